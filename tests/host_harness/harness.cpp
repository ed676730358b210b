// harness.cpp -- TEST INFRASTRUCTURE: compiles the product's device header (rt_device.h) and
// scene packer (rt_pack.h) for the HOST so that the tracer's logic can be compared with the
// oracle pixel by pixel without a GPU (tests/test_host_harness.py, -m "not gpu").
// It is not a product path: nothing in raytracing_opengl_amd/ or librtx_hip.so uses it, and the
// product has no CPU fallback. Built by tests/host_harness/Makefile with g++ -ffp-contract=off.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "rt_device.h"
#include "rt_pack.h"

using namespace rtdev;

extern "C" {

struct harness_texture { int32_t width, height, channels, wrap; const uint8_t* texels; };
struct harness_frame {
    int32_t fb_width, fb_height;
    rtpack::Defines defines;
    const void* blocks[9];
    uint64_t block_sizes[9];
    int32_t sky_size, sky_channels;
    const uint8_t* sky_faces[6];
    harness_texture tex[6];
    int32_t cull;
};

// renders rows [y0,y1) into out (RGBA float, row 0 = bottom); counters[4] = closest, shadow_ref, shadow_cast, torus_solves
// per-pixel counters of the last harness_render call (4 x uint64 per pixel, row-major over the rendered rows), kept only
// when harness_keep_pixel_counters(1) was called: a debugging aid for locating count mismatches
static std::vector<uint64_t> g_pixel_counters;
static int g_keep_pixel_counters = 0;
void harness_keep_pixel_counters(int on) { g_keep_pixel_counters = on; }
const uint64_t* harness_pixel_counters() { return g_pixel_counters.data(); }

int harness_render(const harness_frame* fr, int y0, int y1, float* out, uint64_t* counters)
{
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    for (int b = 0; b < 9; b++) {
        const unsigned char* p = static_cast<const unsigned char*>(fr->blocks[b]);
        if (p && fr->block_sizes[b]) blocks[b].assign(p, p + fr->block_sizes[b]);
    }
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(fr->defines, blocks, blob, err)) return -1;
    // 16-byte aligned copy
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    // pencil masks, built by the product's own per-cell builder (a kernel on the device, a loop here). cull == 2: culls without pencils
    const DevSceneHeader* hdr = reinterpret_cast<const DevSceneHeader*>(aligned.data());
    std::vector<uint32_t> pencil_masks;
    if (fr->cull == 1 && hdr->n_pencil > 0) {
        pencil_masks.assign(hdr->pencil_mask_words, 0u);
        const SceneView S0 = make_view(reinterpret_cast<const char*>(aligned.data()));
        for (uint32_t k = 0; k < hdr->n_pencil + (hdr->pencil_dir != 0xffffffffu ? 1u : 0u); k++) {   // pencils, then the direction table
            const DevPencil P = S0.pencils()[k];
            if (P.kind == RT_PENCIL_OFF) continue;
            std::vector<PencilPrim> prims(hdr->n_surface + hdr->n_torus);
            for (size_t i = 0; i < prims.size(); i++) prims[i] = pencil_prim_at(S0, P, static_cast<int>(i));
#pragma omp parallel for schedule(static, 256)
            for (int64_t cell = 0; cell <= static_cast<int64_t>(P.cells); cell++) {
                const PencilCell C = pencil_cell_geometry(P, static_cast<uint32_t>(cell));
                for (uint32_t w = 0; w < hdr->pencil_stride; w++)
                    pencil_masks[P.mask_off + static_cast<size_t>(cell) * hdr->pencil_stride + w] = pencil_cell_word(S0, P, prims.data(), C, static_cast<uint32_t>(cell), static_cast<int>(w));
            }
        }
    }
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()), hdr, pencil_masks.empty() ? nullptr : pencil_masks.data());

    TexTable T;
    std::memset(&T, 0, sizeof T);
    std::vector<std::vector<uint32_t>> keep;
    for (int s = 0; s < 6; s++) {
        const harness_texture& t = fr->tex[s];
        if (t.width > 0 && t.texels) {
            keep.emplace_back(static_cast<size_t>(t.width) * t.height);
            rtpack::to_rgba8(t.texels, t.width, t.height, t.channels, keep.back().data());
            T.tex[s].texels = keep.back().data();
            T.tex[s].width = t.width; T.tex[s].height = t.height; T.tex[s].wrap = t.wrap; T.tex[s].levels = 1;
            T.tex[s].fwidth = (float)t.width; T.tex[s].fheight = (float)t.height;
        }
    }
    if (fr->sky_size > 0) {
        const size_t fsz = static_cast<size_t>(fr->sky_size) * fr->sky_size;
        keep.emplace_back(fsz * 6, 0u);
        int mask = 0;
        for (int f = 0; f < 6; f++)
            if (fr->sky_faces[f]) { rtpack::to_rgba8(fr->sky_faces[f], fr->sky_size, fr->sky_size, fr->sky_channels, keep.back().data() + fsz * f); mask |= 1 << f; }
        T.sky.texels = keep.back().data();
        T.sky.size = fr->sky_size;
        T.sky.fsize = (float)fr->sky_size;
        T.sky.face_mask = mask;
    }
    uint64_t tot[4] = {0, 0, 0, 0};
    if (g_keep_pixel_counters) g_pixel_counters.assign(static_cast<size_t>(y1 - y0) * fr->fb_width * 4, 0);
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : tot[:4])
    for (int y = y0; y < y1; y++) {
        for (int x = 0; x < fr->fb_width; x++) {
            LaneCounters c = {0, 0, 0, 0};
            f4 px;
            PathStore path;
            if (fr->cull) px = trace_pixel<true, true>(S, T, path, true, (float)x + 0.5f, (float)y + 0.5f, c);
            else px = trace_pixel<false, true>(S, T, path, true, (float)x + 0.5f, (float)y + 0.5f, c);
            float* o = out + (static_cast<size_t>(y - y0) * fr->fb_width + x) * 4;
            o[0] = px.x; o[1] = px.y; o[2] = px.z; o[3] = px.w;
            tot[0] += c.closest; tot[1] += c.shadow_ref; tot[2] += c.shadow_cast; tot[3] += c.torus_solves;
            if (g_keep_pixel_counters) {
                uint64_t* pc = g_pixel_counters.data() + (static_cast<size_t>(y - y0) * fr->fb_width + x) * 4;
                pc[0] = c.closest; pc[1] = c.shadow_ref; pc[2] = c.shadow_cast; pc[3] = c.torus_solves;
            }
        }
    }
    if (counters) std::memcpy(counters, tot, sizeof tot);
    return 0;
}

// The torus culls rest on a premise about the reference's solver ("no root reported for a ray the cull rejects"), which can only be
// checked statistically: n random rays around the torus `record` from distances [dist_lo, dist_hi] (log-uniform), aimed at or near it.
// counts: [0] rays, [1] culled, [2] reference hits, [3] VIOLATIONS (culled but the un-culled solver reports a hit); the first
// violating rays (ro, rd, tmin: 7 floats each, at most max_bad) go to bad.
int harness_torus_premise(const void* record, int64_t n, uint64_t seed, float dist_lo, float dist_hi, int64_t counts[4], float* bad, int max_bad)
{
    rtpack::Defines d;
    std::memset(&d, 0, sizeof d);
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    blocks[rtpack::BLK_SCENE].assign(64, 0);
    d.torus_size = 1;
    const unsigned char* p = static_cast<const unsigned char*>(record);
    blocks[rtpack::BLK_TORUSES].assign(p, p + rtpack::kRecordSize[rtpack::BLK_TORUSES]);
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(d, blocks, blob, err)) return -2;
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()));
    const DevTorus T = S.tori()[0];
    const double ext = std::fabs(T.radii.x) + std::fabs(T.radii.y);
    int64_t c_cull = 0, c_hit = 0, c_bad = 0;
    int n_bad = 0;
#pragma omp parallel for schedule(static, 4096) reduction(+ : c_cull, c_hit, c_bad)
    for (int64_t k = 0; k < n; k++) {
        uint64_t x = seed * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(k) * 0xbf58476d1ce4e5b9ull + 1;
        auto u01 = [&]() { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return (x >> 11) * (1.0 / 9007199254740992.0); };
        auto gauss = [&]() { double a = 0; for (int i = 0; i < 6; i++) a += u01(); return (a - 3.0) * 1.41421356; };
        const double dist = dist_lo * std::pow(static_cast<double>(dist_hi) / dist_lo, u01());
        double dx = gauss(), dy = gauss(), dz = gauss();
        const double dl = std::sqrt(dx * dx + dy * dy + dz * dz) + 1e-30;
        const double ox = T.pos.x + dx / dl * dist, oy = T.pos.y + dy / dl * dist, oz = T.pos.z + dz / dl * dist;
        const double spread = ext * (u01() < 0.6 ? 1.2 : 6.0);
        const double tx = T.pos.x + gauss() * spread, ty = T.pos.y + gauss() * spread, tz = T.pos.z + gauss() * spread;
        double rx = tx - ox, ry = ty - oy, rz = tz - oz;
        const double rl = std::sqrt(rx * rx + ry * ry + rz * rz) + 1e-30;
        const f3 ro = mk3((float)ox, (float)oy, (float)oz);
        f3 rd = mk3((float)(rx / rl), (float)(ry / rl), (float)(rz / rl));
        if (u01() < 0.1) rd = -rd;
        const float tmin = u01() < 0.5 ? 1.0e6f : (float)std::pow(10.0, -1.0 + 5.0 * u01());
        bool solved = false;
        float t = 0.0f, t2 = 0.0f;
        bool cull;
        if (seed >> 63) {
            // the premise of the ray pencils and slab tables: the ray's LINE passes the (padded) bounding sphere by 4e-3 or more, in
            // exact arithmetic -- whatever the distance of the origin (torus_cull asks for a margin that grows with the distance)
            const f4 b = S.torus_bound()[0];
            const double ocx = ox - b.x, ocy = oy - b.y, ocz = oz - b.z, ddx = rd.x, ddy = rd.y, ddz = rd.z;
            const double dl2 = ddx * ddx + ddy * ddy + ddz * ddz, bq = ocx * ddx + ocy * ddy + ocz * ddz;
            const double m2 = ocx * ocx + ocy * ocy + ocz * ocz - bq * bq / dl2, rr = std::sqrt(static_cast<double>(b.w)) + 4e-3;
            cull = std::isfinite(static_cast<double>(b.w)) && m2 > rr * rr && std::fabs(dl2 - 1.0) <= 1e-3;
            (void)t2; (void)solved;
        } else {
            cull = torus_cull(S.torus_bound()[0], ro, rd);
            if (!cull) { intersect_torus_c<true>(T, ro, rd, tmin, t2, solved); cull = !solved; }
        }
        const bool hit = intersect_torus(T, ro, rd, tmin, t);
        c_cull += cull; c_hit += hit;
        if (cull && hit) {
            c_bad++;
#pragma omp critical
            if (n_bad < max_bad) { float* o = bad + 7 * n_bad++; o[0] = ro.x; o[1] = ro.y; o[2] = ro.z; o[3] = rd.x; o[4] = rd.y; o[5] = rd.z; o[6] = tmin; }
        }
    }
    counts[0] = n; counts[1] = c_cull; counts[2] = c_hit; counts[3] = c_bad;
    return n_bad;
}

// Rays that START on the torus -- its own shadow and mirror rays, which the convex-hull cull of torus_local_cull is for: a point of the
// surface, pushed out along the normal by a gap in [gap_lo, gap_hi] (log-uniform; the shader's hit bias is ~1e-3) and, like a hit point
// that comes from a Durand-Kerner root, displaced ALONG the incoming ray by up to +-jitter; directions over the whole outward
// hemisphere (with a share of grazing ones) and a tenth over the inward one. counts / bad as above.
int harness_torus_surface_premise(const void* record, int64_t n, uint64_t seed, float gap_lo, float gap_hi, float jitter, int64_t counts[4], float* bad, int max_bad)
{
    rtpack::Defines d;
    std::memset(&d, 0, sizeof d);
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    blocks[rtpack::BLK_SCENE].assign(64, 0);
    d.torus_size = 1;
    const unsigned char* p = static_cast<const unsigned char*>(record);
    blocks[rtpack::BLK_TORUSES].assign(p, p + rtpack::kRecordSize[rtpack::BLK_TORUSES]);
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(d, blocks, blob, err)) return -2;
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()));
    const DevTorus T = S.tori()[0];
    const double R = T.radii.x, r = T.radii.y;
    int64_t c_cull = 0, c_hit = 0, c_bad = 0;
    int n_bad = 0;
#pragma omp parallel for schedule(static, 4096) reduction(+ : c_cull, c_hit, c_bad)
    for (int64_t k = 0; k < n; k++) {
        uint64_t x = seed * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(k) * 0xbf58476d1ce4e5b9ull + 1;
        auto u01 = [&]() { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return (x >> 11) * (1.0 / 9007199254740992.0); };
        auto gauss = [&]() { double a = 0; for (int i = 0; i < 6; i++) a += u01(); return (a - 3.0) * 1.41421356; };
        const double phi = 6.283185307179586 * u01(), th = 6.283185307179586 * u01();
        const double cp = std::cos(phi), sp = std::sin(phi), ct = std::cos(th), st = std::sin(th);
        const double nx = ct * cp, ny = ct * sp, nz = st;                       // outward normal of the tube (torus frame, axis z)
        const double gap = gap_lo * std::pow(static_cast<double>(gap_hi) / gap_lo, u01());
        // outgoing direction: around the normal, a third of them within a few degrees of the tangent plane
        double dx, dy, dz, dn;
        do { dx = gauss(); dy = gauss(); dz = gauss(); const double l = std::sqrt(dx * dx + dy * dy + dz * dz) + 1e-30; dx /= l; dy /= l; dz /= l; dn = dx * nx + dy * ny + dz * nz; } while (false);
        if (u01() < 0.33) {   // grazing: squash the normal component
            const double f = 0.05 * u01();
            dx -= (1.0 - f) * dn * nx; dy -= (1.0 - f) * dn * ny; dz -= (1.0 - f) * dn * nz;
            const double l = std::sqrt(dx * dx + dy * dy + dz * dz) + 1e-30; dx /= l; dy /= l; dz /= l; dn = dx * nx + dy * ny + dz * nz;
        }
        if ((dn < 0.0) != (u01() < 0.1)) { dx = -dx; dy = -dy; dz = -dz; }
        // the incoming ray the point was "hit" by: the error of its root moves the point along it
        double ix = gauss(), iy = gauss(), iz = gauss();
        const double il = std::sqrt(ix * ix + iy * iy + iz * iz) + 1e-30;
        const double jt = jitter * (2.0 * u01() - 1.0);
        const double lx = (R + r * ct) * cp + nx * gap + ix / il * jt, ly = (R + r * ct) * sp + ny * gap + iy / il * jt, lz = r * st + nz * gap + iz / il * jt;
        const f3 ol = mk3((float)lx, (float)ly, (float)lz), dl = mk3((float)dx, (float)dy, (float)dz);
        const f3 ro = quat_rotate(T.qinv, ol) + xyz(T.pos);
        const f3 rd = quat_rotate(T.qinv, dl);
        const float tmin = u01() < 0.5 ? 1.0e6f : (float)std::pow(10.0, -1.0 + 5.0 * u01());
        bool solved = false;
        float t = 0.0f, t2 = 0.0f;
        bool cull = torus_cull(S.torus_bound()[0], ro, rd);
        if (!cull) { intersect_torus_c<true>(T, ro, rd, tmin, t2, solved); cull = !solved; }
        const bool hit = intersect_torus(T, ro, rd, tmin, t);
        c_cull += cull; c_hit += hit;
        if (cull && hit) {
            c_bad++;
#pragma omp critical
            if (n_bad < max_bad) { float* o = bad + 7 * n_bad++; o[0] = ro.x; o[1] = ro.y; o[2] = ro.z; o[3] = rd.x; o[4] = rd.y; o[5] = rd.z; o[6] = tmin; }
        }
    }
    counts[0] = n; counts[1] = c_cull; counts[2] = c_hit; counts[3] = c_bad;
    return n_bad;
}

// The packed first-level cull record of one quadric: out = bound xyz, radius^2 (negative: none), |p2| margin.
int harness_surface_bound(const void* record, float out[5])
{
    rtpack::Defines d;
    std::memset(&d, 0, sizeof d);
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    blocks[rtpack::BLK_SCENE].assign(64, 0);
    d.surface_size = 1;
    const unsigned char* p = static_cast<const unsigned char*>(record);
    blocks[rtpack::BLK_SURFACES].assign(p, p + rtpack::kRecordSize[rtpack::BLK_SURFACES]);
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(d, blocks, blob, err)) return -2;
    const DevSceneHeader* h = reinterpret_cast<const DevSceneHeader*>(blob.data());
    DevSurfaceCull C;
    std::memcpy(&C, blob.data() + h->off_surf_cull, sizeof C);
    out[0] = C.bound.x; out[1] = C.bound.y; out[2] = C.bound.z; out[3] = C.bound.w; out[4] = C.sym1.z;
    return 0;
}

// The same for a quadric (`record`: one std140 rt_surface): rays around its position from distances [dist_lo, dist_hi].
// extent: the size of the region the rays are aimed at. counts / bad as above.
// counts: [0] rays, [1] culled by surface_cull (the sphere tests), [2] hits, [3] VIOLATIONS (culled by either test and hit), [4] culled by the
// clip-box test behind it
int harness_quadric_premise(const void* record, int64_t n, uint64_t seed, float dist_lo, float dist_hi, float extent, int64_t counts[5], float* bad, int max_bad)
{
    rtpack::Defines d;
    std::memset(&d, 0, sizeof d);
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    blocks[rtpack::BLK_SCENE].assign(64, 0);
    d.surface_size = 1;
    const unsigned char* p = static_cast<const unsigned char*>(record);
    blocks[rtpack::BLK_SURFACES].assign(p, p + rtpack::kRecordSize[rtpack::BLK_SURFACES]);
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(d, blocks, blob, err)) return -2;
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()));
    const DevSurface Q = S.surfaces()[0];
    const DevSurfaceCull C = S.surf_cull()[0];
    int64_t c_cull = 0, c_hit = 0, c_bad = 0, c_box = 0;
    int n_bad = 0;
#pragma omp parallel for schedule(static, 4096) reduction(+ : c_cull, c_hit, c_bad, c_box)
    for (int64_t k = 0; k < n; k++) {
        uint64_t x = seed * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(k) * 0xbf58476d1ce4e5b9ull + 1;
        auto u01 = [&]() { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return (x >> 11) * (1.0 / 9007199254740992.0); };
        auto gauss = [&]() { double a = 0; for (int i = 0; i < 6; i++) a += u01(); return (a - 3.0) * 1.41421356; };
        const double dist = dist_lo * std::pow(static_cast<double>(dist_hi) / dist_lo, u01());
        double dx = gauss(), dy = gauss(), dz = gauss();
        const double dl = std::sqrt(dx * dx + dy * dy + dz * dz) + 1e-30;
        const double ox = Q.pos_a.x + dx / dl * dist, oy = Q.pos_a.y + dy / dl * dist, oz = Q.pos_a.z + dz / dl * dist;
        const double spread = extent * (u01() < 0.6 ? 1.5 : 8.0);
        const double tx = Q.pos_a.x + gauss() * spread, ty = Q.pos_a.y + gauss() * spread, tz = Q.pos_a.z + gauss() * spread;
        double rx = tx - ox, ry = ty - oy, rz = tz - oz;
        const double rl = std::sqrt(rx * rx + ry * ry + rz * rz) + 1e-30;
        const f3 ro = mk3((float)ox, (float)oy, (float)oz);
        f3 rd = mk3((float)(rx / rl), (float)(ry / rl), (float)(rz / rl));
        if (u01() < 0.1) rd = -rd;
        const float tmin = u01() < 0.5 ? 1.0e6f : (float)std::pow(10.0, -1.0 + 5.0 * u01());
        float t = 0.0f;
        bool safe = false;
        const bool sphere = surface_cull(C, ro, rd, tmin, safe);
        const bool box = !sphere && safe && surface_box_miss(Q, ro, rd, tmin);   // the clip-box test behind the sphere, as the candidate scans compose it
        const bool cull = sphere || box;
        const bool hit = intersect_surface(Q, ro, rd, tmin, t);
        c_cull += sphere; c_box += box; c_hit += hit;
        if (cull && hit) {
            c_bad++;
#pragma omp critical
            if (n_bad < max_bad) { float* o = bad + 7 * n_bad++; o[0] = ro.x; o[1] = ro.y; o[2] = ro.z; o[3] = rd.x; o[4] = rd.y; o[5] = rd.z; o[6] = tmin; }
        }
    }
    counts[0] = n; counts[1] = c_cull; counts[2] = c_hit; counts[3] = c_bad; counts[4] = c_box;
    return n_bad;
}

// The product's intersect_surface against the shader's sequence written out (rt.frag:513-572, no early exit) on rays that START on the
// quadric -- the shadow and mirror rays of its own hits, where F(origin) is rounding noise of either sign and one root sits next to the
// `t > 1e-4` test. Origins: hit
// points of rays from outside, as the shader forms them (ro + t rd in float), half of them pushed off by 1e-7 ... 1e-3; directions: any,
// a third of them grazing. counts: [0] rays, [1] hits (literal), [2] rays the product left early although the roots are real (none today), [3] MISMATCHES (hit flag, or t on a hit).
static bool surface_literal(const DevSurface& Q, f3 ro_w, f3 rd_w, float tmin, float& t, bool& real_roots)
{
    const f3 ro = quat_rotate(Q.quat, ro_w - xyz(Q.pos_a));
    const f3 rd = quat_rotate(Q.quat, rd_w);
    const float a = Q.pos_a.w, b = Q.bcde.x, c = Q.bcde.y, d = Q.bcde.z, e = Q.bcde.w, f = Q.f_vmin.x;
    const float d1 = rd.x, d2 = rd.y, d3 = rd.z, o1 = ro.x, o2 = ro.y, o3 = ro.z;
    const float p1 = 2.0f * a * d1 * o1 + 2.0f * b * d2 * o2 + 2.0f * c * d3 * o3 + d * d3 + d2 * e;
    const float p2 = a * d1 * d1 + b * d2 * d2 + c * d3 * d3;
    const float p3 = a * o1 * o1 + b * o2 * o2 + c * o3 * o3 + d * o3 + e * o2 + f;
    real_roots = false;
    if (fabsf(p2) < 1e-6f) { t = -p3 / p1; return t > tmin; }
    const float disc = p1 * p1 - 4.0f * p2 * p3;
    real_roots = !(disc < 0.0f);
    const float p4 = sqrtf(disc);
    float mn = RT_FLT_MAX, mx = RT_FLT_MAX;
    const float t1 = (-p1 - p4) / (2.0f * p2), t2 = (-p1 + p4) / (2.0f * p2);
    const float epsilon = 1e-4f;
    if (t1 > epsilon && t1 < mn) { mn = t1; mx = t2; }
    if (t2 > epsilon && t2 < mn) { mn = t2; mx = t1; }
    const f3 vmin = mk3(Q.f_vmin.y, Q.f_vmin.z, Q.f_vmin.w), vmax = xyz(Q.vmax);
    f3 pt = rd_w * mn + ro_w;
    if (!is_between(pt, vmin, vmax)) {
        if (mx < epsilon) return false;
        pt = rd_w * mx + ro_w;
        if (!is_between(pt, vmin, vmax)) return false;
        const float tmp = mn; mn = mx; mx = tmp;
    }
    t = mn;
    return t < tmin;
}
int harness_quadric_self_rays(const void* record, int64_t n, uint64_t seed, float extent, int64_t counts[4], float* bad, int max_bad)
{
    rtpack::Defines d;
    std::memset(&d, 0, sizeof d);
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    blocks[rtpack::BLK_SCENE].assign(64, 0);
    d.surface_size = 1;
    const unsigned char* p = static_cast<const unsigned char*>(record);
    blocks[rtpack::BLK_SURFACES].assign(p, p + rtpack::kRecordSize[rtpack::BLK_SURFACES]);
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(d, blocks, blob, err)) return -2;
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()));
    const DevSurface Q = S.surfaces()[0];
    int64_t c_rays = 0, c_hit = 0, c_exit = 0, c_bad = 0;
    int n_bad = 0;
#pragma omp parallel for schedule(static, 4096) reduction(+ : c_rays, c_hit, c_exit, c_bad)
    for (int64_t k = 0; k < n; k++) {
        uint64_t x = seed * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(k) * 0xbf58476d1ce4e5b9ull + 1;
        auto u01 = [&]() { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return (x >> 11) * (1.0 / 9007199254740992.0); };
        auto gauss = [&]() { double a = 0; for (int i = 0; i < 6; i++) a += u01(); return (a - 3.0) * 1.41421356; };
        auto unit = [&]() { double ux = gauss(), uy = gauss(), uz = gauss(); const double l = std::sqrt(ux * ux + uy * uy + uz * uz) + 1e-30; return mk3((float)(ux / l), (float)(uy / l), (float)(uz / l)); };
        const float dist = (float)(0.05 * std::pow(200.0 / 0.05, u01()));
        const f3 cen = xyz(Q.pos_a);
        const f3 o0 = cen + unit() * dist;
        f3 aim = cen + mk3((float)gauss(), (float)gauss(), (float)gauss()) * (extent * 0.8f) - o0;
        const f3 d0 = aim * (1.0f / (length3(aim) + 1e-30f));
        float th = 0.0f;
        bool rr = false;
        if (!surface_literal(Q, o0, d0, RT_FLT_MAX, th, rr) || !(th < 1.0e4f)) continue;
        f3 ro = d0 * th + o0;
        if (u01() < 0.5) ro = ro + unit() * (float)(u01() < 0.5 ? 1e-7 * std::pow(1e3, u01()) : 1e-4 * std::pow(10.0, u01()));
        f3 rd = unit();
        if (u01() < 0.3) { const f3 g = rd - d0 * dot3(rd, d0) + d0 * (float)(0.02 * (2.0 * u01() - 1.0)); rd = g * (1.0f / (length3(g) + 1e-30f)); }
        const float tmin = u01() < 0.5 ? RT_FLT_MAX : (float)std::pow(10.0, -3.0 + 4.7 * u01());
        float t_lit = 0.0f, t_prod = 0.0f;
        const bool hit = surface_literal(Q, ro, rd, tmin, t_lit, rr);
        const bool hit_p = intersect_surface(Q, ro, rd, tmin, t_prod);   // host build: every early exit on the ray's own condition
        c_rays++; c_hit += hit;
        const float fmax = RT_FLT_MAX;
        c_exit += !hit_p && rr && std::memcmp(&t_prod, &fmax, 4) == 0;
        if (hit != hit_p || (hit && std::memcmp(&t_lit, &t_prod, 4) != 0)) {
            c_bad++;
#pragma omp critical
            if (n_bad < max_bad) { float* o = bad + 7 * n_bad++; o[0] = ro.x; o[1] = ro.y; o[2] = ro.z; o[3] = rd.x; o[4] = rd.y; o[5] = rd.z; o[6] = tmin; }
        }
    }
    counts[0] = c_rays; counts[1] = c_hit; counts[2] = c_exit; counts[3] = c_bad;
    return n_bad;
}

// The candidate tables (ray pencils, slab tables) primitive by primitive: n random rays of three kinds -- from the camera (pencil 0), from
// random points towards every light with a pencil, and arbitrary rays (slab tables) -- and for each ray every quadric and torus:
// if the un-culled intersector reports a hit, the primitive's bit must be set in the ray's candidate mask.
// counts: [0] rays, [1] hits, [2] VIOLATIONS (hit, bit clear), [3] set bits summed over the rays, [4] rays that read the all-ones cell.
int harness_table_premise(const harness_frame* fr, int64_t n, uint64_t seed, int64_t counts[5])
{
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    for (int b = 0; b < 9; b++) {
        const unsigned char* p = static_cast<const unsigned char*>(fr->blocks[b]);
        if (p && fr->block_sizes[b]) blocks[b].assign(p, p + fr->block_sizes[b]);
    }
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(fr->defines, blocks, blob, err)) return -1;
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    const DevSceneHeader* hdr = reinterpret_cast<const DevSceneHeader*>(aligned.data());
    if (hdr->n_pencil == 0) return -2;
    std::vector<uint32_t> masks(hdr->pencil_mask_words, 0u);
    const SceneView S0 = make_view(reinterpret_cast<const char*>(aligned.data()));
    for (uint32_t k = 0; k < hdr->n_pencil + (hdr->pencil_dir != 0xffffffffu ? 1u : 0u); k++) {
        const DevPencil P = S0.pencils()[k];
        if (P.kind == RT_PENCIL_OFF) continue;
        std::vector<PencilPrim> prims(hdr->n_surface + hdr->n_torus);
        for (size_t i = 0; i < prims.size(); i++) prims[i] = pencil_prim_at(S0, P, static_cast<int>(i));
#pragma omp parallel for schedule(static, 256)
        for (int64_t cell = 0; cell <= static_cast<int64_t>(P.cells); cell++) {
            const PencilCell C = pencil_cell_geometry(P, static_cast<uint32_t>(cell));
            for (uint32_t w = 0; w < hdr->pencil_stride; w++)
                masks[P.mask_off + static_cast<size_t>(cell) * hdr->pencil_stride + w] = pencil_cell_word(S0, P, prims.data(), C, static_cast<uint32_t>(cell), static_cast<int>(w));
        }
    }
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()), hdr, masks.data());
    const int ns = hdr->n_surface, nt = hdr->n_torus, nws = (ns + 31) >> 5, W = static_cast<int>(hdr->pencil_stride);
    const int n_lights = hdr->n_light_point + hdr->n_light_direct;
    int64_t c_hit = 0, c_bad = 0, c_bits = 0, c_all = 0;
#pragma omp parallel for schedule(static, 4096) reduction(+ : c_hit, c_bad, c_bits, c_all)
    for (int64_t k = 0; k < n; k++) {
        uint64_t x = seed * 0x9e3779b97f4a7c15ull + static_cast<uint64_t>(k) * 0xbf58476d1ce4e5b9ull + 1;
        auto u01 = [&]() { x ^= x >> 30; x *= 0xbf58476d1ce4e5b9ull; x ^= x >> 27; x *= 0x94d049bb133111ebull; x ^= x >> 31; return (x >> 11) * (1.0 / 9007199254740992.0); };
        auto gauss = [&]() { double a = 0; for (int i = 0; i < 6; i++) a += u01(); return (a - 3.0) * 1.41421356; };
        // a point in or around the crowd (tests/random_scenes.py: x, y within +-15, z 8 .. 24), sometimes far out
        auto point = [&]() { const double far = u01() < 0.1 ? 40.0 : 1.0; return mk3((float)(gauss() * 7.0 * far), (float)(gauss() * 6.0 * far), (float)(16.0 + gauss() * 5.0 * far)); };
        const int kind = static_cast<int>(k % 3);
        f3 ro, rd;
        float tlimit = RT_MAXDIST;
        uint32_t words[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool have = false;
        if (kind == 0) {                         // camera ray
            ro = xyz(S.h->cam_pos);
            rd = normalize3(point() - ro);
            const PencilScan ps = pencil_open<true>(S, 0, ro, rd, 0.0f, true);
            if (ps.use) { have = true; for (int w = 0; w < W; w++) words[w] = S.pen[ps.cell + w]; }
        } else if (kind == 1 && n_lights > 0) {  // shadow ray, built like calc_shade builds it
            const int li = static_cast<int>(u01() * n_lights) % n_lights;
            ro = point();
            if (li < hdr->n_light_point) {
                const f3 ld = xyz(S.lights_point()[li].pos_r2) - ro;
                tlimit = length3(ld);
                rd = normalize3(ld);
            } else {
                rd = xyz(S.lights_direct()[li - hdr->n_light_point].dir_n);
            }
            const PencilScan ps = pencil_open<true>(S, 1 + li, ro, rd, tlimit, false);
            if (ps.use) { have = true; for (int w = 0; w < W; w++) words[w] = S.pen[ps.cell + w]; }
        }
        if (!have) {                             // any ray: slab tables
            if (!slabs_available(S)) continue;
            ro = point();
            rd = normalize3(point() - ro);
            tlimit = u01() < 0.5 ? RT_MAXDIST : (float)(1.0 + 40.0 * u01());
            slab_ray_mask(S, ro, rd, tlimit, words);
            have = true;
        }
        int bits = 0;
        for (int w = 0; w < W; w++) bits += __builtin_popcount(words[w]);
        c_bits += bits;
        c_all += bits == ns + nt;
        for (int i = 0; i < ns + nt; i++) {
            float t = 0.0f;
            const bool hit = i < ns ? intersect_surface(S.surfaces()[i], ro, rd, tlimit, t) : intersect_torus(S.tori()[i - ns], ro, rd, tlimit, t);
            if (!hit) continue;
            c_hit++;
            const int w = i < ns ? i >> 5 : nws + ((i - ns) >> 5), b = (i < ns ? i : i - ns) & 31;
            if (!((words[w] >> b) & 1u)) c_bad++;
        }
    }
    counts[0] = n; counts[1] = c_hit; counts[2] = c_bad; counts[3] = c_bits; counts[4] = c_all;
    return 0;
}

// Single rays through the product's own scans on the host (tests/test_culls.py; tools/audit/cull_audit.hip probe_kernel is the same on the
// device): per ray (ro, rd, limit, torus index) out[0..1] the literal torus intersector: hit, t; out[2..3] the product's composition for that
// torus (torus_cull, group sphere, intersect_torus_c<true>): hit, t; out[4..5] in_shadow over the whole scene with `limit` as the distance
// to the light, culls (incl. slab tables) on / off; out[6..8] calc_inter, culls on: t, num, type; out[9..11] culls off.
int harness_probe(const harness_frame* fr, const float* rays, int n, float* out)
{
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    for (int b = 0; b < 9; b++) {
        const unsigned char* p = static_cast<const unsigned char*>(fr->blocks[b]);
        if (p && fr->block_sizes[b]) blocks[b].assign(p, p + fr->block_sizes[b]);
    }
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(fr->defines, blocks, blob, err)) return -1;
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    const DevSceneHeader* hdr = reinterpret_cast<const DevSceneHeader*>(aligned.data());
    std::vector<uint32_t> masks(hdr->pencil_mask_words, 0u);
    const SceneView S0 = make_view(reinterpret_cast<const char*>(aligned.data()));
    for (uint32_t k = 0; hdr->n_pencil > 0 && k < hdr->n_pencil + (hdr->pencil_dir != 0xffffffffu ? 1u : 0u); k++) {
        const DevPencil P = S0.pencils()[k];
        if (P.kind == RT_PENCIL_OFF) continue;
        std::vector<PencilPrim> prims(hdr->n_surface + hdr->n_torus);
        for (size_t i = 0; i < prims.size(); i++) prims[i] = pencil_prim_at(S0, P, static_cast<int>(i));
#pragma omp parallel for schedule(static, 256)
        for (int64_t cell = 0; cell <= static_cast<int64_t>(P.cells); cell++) {
            const PencilCell C = pencil_cell_geometry(P, static_cast<uint32_t>(cell));
            for (uint32_t w = 0; w < hdr->pencil_stride; w++)
                masks[P.mask_off + static_cast<size_t>(cell) * hdr->pencil_stride + w] = pencil_cell_word(S0, P, prims.data(), C, static_cast<uint32_t>(cell), static_cast<int>(w));
        }
    }
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()), hdr, masks.empty() ? nullptr : masks.data());
    TexTable TT;
    std::memset(&TT, 0, sizeof TT);
    const int nt = hdr->n_torus;
    for (int k = 0; k < n; k++) {
        const float* r = rays + static_cast<size_t>(k) * 8;
        const f3 ro = mk3(r[0], r[1], r[2]), rd = mk3(r[3], r[4], r[5]);
        const float limit = r[6];
        int i = static_cast<int>(r[7]);
        i = i < 0 ? 0 : (i >= nt ? nt - 1 : i);
        float* o = out + static_cast<size_t>(k) * 12;
        for (int j = 0; j < 12; j++) o[j] = 0.0f;
        if (nt > 0) {
            const DevTorus T = S.tori()[i];
            float t = 0.0f;
            o[0] = intersect_torus(T, ro, rd, limit, t) ? 1.0f : 0.0f; o[1] = t;
            bool culled = torus_cull(S.torus_bound()[i], ro, rd);
            if (nt >= RT_GROUP_MIN) culled = culled || torus_group_cull(S.torus_group()[i / RT_GROUP], ro, rd);
            bool solved = false;
            float t2 = 0.0f;
            const bool h2 = !culled && intersect_torus_c<true, true>(T, ro, rd, limit, t2, solved);
            o[2] = h2 ? 1.0f : 0.0f; o[3] = h2 ? t2 : 0.0f;
        }
        LaneCounters cnt;
        std::memset(&cnt, 0, sizeof cnt);
        o[4] = in_shadow<true, false, true>(S, TT, true, true, ro, rd, limit, cnt, -1);
        o[5] = in_shadow<false, false, true>(S, TT, true, true, ro, rd, limit, cnt, -1);
        int num = -1, type = -1;
        o[6] = calc_inter<true, false, true>(S, ro, rd, num, type, cnt, -1); o[7] = (float)num; o[8] = (float)type;
        num = -1; type = -1;
        o[9] = calc_inter<false, false, true>(S, ro, rd, num, type, cnt, -1); o[10] = (float)num; o[11] = (float)type;
    }
    return 0;
}

// Pencil diagnostics for the tests: out[0] = pencils, out[1] = mask words per cell, then per pencil (kind, cells, mean set bits per cell).
int harness_pencil_stats(const harness_frame* fr, double* out, int max_out)
{
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    for (int b = 0; b < 9; b++) {
        const unsigned char* p = static_cast<const unsigned char*>(fr->blocks[b]);
        if (p && fr->block_sizes[b]) blocks[b].assign(p, p + fr->block_sizes[b]);
    }
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(fr->defines, blocks, blob, err)) return -1;
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()));
    int n = 0;
    if (max_out < 2) return -1;
    out[n++] = S.h->n_pencil;
    out[n++] = S.h->pencil_stride;
    std::vector<uint32_t> cellw(S.h->pencil_stride + 1);
    for (uint32_t k = 0; k < S.h->n_pencil + (S.h->pencil_dir != 0xffffffffu ? 1u : 0u) && n + 3 <= max_out; k++) {
        const DevPencil P = S.pencils()[k];
        double bits = 0.0;
        std::vector<PencilPrim> prims(S.h->n_surface + S.h->n_torus);
        for (size_t i = 0; i < prims.size(); i++) prims[i] = pencil_prim_at(S, P, static_cast<int>(i));
        if (P.kind != RT_PENCIL_OFF)
            for (uint32_t cell = 0; cell < P.cells; cell++) {
                const PencilCell C = pencil_cell_geometry(P, cell);
                for (uint32_t w = 0; w < S.h->pencil_stride; w++) bits += __builtin_popcount(pencil_cell_word(S, P, prims.data(), C, cell, static_cast<int>(w)));
            }
        out[n++] = P.kind; out[n++] = P.cells; out[n++] = P.cells ? bits / P.cells : 0.0;
    }
    return n;
}

// Single-primitive entry points on the DEVICE functions (packed through rt_pack.h like the product does).
// type: PrimType; record: one std140 record of that type. out[0]=hit, out[1]=t, out[2]=cull decision (1 = skipped).
int harness_kat(int type, const void* record, const float ro[3], const float rd[3], float tmin, float out[3])
{
    rtpack::Defines d;
    std::memset(&d, 0, sizeof d);
    std::vector<unsigned char> blocks[rtpack::BLK_COUNT];
    blocks[rtpack::BLK_SCENE].assign(64, 0);
    const unsigned char* p = static_cast<const unsigned char*>(record);
    int blk = -1;
    if (type == TYPE_SURFACE) { d.surface_size = 1; blk = rtpack::BLK_SURFACES; }
    if (type == TYPE_BOX) { d.box_size = 1; blk = rtpack::BLK_BOXES; }
    if (type == TYPE_TORUS) { d.torus_size = 1; blk = rtpack::BLK_TORUSES; }
    if (type == TYPE_RING) { d.ring_size = 1; blk = rtpack::BLK_RINGS; }
    if (blk < 0) return -1;
    blocks[blk].assign(p, p + rtpack::kRecordSize[blk]);
    std::vector<unsigned char> blob;
    std::string err;
    if (!rtpack::pack_scene(d, blocks, blob, err)) return -2;
    std::vector<f4> aligned((blob.size() + 15) / 16);
    std::memcpy(aligned.data(), blob.data(), blob.size());
    const SceneView S = make_view(reinterpret_cast<const char*>(aligned.data()));
    const f3 o = mk3(ro[0], ro[1], ro[2]), dd = mk3(rd[0], rd[1], rd[2]);
    float t = 0.0f;
    bool hit = false, cull = false;
    f3 nor = mk3(0, 0, 0);
    f2 uv = mk2(0, 0);
    if (type == TYPE_SURFACE) { cull = surface_cull(S.surf_cull()[0], o, dd, tmin); hit = intersect_surface(S.surfaces()[0], o, dd, tmin, t); }
    if (type == TYPE_BOX) { RayBoxCtx bctx; hit = intersect_box(S.boxes()[0], o, dd, tmin, t, nor, bctx); }
    if (type == TYPE_TORUS) { bool solved; cull = torus_cull(S.torus_bound()[0], o, dd); if (!cull) { float t2; intersect_torus_c<true>(S.tori()[0], o, dd, tmin, t2, solved); cull = !solved; } hit = intersect_torus(S.tori()[0], o, dd, tmin, t); }
    if (type == TYPE_RING) { cull = ring_cull(S.ring_bound()[0], o, dd, tmin); hit = intersect_ring(S.rings()[0], o, dd, tmin, t, uv); }
    out[0] = hit ? 1.0f : 0.0f;
    out[1] = t;
    out[2] = cull ? 1.0f : 0.0f;
    return 0;
}

// quat_rotate_id's shortcut: for every identity quaternion (all 8 zero-sign patterns) and finite v the full
// rotation formula must give v + 0.0f bit for bit; with a non-finite component the shortcut must not be taken.
// Returns the number of mismatches over special values (signed zeros, denormals, huge) and n random bit patterns.
int harness_identity_rotation_mismatches(int n_random, unsigned seed)
{
    auto bits = [](float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; };
    auto same = [&](f3 a, f3 b) {
        auto eq = [&](float x, float y) { return bits(x) == bits(y) || (x != x && y != y); };
        return eq(a.x, b.x) && eq(a.y, b.y) && eq(a.z, b.z);
    };
    const float inf = __builtin_inff(), nan = __builtin_nanf("");
    const float vals[] = {-1.0f, -0.0f, 0.0f, 1.0f, 0.3f, -2.5f, 1e-42f, -1e-42f, 3e38f, -3e38f, 1e-30f, inf, -inf, nan};
    int bad = 0;
    uint32_t st = seed ? seed : 1u;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 17; st ^= st << 5; return st; };
    for (int sgn = 0; sgn < 8; sgn++) {
        const f4 q = mk4((sgn & 1) ? -0.0f : 0.0f, (sgn & 2) ? -0.0f : 0.0f, (sgn & 4) ? -0.0f : 0.0f, 1.0f);
        if (!quat_is_identity(q)) bad++;
        for (float x : vals) for (float y : vals) for (float z : vals) {
            const f3 v = mk3(x, y, z);
            if (!same(quat_rotate_id(q, true, v), quat_rotate(q, v))) bad++;
        }
        for (int k = 0; k < n_random; k++) {
            float f[3];
            for (int j = 0; j < 3; j++) {
                const uint32_t u = rnd();
                std::memcpy(&f[j], &u, 4);
                if ((rnd() % 7u) == 0u) f[j] = (rnd() & 1u) ? -0.0f : 0.0f;
            }
            const f3 v = mk3(f[0], f[1], f[2]);
            if (!same(quat_rotate_id(q, true, v), quat_rotate(q, v))) bad++;
        }
    }
    return bad;
}

// exhaustive check of the divide-free unorm8 against byte/255.0f
int harness_unorm8_mismatches()
{
    int bad = 0;
    for (uint32_t b = 0; b < 256; b++) {
        volatile float ref = (float)b / 255.0f;
        if (unorm8(b) != ref) bad++;
    }
    return bad;
}

}  // extern "C"
