// rt_pack.h -- host-side packer: nine std140 uniform blocks (+ rt_defines) -> DevScene blob.
//
// Runs inside rtx_block_create / rtx_block_update / rtx_specialize (the reference's
// init_buffer / update_buffer / init_shaders, GLWrapper.cpp:232-277,365-386). Pure C++, no HIP.
// Derived per-primitive fields are computed with the tracer's own inline functions from
// rt_device.h compiled for the host with -ffp-contract=off, i.e. the same IEEE operations the
// kernel would otherwise repeat per ray.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "rt_device.h"

namespace rtpack {
using namespace rtdev;

// std140 record sizes (include/rtx/scene.h static_asserts)
enum { SZ_SCENE = 64, SZ_SPHERE = 112, SZ_PLANE = 96, SZ_SURFACE = 160, SZ_BOX = 112, SZ_TORUS = 112, SZ_RING = 112, SZ_LIGHT_POINT = 48, SZ_LIGHT_DIRECT = 32 };
enum { BLK_SCENE = 0, BLK_SPHERES, BLK_PLANES, BLK_SURFACES, BLK_BOXES, BLK_TORUSES, BLK_RINGS, BLK_LIGHTS_POINT, BLK_LIGHTS_DIRECT, BLK_COUNT };
static const char* const kBlockNames[BLK_COUNT] = {"scene_buf", "spheres_buf", "planes_buf", "surfaces_buf", "boxes_buf",
                                                   "toruses_buf", "rings_buf", "lights_point_buf", "lights_direct_buf"};
static const int kRecordSize[BLK_COUNT] = {SZ_SCENE, SZ_SPHERE, SZ_PLANE, SZ_SURFACE, SZ_BOX, SZ_TORUS, SZ_RING, SZ_LIGHT_POINT, SZ_LIGHT_DIRECT};

struct Defines {  // == rtx_defines / reference rt_defines
    int32_t sphere_size, plane_size, surface_size, box_size, torus_size, ring_size, light_point_size, light_direct_size, iterations;
    float ambient_color[3];
    float shadow_ambient[3];
};

inline float rdf(const unsigned char* p, size_t off) { float v; std::memcpy(&v, p + off, 4); return v; }
inline int32_t rdi(const unsigned char* p, size_t off) { int32_t v; std::memcpy(&v, p + off, 4); return v; }
inline f4 rd4(const unsigned char* p, size_t off) { return mk4(rdf(p, off), rdf(p, off + 4), rdf(p, off + 8), rdf(p, off + 12)); }
inline f4 rd3(const unsigned char* p, size_t off, float w) { return mk4(rdf(p, off), rdf(p, off + 4), rdf(p, off + 8), w); }
inline float int_bits(int32_t v) { float f; std::memcpy(&f, &v, 4); return f; }

// std::to_string(float) == "%f", then parsed back as a GLSL float literal
// (GLWrapper.cpp:246-247,279-282; trap T9)
inline float text_round_trip(float v)
{
    char buf[64];
    std::snprintf(buf, sizeof buf, "%f", static_cast<double>(v));
    return std::strtof(buf, nullptr);
}

inline size_t align16(size_t n) { return (n + 15) & ~static_cast<size_t>(15); }

// blocks[b] may be shorter than count*record (or empty): returns false and sets err.
inline bool pack_scene(const Defines& d, const std::vector<unsigned char> blocks[BLK_COUNT], std::vector<unsigned char>& blob, std::string& err)
{
    const int counts[BLK_COUNT] = {1, d.sphere_size, d.plane_size, d.surface_size, d.box_size, d.torus_size, d.ring_size,
                                   d.light_point_size, d.light_direct_size};
    for (int b = 0; b < BLK_COUNT; b++) {
        if (counts[b] < 0) { err = std::string("negative count for ") + kBlockNames[b]; return false; }
        if (blocks[b].size() < static_cast<size_t>(counts[b]) * kRecordSize[b]) {
            err = std::string("block ") + kBlockNames[b] + " holds fewer bytes than rt_defines announces";
            return false;
        }
    }
    DevSceneHeader h;
    std::memset(&h, 0, sizeof h);
    h.n_sphere = d.sphere_size; h.n_plane = d.plane_size; h.n_surface = d.surface_size; h.n_box = d.box_size;
    h.n_torus = d.torus_size; h.n_ring = d.ring_size; h.n_light_point = d.light_point_size; h.n_light_direct = d.light_direct_size;
    h.iterations = d.iterations;
    const unsigned char* sc = blocks[BLK_SCENE].data();
    h.cam_quat = rd4(sc, 0);
    h.cam_pos = rd3(sc, 16, 0.0f);
    h.canvas_w = rdi(sc, 44);
    h.canvas_h = rdi(sc, 48);
    h.ambient = mk4(text_round_trip(d.ambient_color[0]), text_round_trip(d.ambient_color[1]), text_round_trip(d.ambient_color[2]), 0.0f);
    h.shadow_ambient = mk4(text_round_trip(d.shadow_ambient[0]), text_round_trip(d.shadow_ambient[1]), text_round_trip(d.shadow_ambient[2]), 0.0f);

    size_t off = align16(sizeof(DevSceneHeader));
    auto reserve = [&](size_t bytes) { size_t o = off; off = align16(off + bytes); return static_cast<uint32_t>(o); };
    h.off_sphere = reserve(sizeof(DevSphere) * d.sphere_size);
    h.off_plane = reserve(sizeof(DevPlane) * d.plane_size);
    h.off_surface = reserve(sizeof(DevSurface) * d.surface_size);
    h.off_box = reserve(sizeof(DevBox) * d.box_size);
    h.off_torus = reserve(sizeof(DevTorus) * d.torus_size);
    h.off_ring = reserve(sizeof(DevRing) * d.ring_size);
    h.off_light_point = reserve(sizeof(DevLightPoint) * d.light_point_size);
    h.off_light_direct = reserve(sizeof(DevLightDirect) * d.light_direct_size);
    const int mat_counts[6] = {d.sphere_size, d.plane_size, d.surface_size, d.box_size, d.torus_size, d.ring_size};
    for (int t = 0; t < 6; t++) h.off_mat[t] = reserve(sizeof(DevMaterial) * mat_counts[t]);
    h.total_bytes = static_cast<int32_t>(off);
    blob.assign(off, 0);
    std::memcpy(blob.data(), &h, sizeof h);

    auto mat_at = [&](int type, int i) { return reinterpret_cast<DevMaterial*>(blob.data() + h.off_mat[type]) + i; };
    static_assert(sizeof(DevMaterial) == 64, "material record");

    for (int i = 0; i < d.sphere_size; i++) {
        const unsigned char* p = blocks[BLK_SPHERES].data() + static_cast<size_t>(i) * SZ_SPHERE;
        DevSphere s;
        std::memset(&s, 0, sizeof s);
        const f4 obj = rd4(p, 64);
        s.geom = mk4(obj.x, obj.y, obj.z, obj.w * obj.w);
        s.radius = obj.w;
        s.quat = rd4(p, 80);
        s.texture = rdi(p, 96);
        s.hollow = rdi(p, 100) != 0;  // std140 bool = 4 bytes; the host writes 0/1 + zero padding
        std::memcpy(reinterpret_cast<DevSphere*>(blob.data() + h.off_sphere) + i, &s, sizeof s);
        std::memcpy(mat_at(TYPE_SPHERE, i), p, 64);
    }
    for (int i = 0; i < d.plane_size; i++) {
        const unsigned char* p = blocks[BLK_PLANES].data() + static_cast<size_t>(i) * SZ_PLANE;
        DevPlane s;
        s.pos = rd3(p, 64, 0.0f);
        s.normal = rd3(p, 80, 0.0f);
        std::memcpy(reinterpret_cast<DevPlane*>(blob.data() + h.off_plane) + i, &s, sizeof s);
        std::memcpy(mat_at(TYPE_PLANE, i), p, 64);
    }
    for (int i = 0; i < d.surface_size; i++) {
        const unsigned char* p = blocks[BLK_SURFACES].data() + static_cast<size_t>(i) * SZ_SURFACE;
        DevSurface s;
        std::memset(&s, 0, sizeof s);
        s.quat = rd4(p, 64);
        const f3 vmin = mk3(rdf(p, 80), rdf(p, 84), rdf(p, 88)), vmax = mk3(rdf(p, 96), rdf(p, 100), rdf(p, 104));
        const float a = rdf(p, 124), b = rdf(p, 128), c = rdf(p, 132), dd = rdf(p, 136), e = rdf(p, 140), f = rdf(p, 144);
        s.pos_a = mk4(rdf(p, 112), rdf(p, 116), rdf(p, 120), a);
        s.bcde = mk4(b, c, dd, e);
        s.f_vmin = mk4(f, vmin.x, vmin.y, vmin.z);
        s.vmax = mk4(vmax.x, vmax.y, vmax.z, 0.0f);
        s.qinv = quat_inv(s.quat);
        // cull data (surface_cull in rt_device.h)
        const float big = 1.0e30f;
        const bool finite_box = std::fabs(vmin.x) < big && std::fabs(vmin.y) < big && std::fabs(vmin.z) < big && std::fabs(vmax.x) < big &&
                                std::fabs(vmax.y) < big && std::fabs(vmax.z) < big;
        if (finite_box) {
            const double cx = 0.5 * (static_cast<double>(vmin.x) + vmax.x), cy = 0.5 * (static_cast<double>(vmin.y) + vmax.y),
                         cz = 0.5 * (static_cast<double>(vmin.z) + vmax.z);
            const double hx = 0.5 * (static_cast<double>(vmax.x) - vmin.x), hy = 0.5 * (static_cast<double>(vmax.y) - vmin.y),
                         hz = 0.5 * (static_cast<double>(vmax.z) - vmin.z);
            const double rad = std::sqrt(hx * hx + hy * hy + hz * hz) * 1.01 + 0.01;
            s.bound = mk4(static_cast<float>(cx), static_cast<float>(cy), static_cast<float>(cz), static_cast<float>(rad * rad));
            // columns of the world->local rotation, then M = R^T diag(a,b,c) R in double
            const f3 ex = quat_rotate(s.quat, mk3(1, 0, 0)), ey = quat_rotate(s.quat, mk3(0, 1, 0)), ez = quat_rotate(s.quat, mk3(0, 0, 1));
            const double Rm[3][3] = {{ex.x, ey.x, ez.x}, {ex.y, ey.y, ez.y}, {ex.z, ey.z, ez.z}};  // Rm[k][j]: local k <- world j
            const double dg[3] = {a, b, c};
            double M[3][3];
            for (int r = 0; r < 3; r++)
                for (int q = 0; q < 3; q++) {
                    double acc = 0.0;
                    for (int k = 0; k < 3; k++) acc += Rm[k][r] * dg[k] * Rm[k][q];
                    M[r][q] = acc;
                }
            s.sym0 = mk4(static_cast<float>(M[0][0]), static_cast<float>(M[0][1]), static_cast<float>(M[0][2]), static_cast<float>(M[1][1]));
            const float margin = 1e-6f + 1e-5f * (std::fabs(a) + std::fabs(b) + std::fabs(c));
            s.sym1 = mk4(static_cast<float>(M[1][2]), static_cast<float>(M[2][2]), margin, 0.0f);
        } else {
            s.bound = mk4(0.0f, 0.0f, 0.0f, -1.0f);
        }
        std::memcpy(reinterpret_cast<DevSurface*>(blob.data() + h.off_surface) + i, &s, sizeof s);
        std::memcpy(mat_at(TYPE_SURFACE, i), p, 64);
    }
    for (int i = 0; i < d.box_size; i++) {
        const unsigned char* p = blocks[BLK_BOXES].data() + static_cast<size_t>(i) * SZ_BOX;
        DevBox s;
        s.quat = rd4(p, 64);
        s.pos = rd3(p, 80, 0.0f);
        s.form_tex = rd3(p, 96, int_bits(rdi(p, 108)));
        s.qinv = quat_inv(s.quat);
        std::memcpy(reinterpret_cast<DevBox*>(blob.data() + h.off_box) + i, &s, sizeof s);
        std::memcpy(mat_at(TYPE_BOX, i), p, 64);
    }
    for (int i = 0; i < d.torus_size; i++) {
        const unsigned char* p = blocks[BLK_TORUSES].data() + static_cast<size_t>(i) * SZ_TORUS;
        DevTorus s;
        s.quat = rd4(p, 64);
        s.pos = rd3(p, 80, 0.0f);
        const float R = rdf(p, 96), r = rdf(p, 100);
        const float R2 = R * R, r2 = r * r;
        s.radii = mk4(R, r, R2, r2);
        const double rb = (std::fabs(static_cast<double>(R)) + std::fabs(static_cast<double>(r))) * 1.01 + 0.01;
        const double rf = (100.0 + rb) * 1.001;
        s.k = mk4(4.0f * R2, static_cast<float>(rb * rb), static_cast<float>(rf * rf), 0.0f);
        s.qinv = quat_inv(s.quat);
        std::memcpy(reinterpret_cast<DevTorus*>(blob.data() + h.off_torus) + i, &s, sizeof s);
        std::memcpy(mat_at(TYPE_TORUS, i), p, 64);
    }
    for (int i = 0; i < d.ring_size; i++) {
        const unsigned char* p = blocks[BLK_RINGS].data() + static_cast<size_t>(i) * SZ_RING;
        DevRing s;
        s.quat = rd4(p, 64);
        s.pos_tex = rd3(p, 80, int_bits(rdi(p, 92)));
        const float r1 = rdf(p, 96), r2 = rdf(p, 100);
        s.radii = mk4(r1, r2, r2 - r1, 0.0f);
        const f3 nrm = quat_rotate(quat_inv(s.quat), mk3(0.0f, 0.0f, -1.0f));
        s.normal = mk4(nrm.x, nrm.y, nrm.z, 0.0f);
        std::memcpy(reinterpret_cast<DevRing*>(blob.data() + h.off_ring) + i, &s, sizeof s);
        std::memcpy(mat_at(TYPE_RING, i), p, 64);
    }
    for (int i = 0; i < d.light_point_size; i++) {
        const unsigned char* p = blocks[BLK_LIGHTS_POINT].data() + static_cast<size_t>(i) * SZ_LIGHT_POINT;
        DevLightPoint s;
        const f4 pos = rd4(p, 0);
        s.pos_r2 = mk4(pos.x, pos.y, pos.z, pos.w * pos.w);
        s.color_intensity = rd4(p, 16);  // color xyz @16, intensity @28
        s.atten = mk4(rdf(p, 32), rdf(p, 36), pos.w, 0.0f);
        std::memcpy(reinterpret_cast<DevLightPoint*>(blob.data() + h.off_light_point) + i, &s, sizeof s);
    }
    for (int i = 0; i < d.light_direct_size; i++) {
        const unsigned char* p = blocks[BLK_LIGHTS_DIRECT].data() + static_cast<size_t>(i) * SZ_LIGHT_DIRECT;
        DevLightDirect s;
        s.direction = rd3(p, 0, 0.0f);
        s.color_intensity = rd4(p, 16);
        std::memcpy(reinterpret_cast<DevLightDirect*>(blob.data() + h.off_light_direct) + i, &s, sizeof s);
    }
    return true;
}

// 8-bit interleaved texels (1/3/4 channels) -> RGBA8 dwords (little endian: R in bits 0..7)
inline void to_rgba8(const unsigned char* src, int w, int h, int channels, uint32_t* dst)
{
    const size_t n = static_cast<size_t>(w) * h;
    for (size_t i = 0; i < n; i++) {
        uint32_t r, g, b, a;
        if (channels == 4) { r = src[4 * i]; g = src[4 * i + 1]; b = src[4 * i + 2]; a = src[4 * i + 3]; }
        else if (channels == 3) { r = src[3 * i]; g = src[3 * i + 1]; b = src[3 * i + 2]; a = 255; }
        else { r = src[i]; g = 0; b = 0; a = 255; }
        dst[i] = r | (g << 8) | (b << 16) | (a << 24);
    }
}

}  // namespace rtpack
