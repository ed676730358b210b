// cull_audit.hip -- TOOL, not product: audits the culls of the tracer (csrc/rt_device.h) at a scale the host tests cannot reach.
//
// Every cull of the tracer is a conservative predicate in front of an intersector whose arithmetic is the reference's (rt.frag:342-572).
// "Conservative" has a proof only where the intersector is geometric; for the torus it rests on a premise about the reference's
// Durand-Kerner iteration (rt.frag:462-487: "no root is reported for a ray that misses the inflated torus"), for open quadrics on a bound of
// what the reference's FLOAT evaluation can take for the surface. This tool draws random and near-boundary rays per primitive record on the
// GPU, evaluates the product's cull AND the literal intersector in the same lane and counts "culled and hit" -- 1e10 rays per family are a
// few minutes here against hours on the host (tests/host_harness has the same checks at 1e6-1e8). It compiles the product's own headers
// (rt_device.h, rt_pack.h): what is audited is the code that ships. The literal intersectors are the product's un-culled entry points
// (intersect_torus, intersect_ring) and, for quadrics, a restatement of rt.frag:513-572 without the product's wave-level early exit.
//
// Built by tools/audit/Makefile into tools/audit/libcull_audit.so, driven by tools/cull_audit.py. Nothing under raytracing_opengl_amd/ uses it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_pack.h"

using namespace rtdev;

namespace {

enum { F_TORUS = 0, F_TORUS_MARGIN = 1, F_QUADRIC = 2, F_RING = 3, F_TABLES = 4, F_TORUS_LEAD = 5, F_TORUS_FAR = 6, F_TORUS_BEHIND = 7, F_TORUS_BEHIND_FAR = 8 };
enum { N_COUNTERS = 128, BAD_FLOATS = 12 };

struct AuditParams {
    const char* scene;
    const uint32_t* pencil_masks;
    unsigned long long* counters;   // N_COUNTERS
    float* bad;                     // max_bad x BAD_FLOATS: family-specific record of the first violations
    unsigned int* n_bad;
    int max_bad;
    unsigned long long seed;
    int iters;                      // rays per thread
    int family;
};

// ---- random numbers: one stream per ray, SplitMix64 ----
struct Rng {
    unsigned long long x;
    __device__ unsigned long long next() { x += 0x9e3779b97f4a7c15ull; unsigned long long z = x; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
    __device__ float u01() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }               // [0, 1)
    __device__ float sym() { return 2.0f * u01() - 1.0f; }
    __device__ float gauss() { float a = 0.0f; for (int i = 0; i < 6; i++) a += u01(); return (a - 3.0f) * 1.41421356f; }
    __device__ float logu(float lo, float hi) { return lo * __powf(hi / lo, u01()); }
    __device__ f3 unit() { f3 v; float l2; do { v = mk3(gauss(), gauss(), gauss()); l2 = dot3(v, v); } while (!(l2 > 1e-12f)); return v * (1.0f / sqrtf(l2)); }
};
__device__ f3 any_perp(f3 n)
{
    const f3 a = fabsf(n.x) < 0.6f ? mk3(1.0f, 0.0f, 0.0f) : mk3(0.0f, 1.0f, 0.0f);
    return normalize3(cross3(n, a));
}
// a unit direction within a few degrees of the plane across n (any direction in that plane)
__device__ f3 grazing_dir(Rng& R, f3 n)
{
    const f3 u = any_perp(n), v = cross3(n, u);
    const float a = 6.2831853f * R.u01();
    return normalize3(u * __cosf(a) + v * __sinf(a) + n * (0.05f * R.sym() * R.u01()));
}
__device__ float ray_tmin(Rng& R) { return R.u01() < 0.5f ? RT_MAXDIST : __powf(10.0f, -1.0f + 5.0f * R.u01()); }

__device__ void record_bad(const AuditParams& p, int kind, int prim, f3 ro, f3 rd, float tmin, float t, float extra)
{
    const unsigned k = atomicAdd(p.n_bad, 1u);
    if ((int)k >= p.max_bad) return;
    float* o = p.bad + (size_t)k * BAD_FLOATS;
    o[0] = (float)kind; o[1] = (float)prim; o[2] = ro.x; o[3] = ro.y; o[4] = ro.z; o[5] = rd.x; o[6] = rd.y; o[7] = rd.z; o[8] = tmin; o[9] = t; o[10] = extra; o[11] = 0.0f;
}

__device__ void flush(const AuditParams& p, const unsigned int* c)
{
    for (int k = 0; k < N_COUNTERS; k++) {
        unsigned long long s = c[k];
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if ((threadIdx.x & 63u) == 0u && s) atomicAdd(p.counters + k, s);
    }
}

// ================================================================================================================================
// tori: sphere cull (torus_cull), group sphere, convex-hull cull, puck / hole cull, and the premise of the candidate tables (the ray's LINE
// 's part up to the limit stays 6 mm clear of the real tube, in exact arithmetic) -- each judged ON ITS OWN against the un-culled solver
// counters: 0 rays, 1 culled by any, 2 sphere, 3 group, 4 hull, 5 puck, 6 line-miss, 7 solver runs, 8 literal hits among the solved,
//           10 VIOLATIONS sphere, 11 group, 12 hull, 13 puck, 14 line-miss, 15 a non-unit direction was culled, 16 rays with a non-unit direction,
//           17 culled by the tube (Bernstein) test behind the puck test, 18 VIOLATIONS of it
// ================================================================================================================================
__device__ void torus_ray(Rng& R, const DevTorus& T, const f4 bound, int mode, f3& ro, f3& rd, float& tmin)
{
    const float Rm = fabsf(T.radii.x), rt = fabsf(T.radii.y), ext = Rm + rt;
    const f3 c = xyz(T.pos);
    tmin = ray_tmin(R);
    if (mode <= 3) {                          // around the torus: from 0.02 ... 200 units (3 of 4) or 200 ... 1e5 units out, aimed at or near it
        const float dist = mode == 3 ? R.logu(200.0f, 1.0e5f) : R.logu(0.02f, 200.0f);
        ro = c + R.unit() * dist;
        const float spread = ext * (R.u01() < 0.6f ? 1.2f : 6.0f);
        const f3 target = c + mk3(R.gauss(), R.gauss(), R.gauss()) * spread;
        rd = normalize3(target - ro);
        if (R.u01() < 0.1f) rd = -rd;
    } else if (mode <= 5) {                   // rays that START on the torus (its own shadow / mirror rays): a surface point pushed out by a gap,
                                              // displaced along an incoming ray by the solver's +-1e-3; outward directions incl. grazing ones
        const float phi = 6.2831853f * R.u01(), th = 6.2831853f * R.u01();
        const float cp = __cosf(phi), sp = __sinf(phi), ct = __cosf(th), st = __sinf(th);
        const f3 n = mk3(ct * cp, ct * sp, st);
        const float gap = R.logu(1.0e-5f, 1.0e-1f);
        f3 d = R.unit();
        float dn = dot3(d, n);
        if (R.u01() < 0.33f) { const float f = 0.05f * R.u01(); d = normalize3(d - n * ((1.0f - f) * dn)); dn = dot3(d, n); }
        if ((dn < 0.0f) != (R.u01() < 0.1f)) d = -d;
        const f3 in = R.unit();
        const float jt = 1.0e-3f * R.sym();
        const f3 ol = mk3((Rm + rt * ct) * cp, (Rm + rt * ct) * sp, rt * st) + n * gap + in * jt;
        ro = quat_rotate(T.qinv, ol) + c;
        rd = normalize3(quat_rotate(T.qinv, d));
    } else if (mode == 6) {                   // grazing the cull sphere: a point within +-2 % of its surface, a direction near the tangent plane
        const float rb = sqrtf(bound.w);
        const f3 n = R.unit();
        const f3 pb = c + n * (rb * (1.0f + 0.02f * R.sym()));
        const f3 d = grazing_dir(R, n);
        ro = pb - d * R.logu(0.01f, 300.0f);
        rd = d;
    } else {                                  // grazing the puck / the hole cylinder in the torus' frame
        const float hz = T.cull.x, ro2 = sqrtf(T.k.z), rh = sqrtf(T.k.w);
        const int which = (int)(R.u01() * 3.0f);
        f3 pl;
        if (which == 0) { const float rho = ro2 * sqrtf(R.u01()), a = 6.2831853f * R.u01(); pl = mk3(rho * __cosf(a), rho * __sinf(a), (R.u01() < 0.5f ? -hz : hz) * (1.0f + 0.02f * R.sym())); }
        else { const float rad = (which == 1 || rh == 0.0f ? ro2 : rh) * (1.0f + 0.02f * R.sym()), a = 6.2831853f * R.u01(); pl = mk3(rad * __cosf(a), rad * __sinf(a), hz * R.sym()); }
        f3 d = R.unit();
        if (R.u01() < 0.5f) { const f3 n = which == 0 ? mk3(0.0f, 0.0f, 1.0f) : normalize3(mk3(pl.x, pl.y, 0.0f)); d = normalize3(d - n * (dot3(d, n) * (1.0f - 0.05f * R.u01()))); }
        const f3 ol = pl - d * R.logu(0.01f, 120.0f);
        ro = quat_rotate(T.qinv, ol) + c;
        rd = normalize3(quat_rotate(T.qinv, d));
    }
}
// Distances of a point (torus frame) from the zero set of the quartic. The zero set is { D- = r } and, for a self-intersecting torus (r > R:
// tests/random_scenes.py nasty_scene has them), also { D+ = r }: D-, D+ = the distances to the nearest and the farthest point of the circle of
// radius R through the point's meridian plane. tube_dist: unsigned distance from the zero set; tube_sd: negative inside the solid tube.
__device__ double tube_sd(double Rm, double rt, double x, double y, double z)
{
    const double rho = sqrt(x * x + y * y);
    return sqrt((rho - Rm) * (rho - Rm) + z * z) - rt;
}
__device__ double tube_dist(double Rm, double rt, double x, double y, double z)
{
    const double rho = sqrt(x * x + y * y);
    const double dm = fabs(sqrt((rho - Rm) * (rho - Rm) + z * z) - rt);
    if (rt <= Rm) return dm;
    const double dp = fabs(sqrt((rho + Rm) * (rho + Rm) + z * z) - rt);
    return dm < dp ? dm : dp;
}
// How close does the part 0 <= t <= tl of the ray come to the surface of the tube (torus frame, double)? 0 if it touches or enters it.
// Sampled (256 points over the part inside the torus' bounding sphere + 0.5) and refined around the three best samples: an estimate from
// above (a missed dip only makes the clearance look larger, i.e. the audit stricter).
__device__ double ray_tube_clearance(const DevTorus& T, f3 of, f3 df, double tl)
{
    const double Rm = fabs((double)T.radii.x), rt = fabs((double)T.radii.y), Rb = Rm + rt + 0.5;
    const double ox = of.x, oy = of.y, oz = of.z, dx = df.x, dy = df.y, dz = df.z;
    const double a = dx * dx + dy * dy + dz * dz, b = ox * dx + oy * dy + oz * dz, c = ox * ox + oy * oy + oz * oz - Rb * Rb;
    const double h = b * b - a * c;
    if (!(h >= 0.0)) return 1.0e9;
    const double sh = sqrt(h);
    double t0 = (-b - sh) / a, t1 = (-b + sh) / a;
    t0 = t0 < 0.0 ? 0.0 : t0;
    t1 = t1 > tl ? tl : t1;
    if (!(t0 <= t1)) return 1.0e9;
    auto g = [&](double t) { return tube_dist(Rm, rt, ox + t * dx, oy + t * dy, oz + t * dz); };
    const int N = 256;
    const double dt = (t1 - t0) / N;
    double bt[3] = {t0, t0, t0}, bg[3] = {1.0e30, 1.0e30, 1.0e30};
    double prev_m = 0.0, prev_p = 0.0;
    for (int k = 0; k <= N; k++) {
        const double t = t0 + dt * k, v = g(t);
        // a sign change of D- - r (or, for a self-intersecting torus, of D+ - r) between two samples is a crossing of the surface, whatever the
        // samples' own distances are (a near miss elsewhere on the ray may well beat them)
        const double x = ox + t * dx, y = oy + t * dy, z = oz + t * dz, rho = sqrt(x * x + y * y);
        const double sm = sqrt((rho - Rm) * (rho - Rm) + z * z) - rt, sp = rt > Rm ? sqrt((rho + Rm) * (rho + Rm) + z * z) - rt : 1.0;
        if (k > 0 && ((sm <= 0.0) != (prev_m <= 0.0) || (sp <= 0.0) != (prev_p <= 0.0))) return 0.0;
        prev_m = sm; prev_p = sp;
        if (v < bg[0]) { bg[2] = bg[1]; bt[2] = bt[1]; bg[1] = bg[0]; bt[1] = bt[0]; bg[0] = v; bt[0] = t; }
        else if (v < bg[1]) { bg[2] = bg[1]; bt[2] = bt[1]; bg[1] = v; bt[1] = t; }
        else if (v < bg[2]) { bg[2] = v; bt[2] = t; }
    }
    double best = bg[0];
    for (int s = 0; s < 3; s++) {
        double lo = bt[s] - dt, hi = bt[s] + dt;
        lo = lo < t0 ? t0 : lo; hi = hi > t1 ? t1 : hi;
        for (int k = 0; k < 48; k++) { const double m1 = lo + (hi - lo) / 3.0, m2 = hi - (hi - lo) / 3.0; if (g(m1) < g(m2)) hi = m2; else lo = m1; }
        const double v = g(0.5 * (lo + hi));
        best = v < best ? v : best;
    }
    return best > 1.0e-9 ? best : 0.0;      // a crossing of the surface is a V-shaped minimum of the unsigned distance: the search ends within 1e-12 of 0
}
// The smallest t >= 0 at which the ray is within `thresh` of the tube's surface (torus frame, double; 512 samples over the part of the ray
// inside the bounding sphere + 1, then bisection); a negative value if there is none up to t = 400.
__device__ double ray_tube_first_entry(const DevTorus& T, f3 of, f3 df, double thresh)
{
    const double Rm = fabs((double)T.radii.x), rt = fabs((double)T.radii.y), Rb = Rm + rt + 1.0;
    const double ox = of.x, oy = of.y, oz = of.z, dx = df.x, dy = df.y, dz = df.z;
    const double a = dx * dx + dy * dy + dz * dz, b = ox * dx + oy * dy + oz * dz, c = ox * ox + oy * oy + oz * oz - Rb * Rb;
    const double h = b * b - a * c;
    if (!(h >= 0.0)) return -1.0;
    const double sh = sqrt(h);
    double t0 = (-b - sh) / a, t1 = (-b + sh) / a;
    t0 = t0 < 0.0 ? 0.0 : t0;
    t1 = t1 > 400.0 ? 400.0 : t1;
    if (!(t0 <= t1)) return -1.0;
    auto g = [&](double t) { return tube_sd(Rm, rt, ox + t * dx, oy + t * dy, oz + t * dz) - thresh; };
    if (g(t0) <= 0.0) return t0;
    const int N = 512;
    const double dt = (t1 - t0) / N;
    double prev = t0;
    for (int k = 1; k <= N; k++) {
        const double t = t0 + dt * k;
        if (g(t) <= 0.0) {
            double lo = prev, hi = t;
            for (int j = 0; j < 50; j++) { const double m = 0.5 * (lo + hi); if (g(m) <= 0.0) hi = m; else lo = m; }
            return hi;
        }
        prev = t;
    }
    return -1.0;
}
__device__ void audit_torus(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c)
{
    const int n = S.h->n_torus;
    if (n == 0) return;
    const bool grouped = n >= RT_GROUP_MIN;
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int i = (int)(R.next() % (unsigned long long)n);
        const DevTorus T = S.tori()[i];
        const f4 bound = S.torus_bound()[i];
        f3 ro, rd;
        float tmin;
        torus_ray(R, T, bound, (int)(R.next() & 7ull), ro, rd, tmin);
        const bool scaled = R.u01() < 0.02f;                       // a share of non-unit directions: nothing may cull them
        if (scaled) rd = rd * (R.u01() < 0.5f ? 1.0f + R.logu(2.0e-3f, 1.0f) : 1.0f - R.logu(2.0e-3f, 0.7f));
        const bool ident = ident_flag(T.pos.w);
        const f3 o = quat_rotate_id(T.quat, ident, ro - xyz(T.pos)), d = quat_rotate_id(T.quat, ident, rd);
        const bool unit = unit_direction(dot3(d, d));
        const bool c_sphere = torus_cull(bound, ro, rd);
        const bool c_group = grouped && torus_group_cull(S.torus_group()[i / RT_GROUP], ro, rd);
        // the local culls as the product composes them (round 6: from an origin outside the bounding sphere the hull / puck / tube tests run on the ONE half of the line that can meet the torus
        // -- rt_device.h torus_local_cull, the "behind" rule), attributed to the test that fired
        const bool c_local = unit && torus_local_cull<true>(T, o, d);
        // (the half-line the product judges: the reversed ray from an origin outside the bounding sphere with the centre behind it)
        const bool back_half = RT_TORUS_BEHIND_RULE && dot3(o, o) > T.k.z && dot3(o, d) >= 0.0f;
        const f3 dj = back_half ? mk3(-d.x, -d.y, -d.z) : d;
        const bool f_hull = unit && torus_hull_cull(T, o, dj);
        float pk0 = 0.0f, pk1 = 0.0f;
        const bool f_puck = unit && torus_puck_cull(T, o, dj, pk0, pk1);
        (void)pk0; (void)pk1;
        const bool c_hull = c_local && f_hull, c_puck = c_local && !f_hull && f_puck, c_tube = c_local && !f_hull && !f_puck;
        // the premise in its strongest form: the ray's part up to the reference's own reach (t < 100: RT_TORUS_REACH, never the ray's limit --
        // rt_device.h torus_cull) stays 6 mm clear of the REAL tube (exact) -- every cull and every clear bit of a candidate table implies it
        // (their margins are 1 % + 0.01 and more); from an origin outside the bounding sphere (the "behind" rule) the backward half up to RT_TORUS_REACH_BACK as well
        // (round 6: "6 mm" was 0.6 of the smallest inflation any cull uses, 1 cm -- now 0.6 of the tube's own inflation T.cull.x - |r|, which is 1 cm + 1 %
        // for r >= 0.3 and grows to RT_TORUS_IM_NOISE for thin tubes: rt_pack.h)
        const double clear_min = 0.6 * ((double)T.cull.x - fabs((double)T.radii.y));
        bool c_line = unit && isfinite(bound.w) && ray_tube_clearance(T, o, d, (double)RT_TORUS_REACH * 1.001 + 0.01) >= clear_min;
        if (c_line && RT_TORUS_BEHIND_RULE && dot3(o, o) > 0.98f * T.k.z)       // (T.k.z: the bounding sphere's radius^2 -- outside it the product's culls look at the whole line)
            c_line = ray_tube_clearance(T, o, mk3(-d.x, -d.y, -d.z), (double)RT_TORUS_REACH_BACK * 1.001 + 0.01) >= clear_min;
        const bool any = c_sphere || c_group || c_hull || c_puck || c_tube || c_line;
        c[0]++; c[1] += any; c[2] += c_sphere; c[3] += c_group; c[4] += c_hull; c[5] += c_puck; c[6] += c_line; c[16] += scaled; c[17] += c_tube;
        // product's own composition must agree with the parts (intersect_torus_c<true> is what the scans call)
        if (scaled && (c_sphere || c_group || c_hull || c_puck || c_tube) && !unit_direction(dot3_fma(rd, rd))) { c[15]++; record_bad(p, 15, i, ro, rd, tmin, 0.0f, dot3(rd, rd)); }
        if (any) {
            float t = 0.0f;
            const bool hit = intersect_torus(T, ro, rd, tmin, t);
            c[7]++; c[8] += hit;
            if (hit) {
                if (c_sphere) { c[10]++; record_bad(p, 10, i, ro, rd, tmin, t, 0.0f); }
                if (c_group) { c[11]++; record_bad(p, 11, i, ro, rd, tmin, t, 0.0f); }
                if (c_hull) { c[12]++; record_bad(p, 12, i, ro, rd, tmin, t, 0.0f); }
                if (c_puck) { c[13]++; record_bad(p, 13, i, ro, rd, tmin, t, 0.0f); }
                if (c_tube) { c[18]++; record_bad(p, 18, i, ro, rd, tmin, t, 0.0f); }
                if (c_line) { c[14]++; record_bad(p, 14, i, ro, rd, tmin, t, 0.0f); }
            }
        }
    }
}

// The premise itself, measured: when the reference's solver reports a hit, how far from the REAL tube does the ray pass (0 if it touches
// it)? Every ray is solved; for a reported hit the clearance of the ray's part 0 <= t <= 1.001 torus_limit(tmin) + 0.01 from the tube's
// surface is evaluated in double (ray_tube_clearance). A hit with a positive clearance is a phantom: the culls stay correct as long as no
// phantom has a clearance beyond their inflation (1 % of R + r, + 0.01: a ray that far out is what the sphere / puck culls remove).
// Also: how far from the tube's surface is the reported hit POINT (accuracy of the accepted root, not a matter of the culls).
// counters: 0 rays, 1 hits, 2 hits whose ray touches the tube, 3..11 phantoms with a clearance in [10^(k-10), 10^(k-9)) (3: below 1e-6 ...
//           11: >= 10), 12 VIOLATIONS phantoms beyond the inflation, 13..18 |distance| of the hit point from the surface < 1e-5, < 1e-4, < 1e-3,
//           < 1e-2, < 1e-1, >= 1e-1; 20 largest phantom clearance (float bits, atomicMax), 21 largest hit-point distance;
//           22..27 hits by class of t, 28..33 of them reported earlier than 1e-3 t + 0.01 before the ray enters the inflated tube, 34..39 the largest lead
__device__ void audit_torus_margin(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c, unsigned int& worst, unsigned int& worst_pt, unsigned int* lead_bits, unsigned int& worst_im)
{
    const int n = S.h->n_torus;
    if (n == 0) return;
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int i = (int)(R.next() % (unsigned long long)n);
        const DevTorus T = S.tori()[i];
        if (!(T.cull.y < RT_FLT_MAX)) continue;          // tori that are never culled (zero tube, non-unit quaternion) are outside every premise
        f3 ro, rd;
        float tmin;
        torus_ray(R, T, S.torus_bound()[i], (int)(R.next() & 7ull), ro, rd, tmin);
        float t = 0.0f;
        c[0]++;
        if (!intersect_torus(T, ro, rd, tmin, t)) continue;
        c[1]++;
        const bool ident = ident_flag(T.pos.w);
        const f3 o = quat_rotate_id(T.quat, ident, ro - xyz(T.pos)), d = quat_rotate_id(T.quat, ident, rd);
        const double tl = (double)RT_TORUS_REACH * 1.001 + 0.01;     // the reference's own reach: the culls never use the ray's limit
        const double clr = ray_tube_clearance(T, o, d, tl);
        const double infl = (double)T.cull.x - fabs((double)T.radii.y);      // the smallest inflation any cull of this torus uses (the puck's half height)
        if (clr <= 0.0) c[2]++;
        else {
            // |Im| of the complex root pair of a ray that clears a tube of radius r by clr (locally a cylinder): what the solver took for real
            const double rt_ = fabs((double)T.radii.y);
            const unsigned ib = __builtin_bit_cast(unsigned, (float)sqrt(clr * (2.0 * rt_ + clr)));
            worst_im = ib > worst_im ? ib : worst_im;
            int k = 3;
            for (double lim = 1.0e-6; k < 11 && clr >= lim; lim *= 10.0) k++;
            c[k]++;
            if (clr > infl) { c[12]++; record_bad(p, 12, i, ro, rd, tmin, t, (float)clr); }
            const unsigned bits = __builtin_bit_cast(unsigned, (float)clr);
            worst = bits > worst ? bits : worst;
        }
        // how much EARLIER than the ray's true entry into the inflated tube is the hit reported? (the limit tests of the culls -- "entered beyond
        // tlimit" -- assume at most 1e-3 tlimit + 0.01). Per class of the reported t: below 4, 8, 16, 32, 64, beyond.
        {
            const double tin = ray_tube_first_entry(T, o, d, infl);
            if (tin >= 0.0) {
                const float lead = (float)(tin - (double)t);
                const int bin = t < 4.0f ? 0 : t < 8.0f ? 1 : t < 16.0f ? 2 : t < 32.0f ? 3 : t < 64.0f ? 4 : 5;
                c[22 + bin]++;
                if (lead > 1.0e-3f * t + 0.01f) { c[28 + bin]++; }
                // candidates for a wider limit margin: 40 lead > 1e-3 t + 0.01 + 0.025 (t - 8) for t > 8; 41 lead > 1; 42 lead > 5
                if (t > 8.0f && lead > 1.0e-3f * t + 0.01f + 0.025f * (t - 8.0f)) c[40]++;
                if (lead > 1.0f) c[41]++;
                if (lead > 5.0f) c[42]++;
                if (lead > 0.0f) { const unsigned lb = __builtin_bit_cast(unsigned, lead); lead_bits[bin] = lb > lead_bits[bin] ? lb : lead_bits[bin]; }
            }
        }
        const double px = (double)o.x + (double)t * d.x, py = (double)o.y + (double)t * d.y, pz = (double)o.z + (double)t * d.z;
        const float a = (float)tube_dist(fabs((double)T.radii.x), fabs((double)T.radii.y), px, py, pz);
        c[a < 1.0e-5f ? 13 : a < 1.0e-4f ? 14 : a < 1.0e-3f ? 15 : a < 1.0e-2f ? 16 : a < 1.0e-1f ? 17 : 18]++;
#ifdef AUDIT_DEBUG_POINT
        if (a >= 0.1f && t >= 4.0f && t < 8.0f) record_bad(p, 99, i, ro, rd, tmin, t, a);
#endif
        const unsigned pb = __builtin_bit_cast(unsigned, a);
        worst_pt = pb > worst_pt ? pb : worst_pt;
    }
}

// The length premise by distance: how much EARLIER than the ray's entry into the inflated tube does the solver report a hit, as a function of
// |o| = the origin's distance from the torus' centre? (The measurement behind "no length limit in any torus cull", rt_device.h torus_cull.) Every ray is solved; origins 1.5 ... 64 units out, aimed at the tube's surface (grazing-heavy).
// bins b = 0..9 of |o|: < 2, 4, 6, 8, 10, 12, 16, 24, 48, beyond.
// counters: 0 rays, 1 hits, 10+b hits, 20+b reported more than 1e-3 t + 0.01 before the entry, 30+b ... + 0.025 (t - 8) (the widened limit),
//           40+b more than 0.1 + 1e-3 t early, 50+b more than 1 early, 60+b the largest lead (float bits), 70+b phantoms that clear the real tube by more
//           than 1 mm, 80+b the largest clearance of a phantom (float bits), 90+b hits with no entry into the inflated tube at all (up to t = 400)
__device__ int lead_bin(float dist) { return dist < 2.0f ? 0 : dist < 4.0f ? 1 : dist < 6.0f ? 2 : dist < 8.0f ? 3 : dist < 10.0f ? 4 : dist < 12.0f ? 5 : dist < 16.0f ? 6 : dist < 24.0f ? 7 : dist < 48.0f ? 8 : 9; }
__device__ void audit_torus_lead(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c, unsigned int* maxbits)
{
    const int n = S.h->n_torus;
    if (n == 0) return;
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int i = (int)(R.next() % (unsigned long long)n);
        const DevTorus T = S.tori()[i];
        if (!(T.cull.y < RT_FLT_MAX)) continue;
        const float Rm = fabsf(T.radii.x), rt = fabsf(T.radii.y);
        // a point of the tube's surface (torus frame), jittered; an origin 1.5 ... 64 from the centre; half the rays graze (the direction is
        // turned into the tangent plane at the surface point, up to a few degrees)
        const float phi = 6.2831853f * R.u01(), th = 6.2831853f * R.u01();
        const float cp = __cosf(phi), sp = __sinf(phi), ct = __cosf(th), st = __sinf(th);
        const f3 nl = mk3(ct * cp, ct * sp, st);
        const f3 sl = mk3((Rm + rt * ct) * cp, (Rm + rt * ct) * sp, rt * st) + mk3(R.gauss(), R.gauss(), R.gauss()) * (R.u01() < 0.5f ? 1.0e-3f : 0.05f * (Rm + rt));
        const float dist = R.logu(1.5f, 64.0f);
        f3 ol = R.unit() * dist;
        if (R.u01() < 0.5f) {      // grazing: origin in (almost) the tangent plane of the surface point
            f3 dl = normalize3(sl - ol);
            dl = normalize3(dl - nl * (dot3(dl, nl) * (1.0f - 0.1f * R.u01())));
            ol = sl - dl * dist;
        }
        const f3 ro = quat_rotate(T.qinv, ol) + xyz(T.pos);
        const f3 rd = normalize3(quat_rotate(T.qinv, normalize3(sl - ol)));
        float t = 0.0f;
        c[0]++;
        if (!intersect_torus(T, ro, rd, RT_MAXDIST, t)) continue;
        c[1]++;
        const bool ident = ident_flag(T.pos.w);
        const f3 o = quat_rotate_id(T.quat, ident, ro - xyz(T.pos)), d = quat_rotate_id(T.quat, ident, rd);
        const int b = lead_bin(sqrtf(dot3(o, o)));
        c[10 + b]++;
        const double infl = (double)T.cull.x - (double)rt;       // the tube's own inflation (rt_pack.h rinf - r): 1 cm + 1 % for r >= 0.3, up to RT_TORUS_IM_NOISE for thin tubes
        const double tin = ray_tube_first_entry(T, o, d, infl);
        if (tin < 0.0) { c[90 + b]++; record_bad(p, 90, i, ro, rd, RT_MAXDIST, t, 0.0f); }
        else {
            const float lead = (float)(tin - (double)t);
            if (lead > 1.0e-3f * t + 0.01f) c[20 + b]++;
            if (lead > 1.0e-3f * t + 0.01f + 0.025f * fmaxf(t - 8.0f, 0.0f)) c[30 + b]++;
            if (lead > 1.0e-3f * t + 0.1f) c[40 + b]++;
            if (lead > 1.0f) c[50 + b]++;
            if (lead > 0.0f) { const unsigned lb = __builtin_bit_cast(unsigned, lead); maxbits[b] = lb > maxbits[b] ? lb : maxbits[b]; }
        }
        const double clr = ray_tube_clearance(T, o, d, 100.11);
        if (clr > 1.0e-3) c[70 + b]++;
        if (clr > 0.0) { const unsigned cb = __builtin_bit_cast(unsigned, (float)clr); maxbits[10 + b] = cb > maxbits[10 + b] ? cb : maxbits[10 + b]; }
    }
}

// "Behind" rays (round 5, last session): rays that point AWAY from a torus their backward extension goes through -- every real root
// of the quartic is negative, every cull rejects them, and the reference's solver, which in float32 cannot meet its stop criterion on far real
// roots, occasionally ends its 60 sweeps with an iterate thrown to a positive t on the real axis: a phantom hit. The rate where it lives: every ray
// of this family is such a ray (a point of the tube's surface, jittered; an origin 1.5 ... 100 units from the centre aimed at it, half of them
// grazing; then the direction REVERSED), every one is solved. bins b = 0..9 of the origin's distance as in the lead family.
// counters: 0 rays, 1 culled by the product's composition (torus_cull or the local culls), 2 hits reported, 20+b rays per bin, 30+b phantom hits
//           per bin (hit reported, the half-line clears the real tube by more than 1 mm in double arithmetic), 40+b VIOLATIONS of those: culled by the
//           product (round 6: the "behind" rule of rt_device.h torus_cull lets origins outside the bounding sphere through to the solver -- a phantom that is SOLVED is reproduced)
__device__ void audit_torus_behind(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c)
{
    const int n = S.h->n_torus;
    if (n == 0) return;
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int i = (int)(R.next() % (unsigned long long)n);
        const DevTorus T = S.tori()[i];
        if (!(T.cull.y < RT_FLT_MAX)) continue;
        const float Rm = fabsf(T.radii.x), rt = fabsf(T.radii.y);
        const float phi = 6.2831853f * R.u01(), th = 6.2831853f * R.u01();
        const float cp = __cosf(phi), sp = __sinf(phi), ct = __cosf(th), st = __sinf(th);
        const f3 nl = mk3(ct * cp, ct * sp, st);
        const f3 sl = mk3((Rm + rt * ct) * cp, (Rm + rt * ct) * sp, rt * st) + mk3(R.gauss(), R.gauss(), R.gauss()) * (R.u01() < 0.5f ? 1.0e-3f : 0.05f * (Rm + rt));
        const float dist = R.logu(1.5f, 100.0f);
        f3 ol = R.unit() * dist;
        if (R.u01() < 0.5f) {
            f3 dl = normalize3(sl - ol);
            dl = normalize3(dl - nl * (dot3(dl, nl) * (1.0f - 0.1f * R.u01())));
            ol = sl - dl * dist;
        }
        const f3 ro = quat_rotate(T.qinv, ol) + xyz(T.pos);
        const f3 rd = -normalize3(quat_rotate(T.qinv, normalize3(sl - ol)));       // away from the torus
        const float tmin = ray_tmin(R);
        bool culled = torus_cull(S.torus_bound()[i], ro, rd);
        float t2 = 0.0f;
        bool solved = false;
        if (!culled) { intersect_torus_c<true, true>(T, ro, rd, tmin, t2, solved); culled = !solved; }
        const bool ident = ident_flag(T.pos.w);
        const f3 o = quat_rotate_id(T.quat, ident, ro - xyz(T.pos)), d = quat_rotate_id(T.quat, ident, rd);
        const int b = lead_bin(sqrtf(dot3(o, o)));
        c[0]++; c[1] += culled; c[20 + b]++;
        float t = 0.0f;
        if (!intersect_torus(T, ro, rd, tmin, t)) continue;
        c[2]++;
        if (ray_tube_clearance(T, o, d, 100.11) > 1.0e-3) { c[30 + b]++; c[40 + b] += culled; if (culled) record_bad(p, 40 + b, i, ro, rd, tmin, t, sqrtf(dot3(o, o))); }
    }
}

// The backward reach (RT_TORUS_REACH_BACK): the same rays from 104 ... 3000 units out -- the quartic's real roots lie more than 100 units BEHIND the
// origin. Every ray is solved; a hit is a phantom by construction. bins as in torus_far: < 120, 150, 200, 400, 1000, beyond.
// counters: 0 rays, 1 culled by the product's composition, 20+b rays per bin, 30+b hits reported per bin, 40+b VIOLATIONS of those: culled
__device__ void audit_torus_behind_far(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c)
{
    const int n = S.h->n_torus;
    if (n == 0) return;
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int i = (int)(R.next() % (unsigned long long)n);
        const DevTorus T = S.tori()[i];
        if (!(T.cull.y < RT_FLT_MAX)) continue;
        const float Rm = fabsf(T.radii.x), rt = fabsf(T.radii.y);
        const float phi = 6.2831853f * R.u01(), th = 6.2831853f * R.u01();
        const float cp = __cosf(phi), sp = __sinf(phi), ct = __cosf(th), st = __sinf(th);
        const f3 nl = mk3(ct * cp, ct * sp, st);
        const f3 sl = mk3((Rm + rt * ct) * cp, (Rm + rt * ct) * sp, rt * st) + mk3(R.gauss(), R.gauss(), R.gauss()) * (R.u01() < 0.5f ? 1.0e-3f : 0.05f * (Rm + rt));
        const float dist = (100.0f + 1.5f + Rm + rt) * R.logu(1.0f, 30.0f);
        f3 ol = R.unit() * dist;
        if (R.u01() < 0.5f) {
            f3 dl = normalize3(sl - ol);
            dl = normalize3(dl - nl * (dot3(dl, nl) * (1.0f - 0.1f * R.u01())));
            ol = sl - dl * dist;
        }
        const f3 ro = quat_rotate(T.qinv, ol) + xyz(T.pos);
        const f3 rd = -normalize3(quat_rotate(T.qinv, normalize3(sl - ol)));       // away from the torus
        const float tmin = ray_tmin(R);
        bool culled = torus_cull(S.torus_bound()[i], ro, rd);
        float t2 = 0.0f;
        bool solved = false;
        if (!culled) { intersect_torus_c<true, true>(T, ro, rd, tmin, t2, solved); culled = !solved; }
        const float dd = length3(ro - xyz(T.pos));
        const int b = dd < 120.0f ? 0 : dd < 150.0f ? 1 : dd < 200.0f ? 2 : dd < 400.0f ? 3 : dd < 1000.0f ? 4 : 5;
        c[0]++; c[1] += culled; c[20 + b]++;
        float t = 0.0f;
        if (!intersect_torus(T, ro, rd, tmin, t)) continue;
        c[30 + b]++;
        if (culled) { c[40 + b]++; record_bad(p, 40 + b, i, ro, rd, tmin, t, dd); }
    }
}

// The one length premise the torus culls keep: a torus the ray enters beyond the reference's own reach (roots are accepted for t < 100 only,
// rt.frag:486; RT_TORUS_REACH = 102.5) is culled -- i.e. "a solve from more than 100 units out does not report a root below 100". Every ray
// of this family is such a ray (origin 104 ... 3000 from the centre, aimed at the tube; the product's torus_cull is checked to fire) and
// every one is SOLVED: a reported hit is a violation. bins b = 0..5 of the origin's distance: < 120, 150, 200, 400, 1000, beyond.
// counters: 0 rays, 1 culled by torus_cull (must be all), 2 rays that really cross the tube, 10 VIOLATIONS a hit is reported, 20+b rays, 30+b hits
__device__ void audit_torus_far(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c)
{
    const int n = S.h->n_torus;
    if (n == 0) return;
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int i = (int)(R.next() % (unsigned long long)n);
        const DevTorus T = S.tori()[i];
        if (!(T.cull.y < RT_FLT_MAX)) continue;
        const float Rm = fabsf(T.radii.x), rt = fabsf(T.radii.y);
        const float phi = 6.2831853f * R.u01(), th = 6.2831853f * R.u01();
        const float cp = __cosf(phi), sp = __sinf(phi), ct = __cosf(th), st = __sinf(th);
        const f3 nl = mk3(ct * cp, ct * sp, st);
        const f3 sl = mk3((Rm + rt * ct) * cp, (Rm + rt * ct) * sp, rt * st) + mk3(R.gauss(), R.gauss(), R.gauss()) * (R.u01() < 0.5f ? 1.0e-3f : 0.05f * (Rm + rt));
        const float dist = (RT_TORUS_REACH + 1.5f + Rm + rt) * R.logu(1.0f, 30.0f);
        f3 ol = R.unit() * dist;
        if (R.u01() < 0.5f) {
            f3 dl = normalize3(sl - ol);
            dl = normalize3(dl - nl * (dot3(dl, nl) * (1.0f - 0.1f * R.u01())));
            ol = sl - dl * dist;
        }
        const f3 ro = quat_rotate(T.qinv, ol) + xyz(T.pos);
        const f3 rd = normalize3(quat_rotate(T.qinv, normalize3(sl - ol)));
        const float tmin = ray_tmin(R);
        const bool culled = torus_cull(S.torus_bound()[i], ro, rd);
        if (!culled) continue;                       // (a ray that enters within the reach after all: not this family's business)
        const float dd = length3(ro - xyz(T.pos));
        const int b = dd < 120.0f ? 0 : dd < 150.0f ? 1 : dd < 200.0f ? 2 : dd < 400.0f ? 3 : dd < 1000.0f ? 4 : 5;
        c[0]++; c[1]++; c[20 + b]++;
        float t = 0.0f;
        if (intersect_torus(T, ro, rd, tmin, t)) { c[10]++; c[30 + b]++; record_bad(p, 10, i, ro, rd, tmin, t, dd); }
    }
}

// ================================================================================================================================
// quadrics: surface_cull (segment test, tight / far bound, degenerate-branch margin), group sphere + quadric_may_degenerate, and the product's
// intersect_surface (wave-level exit for waves without a real root) against the literal rt.frag:513-572
// counters: 0 rays, 1 culled by surface_cull, 2 culled by the group test, 3 literal hits, 4 literal hits on the degenerate branch,
//           5 rays the product's intersector leaves early (no real root),
//           6 culled by the clip-box test behind surface_cull,
//           10 VIOLATIONS surface_cull, 11 group, 12 product intersector != literal (hit flag, or t on a hit), 13 clip-box test
// ================================================================================================================================
__device__ bool intersect_surface_literal(const DevSurface& Q, f3 ro_w, f3 rd_w, float tmin, float& t, bool& degenerate, bool* real_roots = nullptr)
{
    const f3 ro = quat_rotate(Q.quat, ro_w - xyz(Q.pos_a));
    const f3 rd = quat_rotate(Q.quat, rd_w);
    const float a = Q.pos_a.w, b = Q.bcde.x, c = Q.bcde.y, d = Q.bcde.z, e = Q.bcde.w, f = Q.f_vmin.x;
    const float d1 = rd.x, d2 = rd.y, d3 = rd.z, o1 = ro.x, o2 = ro.y, o3 = ro.z;
    const float p1 = 2.0f * a * d1 * o1 + 2.0f * b * d2 * o2 + 2.0f * c * d3 * o3 + d * d3 + d2 * e;
    const float p2 = a * d1 * d1 + b * d2 * d2 + c * d3 * d3;
    const float p3 = a * o1 * o1 + b * o2 * o2 + c * o3 * o3 + d * o3 + e * o2 + f;
    degenerate = fabsf(p2) < 1e-6f;
    if (real_roots) *real_roots = false;
    if (degenerate) { t = -p3 / p1; return t > tmin; }
    if (real_roots) *real_roots = !(p1 * p1 - 4.0f * p2 * p3 < 0.0f);
    const float p4 = sqrtf(p1 * p1 - 4.0f * p2 * p3);
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
__device__ f3 clip_centre(const DevSurface& Q, const DevSurfaceCull& C, float& ext)
{
    const f3 lo = mk3(Q.f_vmin.y, Q.f_vmin.z, Q.f_vmin.w), hi = xyz(Q.vmax);
    const f3 pos = xyz(Q.pos_a);
    f3 c = pos;
    ext = 2.0f;
    if (C.bound.w >= 0.0f && C.bound.w < 1.0e12f) { c = xyz(C.bound); ext = sqrtf(C.bound.w); }
    (void)lo; (void)hi;
    return c;
}
__device__ void quadric_ray(Rng& R, const DevSurface& Q, const DevSurfaceCull& C, int mode, f3& ro, f3& rd, float& tmin)
{
    float ext;
    const f3 c = clip_centre(Q, C, ext);
    tmin = ray_tmin(R);
    if (mode <= 2 || mode == 3) {             // around the quadric: 0.05 ... 200 units (3 of 4) or 200 ... 1e5 units out
        const float dist = mode == 3 ? R.logu(200.0f, 1.0e5f) : R.logu(0.05f, 200.0f);
        ro = c + R.unit() * dist;
        const float spread = ext * (R.u01() < 0.6f ? 1.5f : 8.0f);
        rd = normalize3(c + mk3(R.gauss(), R.gauss(), R.gauss()) * spread - ro);
        if (R.u01() < 0.1f) rd = -rd;
    } else if (mode == 4) {                   // directions next to the asymptotic cone: |p2| from 1e-8 to a few times the cull's margin, origins
                                              // anywhere incl. inside the clip box, finite limits (the ill-conditioned regime)
        const float a = Q.pos_a.w, b = Q.bcde.x, cc = Q.bcde.y;
        f3 u = R.unit(), v = R.unit();
        auto p2 = [&](f3 w) { return a * w.x * w.x + b * w.y * w.y + cc * w.z * w.z; };
        float pu = p2(u), pv = p2(v);
        f3 dl = u;
        if ((pu < 0.0f) != (pv < 0.0f)) {     // indefinite form: bisect along the arc for a root of p2, then step off it by a tiny angle
            for (int k = 0; k < 24; k++) { const f3 m = normalize3(u + v); const float pm = p2(m); if ((pm < 0.0f) == (pu < 0.0f)) { u = m; pu = pm; } else { v = m; pv = pm; } }
            dl = normalize3(u + R.unit() * R.logu(1.0e-8f, 3.0e-2f));
        }
        rd = normalize3(quat_rotate(Q.qinv, dl));
        const f3 through = c + mk3(R.sym(), R.sym(), R.sym()) * (ext * (R.u01() < 0.7f ? 0.6f : 3.0f));
        ro = through - rd * (R.u01() < 0.3f ? R.logu(1.0e-3f, 1.0f) * ext : R.logu(0.05f, 3000.0f));
        if (R.u01() < 0.7f) tmin = R.logu(0.1f, 1000.0f);
    } else if (mode == 5) {                   // origins inside the bound / the clip box, every direction, finite limits
        ro = c + mk3(R.sym(), R.sym(), R.sym()) * (ext * 0.7f);
        rd = R.unit();
        if (R.u01() < 0.7f) tmin = R.logu(1.0e-3f, 50.0f);
    } else if (mode == 6) {                   // grazing the cull sphere (the tight one within RT_QUADRIC_FAR, the clip box's beyond)
        const f3 n = R.unit();
        const bool far = R.u01() < 0.3f && C.sym1.w >= 0.0f;
        const float rb = far ? sqrtf(C.sym1.w) : ext;
        const f3 pb = c + n * (rb * (1.0f + 0.02f * R.sym()));
        const f3 d = grazing_dir(R, n);
        ro = pb - d * (far ? R.logu(60.0f, 3000.0f) : R.logu(0.01f, 60.0f));
        rd = d;
    } else if (mode == 7) {                   // origins around the switch between the two bounds (RT_QUADRIC_FAR +- 10)
        ro = c + R.unit() * ((float)RT_QUADRIC_FAR + 10.0f * R.sym());
        rd = normalize3(c + mk3(R.gauss(), R.gauss(), R.gauss()) * (ext * 1.5f) - ro);
    } else {                                  // rays that START on the quadric (shadow / mirror rays of its own hits: F(origin) is rounding noise of
                                              // either sign -- what the sign-based exit of intersect_surface judges): a hit point of the literal
                                              // intersector, as the shader forms it, left where it is or pushed off by up to 1e-3; any direction
        f3 o0 = c + R.unit() * R.logu(0.05f, 200.0f);
        f3 d0 = normalize3(c + mk3(R.gauss(), R.gauss(), R.gauss()) * (ext * 0.8f) - o0);
        float th = 0.0f;
        bool dg = false;
        if (intersect_surface_literal(Q, o0, d0, RT_FLT_MAX, th, dg) && th < 1.0e4f) {
            ro = d0 * th + o0;
            if (R.u01() < 0.5f) ro = ro + R.unit() * (R.u01() < 0.5f ? R.logu(1.0e-7f, 1.0e-4f) : R.logu(1.0e-4f, 1.0e-3f));
            rd = R.unit();
            if (R.u01() < 0.3f) rd = normalize3(rd - d0 * dot3(rd, d0) + d0 * (0.02f * R.sym()));   // grazing the incoming ray's normal plane, roughly
            if (R.u01() < 0.5f) tmin = R.logu(1.0e-3f, 50.0f);
        } else {
            ro = o0;
            rd = d0;
        }
    }
}
__device__ void audit_quadric(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c)
{
    const int n = S.h->n_surface;
    if (n == 0) return;
    const bool grouped = n >= RT_GROUP_MIN;
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int i = (int)(R.next() % (unsigned long long)n);
        const DevSurface Q = S.surfaces()[i];
        const DevSurfaceCull C = S.surf_cull()[i];
        f3 ro, rd;
        float tmin;
        quadric_ray(R, Q, C, (int)(R.next() % 10ull), ro, rd, tmin);
        bool safe = false;
        const bool c_cull = surface_cull(C, ro, rd, tmin, safe);
        const bool c_box = !c_cull && safe && surface_box_miss(Q, ro, rd, tmin);   // as the product composes it: behind the sphere test
        const bool c_group = grouped && surface_group_cull(S.surf_group()[i / RT_GROUP], ro, rd) && !quadric_may_degenerate(C, rd);
        float t_lit = 0.0f, t_prod = 0.0f;
        bool deg = false, real_roots = false;
        const bool hit = intersect_surface_literal(Q, ro, rd, tmin, t_lit, deg, &real_roots);
        const bool hit_p = intersect_surface<false>(Q, ro, rd, tmin, t_prod);   // every lane takes the early exits on its own condition
        c[0]++; c[1] += c_cull; c[2] += c_group; c[3] += hit; c[4] += hit && deg; c[6] += c_box;
        c[5] += !hit_p && !real_roots && !deg;   // left by the product's exit for lanes without a real root
        if (c_cull && hit) { c[10]++; record_bad(p, 10, i, ro, rd, tmin, t_lit, deg ? 1.0f : 0.0f); }
        if (c_group && hit) { c[11]++; record_bad(p, 11, i, ro, rd, tmin, t_lit, deg ? 1.0f : 0.0f); }
        if (c_box && hit) { c[13]++; record_bad(p, 13, i, ro, rd, tmin, t_lit, deg ? 1.0f : 0.0f); }
        if (hit != hit_p || (hit && __builtin_bit_cast(unsigned, t_lit) != __builtin_bit_cast(unsigned, t_prod))) { c[12]++; record_bad(p, 12, i, ro, rd, tmin, t_lit, t_prod); }
    }
}

// ================================================================================================================================
// rings: counters 0 rays, 1 culled, 2 literal hits, 10 VIOLATIONS
// ================================================================================================================================
__device__ void audit_ring(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c)
{
    const int n = S.h->n_ring;
    if (n == 0) return;
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int i = (int)(R.next() % (unsigned long long)n);
        const DevRing G = S.rings()[i];
        const f4 bound = S.ring_bound()[i];
        const f3 cen = xyz(G.pos_tex);
        const float ext = sqrtf(fabsf(G.radii.y)) + 1.0e-3f;
        f3 ro, rd;
        float tmin = ray_tmin(R);
        const int mode = (int)(R.next() & 3ull);
        if (mode <= 1) {
            ro = cen + R.unit() * (mode == 0 ? R.logu(0.02f, 200.0f) : R.logu(200.0f, 1.0e5f));
            rd = normalize3(cen + mk3(R.gauss(), R.gauss(), R.gauss()) * (ext * (R.u01() < 0.6f ? 1.2f : 6.0f)) - ro);
            if (R.u01() < 0.1f) rd = -rd;
        } else if (mode == 2) {               // through the rim: a point of the ring's plane within +-2 % of the outer radius
            const float a = 6.2831853f * R.u01(), rad = ext * (1.0f + 0.02f * R.sym());
            const f3 pl = mk3(rad * __cosf(a), rad * __sinf(a), 0.0f);
            const f3 pw = quat_rotate(quat_inv(G.quat), pl) + cen;
            rd = R.unit();
            ro = pw - rd * R.logu(0.01f, 300.0f);
        } else {                              // grazing the cull sphere
            const f3 nn = R.unit();
            const f3 pb = cen + nn * (sqrtf(bound.w) * (1.0f + 0.02f * R.sym()));
            rd = grazing_dir(R, nn);
            ro = pb - rd * R.logu(0.01f, 300.0f);
        }
        const bool cull = ring_cull(bound, ro, rd, tmin);
        float t = 0.0f;
        f2 uv;
        const bool hit = intersect_ring(G, ro, rd, tmin, t, uv);
        c[0]++; c[1] += cull; c[2] += hit;
        if (cull && hit) { c[10]++; record_bad(p, 10, i, ro, rd, tmin, t, 0.0f); }
    }
}

// ================================================================================================================================
// candidate tables (ray pencils + slab tables + direction table): rays built like the tracer builds them; for a sample of the primitives
// whose bit is CLEAR in the ray's mask: a quadric must not be hit by the literal intersector; a torus must stay 5 mm clear of the ray's part
// up to its limit in exact arithmetic (the premise the torus family audits against the solver).
// counters: 0 rays, 1 camera-pencil rays, 2 light-pencil rays, 3 slab-table rays, 4 rays that read every bit set, 5 set bits,
//           6 quadric checks, 7 torus checks, 10 VIOLATIONS quadric, 11 torus
// ================================================================================================================================
__device__ f3 crowd_point(Rng& R, f3 c, f3 half)
{
    const float far = R.u01() < 0.1f ? 40.0f : 1.0f;
    return c + mk3(R.gauss() * half.x, R.gauss() * half.y, R.gauss() * half.z) * (0.6f * far);
}
__device__ void audit_tables(const AuditParams& p, const SceneView& S, unsigned long long gid, unsigned int* c)
{
    const int ns = S.h->n_surface, nt = S.h->n_torus, nws = (ns + 31) >> 5, W = (int)S.h->pencil_stride;
    if (S.pen == nullptr || ns + nt == 0 || W == 0) return;
    const int n_lights = S.h->n_light_point + S.h->n_light_direct;
    // where the primitives are: the slab box if there is one, else a default crowd box
    f3 cen = mk3(0.0f, 0.0f, 16.0f), half = mk3(12.0f, 10.0f, 8.0f);
    if (slabs_available(S)) { const DevSlabs& B = *S.slabs(); cen = (xyz(B.lo) + xyz(B.hi)) * 0.5f; half = (xyz(B.hi) - xyz(B.lo)) * 0.5f; }
    for (int it = 0; it < p.iters; it++) {
        const unsigned long long ray = gid * (unsigned long long)p.iters + (unsigned long long)it;
        Rng R{p.seed * 0x2545f4914f6cdd1dull + ray * 0xd1342543de82ef95ull};
        const int kind = (int)(ray % 3ull);
        f3 ro, rd;
        float tlimit = RT_MAXDIST;
        uint32_t words[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
        bool have = false;
        if (kind == 0) {
            ro = xyz(S.h->cam_pos);
            rd = normalize3(crowd_point(R, cen, half) - ro);
            const PencilScan ps = pencil_open<true>(S, 0, ro, rd, 0.0f, true);
            if (ps.use) { have = true; c[1]++; for (int w = 0; w < W && w < 8; w++) words[w] = S.pen[ps.cell + w]; }
        } else if (kind == 1 && n_lights > 0) {
            const int li = (int)(R.next() % (unsigned long long)n_lights);
            ro = crowd_point(R, cen, half);
            if (li < S.h->n_light_point) { const f3 ld = xyz(S.lights_point()[li].pos_r2) - ro; tlimit = length3(ld); rd = normalize3(ld); }
            else rd = xyz(S.lights_direct()[li - S.h->n_light_point].dir_n);
            const PencilScan ps = pencil_open<true>(S, 1 + li, ro, rd, tlimit, false);
            if (ps.use) { have = true; c[2]++; for (int w = 0; w < W && w < 8; w++) words[w] = S.pen[ps.cell + w]; }
        }
        if (!have) {
            if (!slabs_available(S)) continue;
            ro = crowd_point(R, cen, half);
            rd = normalize3(crowd_point(R, cen, half) - ro);
            tlimit = R.u01() < 0.5f ? RT_MAXDIST : 1.0f + 40.0f * R.u01();
            slab_ray_mask(S, ro, rd, tlimit, words);
            c[3]++;
        }
        c[0]++;
        int bits = 0;
        for (int w = 0; w < W && w < 8; w++) bits += __builtin_popcount(words[w]);
        c[5] += (unsigned)bits;
        c[4] += bits == ns + nt;
        for (int s = 0; s < 6; s++) {           // six primitives per ray, clear bits only
            const int i = (int)(R.next() % (unsigned long long)(ns + nt));
            const int w = i < ns ? i >> 5 : nws + ((i - ns) >> 5), b = (i < ns ? i : i - ns) & 31;
            if ((words[w] >> b) & 1u) continue;
            if (i < ns) {
                float t = 0.0f;
                bool deg;
                c[6]++;
                if (intersect_surface_literal(S.surfaces()[i], ro, rd, tlimit, t, deg)) { c[10]++; record_bad(p, 10 + kind * 100, i, ro, rd, tlimit, t, deg ? 1.0f : 0.0f); }
            } else {
                c[7]++;
                const DevTorus T = S.tori()[i - ns];
                const bool ident = ident_flag(T.pos.w);
                const f3 o = quat_rotate_id(T.quat, ident, ro - xyz(T.pos)), d = quat_rotate_id(T.quat, ident, rd);
                // up to the reference's own reach, whatever the ray's limit: no torus cull uses that, so a clear bit must not depend on it either
                const double clr = ray_tube_clearance(T, o, d, (double)RT_TORUS_REACH * 1.001 + 0.01);
                const double clear_min = T.cull.y < RT_FLT_MAX ? 0.5 * ((double)T.cull.x - fabs((double)T.radii.y)) : 5.0e-3;    // half the tube's own inflation (5 mm for r >= 0.3)
                if (!(clr >= clear_min)) { c[11]++; record_bad(p, 11 + kind * 100, i - ns, ro, rd, tlimit, 0.0f, (float)clr); }
                // the "behind" rule (rt_device.h torus_cull): from an origin outside the bounding sphere the LINE's part behind the origin, up to the backward reach, as well
                if (RT_TORUS_BEHIND_RULE && T.cull.y < RT_FLT_MAX && dot3(o, o) > T.k.z) {
                    c[8]++;
                    const double back = ray_tube_clearance(T, o, mk3(-d.x, -d.y, -d.z), (double)RT_TORUS_REACH_BACK * 1.001 + 0.01);
                    if (!(back >= clear_min)) { c[12]++; record_bad(p, 12 + kind * 100, i - ns, ro, rd, tlimit, -1.0f, (float)back); }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void audit_kernel(const AuditParams p)
{
    const SceneView S = make_view(p.scene, reinterpret_cast<const DevSceneHeader*>(p.scene), p.pencil_masks);
    const unsigned long long gid = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    unsigned int c[N_COUNTERS];                     // per thread: at most iters x 160 each
    for (int k = 0; k < N_COUNTERS; k++) c[k] = 0u;
    unsigned int worst = 0u, worst_pt = 0u, worst_im = 0u, lead_bits[6] = {0u, 0u, 0u, 0u, 0u, 0u};
    if (p.family == F_TORUS) audit_torus(p, S, gid, c);
    else if (p.family == F_TORUS_MARGIN) audit_torus_margin(p, S, gid, c, worst, worst_pt, lead_bits, worst_im);
    else if (p.family == F_QUADRIC) audit_quadric(p, S, gid, c);
    else if (p.family == F_RING) audit_ring(p, S, gid, c);
    else if (p.family == F_TORUS_LEAD) {
        unsigned int mb[20];
        for (int k = 0; k < 20; k++) mb[k] = 0u;
        audit_torus_lead(p, S, gid, c, mb);
        for (int k = 0; k < 10; k++) {
            if (mb[k]) atomicMax(reinterpret_cast<unsigned int*>(p.counters + 60 + k), mb[k]);
            if (mb[10 + k]) atomicMax(reinterpret_cast<unsigned int*>(p.counters + 80 + k), mb[10 + k]);
        }
    }
    else if (p.family == F_TORUS_FAR) audit_torus_far(p, S, gid, c);
    else if (p.family == F_TORUS_BEHIND) audit_torus_behind(p, S, gid, c);
    else if (p.family == F_TORUS_BEHIND_FAR) audit_torus_behind_far(p, S, gid, c);
    else audit_tables(p, S, gid, c);
    flush(p, c);
    if (p.family == F_TORUS_MARGIN && worst) atomicMax(reinterpret_cast<unsigned int*>(p.counters + 20), worst);
    if (p.family == F_TORUS_MARGIN && worst_pt) atomicMax(reinterpret_cast<unsigned int*>(p.counters + 21), worst_pt);
    if (p.family == F_TORUS_MARGIN && worst_im) atomicMax(reinterpret_cast<unsigned int*>(p.counters + 43), worst_im);
    if (p.family == F_TORUS_MARGIN)
        for (int b = 0; b < 6; b++) if (lead_bits[b]) atomicMax(reinterpret_cast<unsigned int*>(p.counters + 34 + b), lead_bits[b]);
}

__global__ __launch_bounds__(256) void audit_pencil_build(const char* scene, uint32_t* masks)
{
    __shared__ PencilPrim prims[2 * RT_PENCIL_MAX_PRIMS];
    const SceneView S = make_view(scene);
    const DevPencil P = S.pencils()[blockIdx.y];
    if (P.kind == RT_PENCIL_OFF || blockIdx.x * 256u > P.cells) return;
    const int n = S.h->n_surface + S.h->n_torus;
    for (int k = threadIdx.x; k < n; k += 256) prims[k] = pencil_prim_at(S, P, k);
    __syncthreads();
    const uint32_t cell = blockIdx.x * 256u + threadIdx.x;
    if (cell > P.cells) return;
    const PencilCell C = pencil_cell_geometry(P, cell);
    masks[P.mask_off + (size_t)cell * S.h->pencil_stride + blockIdx.z] = pencil_cell_word(S, P, prims, C, cell, (int)blockIdx.z);
}

std::string g_err;
#define TRY(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { g_err = std::string(#expr) + ": " + hipGetErrorString(e_); return -1; } } while (0)

}  // namespace

extern "C" {

__attribute__((visibility("default"))) const char* cull_audit_error() { return g_err.c_str(); }

}  // extern "C"
namespace {
// the scene as the product packs it (rt_pack.h), on the device, with its pencil masks built
struct DeviceScene { char* scene = nullptr; uint32_t* masks = nullptr; };
int upload_scene(const rtpack::Defines* d, const void* const* blocks, const uint64_t* sizes, DeviceScene& out)
{
    std::vector<unsigned char> blk[rtpack::BLK_COUNT];
    for (int b = 0; b < rtpack::BLK_COUNT; b++) {
        const unsigned char* p = static_cast<const unsigned char*>(blocks[b]);
        if (p && sizes[b]) blk[b].assign(p, p + sizes[b]);
    }
    std::vector<unsigned char> blob;
    if (!rtpack::pack_scene(*d, blk, blob, g_err)) return -1;
    const DevSceneHeader hdr = *reinterpret_cast<const DevSceneHeader*>(blob.data());
    TRY(hipMalloc(&out.scene, (blob.size() + 15) & ~size_t(15)));
    TRY(hipMemcpy(out.scene, blob.data(), blob.size(), hipMemcpyHostToDevice));
    if (hdr.n_pencil > 0 && hdr.pencil_mask_words > 0) {
        TRY(hipMalloc(&out.masks, (size_t)hdr.pencil_mask_words * 4));
        TRY(hipMemset(out.masks, 0, (size_t)hdr.pencil_mask_words * 4));
        const DevPencil* pencils = reinterpret_cast<const DevPencil*>(blob.data() + hdr.off_pencil);
        const uint32_t records = hdr.n_pencil + (hdr.pencil_dir != 0xffffffffu ? 1u : 0u);
        uint32_t most = 0;
        for (uint32_t k = 0; k < records; k++) if (pencils[k].kind != RT_PENCIL_OFF && pencils[k].cells > most) most = pencils[k].cells;
        if (most) hipLaunchKernelGGL(audit_pencil_build, dim3((most + 1 + 255) / 256, records, hdr.pencil_stride), dim3(256), 0, 0, out.scene, out.masks);
        TRY(hipGetLastError());
    }
    return 0;
}

// ---- replay of single rays through the product's own scans (tests/test_gpu_culls.py): per ray (ro, rd, limit, torus index)
//   out[0..1]  the literal intersector of that torus (rt.frag:462-487 as the product restates it, no cull): hit, t
//   out[2..3]  the product's composition for that torus -- torus_cull, the group sphere, intersect_torus_c<true> (hull, puck, tube) --: hit, t
//   out[4..5]  in_shadow over the WHOLE scene with `limit` as the distance to the light, culls (incl. slab tables) on / off
//   out[6..8]  calc_inter over the whole scene, culls on: t, num, type;  out[9..11] the same with the culls off
// A lane holds one ray; the scans' wave-level votes see unrelated rays, as they do at an object's silhouette.
__global__ __launch_bounds__(64) void probe_kernel(const char* scene, const uint32_t* masks, const float* rays, int n, float* out)
{
    const SceneView S = make_view(scene, reinterpret_cast<const DevSceneHeader*>(scene), masks);
    const int k = (int)(blockIdx.x * 64u + threadIdx.x);
    const bool live = k < n;
    const float* r = rays + (size_t)(live ? k : 0) * 8;
    const f3 ro = mk3(r[0], r[1], r[2]), rd = mk3(r[3], r[4], r[5]);
    const float limit = r[6];
    const int nt = S.h->n_torus;
    int i = (int)r[7];
    i = i < 0 ? 0 : (i >= nt ? nt - 1 : i);
    float o[12];
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
    TexTable TT;
    memset(&TT, 0, sizeof TT);
    LaneCounters cnt;
    memset(&cnt, 0, sizeof cnt);
    o[4] = in_shadow<true, false, true>(S, TT, live, live, ro, rd, limit, cnt, -1);
    o[5] = in_shadow<false, false, true>(S, TT, live, live, ro, rd, limit, cnt, -1);
    int num = -1, type = -1;
    o[6] = calc_inter<true, false, true>(S, ro, rd, num, type, cnt, -1); o[7] = (float)num; o[8] = (float)type;
    num = -1; type = -1;
    o[9] = calc_inter<false, false, true>(S, ro, rd, num, type, cnt, -1); o[10] = (float)num; o[11] = (float)type;
    if (live) for (int j = 0; j < 12; j++) out[(size_t)k * 12 + j] = o[j];
}
}  // namespace
extern "C" {

// rays: n x 8 floats (ro, rd, limit, torus index); out: n x 12 floats (probe_kernel). Returns 0, or -1 (cull_audit_error()).
__attribute__((visibility("default"))) int cull_audit_probe(const rtpack::Defines* d, const void* const* blocks, const uint64_t* sizes, const float* rays, int n, float* out)
{
    DeviceScene ds;
    if (upload_scene(d, blocks, sizes, ds) != 0) return -1;
    float *d_rays = nullptr, *d_out = nullptr;
    TRY(hipMalloc(&d_rays, (size_t)n * 8 * 4));
    TRY(hipMalloc(&d_out, (size_t)n * 12 * 4));
    TRY(hipMemcpy(d_rays, rays, (size_t)n * 8 * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(probe_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, 0, ds.scene, ds.masks, d_rays, n, d_out);
    TRY(hipGetLastError());
    TRY(hipDeviceSynchronize());
    TRY(hipMemcpy(out, d_out, (size_t)n * 12 * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_rays); (void)hipFree(d_out); (void)hipFree(ds.scene); (void)hipFree(ds.masks);
    return 0;
}

// One scene (the nine std140 blocks the boundary receives), one family, `rays` rays (rounded up to whole launches of 2^20 threads).
// counters: N_COUNTERS x uint64, summed into (not cleared); bad: up to max_bad records of BAD_FLOATS floats; returns the number of records
// written, or -1 (cull_audit_error()).
__attribute__((visibility("default"))) int cull_audit_run(const rtpack::Defines* d, const void* const* blocks, const uint64_t* sizes, int family, uint64_t rays, uint64_t seed,
                                                          uint64_t* counters, float* bad, int max_bad, double* seconds)
{
    DeviceScene ds;
    if (upload_scene(d, blocks, sizes, ds) != 0) return -1;
    char* d_scene = ds.scene;
    uint32_t* d_masks = ds.masks;
    unsigned long long* d_cnt = nullptr;
    float* d_bad = nullptr;
    unsigned int* d_nbad = nullptr;
    TRY(hipMalloc(&d_cnt, N_COUNTERS * 8));
    TRY(hipMemset(d_cnt, 0, N_COUNTERS * 8));
    TRY(hipMalloc(&d_bad, (size_t)(max_bad > 0 ? max_bad : 1) * BAD_FLOATS * 4));
    TRY(hipMalloc(&d_nbad, 4));
    TRY(hipMemset(d_nbad, 0, 4));
    AuditParams p;
    p.scene = d_scene; p.pencil_masks = d_masks; p.counters = d_cnt; p.bad = d_bad; p.n_bad = d_nbad; p.max_bad = max_bad; p.family = family;
    const uint64_t threads = 1ull << 20;                     // 4096 workgroups: 16 per CU
    p.iters = rays >= threads * 256ull ? 256 : (int)((rays + threads - 1) / threads);     // (a short run, as the -m gpu test asks for: one smaller launch)
    if (p.iters < 1) p.iters = 1;
    const uint64_t per_launch = threads * (uint64_t)p.iters;
    const uint64_t launches = (rays + per_launch - 1) / per_launch;
    hipEvent_t e0, e1;
    TRY(hipEventCreate(&e0));
    TRY(hipEventCreate(&e1));
    TRY(hipEventRecord(e0, 0));
    for (uint64_t l = 0; l < launches; l++) {
        p.seed = seed * 1000003ull + l;
        hipLaunchKernelGGL(audit_kernel, dim3((unsigned)(threads / 256)), dim3(256), 0, 0, p);
    }
    TRY(hipGetLastError());
    TRY(hipEventRecord(e1, 0));
    TRY(hipEventSynchronize(e1));
    float ms = 0.0f;
    TRY(hipEventElapsedTime(&ms, e0, e1));
    if (seconds) *seconds = ms * 1e-3;
    unsigned long long host_cnt[N_COUNTERS];
    TRY(hipMemcpy(host_cnt, d_cnt, sizeof host_cnt, hipMemcpyDeviceToHost));
    for (int k = 0; k < N_COUNTERS; k++) {
        if ((family == F_TORUS_MARGIN && (k == 20 || k == 21 || k == 43 || (k >= 34 && k < 40))) || (family == F_TORUS_LEAD && ((k >= 60 && k < 70) || (k >= 80 && k < 90)))) counters[k] = counters[k] > host_cnt[k] ? counters[k] : host_cnt[k];
        else counters[k] += host_cnt[k];
    }
    unsigned int nb = 0;
    TRY(hipMemcpy(&nb, d_nbad, 4, hipMemcpyDeviceToHost));
    const int wrote = (int)(nb < (unsigned)max_bad ? nb : (unsigned)max_bad);
    if (wrote > 0) TRY(hipMemcpy(bad, d_bad, (size_t)wrote * BAD_FLOATS * 4, hipMemcpyDeviceToHost));
    (void)hipFree(d_scene); (void)hipFree(d_masks); (void)hipFree(d_cnt); (void)hipFree(d_bad); (void)hipFree(d_nbad);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return wrote;
}

}  // extern "C"
