// rt_pack.h -- host-side packer: nine std140 uniform blocks (+ rt_defines) -> DevScene blob.
//
// Runs inside rtx_block_create / rtx_block_update / rtx_specialize (the reference's
// init_buffer / update_buffer / init_shaders, GLWrapper.cpp:232-277,365-386). Pure C++, no HIP.
// Derived per-primitive fields are computed with the tracer's own inline functions from
// rt_device.h compiled for the host with -ffp-contract=off, i.e. the same IEEE operations the
// kernel would otherwise repeat per ray.
#pragma once
#include <cmath>
#include <limits>
#include <cstdio>
#include <cstdlib>
#include <algorithm>
#include <cstring>
#include <string>
#include <vector>
#include <mutex>
#include <unordered_map>

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


// ---- bounding box of a quadric clipped by the reference's world-space clip box ------------------
// World form of the surface:  x^T A x + B^T x + C = 0  with A = R^T diag(a,b,c) R,
// B = w - 2 A p, C = p^T A p - w.p + f  (w = R^T (0,e,d), p = surface.pos).
// Clip box axes are either finite (set F) or +-FLT_MAX (set U). For every point v of the F-box the
// equation in the U coordinates u is  (u-u0(v))^T A_UU (u-u0(v)) = g(v); if A_UU is definite the
// solutions lie within sqrt(|g|/lambda_min) of u0(v). u0 is linear and g quadratic in v, so both
// are bounded over the box by centre value + first/second-order terms (rigorous, slightly loose).
// Returns false when the clipped surface is unbounded (A_UU indefinite/singular) -> no cull.
inline void sym_eig_minmax_abs(const double* M, int n, double& min_abs, bool& definite)
{
    // Jacobi rotations on a copy (n <= 3)
    double a[3][3] = {{0}};
    for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) a[i][j] = M[i * n + j];
    for (int sweep = 0; sweep < 50; sweep++) {
        double off = 0.0;
        for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) off += a[i][j] * a[i][j];
        if (off < 1e-300) break;
        for (int pi = 0; pi < n; pi++)
            for (int qi = pi + 1; qi < n; qi++) {
                if (std::fabs(a[pi][qi]) < 1e-300) continue;
                const double theta = (a[qi][qi] - a[pi][pi]) / (2.0 * a[pi][qi]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < n; k++) { const double akp = a[k][pi], akq = a[k][qi]; a[k][pi] = cs * akp - sn * akq; a[k][qi] = sn * akp + cs * akq; }
                for (int k = 0; k < n; k++) { const double apk = a[pi][k], aqk = a[qi][k]; a[pi][k] = cs * apk - sn * aqk; a[qi][k] = sn * apk + cs * aqk; }
            }
    }
    int pos = 0, neg = 0;
    min_abs = 1e300;
    for (int i = 0; i < n; i++) {
        if (a[i][i] > 0) pos++; else if (a[i][i] < 0) neg++;
        min_abs = std::fmin(min_abs, std::fabs(a[i][i]));
    }
    definite = (pos == n || neg == n) && min_abs > 1e-12;
}
inline bool solve_sym(const double* M, int n, const double* rhs, double* x)  // M x = rhs, n <= 3, Gaussian elimination with pivoting
{
    double a[3][4];
    for (int i = 0; i < n; i++) { for (int j = 0; j < n; j++) a[i][j] = M[i * n + j]; a[i][n] = rhs[i]; }
    for (int c = 0; c < n; c++) {
        int piv = c;
        for (int r = c + 1; r < n; r++) if (std::fabs(a[r][c]) > std::fabs(a[piv][c])) piv = r;
        if (std::fabs(a[piv][c]) < 1e-300) return false;
        if (piv != c) for (int k = 0; k <= n; k++) std::swap(a[piv][k], a[c][k]);
        for (int r = 0; r < n; r++) if (r != c) { const double fct = a[r][c] / a[c][c]; for (int k = c; k <= n; k++) a[r][k] -= fct * a[c][k]; }
    }
    for (int i = 0; i < n; i++) x[i] = a[i][n] / a[i][i];
    return true;
}
// tau fattens the surface: the bounds are those of { |x^T A x + B^T x + C| <= tau } inside the clip box (see pack_scene: what the float
// evaluation of the quadratic can mistake for the surface). from_surface: some axis of the result rests on the surface's own extent.
inline bool quadric_clip_bounds(const double A[3][3], const double w[3], const double p[3], double f, const double lo[3], const double hi[3],
                                double out_lo[3], double out_hi[3], double tau, bool& from_surface)
{
    from_surface = false;
    const double big = 1.0e30;
    int U[3], F[3], nu = 0, nf = 0;
    for (int k = 0; k < 3; k++) {
        if (!(lo[k] == lo[k]) || !(hi[k] == hi[k]) || !(lo[k] < hi[k])) return false;
        if (std::fabs(lo[k]) < big && std::fabs(hi[k]) < big) F[nf++] = k; else U[nu++] = k;
    }
    for (int k = 0; k < nf; k++) { out_lo[F[k]] = lo[F[k]]; out_hi[F[k]] = hi[F[k]]; }
    if (nu == 0) return true;
    from_surface = true;
    for (int k = 0; k < nu; k++) if (std::fabs(lo[U[k]]) < big || std::fabs(hi[U[k]]) < big) return false;  // half-bounded axis: treat as unbounded
    double B[3], C = f;
    for (int r = 0; r < 3; r++) {
        double Ap = 0.0;
        for (int q = 0; q < 3; q++) Ap += A[r][q] * p[q];
        B[r] = w[r] - 2.0 * Ap;
        C += p[r] * Ap - w[r] * p[r];
    }
    double Auu[9], minabs;
    bool definite;
    for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++) Auu[i * nu + j] = A[U[i]][U[j]];
    sym_eig_minmax_abs(Auu, nu, minabs, definite);
    if (!definite) return false;
    const double sgn = Auu[0] > 0 ? 1.0 : -1.0;  // definite: sign of any diagonal entry
    // box centre / half widths of the finite axes
    double vc[3] = {0, 0, 0}, hv[3] = {0, 0, 0};
    for (int k = 0; k < nf; k++) { vc[k] = 0.5 * (lo[F[k]] + hi[F[k]]); hv[k] = 0.5 * (hi[F[k]] - lo[F[k]]); }
    // u0(v) = Mm v + m0 :  A_UU u0 = -(A_UF v + B_U/2)
    double Mm[3][3] = {{0}}, m0[3] = {0, 0, 0}, rhs[3], col[3];
    for (int i = 0; i < nu; i++) rhs[i] = -0.5 * B[U[i]];
    if (!solve_sym(Auu, nu, rhs, m0)) return false;
    for (int j = 0; j < nf; j++) {
        for (int i = 0; i < nu; i++) rhs[i] = -A[U[i]][F[j]];
        if (!solve_sym(Auu, nu, rhs, col)) return false;
        for (int i = 0; i < nu; i++) Mm[i][j] = col[i];
    }
    // g(v) = u0^T A_UU u0 - (v^T A_FF v + B_F^T v + C) = v^T Q v + L^T v + K
    double Q[3][3] = {{0}}, L[3] = {0, 0, 0}, K = -C;
    double Am0[3] = {0, 0, 0};
    for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++) Am0[i] += Auu[i * nu + j] * m0[j];
    for (int i = 0; i < nu; i++) K += m0[i] * Am0[i];
    for (int a_ = 0; a_ < nf; a_++) {
        double AMa[3] = {0, 0, 0};
        for (int i = 0; i < nu; i++) for (int j = 0; j < nu; j++) AMa[i] += Auu[i * nu + j] * Mm[j][a_];
        for (int b_ = 0; b_ < nf; b_++) {
            double acc = 0.0;
            for (int i = 0; i < nu; i++) acc += Mm[i][b_] * AMa[i];
            Q[b_][a_] = acc - A[F[b_]][F[a_]];
        }
        double l = 0.0;
        for (int i = 0; i < nu; i++) l += 2.0 * Mm[i][a_] * Am0[i];
        L[a_] = l - B[F[a_]];
    }
    // G = sgn * g ; bound max over the box: value at centre + |gradient| h + second-order terms
    double gc = K, grad[3] = {0, 0, 0};
    for (int a_ = 0; a_ < nf; a_++) {
        gc += L[a_] * vc[a_];
        grad[a_] = L[a_];
        for (int b_ = 0; b_ < nf; b_++) { gc += vc[a_] * Q[a_][b_] * vc[b_]; grad[a_] += (Q[a_][b_] + Q[b_][a_]) * vc[b_]; }
    }
    double Gmax = sgn * gc + tau;
    for (int a_ = 0; a_ < nf; a_++) {
        Gmax += std::fabs(grad[a_]) * hv[a_];
        for (int b_ = 0; b_ < nf; b_++) {
            const double qs = sgn * 0.5 * (Q[a_][b_] + Q[b_][a_]);
            Gmax += (a_ == b_ ? std::fmax(0.0, qs) : std::fabs(qs)) * hv[a_] * hv[b_];
        }
    }
    const double rad = Gmax > 0.0 ? std::sqrt(Gmax / minabs) : 0.0;  // Gmax <= 0: the surface misses the slab entirely
    for (int i = 0; i < nu; i++) {
        double c0 = m0[i], spread = 0.0;
        for (int j = 0; j < nf; j++) { c0 += Mm[i][j] * vc[j]; spread += std::fabs(Mm[i][j]) * hv[j]; }
        out_lo[U[i]] = c0 - spread - rad;
        out_hi[U[i]] = c0 + spread + rad;
    }
    return true;
}

// Round 3 -- where inside a (finite) box can the fattened surface { |F(x)| <= tau }, F(x) = (x-p)^T A (x-p) + w^T (x-p) + f, be at all?
// An octree over the box, three levels deep (8 x 8 x 8 leaves), keeps the cells that cannot be ruled out: for x = c + d in a cell with centre
// c and half widths h, F(x) = F(c) + grad F(c).d + d^T A d, so |F(c)| > sum |g_k| h_k + sum |A_ij| h_i h_j + tau leaves no point of the
// fattened surface in the cell (rigorous; evaluated in double). The clip box of a quadric says where a hit may lie, this says where the
// surface is: a cylinder of radius 0.5 in a 2.4-cube keeps a fifth of the cube's leaves, an ellipsoid of semi-axes 0.8 / 1.1 a tenth, and
// the bounding sphere of the kept leaves is what the first-level test needs. Collects the leaves' corner boxes in `leaves` (lo xyz, hi xyz).
struct QuadricCells {
    double A[3][3], w[3], p[3], f, tau;
    std::vector<double> leaves;
    bool may_hold(const double lo[3], const double hi[3]) const
    {
        double c[3], hw[3], v[3];
        for (int k = 0; k < 3; k++) { c[k] = 0.5 * (lo[k] + hi[k]); hw[k] = 0.5 * (hi[k] - lo[k]); v[k] = c[k] - p[k]; }
        double Fc = f, g[3], slack = tau;
        for (int r = 0; r < 3; r++) {
            double Av = 0.0;
            for (int q = 0; q < 3; q++) { Av += A[r][q] * v[q]; slack += std::fabs(A[r][q]) * hw[r] * hw[q]; }
            Fc += v[r] * Av + w[r] * v[r];
            g[r] = 2.0 * Av + w[r];
        }
        for (int k = 0; k < 3; k++) slack += std::fabs(g[k]) * hw[k];
        return !(std::fabs(Fc) > slack * (1.0 + 1e-9) + 1e-12);    // NaN: may hold
    }
    void descend(const double lo[3], const double hi[3], int depth)
    {
        if (!may_hold(lo, hi)) return;
        if (depth == 0) { leaves.insert(leaves.end(), lo, lo + 3); leaves.insert(leaves.end(), hi, hi + 3); return; }
        for (int o = 0; o < 8; o++) {
            double l[3], u[3];
            for (int k = 0; k < 3; k++) {
                const double m = 0.5 * (lo[k] + hi[k]);
                l[k] = (o >> k) & 1 ? m : lo[k];
                u[k] = (o >> k) & 1 ? hi[k] : m;
            }
            descend(l, u, depth - 1);
        }
    }
    // radius of the smallest sphere about c that holds every kept leaf (0 if none is kept)
    double radius_about(const double c[3]) const
    {
        double r2 = 0.0;
        for (size_t i = 0; i + 6 <= leaves.size(); i += 6) {
            double d2 = 0.0;
            for (int k = 0; k < 3; k++) { const double d = std::fmax(std::fabs(leaves[i + k] - c[k]), std::fabs(leaves[i + 3 + k] - c[k])); d2 += d * d; }
            r2 = std::fmax(r2, d2);
        }
        return std::sqrt(r2);
    }
    // The same to a resolution of `eps`, by branch and bound: always split the cell whose farthest corner is farthest from c, drop the
    // children that cannot hold the surface; when the farthest cell is smaller than eps its farthest corner bounds everything (a few hundred
    // cells; the 8 x 8 x 8 leaves alone would add up to their own diagonal to the radius).
    double radius_about_fine(const double c[3], const double lo[3], const double hi[3], double eps) const
    {
        struct Cell { double far2, lo[3], hi[3]; };
        auto far2_of = [&](const double l[3], const double u[3]) {
            double d2 = 0.0;
            for (int k = 0; k < 3; k++) { const double d = std::fmax(std::fabs(l[k] - c[k]), std::fabs(u[k] - c[k])); d2 += d * d; }
            return d2;
        };
        auto less = [](const Cell& x, const Cell& y) { return x.far2 < y.far2; };
        std::vector<Cell> heap;
        if (!may_hold(lo, hi)) return 0.0;
        Cell root;
        for (int k = 0; k < 3; k++) { root.lo[k] = lo[k]; root.hi[k] = hi[k]; }
        root.far2 = far2_of(lo, hi);
        heap.push_back(root);
        for (int guard = 0; guard < 20000 && !heap.empty(); guard++) {
            std::pop_heap(heap.begin(), heap.end(), less);
            const Cell top = heap.back();
            heap.pop_back();
            const double size = std::fmax(top.hi[0] - top.lo[0], std::fmax(top.hi[1] - top.lo[1], top.hi[2] - top.lo[2]));
            if (size <= eps) return std::sqrt(top.far2);
            for (int o = 0; o < 8; o++) {
                Cell ch;
                for (int k = 0; k < 3; k++) {
                    const double m = 0.5 * (top.lo[k] + top.hi[k]);
                    ch.lo[k] = (o >> k) & 1 ? m : top.lo[k];
                    ch.hi[k] = (o >> k) & 1 ? top.hi[k] : m;
                }
                if (!may_hold(ch.lo, ch.hi)) continue;
                ch.far2 = far2_of(ch.lo, ch.hi);
                heap.push_back(ch);
                std::push_heap(heap.begin(), heap.end(), less);
            }
        }
        if (heap.empty()) return 0.0;     // every cell was ruled out
        double r2 = 0.0;                  // guard hit: the cells still open bound the rest
        for (const Cell& x : heap) r2 = std::fmax(r2, x.far2);
        return std::sqrt(r2);
    }
    // a good centre when it can be chosen freely: start at the middle of the leaves' box and walk towards the farthest leaf corner while
    // that shrinks the radius (Ritter's idea; the radius is always recomputed exactly, so whatever the walk does the sphere holds the leaves)
    double free_sphere(double c[3]) const
    {
        double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
        for (size_t i = 0; i + 6 <= leaves.size(); i += 6)
            for (int k = 0; k < 3; k++) { lo[k] = std::fmin(lo[k], leaves[i + k]); hi[k] = std::fmax(hi[k], leaves[i + 3 + k]); }
        for (int k = 0; k < 3; k++) c[k] = 0.5 * (lo[k] + hi[k]);
        double best = radius_about(c);
        for (int it = 0; it < 24; it++) {
            double far[3] = {c[0], c[1], c[2]}, fd2 = -1.0;
            for (size_t i = 0; i + 6 <= leaves.size(); i += 6) {
                double q[3], d2 = 0.0;
                for (int k = 0; k < 3; k++) { q[k] = std::fabs(leaves[i + k] - c[k]) > std::fabs(leaves[i + 3 + k] - c[k]) ? leaves[i + k] : leaves[i + 3 + k]; d2 += (q[k] - c[k]) * (q[k] - c[k]); }
                if (d2 > fd2) { fd2 = d2; far[0] = q[0]; far[1] = q[1]; far[2] = q[2]; }
            }
            const double step = 0.5 / (it + 2.0);
            double t[3] = {c[0] + step * (far[0] - c[0]), c[1] + step * (far[1] - c[1]), c[2] + step * (far[2] - c[2])};
            const double r = radius_about(t);
            if (r < best) { best = r; c[0] = t[0]; c[1] = t[1]; c[2] = t[2]; }
        }
        return best;
    }
};

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
    h.cam_ident = quat_is_identity(h.cam_quat) ? 1u : 0u;
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
    auto pad = [](int n, int m) { return static_cast<size_t>((n + m - 1) / m * m) + m; };  // one spare group: batched loads may run past n
    h.off_sph_geom = reserve(sizeof(f4) * pad(d.sphere_size, 4));
    h.off_sph_hollow = reserve(sizeof(uint32_t) * (static_cast<size_t>(d.sphere_size) / 32 + 2));
    h.off_surf_cull = reserve(sizeof(DevSurfaceCull) * pad(d.surface_size, 2));
    h.off_torus_bound = reserve(sizeof(f4) * pad(d.torus_size, 4));
    h.off_ring_bound = reserve(sizeof(f4) * pad(d.ring_size, 4));
    h.off_surf_group = reserve(sizeof(f4) * pad(d.surface_size, RT_GROUP) / RT_GROUP + 16);
    h.off_torus_group = reserve(sizeof(f4) * pad(d.torus_size, RT_GROUP) / RT_GROUP + 16);
    // ray pencils (rt_scene_dev.h): the camera, then every point light, then every directional light -- as many as fit. Only for
    // scenes with a long quadric or torus table (and at most RT_PENCIL_MAX_PRIMS of each: four mask words per class).
    const bool long_tables = (d.surface_size >= RT_PENCIL_MIN_PRIMS || d.torus_size >= RT_PENCIL_MIN_PRIMS) &&
                             d.surface_size <= RT_PENCIL_MAX_PRIMS && d.torus_size <= RT_PENCIL_MAX_PRIMS;
    const int n_pencil = !long_tables ? 0 : (1 + d.light_point_size + d.light_direct_size < RT_MAX_PENCILS ? 1 + d.light_point_size + d.light_direct_size : RT_MAX_PENCILS);
    h.n_pencil = static_cast<uint32_t>(n_pencil);
    h.pencil_stride = static_cast<uint32_t>((d.surface_size + 31) / 32 + (d.torus_size + 31) / 32);
    h.off_pencil = reserve(sizeof(DevPencil) * (n_pencil + 2));
    h.pencil_dir = 0xffffffffu;
    const bool slabs = n_pencil > 0 && h.pencil_stride <= RT_SLAB_MAX_WORDS;
    h.off_slabs = slabs ? reserve(sizeof(DevSlabs)) : 0u;
    const uint32_t off_slab_table = slabs ? reserve(sizeof(uint32_t) * 3 * RT_SLAB_LEVELS * RT_SLABS * h.pencil_stride) : 0u;
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
        std::memcpy(reinterpret_cast<f4*>(blob.data() + h.off_sph_geom) + i, &s.geom, sizeof(f4));
        if (s.hollow) reinterpret_cast<uint32_t*>(blob.data() + h.off_sph_hollow)[i >> 5] |= 1u << (i & 31);
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
    std::vector<double> surf_aabb(static_cast<size_t>(d.surface_size) * 6, 0.0);   // lo xyz, hi xyz of the clipped surface (slab tables)
    for (int i = 0; i < d.surface_size; i++) {
        const unsigned char* p = blocks[BLK_SURFACES].data() + static_cast<size_t>(i) * SZ_SURFACE;
        DevSurface s;
        DevSurfaceCull sc;
        std::memset(&s, 0, sizeof s);
        std::memset(&sc, 0, sizeof sc);
        s.quat = rd4(p, 64);
        const f3 vmin = mk3(rdf(p, 80), rdf(p, 84), rdf(p, 88)), vmax = mk3(rdf(p, 96), rdf(p, 100), rdf(p, 104));
        const float a = rdf(p, 124), b = rdf(p, 128), c = rdf(p, 132), dd = rdf(p, 136), e = rdf(p, 140), f = rdf(p, 144);
        s.pos_a = mk4(rdf(p, 112), rdf(p, 116), rdf(p, 120), a);
        s.bcde = mk4(b, c, dd, e);
        s.f_vmin = mk4(f, vmin.x, vmin.y, vmin.z);
        s.vmax = mk4(vmax.x, vmax.y, vmax.z, int_bits(quat_is_identity(s.quat) ? 1 : 0));
        s.qinv = quat_inv(s.quat);
        // cull data (surface_cull in rt_device.h): symmetric M for the p2 pre-check + a bounding
        // sphere of the part of the surface inside the world-space clip box (may not exist)
        {
            const f3 ex = quat_rotate(s.quat, mk3(1, 0, 0)), ey = quat_rotate(s.quat, mk3(0, 1, 0)), ez = quat_rotate(s.quat, mk3(0, 0, 1));
            const double Rm[3][3] = {{ex.x, ey.x, ez.x}, {ex.y, ey.y, ez.y}, {ex.z, ey.z, ez.z}};  // Rm[k][j]: local k <- world j
            const double dg[3] = {a, b, c};
            const double ql[3] = {0.0, e, dd};  // linear local coefficients (x: none, y: e, z: d)
            const double pw[3] = {rdf(p, 112), rdf(p, 116), rdf(p, 120)};
            double A[3][3], wv[3];
            for (int r = 0; r < 3; r++) {
                for (int q = 0; q < 3; q++) {
                    double acc = 0.0;
                    for (int k = 0; k < 3; k++) acc += Rm[k][r] * dg[k] * Rm[k][q];
                    A[r][q] = acc;
                }
                wv[r] = Rm[0][r] * ql[0] + Rm[1][r] * ql[1] + Rm[2][r] * ql[2];
            }
            sc.sym0 = mk4(static_cast<float>(A[0][0]), static_cast<float>(A[0][1]), static_cast<float>(A[0][2]), static_cast<float>(A[1][1]));
            const float margin = 1e-6f + 1e-5f * (std::fabs(a) + std::fabs(b) + std::fabs(c));
            sc.sym1 = mk4(static_cast<float>(A[1][2]), static_cast<float>(A[2][2]), margin, 0.0f);
            // The bounds are computed in the quadric's OWN frame of translation (position at the origin, clip box relative to it) and moved to
            // the position afterwards: they are translation-covariant, and a quadric that only moves between two frames -- the usual
            // animation -- then finds them in the cache below (keyed on the whole record, every moved or re-coloured quadric
            // missed and paid 0.1-0.5 ms of branch and bound inside rtx_draw). Unbounded clip planes (|v| >= 1e30) stay where they are.
            const double pw_abs[3] = {pw[0], pw[1], pw[2]};
            // The relative box is rounded OUTWARDS to 16 significant bits: `vmin = pos - h` computed in float differs from one position to the
            // next in its last bits, and so would the key; a box that only grows keeps every bound valid (the clipped surface of the true
            // box lies inside the clipped surface of the larger one) and costs 1.5e-5 of the box' size.
            auto rel = [&](double v, int k, bool up) {
                if (!(std::fabs(v) < 1.0e30)) return v;
                const double r = v - pw_abs[k];
                const double g = std::ldexp(1.0, std::ilogb(std::fmax(std::fabs(r), 1.0e-3)) - 16);
                return (up ? std::ceil(r / g) : std::floor(r / g)) * g;
            };
            const double lo[3] = {rel(vmin.x, 0, false), rel(vmin.y, 1, false), rel(vmin.z, 2, false)}, hi[3] = {rel(vmax.x, 0, true), rel(vmax.y, 1, true), rel(vmax.z, 2, true)};
            const double pw0[3] = {0.0, 0.0, 0.0};
            double clo[3], chi[3];
            sc.bound = mk4(0.0f, 0.0f, 0.0f, -1.0f);
            sc.sym1.w = std::numeric_limits<float>::quiet_NaN();    // no bound that holds for origins at any distance (yet)
            bool from_surface = false;
            // The bounds are a pure function of the rotation, the six coefficients and the clip box relative to the position, and cost
            // 0.1-0.5 ms of branch and bound: a program that re-uploads all its blocks every frame (the reference's main loop does,
            // main.cpp:246) gets them from a small cache. The material and the position are not part of the key.
            struct CachedBounds { double c[3], tight2, far_w; double aabb[6]; bool has; };
            static std::mutex cache_mu;
            static std::unordered_map<std::string, CachedBounds> cache;
            struct KeyBytes { float quat[4], coef[6]; double lo[3], hi[3]; } kb;
            std::memset(&kb, 0, sizeof kb);
            kb.quat[0] = s.quat.x; kb.quat[1] = s.quat.y; kb.quat[2] = s.quat.z; kb.quat[3] = s.quat.w;
            kb.coef[0] = a; kb.coef[1] = b; kb.coef[2] = c; kb.coef[3] = dd; kb.coef[4] = e; kb.coef[5] = f;
            for (int k = 0; k < 3; k++) { kb.lo[k] = lo[k]; kb.hi[k] = hi[k]; }
            const std::string cache_key(reinterpret_cast<const char*>(&kb), sizeof kb);
            bool cached = false;
            {
                std::lock_guard<std::mutex> g(cache_mu);
                const auto it = cache.find(cache_key);
                if (it != cache.end()) {
                    cached = true;
                    const CachedBounds& cb = it->second;
                    if (cb.has) {
                        sc.bound = mk4(static_cast<float>(cb.c[0] + pw_abs[0]), static_cast<float>(cb.c[1] + pw_abs[1]), static_cast<float>(cb.c[2] + pw_abs[2]), static_cast<float>(cb.tight2));
                        sc.sym1.w = static_cast<float>(cb.far_w);
                        for (int k = 0; k < 6; k++) surf_aabb[i * 6 + k] = cb.aabb[k] + pw_abs[k % 3];
                    }
                }
            }
            CachedBounds fresh;
            std::memset(&fresh, 0, sizeof fresh);
            if (!cached && quadric_clip_bounds(A, wv, pw0, static_cast<double>(f), lo, hi, clo, chi, 0.0, from_surface)) {
                auto sphere = [&](double& cx, double& cy, double& cz) {
                    cx = 0.5 * (clo[0] + chi[0]); cy = 0.5 * (clo[1] + chi[1]); cz = 0.5 * (clo[2] + chi[2]);
                    const double hx = 0.5 * (chi[0] - clo[0]), hy = 0.5 * (chi[1] - clo[1]), hz = 0.5 * (chi[2] - clo[2]);
                    return std::sqrt(hx * hx + hy * hy + hz * hz) * 1.01 + 0.01;
                };
                double cx, cy, cz, rad = sphere(cx, cy, cz);
                bool ok = true;
                double far2 = std::numeric_limits<double>::infinity();
                if (from_surface) {
                    // A clip box that is open along some axis leaves it to the SURFACE to end the piece, and the reference evaluates the
                    // surface in float: at its computed hit point x the quadratic is not 0 but anything up to
                    //     |F(x)| <~ 8 * 2^-24 * (|a| + |b| + |c| + |d| + |e| + |f|) * (|o| + t + 1)^2        (o, t: local origin, distance)
                    // -- from 3000 units away a cylinder of radius 0.5 is "hit" by rays that pass 0.8 units beside it, and where its
                    // axis runs nearly parallel to the open slab such a hit lies hundreds of units beyond the end of the true piece (found
                    // by the pencil-scene fuzz: camera at z = -3000). The bound is therefore taken of the FATTENED surface |F| <= tau for
                    // origins up to `far` units from the bound's centre, and the cull stands aside beyond that distance (sym1.w).
                    // A closed clip box needs none of this: a hit must lie in the box, the box lies in the sphere.
                    const double coef = std::fabs(a) + std::fabs(b) + std::fabs(c) + std::fabs(dd) + std::fabs(e) + std::fabs(f);
                    const double far = RT_QUADRIC_FAR;
                    for (int pass = 0; pass < 3 && ok; pass++) {     // the fattened piece is larger, which raises tau a little: iterate
                        const double reach = 2.0 * (far + rad) + 3.0 * std::sqrt(cx * cx + cy * cy + cz * cz) + 1.0;
                        const double tau = 64.0 / 16777216.0 * coef * reach * reach;     // the estimate above with the rotation into the local frame and
                                                                                 // the solve on top (~22 * 2^-24), times three
                        ok = quadric_clip_bounds(A, wv, pw0, static_cast<double>(f), lo, hi, clo, chi, tau, from_surface);
                        if (ok) rad = sphere(cx, cy, cz);
                    }
                    far2 = far * far;
                }
                if (ok && rad == rad && rad < 1.0e15) {
                    // (cx, cy, cz, rad): the sphere of the box [clo, chi] the clipped surface lies in. If the clip box is closed this bound
                    // holds whatever the reference's arithmetic does and wherever the ray starts (sym1.w); `tight` is the sphere of the
                    // part of that box the (fattened) surface can be in at all -- for origins within RT_QUADRIC_FAR (bound.w).
                    double tight = rad, tc[3] = {cx, cy, cz};
                    if (rad < 1.0e6) {
                        QuadricCells cells;
                        for (int r = 0; r < 3; r++) { for (int q = 0; q < 3; q++) cells.A[r][q] = A[r][q]; cells.w[r] = wv[r]; cells.p[r] = 0.0; }
                        cells.f = f;
                        const double coef = std::fabs(a) + std::fabs(b) + std::fabs(c) + std::fabs(dd) + std::fabs(e) + std::fabs(f);
                        const double reach = 2.0 * (RT_QUADRIC_FAR + rad) + 3.0 * std::sqrt(cx * cx + cy * cy + cz * cz) + 1.0;
                        cells.tau = 64.0 / 16777216.0 * coef * reach * reach;     // as above: what the float evaluation can mistake for the surface
                        if (from_surface) cells.descend(clo, chi, 3);     // the leaves choose the centre; the radius is refined below
                        if (!cells.may_hold(clo, chi) || (from_surface && cells.leaves.empty())) {
                            tight = 0.0;                 // the surface does not reach into its clip box at all
                        } else if (from_surface) {
                            double c2[3];
                            cells.free_sphere(c2);
                            const double r2 = cells.radius_about_fine(c2, clo, chi, 0.02 * rad) * 1.01 + 0.01;
                            if (r2 < rad) { tight = r2; tc[0] = c2[0]; tc[1] = c2[1]; tc[2] = c2[2]; }
                        } else {
                            tight = std::fmin(rad, cells.radius_about_fine(tc, clo, chi, 0.02 * rad) * 1.01 + 0.01);     // same centre: the far bound stays the clip box' own
                        }
                    }
                    if (!(tight == tight)) tight = rad;
                    fresh.has = true;
                    for (int k = 0; k < 3; k++) { fresh.c[k] = tc[k]; fresh.aabb[k] = clo[k]; fresh.aabb[3 + k] = chi[k]; }
                    fresh.tight2 = tight * tight * (1.0 + 1e-6);
                    fresh.far_w = from_surface ? std::numeric_limits<double>::quiet_NaN() : rad * rad;
                    (void)far2;
                    sc.bound = mk4(static_cast<float>(tc[0] + pw_abs[0]), static_cast<float>(tc[1] + pw_abs[1]), static_cast<float>(tc[2] + pw_abs[2]), static_cast<float>(fresh.tight2));
                    sc.sym1.w = static_cast<float>(fresh.far_w);
                    for (int k = 0; k < 6; k++) surf_aabb[i * 6 + k] = fresh.aabb[k] + pw_abs[k % 3];
                }
            }
            if (!cached) {
                std::lock_guard<std::mutex> g(cache_mu);
                if (cache.size() >= 4096) {       // drop half, not all: no frame pays for every quadric at once
                    size_t n = 0;
                    for (auto it = cache.begin(); it != cache.end();) { if ((n++ & 1u) == 0u) it = cache.erase(it); else ++it; }
                }
                cache.emplace(cache_key, fresh);
            }
        }
        std::memcpy(reinterpret_cast<DevSurface*>(blob.data() + h.off_surface) + i, &s, sizeof s);
        std::memcpy(reinterpret_cast<DevSurfaceCull*>(blob.data() + h.off_surf_cull) + i, &sc, sizeof sc);
        std::memcpy(mat_at(TYPE_SURFACE, i), p, 64);
    }
    for (int i = 0; i < d.box_size; i++) {
        const unsigned char* p = blocks[BLK_BOXES].data() + static_cast<size_t>(i) * SZ_BOX;
        DevBox s;
        s.quat = rd4(p, 64);
        s.pos = rd3(p, 80, int_bits(quat_is_identity(s.quat) ? 1 : 0));
        s.form_tex = rd3(p, 96, int_bits(rdi(p, 108)));
        s.qinv = quat_inv(s.quat);
        std::memcpy(reinterpret_cast<DevBox*>(blob.data() + h.off_box) + i, &s, sizeof s);
        std::memcpy(mat_at(TYPE_BOX, i), p, 64);
    }
    for (int i = 0; i < d.torus_size; i++) {
        const unsigned char* p = blocks[BLK_TORUSES].data() + static_cast<size_t>(i) * SZ_TORUS;
        DevTorus s;
        s.quat = rd4(p, 64);
        s.pos = rd3(p, 80, int_bits(quat_is_identity(s.quat) ? 1 : 0));
        const float R = rdf(p, 96), r = rdf(p, 100);
        const float R2 = R * R, r2 = r * r;
        s.radii = mk4(R, r, R2, r2);
        // The inflated tube every cull tests against: radius rinf = max(1.01 |r| + 0.01, sqrt(r^2 + RT_TORUS_IM_NOISE^2)). The second term (round 6):
        // a ray that clears a tube of radius r by delta has a complex root pair with |Im| = sqrt(delta (2 r + delta)), and the reference's solver,
        // run from tens of units away, reports an iterate as real (|Im| <= 1e-3) although the pair's imaginary part is up to ~0.05 (rt_device.h
        // RT_TORUS_IM_NOISE) -- for a thin tube that is a clearance of several centimetres, more than the 1 cm + 1 % of rounds 2-5, which had
        // only been audited on tori of R 0.3 .. 2 (tests/random_scenes.py sized_torus_scene found it). For r >= 0.3 the first term still governs.
        const double ar = std::fabs(static_cast<double>(r)), aR = std::fabs(static_cast<double>(R));
        const double rinf = std::fmax(ar * 1.01 + 0.01, std::sqrt(ar * ar + static_cast<double>(RT_TORUS_IM_NOISE) * RT_TORUS_IM_NOISE));
        const double rb = aR * 1.01 + rinf + 0.01 * ar;              // bounding sphere / puck radius: >= (|R| + |r|) 1.01 + 0.01
        const double hole = aR * 0.99 - rinf - 0.01 * ar;            // <= (|R| - |r|) 0.99 - 0.01
        s.k = mk4(4.0f * R2, static_cast<float>(rb * rb), static_cast<float>(rb * rb), hole > 0.0 ? static_cast<float>(hole * hole) : 0.0f);
        s.qinv = quat_inv(s.quat);
        // y, z: the convex-hull cull of torus_local_cull -- (|r| + margin)^2 and |R|
        // (round 6: + 3e-6 / min(|r|, |R|). The iteration stops once a sweep moves every iterate by less than 1e-3; its last, quadratic step then leaves an
        // error of about 1e-6 / (distance to the next root), and the next root of a ray that has just left the surface is about one tube width
        // away -- 1e-5 for r = 0.3, inside the margin, but 3.6e-4 for a tube of r = 0.0028, whose own shadow rays then "hit" it at t = 1e-5 .. 3e-4
        // from 2.5e-4 outside the hull: 18 rays in 2e10 on one torus of tests/random_scenes.py sized_torus_scene(14).)
        const double hull = ar + RT_TORUS_HULL_MARGIN + 3.0e-6 / std::fmin(ar, aR);
        s.cull = mk4(static_cast<float>(rinf), static_cast<float>(hull * hull), std::fabs(R), static_cast<float>(hull));
        // The culls rest on "a geometric miss makes Durand-Kerner report no root". That holds for tori with a real tube
        // (validated on random rays), but not for degenerate ones: with tube radius 0 the solver, out of sweeps, can stop
        // on an iterate whose imaginary part happens to be below 1e-3 far away from the (zero-thickness) torus -- found by
        // the nasty-scene fuzz. Such tori (tube thinner than 2 % of the major radius, or any non-positive / non-finite
        // radius) get infinite bounds: never culled, the solver decides like in the reference.
        // Round 6: and only tori of the SIZES the premises were measured on are culled at all. Every torus premise is a statement about the reference's
        // float iteration, audited at 1e11 .. 1e12 rays on tori of R 0.3 .. 2, r 0.1 .. 1.5 (the bench scenes and tests/random_scenes.py). Outside that
        // range the same audits, on tests/random_scenes.py sized_torus_scene, meet what the arithmetic predicts: a tube of a few millimetres seen from 30
        // units has its complex root pairs taken for real although the ray clears it by 9 cm (|Im| 0.1); a torus of R = 9 reports a root at t = 0.009 on
        // a ray that has just left its surface; a spindle torus of R = r = 17 throws phantoms at rays from inside 1.25 bounding radii. Such tori are
        // solved for every ray, like in the reference.
        const bool audited_size = R >= RT_TORUS_CULL_R_MIN && R <= RT_TORUS_CULL_R_MAX && r >= RT_TORUS_CULL_TUBE_MIN && r <= RT_TORUS_CULL_TUBE_MAX;
        const bool real_tube = std::isfinite(R) && std::isfinite(r) && R > 0.0f && r > 0.02f * R && audited_size;
        // rotate() (rt.frag:306-311) multiplies by q and conj(q), not by the inverse: a quaternion of squared norm n2 also
        // SCALES the ray by n2, so in world space the torus is 1/n2 times as large as its radii say and, worse, the
        // direction the solver sees is not a unit vector (its roots are then not geometric, rt_device.h unit_direction).
        // Bounds are only valid for unit quaternions; any other torus is never culled. (The fuzzers used to draw unit
        // quaternions only; q = (0,0,0,0.9) changed 540 pixels of a 96x64 frame between culls on and off.)
        const double qn2 = static_cast<double>(s.quat.x) * s.quat.x + static_cast<double>(s.quat.y) * s.quat.y +
                           static_cast<double>(s.quat.z) * s.quat.z + static_cast<double>(s.quat.w) * s.quat.w;
        const bool unit_quat = std::fabs(qn2 - 1.0) <= 1e-4;   // false for NaN; 1e-4 is far inside the 1 % inflation
        if (!real_tube || !unit_quat) {
            const float inf = std::numeric_limits<float>::infinity();
            s.k.y = inf; s.k.z = inf; s.k.w = 0.0f;
            s.cull.x = inf; s.cull.y = inf;
        }
        std::memcpy(reinterpret_cast<DevTorus*>(blob.data() + h.off_torus) + i, &s, sizeof s);
        const f4 tb = mk4(s.pos.x, s.pos.y, s.pos.z, s.k.z);
        std::memcpy(reinterpret_cast<f4*>(blob.data() + h.off_torus_bound) + i, &tb, sizeof tb);
        std::memcpy(mat_at(TYPE_TORUS, i), p, 64);
    }
    for (int i = 0; i < d.ring_size; i++) {
        const unsigned char* p = blocks[BLK_RINGS].data() + static_cast<size_t>(i) * SZ_RING;
        DevRing s;
        s.quat = rd4(p, 64);
        s.pos_tex = rd3(p, 80, int_bits(rdi(p, 92)));
        const float r1 = rdf(p, 96), r2 = rdf(p, 100);
        // rotate() scales by the quaternion's squared norm n2 (see the torus above): a local squared radius r2 is a world
        // radius sqrt(r2)/n2. NaN for a negative r2 or a zero / non-finite quaternion: never culled.
        const double ring_qn2 = static_cast<double>(s.quat.x) * s.quat.x + static_cast<double>(s.quat.y) * s.quat.y +
                                static_cast<double>(s.quat.z) * s.quat.z + static_cast<double>(s.quat.w) * s.quat.w;
        double ring_rb = std::sqrt(static_cast<double>(r2)) / ring_qn2 * 1.001 + 0.01;
        if (!(ring_qn2 > 1e-12) || !std::isfinite(ring_qn2) || !std::isfinite(ring_rb) || ring_rb > 1e15) ring_rb = std::numeric_limits<double>::quiet_NaN();
        s.radii = mk4(r1, r2, r2 - r1, static_cast<float>(ring_rb * ring_rb));
        const f3 nrm = quat_rotate(quat_inv(s.quat), mk3(0.0f, 0.0f, -1.0f));
        s.normal = mk4(nrm.x, nrm.y, nrm.z, int_bits(quat_is_identity(s.quat) ? 1 : 0));
        std::memcpy(reinterpret_cast<DevRing*>(blob.data() + h.off_ring) + i, &s, sizeof s);
        const f4 rbnd = mk4(s.pos_tex.x, s.pos_tex.y, s.pos_tex.z, s.radii.w);
        std::memcpy(reinterpret_cast<f4*>(blob.data() + h.off_ring_bound) + i, &rbnd, sizeof rbnd);
        std::memcpy(mat_at(TYPE_RING, i), p, 64);
    }
    // second-level bounds: a sphere around the first-level bounds of every RT_GROUP consecutive quadrics / tori (rt_device.h group_cull).
    // Conservative by construction: it contains each member's (already inflated) bound, so a ray that provably misses it provably misses
    // every member's bound. A member without a finite bound makes its group unbounded (never culled).
    auto group_bounds = [&](int count, uint32_t off_group, auto member_bound) {
        for (int g = 0; g * RT_GROUP < count; g++) {
            const int i0 = g * RT_GROUP, i1 = i0 + RT_GROUP < count ? i0 + RT_GROUP : count;
            double cx = 0, cy = 0, cz = 0;
            bool bounded = true;
            for (int i = i0; i < i1; i++) {
                const f4 b = member_bound(i);
                if (!(b.w >= 0.0f) || !std::isfinite(b.w) || !std::isfinite(b.x) || !std::isfinite(b.y) || !std::isfinite(b.z)) bounded = false;
                cx += b.x; cy += b.y; cz += b.z;
            }
            f4 out = mk4(0.0f, 0.0f, 0.0f, -1.0f);
            if (bounded) {
                cx /= (i1 - i0); cy /= (i1 - i0); cz /= (i1 - i0);
                double rad = 0.0;
                for (int i = i0; i < i1; i++) {
                    const f4 b = member_bound(i);
                    const double dx = b.x - cx, dy = b.y - cy, dz = b.z - cz;
                    rad = std::fmax(rad, std::sqrt(dx * dx + dy * dy + dz * dz) + std::sqrt(static_cast<double>(b.w)));
                }
                rad = rad * 1.0001 + 1e-3;
                if (rad < 1.0e15) out = mk4(static_cast<float>(cx), static_cast<float>(cy), static_cast<float>(cz), static_cast<float>(rad * rad));
            }
            std::memcpy(reinterpret_cast<f4*>(blob.data() + off_group) + g, &out, sizeof out);
        }
    };
    group_bounds(d.surface_size, h.off_surf_group, [&](int i) {
        const DevSurfaceCull* q = reinterpret_cast<const DevSurfaceCull*>(blob.data() + h.off_surf_cull) + i;
        return q->sym1.w >= 0.0f && q->bound.w >= 0.0f ? mk4(q->bound.x, q->bound.y, q->bound.z, q->sym1.w) : mk4(0.0f, 0.0f, 0.0f, -1.0f);     // the bound that holds at any distance, or no group cull
    });
    group_bounds(d.torus_size, h.off_torus_group, [&](int i) { return reinterpret_cast<const f4*>(blob.data() + h.off_torus_bound)[i]; });
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
        const f3 ln = normalize3(-xyz(s.direction));
        s.dir_n = mk4(ln.x, ln.y, ln.z, 0.0f);
        std::memcpy(reinterpret_cast<DevLightDirect*>(blob.data() + h.off_light_direct) + i, &s, sizeof s);
    }
    // pencil headers (the masks are built on the device from these and the cull records above)
    {
        uint32_t mask_words = 0;
        auto finite3 = [](f4 v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); };
        auto small3 = [&](f4 v) { return finite3(v) && std::fabs(v.x) <= 1.0e3f && std::fabs(v.y) <= 1.0e3f && std::fabs(v.z) <= 1.0e3f; };
        for (int k = 0; k < n_pencil; k++) {
            DevPencil P;
            std::memset(&P, 0, sizeof P);
            P.kind = RT_PENCIL_OFF;
            if (k == 0 || k - 1 < d.light_point_size) {
                const f4 apex = k == 0 ? h.cam_pos : (reinterpret_cast<const DevLightPoint*>(blob.data() + h.off_light_point) + (k - 1))->pos_r2;
                if (small3(apex)) {   // a far-away apex loses the float precision the builder's margins assume: no pencil then
                    P.kind = RT_PENCIL_APEX;
                    P.a = mk4(apex.x, apex.y, apex.z, k == 0 ? 0.0f : 1.0f);   // w != 0: the rays run TOWARDS the apex (a light), rt_device.h pencil_cell_word
                    P.res = RT_PENCIL_APEX_RES;
                    P.cells = 6u * RT_PENCIL_APEX_RES * RT_PENCIL_APEX_RES;
                }
            } else {
                const f4 dn = (reinterpret_cast<const DevLightDirect*>(blob.data() + h.off_light_direct) + (k - 1 - d.light_point_size))->dir_n;
                const double ax = dn.x, ay = dn.y, az = dn.z, n2 = ax * ax + ay * ay + az * az;
                if (finite3(dn) && std::fabs(n2 - 1.0) <= 1e-5) {
                    // e1, e2: an orthonormal pair across the direction, in double
                    double bx = 0, by = 0, bz = 0;
                    if (std::fabs(ax) <= std::fabs(ay) && std::fabs(ax) <= std::fabs(az)) bx = 1; else if (std::fabs(ay) <= std::fabs(az)) by = 1; else bz = 1;
                    double e1x = ay * bz - az * by, e1y = az * bx - ax * bz, e1z = ax * by - ay * bx;
                    const double l1 = std::sqrt(e1x * e1x + e1y * e1y + e1z * e1z);
                    e1x /= l1; e1y /= l1; e1z /= l1;
                    double e2x = ay * e1z - az * e1y, e2y = az * e1x - ax * e1z, e2z = ax * e1y - ay * e1x;
                    const double l2 = std::sqrt(e2x * e2x + e2y * e2y + e2z * e2z);
                    e2x /= l2; e2y /= l2; e2z /= l2;
                    // extent of the bounded quadrics / tori in that plane
                    double ulo = 1e300, uhi = -1e300, vlo = 1e300, vhi = -1e300;
                    auto add = [&](f4 b) {
                        if (!(b.w >= 0.0f) || !std::isfinite(b.w) || !finite3(b)) return;
                        const double r = std::sqrt(static_cast<double>(b.w)), u = b.x * e1x + b.y * e1y + b.z * e1z, v = b.x * e2x + b.y * e2y + b.z * e2z;
                        ulo = std::fmin(ulo, u - r); uhi = std::fmax(uhi, u + r); vlo = std::fmin(vlo, v - r); vhi = std::fmax(vhi, v + r);
                    };
                    for (int i = 0; i < d.surface_size; i++) add((reinterpret_cast<const DevSurfaceCull*>(blob.data() + h.off_surf_cull) + i)->bound);
                    for (int i = 0; i < d.torus_size; i++) add(reinterpret_cast<const f4*>(blob.data() + h.off_torus_bound)[i]);
                    if (!(ulo <= uhi)) { ulo = vlo = -1.0; uhi = vhi = 1.0; }
                    const int R = RT_PENCIL_PAR_RES;
                    const double su = std::fmax((uhi - ulo) / (R - 2), 1e-3), sv = std::fmax((vhi - vlo) / (R - 2), 1e-3);   // cells 1 .. R-2 tile the extent
                    if (std::isfinite(su) && std::isfinite(sv) && su < 1e6 && sv < 1e6 && std::fabs(ulo) < 1e6 && std::fabs(vlo) < 1e6) {
                        P.kind = RT_PENCIL_PARALLEL;
                        P.a = mk4(dn.x, dn.y, dn.z, 0.0f);
                        P.e1 = mk4(static_cast<float>(e1x), static_cast<float>(e1y), static_cast<float>(e1z), static_cast<float>(ulo - su));
                        P.e2 = mk4(static_cast<float>(e2x), static_cast<float>(e2y), static_cast<float>(e2z), static_cast<float>(vlo - sv));
                        P.grid = mk4(static_cast<float>(1.0 / su), static_cast<float>(1.0 / sv), static_cast<float>(su), static_cast<float>(sv));
                        P.res = R;
                        P.cells = static_cast<uint32_t>(R) * R;
                    }
                }
            }
            P.mask_off = mask_words;
            if (P.kind != RT_PENCIL_OFF) mask_words += (P.cells + 1u) * h.pencil_stride;
            std::memcpy(reinterpret_cast<DevPencil*>(blob.data() + h.off_pencil) + k, &P, sizeof P);
        }
        DevSceneHeader* hp = reinterpret_cast<DevSceneHeader*>(blob.data());
        // slab tables + the direction table, for the rays of no pencil
        if (slabs) {
            const int W = static_cast<int>(h.pencil_stride), nws = (d.surface_size + 31) / 32;
            DevSlabs B;
            std::memset(&B, 0, sizeof B);
            auto bit = [&](uint32_t* words, int k) { words[k < d.surface_size ? k / 32 : nws + (k - d.surface_size) / 32] |= 1u << ((k < d.surface_size ? k : k - d.surface_size) & 31); };
            struct Box { double c[3], r[3]; };   // centre, half extents (padded like the bounding spheres: 1 % + 0.01)
            std::vector<Box> boxes(d.surface_size + d.torus_size);
            std::vector<char> bounded(boxes.size(), 0);
            double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
            for (int k = 0; k < d.surface_size + d.torus_size; k++) {
                bit(B.valid, k);
                f4 b;
                bool usable;
                if (k < d.surface_size) {
                    const DevSurfaceCull* q = reinterpret_cast<const DevSurfaceCull*>(blob.data() + h.off_surf_cull) + k;
                    b = mk4(q->bound.x, q->bound.y, q->bound.z, q->sym1.w);
                    usable = q->bound.w >= 0.0f && q->sym1.w >= 0.0f;        // a bound that only holds near the quadric is no bound here
                } else {
                    b = reinterpret_cast<const f4*>(blob.data() + h.off_torus_bound)[k - d.surface_size];
                    usable = true;
                }
                usable = usable && b.w >= 0.0f && std::isfinite(b.w) && finite3(b) && std::fabs(b.x) < 1e6f && std::fabs(b.y) < 1e6f && std::fabs(b.z) < 1e6f && b.w < 1e12f;
                if (!usable) { bit(B.always, k); continue; }
                bounded[k] = 1;
                const double rs = std::sqrt(static_cast<double>(b.w));
                boxes[k] = {{b.x, b.y, b.z}, {rs, rs, rs}};
                if (k < d.surface_size) {      // the box the clipped surface lies in (a hit must lie strictly inside the clip box, rt.frag:500-512)
                    for (int a = 0; a < 3; a++) {
                        const double l = surf_aabb[k * 6 + a], u = surf_aabb[k * 6 + 3 + a];
                        boxes[k].c[a] = 0.5 * (l + u);
                        boxes[k].r[a] = std::fmin(rs + std::fabs(boxes[k].c[a] - b.x * (a == 0) - b.y * (a == 1) - b.z * (a == 2)), 0.5 * (u - l) * 1.01 + 0.01);
                    }
                } else {                       // a torus reaches R * sqrt(1 - n_a^2) + r along axis a (n: its axis)
                    const DevTorus* t = reinterpret_cast<const DevTorus*>(blob.data() + h.off_torus) + (k - d.surface_size);
                    const f3 n = quat_rotate(t->qinv, mk3(0.0f, 0.0f, 1.0f));
                    const double nv[3] = {n.x, n.y, n.z}, R = std::fabs(static_cast<double>(t->radii.x)), r = std::fabs(static_cast<double>(t->radii.y));
                    for (int a = 0; a < 3; a++) {
                        const double e = (R * std::sqrt(std::fmax(0.0, 1.0 - nv[a] * nv[a])) + r) * 1.01 + 0.01 + 1e-3 * R;   // (n is a float, unit within 1e-4)
                        if (e == e) boxes[k].r[a] = std::fmin(rs, e);
                    }
                }
                for (int a = 0; a < 3; a++) { lo[a] = std::fmin(lo[a], boxes[k].c[a] - boxes[k].r[a]); hi[a] = std::fmax(hi[a], boxes[k].c[a] + boxes[k].r[a]); }
            }
            if (!(lo[0] <= hi[0])) { for (int a = 0; a < 3; a++) { lo[a] = -1.0; hi[a] = 1.0; } }
            uint32_t* T = reinterpret_cast<uint32_t*>(blob.data() + off_slab_table);
            double size[3], step[3];
            for (int a = 0; a < 3; a++) {
                const double pad = 0.05 + 1e-3 * (hi[a] - lo[a]);      // hit points are computed in float: keep them inside
                lo[a] -= pad; hi[a] += pad;
                size[a] = hi[a] - lo[a];
                step[a] = size[a] / RT_SLABS;
            }
            B.lo = mk4(static_cast<float>(lo[0]), static_cast<float>(lo[1]), static_cast<float>(lo[2]), 0.0f);
            B.hi = mk4(static_cast<float>(hi[0]), static_cast<float>(hi[1]), static_cast<float>(hi[2]), 0.0f);
            B.inv = mk4(static_cast<float>(1.0 / step[0]), static_cast<float>(1.0 / step[1]), static_cast<float>(1.0 / step[2]), 0.0f);
            auto entry = [&](int a, int l, int i) { return T + ((static_cast<size_t>(a) * RT_SLAB_LEVELS + l) * RT_SLABS + i) * W; };
            for (int k = 0; k < d.surface_size + d.torus_size; k++) {
                if (!bounded[k]) continue;
                for (int a = 0; a < 3; a++) {
                    // the slab index of a point is computed in float by the kernel: pad the interval by more than that can be off
                    const double eps = 2e-3 + 2e-3 * step[a] + 1e-5 * (std::fabs(boxes[k].c[a]) + boxes[k].r[a]);
                    int i0 = static_cast<int>(std::floor((boxes[k].c[a] - boxes[k].r[a] - eps - lo[a]) / step[a]));
                    int i1 = static_cast<int>(std::floor((boxes[k].c[a] + boxes[k].r[a] + eps - lo[a]) / step[a]));
                    i0 = i0 < 0 ? 0 : (i0 > RT_SLABS - 1 ? RT_SLABS - 1 : i0);
                    i1 = i1 < 0 ? 0 : (i1 > RT_SLABS - 1 ? RT_SLABS - 1 : i1);
                    for (int i = i0; i <= i1; i++) bit(entry(a, 0, i), k);
                }
            }
            for (int a = 0; a < 3; a++)
                for (int l = 1; l < RT_SLAB_LEVELS; l++)
                    for (int i = 0; i < RT_SLABS; i++) {
                        const int j = i + (1 << (l - 1)) < RT_SLABS ? i + (1 << (l - 1)) : RT_SLABS - 1;
                        for (int w = 0; w < W; w++) entry(a, l, i)[w] = entry(a, l - 1, i)[w] | entry(a, l - 1, j)[w];
                    }
            B.table_off = off_slab_table;
            std::memcpy(blob.data() + h.off_slabs, &B, sizeof B);
            if (d.surface_size > 0) {   // the direction table, behind the pencils
                DevPencil P;
                std::memset(&P, 0, sizeof P);
                P.kind = RT_PENCIL_DIRECTION;
                P.res = RT_PENCIL_DIR_RES;
                P.cells = 6u * RT_PENCIL_DIR_RES * RT_PENCIL_DIR_RES;
                P.mask_off = mask_words;
                mask_words += (P.cells + 1u) * h.pencil_stride;
                std::memcpy(reinterpret_cast<DevPencil*>(blob.data() + h.off_pencil) + n_pencil, &P, sizeof P);
                hp->pencil_dir = static_cast<uint32_t>(n_pencil);
            }
        }
        hp->pencil_mask_words = n_pencil > 0 ? mask_words + 4u : 0u;   // + spare words: the scans request one word ahead
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

// glGenerateMipmap stand-in (DESIGN.md "Texture rule"): appends levels 1.. to `texels` (level 0 on
// entry, RGBA8 dwords). Level L is max(1,w>>L) x max(1,h>>L); each texel is the rounded integer mean
// ((a+b+c+d+2)>>2 per channel) of source texels (min(2i,ws-1), min(2i+1,ws-1)) x (same in j).
// Returns the level count and fills the dword offset of every level.
inline int build_mip_chain(std::vector<uint32_t>& texels, int w, int h, uint32_t* level_off, int max_levels)
{
    int levels = 1;
    size_t src_off = 0;
    level_off[0] = 0;
    while ((w > 1 || h > 1) && levels < max_levels) {
        const int ws = w, hs = h;
        w = w > 1 ? w >> 1 : 1;
        h = h > 1 ? h >> 1 : 1;
        const size_t dst_off = texels.size();
        texels.resize(dst_off + static_cast<size_t>(w) * h);
        level_off[levels] = static_cast<uint32_t>(dst_off);
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                const int i0 = 2 * i < ws - 1 ? 2 * i : ws - 1, i1 = 2 * i + 1 < ws - 1 ? 2 * i + 1 : ws - 1;
                const int j0 = 2 * j < hs - 1 ? 2 * j : hs - 1, j1 = 2 * j + 1 < hs - 1 ? 2 * j + 1 : hs - 1;
                const uint32_t p00 = texels[src_off + static_cast<size_t>(j0) * ws + i0], p10 = texels[src_off + static_cast<size_t>(j0) * ws + i1];
                const uint32_t p01 = texels[src_off + static_cast<size_t>(j1) * ws + i0], p11 = texels[src_off + static_cast<size_t>(j1) * ws + i1];
                uint32_t out = 0;
                for (int c = 0; c < 4; c++) {
                    const uint32_t sum = ((p00 >> (8 * c)) & 255u) + ((p10 >> (8 * c)) & 255u) + ((p01 >> (8 * c)) & 255u) + ((p11 >> (8 * c)) & 255u);
                    out |= ((sum + 2u) >> 2) << (8 * c);
                }
                texels[dst_off + static_cast<size_t>(j) * w + i] = out;
            }
        src_off = dst_off;
        levels++;
    }
    return levels;
}

}  // namespace rtpack
