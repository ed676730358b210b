// rt_device.h -- the per-pixel tracer (device code of the MI355X path).
//
// Replaces the reference's fragment shader assets/shaders/rt.frag. This is NOT a transliteration:
// the shader's recursive-looking structure (main -> getReflectedColor -> calcShade -> inShadow,
// each with its own copy of the primitive scans) is rebuilt as ONE wave-uniform segment loop with
// a single closest-hit site and a single shading site; per-primitive constants are hoisted into
// the DevScene tables (rt_scene_dev.h); ray-invariant terms of the torus quartic are hoisted out
// of the Durand-Kerner sweeps; shadow rays whose result is provably unused are not cast;
// conservative bounding culls sit in front of the expensive solvers.
//
// Numerics contract (DESIGN.md "Numerics"): every value that feeds a hit/miss decision is
// computed with exactly the float operations, in exactly the order, that rt.frag spells out
// (IEEE binary32, no FMA contraction: the file must be compiled with -ffp-contract=off; IEEE
// division and sqrt). Hoisting and culling never change an operand. Citations "rt.frag:N" give
// the shader lines whose semantics a function carries.
//
// The header is also compilable by a plain host C++ compiler (tests/host_harness) so that the
// device logic can be checked against the oracle without a GPU; wave intrinsics collapse to
// their one-lane meaning there.
#pragma once

#include <math.h>
#include <stdint.h>

#include "rt_scene_dev.h"

#if defined(__HIPCC__)
#define RT_HD __host__ __device__ __forceinline__
#define RT_HDM __host__ __device__ __forceinline__
// Cold, register-hungry leaves (Durand-Kerner, trilinear taps, float64 atan/asin) can be compiled out
// of line as a register-allocation boundary.
// (Measured: real calls make the whole kernel take the 256-VGPR call ABI budget and add scratch
// frames -- 912 B/lane -- so the leaves stay inlined unless RT_OUTLINE_COLD is defined.)
#if defined(RT_OUTLINE_COLD)
#define RT_COLD __host__ __device__ __attribute__((noinline))
#else
#define RT_COLD __host__ __device__ __forceinline__
#endif
#else
#define RT_HD static inline
#define RT_HDM inline
#define RT_COLD static inline
#endif

#if defined(__HIP_DEVICE_COMPILE__)
#define RT_ANY(x) (__any((x)) != 0)
#else
#define RT_ANY(x) (x)
#endif

namespace rtdev {

// two-wide float vector: element-wise IEEE operations (v_pk_*_f32 on the device, SSE on the host)
#if defined(__clang__)
typedef float v2f __attribute__((ext_vector_type(2)));
#else
typedef float v2f __attribute__((vector_size(8)));
#endif

// ------------------------------------------------------------------------------------------
// small vector algebra (component order and association exactly as GLSL evaluates them)
// ------------------------------------------------------------------------------------------
RT_HD f3 mk3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
RT_HD f4 mk4(float x, float y, float z, float w) { f4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }
RT_HD f3 xyz(const f4& v) { return mk3(v.x, v.y, v.z); }
RT_HD f3 operator+(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD f3 operator-(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD f3 operator*(f3 a, f3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD f3 operator*(f3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
RT_HD f3 operator/(f3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
RT_HD f3 operator-(f3 a) { return mk3(-a.x, -a.y, -a.z); }
RT_HD float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
RT_HD float dot2(float ax, float ay, float bx, float by) { return ax * bx + ay * by; }
RT_HD float length3(f3 a) { return sqrtf(dot3(a, a)); }
RT_HD f3 normalize3(f3 a) { return a / length3(a); }
RT_HD float gl_min(float a, float b) { return b < a ? b : a; }   // GLSL min: NaN-asymmetric on purpose
RT_HD float gl_max(float a, float b) { return a < b ? b : a; }
RT_HD float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }
RT_HD float gl_step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
RT_HD float gl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
RT_HD f3 gl_reflect(f3 I, f3 N) { return I - N * (2.0f * dot3(N, I)); }
RT_HD f3 gl_refract(f3 I, f3 N, float eta)
{
    const float d = dot3(N, I);
    const float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return mk3(0.0f, 0.0f, 0.0f);
    return I * eta - N * (eta * d + sqrtf(k));
}

// rt.frag:285-311. quat_rotate keeps every product of quat_mult, including the ones with the
// zero w of the embedded vector: dropping them would change signed zeros / NaN propagation.
RT_HD f4 quat_conj(f4 q) { return mk4(-q.x, -q.y, -q.z, q.w); }
RT_HD f4 quat_inv(f4 q)
{
    const f4 c = quat_conj(q);
    const float s = 1.0f / (q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    return mk4(c.x * s, c.y * s, c.z * s, c.w * s);
}
RT_HD f3 quat_rotate(f4 q, f3 v)
{
    const float pw = 0.0f;
    // q_tmp = quat_mult(q, (v,0))
    const float tx = (q.w * v.x) + (q.x * pw) + (q.y * v.z) - (q.z * v.y);
    const float ty = (q.w * v.y) - (q.x * v.z) + (q.y * pw) + (q.z * v.x);
    const float tz = (q.w * v.z) + (q.x * v.y) - (q.y * v.x) + (q.z * pw);
    const float tw = (q.w * pw) - (q.x * v.x) - (q.y * v.y) - (q.z * v.z);
    // quat_mult(q_tmp, conj(q)).xyz
    const float cx = -q.x, cy = -q.y, cz = -q.z, cw = q.w;
    f3 r;
    r.x = (tw * cx) + (tx * cw) + (ty * cz) - (tz * cy);
    r.y = (tw * cy) - (tx * cz) + (ty * cw) + (tz * cx);
    r.z = (tw * cz) + (tx * cy) - (ty * cx) + (tz * cw);
    return r;
}

// Exact shortcut for rotations by the identity quaternion (floor slabs, un-rotated crates, ...):
// with q = (+-0,+-0,+-0,1) every product in quat_rotate other than 1*v_i is a signed zero and the
// sums of those zeros come out as +0 whatever their signs, so for FINITE v the result is v with
// each -0 component turned into +0, i.e. v + 0.0f component-wise, bit for bit (an infinite component
// would turn 0*inf into NaN: then the full formula runs). Checked for all 8 zero-sign patterns of q
// against quat_rotate on signed zeros, denormals, huge and random finite vectors (tests/test_culls.py).
// `plain3` (finite and no zero component) is the stronger proviso under which the result is v itself.
RT_HD bool quat_is_identity(f4 q) { return q.x == 0.0f && q.y == 0.0f && q.z == 0.0f && q.w == 1.0f; }
// The packer evaluates quat_is_identity once per primitive and stores the answer as an integer in a spare
// record field: on the device a wave-uniform FLOAT comparison still costs VALU instructions (four v_cmp per
// test, no scalar float compare on gfx950), an integer one is a scalar s_cmp.
RT_HD bool ident_flag(float w) { return __builtin_bit_cast(int, w) != 0; }
RT_HD bool finite3(f3 v)
{
    const float z = v.x * 0.0f + v.y * 0.0f + v.z * 0.0f;  // NaN iff some component is inf/NaN
    return z == 0.0f;
}
RT_HD bool plain3(f3 v) { return finite3(v) && v.x != 0.0f && v.y != 0.0f && v.z != 0.0f; }
RT_HD f3 quat_rotate_id(f4 q, bool identity, f3 v)
{
    f3 r = mk3(v.x + 0.0f, v.y + 0.0f, v.z + 0.0f);
    if (!(identity && finite3(v))) r = quat_rotate(q, v);
    return r;
}
// pow() of the shading terms (rt.frag:689 specular, :716 Fresnel): x in [0,1], y >= 1. GLSL defines
// pow(x,y) as exp2(y*log2(x)) at the implementation's precision; on the device it is exactly that on the
// hardware log/exp units (4 instructions instead of ~160 for the correctly rounded libm routine).
// Measured max abs error against double pow over [0,1] for y = 1..5000: 8.0e-8 (ocml powf: 6.2e-8),
// tools/micro/pow_err.hip. Colour-only, covered by the 1e-4 tolerance (DESIGN.md "Numerics").
RT_HD float rt_pow(float x, float y)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x));
#else
    return powf(x, y);
#endif
}

// atan(y,x) / asin(x) of the equirect mapping (rt.frag:323-324): fixed float64 series, only
// + - * / sqrt, one final rounding to float -> bit-reproducible on host and device
// (DESIGN.md "Numerics"; the oracle states the same algorithm independently).
RT_HD double atan_unit(double a)  // 0 <= a <= 1
{
    double off = 0.0;
    if (a > 0.4142135623730950488) { a = (a - 1.0) / (a + 1.0); off = 0.78539816339744830962; }
    const double s = a * a;
    double p = 1.0 / 25.0;
    p = 1.0 / 23.0 - s * p;
    p = 1.0 / 21.0 - s * p;
    p = 1.0 / 19.0 - s * p;
    p = 1.0 / 17.0 - s * p;
    p = 1.0 / 15.0 - s * p;
    p = 1.0 / 13.0 - s * p;
    p = 1.0 / 11.0 - s * p;
    p = 1.0 / 9.0 - s * p;
    p = 1.0 / 7.0 - s * p;
    p = 1.0 / 5.0 - s * p;
    p = 1.0 / 3.0 - s * p;
    p = 1.0 - s * p;
    return off + a * p;
}
RT_COLD float rt_atan2(float yf, float xf)
{
    const double y = (double)yf, x = (double)xf;
    const double ay = y < 0.0 ? -y : y, ax = x < 0.0 ? -x : x;
    const double mx = ax < ay ? ay : ax, mn = ax < ay ? ax : ay;
    double r;
    if (!(mx > 0.0)) r = 0.0;
    else {
        r = atan_unit(mn / mx);
        if (ax < ay) r = 1.57079632679489661923 - r;
        if (x < 0.0) r = 3.14159265358979323846 - r;
        if (y < 0.0) r = -r;
    }
    return (float)r;
}
RT_COLD float rt_asin(float xf)
{
    const double x = (double)xf;
    const double c = sqrt(1.0 - x * x);
    if (!(c == c)) return NAN;
    const double ax = x < 0.0 ? -x : x;
    const double mx = ax < c ? c : ax, mn = ax < c ? ax : c;
    double r;
    if (!(mx > 0.0)) r = 0.0;
    else {
        r = atan_unit(mn / mx);
        if (c < ax) r = 1.57079632679489661923 - r;
        if (x < 0.0) r = -r;
    }
    return (float)r;
}

// log2 for the texture LOD (rt.frag:331-337 and the implicit-LOD rule). The LOD decides the blend
// weight between two mip levels, and a trilinear alpha of (1-f)*1 + f*1 may or may not round to
// exactly 1.0 depending on the last bit of f -- which flips the `alpha < 1` pass-through test
// (rt.frag:884). So log2, too, is a fixed float64 sequence (exponent split + atanh series), bit-
// reproducible on host and device; the oracle states the same algorithm independently.
RT_HD float rt_log2(float xf)
{
    if (!(xf > 0.0f)) return xf == 0.0f ? -INFINITY : NAN;
    if (xf > 3.0e38f) return INFINITY;
    const double x = (double)xf;
    const uint64_t bits = __builtin_bit_cast(uint64_t, x);
    int e = (int)((bits >> 52) & 0x7ffu) - 1023;
    double m = __builtin_bit_cast(double, (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);  // [1,2)
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double z = (m - 1.0) / (m + 1.0);
    const double z2 = z * z;
    double p = 1.0 / 19.0;
    p = 1.0 / 17.0 + z2 * p;
    p = 1.0 / 15.0 + z2 * p;
    p = 1.0 / 13.0 + z2 * p;
    p = 1.0 / 11.0 + z2 * p;
    p = 1.0 / 9.0 + z2 * p;
    p = 1.0 / 7.0 + z2 * p;
    p = 1.0 / 5.0 + z2 * p;
    p = 1.0 / 3.0 + z2 * p;
    p = 1.0 + z2 * p;
    const double ln_m = 2.0 * z * p;
    return (float)((double)e + ln_m * 1.4426950408889634074);
}

// ------------------------------------------------------------------------------------------
// scene view + constants
// ------------------------------------------------------------------------------------------
#define RT_MAXDIST 1000000.0f         /* rt.frag:145 */
#define RT_PI_F 3.14159265358979f     /* rt.frag:5 */
#define RT_FLT_MAX 3.402823466e+38f   /* rt.frag:4 */
#define RT_SEGMENT_CAP 256            /* bound on main-loop trips (refraction does i--, trap T2) */

// Typed access to the DevScene blob. Array addresses are re-derived from the header offsets at
// each use (two scalar ops) instead of being kept in 30 SGPRs for the whole kernel.
struct SceneView {
    const DevSceneHeader* h;   // counts, camera, offsets
    const char* blob;
    template <class T> RT_HDM const T* at(uint32_t off) const { return reinterpret_cast<const T*>(blob + off); }
    RT_HDM const DevSphere* spheres() const { return at<DevSphere>(h->off_sphere); }
    RT_HDM const DevPlane* planes() const { return at<DevPlane>(h->off_plane); }
    RT_HDM const DevSurface* surfaces() const { return at<DevSurface>(h->off_surface); }
    RT_HDM const DevBox* boxes() const { return at<DevBox>(h->off_box); }
    RT_HDM const DevTorus* tori() const { return at<DevTorus>(h->off_torus); }
    RT_HDM const DevRing* rings() const { return at<DevRing>(h->off_ring); }
    RT_HDM const DevLightPoint* lights_point() const { return at<DevLightPoint>(h->off_light_point); }
    RT_HDM const DevLightDirect* lights_direct() const { return at<DevLightDirect>(h->off_light_direct); }
    RT_HDM const DevMaterial* mats(int type) const { return at<DevMaterial>(h->off_mat[type]); }
    RT_HDM const f4* sph_geom() const { return at<f4>(h->off_sph_geom); }
    RT_HDM const uint32_t* sph_hollow() const { return at<uint32_t>(h->off_sph_hollow); }
    RT_HDM const DevSurfaceCull* surf_cull() const { return at<DevSurfaceCull>(h->off_surf_cull); }
    RT_HDM const f4* torus_bound() const { return at<f4>(h->off_torus_bound); }
    RT_HDM const f4* ring_bound() const { return at<f4>(h->off_ring_bound); }
    RT_HDM const f4* surf_group() const { return at<f4>(h->off_surf_group); }
    RT_HDM const f4* torus_group() const { return at<f4>(h->off_torus_group); }
    RT_HDM const DevPencil* pencils() const { return at<DevPencil>(h->off_pencil); }
    RT_HDM const DevSlabs* slabs() const { return at<DevSlabs>(h->off_slabs); }
    const uint32_t* pen;       // pencil masks (a buffer of their own, built on the device), nullptr = none
};
// `hdr` normally is the blob's own first record; with the tables staged in LDS it stays in global memory.
RT_HD SceneView make_view(const char* blob, const DevSceneHeader* hdr, const uint32_t* pencil_masks = nullptr)
{
    SceneView S;
    S.h = hdr;
    S.blob = blob;
    S.pen = pencil_masks;
    return S;
}
RT_HD SceneView make_view(const char* blob) { return make_view(blob, reinterpret_cast<const DevSceneHeader*>(blob)); }

struct TexTable {
    DevTexture tex[TEX_SLOTS];
    DevCubemap sky;
    int32_t lod;  // 0: level-0 bilinear everywhere; 1: mip chain + quad-derivative LOD (DESIGN.md "Texture rule")
};

// Optional per-wave phase timers (profiling builds only, -DRT_PHASE_TIMERS): wave-cycles spent in
// each phase of the segment loop, read with s_memtime. Nested: DK time is also inside SCAN/SHADE.
enum { PH_SETUP = 0, PH_SCAN, PH_HITINFO, PH_CLASSIFY, PH_SKY, PH_SHADE, PH_DK, PH_APPLY, PH_SHADOW, PH_TRIPS, PH_C_SPH, PH_C_SURF, PH_C_BOX, PH_C_TORUS, PH_C_RING, PH_C_LIGHT, PH_COUNT = 18 };
#if defined(RT_PHASE_TIMERS)
struct PhaseClock { unsigned long long acc[PH_COUNT]; };
#endif
#if defined(RT_PHASE_TIMERS) && defined(__HIP_DEVICE_COMPILE__)
// s_memtime has no data dependence on the VALU work around it, so it is fenced with scheduling
// barriers (nothing may be moved across) and an explicit wait for its own result.
__device__ __forceinline__ unsigned long long rt_ph_now()
{
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long t = __builtin_amdgcn_s_memtime();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define RT_PH_DECL unsigned long long _ph_t0 = rt_ph_now()
#define RT_PH_LAP(cnt, id) { const unsigned long long _now = rt_ph_now(); (cnt).pc.acc[id] += _now - _ph_t0; _ph_t0 = _now; }
#define RT_PH_BEGIN(name) const unsigned long long name = rt_ph_now()
#define RT_PH_END(cnt, id, name) (cnt).pc.acc[id] += rt_ph_now() - (name)
#define RT_PH_ADD(cnt, id, v) (cnt).pc.acc[id] += (v)
#else
#if !defined(RT_PHASE_TIMERS)
struct PhaseClock {};
#endif
#define RT_PH_DECL
#define RT_PH_LAP(cnt, id)
#define RT_PH_BEGIN(name)
#define RT_PH_END(cnt, id, name)
#define RT_PH_ADD(cnt, id, v)
#endif

struct LaneCounters {
    uint32_t closest;      // calcInter invocations (reference-defined closest-hit rays)
    uint32_t shadow_ref;   // inShadow invocations the reference would make
    uint32_t shadow_cast;  // shadow scans actually executed
    uint32_t torus_solves; // Durand-Kerner solves actually executed
    PhaseClock pc;
};

// ------------------------------------------------------------------------------------------
// texture sampling (DESIGN.md "Texture rule"): exact float weights, RGBA8 taps
// ------------------------------------------------------------------------------------------
// byte / 255.0f, correctly rounded, without the IEEE divide: one multiply by RN(1/255) and one
// fma-based correction step. Equality with the division for all 256 inputs is asserted on the
// host at library load and on the device by tests (rtx_selftest).
RT_HD float unorm8(uint32_t b)
{
    const float x = (float)b;
    const float rcp = 0.0039215688593685626983642578125f;  // RN(1/255)
    float q = x * rcp;
    const float r = __builtin_fmaf(-q, 255.0f, x);
    q = __builtin_fmaf(r, rcp, q);
    return q;
}
RT_HD f4 unpack_rgba8(uint32_t p) { return mk4(unorm8(p & 255u), unorm8((p >> 8) & 255u), unorm8((p >> 16) & 255u), unorm8(p >> 24)); }

RT_HD void axis_taps(float u, int n, float fn, int wrap, int& i0, int& i1, float& a)
{
    if (wrap == 0) u = u - floorf(u);
    else u = gl_min(gl_max(u, -1.0f), 2.0f);
    if (!(u == u)) u = 0.0f;
    const float x = u * fn - 0.5f;  // fn == (float)n
    const float fl = floorf(x);
    a = x - fl;
    int i = (int)fl;
    int k = i + 1;
    if (wrap == 0) {
        if (i < 0) i += n;
        if (i >= n) i -= n;
        if (k >= n) k -= n;
    } else {
        i = i < 0 ? 0 : (i > n - 1 ? n - 1 : i);
        k = k < 0 ? 0 : (k > n - 1 ? n - 1 : k);
    }
    i0 = i;
    i1 = k;
}
// `first` = dword index of the image's first texel inside `base` (a mip level / cube face): all
// addressing is base (wave-uniform pointer) + 32-bit per-lane index.
RT_HD f4 bilinear_taps(const uint32_t* base, uint32_t first, int w, int i0, int i1, int j0, int j1, float a, float b)
{
    const uint32_t r0 = first + (uint32_t)j0 * (uint32_t)w, r1 = first + (uint32_t)j1 * (uint32_t)w;
    const uint32_t p00 = base[r0 + (uint32_t)i0];
    const uint32_t p10 = base[r0 + (uint32_t)i1];
    const uint32_t p01 = base[r1 + (uint32_t)i0];
    const uint32_t p11 = base[r1 + (uint32_t)i1];
    const f4 t00 = unpack_rgba8(p00), t10 = unpack_rgba8(p10), t01 = unpack_rgba8(p01), t11 = unpack_rgba8(p11);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    f4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}
RT_HD f4 sample2d_level0(const DevTexture& t, float u, float v)
{
    if (t.texels == nullptr) return mk4(0.0f, 0.0f, 0.0f, 1.0f);
    int i0, i1, j0, j1;
    float a, b;
    axis_taps(u, t.width, t.fwidth, t.wrap, i0, i1, a);
    axis_taps(v, t.height, t.fheight, t.wrap, j0, j1, b);
    return bilinear_taps(t.texels, 0u, t.width, i0, i1, j0, j1, a, b);
}
// One mip level: dimensions max(1, w>>l) x max(1, h>>l), texels at level_off[l].
RT_HD f4 sample2d_level(const DevTexture& t, int level, float u, float v)
{
    int w = t.width >> level, h = t.height >> level;
    w = w < 1 ? 1 : w;
    h = h < 1 ? 1 : h;
    int i0, i1, j0, j1;
    float a, b;
    axis_taps(u, w, (float)w, t.wrap, i0, i1, a);
    axis_taps(v, h, (float)h, t.wrap, j0, j1, b);
    return bilinear_taps(t.texels, t.level_off[level], w, i0, i1, j0, j1, a, b);
}
// lambda <= 0 (also -inf, NaN): level-0 bilinear; otherwise floor(lambda) and the next level
// (clamped to the last), blended (1-f)*c0 + f*c1.
RT_COLD f4 sample2d_lod(const DevTexture& t, float u, float v, float lambda)
{
    if (t.texels == nullptr) return mk4(0.0f, 0.0f, 0.0f, 1.0f);
    int l0 = 0;
    float f = 0.0f;
    bool two = false;
    if (lambda > 0.0f) {
        const float top = (float)(t.levels - 1);
        if (lambda > top) lambda = top;
        const float fl = floorf(lambda);
        l0 = (int)fl;
        f = lambda - fl;
        two = l0 + 1 <= t.levels - 1;
    }
    // the two levels are sampled one after the other by the SAME code (not unrolled): four taps in
    // flight at a time keeps the sampler's register footprint at the bilinear one
    f4 c0 = mk4(0.0f, 0.0f, 0.0f, 0.0f), c1 = c0;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && !RT_ANY(two)) break;
        const f4 c = sample2d_level(t, l0 + pass * (two ? 1 : 0), u, v);
        if (pass == 0) c0 = c; else c1 = c;
    }
    if (!two) return c0;
    return mk4((1.0f - f) * c0.x + f * c1.x, (1.0f - f) * c0.y + f * c1.y, (1.0f - f) * c0.z + f * c1.z, (1.0f - f) * c0.w + f * c1.w);
}
// texture(skybox, d): GL face table, per-face bilinear, CLAMP_TO_EDGE, not seamless (rt.frag:893)
RT_HD f4 sample_cube(const DevCubemap& c, f3 d)
{
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    int face;
    float sc, tc, ma;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0.0f) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az)        { ma = ay; if (d.y >= 0.0f) { face = 2; sc = d.x; tc = d.z; }  else { face = 3; sc = d.x; tc = -d.z; } }
    else                      { ma = az; if (d.z >= 0.0f) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    if (c.texels == nullptr || !((c.face_mask >> face) & 1)) return mk4(0.0f, 0.0f, 0.0f, 1.0f);
    const float s = 0.5f * (sc / ma + 1.0f);
    const float t = 0.5f * (tc / ma + 1.0f);
    int i0, i1, j0, j1;
    float a, b;
    axis_taps(s, c.size, c.fsize, 1, i0, i1, a);
    axis_taps(t, c.size, c.fsize, 1, j0, j1, b);
    return bilinear_taps(c.texels, (uint32_t)face * (uint32_t)c.size * (uint32_t)c.size, c.size, i0, i1, j0, j1, a, b);
}

// ---- mip-mapped sky box: GLWrapper::load_cubemap(faces, genMipmap = true), GLWrapper.cpp:307-310 -> texture(skybox, rd) (rt.frag:893)
// is min-filtered GL_LINEAR_MIPMAP_LINEAR. Rule (DESIGN.md "Texture rule", cube part; oracle: cube_lambda / sample_cube_lod): the level of
// detail comes from the derivatives of the face coordinates s = (sc/ma + 1)/2, t = (tc/ma + 1)/2 on the pixel's OWN face, obtained from
// the quad differences of the DIRECTION by the quotient rule d(sc/ma) = (dsc * ma - sc * dma) / ma^2.
RT_HD int cube_face(f3 d)
{
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (ax >= ay && ax >= az) return d.x >= 0.0f ? 0 : 1;
    if (ay >= az) return d.y >= 0.0f ? 2 : 3;
    return d.z >= 0.0f ? 4 : 5;
}
// (sc, tc, ma) of the GL face table for a FIXED face: a linear map, applied to the direction and to its derivatives alike
RT_HD f3 cube_project(int face, f3 v)
{
    const int axis = face >> 1;
    const bool neg = (face & 1) != 0;
    const float m = axis == 0 ? v.x : (axis == 1 ? v.y : v.z);
    const float sc = axis == 0 ? (neg ? v.z : -v.z) : (axis == 1 ? v.x : (neg ? -v.x : v.x));
    const float tc = axis == 1 ? (neg ? -v.z : v.z) : -v.y;
    return mk3(sc, tc, neg ? -m : m);
}
RT_HD float cube_lambda(const DevCubemap& c, f3 d, f3 ddx, f3 ddy)
{
    const int face = cube_face(d);
    const f3 p = cube_project(face, d), px = cube_project(face, ddx), py = cube_project(face, ddy);
    const float ma = p.z, ma2 = ma * ma;
    const float dsdx = 0.5f * ((px.x * ma - p.x * px.z) / ma2), dtdx = 0.5f * ((px.y * ma - p.y * px.z) / ma2);
    const float dsdy = 0.5f * ((py.x * ma - p.x * py.z) / ma2), dtdy = 0.5f * ((py.y * ma - p.y * py.z) / ma2);
    const float n = c.fsize;
    const float rx = sqrtf((dsdx * n) * (dsdx * n) + (dtdx * n) * (dtdx * n));
    const float ry = sqrtf((dsdy * n) * (dsdy * n) + (dtdy * n) * (dtdy * n));
    return rt_log2(gl_max(rx, ry));
}
// lambda as in sample2d_lod; level L of the cube: 6 faces of max(1, size>>L)^2 texels behind the levels before it
RT_COLD f4 sample_cube_lod(const DevCubemap& c, f3 d, float lambda)
{
    const int face = cube_face(d);
    const f3 p = cube_project(face, d);
    if (c.texels == nullptr || !((c.face_mask >> face) & 1)) return mk4(0.0f, 0.0f, 0.0f, 1.0f);
    const float s = 0.5f * (p.x / p.z + 1.0f);
    const float t = 0.5f * (p.y / p.z + 1.0f);
    int l0 = 0;
    float f = 0.0f;
    bool two = false;
    if (lambda > 0.0f) {
        const float top = (float)(c.levels - 1);
        if (lambda > top) lambda = top;
        const float fl = floorf(lambda);
        l0 = (int)fl;
        f = lambda - fl;
        two = l0 + 1 <= c.levels - 1;
    }
    int w = c.size >> l0;            // level l0: faces of max(1, size >> l0) texels, at c.level_off[l0] (the host's table, as for the 2-D textures)
    w = w > 1 ? w : 1;
    uint32_t first = c.level_off[l0];
    f4 c0 = mk4(0.0f, 0.0f, 0.0f, 0.0f), c1 = c0;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1 && !RT_ANY(two)) break;
        if (pass == 1 && two) {
            first = c.level_off[l0 + 1];
            w = w > 1 ? w >> 1 : 1;
        }
        int i0, i1, j0, j1;
        float a, b;
        axis_taps(s, w, (float)w, 1, i0, i1, a);
        axis_taps(t, w, (float)w, 1, j0, j1, b);
        const f4 v = bilinear_taps(c.texels, first + (uint32_t)face * (uint32_t)w * (uint32_t)w, w, i0, i1, j0, j1, a, b);
        if (pass == 0) c0 = v; else c1 = v;
    }
    if (!two) return c0;
    return mk4((1.0f - f) * c0.x + f * c1.x, (1.0f - f) * c0.y + f * c1.y, (1.0f - f) * c0.z + f * c1.z, (1.0f - f) * c0.w + f * c1.w);
}

// The single 2-D texture fetch site. Each lane may request a fetch from a different sampler slot;
// the wave serves one slot per pass (wave-uniform slot -> the sampler state is read with scalar
// loads at a computed address, nothing is hoisted into long-lived registers), lanes of other slots
// wait their turn. One inlined copy of the sampler instead of one per call site.
RT_HD int rt_first_slot(bool pending, int slot)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long m = __ballot(pending);
    const int src = __ffsll((long long)m) - 1;
    return __builtin_amdgcn_readlane(slot, src);
#else
    (void)pending;
    return slot;
#endif
}
// Values of the horizontal / vertical neighbour inside the lane's 2x2 pixel quad. The lane mapping
// of the kernel puts a quad in 4 consecutive lanes (bit0 = x&1, bit1 = y&1), so this is a DPP
// quad_perm move, no LDS. Must be called with all four lanes of the quad active.
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD int quad_other_x(int v) { return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true); }  // quad_perm [1,0,3,2]
RT_HD int quad_other_y(int v) { return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true); }  // quad_perm [2,3,0,1]
RT_HD int rt_lane_id() { return (int)(threadIdx.x & 63u); }
#else
RT_HD int quad_other_x(int v) { return v; }  // the host harness has no quads: texture_lod must be 0 there
RT_HD int quad_other_y(int v) { return v; }
RT_HD int rt_lane_id() { return 0; }
#endif
RT_HD float quad_other_x(float v) { return __builtin_bit_cast(float, quad_other_x(__builtin_bit_cast(int, v))); }
RT_HD float quad_other_y(float v) { return __builtin_bit_cast(float, quad_other_y(__builtin_bit_cast(int, v))); }

// `prim` identifies the primitive being textured (type and index): a quad neighbour contributes to
// a derivative only if it executes this very fetch for the same sampler and the same primitive.
RT_HD f4 fetch2d(const TexTable& T, bool want, int slot, int prim, float u, float v)
{
    float dudx = 0.0f, dvdx = 0.0f, dudy = 0.0f, dvdy = 0.0f;
    if (T.lod) {
        const int key = want ? (slot | (prim << 3)) : -1;
        const int lane = rt_lane_id();
        const int kx = quad_other_x(key), ky = quad_other_y(key);
        const float ux = quad_other_x(u), vx = quad_other_x(v), uy = quad_other_y(u), vy = quad_other_y(v);
        if (want && kx == key) {  // dFdx = right - left
            const bool right = (lane & 1) != 0;
            dudx = right ? u - ux : ux - u;
            dvdx = right ? v - vx : vx - v;
        }
        if (want && ky == key) {  // dFdy = top - bottom (y grows upwards, gl_FragCoord)
            const bool top = (lane & 2) != 0;
            dudy = top ? u - uy : uy - u;
            dvdy = top ? v - vy : vy - v;
        }
    }
    f4 out = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    bool pending = want;
    while (RT_ANY(pending)) {
        const int s = rt_first_slot(pending, slot);
        const bool mine = pending && slot == s;
        if (mine) {
            const DevTexture& t = T.tex[s];
            if (T.lod) {
                float lambda;
                if (s <= TEX_SPHERE_4) {  // getSphereTexture: textureLod(.., log2(max(df.x,df.y)*1024)), rt.frag:326-338
                    float dfx = fabsf(dudx) + fabsf(dudy);
                    const float dfy = fabsf(dvdx) + fabsf(dvdy);
                    if (dfx > 0.5f) dfx = 0.0f;
                    lambda = rt_log2(gl_max(dfx, dfy) * 1024.0f);
                } else {                  // texture(): implicit LOD from the texel-space footprint
                    const float rx = sqrtf((dudx * t.fwidth) * (dudx * t.fwidth) + (dvdx * t.fheight) * (dvdx * t.fheight));
                    const float ry = sqrtf((dudy * t.fwidth) * (dudy * t.fwidth) + (dvdy * t.fheight) * (dvdy * t.fheight));
                    lambda = rt_log2(gl_max(rx, ry));
                }
                out = sample2d_lod(t, u, v, lambda);
            } else {
                out = sample2d_level0(t, u, v);
            }
        }
        pending = pending && !mine;
    }
    return out;
}

// ------------------------------------------------------------------------------------------
// intersectors
// ------------------------------------------------------------------------------------------
// rt.frag:342-354; geom.w is the hoisted r*r
RT_HD bool intersect_sphere(f3 ro, f3 rd, f4 geom, bool hollow, float tmin, float& t)
{
    const f3 oc = ro - xyz(geom);
    const float b = dot3(oc, rd);
    const float c = dot3(oc, oc) - geom.w;
    const float h = b * b - c;
    if (h < 0.0f) return false;
    const float hs = sqrtf(h);
    t = -b - hs;
    if (hollow && t < 0.0f) t = -b + hs;
    return t > 0.0f && t < tmin;
}

// rt.frag:356-370 (PLANE_ONESIDE defined -> one-sided, trap T1)
RT_HD bool intersect_plane(f3 ro, f3 rd, f3 n, f3 p, float tmin, float& t)
{
    const float denom = gl_clamp(dot3(n, rd), -1.0f, 1.0f);
    if (denom < -1e-6f) {
        t = dot3(p - ro, n) / denom;
        return t > 0.0f && t < tmin;
    }
    return false;
}

// rt.frag:372-390. uv is written only on a hit (the shader's global opt_uv)
RT_HD bool intersect_ring(const DevRing& R, f3 ro, f3 rd, float tmin, float& t, f2& uv)
{
    const bool ident = ident_flag(R.normal.w);
    const f3 d = quat_rotate_id(R.quat, ident, rd);
    const f3 o = quat_rotate_id(R.quat, ident, ro - xyz(R.pos_tex));
    t = -o.z / d.z;
    const float x = o.x + d.x * t;
    const float y = o.y + d.y * t;
    const float p = x * x + y * y;
    if (t > 0.0f && t < tmin && p < R.radii.y && p > R.radii.x) {
        const float len = sqrtf(x * x + y * y);
        const float nx = x / len, ny = y / len;
        uv = mk2((p - R.radii.x) / R.radii.z, nx * 1.0f + ny * 0.0f);
        return true;
    }
    return false;
}

// rt.frag:399-427. nor (box-space normal) is written only on a hit; NaN falls through the
// early-outs exactly like the shader (trap T5); no t>0 test (trap T21).
// Per-ray values shared by all identity-rotated boxes: for those rdd == rd bit for bit when rd is
// "plain" (quat_rotate_id), so m = 1/rdd is the same three IEEE divisions for every such box and
// is computed once per ray, lazily (wave-uniform flag).
struct RayBoxCtx {
    bool have = false;   // wave-uniform
    bool rd_plain = false;
    f3 inv_rd;
};
RT_HD bool intersect_box(const DevBox& B, f3 ro, f3 rd, float tmin, float& t, f3& nor, RayBoxCtx& ctx)
{
    const bool ident = ident_flag(B.pos.w);
    if (ident && !ctx.have) {
        ctx.have = true;
        ctx.rd_plain = plain3(rd);
        ctx.inv_rd = mk3(1.0f / rd.x, 1.0f / rd.y, 1.0f / rd.z);
    }
    f3 rdd = rd;
    f3 m = ctx.inv_rd;
    if (!(ident && ctx.rd_plain)) {
        rdd = quat_rotate(B.quat, rd);
        m = mk3(1.0f / rdd.x, 1.0f / rdd.y, 1.0f / rdd.z);
    }
    const f3 roo = quat_rotate_id(B.quat, ident, ro - xyz(B.pos));
    const f3 n = m * roo;
    const f3 k = mk3(fabsf(m.x), fabsf(m.y), fabsf(m.z)) * xyz(B.form_tex);
    const f3 t1 = -n - k;
    const f3 t2 = -n + k;
    const float tN = gl_max(gl_max(t1.x, t1.y), t1.z);
    const float tF = gl_min(gl_min(t2.x, t2.y), t2.z);
    if (tN > tF || tF < 0.0f) return false;
    if (tN >= tmin) return false;
    nor.x = -gl_sign(rdd.x) * gl_step(t1.y, t1.x) * gl_step(t1.z, t1.x);
    nor.y = -gl_sign(rdd.y) * gl_step(t1.z, t1.y) * gl_step(t1.x, t1.y);
    nor.z = -gl_sign(rdd.z) * gl_step(t1.x, t1.z) * gl_step(t1.y, t1.z);
    t = tN;
    return true;
}

// ---- torus: Durand-Kerner on the ray/torus quartic (rt.frag:438-487) ----
struct TorusRay {  // ray-invariant terms of cTorus (rt.frag:445-455), hoisted out of the sweeps
    float a, b, c;        // dot(rd,rd), dot(ro,rd), dot(ro,ro)+R2-r2
    float axy, bxy, cxy;  // the same three over .xy
    float k;              // 4*R2
};
// Complex numbers live in two-wide vectors (re, im), so the complex products and the polynomial's
// real/imaginary halves evaluate as packed FP32 instructions: per half exactly the operations, operands
// and association of the shader's cMul / cTorus / the Durand-Kerner update (rt.frag:438-476).
RT_HD v2f c_xx(v2f v) { v2f r; r[0] = v[0]; r[1] = v[0]; return r; }
RT_HD v2f c_yy(v2f v) { v2f r; r[0] = v[1]; r[1] = v[1]; return r; }
RT_HD v2f c_yx(v2f v) { v2f r; r[0] = v[1]; r[1] = v[0]; return r; }
RT_HD v2f mk2v(float x, float y) { v2f r; r[0] = x; r[1] = y; return r; }
// (p.x*q.x - p.y*q.y, p.x*q.y + p.y*q.x); a - b is a + (-b) in IEEE arithmetic
RT_HD v2f cmul(v2f p, v2f q)
{
    const v2f a = c_xx(p) * q;
    const v2f b = c_yy(p) * c_yx(q);
    // a.x - b.x is a.x + (-b.x) in IEEE arithmetic. Spelled as two scalar operations: as "negate b.x, packed add" it compiled to a packed
    // negation of both halves, a move to restore the upper one and the packed add (round 4, ISA of the sweep: 140 packed + 21 moves per
    // sweep before, 100 packed + 53 scalar after; with -disable-vector-combine, which otherwise re-packs the pair into TWO packed
    // operations and a move).
    return mk2v(a[0] - b[0], a[1] + b[1]);
}
RT_HD v2f torus_poly(v2f t, const TorusRay& w)
{
    const v2f sq = t * t;                                        // (x*x, y*y)
    const v2f two_t = t * 2.0f;                                  // (2*x, 2*y)
    const v2f t2 = mk2v(sq[0] - sq[1], two_t[0] * t[1]);          // t*t as a complex number
    v2f res = t2 * w.a + two_t * w.b + mk2v(w.c, 0.0f);
    res = cmul(res, res);
    const v2f res2 = (t2 * w.axy + two_t * w.bxy + mk2v(w.cxy, 0.0f)) * w.k;
    return res - res2;
}
RT_HD float dk_step(v2f& c0, v2f c1, v2f c2, v2f c3, const TorusRay& w)
{
    v2f fc = torus_poly(c0, w);
    const v2f den = cmul(c0 - c1, cmul(c0 - c2, c0 - c3));
    const v2f d2 = den * den;
    const float dd = d2[0] + d2[1];
    fc = cmul(fc, mk2v(den[0] / dd, -den[1] / dd));
    c0 = c0 - fc;
    return gl_max(fabsf(fc[0]), fabsf(fc[1]));
}
#if defined(RT_DK_STATS) && defined(__HIPCC__)
__device__ unsigned long long g_dk[16];  // diagnostic build only: wave execs, lane solves, wave sweeps, lane sweeps, wave cycles, ...;
                                         // [8] scans with a candidate, [9] sum over scans of the busiest lane's candidates (= passes),
                                         // [10] candidates of all lanes, [11] sum of ceil(candidates / 64), [12] lanes with a candidate
#endif
RT_HD void dk_stats_scan(unsigned long long cand)
{
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
    const int mine = __builtin_popcountll(cand);
    if (__ballot(mine != 0) == 0ull) return;
    int mx = 0, tot = 0, lanes = 0;
    unsigned long long m = __ballot(1);
    const int first = __ffsll((long long)m) - 1;
    while (m) { const int l = __ffsll((long long)m) - 1; const int v = __shfl(mine, l, 64); mx = v > mx ? v : mx; tot += v; lanes += v != 0; m &= m - 1; }
    if ((int)(threadIdx.x & 63) == first) {
        atomicAdd(&g_dk[8], 1ull);
        atomicAdd(&g_dk[9], (unsigned long long)mx);
        atomicAdd(&g_dk[10], (unsigned long long)tot);
        atomicAdd(&g_dk[11], (unsigned long long)((tot + 63) / 64));
        atomicAdd(&g_dk[12], (unsigned long long)lanes);
    }
#else
    (void)cand;
#endif
}
// the ray-invariant terms of the quartic (rt.frag:445-455), from the ray in the torus' own frame
RT_HD TorusRay torus_ray_setup(const DevTorus& T, f3 ro, f3 rd)
{
    TorusRay w;
    w.a = dot3(rd, rd);
    w.b = dot3(ro, rd);
    w.c = dot3(ro, ro) + T.radii.z - T.radii.w;
    w.axy = dot2(rd.x, rd.y, rd.x, rd.y);
    w.bxy = dot2(ro.x, ro.y, rd.x, rd.y);
    w.cxy = dot2(ro.x, ro.y, ro.x, ro.y);
    w.k = T.k.x;
    return w;
}
// The iteration itself (rt.frag:462-485): the smallest iterate that looks like a real, non-negative root, 10000 if none does. Takes nothing but the
// seven ray terms -- which is what would let a solve run in ANOTHER lane than the ray's (round 5 built that as a workgroup-wide pool: profiles/r05s_torus_pool_ab.txt, not shipped).
RT_COLD float dk_solve(const TorusRay& w)
{
    const float eps = 0.001f;
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long _dk_t0 = clock64();
    int _dk_sweeps = 0;
#endif
    v2f c0 = mk2v(1.0f, 0.0f);
    v2f c1 = mk2v(0.4f, 0.9f);
    v2f c2 = cmul(c1, mk2v(0.4f, 0.9f));
    v2f c3 = cmul(c2, mk2v(0.4f, 0.9f));
    for (int i = 0; i < 60; i++) {
        float e = dk_step(c0, c1, c2, c3, w);
        e = gl_max(e, dk_step(c1, c2, c3, c0, w));
        e = gl_max(e, dk_step(c2, c3, c0, c1, w));
        e = gl_max(e, dk_step(c3, c0, c1, c2, w));
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
        _dk_sweeps++;
#endif
        if (e < eps) break;  // per-lane exit: a converged lane must stop updating its roots (trap T14)
    }
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
    {
        unsigned long long m = __ballot(1);
        const int first = __ffsll((long long)m) - 1;
        int mx = 0;
        while (m) { const int l = __ffsll((long long)m) - 1; const int v = __shfl(_dk_sweeps, l, 64); mx = v > mx ? v : mx; m &= m - 1; }
        atomicAdd(&g_dk[1], 1ull);
        atomicAdd(&g_dk[3], (unsigned long long)_dk_sweeps);
        if ((int)(threadIdx.x & 63) == first) {
            atomicAdd(&g_dk[0], 1ull);
            atomicAdd(&g_dk[2], (unsigned long long)mx);
            atomicAdd(&g_dk[4], (unsigned long long)clock64() - _dk_t0);
        }
        {   // [13] solver runs with a lane at the 60-sweep cap, [14] sum over those runs of the SECOND-largest distinct stop (sweeps the run would take without its capped lanes)
            const unsigned long long capm = __ballot(_dk_sweeps >= 60);
            if (capm != 0ull) {
                unsigned long long m2 = __ballot(_dk_sweeps < 60);
                int mx2 = 0;
                while (m2) { const int l = __ffsll((long long)m2) - 1; const int v = __shfl(_dk_sweeps, l, 64); mx2 = v > mx2 ? v : mx2; m2 &= m2 - 1; }
                if ((int)(threadIdx.x & 63) == first) { atomicAdd(&g_dk[13], 1ull); atomicAdd(&g_dk[14], (unsigned long long)mx2); }
            }
        }
        if (_dk_sweeps >= 60) atomicAdd(&g_dk[6], 1ull);
    }
#endif
    float r0 = c0[0], r1 = c1[0], r2 = c2[0], r3 = c3[0];
    if (fabsf(c0[1]) > eps || r0 < 0.0f) r0 = 10000.0f;
    if (fabsf(c1[1]) > eps || r1 < 0.0f) r1 = 10000.0f;
    if (fabsf(c2[1]) > eps || r2 < 0.0f) r2 = 10000.0f;
    if (fabsf(c3[1]) > eps || r3 < 0.0f) r3 = 10000.0f;
    return gl_min(gl_min(r0, r1), gl_min(r2, r3));
}
RT_HD bool torus_root_accepted(float t, float tmin) { return t > 0.0f && t < 100.0f && t < tmin; }   // rt.frag:486
RT_HD bool intersect_torus_local(const DevTorus& T, f3 ro, f3 rd, float tmin, float& t)
{
    t = dk_solve(torus_ray_setup(T, ro, rd));
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
    if (t > 0.0f && t < 100.0f && t < tmin) atomicAdd(&g_dk[7], 1ull);
    else if (t > 0.0f && t < 100.0f) atomicAdd(&g_dk[5], 1ull);   // real root, but beyond the limit
#endif
    return torus_root_accepted(t, tmin);
}
// Conservative pre-tests in WORLD space (no rotation needed): true = the exact test can be skipped.
// Rings and quadrics (geometric intersectors): a hit needs 0 < t < tlimit, so a bounding sphere that the ray's line misses, that lies
// behind the origin or that is entered beyond the limit rules the primitive out. Tori: see torus_cull below -- their solver is an
// iteration whose reported root need not be where the ray enters the tube, so they get no length limit.
// true = PROVABLY no point of the ray with 0 < t <= tlimit (plus slack) lies inside the sphere
// (centre c, squared radius r2, already inflated by the caller).
// The discriminant b*b - a*cc cancels catastrophically when the origin is far from the sphere
// (|oc| >> r): its rounding error is bounded by ~1e-6 * a * |oc|^2 (three-term dot products,
// 2^-24 per operation), so "misses" is only concluded when h is below minus ten times that.
// NaNs compare false -> "not culled".
// The culls are conservative predicates, not reference arithmetic: their dot products use explicit fused multiply-adds
// (half the instructions, smaller rounding error than the margins were derived for) and everything that depends on the
// ray direction only (dot(rd,rd), the six products of the quadric form) is loop-invariant and hoisted by the compiler.
RT_HD float dot3_fma(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// The earliest possible entry into the sphere, t_in = (-b - sqrt(hp)) / a (b < 0, hp >= 0 the padded discriminant), lies beyond the
// limit L = 1.001 tlimit + 0.01 + 1e-5 (1 + |oc|^2):  t_in > L  <=>  s = -b - a L > sqrt(hp)  <=>  s > 0 and s^2 > hp -- no square root and
// no division (round 3: the IEEE expansions of both were a fifth of this predicate's instructions). NaN anywhere -> false -> not culled.
RT_HD bool sphere_entry_beyond(float a, float b, float hp, float d2, float tlimit)
{
    const float L = fmaf(tlimit, 1.001f, fmaf(1e-5f, d2, 0.01001f));
    const float s = fmaf(-a, L, -b);
    return (s > 0.0f) & (s * s > hp);
}
// (round 6: straight-line. With early returns the compiler built six nested exec-mask regions per test -- s_and_saveexec / s_cbranch_execz
// pairs around two or three VALU instructions each, four tests per scan step; every operation here is cheap and has no side effect, so all
// of it is evaluated and the conditions are combined bitwise: the same predicate, NaNs still compare false -> "not culled".)
RT_HD bool sphere_cull(f3 c, float r2, f3 ro, f3 rd, float tlimit)
{
    const f3 oc = ro - c;
    const float d2 = dot3_fma(oc, oc);
    const float cc = d2 - r2;                 // > 0: origin outside
    const float a = dot3_fma(rd, rd);
    const float b = dot3_fma(oc, rd);
    const float h = fmaf(b, b, -(a * cc));
    const float err = 1e-5f * a * d2;
    // origin outside; not a degenerate direction (refract() yields the zero vector on a total reflection it disagrees about with the
    // Fresnel test: never cull)
    const bool may = (cc > 0.0f) & (a > 0.25f) & (a < 4.0f);
    // sphere behind the origin | the line misses the sphere, beyond rounding doubt | entered beyond the limit
    const bool out = (b >= 0.0f) | (h < -err) | sphere_entry_beyond(a, b, h + err, d2, tlimit);
    return may & out;
}
// The premise above holds for UNIT directions only. The shader's Durand-Kerner update divides by the product of root
// differences but not by the quartic's leading coefficient dot(rd,rd)^2, so for a direction of length L its steps are L^4
// times too long: for L^4 >= 2 the iteration does not converge at all, runs out of its 60 sweeps and reports whatever
// iterate happens to have a small imaginary part -- also for rays that miss the torus by a wide margin (found by the
// degenerate-scene fuzz: a ray refracted with a non-unit normal, |rd| = 1.29, going away from a torus 2 units to its side,
// "hit" it at t = 1.18). Such directions are never culled: the solver has to run to reproduce its own garbage.
RT_HD bool unit_direction(float dd) { return fabsf(dd - 1.0f) <= 1e-3f; }   // false for NaN
// THE RAY'S OWN LENGTH LIMIT IS NOT USED by any torus cull (round 5). A root is accepted for t < min(tmin, 100) only (rt.frag:486), and rounds
// 2-4 culled a torus whose inflated bound the ray enters beyond min(tmin, 100). That assumed that the solver reports a hit no earlier than the
// ray really enters the tube -- a statement about the reference's Durand-Kerner iteration that does not hold: a solve that runs out of its 60
// sweeps reports whatever iterate looks real, and on a ray that does cross the tube such an iterate can lie anywhere along it. Measured
// (tools/cull_audit.py family torus_lead, 2.9e10 reported hits, profiles/r05a_torus_lead_by_origin_distance_4e10.txt): reported more than
// 1e-3 t + 0.01 BEFORE the entry into the inflated tube -- 0 of 2.2e9 hits for origins within 2 units of the torus' centre, 4 of 5.5e9 at 2..4
// (one of them by 1.04), 52 of 3.2e9 at 4..6 (up to 4.8), 1 117 of 1.4e9 at 10..12 (up to 10.5), 281 037 of 5.4e9 at 24..48 (up to 46): rarer
// near the torus, where the quartic's coefficients are small and the iteration converges, but there is no distance from which on it is never.
// Round 4's audit met four such rays in 2.7e11 that its culls dropped (tests/golden/torus_far_rays.json; tests/test_culls.py,
// tests/test_gpu_culls.py). The reference never culls (rt.frag:462-487): whatever the solver reports below tmin is a hit. What is left:
//   * LATERAL: a torus is culled when the ray's forward half-line misses its inflated bound. Of another kind than the length premise -- a ray
//     that never comes near the tube has four complex roots, and an iterate would have to sit within 1e-3 of the real axis by chance: of 3.6e10
//     reported hits at every distance none belongs to a ray that stays outside the tube inflated by 1 % + 0.01 (the farthest phantom passes
//     3.8 mm from the real tube), and 1.3e11 laterally culled, solved rays of the torus family had no hit (profiles/r05b_*).
//   * THE REFERENCE'S OWN t < 100: a torus that the ray enters beyond RT_TORUS_REACH = 102.5 is culled whatever tmin is. An accepted root would
//     have to be reported more than 2.4 before the entry by a solve whose origin is more than 100 units out -- every such solve runs all 60
//     sweeps. Dropping this limit as well is exact by construction and was built and measured: the torus-heavy 4K frame takes 9.7 ms instead of
//     1.8 and the default frame 0.67 instead of 0.47 (the far floor and the planets: millions of rays from hundreds to 1e5 units out whose line
//     crosses a torus, each 60 sweeps), with bit-identical frames -- not one of those solves reported a root below 100
//     (profiles/r05b_torus_no_limit_at_all_cost_ab.txt). The premise is measured, not proven: tools/cull_audit.py family torus_far solves
//     rays that enter beyond the reach (DESIGN.md section 3 has the count).
// Cost of dropping the ray's own limit: torus-heavy 4K frame +3 %, default frame +0.3 % (profiles/r05a_torus_limit_rule_cost_ab.txt, `near0`).
#ifndef RT_TORUS_REACH
#define RT_TORUS_REACH 102.5f      /* 100 + 2.5 % (round 4's widening at t = 100); sphere_cull and the puck test add their own slack */
#endif
// "BEHIND" RAYS (round 6). A ray that points AWAY from a torus its backward extension goes through has four real NEGATIVE roots, and the
// reference still solves it (rt.frag:462-487 never culls). Mostly the iteration converges on them and reports nothing. But the quartic's
// coefficients grow as |o|^4, and once the float noise of a Durand-Kerner step exceeds the 1e-3 stop criterion the solve runs all 60 sweeps with
// its iterates jittering around the roots; when two of them nearly coincide in the last sweep one is thrown along the real axis -- now and then
// to a positive t below the limit: a hit where there is no torus, which the reference shows. Measured (tools/cull_audit.py family torus_behind,
// every ray of that kind solved; profiles/r06d_cull_audit_torus_behind_1e12.txt, 7.8e11 rays over 125 scenes): 66 phantoms from 2..4 units, 223
// from 4..6, 13 667 from 24..48 -- the rate falls towards the torus but there is NO distance from which on it is zero (round 5's "none within 4
// units" was 1.4e10 rays; two of the 66 come from 3.2 and 3.3 units of tori of R = 1 .. 1.3). Rule: "the torus lies behind the origin" is no
// reason to cull it. For an origin OUTSIDE the torus' inflated bounding sphere every test looks at the ray's whole LINE -- the backward half is
// culled only where it clears the inflated torus (then the roots are two complex pairs: the lateral premise of the comment above) or enters
// it beyond RT_TORUS_REACH_BACK (measured like the forward reach: family torus_behind_far, 1.1e11 rays from 104 .. 3000 units, no hit). An
// origin INSIDE the bounding sphere (a torus' own shadow and mirror rays: |o| <= R + r) keeps the half-line culls -- hull, puck, tube --
// whose premises were measured on exactly those rays.
#ifndef RT_TORUS_BEHIND_RULE
#define RT_TORUS_BEHIND_RULE 1      /* A/B switch: 0 = the culls of round 5 (a torus behind the origin is culled from any distance) */
#endif
#ifndef RT_TORUS_REACH_BACK
#define RT_TORUS_REACH_BACK 102.5f  /* how far back along the line a torus still has to be looked at (measured: tools/cull_audit.py family torus_behind_far) */
#endif
// sphere_cull(c, r2, ro, rd, RT_TORUS_REACH) with the "behind" rule: the origin is outside the sphere here, so a sphere behind it is judged by
// the REVERSED ray (-|b|) against the backward reach. Straight-line (see sphere_cull).
RT_HD bool torus_sphere_cull(f3 c, float r2, f3 ro, f3 rd)
{
    const float a = dot3_fma(rd, rd);
    const f3 oc = ro - c;
    const float d2 = dot3_fma(oc, oc);
    const float cc = d2 - r2;                 // > 0: origin outside (false also for a torus that is never culled, r2 = +inf, and for NaN)
    const float b = dot3_fma(oc, rd);
    const float h = fmaf(b, b, -(a * cc));
    const float err = 1e-5f * a * d2;
    const bool may = unit_direction(a) & (cc > 0.0f);
    const bool behind = b >= 0.0f;
    const float reach = RT_TORUS_REACH_BACK == RT_TORUS_REACH ? RT_TORUS_REACH : (behind ? RT_TORUS_REACH_BACK : RT_TORUS_REACH);
    // the LINE misses the sphere, beyond rounding doubt | round 5: behind the origin | the (reversed) ray enters it beyond the reach
    const bool out = (h < -err) | (behind & !RT_TORUS_BEHIND_RULE) | sphere_entry_beyond(a, RT_TORUS_BEHIND_RULE ? -fabsf(b) : b, h + err, d2, reach);
    return may & out;
}
RT_HD bool torus_cull(f4 bound, f3 ro, f3 rd)
{
    return torus_sphere_cull(xyz(bound), bound.w, ro, rd);
}
// A ring hit lies within sqrt(r2) of the ring centre (p < r2, rt.frag:384) and needs 0 < t < tmin;
// intersect_ring has no NaN-accepting path (all four comparisons must hold), so missing the
// inflated sphere means "false".
RT_HD bool ring_cull(f4 bound, f3 ro, f3 rd, float tlimit)
{
    return sphere_cull(xyz(bound), bound.w, ro, rd, tlimit);
}
// Second, tighter pre-test in the torus' own frame (axis = local z), after the rotation the solver
// needs anyway. The torus lies inside the "puck" |z| <= r, x^2+y^2 <= (R+r)^2 and outside the
// hole x^2+y^2 < (R-r)^2. true = the part of the ray with 0 < t <= RT_TORUS_REACH never meets the
// (1 %-inflated) puck, or crosses the puck's slab entirely inside the (deflated) hole. Margins as
// in sphere_cull. Same premise as torus_cull: Durand-Kerner reports no root for a geometric miss.
RT_HD float rt_sqrt_approx(float x)   // conservative predicates only: 1 ulp is as good as correctly rounded there
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_sqrtf(x);
#else
    return sqrtf(x);
#endif
}
// (the two halves are separate functions so that tools/audit can attribute a culled ray to the one that fired; both are inlined)
RT_HD bool torus_hull_cull(const DevTorus& T, f3 o, f3 d)
{
    // Round 3 -- the convex hull of the torus: the points within r of the disc of radius R in the plane z = 0. With q the point of that
    // disc nearest to the origin o and w = o - q, the disc lies in the half-space (x - q).w <= 0, so a ray with d.w >= 0 only ever moves
    // away from it: dist(o + t d, disc) >= |w| for every t >= 0. |w| >= r + margin (T.cull.y, squared) then keeps the whole half-line outside
    // the hull, and the torus inside it. This is the cull for the rays that START on a torus -- its own shadow and mirror rays, half of all
    // solves that survive the other culls -- wherever the surface is convex (the outer half of the tube, where w is the surface normal
    // and d.w > 0 for every shadow ray that is cast at all and every mirror ray). Outside the disc's rim (rho > R): q = R (o.x, o.y) / rho,
    // w = ((rho - R) o.x / rho, (rho - R) o.y / rho, o.z), compared after multiplying by rho > 0; above the disc: w = (0, 0, o.z).
    const float q2 = o.x * o.x + o.y * o.y, Ra = T.cull.z;
    const bool rim = q2 > Ra * Ra;
    const float rho = rt_sqrt_approx(q2), e = rim ? rho - Ra : 0.0f;
    const float w2 = e * e + o.z * o.z;
    const float away = rim ? e * (o.x * d.x + o.y * d.y) + rho * (o.z * d.z) : o.z * d.z;
    // w2 < FLT_MAX: tori that are never culled carry cull.y = +inf, and an origin beyond 1.8e19 overflows w2 to +inf as well (origins that
    // large follow a degenerate-quadric "hit", trap T4) -- "inf >= inf" must not cull them
    return w2 >= T.cull.y && w2 < RT_FLT_MAX && away >= 0.0f;   // NaN -> false -> not culled
}
// The quartic of the INFLATED torus (tube radius T.cull.x = 1.01 r + 0.01, the puck's half height) has no root on the ray's part t0 <= t <= t1
// (the part inside the inflated puck): all Bernstein coefficients of F(t) = (|p|^2 + R^2 - r'^2)^2 - 4 R^2 (p.x^2 + p.y^2), p = o + t d, over
// the interval -- and over its two halves -- are positive, so F > 0 there: the part stays outside the inflated tube. The polynomial is
// set up about the interval's midpoint, where |p| is of the torus' own size whatever the origin's distance, so float evaluates it to ~1e-6 of
// its terms; `eps` asks for 1e-4 of them. Same premise as every torus cull (measured in DESIGN.md section 3: a ray that clears the real tube by
// more than 3.7 mm is never reported as a hit; the inflation is 10 mm + 1 %).
// (rp: the tube radius the quartic is set up for; eps_rel: the share of the sum of the terms' magnitudes every coefficient has to exceed)
RT_HD bool torus_quartic_positive(const DevTorus& T, f3 o, f3 d, float t0, float t1, float rp, float eps_rel)
{
    const float tm = 0.5f * (t0 + t1), hl = 0.5f * (t1 - t0);
    if (!(hl >= 0.0f) || !(tm < 1.0e3f)) return false;
    const f3 p = mk3(fmaf(d.x, tm, o.x), fmaf(d.y, tm, o.y), fmaf(d.z, tm, o.z));
    const float k4 = T.k.x;                                                 // 4 R^2
    const float a = dot3_fma(d, d), b = dot3_fma(p, d), c = dot3_fma(p, p) + (T.radii.z - rp * rp);
    const float axy = fmaf(d.y, d.y, d.x * d.x), bxy = fmaf(p.y, d.y, p.x * d.x), cxy = fmaf(p.y, p.y, p.x * p.x);
    // F(tm + s) = q0 + q1 s + q2 s^2 + q3 s^3 + q4 s^4; with s = hl u, u in [-1, 1]: coefficients c_k = q_k hl^k
    const float h2 = hl * hl;
    const float c0 = fmaf(c, c, -(k4 * cxy));
    const float c1 = (4.0f * b * c - 2.0f * k4 * bxy) * hl;
    const float c2 = (4.0f * b * b + 2.0f * a * c - k4 * axy) * h2;
    const float c3 = (4.0f * a * b) * (h2 * hl);
    const float c4 = (a * a) * (h2 * h2);
    const float eps = eps_rel * (fabsf(c0) + fabsf(c1) + fabsf(c2) + fabsf(c3) + fabsf(c4)) + 1.0e-12f;
    // Bernstein coefficients over u in [-1, 1] (blossoms of the monomials at -1 / +1)
    const float b0 = c0 - c1 + c2 - c3 + c4, b4 = c0 + c1 + c2 + c3 + c4;
    const float b1 = c0 - 0.5f * c1 + 0.5f * c3 - c4, b3 = c0 + 0.5f * c1 - 0.5f * c3 - c4;
    const float b2 = c0 - c2 * (1.0f / 3.0f) + c4;
    if (!(b0 > eps && b4 > eps)) return false;                              // an end point inside the inflated tube (or NaN)
    // one de Casteljau split at u = 0: the control points of the two halves
    const float l1 = 0.5f * (b0 + b1), m1 = 0.5f * (b1 + b2), m2 = 0.5f * (b2 + b3), r3 = 0.5f * (b3 + b4);
    const float l2 = 0.5f * (l1 + m1), mm = 0.5f * (m1 + m2), r2 = 0.5f * (m2 + r3);
    const float l3 = 0.5f * (l2 + mm), r1 = 0.5f * (mm + r2);
    const float mid = 0.5f * (l3 + r1);
    return l1 > eps && l2 > eps && l3 > eps && mid > eps && r1 > eps && r2 > eps && r3 > eps;
}
// The tube test proper: the inflated tube (T.cull.x) over the ray's part inside the puck, for rays whose origin is outside the inflated tube.
// Round 5, last session -- START: the rays that start ON a torus and inside its convex hull (the inner half of the tube: its own shadow and
// mirror rays there; the hull cull takes the outer half). Measured on the host build (64 tori, 960 x 540, depth 6): 76 k of 282 k solves that
// survive every other cull start inside the inflated tube of their own torus, 66 k of them report no hit -- three quarters of ALL solves
// without a hit. Such an origin sits one hit bias (~1e-3) off the surface, inside the inflation, so the first stretch is judged against the
// tube of radius r + RT_TORUS_HULL_MARGIN (T.cull.w; the margin the hull cull's premise was measured with: a ray that starts >= 2.5e-4 off
// the surface and only moves away has no reported root) at a threshold of 2e-5 of the terms (F there is ~3e-4 of them; float gives 1e-6):
//     F_(r + margin) > 0 on [0, ts],  ts = min(t1, 4 (r' - r)),    and    F_(r') > 0 on [ts, t1]  (the inflated tube, as for every other ray)
// -- by the first the ray never comes back within the margin of the surface it left, by the second it has left the inflated tube at ts and
// stays outside it. One evaluation per pass of a two-trip loop (the second trip only for lanes whose stretch inside the puck is longer
// than ts), so that the code exists once.
#ifndef RT_TORUS_START
#define RT_TORUS_START 1        /* A/B switch: 0 = the tube test as it was before the START part (no ray that starts inside the inflated tube is culled) */
#endif
RT_HD bool torus_tube_cull(const DevTorus& T, f3 o, f3 d, float t0, float t1)
{
    const float rp = T.cull.x;
    const float q = rt_sqrt_approx(o.x * o.x + o.y * o.y) - T.cull.z, dist2 = q * q + o.z * o.z;   // distance^2 to the centre circle
    const bool start = dist2 < rp * rp;          // the origin is inside the inflated tube (hence inside the puck: t0 = 0)
    if (start && (!RT_TORUS_START || !(dist2 > T.cull.y))) return false;   // closer to the surface than the margin (or a torus that is never culled: +inf)
    const float ts = 4.0f * (rp - T.radii.y);
    bool ok = true, more = true;
    float ta = start ? 0.0f : t0, tb = start ? gl_min(t1, ts) : t1;
#pragma unroll 1
    for (int pass = 0; pass < 2; pass++) {
        const bool first = start && pass == 0;
        if (more) ok = ok && torus_quartic_positive(T, o, d, ta, tb, first ? T.cull.w : rp, first ? 2.0e-5f : 1.0e-4f);
        more = more && ok && start && pass == 0 && t1 > ts;
        if (!RT_ANY(more)) break;
        ta = ts;
        tb = t1;
    }
    return ok;
}
// (t0, t1: on a `false` return, the part of the ray inside the inflated puck and the reach -- what torus_tube_cull then looks at)
RT_HD bool torus_puck_cull(const DevTorus& T, f3 o, f3 d, float& t0, float& t1, float reach = RT_TORUS_REACH)
{
    t0 = 0.0f;
    t1 = reach * 1.001f + 0.01f;              // the reference's own t < 100, never the ray's limit (see torus_cull)
    // slab |z| <= hz
    const float hz = T.cull.x;
    if (d.z != 0.0f) {
        const float ta = (-hz - o.z) / d.z, tb = (hz - o.z) / d.z;
        t0 = gl_max(t0, gl_min(ta, tb));
        t1 = gl_min(t1, gl_max(ta, tb));
    } else if (fabsf(o.z) > hz) {
        return true;
    }
    if (t0 > t1) return true;
    // outer cylinder x^2 + y^2 <= Ro2
    const float a = d.x * d.x + d.y * d.y;
    const float b = o.x * d.x + o.y * d.y;
    const float q = o.x * o.x + o.y * o.y;
    const float c = q - T.k.z;
    const float err = 1e-5f * (a * q + b * b);
    // a = sin^2 of the angle between ray and torus axis. Below 1e-12 (the ray drifts < 1e-4 radially over the whole
    // 100-unit range, well inside the puck's inflation) the quadratic is treated as the parallel case: with a in the
    // denormal range -- a mirror ray with direction (4e-23, 2e-24, -1) did this -- b*b - a*c and the two roots are
    // rounding noise, and the interval came out empty for a ray that goes straight down through the tube.
    if (a > 1e-12f) {
        const float h = b * b - a * c;
        if (h < -err) return true;
        const float sh = sqrtf(gl_max(h + err, 0.0f));
        t0 = gl_max(t0, (-b - sh) / a);
        t1 = gl_min(t1, (-b + sh) / a);
        if (t0 > t1) return true;
    } else if (c > 0.0f) {
        return true;
    }
    // hole: radial distance^2 is convex in t, so its maximum over [t0,t1] is at an end point
    const float hole = T.k.w;
    if (hole > 0.0f) {
        const float x0 = o.x + d.x * t0, y0 = o.y + d.y * t0, x1 = o.x + d.x * t1, y1 = o.y + d.y * t1;
        if (x0 * x0 + y0 * y0 < hole && x1 * x1 + y1 * y1 < hole) return true;
    }
    return false;
}
RT_HD bool torus_puck_cull(const DevTorus& T, f3 o, f3 d)
{
    float t0, t1;
    return torus_puck_cull(T, o, d, t0, t1);
}
// TUBE: the Bernstein test of the inflated tube behind the puck test (round 4). Compiled into the many-primitive kernel variant (and the host
// build): 64 tori, 4K, depth 6: 5.25 M -> 4.47 M solves, 141 k -> 128 k solver runs, 1 904 -> 1 794 us; in the default variant its one torus gains
// a tenth fewer runs and loses as much to the 90 instructions per candidate pass (472 -> 477 us: not compiled in there).
// (the half-line o + t d, t >= 0, whichever way it points along the ray's line: torus_local_cull)
template <bool TUBE = true>
RT_HD bool torus_forward_cull(const DevTorus& T, f3 o, f3 d)
{
    if (torus_hull_cull(T, o, d)) return true;
    float t0, t1;
    if (torus_puck_cull(T, o, d, t0, t1)) return true;
    // Ring tori only (hole radius T.k.w > 0, i.e. R - r > 0.01): F = (D-^2 - r^2)(D+^2 - r^2) with D-, D+ the distances to the nearest and the
    // farthest point of the centre circle. With r' < R the second factor is positive everywhere and F' > 0 means "outside the inflated tube".
    // A horn or spindle torus (r >= R) has a second sheet { D+ = r } around its centre, INSIDE which F > 0 again: a ray that starts in there and
    // hits that sheet stays where the inflated quartic is positive. (The first form of this test had no such condition: the bench scenes and
    // the whole GPU suite were bit-identical, and tools/cull_audit.py counted 6.2 M culled hits in 7.9e9 culled rays -- every one on the
    // r >= R tori of tests/random_scenes.py nasty_scene.)
    // (The START part alone in the default kernel variant, whose one torus has no tube test: 459 -> 468 us at 4K, 198 -> 206 us at 1920 x 1080 --
    // measured and not compiled in, profiles/r05u_start_cull_ab.txt.)
    return TUBE && T.k.w > 0.0f && torus_tube_cull(T, o, d, t0, t1);
}
// The "behind" rule (torus_cull) in the torus' own frame. From an origin OUTSIDE the inflated bounding sphere exactly one half of the ray's line
// can meet the torus: with the centre ahead (o.d < 0) the backward half only moves away from the sphere it already is outside of, with the
// centre behind (o.d >= 0) the forward half does. So the one set of tests -- hull, puck, tube -- runs on the half-line that matters: the ray
// itself, or the REVERSED ray (whose roots are the negative ones the reference's solver may throw to the positive side). A sign error of
// o.d next to 0 lets the other half dip into the sphere by ~1e-14 |o|^2, nowhere near the torus inside it (the sphere is 1 % + 0.01 larger).
// An origin inside the sphere (a torus' own shadow and mirror rays) is judged on its forward half-line, as the premises were measured.
// (First form of round 6: forward tests, then the backward ones for every forward-culled far ray -- twice the work on most rays that reach
// this point, default 4K frame 474 us; this form: profiles/r06h_*.)
template <bool TUBE = true>
RT_HD bool torus_local_cull(const DevTorus& T, f3 o, f3 d)
{
    const float dd = dot3(d, d);
    if (!unit_direction(dd)) return false;  // not a unit direction: the solver's result is not geometric (see torus_cull)
    const bool back = RT_TORUS_BEHIND_RULE && (dot3(o, o) > T.k.z) & (dot3(o, d) >= 0.0f);     // T.k.z: the bounding sphere's radius^2
    static_assert(RT_TORUS_REACH_BACK == RT_TORUS_REACH, "torus_puck_cull takes one reach for both halves of the line");
    return torus_forward_cull<TUBE>(T, o, back ? mk3(-d.x, -d.y, -d.z) : d);
}
template <bool CULL, bool TUBE = true>
RT_HD bool intersect_torus_c(const DevTorus& T, f3 ro, f3 rd, float tmin, float& t, bool& solved)
{
    const bool ident = ident_flag(T.pos.w);
    const f3 o = quat_rotate_id(T.quat, ident, ro - xyz(T.pos));
    const f3 d = quat_rotate_id(T.quat, ident, rd);
    solved = false;
    if (CULL && torus_local_cull<TUBE>(T, o, d)) return false;
    solved = true;
    return intersect_torus_local(T, o, d, tmin, t);
}
RT_HD bool intersect_torus(const DevTorus& T, f3 ro, f3 rd, float tmin, float& t)
{
    bool solved;
    return intersect_torus_c<false>(T, ro, rd, tmin, t, solved);
}

RT_HD int lane_pop(unsigned long long& m)
{
    const int j = __builtin_ctzll(m);
    m &= m - 1ull;
    return j;
}

// ---- a lane's candidate tori in ONE solver loop (round 6; SURVEY section 7 step 5, "ray regeneration", inside the wave) ----
// The lane-divergent scans (calc_inter / in_shadow of the many-primitive variant) used to run one pass per candidate: every lane takes its
// next candidate, the wave solves, and the pass lasts as long as its slowest lane -- 0.57 of the lanes of a solver sweep did work on the
// 64-torus frame. The solve does not depend on the ray's limit (rt.frag:486 applies it afterwards), so a lane whose solve has ended can take
// its NEXT candidate -- rotation, culls, torus_ray_setup -- and re-enter the sweeps while the other lanes are still in theirs: no barrier, no
// LDS, no other wave. Per lane the sequence of operations is exactly the sequential one (candidates in index order, each tested against
// the limit as it stands when its solve ends, a converged lane stops updating its roots: trap T14), so results are bit-identical. Lanes are
// refilled in batches: sweeps run in bursts of RT_DK_BURST, and only between bursts do waiting lanes fetch their next candidate (a refill
// is ~350 instructions for however few lanes take part; convergence clusters at 10-15 sweeps, so most lanes of a run wait at the same burst).
// ANYHIT (in_shadow): the first accepted root ends the lane's list.
#ifndef RT_DK_BURST
#define RT_DK_BURST 2        /* (2 / 4 / 8 on the 64-torus 4K frame: 2 081 / 2 089 / 2 093 us; one pass per candidate: 2 164 -- profiles/r06b_dk_restart_ab.txt) */
#endif
#ifndef RT_DK_RESTART
#define RT_DK_RESTART 1      /* A/B switch: 0 = one solver pass per candidate (rounds 2-5) */
#endif
template <bool CULL, bool TUBE, bool COUNT, bool ANYHIT>
RT_HD void torus_run_candidates(const SceneView& S, int base, unsigned long long cand, f3 ro, f3 rd, float& tmin, int& num, bool& hit, LaneCounters& cnt)
{
    const float eps = 0.001f;
    // Phase A: the culls of every candidate, before any solver state exists (their temporaries -- the Bernstein test alone holds ~30 values --
    // must not share the registers with four complex iterates and seven ray terms): what is left is the list of tori this lane SOLVES.
    unsigned long long todo = 0ull;
    if (CULL) {
        while (RT_ANY(cand != 0ull)) {
            if (cand != 0ull) {
                const int j = lane_pop(cand);
                const DevTorus& T = S.tori()[base + j];
                const bool ident = ident_flag(T.pos.w);
                const f3 o = quat_rotate_id(T.quat, ident, ro - xyz(T.pos));
                const f3 d = quat_rotate_id(T.quat, ident, rd);
                if (!torus_local_cull<TUBE>(T, o, d)) todo |= 1ull << j;
            }
        }
    } else {
        todo = cand;
    }
    // Phase B: one solver loop over the lane's list. A refill is the rotation and the seven dot products of torus_ray_setup only.
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
    bool _dk_run_counted = false;
#endif
    bool active = false;
    int sweeps = 0, cur = 0;
    TorusRay w;
    w.a = w.b = w.c = w.axy = w.bxy = w.cxy = w.k = 0.0f;
    v2f c0 = mk2v(0.0f, 0.0f), c1 = c0, c2 = c0, c3 = c0;
    for (;;) {
        if (RT_ANY(!active && todo != 0ull)) {
            if (!active && todo != 0ull) {
                cur = base + lane_pop(todo);
                const DevTorus& T = S.tori()[cur];
                const bool ident = ident_flag(T.pos.w);
                w = torus_ray_setup(T, quat_rotate_id(T.quat, ident, ro - xyz(T.pos)), quat_rotate_id(T.quat, ident, rd));
                c0 = mk2v(1.0f, 0.0f);                              // rt.frag:463-466
                c1 = mk2v(0.4f, 0.9f);
                c2 = cmul(c1, mk2v(0.4f, 0.9f));
                c3 = cmul(c2, mk2v(0.4f, 0.9f));
                sweeps = 0;
                active = true;
                if (COUNT) cnt.torus_solves++;
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
                atomicAdd(&g_dk[1], 1ull);                          // lane solves
#endif
            }
        }
        if (!RT_ANY(active)) break;
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
        if (!_dk_run_counted) { _dk_run_counted = true; if ((int)(threadIdx.x & 63) == __ffsll((long long)__ballot(1)) - 1) atomicAdd(&g_dk[0], 1ull); }   // solver runs (one per scan now)
#endif
#pragma unroll 1
        for (int k = 0; k < RT_DK_BURST; k++) {
#if defined(RT_DK_STATS) && defined(__HIP_DEVICE_COMPILE__)
            {   // wave sweeps, lane sweeps
                const unsigned long long am = __ballot(active), all = __ballot(1);
                if ((int)(threadIdx.x & 63) == __ffsll((long long)all) - 1) { atomicAdd(&g_dk[2], 1ull); atomicAdd(&g_dk[3], (unsigned long long)__builtin_popcountll(am)); }
            }
#endif
            if (active) {
                float e = dk_step(c0, c1, c2, c3, w);              // one sweep, rt.frag:468-477
                e = gl_max(e, dk_step(c1, c2, c3, c0, w));
                e = gl_max(e, dk_step(c2, c3, c0, c1, w));
                e = gl_max(e, dk_step(c3, c0, c1, c2, w));
                sweeps++;
                if (e < eps || sweeps >= 60) {                      // this lane's solve has ended (rt.frag:478, the cap of rt.frag:467)
                    float r0 = c0[0], r1 = c1[0], r2 = c2[0], r3 = c3[0];
                    if (fabsf(c0[1]) > eps || r0 < 0.0f) r0 = 10000.0f;
                    if (fabsf(c1[1]) > eps || r1 < 0.0f) r1 = 10000.0f;
                    if (fabsf(c2[1]) > eps || r2 < 0.0f) r2 = 10000.0f;
                    if (fabsf(c3[1]) > eps || r3 < 0.0f) r3 = 10000.0f;
                    const float t = gl_min(gl_min(r0, r1), gl_min(r2, r3));
                    if (torus_root_accepted(t, tmin)) {
                        hit = true;
                        if (ANYHIT) todo = 0ull;
                        else { num = cur; tmin = t; }
                    }
                    active = false;
                }
            }
            if (!RT_ANY(active)) break;
        }
    }
}

// ---- general quadric (rt.frag:499-572) ----
RT_HD bool is_between(f3 v, f3 lo, f3 hi) { return (v.x > lo.x && v.y > lo.y && v.z > lo.z) && (v.x < hi.x && v.y < hi.y && v.z < hi.z); }
// WAVE = false (tools/audit only): the early exits below are taken by each lane on its own condition, so that an audit whose lanes hold
// unrelated rays exercises them for every ray (a wave of the product leaves only if ALL its lanes may).
template <bool WAVE = true>
RT_HD bool intersect_surface(const DevSurface& Q, f3 ro_w, f3 rd_w, float tmin, float& t)
{
    const bool ident = ident_flag(Q.vmax.w);
    const f3 ro = quat_rotate_id(Q.quat, ident, ro_w - xyz(Q.pos_a));
    const f3 rd = quat_rotate_id(Q.quat, ident, rd_w);
    const float a = Q.pos_a.w, b = Q.bcde.x, c = Q.bcde.y, d = Q.bcde.z, e = Q.bcde.w, f = Q.f_vmin.x;
    const float d1 = rd.x, d2 = rd.y, d3 = rd.z, o1 = ro.x, o2 = ro.y, o3 = ro.z;
    const float p1 = 2.0f * a * d1 * o1 + 2.0f * b * d2 * o2 + 2.0f * c * d3 * o3 + d * d3 + d2 * e;
    const float p2 = a * d1 * d1 + b * d2 * d2 + c * d3 * d3;
    const float p3 = a * o1 * o1 + b * o2 * o2 + c * o3 * o3 + d * o3 + e * o2 + f;
    if (fabsf(p2) < 1e-6f) {  // trap T4: inverted comparison, no clip test
        t = -p3 / p1;
        return t > tmin;
    }
    // No real root (negative discriminant): the square root below is NaN, neither root passes its tests, mn = mx = FLT_MAX, and whatever the
    // clip test makes of the "point" at FLT_MAX the result is FLT_MAX < tmin -- false for every finite limit. A wave none of whose lanes
    // has a real root (the line misses the unclipped quadric: common for the lanes a bounding sphere lets through) stops here, before the
    // square root, the two divisions and the clip tests; t is left as the long way round leaves it.
    const float disc = p1 * p1 - 4.0f * p2 * p3;
    // (Round 4 also tried leaving here for real roots of which provably none lies above epsilon -- Descartes' rule on the float values p1, p2, p3,
    // and the noise case F(origin) ~ 0 of a quadric's own shadow rays: exact, 8 % of all rays and 41 % of those that start on their quadric
    // leave by it, audited on 2e10 rays -- and measured no gain: a wave leaves only if ALL its lanes may. profiles/r04a_solver_sweep_ab.txt item 10.)
    const bool may_hit = !(disc < 0.0f) || tmin > RT_FLT_MAX;
    if (!(WAVE ? RT_ANY(may_hit) : may_hit)) { t = RT_FLT_MAX; return false; }
    const float p4 = sqrtf(disc);
    float mn = RT_FLT_MAX, mx = RT_FLT_MAX;
    const float t1 = (-p1 - p4) / (2.0f * p2);
    const float t2 = (-p1 + p4) / (2.0f * p2);
    const float epsilon = 1e-4f;
    if (t1 > epsilon && t1 < mn) { mn = t1; mx = t2; }
    if (t2 > epsilon && t2 < mn) { mn = t2; mx = t1; }
    // checkSurfaceEdges (rt.frag:500-512) on the WORLD-space ray (trap T6)
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
// Conservative pre-test: true = intersect_surface would return false. Only for quadrics whose
// clip box is finite on all axes (bound.w >= 0). Two facts are needed:
//  (1) the degenerate branch |p2| < 1e-6 (trap T4), which ignores the clip box, is not taken:
//      p2 = d^T M d with M = R^T diag(a,b,c) R, evaluated here from the hoisted symmetric M;
//      the two evaluations differ by rounding only, far less than the stored margin;
//  (2) past that branch a hit needs a point of the ray strictly inside the clip box
//      (checkSurfaceEdges); if the ray's LINE misses the box's inflated bounding sphere there is none.
// Round 3: the test is one of the ray's SEGMENT, not of its line. intersect_surface accepts a root only for epsilon < t < tlimit (closest
// hit so far / distance to the light) at a point inside the clip box, so a bound that lies behind the origin, or that the ray enters
// beyond the limit, settles the quadric like a line that misses it (the closest-hit scan passes the tmin of the moment, which is the very
// value intersect_surface would compare with).
// safe (out): the degenerate branch is ruled out for this direction (fact (1) below holds), i.e. a hit needs a point of the ray strictly
// inside the clip box -- what surface_box_miss may then use.
RT_HD bool surface_cull(const DevSurfaceCull& Q, f3 ro, f3 rd, float tlimit, bool& safe)
{
    // (round 6: straight-line like sphere_cull -- every early return was an exec-mask region of its own; same predicate)
    // p2 ~ d^T M d from the six direction products (ray-invariant) and the symmetric M of this quadric
    const float dxx = rd.x * rd.x, dyy = rd.y * rd.y, dzz = rd.z * rd.z;
    const float dxy = 2.0f * (rd.x * rd.y), dxz = 2.0f * (rd.x * rd.z), dyz = 2.0f * (rd.y * rd.z);
    const float p2 = fmaf(Q.sym1.x, dyz, fmaf(Q.sym0.z, dxz, fmaf(Q.sym0.y, dxy, fmaf(Q.sym1.y, dzz, fmaf(Q.sym0.w, dyy, Q.sym0.x * dxx)))));
    const float a = dot3_fma(rd, rd);
    // a bound exists; not too close to the degenerate branch (else: run the full test); not a degenerate direction (never cull)
    safe = (Q.bound.w >= 0.0f) & (fabsf(p2) > Q.sym1.z) & (a > 0.25f) & (a < 4.0f);
    const f3 oc = ro - xyz(Q.bound);
    const float b = dot3_fma(oc, rd);
    const float d2 = dot3_fma(oc, oc);
    // near origins: the tight bound (it may rest on the fattened surface's own extent, which grows with the distance it is looked at from);
    // far ones: the bound of the closed clip box, if there is one (NaN: none -- the comparisons below then never cull)
    const float cc = d2 - (d2 <= (float)(RT_QUADRIC_FAR * RT_QUADRIC_FAR) ? Q.bound.w : Q.sym1.w);
    const float h = fmaf(b, b, -(a * cc)), err = 1e-5f * a * d2;
    // the line misses (rounding-safe, see sphere_cull; NaN -> false -> not culled) | origin outside the bound, and the bound behind the origin
    // or entered beyond the limit
    const bool out = (h < -err) | ((cc > 0.0f) & ((b >= 0.0f) | sphere_entry_beyond(a, b, h + err, d2, tlimit)));
    return safe & out;
}
RT_HD bool surface_cull(const DevSurfaceCull& Q, f3 ro, f3 rd, float tlimit)
{
    bool safe;
    return surface_cull(Q, ro, rd, tlimit, safe);
}
// Second conservative pre-test, for lanes the sphere test lets through and whose direction rules the degenerate branch out (`safe`): the
// reference accepts a root only at a point STRICTLY inside the world-space clip box (checkSurfaceEdges, rt.frag:500-512, on pt = rd t + ro in
// float) with 1e-4 < t < tlimit. true = the ray's part [0, tlimit] misses the box inflated by more than pt's rounding can move a point
// (relative 1e-5 of |origin| + |box|, absolute 1e-5), so there is no such point. Slab test with approximate reciprocals (the margins
// cover them); NaNs from 0 x inf drop out of fminf / fmaxf, i.e. that axis says nothing. Unbounded box axes are +-FLT_MAX and never cull.
RT_HD bool surface_box_miss(const DevSurface& Q, f3 ro, f3 rd, float tlimit)
{
    const f3 lo = mk3(Q.f_vmin.y, Q.f_vmin.z, Q.f_vmin.w), hi = xyz(Q.vmax);
    float t0 = 0.0f, t1 = tlimit;
    const float o[3] = {ro.x, ro.y, ro.z}, d[3] = {rd.x, rd.y, rd.z}, l[3] = {lo.x, lo.y, lo.z}, h[3] = {hi.x, hi.y, hi.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float pad = fmaf(1.0e-5f, fabsf(o[k]) + fabsf(l[k]) + fabsf(h[k]), 1.0e-5f);
#if defined(__HIP_DEVICE_COMPILE__)
        const float inv = __builtin_amdgcn_rcpf(d[k]);
#else
        const float inv = 1.0f / d[k];
#endif
        const float ta = ((l[k] - pad) - o[k]) * inv, tb = ((h[k] + pad) - o[k]) * inv;
        t0 = fmaxf(t0, fminf(ta, tb));
        t1 = fminf(t1, fmaxf(ta, tb));
    }
    return t0 > fmaf(1.0e-5f, fabsf(t1), t1) + 1.0e-6f;   // (NaN -> false -> not culled)
}

// ---- second-level culls for long tables (RT_GROUP consecutive primitives under one sphere, built by the packer) ----
// A scene with 64 tori or 96 quadrics spends most of its instructions on one first-level cull per primitive and ray. A group sphere
// contains its members' (inflated) bounds, so "provably misses the group" implies "provably misses every member's bound" -- the
// conclusion each member's own cull would reach. Groups are runs of consecutive indices and are visited in order: scan order, the
// strict-< tie rule and every per-lane test sequence stay what they were.
//  * tori: a wave none of whose lanes can reach the group skips the group's eight first-level tests. Same preconditions as
//    torus_cull (unit direction only: for other directions the solver's answer is not geometric and nothing may be culled).
//  * quadrics: missing the bounds does not settle a quadric -- its degenerate branch (trap T4, |p2| < 1e-6) ignores the clip box --
//    so a skipped group still evaluates the p2 pre-check of every member (quadric_may_degenerate: the first third of surface_cull) and
//    runs the exact test for a lane whose direction is that close to a member's asymptotic cone.
RT_HD bool torus_group_cull(f4 g, f3 ro, f3 rd)
{
    if (!(g.w >= 0.0f)) return false;   // a member that is never culled (zero tube, non-unit quaternion): neither is the group
    return torus_sphere_cull(xyz(g), g.w, ro, rd);
}
RT_HD bool surface_group_cull(f4 g, f3 ro, f3 rd)
{
    if (!(g.w >= 0.0f)) return false;
    const float a = dot3_fma(rd, rd);
    if (!(a > 0.25f && a < 4.0f)) return false;
    const f3 oc = ro - xyz(g);
    const float b = dot3_fma(oc, rd);
    const float d2 = dot3_fma(oc, oc);
    return fmaf(b, b, -(a * (d2 - g.w))) < -1e-5f * a * d2;
}
RT_HD float quadric_p2(const DevSurfaceCull& Q, f3 rd)
{
    const float dxx = rd.x * rd.x, dyy = rd.y * rd.y, dzz = rd.z * rd.z;
    const float dxy = 2.0f * (rd.x * rd.y), dxz = 2.0f * (rd.x * rd.z), dyz = 2.0f * (rd.y * rd.z);
    return fmaf(Q.sym1.x, dyz, fmaf(Q.sym0.z, dxz, fmaf(Q.sym0.y, dxy, fmaf(Q.sym1.y, dzz, fmaf(Q.sym0.w, dyy, Q.sym0.x * dxx)))));
}
RT_HD bool quadric_may_degenerate(const DevSurfaceCull& Q, f3 rd)
{
    return !(fabsf(quadric_p2(Q, rd)) > Q.sym1.z);
}
#ifndef RT_GROUP_MIN
#define RT_GROUP_MIN 16   /* tables shorter than this keep the one-level scan */
#endif

// ---- third level: ray pencils (rt_scene_dev.h DevPencil) ----
// A cell's mask promises: bit i clear => for EVERY ray of the cell that passes the lane-level preconditions below, primitive i's own
// first-level test (surface_cull / torus_cull) would return "culled". The scans only ever visit FEWER primitives than the two-level
// scan would, in the same index order, and every visited primitive still runs its own first-level test: results cannot change.
//  * cube-map cells (APEX): a ray of the pencil lies on a line through the apex whose direction is in the cell; the builder compares
//    the angle between the cell's axis and the centre of the primitive's bound with (cell half-angle + angular radius of the bound).
//    A shadow ray ends at the light only up to rounding: it passes the apex within ulp(|origin|) + 1e-7 * length, which the lane-level
//    limits (|origin| <= 1e3, length <= 1e3) keep below the 4e-3 the builder adds to every radius.
//  * plane cells (PARALLEL): the ray is the line origin + t * a with the pencil's own direction a, i.e. ONE point of the plane across
//    a; the builder compares the projected bound with the cell's rectangle (the outermost cells reach to infinity).
//  * quadrics also need "not on the degenerate branch" (trap T4, see surface_cull): p2 = d^T M d varies by at most 2 |M| theta over unit
//    directions within theta of the cell axis, so |p2(axis)| > 2 |M| theta + margin rules the branch out for the whole cell; for a
//    PARALLEL pencil the direction is the same for every ray and the test is the kernel's own.
//  * preconditions every lane checks itself: unit direction (|d.d - 1| <= 1e-3: the torus premise, and what the p2 bound assumes);
//    anything else -- NaNs included -- reads the all-ones cell, i.e. falls back to the full table.
RT_HD f3 pencil_face_dir(int face, float u, float v)
{
    const float s = (face & 1) ? -1.0f : 1.0f;
    return (face >> 1) == 0 ? mk3(s, u, v) : ((face >> 1) == 1 ? mk3(u, s, v) : mk3(u, v, s));
}
RT_HD uint32_t pencil_cell_apex(const DevPencil& P, uint32_t stride, f3 w_in, bool ok)
{
    const f3 w = ok ? w_in : mk3(1.0f, 0.0f, 0.0f);
    const float ax = fabsf(w.x), ay = fabsf(w.y), az = fabsf(w.z);
    int face;
    float m, a, b;
    if (ax >= ay && ax >= az) { face = w.x < 0.0f ? 1 : 0; m = ax; a = w.y; b = w.z; }
    else if (ay >= az) { face = w.y < 0.0f ? 3 : 2; m = ay; a = w.x; b = w.z; }
    else { face = w.z < 0.0f ? 5 : 4; m = az; a = w.x; b = w.y; }
    const int R = P.res;
    const float half = 0.5f * (float)R, top = (float)(R - 1);
    const float k = half / m;
    const int i = (int)gl_min(gl_max(a * k + half, 0.0f), top), j = (int)gl_min(gl_max(b * k + half, 0.0f), top);
    const uint32_t cell = ok ? (uint32_t)((face * R + j) * R + i) : P.cells;
    return P.mask_off + cell * stride;
}
RT_HD uint32_t pencil_cell_parallel(const DevPencil& P, uint32_t stride, f3 pt_in, bool ok)
{
    const f3 pt = ok ? pt_in : mk3(0.0f, 0.0f, 0.0f);
    const float top = (float)(P.res - 1);
    const float u = (dot3_fma(pt, xyz(P.e1)) - P.e1.w) * P.grid.x, v = (dot3_fma(pt, xyz(P.e2)) - P.e2.w) * P.grid.y;
    const int i = (int)gl_min(gl_max(u, 0.0f), top), j = (int)gl_min(gl_max(v, 0.0f), top);
    const uint32_t cell = ok ? (uint32_t)(j * P.res + i) : P.cells;
    return P.mask_off + cell * stride;
}
// lane-level preconditions (see above); false for NaN
RT_HD bool pencil_ray_ok(f3 ro, f3 rd, float len) { return unit_direction(dot3_fma(rd, rd)) && dot3_fma(ro, ro) <= 1.0e6f && len <= 1.0e3f; }

// OR over the wave's participating lanes (wave-uniform result). Lanes of a wave mostly share a handful of cells, so "take the first
// lane that still has something new" finishes in a few rounds; butterfly reductions would need every lane of the wave enabled.
RT_HD uint32_t wave_or(uint32_t own, bool on)
{
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t rem = on ? own : 0u, uni = 0u;
    unsigned long long b;
    while ((b = __ballot(rem != 0u)) != 0ull) {
        const uint32_t s = (uint32_t)__builtin_amdgcn_readlane((int)rem, __builtin_ctzll(b));
        uni |= s;
        rem &= ~s;
    }
    return uni;
#else
    return on ? own : 0u;
#endif
}

#if defined(RT_SCAN_STATS) && defined(__HIPCC__)
// diagnostic build only (tools/scan_stats.py): how long are the quadric candidate lists of a WAVE against those of its lanes?
// g_scan[kind][..]: kind 0 closest-hit/pencil, 1 closest-hit/slab tables, 2 shadow/pencil, 3 shadow/slab tables;
// [0] wave-level word walks, [1] set bits of the wave's OR, [2] set bits of the lanes' own words, [3] participating lanes,
// [4] largest lane count per walk, [5] second-level runs (some lane needs the exact test), [6] lanes in those runs, [7] lanes whose exact test HIT
__device__ unsigned long long g_scan[4][8];
#endif
RT_HD void scan_stats_word(int kind, uint32_t own, bool on, uint32_t uni)
{
#if defined(RT_SCAN_STATS) && defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long act = __ballot(on);
    if (act == 0ull) return;
    const int mine = on ? __builtin_popcount(own) : 0;
    int mx = 0; unsigned long long m = act, tot = 0ull;
    while (m) { const int l = __builtin_ctzll(m); const int v = __shfl(mine, l, 64); mx = v > mx ? v : mx; tot += (unsigned long long)v; m &= m - 1ull; }
    if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) {
        atomicAdd(&g_scan[kind][0], 1ull);
        atomicAdd(&g_scan[kind][1], (unsigned long long)__builtin_popcount(uni));
        atomicAdd(&g_scan[kind][2], tot);
        atomicAdd(&g_scan[kind][3], (unsigned long long)__builtin_popcountll(act));
        atomicAdd(&g_scan[kind][4], (unsigned long long)mx);
    }
#endif
}
RT_HD void scan_stats_hit(int kind)
{
#if defined(RT_SCAN_STATS) && defined(__HIP_DEVICE_COMPILE__)
    atomicAdd(&g_scan[kind][7], 1ull);
#endif
}
RT_HD void scan_stats_level2(int kind, bool need)
{
#if defined(RT_SCAN_STATS) && defined(__HIP_DEVICE_COMPILE__)
    const unsigned long long nb = __ballot(need);
    if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) {
        atomicAdd(&g_scan[kind][5], 1ull);
        atomicAdd(&g_scan[kind][6], (unsigned long long)__builtin_popcountll(nb));
    }
#endif
}

// ---- the builder: per pencil one record per primitive (pencil_prim), then one mask word of one cell at a time (pencil_cell_word). On the device a
// workgroup prepares the records in LDS and each of its threads fills one cell (rt_kernel.hip); the host build loops. ----
struct PencilPrim {
    f4 a;   // APEX: unit vector from the apex to the centre of the bound, w = sin(alpha), alpha = angular radius of the (padded) bound
            // PARALLEL: x, y = the centre across the direction, z = padded radius, w = |p2| of the pencil's direction (quadrics)
    f4 b;   // x = cos(alpha); y = 1: set in every cell (no finite bound / the apex is inside it / NaNs); z = |M|_F, w = p2 margin (quadrics); tori: z = 1 if the apex is a far origin (torus_cull, "behind" rule)
};
RT_HD f3 cross3(f3 a, f3 b) { return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
RT_HD float angle_between(f3 a, f3 b) { return atan2f(length3(cross3(a, b)), dot3(a, b)); }   // well-conditioned at 0 and pi, unlike acos
RT_HD PencilPrim pencil_prim(const DevPencil& P, f4 bound, const DevSurfaceCull* Q)
{
    PencilPrim r;
    r.a = mk4(0.0f, 0.0f, 0.0f, 0.0f);
    r.b = mk4(0.0f, 1.0f, 0.0f, 0.0f);
    if (Q) {
        r.b.z = sqrtf(Q->sym0.x * Q->sym0.x + Q->sym0.w * Q->sym0.w + Q->sym1.y * Q->sym1.y +
                      2.0f * (Q->sym0.y * Q->sym0.y + Q->sym0.z * Q->sym0.z + Q->sym1.x * Q->sym1.x));
        r.b.w = 1.01f * Q->sym1.z + 1.0e-5f * r.b.z;
        if (P.kind == RT_PENCIL_PARALLEL) r.a.w = fabsf(quadric_p2(*Q, xyz(P.a)));
    }
    if (!(bound.w >= 0.0f)) return r;                                  // no bound (or NaN): every cell
    if (Q && !(Q->sym1.w >= 0.0f)) return r;                           // no bound that holds for origins at any distance: every cell
    const f3 c = xyz(bound);
    const float rad = sqrtf(Q ? Q->sym1.w : bound.w) + 4.0e-3f + 1.0e-6f * (fabsf(c.x) + fabsf(c.y) + fabsf(c.z));
    if (P.kind == RT_PENCIL_APEX) {
        const f3 v = c - xyz(P.a);
        const float D2 = dot3(v, v);
        if (!(D2 > 1.01f * rad * rad) || !(D2 < 1.0e30f)) return r;    // apex inside (or almost inside) the bound, infinite radius, NaN
        const float D = sqrtf(D2), sa = rad / D;                        // sin(alpha) < 0.996: alpha is well-conditioned
        r.a = mk4(v.x / D, v.y / D, v.z / D, sa);
        r.b.x = sqrtf(1.0f - sa * sa);
        // tori: the apex is outside the bound (it is, here), so torus_cull's "behind" rule looks at the whole line of a ray that STARTS at the
        // apex -- the torus stays a candidate in the cells its bound's antipodal cone meets as well
        if (!Q && RT_TORUS_BEHIND_RULE) r.b.z = 1.0f;
    } else {
        const float cu = dot3(c, xyz(P.e1)), cv = dot3(c, xyz(P.e2));
        const float rr = rad + 1.0e-3f * gl_max(P.grid.z, P.grid.w);
        if (!(fabsf(cu) < 1.0e30f && fabsf(cv) < 1.0e30f && rr < 1.0e30f)) return r;
        r.a.x = cu; r.a.y = cv; r.a.z = rr;
    }
    r.b.y = 0.0f;
    return r;
}
// record k of pencil P: quadrics 0 .. n_surface-1, then tori
RT_HD PencilPrim pencil_prim_at(const SceneView& S, const DevPencil& P, int k)
{
    const int ns = S.h->n_surface;
    if (k < ns) { const DevSurfaceCull Q = S.surf_cull()[k]; return pencil_prim(P, Q.bound, &Q); }
    return pencil_prim(P, S.torus_bound()[k - ns], nullptr);
}
struct PencilCell {     // geometry of one cell
    f3 axis;            // APEX: unit direction of the cell centre ...
    float theta, ct, st;   // ... and the half-angle (plus slack) of the cone around it that holds the cell, its cosine and sine
    float ulo, uhi, vlo, vhi;   // PARALLEL: the cell's rectangle (outer cells: +-inf)
};
RT_HD PencilCell pencil_cell_geometry(const DevPencil& P, uint32_t cell)
{
    PencilCell C;
    C.axis = mk3(0.0f, 0.0f, 1.0f); C.theta = 0.0f; C.ct = 1.0f; C.st = 0.0f; C.ulo = C.uhi = C.vlo = C.vhi = 0.0f;
    const int R = P.res;
    if (P.kind == RT_PENCIL_APEX || P.kind == RT_PENCIL_DIRECTION) {   // cube-map cells
        const int face = (int)cell / (R * R), j = ((int)cell / R) % R, i = (int)cell % R;
        const float step = 2.0f / (float)R;
        const float u0 = -1.0f + step * (float)i, u1 = -1.0f + step * (float)(i + 1), v0 = -1.0f + step * (float)j, v1 = -1.0f + step * (float)(j + 1);
        C.axis = normalize3(pencil_face_dir(face, 0.5f * (u0 + u1), 0.5f * (v0 + v1)));
        float th = angle_between(C.axis, pencil_face_dir(face, u0, v0));
        th = gl_max(th, angle_between(C.axis, pencil_face_dir(face, u1, v0)));
        th = gl_max(th, angle_between(C.axis, pencil_face_dir(face, u0, v1)));
        th = gl_max(th, angle_between(C.axis, pencil_face_dir(face, u1, v1)));
        C.theta = th + 2.0e-3f;   // slack: the lookup's rounding at cell borders, |d| within 1e-3 of 1, the builder's own arithmetic
        C.ct = cosf(C.theta); C.st = sinf(C.theta);
    } else {
        const int j = (int)cell / R, i = (int)cell % R;
        const float inf = __builtin_huge_valf();
        C.ulo = i == 0 ? -inf : P.e1.w + P.grid.z * (float)i;
        C.uhi = i == R - 1 ? inf : P.e1.w + P.grid.z * (float)(i + 1);
        C.vlo = j == 0 ? -inf : P.e2.w + P.grid.w * (float)j;
        C.vhi = j == R - 1 ? inf : P.e2.w + P.grid.w * (float)(j + 1);
    }
    return C;
}
// mask word w (quadric words first, then torus words) of cell `cell`; cell == P.cells is the all-ones cell: every primitive that exists
RT_HD uint32_t pencil_cell_word(const SceneView& S, const DevPencil& P, const PencilPrim* prims, const PencilCell& C, uint32_t cell, int w)
{
    const int ns = S.h->n_surface, nt = S.h->n_torus;
    const int nws = (ns + 31) >> 5;
    const bool quadrics = w < nws;
    const int first = quadrics ? w * 32 : (w - nws) * 32, count = (quadrics ? ns : nt) - first;   // primitives first .. of this class
    if (cell >= P.cells) return count >= 32 ? ~0u : (1u << count) - 1u;
    if (P.kind == RT_PENCIL_DIRECTION) {      // quadrics only: "some direction of the cell may take the degenerate branch"
        uint32_t deg = 0u;
        for (int b = 0; quadrics && b < 32 && b < count; b++) {
            const PencilPrim pp = prims[first + b];
            if (!(fabsf(quadric_p2(S.surf_cull()[first + b], C.axis)) > 2.01f * pp.b.z * C.theta + pp.b.w)) deg |= 1u << b;
        }
        return deg;
    }
    const bool apex = P.kind == RT_PENCIL_APEX;
    // "the cell's rays miss the bound": APEX -- the angle between the cell axis and the centre exceeds theta + alpha, compared through
    // cosines (cos(theta + alpha) = ct * cos(alpha) - st * sin(alpha); theta + alpha < pi/2 + 0.1; the 4e-6 covers the rounding of both
    // sides, 3e-4 rad at the smallest theta + alpha there is, inside theta's slack); PARALLEL -- rectangle against padded disc.
    auto misses = [&](const PencilPrim& pp) {
        if (apex) return dot3_fma(C.axis, xyz(pp.a)) < fmaf(C.ct, pp.b.x, -(C.st * pp.a.w)) - 4.0e-6f;
        return pp.a.x + pp.a.z < C.ulo || pp.a.x - pp.a.z > C.uhi || pp.a.y + pp.a.z < C.vlo || pp.a.y - pp.a.z > C.vhi;
    };
    // A light's pencil (P.a.w != 0: its rays run TOWARDS the apex): the cell of a ray only tells where the ray comes from, so "the bound is
    // not in the cell's cone" also drops every torus BEHIND the light -- the ray's own length limit in disguise (the ray would reach it beyond
    // its distance to the light), which no torus cull may use (torus_cull). A torus therefore stays a candidate of the cells its bound's
    // ANTIPODAL cone meets as well. A pencil of rays that START at the apex (the camera's): the antipodal cells too, where the apex is a far
    // origin for the torus (pp.b.z, round 6: torus_cull's "behind" rule -- the ray's backward extension goes through the bound).
    auto misses_behind = [&](const PencilPrim& pp) { return -dot3_fma(C.axis, xyz(pp.a)) < fmaf(C.ct, pp.b.x, -(C.st * pp.a.w)) - 4.0e-6f; };
    const bool towards_apex = apex && P.a.w != 0.0f;
    uint32_t bits = 0u;
    for (int b = 0; b < 32 && b < count; b++) {
        const PencilPrim pp = prims[(quadrics ? 0 : ns) + first + b];
        bool clear = false;
        if (pp.b.y == 0.0f) {
            if (quadrics) {
                // not on the degenerate branch for any direction of the cell (p2 varies by at most 2 |M| theta around the axis' value)
                const float p2 = apex ? fabsf(quadric_p2(S.surf_cull()[first + b], C.axis)) : pp.a.w;
                if (p2 > 2.01f * pp.b.z * C.theta + pp.b.w) clear = misses(pp);
            } else {
                clear = misses(pp) && ((!towards_apex && pp.b.z == 0.0f) || misses_behind(pp));
            }
        }
        if (!clear) bits |= 1u << b;
    }
    return bits;
}

// ------------------------------------------------------------------------------------------
// closest hit (rt.frag:587-628) and any-hit (rt.frag:630-658)
// ------------------------------------------------------------------------------------------
// Two-level scans. Level 1 walks compact 16-byte records four at a time (ONE batched scalar load
// per four primitives): for spheres the record is the test itself, for tori / rings / quadrics it
// is the conservative cull predicate. Level 2 -- the exact test on the full record -- runs only
// for primitives that at least one lane of the wave still needs (wave ballot), with the other
// lanes masked. Order and strict-< tie breaking of rt.frag:587-628 are kept: a cull evaluated
// with an earlier (larger) tmin only culls less.
// Many tori: "lane-divergent candidates" (measured on the 64-torus scene: 4.84 -> 3.40 ms; the same scheme for
// quadrics was slower -- 3.75 vs 3.54 ms: coherent rays share their few candidates, so nothing is saved and
// the records move from scalar to vector loads -- and is not used). Phase 1 runs the cheap cull
// of every primitive with wave-uniform indices (batched scalar loads) and records the survivors in a
// per-lane bit mask; phase 2 lets every lane walk ITS OWN candidates in index order, loading its own
// primitive record (vector loads), so one pass of the expensive solver serves up to 64 different
// primitives at once. The number of solver passes per scan drops from "distinct primitives any lane
// of the wave needs" to "most candidates of a single lane". Per lane the tests still run in index
// order with the live tmin, so the closest-hit semantics (strict <, first wins) are unchanged.
// Round 6: an exact test takes its primitive record BY VALUE. Through a reference the compiler loaded every field where it was first used --
// the quadric walk's exact test fetched its 96-byte record in seven scalar loads, each behind a branch and each waited for at once
// (profiles/r06_quadric_walk_isa.txt): seven scalar-cache round trips per test where one does. A by-value copy in front of the test is one
// batch of loads (fields the test never reads are dropped again by the compiler). Quadric-heavy 4K frame 920 -> 859 us (profiles/r06e_*).
#ifndef RT_SURF_RECORD_BY_VALUE
#define RT_SURF_RECORD_BY_VALUE 1   /* the candidate walks of the many-primitive variant */
#endif
#ifndef RT_RECORDS_BY_VALUE
#define RT_RECORDS_BY_VALUE 1       /* the short tables' uniform scans (surfaces, boxes, tori, rings) */
#endif
#if RT_RECORDS_BY_VALUE
#define RT_REC(T) T
#else
#define RT_REC(T) T&
#endif
#ifndef RT_SURF_PREFETCH
#define RT_SURF_PREFETCH 0   /* 1: the quadric mask walk loads the NEXT candidate's cull record while the current one is tested (A/B: profiles/r06c_*) */
#endif
#ifndef RT_LANE_DIVERGENT_MIN
#define RT_LANE_DIVERGENT_MIN 3   /* classes with fewer primitives keep the wave-uniform path */
#endif

#define RT_UNROLL4(BODY) { { constexpr int k = 0; BODY } { constexpr int k = 1; BODY } { constexpr int k = 2; BODY } { constexpr int k = 3; BODY } }

// GROUPS: compile the second-level group culls in. Only the many-primitive kernel variant (and the host build) does: in the default
// variant the extra code cost 2.5 % of the default scene's frame time through the instruction cache without ever being executed.
// The pencil of a scan, if it has one: wave-uniform `use`, the lane's cell (dword offset of its first mask word; words of a cell:
// quadrics first, then tori) and the next word's number. A word is loaded where it is walked and nothing of it is kept: fetching one
// word ahead, or keeping the lane's own word to let it skip the other lanes' candidates, each cost a register for the whole scan, and
// registers are what this kernel is short of (4K quadric-heavy frame 1293 us with both, 1260 without the fetch-ahead, 1247 without
// either; six waves per SIMD hide the load).
struct PencilScan {
    bool use;
    bool mem;            // words come from a pencil cell; otherwise from the caller's array (slab_ray_mask)
    uint32_t cell;
    int word;
    RT_HDM uint32_t next(const SceneView& S, const uint32_t* words)
    {
        const uint32_t own = mem ? S.pen[cell + word] : words[word];
        word++;
        return own;
    }
};

// ---- rays of no pencil: slab tables (rt_scene_dev.h DevSlabs) + the direction table ----
// words[0 .. stride): the lane's candidate mask. A lane the tables cannot vouch for (not a unit direction, far-away or non-finite origin)
// gets every primitive. tlimit: hits beyond it do not matter (the closest hit so far / the distance to the light) -- see below.
#ifndef RT_SLAB_BACK_SEGMENTS
#define RT_SLAB_BACK_SEGMENTS 4     /* pieces of the walk over the line's part BEHIND the origin (tori, "behind" rule); the forward part has RT_SLAB_SEGMENTS */
#endif
RT_HD int slab_index(float p, float lo, float inv) { return (int)gl_min(gl_max((p - lo) * inv, 0.0f), (float)(RT_SLABS - 1)); }
RT_HD void slab_ray_mask(const SceneView& S, f3 ro, f3 rd, float tlimit, uint32_t* words)
{
    const DevSlabs& B = *S.slabs();
    const int W = (int)S.h->pencil_stride;
    const uint32_t* T = S.at<uint32_t>(B.table_off);
    const bool ok = unit_direction(dot3_fma(rd, rd)) && dot3_fma(ro, ro) <= 1.0e8f;    // false for NaN
    const f3 o = ok ? ro : mk3(0.0f, 0.0f, 0.0f), d = ok ? rd : mk3(0.0f, 0.0f, 1.0f);
    // the quadrics this direction may put on their degenerate branch
    uint32_t deg[RT_SLAB_MAX_WORDS] = {0u, 0u, 0u, 0u};
    bool any_deg = false;
    if (S.h->pencil_dir != 0xffffffffu) {
        const uint32_t cell = pencil_cell_apex(S.pencils()[S.h->pencil_dir], S.h->pencil_stride, d, ok);
        for (int w = 0; w < RT_SLAB_MAX_WORDS; w++)
            if (w < W) { deg[w] = S.pen[cell + w]; any_deg = any_deg || deg[w] != 0u; }
    }
    // the part of the ray inside the box. The length limit only holds while the closest hit so far can only come closer -- and a quadric
    // on its degenerate branch accepts t > tmin (trap T4: the comparison is inverted), which moves the "closest" hit AWAY and makes
    // primitives behind the old limit eligible again (pencil-scene fuzz, seed 9038: a floor at t = 2992, a degenerate quadric at 22 925,
    // then a cylinder at 3018 that the reference therefore shows). A lane with such a quadric among its candidates gets the whole ray.
    // Tori: never the ray's own limit, only the reference's t < 100 (torus_cull) -- in a scene with tori the walk covers at least that.
    const float tB_quadric = any_deg ? 1.0e6f : gl_min(tlimit + 0.025f * gl_max(gl_min(tlimit, 100.0f) - 8.0f, 0.0f), 1.0e6f);   // (the slack is rounds 2-4's, kept)
    const float tB_forward = S.h->n_torus > 0 ? gl_max(tB_quadric, RT_TORUS_REACH * 1.001f + 0.01f) : tB_quadric;
    const float ov[3] = {o.x, o.y, o.z}, dv[3] = {d.x, d.y, d.z}, lov[3] = {B.lo.x, B.lo.y, B.lo.z}, hiv[3] = {B.hi.x, B.hi.y, B.hi.z};
    const float invv[3] = {B.inv.x, B.inv.y, B.inv.z};
    uint32_t acc[RT_SLAB_MAX_WORDS] = {0u, 0u, 0u, 0u};
    // Pass 1 (scenes with tori; round 6, the "behind" rule of torus_cull): the part of the LINE behind the origin, up to the backward reach --
    // a torus there is a candidate too (its own first-level test decides whether the origin is far enough for that to matter). Torus words only.
    const int nws = (S.h->n_surface + 31) >> 5;
#if defined(RT_AB_NO_SLAB_BACK)  /* measurement only (NOT exact): what the backward walk costs */
    const int passes = 1;
#else
    const int passes = RT_TORUS_BEHIND_RULE && S.h->n_torus > 0 ? 2 : 1;
#endif
#pragma unroll 1
    for (int pass = 0; pass < passes; pass++) {
        const float sg = pass ? -1.0f : 1.0f;
        float tA = 0.0f, tB = pass ? RT_TORUS_REACH_BACK * 1.001f + 0.01f : tB_forward;
        bool inside = true;
        for (int a = 0; a < 3; a++) {
            const float da = dv[a] * sg;
            if (fabsf(da) > 1.0e-9f) {
                const float inv = 1.0f / da, t0 = (lov[a] - ov[a]) * inv, t1 = (hiv[a] - ov[a]) * inv;
                tA = gl_max(tA, gl_min(t0, t1));
                tB = gl_min(tB, gl_max(t0, t1));
            } else if (ov[a] < lov[a] || ov[a] > hiv[a]) {
                inside = false;       // moves less than 1e-3 along this axis over any length that matters, and starts outside
            }
        }
        inside = inside && tA <= tB;
        if (!RT_ANY(inside)) continue;
        const int nseg = pass ? RT_SLAB_BACK_SEGMENTS : RT_SLAB_SEGMENTS;
        const float dt = (tB - tA) * (1.0f / (float)nseg);
        int i0[3];
        for (int a = 0; a < 3; a++) i0[a] = slab_index(fmaf(dv[a] * sg, tA, ov[a]), lov[a], invv[a]);
        for (int j = 1; j <= nseg; j++) {
            const float t = j == nseg ? tB : fmaf(dt, (float)j, tA);
            uint32_t seg[RT_SLAB_MAX_WORDS] = {~0u, ~0u, ~0u, ~0u};
            for (int a = 0; a < 3; a++) {
                const int i1 = slab_index(fmaf(dv[a] * sg, t, ov[a]), lov[a], invv[a]);
                const int lo = i0[a] < i1 ? i0[a] : i1, hi = i0[a] < i1 ? i1 : i0[a];
                const int lvl = 31 - __builtin_clz((unsigned)(hi - lo + 1));
                const uint32_t* e0 = T + (size_t)((a * RT_SLAB_LEVELS + lvl) * RT_SLABS + lo) * W;
                const uint32_t* e1 = T + (size_t)((a * RT_SLAB_LEVELS + lvl) * RT_SLABS + hi - (1 << lvl) + 1) * W;
                for (int w = 0; w < RT_SLAB_MAX_WORDS; w++)
                    if (w < W) seg[w] &= e0[w] | e1[w];
                i0[a] = i1;
            }
            for (int w = 0; w < RT_SLAB_MAX_WORDS; w++) acc[w] |= inside && (pass == 0 || w >= nws) ? seg[w] : 0u;
        }
    }
    for (int w = 0; w < RT_SLAB_MAX_WORDS; w++)
        if (w < W) words[w] = ok ? (acc[w] | deg[w] | B.always[w]) : B.valid[w];
}
RT_HD bool slabs_available(const SceneView& S) { return S.pen != nullptr && S.h->off_slabs != 0u; }

template <bool ENABLED>
RT_HD PencilScan pencil_open(const SceneView& S, int pencil, f3 ro, f3 rd, float len, bool from_apex)
{
    PencilScan ps;
    ps.use = false; ps.mem = true; ps.cell = 0u; ps.word = 0;
    if (!ENABLED || pencil < 0 || S.pen == nullptr || pencil >= (int)S.h->n_pencil) return ps;
    const DevPencil& P = S.pencils()[pencil];
    if (P.kind == RT_PENCIL_OFF) return ps;
    const uint32_t stride = S.h->pencil_stride;
    // from_apex: the ray starts AT the apex (camera rays: exactly, nothing to check about its origin); otherwise it runs towards it
    const bool ok = from_apex ? unit_direction(dot3_fma(rd, rd)) : pencil_ray_ok(ro, rd, P.kind == RT_PENCIL_APEX ? len : 0.0f);
    ps.cell = P.kind == RT_PENCIL_APEX ? pencil_cell_apex(P, stride, from_apex ? rd : -rd, ok) : pencil_cell_parallel(P, stride, ro, ok);
    ps.use = true;
    return ps;
}

template <bool CULL, bool COUNT, bool GROUPS = true>
RT_HD float calc_inter(const SceneView& S, f3 ro, f3 rd, int& num, int& type, LaneCounters& cnt, int pencil = -1)
{
    float tmin = RT_MAXDIST;
    float t = 0.0f;
    if (COUNT) cnt.closest++;
    RT_PH_DECL;
    PencilScan ps = pencil_open<GROUPS && CULL>(S, pencil, ro, rd, 0.0f, true);
    for (int i = 0; i < S.h->n_plane; i++) {
        if (intersect_plane(ro, rd, xyz(S.planes()[i].normal), xyz(S.planes()[i].pos), tmin, t)) { num = i; tmin = t; type = TYPE_PLANE; }
    }
    {
        const int n = S.h->n_sphere;
        const f4* geom = S.sph_geom();
        for (int i = 0; i < n; i += 4) {
            const f4 g[4] = {geom[i], geom[i + 1], geom[i + 2], geom[i + 3]};
            const uint32_t hb = S.sph_hollow()[i >> 5] >> (i & 31);
            RT_UNROLL4(if (i + k < n && intersect_sphere(ro, rd, g[k], ((hb >> k) & 1u) != 0, tmin, t)) { num = i + k; tmin = t; type = TYPE_SPHERE; })
        }
    }
    RT_PH_LAP(cnt, PH_C_SPH);
    uint32_t slabw[RT_SLAB_MAX_WORDS];
    if (GROUPS && CULL && !ps.use && slabs_available(S)) {   // a ray of no pencil: its candidates from the slab tables, up to the closest hit so far
        slab_ray_mask(S, ro, rd, tmin, slabw);
        ps.use = true;
        ps.mem = false;
    }
    if (ps.use) {
        const int nws = (S.h->n_surface + 31) >> 5;
        const DevSurfaceCull* cullrec = S.surf_cull();
        for (int w = 0; w < nws; w++) {
            const uint32_t own_w = ps.next(S, slabw);
            uint32_t u = wave_or(own_w, true);
            scan_stats_word(ps.mem ? 0 : 1, own_w, true, u);
#if RT_SURF_PREFETCH
            DevSurfaceCull nxt = cullrec[(w << 5) + (u ? __builtin_ctz(u) : 0)];
#endif
            while (u != 0u) {
                const int b = __builtin_ctz(u), i = (w << 5) + b;
                u &= u - 1u;
#if RT_SURF_PREFETCH
                const DevSurfaceCull c0 = nxt;
                nxt = cullrec[(w << 5) + (u ? __builtin_ctz(u) : b)];      // the next candidate's record travels while this one is tested
#else
                const DevSurfaceCull c0 = cullrec[i];
#endif
                // (round 4) behind the sphere: the clip box itself, for the lanes it governs -- quadric-heavy 4K frame 1 015 -> 981 us
                bool safe;
                bool need = !surface_cull(c0, ro, rd, tmin, safe);
#if RT_SURF_RECORD_BY_VALUE
                if (RT_ANY(need)) {
                    const DevSurface Qv = S.surfaces()[i];     // the whole 96-byte record in ONE scalar round trip (the walk's ISA had it in seven: profiles/r06_quadric_walk_isa.txt)
                    need = need && !(safe && surface_box_miss(Qv, ro, rd, tmin));
                    if (RT_ANY(need)) {
                        scan_stats_level2(ps.mem ? 0 : 1, need);
                        if (need && intersect_surface(Qv, ro, rd, tmin, t)) { num = i; tmin = t; type = TYPE_SURFACE; scan_stats_hit(ps.mem ? 0 : 1); }
                    }
                }
#else
                if (RT_ANY(need)) need = need && !(safe && surface_box_miss(S.surfaces()[i], ro, rd, tmin));
                if (RT_ANY(need)) {
                    scan_stats_level2(ps.mem ? 0 : 1, need);
                    if (need && intersect_surface(S.surfaces()[i], ro, rd, tmin, t)) { num = i; tmin = t; type = TYPE_SURFACE; scan_stats_hit(ps.mem ? 0 : 1); }
                }
#endif
            }
        }
    } else {
        const int n = S.h->n_surface;
        const DevSurfaceCull* cullrec = S.surf_cull();
        const bool grouped = GROUPS && CULL && n >= RT_GROUP_MIN;
        bool group_live = true;   // wave-uniform: some lane may reach the current group's sphere
        for (int i = 0; i < n; i += 2) {
            if (grouped && (i & (RT_GROUP - 1)) == 0) group_live = RT_ANY(!surface_group_cull(S.surf_group()[i / RT_GROUP], ro, rd));
            bool need[2] = {true, i + 1 < n};
            if (CULL) {
                const DevSurfaceCull c0 = cullrec[i], c1 = cullrec[i + 1];
                if (group_live) {
                    // the second of the pair is judged before the first has run: a lane that runs the first may see its tmin GROW (trap T4:
                    // the degenerate branch accepts t > tmin), so only a lane that skips the first may hold the second to today's limit
                    need[0] = !surface_cull(c0, ro, rd, tmin);
                    need[1] = need[1] && !surface_cull(c1, ro, rd, need[0] ? RT_FLT_MAX : tmin);
                } else {           // every lane misses the group: only the degenerate branch could still answer
                    need[0] = quadric_may_degenerate(c0, rd);
                    need[1] = need[1] && quadric_may_degenerate(c1, rd);
                }
            }
            for (int k = 0; k < 2; k++) {
                if (RT_ANY(need[k])) {
                    const RT_REC(DevSurface) Qv = S.surfaces()[i + k];
                    if (need[k] && intersect_surface(Qv, ro, rd, tmin, t)) { num = i + k; tmin = t; type = TYPE_SURFACE; }
                }
            }
        }
    }
    RT_PH_LAP(cnt, PH_C_SURF);
    {
        RayBoxCtx bctx;
        for (int i = 0; i < S.h->n_box; i++) {
            f3 nor;
            const RT_REC(DevBox) Bv = S.boxes()[i];
            if (intersect_box(Bv, ro, rd, tmin, t, nor, bctx)) { num = i; tmin = t; type = TYPE_BOX; }
        }
    }
    RT_PH_LAP(cnt, PH_C_BOX);
    if (CULL && S.h->n_torus >= RT_LANE_DIVERGENT_MIN) {
        const int n = S.h->n_torus;
        const f4* bound = S.torus_bound();
        for (int base = 0; base < n; base += 64) {
            unsigned long long cand = 0ull;
            const int end = base + 64 < n ? base + 64 : n;
            const bool grouped = GROUPS && n >= RT_GROUP_MIN;
            bool group_live = true;
            if (ps.use) {   // phase 1 over the wave's pencil candidates only
                for (int w = base >> 5; w << 5 < end; w++) {
                    uint32_t u = wave_or(ps.next(S, slabw), true);
                    while (u != 0u) {
                        const int b = __builtin_ctz(u), i = (w << 5) + b;
                        u &= u - 1u;
                        if (!torus_cull(bound[i], ro, rd)) cand |= 1ull << (i - base);
                    }
                }
            } else
            for (int i = base; i < end; i += 4) {
                if (grouped && (i & (RT_GROUP - 1)) == 0) group_live = RT_ANY(!torus_group_cull(S.torus_group()[i / RT_GROUP], ro, rd));
                if (!group_live) continue;   // wave-uniform: no lane can reach any of the group's tori
                const f4 b[4] = {bound[i], bound[i + 1], bound[i + 2], bound[i + 3]};
                RT_UNROLL4(if (i + k < end && !torus_cull(b[k], ro, rd)) cand |= 1ull << (i + k - base);)
            }
            dk_stats_scan(cand);
            if (GROUPS && RT_DK_RESTART) {      // (the many-primitive variant only: in the default variant this path never runs, and its code costs)
                bool th = false;
                torus_run_candidates<CULL, GROUPS, COUNT, false>(S, base, cand, ro, rd, tmin, num, th, cnt);
                if (th) type = TYPE_TORUS;
            } else
            while (RT_ANY(cand != 0ull)) {
                if (cand != 0ull) {
                    const int i = base + lane_pop(cand);          // differs from lane to lane
                    bool solved;
                    const bool th = intersect_torus_c<CULL, GROUPS>(S.tori()[i], ro, rd, tmin, t, solved);
                    if (COUNT && solved) cnt.torus_solves++;
                    if (th) { num = i; tmin = t; type = TYPE_TORUS; }
                }
            }
        }
    } else {
        const int n = S.h->n_torus;
        const f4* bound = S.torus_bound();
        for (int i = 0; i < n; i += 4) {
            bool need[4] = {true, i + 1 < n, i + 2 < n, i + 3 < n};
            if (CULL) {
                const f4 b[4] = {bound[i], bound[i + 1], bound[i + 2], bound[i + 3]};
                RT_UNROLL4(need[k] = need[k] && !torus_cull(b[k], ro, rd);)
            }
            for (int k = 0; k < 4; k++) {
                if (RT_ANY(need[k])) {
                    if (need[k]) {
                        bool solved;
                        RT_PH_BEGIN(_dk0);
                        const RT_REC(DevTorus) Tv = S.tori()[i + k];
                        const bool th = intersect_torus_c<CULL, GROUPS>(Tv, ro, rd, tmin, t, solved);
                        RT_PH_END(cnt, PH_DK, _dk0);
                        if (COUNT && solved) cnt.torus_solves++;
                        if (th) { num = i + k; tmin = t; type = TYPE_TORUS; }
                    }
                }
            }
        }
    }
    RT_PH_LAP(cnt, PH_C_TORUS);
    {
        const int n = S.h->n_ring;
        const f4* bound = S.ring_bound();
        for (int i = 0; i < n; i += 4) {
            bool need[4] = {true, i + 1 < n, i + 2 < n, i + 3 < n};
            if (CULL) {
                const f4 b[4] = {bound[i], bound[i + 1], bound[i + 2], bound[i + 3]};
                RT_UNROLL4(need[k] = need[k] && !ring_cull(b[k], ro, rd, tmin);)
            }
            for (int k = 0; k < 4; k++) {
                if (RT_ANY(need[k])) {
                    f2 uv;
                    const RT_REC(DevRing) Rv = S.rings()[i + k];
                    if (need[k] && intersect_ring(Rv, ro, rd, tmin, t, uv)) { num = i + k; tmin = t; type = TYPE_RING; }
                }
            }
        }
    }
    RT_PH_LAP(cnt, PH_C_RING);
    for (int i = 0; i < S.h->n_light_point; i++) {
        if (intersect_sphere(ro, rd, S.lights_point()[i].pos_r2, false, tmin, t)) { num = i; tmin = t; type = TYPE_POINT_LIGHT; }
    }
    RT_PH_LAP(cnt, PH_C_LIGHT);
    return tmin;
}

// `on` = this lane casts the ray. Lanes stop scanning once shadow >= 1: later hits could only
// set it to 1 or add a non-negative alpha and the result is min(shadow,1) (rt.frag:657), so the
// early exit is exact. The any-hit scan is an OR (a float sum for textured rings only), so the
// cheap classes go first; ring order is kept for the sum.
template <bool CULL, bool COUNT, bool GROUPS = true>
RT_HD float in_shadow(const SceneView& S, const TexTable& T, bool on, bool ref_on, f3 ro, f3 rd, float dist, LaneCounters& cnt, int pencil = -1)
{
    float shadow = 0.0f;
    float t = 0.0f;
    if (COUNT && on) cnt.shadow_cast++;
    PencilScan ps = pencil_open<GROUPS && CULL>(S, RT_ANY(on) ? pencil : -1, ro, rd, dist, false);
    if (RT_ANY(on)) {
        const int n = S.h->n_sphere;
        const f4* geom = S.sph_geom();
        for (int i = 0; i < n; i += 4) {
            const f4 g[4] = {geom[i], geom[i + 1], geom[i + 2], geom[i + 3]};
            RT_UNROLL4(if (on && i + k < n && intersect_sphere(ro, rd, g[k], false, dist, t)) { shadow = 1.0f; on = false; })
            if (!RT_ANY(on)) break;
        }
    }
    if (RT_ANY(on)) {
        RayBoxCtx bctx;
        for (int i = 0; i < S.h->n_box; i++) {
            f3 nor;
            const RT_REC(DevBox) Bv = S.boxes()[i];
            if (on && intersect_box(Bv, ro, rd, dist, t, nor, bctx)) { shadow = 1.0f; on = false; }
            if (!RT_ANY(on)) break;
        }
    }
    uint32_t slabw[RT_SLAB_MAX_WORDS];
    if (GROUPS && CULL && !ps.use && RT_ANY(on) && slabs_available(S)) {   // a light without a pencil
        slab_ray_mask(S, ro, rd, dist, slabw);
        ps.use = true;
        ps.mem = false;
    }
    if (ps.use) {
        // the word counter must advance past the quadric words even when every lane is already in shadow
        const int nws = (S.h->n_surface + 31) >> 5;
        const DevSurfaceCull* cullrec = S.surf_cull();
        for (int w = 0; w < nws; w++) {
            const uint32_t own_w = ps.next(S, slabw);
            uint32_t u = wave_or(own_w, on);
            scan_stats_word(ps.mem ? 2 : 3, own_w, on, u);
#if RT_SURF_PREFETCH
            DevSurfaceCull nxt = cullrec[(w << 5) + (u ? __builtin_ctz(u) : 0)];
#endif
            while (u != 0u) {
                const int b = __builtin_ctz(u), i = (w << 5) + b;
                u &= u - 1u;
#if RT_SURF_PREFETCH
                const DevSurfaceCull c0 = nxt;
                nxt = cullrec[(w << 5) + (u ? __builtin_ctz(u) : b)];
#else
                const DevSurfaceCull c0 = cullrec[i];
#endif
                bool safe;
                bool need = on && !surface_cull(c0, ro, rd, dist, safe);
#if RT_SURF_RECORD_BY_VALUE
                if (RT_ANY(need)) {
                    const DevSurface Qv = S.surfaces()[i];
                    need = need && !(safe && surface_box_miss(Qv, ro, rd, dist));
                    if (RT_ANY(need)) {
                        scan_stats_level2(ps.mem ? 2 : 3, need);
                        if (need && intersect_surface(Qv, ro, rd, dist, t)) { shadow = 1.0f; on = false; scan_stats_hit(ps.mem ? 2 : 3); }
                        if (!RT_ANY(on)) u = 0u;
                    }
                }
#else
                if (RT_ANY(need)) need = need && !(safe && surface_box_miss(S.surfaces()[i], ro, rd, dist));
                if (RT_ANY(need)) {
                    scan_stats_level2(ps.mem ? 2 : 3, need);
                    if (need && intersect_surface(S.surfaces()[i], ro, rd, dist, t)) { shadow = 1.0f; on = false; scan_stats_hit(ps.mem ? 2 : 3); }
                    if (!RT_ANY(on)) u = 0u;
                }
#endif
            }
        }
    } else if (RT_ANY(on)) {
        const int n = S.h->n_surface;
        const DevSurfaceCull* cullrec = S.surf_cull();
        const bool grouped = GROUPS && CULL && n >= RT_GROUP_MIN;
        bool group_live = true;
        for (int i = 0; i < n; i += 2) {
            if (grouped && (i & (RT_GROUP - 1)) == 0) group_live = RT_ANY(on && !surface_group_cull(S.surf_group()[i / RT_GROUP], ro, rd));
            bool need[2] = {on, on && i + 1 < n};
            if (CULL) {
                const DevSurfaceCull c0 = cullrec[i], c1 = cullrec[i + 1];
                if (group_live) {
                    need[0] = need[0] && !surface_cull(c0, ro, rd, dist);
                    need[1] = need[1] && !surface_cull(c1, ro, rd, dist);
                } else {
                    need[0] = need[0] && quadric_may_degenerate(c0, rd);
                    need[1] = need[1] && quadric_may_degenerate(c1, rd);
                }
            }
            for (int k = 0; k < 2; k++) {
                if (RT_ANY(need[k])) {
                    const RT_REC(DevSurface) Qv = S.surfaces()[i + k];
                    if (need[k] && on && intersect_surface(Qv, ro, rd, dist, t)) { shadow = 1.0f; on = false; }
                }
            }
            if (!RT_ANY(on)) break;
        }
    }
    if (CULL && S.h->n_torus >= RT_LANE_DIVERGENT_MIN) {
        if (RT_ANY(on)) {
            const int n = S.h->n_torus;
            const f4* bound = S.torus_bound();
            for (int base = 0; base < n; base += 64) {
                unsigned long long cand = 0ull;
                const int end = base + 64 < n ? base + 64 : n;
                const bool grouped = GROUPS && n >= RT_GROUP_MIN;
                bool group_live = true;
                if (ps.use) {
                    for (int w = base >> 5; w << 5 < end; w++) {
                        uint32_t u = wave_or(ps.next(S, slabw), on);
                        while (u != 0u) {
                            const int b = __builtin_ctz(u), i = (w << 5) + b;
                            u &= u - 1u;
                            if (on && !torus_cull(bound[i], ro, rd)) cand |= 1ull << (i - base);
                        }
                    }
                } else
                for (int i = base; i < end; i += 4) {
                    if (grouped && (i & (RT_GROUP - 1)) == 0) group_live = RT_ANY(on && !torus_group_cull(S.torus_group()[i / RT_GROUP], ro, rd));
                    if (!group_live) continue;
                    const f4 b[4] = {bound[i], bound[i + 1], bound[i + 2], bound[i + 3]};
                    RT_UNROLL4(if (on && i + k < end && !torus_cull(b[k], ro, rd)) cand |= 1ull << (i + k - base);)
                }
                dk_stats_scan(cand);
                if (GROUPS && RT_DK_RESTART) {
                    bool th = false;
                    float lim = dist;
                    int unused = 0;
                    torus_run_candidates<CULL, GROUPS, COUNT, true>(S, base, cand, ro, rd, lim, unused, th, cnt);
                    if (th) { shadow = 1.0f; on = false; }
                } else
                while (RT_ANY(cand != 0ull)) {
                    if (cand != 0ull) {
                        const int i = base + lane_pop(cand);
                        bool solved;
                        const bool th = intersect_torus_c<CULL, GROUPS>(S.tori()[i], ro, rd, dist, t, solved);
                        if (COUNT && solved) cnt.torus_solves++;
                        if (th) { shadow = 1.0f; on = false; cand = 0ull; }
                    }
                }
                if (!RT_ANY(on)) break;
            }
        }
    } else if (RT_ANY(on)) {
        const int n = S.h->n_torus;
        const f4* bound = S.torus_bound();
        for (int i = 0; i < n; i += 4) {
            bool need[4] = {on, on && i + 1 < n, on && i + 2 < n, on && i + 3 < n};
            if (CULL) {
                const f4 b[4] = {bound[i], bound[i + 1], bound[i + 2], bound[i + 3]};
                RT_UNROLL4(need[k] = need[k] && !torus_cull(b[k], ro, rd);)
            }
#ifdef RT_ABL_TORUS   /* timing ablation only (wrong frames): 1 = the second light's shadow rays skip the tori, 2 = every shadow ray does (profiles/r05q_*) */
            if ((RT_ABL_TORUS & 2) || ((RT_ABL_TORUS & 1) && pencil == 2)) { need[0] = need[1] = need[2] = need[3] = false; }
#endif
            for (int k = 0; k < 4; k++) {
                if (RT_ANY(need[k])) {
                    if (need[k] && on) {
                        bool solved;
                        RT_PH_BEGIN(_dk0);
                        const RT_REC(DevTorus) Tv = S.tori()[i + k];
                        const bool th = intersect_torus_c<CULL, GROUPS>(Tv, ro, rd, dist, t, solved);
                        RT_PH_END(cnt, PH_DK, _dk0);
                        if (COUNT && solved) cnt.torus_solves++;
                        if (th) { shadow = 1.0f; on = false; }
                    }
                }
            }
            if (!RT_ANY(on)) break;
        }
    }
    // rings: textured rings ADD their alpha (trap T10) -- ring order is kept for the float sum.
    // With quad-derivative LOD every lane for which the REFERENCE runs inShadow (ref_on: it neither
    // skips dp == 0 lights nor stops at shadow >= 1) must still present its ring uv at the fetch, because
    // its quad neighbours difference against it; such lanes fetch but do not accumulate.
    const bool lod = T.lod != 0;
    const bool ring_on = lod ? ref_on : on;
    if (RT_ANY(ring_on)) {
        const int n = S.h->n_ring;
        const f4* bound = S.ring_bound();
        for (int i = 0; i < n; i += 4) {
            const bool live = lod ? ref_on : on;
            bool need[4] = {live, live && i + 1 < n, live && i + 2 < n, live && i + 3 < n};
            if (CULL) {
                const f4 b[4] = {bound[i], bound[i + 1], bound[i + 2], bound[i + 3]};
                RT_UNROLL4(need[k] = need[k] && !ring_cull(b[k], ro, rd, dist);)
            }
            for (int k = 0; k < 4; k++) {
                if (RT_ANY(need[k])) {
                    f2 uv = mk2(0.0f, 0.0f);
                    const RT_REC(DevRing) Rv = S.rings()[i + k];
                    const bool geom_hit = need[k] && intersect_ring(Rv, ro, rd, dist, t, uv);
                    const bool hit = geom_hit && on;
                    const int texnum = __builtin_bit_cast(int, S.rings()[i + k].pos_tex.w);
                    if (texnum > 0) {
                        if (RT_ANY(geom_hit)) {
                            const f4 c = fetch2d(T, geom_hit, TEX_RING, (TYPE_RING << 20) | (i + k), uv.x, uv.y);
                            if (hit) shadow += c.w;
                        }
                    } else if (hit) {
                        shadow = 1.0f;
                    }
                    on = on && shadow < 1.0f;
                }
            }
            if (!lod && !RT_ANY(on)) break;
        }
    }
    return gl_min(shadow, 1.0f);
}

// ------------------------------------------------------------------------------------------
// shading (rt.frag:660-709)
// ------------------------------------------------------------------------------------------
struct Surf {           // what calcShade needs from a hit
    f3 color;           // material colour (after texturing)
    float diffuse;
    int specular;
    float kd, ks;
};

template <bool CULL, bool COUNT, bool GROUPS = true>
RT_HD f3 calc_shade(const SceneView& S, const TexTable& T, bool on, f3 pt, f3 rd, const Surf& m, f3 normal, LaneCounters& cnt)
{
    f3 diffuse = mk3(0.0f, 0.0f, 0.0f);
    f3 specular = mk3(0.0f, 0.0f, 0.0f);
    const f3 ambient = xyz(S.h->ambient), shadow_ambient = xyz(S.h->shadow_ambient);
    const int n_lp = S.h->n_light_point, n_ld = S.h->n_light_direct;
    for (int li = 0; li < n_lp + n_ld; li++) {
        f3 light_color, light_dir;
        float intensity, dist, distDiv;
        if (li < n_lp) {
            const DevLightPoint& L = S.lights_point()[li];
            light_color = xyz(L.color_intensity);
            intensity = L.color_intensity.w;
            light_dir = xyz(L.pos_r2) - pt;
            dist = length3(light_dir);
            distDiv = 1.0f + L.atten.x * dist + L.atten.y * dist * dist;
            light_dir = normalize3(light_dir);   // calcShade2
        } else {
            const DevLightDirect& L = S.lights_direct()[li - n_lp];
            light_color = xyz(L.color_intensity);
            intensity = L.color_intensity.w;
            light_dir = xyz(L.dir_n);            // normalize(-direction), hoisted to the packer
            dist = RT_MAXDIST;
            distDiv = 1.0f;
        }
        const float dp = gl_clamp(dot3(normal, light_dir), 0.0f, 1.0f);
        light_color = light_color * dp;
        if (COUNT && on) cnt.shadow_ref++;
        // dp == 0 zeroes light_color, so neither term below can receive anything from this light:
        // the shadow ray's result is unused and the ray is not cast (the reference casts it, trap T10).
        // (NaN dp must still take the full path so that it propagates like in the shader.)
        const bool cast = on && !(dp == 0.0f);
        RT_PH_BEGIN(_sh0);
        const float sh = 1.0f - in_shadow<CULL, COUNT, GROUPS>(S, T, cast, on, pt, light_dir, dist, cnt, 1 + li);   // pencil 0 is the camera's
        RT_PH_END(cnt, PH_SHADOW, _sh0);
        if (cast) {
            light_color = light_color * mk3(gl_max(sh, shadow_ambient.x), gl_max(sh, shadow_ambient.y), gl_max(sh, shadow_ambient.z));
            // directional lights have distDiv == 1 (rt.frag:703) and x / 1.0f == x exactly: the three IEEE
            // divisions are skipped for them (wave-uniform branch on the light index)
            const bool unit_div = li >= n_lp;
            const f3 dterm = ((light_color * m.color) * m.diffuse) * intensity;
            diffuse = diffuse + (unit_div ? dterm : dterm / distDiv);
            if (m.specular > 0) {
                const f3 refl = gl_reflect(light_dir, normal);
                const float specDp = gl_clamp(dot3(rd, refl), 0.0f, 1.0f);
                const f3 sterm = (light_color * rt_pow(specDp, (float)m.specular)) * intensity;
                specular = specular + (unit_div ? sterm : sterm / distDiv);
            }
        }
    }
    return ambient * m.color + (diffuse * m.kd + specular * m.ks);
}

// rt.frag:711-742
RT_HD float get_fresnel(f3 normal, f3 rd, float reflection)
{
    const float ndotv = gl_clamp(dot3(normal, -rd), 0.0f, 1.0f);
    return reflection + (1.0f - reflection) * rt_pow(1.0f - ndotv, 5.0f);
}
RT_HD float fresnel_reflect_amount(float n1, float n2, f3 normal, f3 incident, float refl)
{
    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float cosX = -dot3(normal, incident);
    if (n1 > n2) {
        const float n = n1 / n2;
        const float sinT2 = n * n * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return 1.0f;
        cosX = sqrtf(1.0f - sinT2);
    }
    const float x = 1.0f - cosX;
    float ret = r0 + (1.0f - r0) * x * x * x * x * x;
    ret = (refl + (1.0f - refl) * ret);
    return ret;
}

// ------------------------------------------------------------------------------------------
// hit attributes (rt.frag:744-784). Box normal and ring uv are RE-DERIVED from the winning
// primitive instead of being carried through the scan in registers (the shader's opt_normal /
// opt_uv globals): same operands, same operations -> same bits, fewer live VGPRs in the scan.
// ------------------------------------------------------------------------------------------
struct Hit {
    f3 normal;
    float alpha;
    float bias;
    Surf surf;
    float reflection, refraction;
    f3 absorb;
};

RT_HD void load_material(const DevMaterial& M, Hit& h)
{
    h.surf.color = mk3(M.color[0], M.color[1], M.color[2]);
    h.surf.diffuse = M.diffuse;
    h.surf.specular = M.specular;
    h.surf.kd = M.kd;
    h.surf.ks = M.ks;
    h.reflection = M.reflection;
    h.refraction = M.refraction;
    h.absorb = mk3(M.absorb[0], M.absorb[1], M.absorb[2]);
}

RT_HD void clear_hit(Hit& h)
{
    h.normal = mk3(0.0f, 0.0f, 0.0f);
    h.alpha = 1.0f;
    h.bias = 0.0f;
    h.surf.color = mk3(0.0f, 0.0f, 0.0f);
    h.surf.diffuse = 0.0f; h.surf.specular = 0; h.surf.kd = 0.0f; h.surf.ks = 0.0f;
    h.reflection = 0.0f; h.refraction = 0.0f; h.absorb = mk3(0.0f, 0.0f, 0.0f);
}
// `on` = lane has a hit to describe (h was cleared by the caller). Texture fetches sit in wave-uniform control flow.
RT_HD void get_hit_info(const SceneView& S, const TexTable& T, bool on, f3 ro, f3 rd, f3 pt, float t, int num, int type, Hit& h)
{
    // texture request of this lane: slot < 0 = none. Boxes need three taps (tri-planar).
    int slot = -1;
    float u = 0.0f, v = 0.0f;
    bool box_tex = false;
    f3 lp = mk3(0.0f, 0.0f, 0.0f), lpos = lp, ln = lp;

    if (on && type == TYPE_SPHERE) {  // + equirect texture, rt.frag:319-340
        const DevSphere& P = S.spheres()[num];
        load_material(S.mats(TYPE_SPHERE)[num], h);
        h.normal = normalize3(pt - xyz(P.geom));
        const int texnum = P.texture;
        if (texnum != 0) {
            f3 sn = h.normal;
            const f4 q = P.quat;
            if (q.x != 0.0f || q.y != 0.0f || q.z != 0.0f || q.w != 1.0f) sn = quat_rotate(q, sn);
            u = 0.5f + rt_atan2(sn.z, sn.x) / (2.0f * RT_PI_F);
            v = 0.5f - rt_asin(sn.y) / RT_PI_F;
            // texNum outside {1,2,3}: `color` stays undefined in GLSL (trap T15) -- pinned to 0 here
            if (texnum >= 1 && texnum <= 3) slot = TEX_SPHERE_1 + (texnum - 1);
            else { h.surf.color = mk3(0.0f, 0.0f, 0.0f); h.alpha = 0.0f; }
        }
    }
    if (on && type == TYPE_PLANE) {
        load_material(S.mats(TYPE_PLANE)[num], h);
        h.normal = normalize3(xyz(S.planes()[num].normal));
    }
    if (on && type == TYPE_SURFACE) {  // getSurfaceNormal rt.frag:573-584
        const DevSurface& Q = S.surfaces()[num];
        load_material(S.mats(TYPE_SURFACE)[num], h);
        const bool ident = ident_flag(Q.vmax.w);
        const f3 o = quat_rotate_id(Q.quat, ident, ro - xyz(Q.pos_a));
        const f3 d = quat_rotate_id(Q.quat, ident, rd);
        const f3 tm = d * t + o;
        const f3 n = mk3(2.0f * Q.pos_a.w * tm.x, 2.0f * Q.bcde.x * tm.y + Q.bcde.w, 2.0f * Q.bcde.y * tm.z + Q.bcde.z);
        h.normal = normalize3(quat_rotate_id(Q.qinv, ident, n));
    }
    if (on && type == TYPE_BOX) {  // + tri-planar texture, rt.frag:428-436
        const DevBox& B = S.boxes()[num];
        load_material(S.mats(TYPE_BOX)[num], h);
        float tt;
        f3 nor = mk3(0.0f, 0.0f, 0.0f);
        RayBoxCtx bctx;
        intersect_box(B, ro, rd, RT_FLT_MAX, tt, nor, bctx);  // re-derive the winning box's normal (+inf tmin: same result path)
        const bool ident = ident_flag(B.pos.w);   // then qinv = (-0,-0,-0,1) is one too
        h.normal = quat_rotate_id(B.qinv, ident, nor);
        if (__builtin_bit_cast(int, B.form_tex.w) != 0) {
            lpos = quat_rotate_id(B.quat, ident, xyz(B.pos));
            lp = quat_rotate_id(B.quat, ident, pt);
            ln = quat_rotate_id(B.quat, ident, h.normal);
            box_tex = true;
            slot = TEX_BOX;
            u = 0.5f * (lp.z - lpos.z) - 0.5f;
            v = 0.5f * (lp.y - lpos.y) - 0.5f;
        }
    }
    if (on && type == TYPE_TORUS) {  // getTorusNormal rt.frag:488-496
        const DevTorus& P = S.tori()[num];
        load_material(S.mats(TYPE_TORUS)[num], h);
        const bool ident = ident_flag(P.pos.w);
        const f3 o = quat_rotate_id(P.quat, ident, ro - xyz(P.pos));
        const f3 d = quat_rotate_id(P.quat, ident, rd);
        const f3 pos = o + d * t;
        const float s = dot3(pos, pos) - P.radii.w;
        const f3 n = pos * mk3(s - P.radii.z * 1.0f, s - P.radii.z * 1.0f, s - P.radii.z * -1.0f);
        h.normal = normalize3(quat_rotate_id(P.qinv, ident, n));
    }
    if (on && type == TYPE_RING) {  // + strip texture, rt.frag:391-397
        const DevRing& R = S.rings()[num];
        load_material(S.mats(TYPE_RING)[num], h);
        h.normal = xyz(R.normal);
        if (__builtin_bit_cast(int, R.pos_tex.w) != 0) {
            float tt;
            f2 uv = mk2(0.0f, 0.0f);
            intersect_ring(R, ro, rd, RT_FLT_MAX, tt, uv);  // re-derive opt_uv of the winning ring
            slot = TEX_RING;
            u = uv.x;
            v = uv.y;
        }
    }

    // ---- texture taps (wave-uniform sites) ----
    if (RT_ANY(slot >= 0)) {
        const int prim = (type << 20) | num;
        const f4 c0 = fetch2d(T, slot >= 0, slot, prim, u, v);
        if (slot >= 0 && !box_tex) {
            h.surf.color = mk3(c0.x, c0.y, c0.z);
            h.alpha = c0.w;
        }
        if (RT_ANY(box_tex)) {
            const f4 c1 = fetch2d(T, box_tex, TEX_BOX, prim, 0.5f * (lp.z - lpos.z) - 0.5f, 0.5f * (lp.x - lpos.x) - 0.5f);
            const f4 c2 = fetch2d(T, box_tex, TEX_BOX, prim, 0.5f * (lp.x - lpos.x) - 0.5f, 0.5f * (lp.y - lpos.y) - 0.5f);
            if (box_tex) {
                const float wx = fabsf(ln.x), wy = fabsf(ln.y), wz = fabsf(ln.z);
                h.surf.color = mk3(wx * c0.x + wy * c1.x + wz * c2.x, wx * c0.y + wy * c1.y + wz * c2.y, wx * c0.z + wy * c1.z + wz * c2.z);
            }
        }
    }
    const float distance = length3(pt - ro);
    h.bias = (9e-3f * distance + 35.0f) / 35e3f;
}

// rt.frag:313-317
RT_HD f3 ray_dir(const SceneView& S, float frag_x, float frag_y)
{
    const float cw = (float)S.h->canvas_w, ch = (float)S.h->canvas_h;
    const f3 v = mk3((frag_x - cw / 2.0f) / ch, (frag_y - ch / 2.0f) / ch, 1.0f);
    return normalize3(quat_rotate_id(S.h->cam_quat, S.h->cam_ident != 0, v));   // un-rotated camera: exact shortcut
}

// ------------------------------------------------------------------------------------------
// the pixel program: rt.frag main() (:804-902) + getReflectedColor (:787-802) as one segment
// loop. Per trip every live lane traces ONE closest-hit ray -- either the next segment of its
// main path or the one-bounce "side" ray that getReflectedColor casts for refractive surfaces
// -- then at most one shading evaluation. All lanes of a wave walk the loop together; `alive`
// predicates the work, so texture fetch sites stay in wave-uniform control flow.
// ------------------------------------------------------------------------------------------
// Path state that is only touched a few times per bounce-loop trip but would otherwise sit in ~19 VGPRs across the
// closest-hit scan AND the shadow scans of the shading site: the accumulated colour, the path mask, the next ray, the
// refracted continuation that waits while a mirror ray runs, the absorb distance. On the device these live in LDS (one
// dword column per lane, slot k at base[k * RT_PS_STRIDE]: lanes hit consecutive banks), which is what makes 5 and 6 waves
// per SIMD affordable: at 5 the kernel fits its 96 VGPRs without any scratch, where the compiler's own spills of the same
// values went through L2 to HBM (3x the frame in write traffic). Values round-trip bit for bit, the arithmetic and its
// order are untouched. The host build keeps them in a plain array.
// fence(): an ordinary LDS store to a slot whose index the compiler cannot see (a kernel argument): it may alias every
// slot, so loads after the shading site are not merged with loads before it (which would pin the values in registers
// across the shadow scans again) -- and, being LDS-only, it leaves the scene tables' scalar loads invariant, unlike a
// compiler-level memory clobber.
// The four loop scalars behind PS_SCALARS (mirror weight, shaded-term weight, mask factor, the two trip counters packed
// into one word) join them only in the WIDE layout: 23 + 1 slots = 24 KB per workgroup fit six workgroups per CU (6
// waves/SIMD) into the 160 KB of LDS but not seven, so the build for 7 waves/SIMD keeps them in registers (19 + 1 slots).
enum { PS_COLOR = 0, PS_MASK = 3, PS_RO = 6, PS_RD = 9, PS_CONT_RO = 12, PS_CONT_RD = 15, PS_ABSORB = 18, PS_SLOTS_NARROW = 19,
       PS_SCALARS = 19, QS_SIDE_R = 0, QS_W = 1, QS_K = 2, QS_COUNT = 3, PS_SLOTS_WIDE = 23 };
constexpr int path_slots(bool wide) { return wide ? PS_SLOTS_WIDE : PS_SLOTS_NARROW; }
#ifndef RT_PS_STRIDE
#define RT_PS_STRIDE 256
#endif
struct PathStore {
#if defined(__HIP_DEVICE_COMPILE__)
    float* base;      // this lane's column in LDS
    int fence_slot;   // run-time value (= path_slots(wide)): a pad slot
    RT_HDM float ld(int k) const { return base[k * RT_PS_STRIDE]; }
    RT_HDM void st(int k, float v) const { base[k * RT_PS_STRIDE] = v; }
    RT_HDM void fence() const { base[fence_slot * RT_PS_STRIDE] = 0.0f; }
#else
    mutable float v[PS_SLOTS_WIDE];
    RT_HDM float ld(int k) const { return v[k]; }
    RT_HDM void st(int k, float x) const { v[k] = x; }
    RT_HDM void fence() const {}
#endif
    RT_HDM int ldi(int k) const { return __builtin_bit_cast(int, ld(k)); }
    RT_HDM void sti(int k, int x) const { st(k, __builtin_bit_cast(float, x)); }
    RT_HDM f3 ld3(int k) const { return mk3(ld(k), ld(k + 1), ld(k + 2)); }
    RT_HDM void st3(int k, f3 x) const { st(k, x.x); st(k + 1, x.y); st(k + 2, x.z); }
};

template <bool IN_LDS>
struct PathScalars {   // the four loop scalars: LDS slots PS_SCALARS.. (WIDE layout) or registers
    float r[4];
    RT_HDM float ld(const PathStore& P, int k) const { return IN_LDS ? P.ld(PS_SCALARS + k) : r[k]; }
    RT_HDM void st(const PathStore& P, int k, float v) { if (IN_LDS) P.st(PS_SCALARS + k, v); else r[k] = v; }
    RT_HDM int ldi(const PathStore& P, int k) const { return __builtin_bit_cast(int, ld(P, k)); }
    RT_HDM void sti(const PathStore& P, int k, int v) { st(P, k, __builtin_bit_cast(float, v)); }
};

// SKYLOD: the sky box has a mip chain (load_cubemap(faces, true)) -- an instantiation of its own, so that the kernels of the reference's
// default (genMipmap = false) are the same machine code with and without the feature in the source
template <bool CULL, bool COUNT, bool WIDE = false, bool GROUPS = true, bool SKYLOD = false>
RT_HD f4 trace_pixel(const SceneView& S, const TexTable& T, const PathStore& P, bool alive, float frag_x, float frag_y, LaneCounters& cnt)
{
    PathScalars<WIDE> Q;
    P.st3(PS_MASK, mk3(1.0f, 1.0f, 1.0f));
    P.st3(PS_COLOR, mk3(0.0f, 0.0f, 0.0f));
    P.st3(PS_RO, xyz(S.h->cam_pos));
    P.st3(PS_RD, ray_dir(S, frag_x, frag_y));
    P.st(PS_ABSORB, 0.0f);
    Q.st(P, QS_SIDE_R, 0.0f);
    Q.st(P, QS_W, 0.0f);
    Q.st(P, QS_K, 1.0f);
    Q.sti(P, QS_COUNT, 0);   // low half: i, the shader's loop variable; high half: segments = main-loop trips taken (cap, trap T2)
    const int iterations = S.h->iterations;
    bool side = false;  // the NEXT trip traces getReflectedColor's ray; the refracted continuation of the main path waits
                        // in PS_CONT_RO / PS_CONT_RD while it runs
    int cam_pencil = 0; // the first trip traces the camera rays: pencil 0; every later ray starts somewhere else

    alive = alive && iterations > 0;
    RT_PH_DECL;
    RT_PH_LAP(cnt, PH_SETUP);
    while (RT_ANY(alive)) {
        RT_PH_ADD(cnt, PH_TRIPS, 1);
        const bool is_side = side;  // what THIS trip traces
        if (alive && !is_side) Q.sti(P, QS_COUNT, Q.ldi(P, QS_COUNT) + 0x10000);   // segments++
        const f3 ro = P.ld3(PS_RO), rd = P.ld3(PS_RD);

        // ---- one closest-hit ray per live lane ----
        int num = 0, type = -1;  // type is written only on a hit (rt.frag:593...); -1 = "nothing" (trap T3)
        float tm = RT_MAXDIST;
        if (alive) tm = calc_inter<CULL, COUNT, GROUPS>(S, ro, rd, num, type, cnt, cam_pencil);
        cam_pencil = -1;
        const bool hit = alive && (tm < RT_MAXDIST);  // false for NaN tm (trap T5)
        RT_PH_LAP(cnt, PH_SCAN);
        const f3 pt = ro + rd * tm;
        Hit h;
        clear_hit(h);
        // wave-uniform skip: a wave whose live lanes all missed (sky tiles, mirror rays leaving the scene)
        // has no hit to describe and, below, nothing to shade
        if (RT_ANY(hit)) get_hit_info(S, T, hit, ro, rd, pt, tm, num, type, h);
        RT_PH_LAP(cnt, PH_HITINFO);

        // ---- classify: everything that does not need the shaded colour happens BEFORE the shading call
        // (next ray, iteration count, unshaded radiance), so that only a weight and a mask factor stay
        // live across the shadow scans ----
        enum { ACT_NONE = 0, ACT_SIDE, ACT_REFLECT, ACT_DIFFUSE };
        int act = ACT_NONE;
        f3 sh_pt = pt;
        f3 n = h.normal;
        float w_s = 0.0f;     // scalar weight of the shaded term (R, T or alpha, by branch)
        float k_mask = 1.0f;  // mask factor applied AFTER the shaded term was weighted with the old mask
        bool finished = false;  // lane leaves the loop after this trip
        bool sky = false;

        if (alive) {
            if (is_side) {
                const float side_R = Q.ld(P, QS_SIDE_R);
                // getReflectedColor: light sphere -> its colour; miss -> BLACK (trap T3); else one shade
                if (type == TYPE_POINT_LIGHT) {
                    P.st3(PS_COLOR, P.ld3(PS_COLOR) + (xyz(S.lights_point()[num].color_intensity) * side_R) * P.ld3(PS_MASK));
                } else if (hit) {
                    act = ACT_SIDE;
                    sh_pt = dot3(rd, n) < 0.0f ? pt + n * h.bias : pt - n * h.bias;  // n stays unflipped (trap T18)
                    w_s = side_R;
                } else {
                    // miss: the shader still executes color += vec3(0) * reflectMultiplier * mask (rt.frag:855). That is
                    // +0 for a finite mask, but NaN once the mask has overflowed (exp(-absorb * negative distance) inside
                    // a box, traps T12 + T21) -- and then the pixel must come out NaN like the reference's, not inf.
                    P.st3(PS_COLOR, P.ld3(PS_COLOR) + (mk3(0.0f, 0.0f, 0.0f) * side_R) * P.ld3(PS_MASK));
                }
                k_mask = 1.0f - side_R;
                if (side_R >= 1.0f) finished = true;  // total reflection: checked after the mirror term (rt.frag:865)
                P.st3(PS_RO, P.ld3(PS_CONT_RO));
                P.st3(PS_RD, P.ld3(PS_CONT_RD));
                side = false;
            } else if (!hit) {
                sky = true;
                finished = true;
            } else if (type == TYPE_POINT_LIGHT) {
                P.st3(PS_COLOR, P.ld3(PS_COLOR) + xyz(S.lights_point()[num].color_intensity) * P.ld3(PS_MASK));
                finished = true;
            } else {
                const bool outside = dot3(rd, n) < 0.0f;
                n = outside ? n : -n;
                float R;
                if (h.refraction > 0.0f)
                    R = fresnel_reflect_amount(outside ? 1.0f : h.refraction, outside ? h.refraction : 1.0f, rd, n, h.reflection);
                else
                    R = get_fresnel(n, rd, h.reflection);
                if (h.refraction > 0.0f) {  // refractive, rt.frag:851-873
                    const f3 next_ro = pt - n * h.bias;
                    const f3 next_rd = gl_refract(rd, n, outside ? 1.0f / h.refraction : h.refraction);
                    if (outside && h.reflection > 0.0f) {
                        // next trip: the mirror ray; the refracted ray waits in the continuation slots
                        side = true;
                        Q.st(P, QS_SIDE_R, R);
                        P.st3(PS_CONT_RO, next_ro);
                        P.st3(PS_CONT_RD, next_rd);
                        P.st3(PS_RO, pt + n * h.bias);
                        P.st3(PS_RD, gl_reflect(rd, n));
                    } else {
                        if (!outside) {
                            const float absorbDistance = P.ld(PS_ABSORB) + tm;  // accumulates over all inside segments (trap T12)
                            P.st(PS_ABSORB, absorbDistance);
                            P.st3(PS_MASK, P.ld3(PS_MASK) * mk3(expf(-h.absorb.x * absorbDistance), expf(-h.absorb.y * absorbDistance),
                                                                 expf(-h.absorb.z * absorbDistance)));
                        }
                        if (R >= 1.0f) finished = true;
                        P.st3(PS_RO, next_ro);
                        P.st3(PS_RD, next_rd);
                    }
                    // i is not advanced: "i--" cancels the loop increment (rt.frag:870-872, trap T2)
                } else if (h.reflection > 0.0f) {  // reflective, rt.frag:874-880
                    act = ACT_REFLECT;
                    sh_pt = pt + n * h.bias;
                    w_s = 1.0f - R;
                    k_mask = R;
                    P.st3(PS_RO, sh_pt);
                    P.st3(PS_RD, gl_reflect(rd, n));
                    Q.sti(P, QS_COUNT, Q.ldi(P, QS_COUNT) + 1);   // i++
                } else {  // diffuse, rt.frag:881-890
                    act = ACT_DIFFUSE;
                    sh_pt = pt + n * h.bias;
                    w_s = h.alpha;
                    if (h.alpha < 1.0f) {  // alpha pass-through keeps rd, costs an iteration (trap T13)
                        P.st3(PS_RO, pt - n * h.bias);
                        k_mask = 1.0f - h.alpha;
                        Q.sti(P, QS_COUNT, Q.ldi(P, QS_COUNT) + 1);   // i++
                    } else {
                        finished = true;
                    }
                }
            }
        }

        RT_PH_LAP(cnt, PH_CLASSIFY);
        // ---- sky fetch (wave-uniform site) ----
        if (RT_ANY(sky)) {
            if (SKYLOD) {
                // mip-mapped sky box: the direction's quad differences (a neighbour counts only if it fetches the sky in this very trip,
                // i.e. at the same lock-step index); every lane of the quad takes part in the DPP exchange, whatever it is doing
                const int lane = rt_lane_id();
                const int kx = quad_other_x(sky ? 1 : 0), ky = quad_other_y(sky ? 1 : 0);
                const f3 ox = mk3(quad_other_x(rd.x), quad_other_x(rd.y), quad_other_x(rd.z));
                const f3 oy = mk3(quad_other_y(rd.x), quad_other_y(rd.y), quad_other_y(rd.z));
                if (sky) {
                    f3 ddx = mk3(0.0f, 0.0f, 0.0f), ddy = ddx;
                    if (kx) ddx = (lane & 1) != 0 ? rd - ox : ox - rd;   // right - left
                    if (ky) ddy = (lane & 2) != 0 ? rd - oy : oy - rd;   // top - bottom
                    const f4 c = sample_cube_lod(T.sky, rd, cube_lambda(T.sky, rd, ddx, ddy));
                    P.st3(PS_COLOR, P.ld3(PS_COLOR) + mk3(c.x, c.y, c.z) * P.ld3(PS_MASK));
                }
            } else if (sky) {
                const f4 c = sample_cube(T.sky, rd);
                P.st3(PS_COLOR, P.ld3(PS_COLOR) + mk3(c.x, c.y, c.z) * P.ld3(PS_MASK));
            }
        }

        RT_PH_LAP(cnt, PH_SKY);
        // ---- the single shading site ----
        f3 col = mk3(0.0f, 0.0f, 0.0f);
        if (RT_ANY(act != ACT_NONE)) {
            Q.st(P, QS_W, w_s);
            Q.st(P, QS_K, k_mask);
            P.fence();
            col = calc_shade<CULL, COUNT, GROUPS>(S, T, act != ACT_NONE, sh_pt, rd, h.surf, n, cnt);
        }
        RT_PH_LAP(cnt, PH_SHADE);

        // ---- apply + advance ----
        if (alive) {
            const f3 mask = P.ld3(PS_MASK);
            if (act != ACT_NONE) { w_s = Q.ld(P, QS_W); k_mask = Q.ld(P, QS_K); }
            const int counts = Q.ldi(P, QS_COUNT);
            const int i = counts & 0xffff, segments = counts >> 16;
            if (act == ACT_DIFFUSE) P.st3(PS_COLOR, P.ld3(PS_COLOR) + (col * mask) * w_s);          // calcShade * mask * alpha
            else if (act != ACT_NONE) P.st3(PS_COLOR, P.ld3(PS_COLOR) + (col * w_s) * mask);        // calcShade * R|T * mask
            P.st3(PS_MASK, mask * k_mask);  // x * 1.0f == x: lanes without a mask change are untouched
            // loop condition of the shader's for(), evaluated before the next MAIN trip
            if (finished || (!side && (i >= iterations || segments >= RT_SEGMENT_CAP))) alive = false;
        }
        RT_PH_LAP(cnt, PH_APPLY);
    }
    const f3 color = P.ld3(PS_COLOR);
    return mk4(color.x, color.y, color.z, 1.0f);
}

}  // namespace rtdev
