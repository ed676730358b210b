/* rt_oracle.c -- CPU ORACLE for the per-pixel ray-trace loop. TEST INFRASTRUCTURE ONLY.
 *
 * This file is a scalar, plain-C restatement of the reference's fragment shader
 * assets/shaders/rt.frag (all citations below are rt.frag line numbers unless another file is
 * named). It exists to CHECK the HIP tracer; nothing in the product path (raytracing_opengl_amd/,
 * include/, librtx_hip.so) includes, links or calls it. Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg use it.
 *
 * PARITY PINNING STATUS: PINNED against outputs of the reference itself. The reference ships no tests,
 * fixtures or golden images for this path, but its device half -- the GLSL fragment shader
 * -- runs in the build container on Mesa's llvmpipe software rasteriser, head-less (oracle/ref_gl/glref.c loads
 * swrast_dri.so through the DRI loader interface; the shader text is read from /root/reference at run time).
 * tools/gen_reference_frames.py executes it for nine cases (default scene with and without textures,
 * quadric-heavy, torus-heavy, four trap scenes) and commits the pixels as tests/golden/ref_frame_*.npz;
 * tests/test_reference_frames.py compares this oracle, the host build of the product's device code and the HIP
 * kernel with them. Agreement (fraction of pixels beyond 1e-4): 0.000 % on the two trap scenes without
 * implementation-defined ingredients (max difference 3e-5), 0.13 % / 0.17 % on the untextured default and the
 * quadric scene (silhouette pixels: llvmpipe evaluates normalize() as v*rsqrt(dot(v,v))), 4.8 % on the torus
 * scene (Durand-Kerner stops at 1e-3), 7-8 % on mip-mapped textures (level-of-detail selection is an
 * approximation the GL specification leaves to the implementation). DESIGN.md section 2 has the table.
 * Also pinned:
 *   - the scene bytes fed to this oracle are checked against golden uniform-block dumps made
 *     from the reference's own SceneManager.cpp/Surface.h/GLM (tests/golden/, tools/gen_golden_blocks.sh);
 *   - each intersector is checked against closed-form / float64 known answers (tests/test_oracle_kat.py);
 *   - ray-count pins of SURVEY.md Appendix C.3.
 *
 * Arithmetic contract (shared, by specification, with the HIP kernel -- DESIGN.md "Numerics"):
 *   - IEEE-754 binary32, round-to-nearest-even, no FMA contraction (-ffp-contract=off), no
 *     fast-math, denormals kept; + - * / sqrt are correctly rounded;
 *   - every GLSL expression is evaluated strictly left to right as written in rt.frag;
 *   - GLSL built-ins are restated by their specification formulas (GLSL 3.30 section 8):
 *     dot = x*x' + y*y' + z*z' (left to right), length = sqrt(dot), normalize = v / length(v),
 *     reflect = I - 2*dot(N,I)*N, refract per spec, min(a,b) = b<a?b:a, max(a,b) = a<b?b:a,
 *     clamp = min(max(x,lo),hi), step(e,x) = x<e?0:1, sign;
 *   - pow/exp use libm (they only shape colours; differences vs the device are ~1 ulp);
 *   - atan/asin (equirect uv, :323-324) and log2 (texture LOD) use the fixed float64 series below so
 *     that texel addresses, blend weights and the alpha they produce are bit-reproducible;
 *   - texture filtering, which GL leaves implementation-defined, follows DESIGN.md "Texture rule".
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off -fopenmp -shared).
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ---------------------------------------------------------------------------------------------
 * Public interface (bound from Python with ctypes: oracle/oracle.py)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t sphere_size, plane_size, surface_size, box_size, torus_size, ring_size;
    int32_t light_point_size, light_direct_size;
    int32_t iterations;
    float ambient_color[3];
    float shadow_ambient[3];
} orc_defines; /* == reference rt_defines, src/scene.h:7-20 */

typedef struct {
    int32_t width, height, channels; /* level 0; channels 1/3/4; width==0 -> unbound sampler */
    int32_t wrap;                    /* 0 = REPEAT, 1 = CLAMP_TO_EDGE */
    const uint8_t* texels;           /* row 0 = t 0 */
} orc_texture;

typedef struct {
    int32_t face_size, channels;
    const uint8_t* faces[6]; /* +X,-X,+Y,-Y,+Z,-Z; NULL face = black */
    int32_t gen_mipmap;      /* load_cubemap(faces, genMipmap): != 0 -> glGenerateMipmap + GL_LINEAR_MIPMAP_LINEAR (GLWrapper.cpp:307-310) */
} orc_cubemap;

enum { ORC_TEX_SPHERE_1 = 0, ORC_TEX_SPHERE_2, ORC_TEX_SPHERE_3, ORC_TEX_SPHERE_4, ORC_TEX_RING, ORC_TEX_BOX, ORC_TEX_COUNT };

typedef struct {
    int32_t fb_width, fb_height; /* framebuffer (gl_FragCoord range) */
    orc_defines defines;
    /* raw std140 block bytes exactly as SceneManager hands them to GLWrapper::init_buffer */
    const void* scene_buf;
    const void* spheres_buf;
    const void* planes_buf;
    const void* surfaces_buf;
    const void* boxes_buf;
    const void* toruses_buf;
    const void* rings_buf;
    const void* lights_point_buf;
    const void* lights_direct_buf;
    orc_cubemap skybox;
    orc_texture tex[ORC_TEX_COUNT];
    int32_t texture_lod; /* 0 = level-0 bilinear everywhere; 1 = mip chain + quad-derivative LOD (DESIGN.md "Texture rule");
                            2 = as 1, but the implicit level of detail is computed the way Mesa llvmpipe does it,
                            lambda = 0.5 * L(rho^2) with L(x) = exponent(x) + mantissa(x) - 1 (piecewise-linear log2).
                            DIAGNOSTIC: used only to show that the residual against the reference's shader run on
                            llvmpipe is that approximation (tests/test_reference_frames.py) */
} orc_frame;

typedef struct {
    uint64_t rays_closest;  /* calcInter invocations */
    uint64_t rays_shadow;   /* inShadow invocations */
    uint64_t tests[7];      /* per primitive type (TYPE_* index), closest + shadow scans */
    uint64_t dk_solves;     /* intersectTorus invocations */
    uint64_t dk_sweeps;     /* Durand-Kerner sweeps executed */
    uint64_t dk_capped;     /* solves that ran all 60 sweeps */
    uint64_t t4_taken;      /* degenerate-quadric branch returned true (trap T4) */
    uint64_t refract_segments;
    uint64_t tir_breaks;
    uint64_t alpha_pass;    /* trap T13 */
    uint64_t side_miss;     /* getReflectedColor black-on-miss (trap T3) */
    uint64_t light_hits;
    uint64_t box_nan_hits;  /* trap T5 */
    uint64_t box_inside_hits; /* trap T21 */
    uint64_t segment_cap_hits;
    uint64_t max_segments;
} orc_counters;

/* ---------------------------------------------------------------------------------------------
 * std140 records, rt.frag:24-113 (byte offsets: SURVEY.md Appendix B)
 * ------------------------------------------------------------------------------------------- */
typedef struct { float x, y; } vec2;
typedef struct { float x, y, z; } vec3;
typedef struct { float x, y, z, w; } vec4;

typedef struct {        /* :24-34, 64 B */
    vec3 color; float _p0;
    vec3 absorb;
    float diffuse;
    float reflection;
    float refraction;
    int32_t specular;
    float kd;
    float ks;
    float _p1[3];
} rt_material;
typedef struct { rt_material mat; vec4 obj; vec4 quat_rotation; int32_t textureNum; int32_t hollow; float _p[2]; } rt_sphere; /* :36-42 */
typedef struct { rt_material mat; vec3 pos; float _p0; vec3 normal; float _p1; } rt_plane;                                    /* :44-48 */
typedef struct { rt_material mat; vec4 quat_rotation; vec3 pos; float _p0; vec3 form; int32_t textureNum; } rt_box;            /* :50-56 */
typedef struct { rt_material mat; vec4 quat_rotation; vec3 pos; int32_t textureNum; float r1; float r2; float _p[2]; } rt_ring; /* :58-65 */
typedef struct {        /* :67-79, 160 B */
    rt_material mat; vec4 quat_rotation;
    vec3 v_min; float _p0; vec3 v_max; float _p1;
    vec3 pos; float a; float b; float c; float d; float e; float f; float _p2[3];
} rt_surface;
typedef struct { rt_material mat; vec4 quat_rotation; vec3 pos; float _p0; vec2 form; float _p1[2]; } rt_torus;                /* :81-86 */
typedef struct { vec3 direction; float _p0; vec3 color; float intensity; } rt_light_direct;                                  /* :88-93 */
typedef struct { vec4 pos; vec3 color; float intensity; float linear_k; float quadratic_k; float _p[2]; } rt_light_point;    /* :95-102 */
typedef struct {        /* :104-113 */
    vec4 quat_camera_rotation; vec3 camera_pos; float _p0; vec3 bg_color;
    int32_t canvas_width; int32_t canvas_height; int32_t reflect_depth; float _p1[2];
} rt_scene;

_Static_assert(sizeof(rt_material) == 64, "std140");
_Static_assert(sizeof(rt_sphere) == 112 && sizeof(rt_plane) == 96 && sizeof(rt_box) == 112, "std140");
_Static_assert(sizeof(rt_ring) == 112 && sizeof(rt_surface) == 160 && sizeof(rt_torus) == 112, "std140");
_Static_assert(sizeof(rt_light_direct) == 32 && sizeof(rt_light_point) == 48 && sizeof(rt_scene) == 64, "std140");

typedef struct { rt_material mat; vec3 normal; float bias_mult; float alpha; } hit_record; /* :115-120 */

#define FLT_MAX_GLSL 3.402823466e+38f /* :4 */
#define PI_F 3.14159265358979f        /* :5 */
#define TYPE_SPHERE 0
#define TYPE_PLANE 1
#define TYPE_SURFACE 2
#define TYPE_BOX 3
#define TYPE_TORUS 4
#define TYPE_RING 5
#define TYPE_POINT_LIGHT 6
static const float maxDist = 1000000.0f; /* :145 */

/* Hard cap on loop trips of main(): refraction does i-- (:870-872) so the GLSL loop has no
 * static bound. 256 is never reached on the bench scenes (counter segment_cap_hits). The HIP
 * kernel applies the same cap so that both terminate identically. */
#define ORC_SEGMENT_CAP 256

/* per-pixel "shader invocation" state: uniforms + the shader's two globals (:148-149) */
typedef struct {
    const orc_frame* fr;
    const rt_scene* scene;
    const rt_sphere* spheres;
    const rt_plane* planes;
    const rt_surface* surfaces;
    const rt_box* boxes;
    const rt_torus* toruses;
    const rt_ring* rings;
    const rt_light_point* lights_point;
    const rt_light_direct* lights_direct;
    int SPHERE_SIZE, PLANE_SIZE, SURFACE_SIZE, BOX_SIZE, TORUS_SIZE, RING_SIZE, LIGHT_POINT_SIZE, LIGHT_DIRECT_SIZE, ITERATIONS;
    vec3 AMBIENT_COLOR, SHADOW_AMBIENT;
    vec3 opt_normal; /* :148 */
    vec2 opt_uv;     /* :149 */
    float frag_x, frag_y; /* gl_FragCoord.xy */
    orc_counters* cnt;
    /* texture_lod == 1: this invocation runs as one of the 4 coroutines of a 2x2 pixel quad */
    int step;             /* closest-hit rays traced so far by this pixel = lock-step index (DESIGN.md texture rule) */
    int light_index;      /* which light calcShade is processing (part of the fetch key) */
    struct quad_s* quad;  /* NULL when texture_lod == 0 */
    int quad_slot;        /* 0..3: bit0 = x&1, bit1 = y&1 */
    uint32_t tag;         /* diagnostic: ORC_TAG_* events of this pixel (orc_set_tag_buffer) */
} inv_t;

/* Per-pixel event tags (diagnostic output for tests/reference_classify.py: WHICH implementation-defined mechanism a pixel touched). */
enum { ORC_TAG_BOX_INSIDE = 1,   /* intersectBox returned a negative distance (trap T21) */
       ORC_TAG_REFRACT = 2,      /* a refraction segment was taken (i--, trap T2) */
       ORC_TAG_TORUS = 4,        /* a Durand-Kerner root was accepted (closest-hit or shadow scan): the root is only good to the solver's 1e-3 stop */
       ORC_TAG_TEXTURE = 8,      /* a 2-D texture was sampled (mip level selection, atan/asin uv) */
       ORC_TAG_BOX_NAN = 16,     /* NaN through intersectBox (trap T5) */
       ORC_TAG_TIR = 32,
       ORC_TAG_SKY_LOD = 128,    /* the sky was fetched from a mip-mapped cube map (load_cubemap(faces, true)): level selection as for ORC_TAG_TEXTURE */
       ORC_TAG_QUAD_DIVERGENT = 64 };   /* a mip-mapped fetch for which a 2x2-quad neighbour did not execute the same fetch: GLSL leaves
                                          derivatives undefined in non-uniform control flow (GLSL 4.50 section 8.13.1); the oracle's rule
                                          takes that derivative as 0, llvmpipe differences whatever its masked-off lanes hold */
static uint32_t* g_tag_buffer = NULL;   /* fb_width * rows uint32, or NULL */
void orc_set_tag_buffer(uint32_t* p) { g_tag_buffer = p; }
/* Diagnostics for pinning Durand-Kerner's accepted root against the reference (tests/test_reference_frames.py): what the FIRST calcInter
 * of every pixel returned -- (t, type, num, 0) per pixel of the WHOLE frame, row 0 = bottom -- and the reverse: a per-pixel distance that
 * replaces the camera ray's own root when it hit a torus (the reference's root substituted), so that the rest of the path can be
 * compared without the solver's 1e-3 between the two. */
static float* g_primary_out = NULL;
static const float* g_primary_t_in = NULL;
void orc_set_primary_buffers(float* out_t_type_num, const float* torus_t_in) { g_primary_out = out_t_type_num; g_primary_t_in = torus_t_in; }

/* ---------------------------------------------------------------------------------------------
 * GLSL built-ins, restated
 * ------------------------------------------------------------------------------------------- */
static inline vec3 v3(float x, float y, float z) { vec3 r = {x, y, z}; return r; }
static inline vec2 v2(float x, float y) { vec2 r = {x, y}; return r; }
static inline vec4 v4(float x, float y, float z, float w) { vec4 r = {x, y, z, w}; return r; }
static inline vec3 add3(vec3 a, vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline vec3 sub3(vec3 a, vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline vec3 mul3(vec3 a, vec3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline vec3 scale3(vec3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
static inline vec3 div3s(vec3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
static inline vec3 neg3(vec3 a) { return v3(-a.x, -a.y, -a.z); }
static inline float dot3(vec3 a, vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline float dot2(vec2 a, vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float dot4(vec4 a, vec4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
static inline float length3(vec3 a) { return sqrtf(dot3(a, a)); }
static inline vec3 normalize3(vec3 a) { return div3s(a, length3(a)); }
/* min/max with a NaN operand are undefined in GLSL. Mode 0 (the contract shared with the kernel) follows the specification's
 * wording: min "returns y if y < x, otherwise x". Mode 1 is DIAGNOSTIC: the SSE minps/maxps behaviour of Mesa llvmpipe
 * (x < y ? x : y, i.e. the second operand when unordered), used only to show that the reference-on-llvmpipe frames of
 * degenerate scenes differ from the oracle in nothing but this choice (tools/fuzz_reference.py, orc_set_nan_minmax). */
/* Diagnostic: add a constant to every level of detail before the mip levels are chosen. Level selection is the GL implementation's
 * (its log2, its derivative estimates, its rounding: GL 4.5 section 8.14.1 allows a range); a reference pixel that lies between the
 * oracle's pixels for lambda - b and lambda + b is reproduced by a level of detail within b of the oracle's. Normally 0. */
static float g_lod_bias = 0.0f;
void orc_set_lod_bias(float b) { g_lod_bias = b; }
/* Diagnostic: every mip-mapped fetch takes THIS level of detail (>= 0; < 0 = off). A fetch in a divergent 2x2 quad has no defined
 * derivative (GLSL 4.50 section 8.13.1), so a GL implementation may sample it at ANY level; a trilinear sample is piecewise linear in
 * the level with knots at the integers, so the pixel's values for level 0, 1, ..., top bracket whatever level was taken
 * (tests/reference_classify.py: the `divergent` and `quad_neighbour` classes demand the reference's pixel inside that bracket). */
static float g_lod_force = -1.0f;
void orc_set_lod_force(float level) { g_lod_force = level; }
/* diagnostic: the fetches of ONE site -- sampler slot `slot` at site `site` (SITE_HIT 0 / SITE_SHADOW 1) -- at a level of their own, whatever
 * orc_set_lod_force says for the rest; level < 0: off; two such overrides (idx 0, 1). tests/reference_classify.py brackets pixels whose PATH depends on one fetch's value
 * (the ring's alpha decides the pass-through, rt.frag:884) with the levels of that fetch and of all others chosen independently. */
static int g_lod_force2_slot[2] = {-1, -1}, g_lod_force2_site[2] = {-1, -1};
static float g_lod_force2_level[2] = {-1.0f, -1.0f};
void orc_set_lod_force_site(int idx, int slot, int site, float level) { if (idx >= 0 && idx < 2) { g_lod_force2_slot[idx] = slot; g_lod_force2_site[idx] = site; g_lod_force2_level[idx] = level; } }
static int g_nan_minmax = 0;
void orc_set_nan_minmax(int mode) { g_nan_minmax = mode; }
static inline float gl_min(float a, float b) { return g_nan_minmax ? (a < b ? a : b) : (b < a ? b : a); }
static inline float gl_max(float a, float b) { return g_nan_minmax ? (a > b ? a : b) : (a < b ? b : a); }
static inline float gl_clamp(float x, float lo, float hi) { return gl_min(gl_max(x, lo), hi); }
static inline float gl_step(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static inline float gl_sign(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
static inline vec3 gl_reflect(vec3 I, vec3 N) { return sub3(I, scale3(N, 2.0f * dot3(N, I))); }
static inline vec3 gl_refract(vec3 I, vec3 N, float eta)
{
    float d = dot3(N, I);
    float k = 1.0f - eta * eta * (1.0f - d * d);
    if (k < 0.0f) return v3(0.0f, 0.0f, 0.0f);
    return sub3(scale3(I, eta), scale3(N, eta * d + sqrtf(k)));
}

/* atan(y,x) and asin(x) for the equirect mapping (:323-324): float64 series with a fixed
 * operation order (only + - * / sqrt, all IEEE, no contraction), rounded once to float. */
static double orc_atan_unit(double a) /* |a| <= 1 */
{
    double off = 0.0;
    if (a > 0.4142135623730950488) { a = (a - 1.0) / (a + 1.0); off = 0.78539816339744830962; }
    double s = a * a;
    /* Taylor series of atan, 13 terms, Horner from the highest power */
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
static float orc_atan2(float yf, float xf)
{
    double y = (double)yf, x = (double)xf;
    double ay = y < 0.0 ? -y : y, ax = x < 0.0 ? -x : x;
    double mx = ax < ay ? ay : ax, mn = ax < ay ? ax : ay;
    double r;
    if (!(mx > 0.0)) r = 0.0; /* atan(0,0) undefined in GLSL: pinned to 0 (also NaN inputs) */
    else {
        r = orc_atan_unit(mn / mx);
        if (ax < ay) r = 1.57079632679489661923 - r;
        if (x < 0.0) r = 3.14159265358979323846 - r;
        if (y < 0.0) r = -r;
    }
    return (float)r;
}
static float orc_asin(float xf)
{
    double x = (double)xf;
    double c = sqrt(1.0 - x * x); /* NaN for |x| > 1, as GLSL leaves it undefined (trap T15) */
    double ax = x < 0.0 ? -x : x;
    double mx = ax < c ? c : ax, mn = ax < c ? ax : c;
    double r;
    if (!(c == c)) return NAN;
    if (!(mx > 0.0)) r = 0.0;
    else {
        r = orc_atan_unit(mn / mx);
        if (c < ax) r = 1.57079632679489661923 - r;
        if (x < 0.0) r = -r;
    }
    return (float)r;
}

/* ---------------------------------------------------------------------------------------------
 * Texture rule (DESIGN.md "Texture rule") -- level-0 bilinear part
 * ------------------------------------------------------------------------------------------- */
static inline vec4 texel_rgba(const uint8_t* base, int channels, int w, int i, int j)
{
    const uint8_t* p = base + ((size_t)j * (size_t)w + (size_t)i) * (size_t)channels;
    if (channels == 4) return v4(p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f, p[3] / 255.0f);
    if (channels == 3) return v4(p[0] / 255.0f, p[1] / 255.0f, p[2] / 255.0f, 1.0f);
    return v4(p[0] / 255.0f, 0.0f, 0.0f, 1.0f); /* GL_RED */
}
static inline vec4 bilerp(vec4 t00, vec4 t10, vec4 t01, vec4 t11, float a, float b)
{
    float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    vec4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}
/* one axis: normalised coordinate -> two texel indices + weight */
static inline void axis_taps(float u, int n, int wrap, int* i0, int* i1, float* a)
{
    if (wrap == 0) u = u - floorf(u);               /* REPEAT: fract first (inf -> NaN) */
    else u = gl_min(gl_max(u, -1.0f), 2.0f);        /* CLAMP_TO_EDGE: anything outside [-1,2] hits the edge texel anyway */
    if (!(u == u)) u = 0.0f;                        /* NaN coordinates: pinned to 0 */
    float x = u * (float)n - 0.5f;
    float fl = floorf(x);
    *a = x - fl;
    int i = (int)fl;
    int k = i + 1;
    if (wrap == 0) {
        if (i < 0) i += n;
        if (i >= n) i -= n; /* u == 1.0 after fract of a tiny negative */
        if (k >= n) k -= n;
    } else {
        if (i < 0) i = 0;
        if (i > n - 1) i = n - 1;
        if (k < 0) k = 0;
        if (k > n - 1) k = n - 1;
    }
    *i0 = i;
    *i1 = k;
}
static vec4 sample2d_level0(const orc_texture* t, vec2 uv)
{
    if (t->width <= 0 || !t->texels) return v4(0.0f, 0.0f, 0.0f, 1.0f); /* unbound sampler: black (GL incomplete texture) */
    int i0, i1, j0, j1;
    float a, b;
    axis_taps(uv.x, t->width, t->wrap, &i0, &i1, &a);
    axis_taps(uv.y, t->height, t->wrap, &j0, &j1, &b);
    return bilerp(texel_rgba(t->texels, t->channels, t->width, i0, j0), texel_rgba(t->texels, t->channels, t->width, i1, j0),
                  texel_rgba(t->texels, t->channels, t->width, i0, j1), texel_rgba(t->texels, t->channels, t->width, i1, j1), a, b);
}
/* GL cube face selection table (OpenGL 3.3 spec table 3.19; ties x >= y >= z). */
static int cube_face(vec3 d)
{
    float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    if (ax >= ay && ax >= az) return d.x >= 0.0f ? 0 : 1;
    if (ay >= az) return d.y >= 0.0f ? 2 : 3;
    return d.z >= 0.0f ? 4 : 5;
}
/* (sc, tc, ma) of table 3.19 for a FIXED face as a linear map of the vector: applied to the direction it yields the face coordinates
 * and |major axis| (ma >= 0 on the direction's own face), applied to a derivative of the direction it yields their derivatives. */
static vec3 cube_project(int face, vec3 v)
{
    switch (face) {
    case 0: return v3(-v.z, -v.y, v.x);
    case 1: return v3(v.z, -v.y, -v.x);
    case 2: return v3(v.x, v.z, v.y);
    case 3: return v3(v.x, -v.z, -v.y);
    case 4: return v3(v.x, -v.y, v.z);
    default: return v3(-v.x, -v.y, -v.z);
    }
}
/* load_cubemap(faces, genMipmap = true) with a face that failed to load (GLWrapper.cpp:296-310): the texture is not cube complete, so
 * glGenerateMipmap raises GL_INVALID_OPERATION and builds nothing, and with GL_LINEAR_MIPMAP_LINEAR an incomplete texture samples
 * (0, 0, 0, 1) on EVERY face -- not only on the missing one. */
static int cube_incomplete(const orc_cubemap* c)
{
    if (!c->gen_mipmap) return 0;
    for (int f = 0; f < 6; f++) if (!c->faces[f]) return 1;
    return 0;
}
/* texture(skybox, dir) at level 0: per-face bilinear with CLAMP_TO_EDGE, not seamless (GLWrapper.cpp:310-314). */
static vec4 sample_cube(const orc_cubemap* c, vec3 d)
{
    const int face = cube_face(d);
    const vec3 p = cube_project(face, d);
    const float sc = p.x, tc = p.y, ma = p.z;
    if (c->face_size <= 0 || !c->faces[face] || cube_incomplete(c)) return v4(0.0f, 0.0f, 0.0f, 1.0f);
    float s = 0.5f * (sc / ma + 1.0f);
    float t = 0.5f * (tc / ma + 1.0f);
    orc_texture ft;
    ft.width = ft.height = c->face_size;
    ft.channels = c->channels;
    ft.wrap = 1;
    ft.texels = c->faces[face];
    return sample2d_level0(&ft, v2(s, t));
}

/* ---------------------------------------------------------------------------------------------
 * Texture rule, phase B: mip chain, trilinear, quad-derivative LOD (DESIGN.md "Texture rule";
 * SURVEY.md Appendix E). GL leaves all of this implementation-defined; the rule is:
 *  - mips: RGBA8, level L is max(1,w>>L) x max(1,h>>L); a texel is the rounded integer mean
 *    ((a+b+c+d+2)>>2) of the source texels (min(2i,ws-1), min(2i+1,ws-1)) x (same in j);
 *  - lambda <= 0 (or NaN): level-0 bilinear; else levels floor(lambda), +1 (clamped to the last),
 *    blended (1-f)*c0 + f*c1 with f = lambda - floor(lambda);
 *  - explicit LOD of getSphereTexture (:326-338): lambda = log2(max(df.x,df.y)*1024),
 *    df = |dFdx uv| + |dFdy uv|, df.x zeroed when > 0.5;
 *  - implicit LOD of texture() (ring :396,647; box :433-435): lambda = log2(max(|d(uv*size)/dx|, |d(uv*size)/dy|));
 *  - derivatives are differences inside the 2x2 pixel quad (aligned to even pixel coordinates):
 *    dFdx = right - left in the pixel's row, dFdy = top - bottom in its column; a neighbour counts only
 *    if, at the same lock-step index, it executes the same fetch (same site, light, ring, tap) of the
 *    same sampler on the same primitive; otherwise that axis' derivative is 0;
 *  - pixels of a quad that fall outside an odd-sized framebuffer run as helper invocations.
 * ------------------------------------------------------------------------------------------- */
#define ORC_MAX_MIPS 15
typedef struct {
    const uint8_t* key_texels; int key_w, key_h, key_c;
    uint64_t key_sum; /* FNV-1a of the level-0 bytes: the caller may reuse an address for other texels */
    int levels; int w[ORC_MAX_MIPS], h[ORC_MAX_MIPS];
    uint8_t* data[ORC_MAX_MIPS]; /* RGBA8 */
} orc_mipchain;
#define ORC_MIP_CACHE 48
static orc_mipchain g_mips[ORC_MIP_CACHE];
static int g_mips_n = 0;
/* Diagnostic callers that render the SAME frame description many times (tests/reference_classify.py: thousands of two-row renders) switch the
 * per-call re-hash of the level-0 texels off after the first render: the arrays are pinned by the caller, so no address can have been reused. */
static int g_mip_validate = 1;
void orc_set_mip_validation(int on) { g_mip_validate = on; }
static int g_mips_built = 0; /* chains built so far (orc_render repeats its serial pre-pass until a pass builds none) */

static uint64_t fnv1a(const uint8_t* p, size_t n)
{
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; i++) { h ^= p[i]; h *= 1099511628211ull; }
    return h;
}
/* validate != 0: re-hash the texels (done once per orc_render call, before the parallel region);
 * validate == 0: trust the address key (lookups from inside the render). */
static const orc_mipchain* mip_lookup(const orc_texture* t, int validate)
{
    if (t->width <= 0 || !t->texels) return NULL;
    const size_t nbytes = (size_t)t->width * t->height * t->channels;
    const uint64_t sum = validate ? fnv1a(t->texels, nbytes) : 0;
    for (int k = 0; k < g_mips_n; k++)
        if (g_mips[k].key_texels == t->texels && g_mips[k].key_w == t->width && g_mips[k].key_h == t->height && g_mips[k].key_c == t->channels) {
            if (!validate || g_mips[k].key_sum == sum) return &g_mips[k];
            for (int l = 0; l < g_mips[k].levels; l++) free(g_mips[k].data[l]); /* same address, other content: rebuild */
            memmove(&g_mips[k], &g_mips[k + 1], (size_t)(g_mips_n - k - 1) * sizeof g_mips[0]);
            g_mips_n--;
            break;
        }
    return NULL;
}
static const orc_mipchain* mip_get(const orc_texture* t)
{
    if (t->width <= 0 || !t->texels) return NULL;
    const orc_mipchain* hit = mip_lookup(t, 0);
    if (hit) return hit;
    if (g_mips_n == ORC_MIP_CACHE) { /* recycle the oldest */
        for (int l = 0; l < g_mips[0].levels; l++) free(g_mips[0].data[l]);
        memmove(&g_mips[0], &g_mips[1], (ORC_MIP_CACHE - 1) * sizeof g_mips[0]);
        g_mips_n = ORC_MIP_CACHE - 1;
    }
    g_mips_built++;
    orc_mipchain* m = &g_mips[g_mips_n++];
    memset(m, 0, sizeof *m);
    m->key_texels = t->texels; m->key_w = t->width; m->key_h = t->height; m->key_c = t->channels;
    m->key_sum = fnv1a(t->texels, (size_t)t->width * t->height * t->channels);
    int w = t->width, h = t->height;
    m->w[0] = w; m->h[0] = h;
    m->data[0] = (uint8_t*)malloc((size_t)w * h * 4);
    for (size_t i = 0; i < (size_t)w * h; i++) {
        const uint8_t* p = t->texels + i * t->channels;
        uint8_t* o = m->data[0] + i * 4;
        if (t->channels == 4) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = p[3]; }
        else if (t->channels == 3) { o[0] = p[0]; o[1] = p[1]; o[2] = p[2]; o[3] = 255; }
        else { o[0] = p[0]; o[1] = 0; o[2] = 0; o[3] = 255; }
    }
    int L = 0;
    while ((w > 1 || h > 1) && L + 1 < ORC_MAX_MIPS) {
        const int ws = w, hs = h;
        w = w > 1 ? w >> 1 : 1;
        h = h > 1 ? h >> 1 : 1;
        L++;
        m->w[L] = w; m->h[L] = h;
        m->data[L] = (uint8_t*)malloc((size_t)w * h * 4);
        const uint8_t* src = m->data[L - 1];
        for (int j = 0; j < h; j++)
            for (int i = 0; i < w; i++) {
                int i0 = 2 * i < ws - 1 ? 2 * i : ws - 1, i1 = 2 * i + 1 < ws - 1 ? 2 * i + 1 : ws - 1;
                int j0 = 2 * j < hs - 1 ? 2 * j : hs - 1, j1 = 2 * j + 1 < hs - 1 ? 2 * j + 1 : hs - 1;
                for (int c = 0; c < 4; c++) {
                    int sum = src[((size_t)j0 * ws + i0) * 4 + c] + src[((size_t)j0 * ws + i1) * 4 + c] + src[((size_t)j1 * ws + i0) * 4 + c] +
                              src[((size_t)j1 * ws + i1) * 4 + c];
                    m->data[L][((size_t)j * w + i) * 4 + c] = (uint8_t)((sum + 2) >> 2);
                }
            }
    }
    m->levels = L + 1;
    return m;
}

static vec4 sample_mip_level(const orc_mipchain* m, int level, int wrap, vec2 uv)
{
    orc_texture t;
    t.width = m->w[level]; t.height = m->h[level]; t.channels = 4; t.wrap = wrap; t.texels = m->data[level];
    return sample2d_level0(&t, uv);
}
static vec4 sample2d_lod(const orc_texture* t, vec2 uv, float lambda)
{
    const orc_mipchain* m = mip_get(t);
    if (!m) return v4(0.0f, 0.0f, 0.0f, 1.0f);
    if (!(lambda > 0.0f)) return sample_mip_level(m, 0, t->wrap, uv); /* magnification, -inf and NaN */
    const float top = (float)(m->levels - 1);
    if (lambda > top) lambda = top;
    const float fl = floorf(lambda);
    const int l0 = (int)fl;
    const float f = lambda - fl;
    const vec4 c0 = sample_mip_level(m, l0, t->wrap, uv);
    if (l0 + 1 > m->levels - 1) return c0;
    const vec4 c1 = sample_mip_level(m, l0 + 1, t->wrap, uv);
    return v4((1.0f - f) * c0.x + f * c1.x, (1.0f - f) * c0.y + f * c1.y, (1.0f - f) * c0.z + f * c1.z, (1.0f - f) * c0.w + f * c1.w);
}

/* log2 of the LOD rule: a fixed float64 sequence (exponent split, then ln(m) = 2 atanh((m-1)/(m+1)) as a
 * 10-term series, times 1/ln 2), rounded once to float -- so that the trilinear blend weight, and through a
 * blended alpha the `alpha < 1` pass-through decision (:884), are bit-reproducible across libm / device. */
static float orc_log2(float xf)
{
    if (!(xf > 0.0f)) return xf == 0.0f ? -INFINITY : NAN;
    if (xf > 3.0e38f) return INFINITY;
    double x = (double)xf;
    uint64_t bits;
    memcpy(&bits, &x, 8);
    int e = (int)((bits >> 52) & 0x7ffu) - 1023;
    uint64_t mb = (bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull;
    double m;
    memcpy(&m, &mb, 8);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    double z = (m - 1.0) / (m + 1.0);
    double z2 = z * z;
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
    double ln_m = 2.0 * z * p;
    return (float)((double)e + ln_m * 1.4426950408889634074);
}

/* Cube mips (load_cubemap(faces, true), GLWrapper.cpp:307-310 -> the sky fetch rt.frag:893 is min-filtered GL_LINEAR_MIPMAP_LINEAR).
 * Rule (DESIGN.md "Texture rule", cube part): every face has the mip chain of a 2-D RGBA8 image of its own (same rounded integer mean);
 * the level of detail comes from the derivatives of the FACE coordinates s = (sc/ma + 1)/2, t = (tc/ma + 1)/2 of the pixel's OWN face,
 * obtained from the quad differences of the direction by the quotient rule, d(sc/ma) = (dsc * ma - sc * dma) / ma^2 -- a neighbour that
 * looks at another face still contributes through its direction (this is also how Mesa llvmpipe does it);
 * lambda = log2(size * max(|(ds/dx, dt/dx)|, |(ds/dy, dt/dy)|)); levels floor(lambda), +1 blended like the 2-D ones. */
static float cube_lambda(const orc_cubemap* c, vec3 d, vec3 ddx, vec3 ddy, int mesa_log2)
{
    const int face = cube_face(d);
    const vec3 p = cube_project(face, d), px = cube_project(face, ddx), py = cube_project(face, ddy);
    const float ma = p.z, ma2 = ma * ma;
    const float dsdx = 0.5f * ((px.x * ma - p.x * px.z) / ma2), dtdx = 0.5f * ((px.y * ma - p.y * px.z) / ma2);
    const float dsdy = 0.5f * ((py.x * ma - p.x * py.z) / ma2), dtdy = 0.5f * ((py.y * ma - p.y * py.z) / ma2);
    const float n = (float)c->face_size;
    const float rx = sqrtf((dsdx * n) * (dsdx * n) + (dtdx * n) * (dtdx * n));
    const float ry = sqrtf((dsdy * n) * (dsdy * n) + (dtdy * n) * (dtdy * n));
    if (mesa_log2) { /* texture_lod == 2, diagnostic: llvmpipe's 0.5 * L(rho^2), see quad_resolve */
        const float r2 = gl_max(rx * rx, ry * ry);
        int e;
        const float m = frexpf(r2, &e);
        return r2 > 0.0f ? 0.5f * ((float)(e - 1) + (2.0f * m - 1.0f)) : -1000.0f;
    }
    return orc_log2(gl_max(rx, ry));
}
static vec4 sample_cube_lod(const orc_cubemap* c, vec3 d, float lambda)
{
    const int face = cube_face(d);
    const vec3 p = cube_project(face, d);
    if (c->face_size <= 0 || !c->faces[face] || cube_incomplete(c)) return v4(0.0f, 0.0f, 0.0f, 1.0f);
    const float s = 0.5f * (p.x / p.z + 1.0f);
    const float t = 0.5f * (p.y / p.z + 1.0f);
    orc_texture ft;
    ft.width = ft.height = c->face_size;
    ft.channels = c->channels;
    ft.wrap = 1;
    ft.texels = c->faces[face];
    return sample2d_lod(&ft, v2(s, t), lambda);
}

/* ---- 2x2 quad lock-step machinery ---- */
enum { SITE_HIT = 0, SITE_SHADOW = 1, SITE_SKY = 2 };
enum { LOD_EXPLICIT_SPHERE = 0, LOD_IMPLICIT = 1, LOD_CUBE = 2 };
typedef struct { int step, site, a, b, slot, ptype, pnum; } fetch_key;
typedef struct {
    int active;      /* pixel exists in this quad (inside the even-rounded framebuffer) */
    int done;
    int waiting;
    fetch_key key;
    vec2 uv;
    vec3 dir;        /* LOD_CUBE: the fetch direction (uv unused) */
    int mode;
    vec4 result;
    ucontext_t ctx;
    inv_t iv;
    vec4 color;
    char* stack;
} quad_lane;
typedef struct quad_s {
    quad_lane lane[4];
    ucontext_t sched;
    int current;
} quad_t;

static int key_cmp(const fetch_key* p, const fetch_key* q)
{
    const int* a = (const int*)p; const int* b = (const int*)q;
    for (int k = 0; k < 7; k++) { if (a[k] < b[k]) return -1; if (a[k] > b[k]) return 1; }
    return 0;
}

/* texture fetch as seen by the pixel program */
static vec4 tex_fetch(inv_t* iv, int slot, int site, int a, int b, int ptype, int pnum, vec2 uv, int mode)
{
    iv->tag |= ORC_TAG_TEXTURE;
    const orc_texture* t = &iv->fr->tex[slot];
    if (!iv->quad) return sample2d_level0(t, uv); /* texture_lod == 0 */
    quad_t* q = iv->quad;
    quad_lane* me = &q->lane[iv->quad_slot];
    me->key.step = iv->step; me->key.site = site; me->key.a = a; me->key.b = b; me->key.slot = slot; me->key.ptype = ptype; me->key.pnum = pnum;
    me->uv = uv;
    me->mode = mode;
    me->waiting = 1;
    swapcontext(&me->ctx, &q->sched); /* yield; the scheduler fills me->result */
    return me->result;
}

/* texture(skybox, rd), rt.frag:893 */
static vec4 sky_fetch(inv_t* iv, vec3 rd)
{
    const orc_cubemap* c = &iv->fr->skybox;
    if (!iv->quad || !c->gen_mipmap) return sample_cube(c, rd); /* no cube mips (the reference's default) or texture_lod == 0 */
    iv->tag |= ORC_TAG_SKY_LOD | ORC_TAG_TEXTURE;
    quad_t* q = iv->quad;
    quad_lane* me = &q->lane[iv->quad_slot];
    me->key.step = iv->step; me->key.site = SITE_SKY; me->key.a = 0; me->key.b = 0; me->key.slot = ORC_TEX_COUNT; me->key.ptype = 0; me->key.pnum = 0;
    me->dir = rd;
    me->mode = LOD_CUBE;
    me->waiting = 1;
    swapcontext(&me->ctx, &q->sched);
    return me->result;
}

static void quad_resolve(quad_t* q, const orc_frame* fr)
{
    /* smallest pending key */
    int first = -1;
    for (int k = 0; k < 4; k++)
        if (q->lane[k].active && q->lane[k].waiting && (first < 0 || key_cmp(&q->lane[k].key, &q->lane[first].key) < 0)) first = k;
    const fetch_key key = q->lane[first].key;
    int in_set[4];
    for (int k = 0; k < 4; k++) in_set[k] = q->lane[k].active && q->lane[k].waiting && key_cmp(&q->lane[k].key, &key) == 0;
    vec4 res[4];
    for (int k = 0; k < 4; k++) {
        if (!in_set[k]) continue;
        const int kx = k ^ 1, ky = k ^ 2; /* bit0 = x&1, bit1 = y&1 */
        vec2 ddx = v2(0.0f, 0.0f), ddy = v2(0.0f, 0.0f);
        if (in_set[kx]) { const quad_lane* r = &q->lane[k | 1]; const quad_lane* l = &q->lane[k & ~1]; ddx = v2(r->uv.x - l->uv.x, r->uv.y - l->uv.y); }
        if (in_set[ky]) { const quad_lane* tp = &q->lane[k | 2]; const quad_lane* bt = &q->lane[k & ~2]; ddy = v2(tp->uv.x - bt->uv.x, tp->uv.y - bt->uv.y); }
        if (!in_set[kx] || !in_set[ky]) q->lane[k].iv.tag |= ORC_TAG_QUAD_DIVERGENT;
        if (q->lane[k].mode == LOD_CUBE) {
            vec3 dx = v3(0.0f, 0.0f, 0.0f), dy = dx;
            if (in_set[kx]) dx = sub3(q->lane[k | 1].dir, q->lane[k & ~1].dir);
            if (in_set[ky]) dy = sub3(q->lane[k | 2].dir, q->lane[k & ~2].dir);
            float lambda = cube_lambda(&fr->skybox, q->lane[k].dir, dx, dy, fr->texture_lod == 2);
            if (g_lod_bias != 0.0f) lambda += g_lod_bias;
            if (g_lod_force >= 0.0f) lambda = g_lod_force;
            for (int f = 0; f < 2; f++)
                if (g_lod_force2_level[f] >= 0.0f && key.slot == g_lod_force2_slot[f] && key.site == g_lod_force2_site[f]) lambda = g_lod_force2_level[f];
            res[k] = sample_cube_lod(&fr->skybox, q->lane[k].dir, lambda);
            continue;
        }
        const orc_texture* t = &fr->tex[key.slot];
        float lambda;
        if (q->lane[k].mode == LOD_EXPLICIT_SPHERE) {
            vec2 df = v2(fabsf(ddx.x) + fabsf(ddy.x), fabsf(ddx.y) + fabsf(ddy.y)); /* fwidth(uv), :326 */
            if (df.x > 0.5f) df.x = 0.0f;                                          /* :327 */
            lambda = orc_log2(gl_max(df.x, df.y) * 1024.0f);                       /* :331 */
        } else {
            const float w = (float)t->width, h = (float)t->height;
            const float rx = sqrtf((ddx.x * w) * (ddx.x * w) + (ddx.y * h) * (ddx.y * h));
            const float ry = sqrtf((ddy.x * w) * (ddy.x * w) + (ddy.y * h) * (ddy.y * h));
            lambda = orc_log2(gl_max(rx, ry));
            if (fr->texture_lod == 2) {
                const float r2 = gl_max(rx * rx, ry * ry);
                int e;
                const float m = frexpf(r2, &e); /* r2 = m * 2^e, m in [0.5,1) */
                lambda = r2 > 0.0f ? 0.5f * ((float)(e - 1) + (2.0f * m - 1.0f)) : -1000.0f;
            }
        }
        if (g_lod_bias != 0.0f) lambda += g_lod_bias;   /* diagnostic, see orc_set_lod_bias */
        if (g_lod_force >= 0.0f) lambda = g_lod_force;  /* diagnostic, see orc_set_lod_force */
        for (int f = 0; f < 2; f++)
            if (g_lod_force2_level[f] >= 0.0f && key.slot == g_lod_force2_slot[f] && key.site == g_lod_force2_site[f]) lambda = g_lod_force2_level[f];
        res[k] = sample2d_lod(t, q->lane[k].uv, lambda);
    }
    for (int k = 0; k < 4; k++)
        if (in_set[k]) { q->lane[k].result = res[k]; q->lane[k].waiting = 0; }
}

/* ---------------------------------------------------------------------------------------------
 * rt.frag functions, in file order
 * ------------------------------------------------------------------------------------------- */
static inline void swapf(float* a, float* b) { float tmp = *a; *a = *b; *b = tmp; } /* :273-278 */

static inline int isBetween(vec3 value, vec3 mn, vec3 mx) /* :280-283 */
{
    return (value.x > mn.x && value.y > mn.y && value.z > mn.z) && (value.x < mx.x && value.y < mx.y && value.z < mx.z);
}
static inline vec4 quat_conj(vec4 q) { return v4(-q.x, -q.y, -q.z, q.w); } /* :285-288 */
static inline vec4 quat_inv(vec4 q)                                         /* :290-293 */
{
    vec4 c = quat_conj(q);
    float s = 1.0f / dot4(q, q);
    return v4(c.x * s, c.y * s, c.z * s, c.w * s);
}
static inline vec4 quat_mult(vec4 q1, vec4 q2) /* :295-303 */
{
    vec4 qr;
    qr.x = (q1.w * q2.x) + (q1.x * q2.w) + (q1.y * q2.z) - (q1.z * q2.y);
    qr.y = (q1.w * q2.y) - (q1.x * q2.z) + (q1.y * q2.w) + (q1.z * q2.x);
    qr.z = (q1.w * q2.z) + (q1.x * q2.y) - (q1.y * q2.x) + (q1.z * q2.w);
    qr.w = (q1.w * q2.w) - (q1.x * q2.x) - (q1.y * q2.y) - (q1.z * q2.z);
    return qr;
}
static inline vec3 rotate(vec4 qr, vec3 v) /* :305-311 */
{
    vec4 qr_conj = quat_conj(qr);
    vec4 q_pos = v4(v.x, v.y, v.z, 0.0f);
    vec4 q_tmp = quat_mult(qr, q_pos);
    vec4 r = quat_mult(q_tmp, qr_conj);
    return v3(r.x, r.y, r.z);
}

/* Diagnostic for the pin against a real GL implementation (tests/reference_classify.py): displace every primary ray by (jx, jy) in
 * the units of the un-normalised view vector (1 / canvas_height = one pixel). A GL implementation is free to evaluate normalize(),
 * dot() and the intersectors with other roundings, fused or reordered; at a pixel where a hit/miss, root-selection or branch decision
 * is within rounding of flipping, the reference's own output is implementation-defined. Such pixels are found by asking whether a
 * displacement of a few ulp changes the oracle's own answer. Normally (0, 0). */
static float g_ray_jitter[2] = {0.0f, 0.0f};
void orc_set_ray_jitter(float jx, float jy) { g_ray_jitter[0] = jx; g_ray_jitter[1] = jy; }

static vec3 getRayDir(const inv_t* iv) /* :313-317 */
{
    float cw = (float)iv->scene->canvas_width, ch = (float)iv->scene->canvas_height;
    vec3 result = v3((iv->frag_x - cw / 2.0f) / ch, (iv->frag_y - ch / 2.0f) / ch, 1.0f);
    if (g_ray_jitter[0] != 0.0f) result.x += g_ray_jitter[0];
    if (g_ray_jitter[1] != 0.0f) result.y += g_ray_jitter[1];
    return normalize3(rotate(iv->scene->quat_camera_rotation, result));
}

static vec4 getSphereTexture(inv_t* iv, vec3 sphereNormal, vec4 quat, int texNum, int sphereNum) /* :319-340 */
{
    if (quat.x != 0.0f || quat.y != 0.0f || quat.z != 0.0f || quat.w != 1.0f) sphereNormal = rotate(quat, sphereNormal);
    float u = 0.5f + orc_atan2(sphereNormal.z, sphereNormal.x) / (2.0f * PI_F);
    float v = 0.5f - orc_asin(sphereNormal.y) / PI_F;
    vec2 uv = v2(u, v);
    /* df = fwidth(uv), the seam fix and textureLod(..., log2(max(df.x,df.y)*1024)) (:326-338) happen in
     * tex_fetch/quad_resolve, which see the whole 2x2 quad; with texture_lod == 0 the fetch is level-0 bilinear */
    vec4 color = v4(0.0f, 0.0f, 0.0f, 0.0f); /* texNum not in {1,2,3}: undefined in GLSL (T15); pinned to 0 */
    if (texNum >= 1 && texNum <= 3) color = tex_fetch(iv, ORC_TEX_SPHERE_1 + (texNum - 1), SITE_HIT, 0, 0, TYPE_SPHERE, sphereNum, uv, LOD_EXPLICIT_SPHERE);
    return color;
}

static int intersectSphere(vec3 ro, vec3 rd, vec4 object, int hollow, float tmin, float* t) /* :342-354 */
{
    vec3 oc = sub3(ro, v3(object.x, object.y, object.z));
    float b = dot3(oc, rd);
    float c = dot3(oc, oc) - object.w * object.w;
    float h = b * b - c;
    if (h < 0.0f) return 0;
    float h_sqrt = sqrtf(h);
    *t = -b - h_sqrt;
    if (hollow && *t < 0.0f) *t = -b + h_sqrt;
    return *t > 0.0f && *t < tmin;
}

static int intersectPlane(vec3 ro, vec3 rd, vec3 n, vec3 p, float tmin, float* t) /* :356-370, PLANE_ONESIDE defined */
{
    float denom = gl_clamp(dot3(n, rd), -1.0f, 1.0f);
    if (denom < -1e-6f) {
        vec3 p_ro = sub3(p, ro);
        *t = dot3(p_ro, n) / denom;
        return (*t > 0.0f) && (*t < tmin);
    }
    return 0;
}

static int intersectRing(inv_t* iv, vec3 ro, vec3 rd, int num, float tmin, float* t) /* :372-390 */
{
    const rt_ring* ring = &iv->rings[num];
    rd = rotate(ring->quat_rotation, rd);
    ro = rotate(ring->quat_rotation, sub3(ro, ring->pos));
    *t = -ro.z / rd.z;
    float x = ro.x + rd.x * *t;
    float y = ro.y + rd.y * *t;
    float p = x * x + y * y;
    if (*t > 0.0f && *t < tmin && p < ring->r2 && p > ring->r1) {
        float len = sqrtf(dot2(v2(x, y), v2(x, y)));
        vec2 nrm = v2(x / len, y / len);
        float cosv = dot2(nrm, v2(1.0f, 0.0f));
        iv->opt_uv = v2((p - ring->r1) / (ring->r2 - ring->r1), cosv);
        return 1;
    }
    return 0;
}
static vec3 getRingNormal(const inv_t* iv, int num) /* :391-394 */
{
    return rotate(quat_inv(iv->rings[num].quat_rotation), v3(0.0f, 0.0f, -1.0f));
}
static vec4 getRingTexture(inv_t* iv, int site, int light, int ringNum, vec2 uv) /* :395-397 */
{
    return tex_fetch(iv, ORC_TEX_RING, site, light, ringNum, TYPE_RING, ringNum, uv, LOD_IMPLICIT);
}

static int intersectBox(inv_t* iv, vec3 ro, vec3 rd, int num, float tmin, float* t) /* :399-427 */
{
    const rt_box* box = &iv->boxes[num];
    vec3 rdd = rotate(box->quat_rotation, rd);
    vec3 roo = rotate(box->quat_rotation, sub3(ro, box->pos));
    vec3 m = v3(1.0f / rdd.x, 1.0f / rdd.y, 1.0f / rdd.z);
    vec3 n = mul3(m, roo);
    vec3 k = mul3(v3(fabsf(m.x), fabsf(m.y), fabsf(m.z)), box->form);
    vec3 t1 = sub3(neg3(n), k);
    vec3 t2 = add3(neg3(n), k);
    float tN = gl_max(gl_max(t1.x, t1.y), t1.z);
    float tF = gl_min(gl_min(t2.x, t2.y), t2.z);
    /* a NaN operand (0 * inf for an axis-parallel ray, trap T5) entered the min / max chains: GLSL leaves min/max of a NaN undefined,
     * and whether this box is hit, missed or hit at another distance is the implementation's choice -- also when the result is finite */
    if (t1.x != t1.x || t1.y != t1.y || t1.z != t1.z || t2.x != t2.x || t2.y != t2.y || t2.z != t2.z) iv->tag |= ORC_TAG_BOX_NAN;
    if (tN > tF || tF < 0.0f) return 0;
    if (tN >= tmin) return 0;
    vec3 nor;
    nor.x = -gl_sign(rdd.x) * gl_step(t1.y, t1.x) * gl_step(t1.z, t1.x);
    nor.y = -gl_sign(rdd.y) * gl_step(t1.z, t1.y) * gl_step(t1.x, t1.y);
    nor.z = -gl_sign(rdd.z) * gl_step(t1.x, t1.z) * gl_step(t1.y, t1.z);
    *t = tN;
    iv->opt_normal = rotate(quat_inv(box->quat_rotation), nor);
    if (!(tN == tN)) { iv->cnt->box_nan_hits++; iv->tag |= ORC_TAG_BOX_NAN; }
    else if (tN < 0.0f) { iv->cnt->box_inside_hits++; iv->tag |= ORC_TAG_BOX_INSIDE; }
    return 1;
}
static vec4 getBoxTexture(inv_t* iv, vec3 pt, vec3 normal, int num) /* :428-436 */
{
    const rt_box* box = &iv->boxes[num];
    vec3 pos = rotate(box->quat_rotation, box->pos);
    pt = rotate(box->quat_rotation, pt);
    normal = rotate(box->quat_rotation, normal);
    vec4 a = tex_fetch(iv, ORC_TEX_BOX, SITE_HIT, 0, 0, TYPE_BOX, num, v2(0.5f * (pt.z - pos.z) - 0.5f, 0.5f * (pt.y - pos.y) - 0.5f), LOD_IMPLICIT);
    vec4 b = tex_fetch(iv, ORC_TEX_BOX, SITE_HIT, 1, 0, TYPE_BOX, num, v2(0.5f * (pt.z - pos.z) - 0.5f, 0.5f * (pt.x - pos.x) - 0.5f), LOD_IMPLICIT);
    vec4 c = tex_fetch(iv, ORC_TEX_BOX, SITE_HIT, 2, 0, TYPE_BOX, num, v2(0.5f * (pt.x - pos.x) - 0.5f, 0.5f * (pt.y - pos.y) - 0.5f), LOD_IMPLICIT);
    float wx = fabsf(normal.x), wy = fabsf(normal.y), wz = fabsf(normal.z);
    return v4(wx * a.x + wy * b.x + wz * c.x, wx * a.y + wy * b.y + wz * c.y, wx * a.z + wy * b.z + wz * c.z,
              wx * a.w + wy * b.w + wz * c.w);
}

/* ---- torus section :438-497 ---- */
static inline vec2 cmul(vec2 c1, vec2 c2) { return v2(c1.x * c2.x - c1.y * c2.y, c1.x * c2.y + c1.y * c2.x); } /* :439-441 */
static inline vec2 cinv(vec2 c) { float d = dot2(c, c); return v2(c.x / d, -c.y / d); }                      /* :442-444 */
static vec2 cTorus(vec2 t, vec3 ro, vec3 rd, vec2 torus) /* :445-455 */
{
    float R2 = torus.x * torus.x;
    float r2 = torus.y * torus.y;
    vec2 t2 = v2(t.x * t.x - t.y * t.y, 2.0f * t.x * t.y);
    float drd = dot3(rd, rd), dro = dot3(ro, rd), doo = dot3(ro, ro);
    vec2 res = v2(t2.x * drd + 2.0f * t.x * dro + (doo + R2 - r2), t2.y * drd + 2.0f * t.y * dro + 0.0f);
    res = cmul(res, res);
    vec2 rdxy = v2(rd.x, rd.y), roxy = v2(ro.x, ro.y);
    float k = 4.0f * R2;
    float axy = dot2(rdxy, rdxy), bxy = dot2(roxy, rdxy), cxy = dot2(roxy, roxy);
    vec2 res2 = v2(k * (t2.x * axy + 2.0f * t.x * bxy + cxy), k * (t2.y * axy + 2.0f * t.y * bxy + 0.0f));
    return v2(res.x - res2.x, res.y - res2.y);
}
static float DKstep(vec2* c0, vec2 c1, vec2 c2, vec2 c3, vec3 ro, vec3 rd, vec2 torus) /* :456-461 */
{
    vec2 fc = cTorus(*c0, ro, rd, torus);
    vec2 d1 = v2(c0->x - c1.x, c0->y - c1.y), d2 = v2(c0->x - c2.x, c0->y - c2.y), d3 = v2(c0->x - c3.x, c0->y - c3.y);
    fc = cmul(fc, cinv(cmul(d1, cmul(d2, d3))));
    c0->x -= fc.x;
    c0->y -= fc.y;
    return gl_max(fabsf(fc.x), fabsf(fc.y));
}
static int intersectTorus(inv_t* iv, vec3 ro, vec3 rd, int num, float tmin, float* t) /* :462-487 */
{
    const float eps = 0.001f;
    const rt_torus* torus = &iv->toruses[num];
    ro = rotate(torus->quat_rotation, sub3(ro, torus->pos));
    rd = rotate(torus->quat_rotation, rd);
    vec2 c0 = v2(1.0f, 0.0f);
    vec2 c1 = v2(0.4f, 0.9f);
    vec2 c2 = cmul(c1, v2(0.4f, 0.9f));
    vec2 c3 = cmul(c2, v2(0.4f, 0.9f));
    int i;
    iv->cnt->dk_solves++;
    for (i = 0; i < 60; i++) {
        iv->cnt->dk_sweeps++;
        float e = DKstep(&c0, c1, c2, c3, ro, rd, torus->form);
        e = gl_max(e, DKstep(&c1, c2, c3, c0, ro, rd, torus->form));
        e = gl_max(e, DKstep(&c2, c3, c0, c1, ro, rd, torus->form));
        e = gl_max(e, DKstep(&c3, c0, c1, c2, ro, rd, torus->form));
        if (e < eps) break;
    }
    if (i == 60) iv->cnt->dk_capped++;
    vec4 rs = v4(c0.x, c1.x, c2.x, c3.x);
    vec4 ri = v4(fabsf(c0.y), fabsf(c1.y), fabsf(c2.y), fabsf(c3.y));
    if (ri.x > eps || rs.x < 0.0f) rs.x = 10000.0f;
    if (ri.y > eps || rs.y < 0.0f) rs.y = 10000.0f;
    if (ri.z > eps || rs.z < 0.0f) rs.z = 10000.0f;
    if (ri.w > eps || rs.w < 0.0f) rs.w = 10000.0f;
    *t = gl_min(gl_min(rs.x, rs.y), gl_min(rs.z, rs.w));
    return *t > 0.0f && *t < 100.0f && *t < tmin;
}
static vec3 getTorusNormal(const inv_t* iv, vec3 ro, vec3 rd, float t, int num) /* :488-496 */
{
    const rt_torus* torus = &iv->toruses[num];
    ro = rotate(torus->quat_rotation, sub3(ro, torus->pos));
    rd = rotate(torus->quat_rotation, rd);
    vec3 pos = add3(ro, scale3(rd, t));
    float s = dot3(pos, pos) - torus->form.y * torus->form.y;
    float RR = torus->form.x * torus->form.x;
    vec3 normal = mul3(pos, v3(s - RR * 1.0f, s - RR * 1.0f, s - RR * -1.0f));
    return normalize3(rotate(quat_inv(torus->quat_rotation), normal));
}

/* ---- surface section :499-585 ---- */
static int checkSurfaceEdges(vec3 o, vec3 d, float* tMin, float* tMax, vec3 v_min, vec3 v_max, float epsilon) /* :500-512 */
{
    vec3 pt = add3(scale3(d, *tMin), o);
    if (!isBetween(pt, v_min, v_max)) {
        if (*tMax < epsilon) return 0;
        pt = add3(scale3(d, *tMax), o);
        if (!isBetween(pt, v_min, v_max)) return 0;
        swapf(tMin, tMax);
    }
    return 1;
}
static int intersectSurface(inv_t* iv, vec3 ro, vec3 rd, int num, float tmin, float* t) /* :513-572 */
{
    vec3 orig_ro = ro;
    vec3 orig_rd = rd;
    const rt_surface* surface = &iv->surfaces[num];
    ro = rotate(surface->quat_rotation, sub3(ro, surface->pos));
    rd = rotate(surface->quat_rotation, rd);

    float a = surface->a, b = surface->b, c = surface->c, d = surface->d, e = surface->e, f = surface->f;
    float d1 = rd.x, d2 = rd.y, d3 = rd.z, o1 = ro.x, o2 = ro.y, o3 = ro.z;

    float p1 = 2.0f * a * d1 * o1 + 2.0f * b * d2 * o2 + 2.0f * c * d3 * o3 + d * d3 + d2 * e;
    float p2 = a * d1 * d1 + b * d2 * d2 + c * d3 * d3;
    float p3 = a * o1 * o1 + b * o2 * o2 + c * o3 * o3 + d * o3 + e * o2 + f;
    float p4 = sqrtf(p1 * p1 - 4.0f * p2 * p3);

    if (fabsf(p2) < 1e-6f) { /* :541-545, trap T4: inverted comparison, no clip-box test */
        *t = -p3 / p1;
        if (*t > tmin) iv->cnt->t4_taken++;
        return *t > tmin;
    }

    float mn = FLT_MAX_GLSL;
    float mx = FLT_MAX_GLSL;
    float t1 = (-p1 - p4) / (2.0f * p2);
    float t2 = (-p1 + p4) / (2.0f * p2);
    float epsilon = 1e-4f;
    if (t1 > epsilon && t1 < mn) { mn = t1; mx = t2; }
    if (t2 > epsilon && t2 < mn) { mn = t2; mx = t1; }
    if (!checkSurfaceEdges(orig_ro, orig_rd, &mn, &mx, surface->v_min, surface->v_max, epsilon)) return 0;
    *t = mn;
    return *t < tmin;
}
static vec3 getSurfaceNormal(const inv_t* iv, vec3 ro, vec3 rd, float t, int num) /* :573-584 */
{
    const rt_surface* surface = &iv->surfaces[num];
    ro = sub3(ro, surface->pos);
    ro = rotate(surface->quat_rotation, ro);
    rd = rotate(surface->quat_rotation, rd);
    vec3 tm = add3(scale3(rd, t), ro);
    vec3 normal = v3(2.0f * surface->a * tm.x, 2.0f * surface->b * tm.y + surface->e, 2.0f * surface->c * tm.z + surface->d);
    normal = rotate(quat_inv(surface->quat_rotation), normal);
    return normalize3(normal);
}

static float calcInter(inv_t* iv, vec3 ro, vec3 rd, int* num, int* type) /* :587-628 */
{
    float tmin = maxDist;
    float t = 0.0f;
    int i;
    iv->cnt->rays_closest++;
    iv->step++;
    for (i = 0; i < iv->PLANE_SIZE; i++) {
        iv->cnt->tests[TYPE_PLANE]++;
        if (intersectPlane(ro, rd, iv->planes[i].normal, iv->planes[i].pos, tmin, &t)) { *num = i; tmin = t; *type = TYPE_PLANE; }
    }
    for (i = 0; i < iv->SPHERE_SIZE; i++) {
        iv->cnt->tests[TYPE_SPHERE]++;
        if (intersectSphere(ro, rd, iv->spheres[i].obj, iv->spheres[i].hollow != 0, tmin, &t)) { *num = i; tmin = t; *type = TYPE_SPHERE; }
    }
    for (i = 0; i < iv->SURFACE_SIZE; i++) {
        iv->cnt->tests[TYPE_SURFACE]++;
        if (intersectSurface(iv, ro, rd, i, tmin, &t)) { *num = i; tmin = t; *type = TYPE_SURFACE; }
    }
    for (i = 0; i < iv->BOX_SIZE; i++) {
        iv->cnt->tests[TYPE_BOX]++;
        if (intersectBox(iv, ro, rd, i, tmin, &t)) { *num = i; tmin = t; *type = TYPE_BOX; }
    }
    for (i = 0; i < iv->TORUS_SIZE; i++) {
        iv->cnt->tests[TYPE_TORUS]++;
        if (intersectTorus(iv, ro, rd, i, tmin, &t)) { *num = i; tmin = t; *type = TYPE_TORUS; iv->tag |= ORC_TAG_TORUS; }
    }
    for (i = 0; i < iv->RING_SIZE; i++) {
        iv->cnt->tests[TYPE_RING]++;
        if (intersectRing(iv, ro, rd, i, tmin, &t)) { *num = i; tmin = t; *type = TYPE_RING; }
    }
    for (i = 0; i < iv->LIGHT_POINT_SIZE; i++) {
        iv->cnt->tests[TYPE_POINT_LIGHT]++;
        if (intersectSphere(ro, rd, iv->lights_point[i].pos, 0, tmin, &t)) { *num = i; tmin = t; *type = TYPE_POINT_LIGHT; }
    }
    return tmin;
}

static float inShadow(inv_t* iv, vec3 ro, vec3 rd, float dist) /* :630-658 */
{
    float t = 0.0f;
    float shadow = 0.0f;
    int i;
    iv->cnt->rays_shadow++;
    for (i = 0; i < iv->SPHERE_SIZE; i++) {
        iv->cnt->tests[TYPE_SPHERE]++;
        if (intersectSphere(ro, rd, iv->spheres[i].obj, 0, dist, &t)) shadow = 1.0f;
    }
    for (i = 0; i < iv->SURFACE_SIZE; i++) {
        iv->cnt->tests[TYPE_SURFACE]++;
        if (intersectSurface(iv, ro, rd, i, dist, &t)) shadow = 1.0f;
    }
    for (i = 0; i < iv->BOX_SIZE; i++) {
        iv->cnt->tests[TYPE_BOX]++;
        if (intersectBox(iv, ro, rd, i, dist, &t)) shadow = 1.0f;
    }
    for (i = 0; i < iv->TORUS_SIZE; i++) {
        iv->cnt->tests[TYPE_TORUS]++;
        if (intersectTorus(iv, ro, rd, i, dist, &t)) { shadow = 1.0f; iv->tag |= ORC_TAG_TORUS; }
    }
    for (i = 0; i < iv->RING_SIZE; i++) {
        iv->cnt->tests[TYPE_RING]++;
        if (intersectRing(iv, ro, rd, i, dist, &t)) {
            const rt_ring* ring = &iv->rings[i];
            if (ring->textureNum > 0) shadow += getRingTexture(iv, SITE_SHADOW, iv->light_index, i, iv->opt_uv).w;
            else shadow = 1.0f;
        }
    }
    /* planes: skipped, "#if PLANE_ONESIDE == 0" is false (:21,652-655; trap T1) */
    return gl_min(shadow, 1.0f);
}

static void calcShade2(inv_t* iv, vec3 light_dir, vec3 light_color, float intensity, vec3 pt, vec3 rd, const rt_material* material,
                       vec3 normal, int doShadow, float dist, float distDiv, vec3* diffuse, vec3* specular) /* :660-679 */
{
    light_dir = normalize3(light_dir);
    float dp = gl_clamp(dot3(normal, light_dir), 0.0f, 1.0f);
    light_color = scale3(light_color, dp);
    if (doShadow) {
        float sh = 1.0f - inShadow(iv, pt, light_dir, dist);
        vec3 shadow = v3(gl_max(sh, iv->SHADOW_AMBIENT.x), gl_max(sh, iv->SHADOW_AMBIENT.y), gl_max(sh, iv->SHADOW_AMBIENT.z));
        light_color = mul3(light_color, shadow);
    }
    *diffuse = add3(*diffuse, div3s(scale3(scale3(mul3(light_color, material->color), material->diffuse), intensity), distDiv));
    if (material->specular > 0) {
        vec3 reflection = gl_reflect(light_dir, normal);
        float specDp = gl_clamp(dot3(rd, reflection), 0.0f, 1.0f);
        *specular = add3(*specular, div3s(scale3(scale3(light_color, powf(specDp, (float)material->specular)), intensity), distDiv));
    }
}

static vec3 calcShade(inv_t* iv, vec3 pt, vec3 rd, const rt_material* material, vec3 normal, int doShadow) /* :681-709 */
{
    float dist, distDiv;
    vec3 light_color, light_dir;
    vec3 diffuse = v3(0.0f, 0.0f, 0.0f);
    vec3 specular = v3(0.0f, 0.0f, 0.0f);
    vec3 pixelColor = mul3(iv->AMBIENT_COLOR, material->color);
    int i;
    for (i = 0; i < iv->LIGHT_POINT_SIZE; i++) {
        const rt_light_point* light = &iv->lights_point[i];
        light_color = light->color;
        light_dir = sub3(v3(light->pos.x, light->pos.y, light->pos.z), pt);
        dist = length3(light_dir);
        distDiv = 1.0f + light->linear_k * dist + light->quadratic_k * dist * dist;
        iv->light_index = i;
        calcShade2(iv, light_dir, light_color, light->intensity, pt, rd, material, normal, doShadow, dist, distDiv, &diffuse, &specular);
    }
    for (i = 0; i < iv->LIGHT_DIRECT_SIZE; i++) {
        light_color = iv->lights_direct[i].color;
        light_dir = neg3(iv->lights_direct[i].direction);
        dist = maxDist;
        distDiv = 1.0f;
        iv->light_index = iv->LIGHT_POINT_SIZE + i;
        calcShade2(iv, light_dir, light_color, iv->lights_direct[i].intensity, pt, rd, material, normal, doShadow, dist, distDiv, &diffuse,
                   &specular);
    }
    pixelColor = add3(pixelColor, add3(scale3(diffuse, material->kd), scale3(specular, material->ks)));
    return pixelColor;
}

static float getFresnel(vec3 normal, vec3 rd, float reflection) /* :711-715 */
{
    float ndotv = gl_clamp(dot3(normal, neg3(rd)), 0.0f, 1.0f);
    return reflection + (1.0f - reflection) * powf(1.0f - ndotv, 5.0f);
}

static float FresnelReflectAmount(float n1, float n2, vec3 normal, vec3 incident, float refl) /* :717-742, DO_FRESNEL 1 */
{
    float r0 = (n1 - n2) / (n1 + n2);
    r0 *= r0;
    float cosX = -dot3(normal, incident);
    if (n1 > n2) {
        float n = n1 / n2;
        float sinT2 = n * n * (1.0f - cosX * cosX);
        if (sinT2 > 1.0f) return 1.0f;
        cosX = sqrtf(1.0f - sinT2);
    }
    float x = 1.0f - cosX;
    float ret = r0 + (1.0f - r0) * x * x * x * x * x;
    ret = (refl + (1.0f - refl) * ret);
    return ret;
}

static hit_record get_hit_info(inv_t* iv, vec3 ro, vec3 rd, vec3 pt, float t, int num, int type) /* :744-784 */
{
    hit_record hr;
    memset(&hr, 0, sizeof hr);
    if (type == TYPE_SPHERE) {
        const rt_sphere* sphere = &iv->spheres[num];
        hr.mat = sphere->mat;
        hr.normal = normalize3(sub3(pt, v3(sphere->obj.x, sphere->obj.y, sphere->obj.z)));
        hr.bias_mult = 0.0f;
        hr.alpha = 1.0f;
        if (sphere->textureNum != 0) {
            vec4 texColor = getSphereTexture(iv, hr.normal, sphere->quat_rotation, sphere->textureNum, num);
            hr.mat.color = v3(texColor.x, texColor.y, texColor.z);
            hr.alpha = texColor.w;
        }
    }
    if (type == TYPE_PLANE) {
        hr.mat = iv->planes[num].mat;
        hr.normal = normalize3(iv->planes[num].normal);
        hr.alpha = 1.0f;
    }
    if (type == TYPE_SURFACE) {
        hr.mat = iv->surfaces[num].mat;
        hr.normal = getSurfaceNormal(iv, ro, rd, t, num);
        hr.alpha = 1.0f;
    }
    if (type == TYPE_BOX) {
        const rt_box* box = &iv->boxes[num];
        hr.mat = box->mat;
        hr.normal = iv->opt_normal;
        hr.alpha = 1.0f;
        if (box->textureNum != 0) {
            vec4 c = getBoxTexture(iv, pt, iv->opt_normal, num);
            hr.mat.color = v3(c.x, c.y, c.z);
        }
    }
    if (type == TYPE_TORUS) {
        hr.mat = iv->toruses[num].mat;
        hr.normal = getTorusNormal(iv, ro, rd, t, num);
        hr.alpha = 1.0f;
    }
    if (type == TYPE_RING) {
        const rt_ring* ring = &iv->rings[num];
        hr.mat = ring->mat;
        hr.normal = getRingNormal(iv, num);
        hr.alpha = 1.0f;
        if (ring->textureNum != 0) {
            vec4 texColor = getRingTexture(iv, SITE_HIT, 0, num, iv->opt_uv);
            hr.mat.color = v3(texColor.x, texColor.y, texColor.z);
            hr.alpha = texColor.w;
        }
    }
    float distance = length3(sub3(pt, ro));
    hr.bias_mult = (9e-3f * distance + 35.0f) / 35e3f;
    return hr;
}

static vec3 getReflectedColor(inv_t* iv, vec3 ro, vec3 rd) /* :787-802 */
{
    vec3 color = v3(0.0f, 0.0f, 0.0f);
    vec3 pt;
    int num = 0, type = -1; /* uninitialised in GLSL (trap T3): pinned to "nothing" */
    float t = calcInter(iv, ro, rd, &num, &type);
    if (type == TYPE_POINT_LIGHT) { iv->cnt->light_hits++; return iv->lights_point[num].color; }
    if (t < maxDist) {
        pt = add3(ro, scale3(rd, t));
        hit_record hr = get_hit_info(iv, ro, rd, pt, t, num, type);
        ro = dot3(rd, hr.normal) < 0.0f ? add3(pt, scale3(hr.normal, hr.bias_mult)) : sub3(pt, scale3(hr.normal, hr.bias_mult));
        color = calcShade(iv, ro, rd, &hr.mat, hr.normal, 1);
    } else {
        iv->cnt->side_miss++;
    }
    return color;
}

static vec4 shade_pixel(inv_t* iv) /* main(), :804-902 */
{
    float reflectMultiplier, refractMultiplier, tm;
    rt_material mat;
    vec3 pt, n;
    vec3 mask = v3(1.0f, 1.0f, 1.0f);
    vec3 color = v3(0.0f, 0.0f, 0.0f);
    vec3 ro = iv->scene->camera_pos;
    vec3 rd = getRayDir(iv);
    float absorbDistance = 0.0f;
    int type = 0;
    int num = 0;
    hit_record hr;
    uint64_t segments = 0;

    for (int i = 0; i < iv->ITERATIONS; i++) {
        if (segments >= ORC_SEGMENT_CAP) { iv->cnt->segment_cap_hits++; break; }
        segments++;
        tm = calcInter(iv, ro, rd, &num, &type);
        if (segments == 1 && (g_primary_out || g_primary_t_in)) {   /* diagnostic, see orc_set_primary_buffers */
            const size_t px = (size_t)(iv->frag_y - 0.5f) * (size_t)iv->fr->fb_width + (size_t)(iv->frag_x - 0.5f);
            if (iv->frag_x < (float)iv->fr->fb_width && iv->frag_y < (float)iv->fr->fb_height) {
                if (g_primary_t_in && tm < maxDist && type == TYPE_TORUS && g_primary_t_in[px] > 0.0f) tm = g_primary_t_in[px];
                if (g_primary_out) { float* o = g_primary_out + px * 4; o[0] = tm; o[1] = (float)(tm < maxDist ? type : -1); o[2] = (float)(tm < maxDist ? num : -1); o[3] = 0.0f; }
            }
        }
        if (tm < maxDist) {
            pt = add3(ro, scale3(rd, tm));
            hr = get_hit_info(iv, ro, rd, pt, tm, num, type);

            if (type == TYPE_POINT_LIGHT) {
                iv->cnt->light_hits++;
                color = add3(color, mul3(iv->lights_point[num].color, mask));
                break;
            }
            mat = hr.mat;
            n = hr.normal;
            int outside = dot3(rd, n) < 0.0f;
            n = outside ? n : neg3(n);

            if (mat.refraction > 0.0f)
                reflectMultiplier = FresnelReflectAmount(outside ? 1.0f : mat.refraction, outside ? mat.refraction : 1.0f, rd, n, mat.reflection);
            else
                reflectMultiplier = getFresnel(n, rd, mat.reflection);
            refractMultiplier = 1.0f - reflectMultiplier;

            if (mat.refraction > 0.0f) { /* refractive :851-873 */
                if (outside && mat.reflection > 0.0f) {
                    vec3 rc = getReflectedColor(iv, add3(pt, scale3(n, hr.bias_mult)), gl_reflect(rd, n));
                    color = add3(color, mul3(scale3(rc, reflectMultiplier), mask));
                    mask = scale3(mask, refractMultiplier);
                } else if (!outside) {
                    absorbDistance += tm;
                    vec3 absorb = v3(expf(-mat.absorb.x * absorbDistance), expf(-mat.absorb.y * absorbDistance), expf(-mat.absorb.z * absorbDistance));
                    mask = mul3(mask, absorb);
                }
                if (reflectMultiplier >= 1.0f) { iv->cnt->tir_breaks++; iv->tag |= ORC_TAG_TIR; break; }
                ro = sub3(pt, scale3(n, hr.bias_mult));
                rd = gl_refract(rd, n, outside ? 1.0f / mat.refraction : mat.refraction);
                iv->cnt->refract_segments++;
                iv->tag |= ORC_TAG_REFRACT;
                i--; /* REFLECT_REDUCE_ITERATION is defined (:22,870-872; trap T2) */
            } else if (mat.reflection > 0.0f) { /* reflective :874-880 */
                ro = add3(pt, scale3(n, hr.bias_mult));
                color = add3(color, mul3(scale3(calcShade(iv, ro, rd, &mat, n, 1), refractMultiplier), mask));
                rd = gl_reflect(rd, n);
                mask = scale3(mask, reflectMultiplier);
            } else { /* diffuse :881-890 */
                color = add3(color, scale3(mul3(calcShade(iv, add3(pt, scale3(n, hr.bias_mult)), rd, &mat, n, 1), mask), hr.alpha));
                if (hr.alpha < 1.0f) {
                    iv->cnt->alpha_pass++;
                    ro = sub3(pt, scale3(n, hr.bias_mult));
                    mask = scale3(mask, 1.0f - hr.alpha);
                } else {
                    break;
                }
            }
        } else {
            vec4 sky = sky_fetch(iv, rd);
            color = add3(color, mul3(v3(sky.x, sky.y, sky.z), mask));
            break;
        }
    }
    if (segments > iv->cnt->max_segments) iv->cnt->max_segments = segments;
    return v4(color.x, color.y, color.z, 1.0f);
}

/* GLWrapper::to_string round trip of the two colour constants: std::to_string(float) = "%f",
 * then the GLSL compiler reads the literal back as a float (GLWrapper.cpp:246-247,279-282; T9). */
static float text_round_trip(float v)
{
    char buf[64];
    snprintf(buf, sizeof buf, "%f", (double)v);
    return strtof(buf, NULL);
}

static void counters_add(orc_counters* a, const orc_counters* b)
{
    a->rays_closest += b->rays_closest; a->rays_shadow += b->rays_shadow;
    for (int k = 0; k < 7; k++) a->tests[k] += b->tests[k];
    a->dk_solves += b->dk_solves; a->dk_sweeps += b->dk_sweeps; a->dk_capped += b->dk_capped; a->t4_taken += b->t4_taken;
    a->refract_segments += b->refract_segments; a->tir_breaks += b->tir_breaks; a->alpha_pass += b->alpha_pass;
    a->side_miss += b->side_miss; a->light_hits += b->light_hits; a->box_nan_hits += b->box_nan_hits;
    a->box_inside_hits += b->box_inside_hits; a->segment_cap_hits += b->segment_cap_hits;
    if (b->max_segments > a->max_segments) a->max_segments = b->max_segments;
}

static void inv_init(inv_t* iv, const orc_frame* fr, orc_counters* cnt)
{
    memset(iv, 0, sizeof *iv);
    iv->fr = fr;
    iv->scene = (const rt_scene*)fr->scene_buf;
    iv->spheres = (const rt_sphere*)fr->spheres_buf;
    iv->planes = (const rt_plane*)fr->planes_buf;
    iv->surfaces = (const rt_surface*)fr->surfaces_buf;
    iv->boxes = (const rt_box*)fr->boxes_buf;
    iv->toruses = (const rt_torus*)fr->toruses_buf;
    iv->rings = (const rt_ring*)fr->rings_buf;
    iv->lights_point = (const rt_light_point*)fr->lights_point_buf;
    iv->lights_direct = (const rt_light_direct*)fr->lights_direct_buf;
    iv->SPHERE_SIZE = fr->defines.sphere_size; iv->PLANE_SIZE = fr->defines.plane_size; iv->SURFACE_SIZE = fr->defines.surface_size;
    iv->BOX_SIZE = fr->defines.box_size; iv->TORUS_SIZE = fr->defines.torus_size; iv->RING_SIZE = fr->defines.ring_size;
    iv->LIGHT_POINT_SIZE = fr->defines.light_point_size; iv->LIGHT_DIRECT_SIZE = fr->defines.light_direct_size;
    iv->ITERATIONS = fr->defines.iterations;
    iv->AMBIENT_COLOR = v3(text_round_trip(fr->defines.ambient_color[0]), text_round_trip(fr->defines.ambient_color[1]), text_round_trip(fr->defines.ambient_color[2]));
    iv->SHADOW_AMBIENT = v3(text_round_trip(fr->defines.shadow_ambient[0]), text_round_trip(fr->defines.shadow_ambient[1]), text_round_trip(fr->defines.shadow_ambient[2]));
    iv->cnt = cnt;
}

/* ---- quad execution (texture_lod == 1) ---- */
static __thread quad_t* tls_quad;
static void lane_entry(void)
{
    quad_t* q = tls_quad;
    quad_lane* L = &q->lane[q->current];
    L->color = shade_pixel(&L->iv);
    L->done = 1; /* returning resumes q->sched through uc_link */
}
#define ORC_CORO_STACK (256 * 1024)

static void render_quad(quad_t* q, const inv_t* base, const orc_frame* fr, int qx, int qy, int y0, int y1, float* out_rgba, orc_counters* local,
                        orc_counters* discard)
{
    for (int k = 0; k < 4; k++) {
        quad_lane* L = &q->lane[k];
        const int x = 2 * qx + (k & 1), y = 2 * qy + (k >> 1);
        L->active = 1; /* inside the even-rounded framebuffer by construction: helper invocation if outside the real one */
        L->done = 0;
        L->waiting = 0;
        L->iv = *base;
        const int stored = x < fr->fb_width && y < fr->fb_height && y >= y0 && y < y1;
        L->iv.cnt = stored ? local : discard;
        L->iv.quad = q;
        L->iv.quad_slot = k;
        L->iv.step = 0;
        L->iv.tag = 0;
        L->iv.frag_x = (float)x + 0.5f;
        L->iv.frag_y = (float)y + 0.5f;
        getcontext(&L->ctx);
        L->ctx.uc_stack.ss_sp = L->stack;
        L->ctx.uc_stack.ss_size = ORC_CORO_STACK;
        L->ctx.uc_link = &q->sched;
        makecontext(&L->ctx, lane_entry, 0);
    }
    tls_quad = q;
    for (;;) {
        int all_done = 1;
        for (int k = 0; k < 4; k++) {
            quad_lane* L = &q->lane[k];
            if (!L->done) all_done = 0;
            if (!L->done && !L->waiting) {
                q->current = k;
                swapcontext(&q->sched, &L->ctx); /* runs lane k until it asks for a texel or finishes */
            }
        }
        if (all_done) break;
        int any_waiting = 0, any_runnable = 0;
        for (int k = 0; k < 4; k++) {
            if (q->lane[k].done) continue;
            if (q->lane[k].waiting) any_waiting = 1; else any_runnable = 1;
        }
        if (!any_runnable && any_waiting) quad_resolve(q, fr);
    }
    for (int k = 0; k < 4; k++) {
        const int x = 2 * qx + (k & 1), y = 2 * qy + (k >> 1);
        if (x < fr->fb_width && y < fr->fb_height && y >= y0 && y < y1) {
            float* o = out_rgba + ((size_t)(y - y0) * (size_t)fr->fb_width + (size_t)x) * 4;
            const vec4 c = q->lane[k].color;
            o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
            if (g_tag_buffer) g_tag_buffer[(size_t)(y - y0) * (size_t)fr->fb_width + (size_t)x] = q->lane[k].iv.tag;
        }
    }
}

/* Render rows [y0, y1) (row 0 = bottom, gl_FragCoord convention) into out_rgba, which holds
 * (y1 - y0) * fb_width RGBA float pixels. nthreads <= 0: all cores. Returns 0. */
int orc_render(const orc_frame* fr, int y0, int y1, float* out_rgba, orc_counters* counters, int nthreads)
{
    orc_counters total;
    memset(&total, 0, sizeof total);
    if (y0 < 0) y0 = 0;
    if (y1 > fr->fb_height) y1 = fr->fb_height;
    /* (re)build the mip chains before the parallel region (mip_get is not thread-safe when it builds); repeated until a pass builds
     * nothing, so that a chain recycled while a later one was built is back before the threads start */
    for (int pass = 0, built = -1; fr->texture_lod && built != g_mips_built && pass < 8; pass++) {
        built = g_mips_built;
        for (int k = 0; k < ORC_TEX_COUNT; k++) {
            (void)mip_lookup(&fr->tex[k], pass == 0 && g_mip_validate); /* drops a stale chain whose address was reused for other texels */
            (void)mip_get(&fr->tex[k]);
        }
        if (fr->skybox.gen_mipmap && fr->skybox.face_size > 0)
            for (int f = 0; f < 6; f++) {
                if (!fr->skybox.faces[f]) continue;
                orc_texture ft;
                ft.width = ft.height = fr->skybox.face_size; ft.channels = fr->skybox.channels; ft.wrap = 1; ft.texels = fr->skybox.faces[f];
                (void)mip_lookup(&ft, pass == 0 && g_mip_validate);
                (void)mip_get(&ft);
            }
    }
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
    else omp_set_num_threads(omp_get_num_procs());
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        orc_counters local, discard;
        memset(&local, 0, sizeof local);
        memset(&discard, 0, sizeof discard);
        inv_t iv;
        inv_init(&iv, fr, &local);
        if (!fr->texture_lod) {
#pragma omp for schedule(dynamic, 1)
            for (int y = y0; y < y1; y++) {
                for (int x = 0; x < fr->fb_width; x++) {
                    iv.frag_x = (float)x + 0.5f;
                    iv.frag_y = (float)y + 0.5f;
                    iv.step = 0;
                    iv.tag = 0;
                    vec4 c = shade_pixel(&iv);
                    float* o = out_rgba + ((size_t)(y - y0) * (size_t)fr->fb_width + (size_t)x) * 4;
                    o[0] = c.x; o[1] = c.y; o[2] = c.z; o[3] = c.w;
                    if (g_tag_buffer) g_tag_buffer[(size_t)(y - y0) * (size_t)fr->fb_width + (size_t)x] = iv.tag;
                }
            }
        } else {
            quad_t* q = (quad_t*)calloc(1, sizeof(quad_t));
            for (int k = 0; k < 4; k++) q->lane[k].stack = (char*)malloc(ORC_CORO_STACK);
            const int qy0 = y0 / 2, qy1 = (y1 + 1) / 2, qx1 = (fr->fb_width + 1) / 2;
#pragma omp for schedule(dynamic, 1)
            for (int qy = qy0; qy < qy1; qy++)
                for (int qx = 0; qx < qx1; qx++) render_quad(q, &iv, fr, qx, qy, y0, y1, out_rgba, &local, &discard);
            for (int k = 0; k < 4; k++) free(q->lane[k].stack);
            free(q);
        }
#pragma omp critical
        counters_add(&total, &local);
    }
    if (counters) *counters = total;
    return 0;
}

/* ---- single-function entry points for the known-answer tests (tests/test_oracle_kat.py) ---- */
static void kat_inv(inv_t* iv, orc_frame* fr, orc_counters* cnt, const void* rec, int type)
{
    memset(fr, 0, sizeof *fr);
    memset(cnt, 0, sizeof *cnt);
    memset(iv, 0, sizeof *iv);
    iv->fr = fr;
    iv->cnt = cnt;
    if (type == TYPE_BOX) iv->boxes = (const rt_box*)rec;
    if (type == TYPE_TORUS) iv->toruses = (const rt_torus*)rec;
    if (type == TYPE_RING) iv->rings = (const rt_ring*)rec;
    if (type == TYPE_SURFACE) iv->surfaces = (const rt_surface*)rec;
}
/* out[0] = hit (0/1), out[1] = t, out[2..4] = opt_normal (box) or opt_uv (ring, 2 values) */
int orc_kat_intersect(int type, const void* record, const float ro[3], const float rd[3], float tmin, int hollow, float out[5])
{
    inv_t iv; orc_frame fr; orc_counters cnt;
    kat_inv(&iv, &fr, &cnt, record, type);
    vec3 o = v3(ro[0], ro[1], ro[2]), d = v3(rd[0], rd[1], rd[2]);
    float t = 0.0f; int hit = 0;
    memset(out, 0, 5 * sizeof(float));
    switch (type) {
    case TYPE_SPHERE: { const float* s = (const float*)record; hit = intersectSphere(o, d, v4(s[0], s[1], s[2], s[3]), hollow, tmin, &t); break; }
    case TYPE_PLANE: { const float* s = (const float*)record; hit = intersectPlane(o, d, v3(s[0], s[1], s[2]), v3(s[3], s[4], s[5]), tmin, &t); break; }
    case TYPE_SURFACE: hit = intersectSurface(&iv, o, d, 0, tmin, &t); break;
    case TYPE_BOX: hit = intersectBox(&iv, o, d, 0, tmin, &t); out[2] = iv.opt_normal.x; out[3] = iv.opt_normal.y; out[4] = iv.opt_normal.z; break;
    case TYPE_TORUS: hit = intersectTorus(&iv, o, d, 0, tmin, &t); break;
    case TYPE_RING: hit = intersectRing(&iv, o, d, 0, tmin, &t); out[2] = iv.opt_uv.x; out[3] = iv.opt_uv.y; break;
    default: return -1;
    }
    out[0] = (float)hit;
    out[1] = t;
    return 0;
}
float orc_kat_atan2(float y, float x) { return orc_atan2(y, x); }
float orc_kat_asin(float x) { return orc_asin(x); }
void orc_kat_rotate(const float q[4], const float v[3], float out[3])
{
    vec3 r = rotate(v4(q[0], q[1], q[2], q[3]), v3(v[0], v[1], v[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_kat_sample2d(const orc_texture* t, float u, float v, float out[4])
{
    vec4 c = sample2d_level0(t, v2(u, v));
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
void orc_kat_sample_cube(const orc_cubemap* c, const float d[3], float out[4])
{
    vec4 r = sample_cube(c, v3(d[0], d[1], d[2]));
    out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = r.w;
}
float orc_kat_text_round_trip(float v) { return text_round_trip(v); }
float orc_kat_log2(float v) { return orc_log2(v); }
/* one pixel with level-0 textures: its colour and its own event counters (debugging aid for count mismatches) */
int orc_kat_pixel(const orc_frame* fr, int x, int y, float out[4], orc_counters* counters)
{
    orc_counters local;
    memset(&local, 0, sizeof local);
    inv_t iv;
    inv_init(&iv, fr, &local);
    iv.frag_x = (float)x + 0.5f;
    iv.frag_y = (float)y + 0.5f;
    iv.step = 0;
    const vec4 c = shade_pixel(&iv);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
    if (counters) *counters = local;
    return 0;
}
void orc_kat_sample2d_lod(const orc_texture* t, float u, float v, float lambda, float out[4])
{
    (void)mip_lookup(t, 1);
    vec4 c = sample2d_lod(t, v2(u, v), lambda);
    out[0] = c.x; out[1] = c.y; out[2] = c.z; out[3] = c.w;
}
/* Diagnostic: replace the texels of mip level `level` (>= 1) of this texture's cached chain by the caller's RGBA8 bytes -- the build
 * container hands over the levels a GL implementation generated (glGenerateMipmap leaves the filter to the implementation; llvmpipe's
 * differs from the integer mean by one LSB on 6-17 % of the texels), so that the plain reference frames can be compared with "the same
 * mip texels" on both sides. The override lives as long as the cached chain (same level-0 bytes at the same address); orc_kat_drop_mips
 * forgets every chain. Returns 0, -1 if the level does not exist. */
int orc_kat_set_mip_level(const orc_texture* t, int level, const uint8_t* in)
{
    (void)mip_lookup(t, 1);
    const orc_mipchain* m = mip_get(t);
    if (!m || level < 1 || level >= m->levels) return -1;
    memcpy(m->data[level], in, (size_t)m->w[level] * m->h[level] * 4);
    return 0;
}
void orc_kat_drop_mips(void)
{
    for (int k = 0; k < g_mips_n; k++)
        for (int l = 0; l < g_mips[k].levels; l++) free(g_mips[k].data[l]);
    g_mips_n = 0;
}
/* copies mip level `level` (RGBA8) into out; returns its width<<16 | height, 0 if the level does not exist */
int orc_kat_mip_level(const orc_texture* t, int level, uint8_t* out)
{
    (void)mip_lookup(t, 1);
    const orc_mipchain* m = mip_get(t);
    if (!m || level < 0 || level >= m->levels) return 0;
    memcpy(out, m->data[level], (size_t)m->w[level] * m->h[level] * 4);
    return (m->w[level] << 16) | m->h[level];
}
