/* smaa_oracle.c -- CPU restatement of the reference's SMAA post-process. TEST INFRASTRUCTURE (the checker):
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product never does.
 *
 * What it restates: the three full-screen passes GLWrapper::draw runs after the tracer
 * (src/GLWrapper.cpp:173-204) with the programs SMAA_Builder assembles (src/SMAA_Builder.h:17-113) from
 * assets/shaders/SMAA.h:
 *   pass 1  SMAAEdgeDetectionVS (SMAA.h:646-651) + SMAALumaEdgeDetectionPS (SMAA.h:689-741)      colour RGBA8 -> edges RG8
 *   pass 2  SMAABlendingWeightCalculationVS (:656-668) + ...PS (:1145-1243) with the diagonal (:835-985), search
 *           (:998-1077), area (:1083-1095) and corner (:1100-1140) helpers                        edges -> weights RGBA8
 *   pass 3  SMAANeighborhoodBlendingVS (:673-676) + ...PS (:1252-1300)                             colour + weights -> screen RGBA8
 * for the four presets (SMAA.h:304-324), SMAA 1x (subsampleIndices = 0, SMAA_Builder.h:169), no predication, no
 * reprojection. The area / search look-up tables are INPUTS (the reference uploads the byte arrays of AreaTex.h / SearchTex.h,
 * SMAA_Builder.h:52-83); nothing of them is stored here.
 *
 * Arithmetic contract (shared with the HIP kernels, raytracing_opengl_amd/csrc/smaa_device.h, written independently):
 *   - float32, no contraction (-ffp-contract=off), IEEE divide and sqrt, round() = round-half-even (rintf);
 *   - every texture is 8-bit UNORM, LINEAR, CLAMP_TO_EDGE, one level. Coordinates are carried in TEXEL space
 *     (t = texcoord * size - 0.5): the pixel (x, y) the fragment shader runs for has texcoord ((x+.5)/W, (y+.5)/H), i.e.
 *     texel-space coordinate exactly (x, y), and every offset the shaders add is a dyadic multiple of a pixel, so the
 *     positions are exact in float32 where the GLSL, working in [0,1] coordinates and on interpolated varyings, carries
 *     ~1e-7 of noise that GL leaves to the implementation (interpolation precision, sub-texel weight precision). This is the
 *     sampler with exact interpolation and infinite sub-texel precision;
 *   - bilinear sample: i0 = floor(t), a = t - i0, taps i0 / i0+1 clamped to [0, n-1] AFTER the integer texel offset of
 *     textureLodOffset is added; texel = byte / 255; result = w00*t00 + w10*t10 + w01*t01 + w11*t11 left to right with
 *     w00 = (1-a)(1-b), w10 = a(1-b), w01 = (1-a)b, w11 = ab (the tracer's texture rule);
 *   - colour write-out to UNORM8: clamp to [0,1], then (uint)(v * 255 + 0.5).
 * Pinned against the reference's own shaders executed on Mesa llvmpipe (oracle/ref_gl/ref_smaa.py,
 * tools/gen_smaa_fixtures.py -> tests/golden/smaa_*.npz; tests/test_smaa_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y; } v2;
typedef struct { float x, y, z, w; } v4;

typedef struct {
    int w, h, c;              /* c interleaved 8-bit channels: 1, 2 or 4 */
    const uint8_t* px;        /* row 0 = t 0 */
} tex_t;

typedef struct {
    float threshold;          /* SMAA_THRESHOLD */
    int max_steps;            /* SMAA_MAX_SEARCH_STEPS */
    int max_steps_diag;       /* SMAA_MAX_SEARCH_STEPS_DIAG; 0 = SMAA_DISABLE_DIAG_DETECTION */
    int corner_rounding;      /* SMAA_CORNER_ROUNDING; < 0 = SMAA_DISABLE_CORNER_DETECTION */
} preset_t;

/* SMAA.h:304-324 */
static const preset_t k_presets[4] = {
    {0.15f, 4, 0, -1},   /* LOW */
    {0.1f, 8, 0, -1},    /* MEDIUM */
    {0.1f, 16, 8, 25},   /* HIGH */
    {0.05f, 32, 16, 25}, /* ULTRA */
};

static inline float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }          /* GLSL step */
static inline float maxf(float a, float b) { return a < b ? b : a; }                       /* GLSL max  */
static inline float saturatef(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float unorm(uint8_t b) { return (float)b / 255.0f; }
static inline uint8_t to_unorm8(float v)
{
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    if (!(v == v)) v = 0.0f;
    return (uint8_t)(uint32_t)(v * 255.0f + 0.5f);
}

static inline v4 texel(const tex_t* t, int i, int j)
{
    const uint8_t* p = t->px + ((size_t)j * (size_t)t->w + (size_t)i) * (size_t)t->c;
    v4 r = {unorm(p[0]), 0.0f, 0.0f, 1.0f};
    if (t->c >= 2) r.y = unorm(p[1]);
    if (t->c >= 4) { r.z = unorm(p[2]); r.w = unorm(p[3]); }
    return r;
}

/* LINEAR + CLAMP_TO_EDGE sample at texel-space position (tx, ty) with an integer texel offset (textureLodOffset) */
static v4 sample_off(const tex_t* t, float tx, float ty, int ox, int oy)
{
    const float fx = floorf(tx), fy = floorf(ty);
    const float a = tx - fx, b = ty - fy;
    const int i0 = clampi((int)fx + ox, 0, t->w - 1), i1 = clampi((int)fx + ox + 1, 0, t->w - 1);
    const int j0 = clampi((int)fy + oy, 0, t->h - 1), j1 = clampi((int)fy + oy + 1, 0, t->h - 1);
    const v4 t00 = texel(t, i0, j0), t10 = texel(t, i1, j0), t01 = texel(t, i0, j1), t11 = texel(t, i1, j1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    v4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}
static v4 sample(const tex_t* t, float tx, float ty) { return sample_off(t, tx, ty, 0, 0); }

/* ---- pass 1: SMAALumaEdgeDetectionPS (SMAA.h:689-741) ------------------------------------------------------ */
static float luma(const tex_t* color, float tx, float ty)
{
    const v4 c = sample(color, tx, ty);
    return c.x * 0.2126f + c.y * 0.7152f + c.z * 0.0722f;   /* dot(rgb, weights), SMAA.h:705-706 */
}

static void edge_pixel(const tex_t* color, const preset_t* P, int x, int y, uint8_t out[2])
{
    const float X = (float)x, Y = (float)y;
    out[0] = out[1] = 0;                                     /* glClear(0) + discard (GLWrapper.cpp:177-178, SMAA.h:718-719) */
    const float L = luma(color, X, Y);
    const float Lleft = luma(color, X - 1.0f, Y), Ltop = luma(color, X, Y - 1.0f);          /* offset[0], SMAA.h:648 */
    const float dx = fabsf(L - Lleft), dy = fabsf(L - Ltop);
    float ex = stepf(P->threshold, dx), ey = stepf(P->threshold, dy);
    if (ex * 1.0f + ey * 1.0f == 0.0f) return;
    const float Lright = luma(color, X + 1.0f, Y), Lbottom = luma(color, X, Y + 1.0f);      /* offset[1] */
    float dz = fabsf(L - Lright), dw = fabsf(L - Lbottom);
    float mx = maxf(dx, dz), my = maxf(dy, dw);
    const float Lleftleft = luma(color, X - 2.0f, Y), Ltoptop = luma(color, X, Y - 2.0f);   /* offset[2] */
    dz = fabsf(Lleft - Lleftleft);
    dw = fabsf(Ltop - Ltoptop);
    mx = maxf(mx, dz);
    my = maxf(my, dw);
    const float final_delta = maxf(mx, my);
    ex *= stepf(final_delta, 2.0f * dx);                     /* SMAA_LOCAL_CONTRAST_ADAPTATION_FACTOR = 2.0 */
    ey *= stepf(final_delta, 2.0f * dy);
    out[0] = to_unorm8(ex);
    out[1] = to_unorm8(ey);
}

/* ---- pass 2 helpers ---------------------------------------------------------------------------------------- */
typedef struct {
    const tex_t *edges, *area, *search;
    const preset_t* P;
} blend_ctx;

/* SMAADecodeDiagBilinearAccess (SMAA.h:835-856) */
static v2 decode_diag2(v2 e)
{
    e.x = e.x * fabsf(5.0f * e.x - 3.75f);
    e.x = rintf(e.x);
    e.y = rintf(e.y);
    return e;
}
static v4 decode_diag4(v4 e)
{
    e.x = e.x * fabsf(5.0f * e.x - 3.75f);
    e.z = e.z * fabsf(5.0f * e.z - 3.75f);
    e.x = rintf(e.x); e.y = rintf(e.y); e.z = rintf(e.z); e.w = rintf(e.w);
    return e;
}

/* SMAASearchDiag1 / SMAASearchDiag2 (SMAA.h:861-892): returns coord.zw, leaves the last fetched edges in *e */
static v2 search_diag1(const blend_ctx* C, float tx, float ty, float dirx, float diry, v2* e)
{
    float cz = -1.0f, cw = 1.0f;
    while (cz < (float)(C->P->max_steps_diag - 1) && cw > 0.9f) {
        tx = 1.0f * dirx + tx;   /* mad(t, float3(dir, 1), coord.xyz) with t = (rt.xy, 1): one texel per step */
        ty = 1.0f * diry + ty;
        cz = 1.0f * 1.0f + cz;
        const v4 s = sample(C->edges, tx, ty);
        e->x = s.x; e->y = s.y;
        cw = e->x * 0.5f + e->y * 0.5f;
    }
    v2 r = {cz, cw};
    return r;
}
static v2 search_diag2(const blend_ctx* C, float tx, float ty, float dirx, float diry, v2* e)
{
    float cz = -1.0f, cw = 1.0f;
    tx += 0.25f;                 /* @SearchDiag2Optimization */
    while (cz < (float)(C->P->max_steps_diag - 1) && cw > 0.9f) {
        tx = 1.0f * dirx + tx;
        ty = 1.0f * diry + ty;
        cz = 1.0f * 1.0f + cz;
        const v4 s = sample(C->edges, tx, ty);
        v2 ee = {s.x, s.y};
        *e = decode_diag2(ee);
        cw = e->x * 0.5f + e->y * 0.5f;
    }
    v2 r = {cz, cw};
    return r;
}

/* SMAAAreaDiag (SMAA.h:898-913), offset = 0: texel space of the 160 x 560 table, diagonal half starts at column 80 */
static v2 area_diag(const blend_ctx* C, v2 dist, v2 e)
{
    const float tx = 20.0f * e.x + dist.x, ty = 20.0f * e.y + dist.y;    /* SMAA_AREATEX_MAX_DISTANCE_DIAG */
    const v4 s = sample(C->area, tx + 80.0f, ty);
    v2 r = {s.x, s.y};
    return r;
}

/* SMAACalculateDiagWeights (SMAA.h:918-985) for the pixel at texel-space (X, Y) whose own edges are e */
static v2 diag_weights(const blend_ctx* C, float X, float Y, v2 e)
{
    v2 weights = {0.0f, 0.0f};
    float dx, dy, dz, dw;
    v2 end = {0.0f, 0.0f};
    if (e.x > 0.0f) {
        const v2 r = search_diag1(C, X, Y, -1.0f, 1.0f, &end);
        dx = r.x; dz = r.y;
        dx += (end.y > 0.9f) ? 1.0f : 0.0f;
    } else {
        dx = 0.0f; dz = 0.0f;
    }
    {
        const v2 r = search_diag1(C, X, Y, 1.0f, -1.0f, &end);
        dy = r.x; dw = r.y;
    }
    if (dx + dy > 2.0f) {
        /* coords = mad((-d.x + 0.25, d.x, d.y, -d.y - 0.25), rt.xyxy, texcoord.xyxy) */
        const float c0x = (-dx + 0.25f) * 1.0f + X, c0y = dx * 1.0f + Y, c1x = dy * 1.0f + X, c1y = (-dy - 0.25f) * 1.0f + Y;
        const v4 s0 = sample_off(C->edges, c0x, c0y, -1, 0), s1 = sample_off(C->edges, c1x, c1y, 1, 0);
        v4 c = {s0.x, s0.y, s1.x, s1.y};
        const v4 dcd = decode_diag4(c);                       /* c.yxwz = decode(c.xyzw) */
        c.y = dcd.x; c.x = dcd.y; c.w = dcd.z; c.z = dcd.w;
        v2 cc = {2.0f * c.x + c.y, 2.0f * c.z + c.w};
        if (stepf(0.9f, dz) != 0.0f) cc.x = 0.0f;             /* SMAAMovc(bool2(step(0.9, d.zw)), cc, 0) */
        if (stepf(0.9f, dw) != 0.0f) cc.y = 0.0f;
        v2 d = {dx, dy};
        const v2 a = area_diag(C, d, cc);
        weights.x += a.x; weights.y += a.y;
    }
    {
        const v2 r = search_diag2(C, X, Y, -1.0f, -1.0f, &end);
        dx = r.x; dz = r.y;
    }
    if (sample_off(C->edges, X, Y, 1, 0).x > 0.0f) {
        const v2 r = search_diag2(C, X, Y, 1.0f, 1.0f, &end);
        dy = r.x; dw = r.y;
        dy += (end.y > 0.9f) ? 1.0f : 0.0f;
    } else {
        dy = 0.0f; dw = 0.0f;
    }
    if (dx + dy > 2.0f) {
        const float c0x = -dx * 1.0f + X, c0y = -dx * 1.0f + Y, c1x = dy * 1.0f + X, c1y = dy * 1.0f + Y;
        v4 c;
        c.x = sample_off(C->edges, c0x, c0y, -1, 0).y;
        c.y = sample_off(C->edges, c0x, c0y, 0, -1).x;
        const v4 s = sample_off(C->edges, c1x, c1y, 1, 0);
        c.z = s.y; c.w = s.x;                                  /* .gr */
        v2 cc = {2.0f * c.x + c.y, 2.0f * c.z + c.w};
        if (stepf(0.9f, dz) != 0.0f) cc.x = 0.0f;
        if (stepf(0.9f, dw) != 0.0f) cc.y = 0.0f;
        v2 d = {dx, dy};
        const v2 a = area_diag(C, d, cc);
        weights.x += a.y; weights.y += a.x;                    /* .gr */
    }
    return weights;
}

/* SMAASearchLength (SMAA.h:998-1015) in the texel space of the 64 x 16 table: texcoord = scale * e + bias with
 * scale = (0.5, -2), bias = ((66 offset + 0.5) / 64, 32.5 / 16), i.e. texel (32 e.x + 66 offset, 32 - 32 e.y). */
static float search_length(const blend_ctx* C, v2 e, float offset)
{
    const float tx = 32.0f * e.x + 66.0f * offset, ty = -32.0f * e.y + 32.0f;
    return sample(C->search, tx, ty).x;
}

/* SMAASearchXLeft / XRight / YUp / YDown (SMAA.h:1020-1077); positions in texel space, one step = two texels */
static float search_x_left(const blend_ctx* C, float tx, float ty, float end)
{
    v2 e = {0.0f, 1.0f};
    while (tx > end && e.y > 0.8281f && e.x == 0.0f) {
        const v4 s = sample(C->edges, tx, ty);
        e.x = s.x; e.y = s.y;
        tx = -2.0f * 1.0f + tx;
    }
    const float offset = -(255.0f / 127.0f) * search_length(C, e, 0.0f) + 3.25f;
    return 1.0f * offset + tx;
}
static float search_x_right(const blend_ctx* C, float tx, float ty, float end)
{
    v2 e = {0.0f, 1.0f};
    while (tx < end && e.y > 0.8281f && e.x == 0.0f) {
        const v4 s = sample(C->edges, tx, ty);
        e.x = s.x; e.y = s.y;
        tx = 2.0f * 1.0f + tx;
    }
    const float offset = -(255.0f / 127.0f) * search_length(C, e, 0.5f) + 3.25f;
    return -1.0f * offset + tx;
}
static float search_y_up(const blend_ctx* C, float tx, float ty, float end)
{
    v2 e = {1.0f, 0.0f};
    while (ty > end && e.x > 0.8281f && e.y == 0.0f) {
        const v4 s = sample(C->edges, tx, ty);
        e.x = s.x; e.y = s.y;
        ty = -2.0f * 1.0f + ty;
    }
    v2 gr = {e.y, e.x};
    const float offset = -(255.0f / 127.0f) * search_length(C, gr, 0.0f) + 3.25f;
    return 1.0f * offset + ty;
}
static float search_y_down(const blend_ctx* C, float tx, float ty, float end)
{
    v2 e = {1.0f, 0.0f};
    while (ty < end && e.x > 0.8281f && e.y == 0.0f) {
        const v4 s = sample(C->edges, tx, ty);
        e.x = s.x; e.y = s.y;
        ty = 2.0f * 1.0f + ty;
    }
    v2 gr = {e.y, e.x};
    const float offset = -(255.0f / 127.0f) * search_length(C, gr, 0.5f) + 3.25f;
    return -1.0f * offset + ty;
}

/* SMAAArea (SMAA.h:1083-1095), offset = 0 */
static v2 area(const blend_ctx* C, v2 dist, float e1, float e2)
{
    const float tx = 16.0f * rintf(4.0f * e1) + dist.x, ty = 16.0f * rintf(4.0f * e2) + dist.y;   /* SMAA_AREATEX_MAX_DISTANCE */
    const v4 s = sample(C->area, tx, ty);
    v2 r = {s.x, s.y};
    return r;
}

/* SMAADetectHorizontalCornerPattern / Vertical (SMAA.h:1100-1140); (ax, ay) and (bx, by) = texcoord.xy / .zw */
static void corner_h(const blend_ctx* C, v2* weights, float ax, float ay, float bx, float by, v2 d)
{
    if (C->P->corner_rounding < 0) return;
    const float lx = stepf(d.x, d.y), ly = stepf(d.y, d.x);
    const float norm = (float)C->P->corner_rounding / 100.0f;
    float rx = (1.0f - norm) * lx, ry = (1.0f - norm) * ly;
    rx /= lx + ly;
    ry /= lx + ly;
    float fx = 1.0f, fy = 1.0f;
    fx -= rx * sample_off(C->edges, ax, ay, 0, 1).x;
    fx -= ry * sample_off(C->edges, bx, by, 1, 1).x;
    fy -= rx * sample_off(C->edges, ax, ay, 0, -2).x;
    fy -= ry * sample_off(C->edges, bx, by, 1, -2).x;
    weights->x *= saturatef(fx);
    weights->y *= saturatef(fy);
}
static void corner_v(const blend_ctx* C, v2* weights, float ax, float ay, float bx, float by, v2 d)
{
    if (C->P->corner_rounding < 0) return;
    const float lx = stepf(d.x, d.y), ly = stepf(d.y, d.x);
    const float norm = (float)C->P->corner_rounding / 100.0f;
    float rx = (1.0f - norm) * lx, ry = (1.0f - norm) * ly;
    rx /= lx + ly;
    ry /= lx + ly;
    float fx = 1.0f, fy = 1.0f;
    fx -= rx * sample_off(C->edges, ax, ay, 1, 0).y;
    fx -= ry * sample_off(C->edges, bx, by, 1, 1).y;
    fy -= rx * sample_off(C->edges, ax, ay, -2, 0).y;
    fy -= ry * sample_off(C->edges, bx, by, -2, 1).y;
    weights->x *= saturatef(fx);
    weights->y *= saturatef(fy);
}

/* SMAABlendingWeightCalculationPS (SMAA.h:1145-1243) with the varyings of ...VS (SMAA.h:656-668) */
static void blend_pixel_at(const blend_ctx* C, float X, float Y, float slack_lo, float slack_hi, int force, uint8_t out[4])
{
    const float S = (float)C->P->max_steps;
    /* offset[0] = (X - 0.25, Y - 0.125, X + 1.25, Y - 0.125); offset[1] = (X - 0.125, Y - 0.25, X - 0.125, Y + 1.25);
     * offset[2] = rt.xxyy * ((-2, 2, -2, 2) * steps) + (offset[0].xz, offset[1].yw) */
    const float o0x = X - 0.25f, o0y = Y - 0.125f, o0z = X + 1.25f, o0w = Y - 0.125f;
    const float o1x = X - 0.125f, o1y = Y - 0.25f, o1z = X - 0.125f, o1w = Y + 1.25f;
    /* slack_lo / slack_hi (diagnostic, normally 0): offset[2] is a varying of its own; a GL implementation interpolates each of its
     * components with noise that is independent of offset[0/1]'s, and after SMAA_MAX_SEARCH_STEPS steps `texcoord > end` compares two
     * numbers that are equal in exact arithmetic -- one more or one fewer search step, per direction, is the implementation's choice. */
    const float o2x = (-2.0f * S) * 1.0f + o0x - slack_lo, o2y = (2.0f * S) * 1.0f + o0z + slack_hi, o2z = (-2.0f * S) * 1.0f + o1y - slack_lo,
                o2w = (2.0f * S) * 1.0f + o1w + slack_hi;
    v4 weights = {0.0f, 0.0f, 0.0f, 0.0f};
    const v4 es = sample(C->edges, X, Y);
    v2 e = {es.x, es.y};
    /* force (diagnostic, normally 0): bit 0 / bit 1 make a pixel WITHOUT a north / west edge take the branch as a GL implementation's
     * noisy centre fetch does when it picks up 1e-5 of a neighbour's edge ("phantom edge", smaa_oracle_blend_pass_forced below) */
    if ((force & 1) && !(e.y > 0.0f)) e.y = 1e-6f;
    if ((force & 2) && !(e.x > 0.0f)) e.x = 1e-6f;
    if (e.y > 0.0f) {   /* edge at north */
        int do_hv = 1;
        if (C->P->max_steps_diag > 0) {
            const v2 dwt = diag_weights(C, X, Y, e);
            weights.x = dwt.x; weights.y = dwt.y;
            do_hv = (weights.x == -weights.y);
        }
        if (do_hv) {
            v2 d;
            const float cx = search_x_left(C, o0x, o0y, o2x);
            float cy = o1y;                                     /* texcoord.y - 0.25 (@CROSSING_OFFSET) */
            d.x = cx;
            const float e1 = sample(C->edges, cx, cy).x;
            const float cz = search_x_right(C, o0z, o0w, o2y);
            d.y = cz;
            /* d = abs(round(rt.zz * d - pixcoord.xx)): texcoord * W - (X + 0.5) = texel-space position - X */
            d.x = fabsf(rintf(d.x - X));
            d.y = fabsf(rintf(d.y - X));
            v2 sq = {sqrtf(d.x), sqrtf(d.y)};
            const float e2 = sample_off(C->edges, cz, cy, 1, 0).x;
            v2 w = area(C, sq, e1, e2);
            cy = Y;
            corner_h(C, &w, cx, cy, cz, cy, d);
            weights.x = w.x; weights.y = w.y;
        } else {
            e.x = 0.0f;                                         /* skip vertical processing */
        }
    }
    if (e.x > 0.0f) {   /* edge at west */
        v2 d;
        const float cy = search_y_up(C, o1x, o1y, o2z);
        float cx = o0x;                                         /* texcoord.x - 0.25 */
        d.x = cy;
        const float e1 = sample(C->edges, cx, cy).y;
        const float cz = search_y_down(C, o1z, o1w, o2w);
        d.y = cz;
        d.x = fabsf(rintf(d.x - Y));
        d.y = fabsf(rintf(d.y - Y));
        v2 sq = {sqrtf(d.x), sqrtf(d.y)};
        const float e2 = sample_off(C->edges, cx, cz, 0, 1).y;
        v2 w = area(C, sq, e1, e2);
        cx = X;
        corner_v(C, &w, cx, cy, cx, cz, d);
        weights.z = w.x; weights.w = w.y;
    }
    out[0] = to_unorm8(weights.x); out[1] = to_unorm8(weights.y); out[2] = to_unorm8(weights.z); out[3] = to_unorm8(weights.w);
}

static void blend_pixel(const blend_ctx* C, int x, int y, uint8_t out[4]) { blend_pixel_at(C, (float)x, (float)y, 0.0f, 0.0f, 0, out); }

/* ---- pass 3: SMAANeighborhoodBlendingPS (SMAA.h:1252-1300) -------------------------------------------------- */
static void neighborhood_pixel(const tex_t* color, const tex_t* blend, int x, int y, uint8_t out[4])
{
    const float X = (float)x, Y = (float)y;
    v4 a;
    a.x = sample(blend, X + 1.0f, Y).w;          /* right  (offset.xy) */
    a.y = sample(blend, X, Y + 1.0f).y;          /* top    (offset.zw) */
    {
        const v4 s = sample(blend, X, Y);
        a.w = s.x; a.z = s.z;                    /* a.wz = .xz */
    }
    v4 c;
    if (a.x * 1.0f + a.y * 1.0f + a.z * 1.0f + a.w * 1.0f < 1e-5f) {
        c = sample(color, X, Y);
    } else {
        const int h = maxf(a.x, a.z) > maxf(a.y, a.w);
        float bx = 0.0f, by = a.y, bz = 0.0f, bw = a.w;
        float wx = a.y, wy = a.w;
        if (h) { bx = a.x; by = 0.0f; bz = a.z; bw = 0.0f; wx = a.x; wy = a.z; }
        const float sum = wx * 1.0f + wy * 1.0f;
        wx /= sum;
        wy /= sum;
        /* blendingCoord = mad(blendingOffset, (rt.xy, -rt.xy), texcoord.xyxy) */
        const float c0x = bx * 1.0f + X, c0y = by * 1.0f + Y, c1x = bz * -1.0f + X, c1y = bw * -1.0f + Y;
        const v4 s0 = sample(color, c0x, c0y), s1 = sample(color, c1x, c1y);
        c.x = wx * s0.x; c.y = wx * s0.y; c.z = wx * s0.z; c.w = wx * s0.w;
        c.x += wy * s1.x; c.y += wy * s1.y; c.z += wy * s1.z; c.w += wy * s1.w;
    }
    out[0] = to_unorm8(c.x); out[1] = to_unorm8(c.y); out[2] = to_unorm8(c.z); out[3] = to_unorm8(c.w);
}

/* ---- entry points --------------------------------------------------------------------------------------------
 * color: W*H RGBA8, row 0 = bottom row of the frame (= texture row 0). area: 160 x 560 x 2, search: 64 x 16 x 1.
 * edges_out: W*H*2, blend_out / screen_out: W*H*4 (any may be NULL except that later passes need the earlier ones:
 * the function keeps its own intermediates). preset: 0 LOW .. 3 ULTRA. Returns 0, or -1 on bad arguments. */
int smaa_oracle_run(const uint8_t* color, int w, int h, int preset, const uint8_t* area_tex, const uint8_t* search_tex,
                    uint8_t* edges_out, uint8_t* blend_out, uint8_t* screen_out)
{
    if (!color || !area_tex || !search_tex || w <= 0 || h <= 0 || preset < 0 || preset > 3) return -1;
    const preset_t* P = &k_presets[preset];
    const size_t n = (size_t)w * (size_t)h;
    uint8_t* edges = edges_out ? edges_out : (uint8_t*)malloc(n * 2);
    uint8_t* blend = blend_out ? blend_out : (uint8_t*)malloc(n * 4);
    if (!edges || !blend) return -1;
    const tex_t tc = {w, h, 4, color}, te = {w, h, 2, edges}, tb = {w, h, 4, blend};
    const tex_t ta = {160, 560, 2, area_tex}, ts = {64, 16, 1, search_tex};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) edge_pixel(&tc, P, x, y, edges + ((size_t)y * w + x) * 2);
    const blend_ctx C = {&te, &ta, &ts, P};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) blend_pixel(&C, x, y, blend + ((size_t)y * w + x) * 4);
    if (screen_out) {
#pragma omp parallel for schedule(dynamic, 4)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) neighborhood_pixel(&tc, &tb, x, y, screen_out + ((size_t)y * w + x) * 4);
    }
    if (!edges_out) free(edges);
    if (!blend_out) free(blend);
    return 0;
}

/* single passes on caller-supplied inputs (tests feed the REFERENCE's intermediate textures to isolate a pass) */
int smaa_oracle_blend_pass(const uint8_t* edges, int w, int h, int preset, const uint8_t* area_tex, const uint8_t* search_tex, uint8_t* blend_out)
{
    if (!edges || !area_tex || !search_tex || !blend_out || w <= 0 || h <= 0 || preset < 0 || preset > 3) return -1;
    const tex_t te = {w, h, 2, edges}, ta = {160, 560, 2, area_tex}, ts = {64, 16, 1, search_tex};
    const blend_ctx C = {&te, &ta, &ts, &k_presets[preset]};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) blend_pixel(&C, x, y, blend_out + ((size_t)y * w + x) * 4);
    return 0;
}
int smaa_oracle_neighborhood_pass(const uint8_t* color, const uint8_t* blend, int w, int h, uint8_t* screen_out)
{
    if (!color || !blend || !screen_out || w <= 0 || h <= 0) return -1;
    const tex_t tc = {w, h, 4, color}, tb = {w, h, 4, blend};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) neighborhood_pixel(&tc, &tb, x, y, screen_out + ((size_t)y * w + x) * 4);
    return 0;
}

/* Diagnostic for the pin against a real GL implementation (tests/test_smaa_oracle.py): pass 2 with the pixel's position
 * displaced by (jx, jy) texels. A GL implementation interpolates the texture-coordinate varyings and computes filter weights
 * with finite precision, so the shader sees the pixel centre ~1e-7..1e-3 texels off; where a branch of the shader tests a
 * bilinear fetch against exactly 0 (SMAA.h:1155,1205: e.g > 0.0, e.r > 0.0) that noise decides. A pixel whose result changes
 * under such a displacement is one where the reference's own output is implementation-defined. */
int smaa_oracle_blend_pass_jitter(const uint8_t* edges, int w, int h, int preset, const uint8_t* area_tex, const uint8_t* search_tex, float jx,
                                  float jy, float slack_lo, float slack_hi, uint8_t* blend_out)
{
    if (!edges || !area_tex || !search_tex || !blend_out || w <= 0 || h <= 0 || preset < 0 || preset > 3) return -1;
    const tex_t te = {w, h, 2, edges}, ta = {160, 560, 2, area_tex}, ts = {64, 16, 1, search_tex};
    const blend_ctx C = {&te, &ta, &ts, &k_presets[preset]};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) blend_pixel_at(&C, (float)x + jx, (float)y + jy, slack_lo, slack_hi, 0, blend_out + ((size_t)y * w + x) * 4);
    return 0;
}

/* Diagnostic for the same pin: pass 2 at EXACT positions, but a pixel that has no north (force bit 0) / west (bit 1) edge of its own while
 * one of its eight neighbours has one takes the `e.g > 0.0` / `e.r > 0.0` branch anyway (SMAA.h:1155,1205) -- what llvmpipe's centre fetch
 * does when its bilinear weights are 1e-5 off. The value such a "phantom" pixel gets is then fully determined (searches, area look-up and
 * corner rounding run on the real edge texture), so the classifier can demand it instead of excusing the pixel. */
int smaa_oracle_blend_pass_forced(const uint8_t* edges, int w, int h, int preset, const uint8_t* area_tex, const uint8_t* search_tex, int force,
                                  float slack_lo, float slack_hi, uint8_t* blend_out)
{
    if (!edges || !area_tex || !search_tex || !blend_out || w <= 0 || h <= 0 || preset < 0 || preset > 3) return -1;
    const tex_t te = {w, h, 2, edges}, ta = {160, 560, 2, area_tex}, ts = {64, 16, 1, search_tex};
    const blend_ctx C = {&te, &ta, &ts, &k_presets[preset]};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int nb_n = 0, nb_w = 0;
            for (int dy = -1; dy <= 1; dy++)
                for (int dx = -1; dx <= 1; dx++) {
                    const int xx = clampi(x + dx, 0, w - 1), yy = clampi(y + dy, 0, h - 1);
                    nb_w |= edges[((size_t)yy * w + xx) * 2 + 0] != 0;
                    nb_n |= edges[((size_t)yy * w + xx) * 2 + 1] != 0;
                }
            const int f = ((force & 1) && nb_n ? 1 : 0) | ((force & 2) && nb_w ? 2 : 0);
            blend_pixel_at(&C, (float)x, (float)y, slack_lo, slack_hi, f, blend_out + ((size_t)y * w + x) * 4);
        }
    return 0;
}
