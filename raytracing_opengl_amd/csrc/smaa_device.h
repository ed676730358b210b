// smaa_device.h -- per-pixel arithmetic of the SMAA post-process, shared by the HIP kernels
// (smaa_kernel.hip) and the host build used by the CPU-side logic tests (tests/host_harness).
//
// Replaces the three programs the reference runs after the tracer (src/GLWrapper.cpp:173-204, assembled by
// src/SMAA_Builder.h:17-113 from assets/shaders/SMAA.h): luma edge detection (SMAA.h:689-741), blending-weight calculation
// (SMAA.h:835-1243) and neighbourhood blending (SMAA.h:1252-1300); presets SMAA.h:304-324; SMAA 1x, no predication, no
// reprojection (what SMAA_Builder compiles in).
//
// Arithmetic contract ("SMAA"): float32, no contraction, IEEE divide / sqrt, round = round-half-even; all textures
// are 8-bit UNORM, LINEAR, CLAMP_TO_EDGE; positions are carried in TEXEL space, where the pixel the shader runs for sits at
// exactly (x, y) and every offset SMAA adds is a dyadic fraction of a texel -- the values a sampler with exact varying
// interpolation and unlimited sub-texel precision returns. Because the positions are exact, most fetches here are known at
// compile time to fall on a texel centre or on a texel row / column, and are written as one- or two-tap fetches; that is the
// SAME value as the general four-tap form (the dropped taps have weight exactly 0, and x + 0 = x), not an approximation.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define SM_HD __host__ __device__ __forceinline__
#define SM_HDM __host__ __device__ __forceinline__   // member functions
#else
#define SM_HD static inline
#define SM_HDM inline
#endif

#ifndef SMAA_DIAG_BATCH
#define SMAA_DIAG_BATCH 4   /* steps fetched ahead per round of the in-step diagonal searches */
#endif
#ifndef SMAA_DIAG_IN_STEP
#define SMAA_DIAG_IN_STEP 1
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define SM_ANY(x) (__any((x)) != 0)     /* some lane of the wave */
#else
#define SM_ANY(x) (x)
#endif
#ifndef SMAA_PH
#define SMAA_PH(k)      /* diagnostic build only (smaa_kernel.hip, -DSMAA_PHASE_TIMES): a timestamp per wave at the convergent points of weights() */
#endif
namespace smaa {

struct Preset {            // SMAA.h:304-324
    float threshold;       // SMAA_THRESHOLD
    int max_steps;         // SMAA_MAX_SEARCH_STEPS
    int max_steps_diag;    // SMAA_MAX_SEARCH_STEPS_DIAG, 0 = SMAA_DISABLE_DIAG_DETECTION
    int corner_rounding;   // SMAA_CORNER_ROUNDING, < 0 = SMAA_DISABLE_CORNER_DETECTION
};
SM_HD Preset preset_of(int id)
{
    switch (id) {
        case 0: return Preset{0.15f, 4, 0, -1};
        case 1: return Preset{0.1f, 8, 0, -1};
        case 2: return Preset{0.1f, 16, 8, 25};
        default: return Preset{0.05f, 32, 16, 25};
    }
}

enum { AREA_W = 160, AREA_H = 560, SEARCH_W = 64, SEARCH_H = 16 };   // SMAA.h:519-522, AreaTex.h / SearchTex.h sizes

struct Views {             // device (or host) pointers of one frame's textures
    int w, h;
    const uint32_t* color;   // RGBA8, R in bits 0..7
    const uint16_t* edges;   // RG8,   R in bits 0..7
    const uint32_t* blend;   // RGBA8
    const uint16_t* area;    // RG8, AREA_W x AREA_H
    const uint8_t* search;   // R8,  SEARCH_W x SEARCH_H
};

struct F2 { float x, y; };
struct F4 { float x, y, z, w; };

// byte / 255.0f without the IEEE divide: one multiply by RN(1/255) and an fma correction step give the correctly rounded quotient
// for every one of the 256 inputs (checked exhaustively: tests/test_smaa_host.py on the host, rtx_selftest on the device).
SM_HD float unorm8(uint32_t b)
{
    const float x = (float)b;
    const float rcp = 0.0039215688593685626983642578125f;   // RN(1/255)
    float q = x * rcp;
    const float r = __builtin_fmaf(-q, 255.0f, x);
    q = __builtin_fmaf(r, rcp, q);
    return q;
}
SM_HD uint32_t to_unorm8(float v)
{
    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
    if (!(v == v)) v = 0.0f;
    return (uint32_t)(v * 255.0f + 0.5f);
}
SM_HD float step_(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
SM_HD float max_(float a, float b) { return a < b ? b : a; }
SM_HD float sat_(float v) { return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v); }
SM_HD int clampi(int v, int hi) { return v < 0 ? 0 : (v > hi ? hi : v); }

// ---- samplers -----------------------------------------------------------------------------------------------
// Where the RG8 edge texture of pass 1 is read from: raw(i, j) = the RG8 texel of an IN-RANGE texel (the samplers clamp first), R in
// bits 0..7, G in bits 8..15.
//   TexEdges   the texture itself (what the reference's fboTexEdge is): 2 bytes per texel. The host reference build, and the form the other
//              source is checked against.
//   PlaneTex   the ROW bit plane the dense HIP kernel writes every frame (SearchPlanes below: 32 pixels per 64-bit word, bit 2k = red,
//              bit 2k + 1 = green of pixel k): the same texel values, 0 or 255. The HIP weight kernel reads the edges from here, so the
//              RG8 texture need not exist on the device at all -- no scattered 2-byte texel stores in the dense pass, and nothing to
//              clear before the next frame (the plane is rewritten densely; round 2 kept the texture zero outside the current edge list
//              with a kernel of its own).
struct TexEdges {
    const uint16_t* t;
    int w;
    SM_HDM uint32_t raw(int i, int j) const { return t[(size_t)j * w + i]; }
};
struct PlaneTex {
    const uint64_t* rows;
    int pw;                  // 64-bit words per row
    // (read as 32-bit halves -- 16 pixels each, little-endian: half h of word q is dword 2q + h -- so that the extraction is a 32-bit shift)
    SM_HDM uint32_t two(int i, int j) const { return (reinterpret_cast<const uint32_t*>(rows)[((size_t)j * pw << 1) + (i >> 4)] >> (2 * (i & 15))) & 3u; }
    SM_HDM uint32_t raw(int i, int j) const
    {
        const uint32_t t = two(i, j);
        return ((t & 1u) ? 0x00ffu : 0u) | ((t & 2u) ? 0xff00u : 0u);
    }
    SM_HDM bool any(int i, int j) const { return two(i, j) != 0u; }   // the pixel has an edge
};
// General LINEAR + CLAMP_TO_EDGE fetch of the RG8 edge texture at texel-space (tx, ty) plus an integer texel offset.
template <class E>
SM_HD F2 sample_edges(const E& src, int w, int h, float tx, float ty, int ox = 0, int oy = 0)
{
    const float fx = floorf(tx), fy = floorf(ty);
    const float a = tx - fx, b = ty - fy;
    const int i0 = clampi((int)fx + ox, w - 1), i1 = clampi((int)fx + ox + 1, w - 1);
    const int j0 = clampi((int)fy + oy, h - 1), j1 = clampi((int)fy + oy + 1, h - 1);
    // A tap whose weight is exactly 0 is not fetched: 0 * texel is +0 for every texel and x + 0 = x, so the sum is the same bits. Most of
    // SMAA's fetches sit on a texel row, column or centre (a == 0 and / or b == 0).
    const uint32_t p00 = src.raw(i0, j0);
    const uint32_t p10 = a != 0.0f ? src.raw(i1, j0) : 0u;
    const uint32_t p01 = b != 0.0f ? src.raw(i0, j1) : 0u;
    const uint32_t p11 = (a != 0.0f && b != 0.0f) ? src.raw(i1, j1) : 0u;
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    F2 r;
    r.x = w00 * unorm8(p00 & 255u) + w10 * unorm8(p10 & 255u) + w01 * unorm8(p01 & 255u) + w11 * unorm8(p11 & 255u);
    r.y = w00 * unorm8(p00 >> 8) + w10 * unorm8(p10 >> 8) + w01 * unorm8(p01 >> 8) + w11 * unorm8(p11 >> 8);
    return r;
}
// The same for any RG8 texture in memory (the area table).
SM_HD F2 sample_rg(const uint16_t* t, int w, int h, float tx, float ty, int ox = 0, int oy = 0)
{
    return sample_edges(TexEdges{t, w}, w, h, tx, ty, ox, oy);
}
// Texel-centre fetch (integer position): one tap.
SM_HD F2 texel_rg(const uint16_t* t, int w, int h, int i, int j)
{
    const uint32_t p = t[(size_t)clampi(j, h - 1) * w + clampi(i, w - 1)];
    F2 r;
    r.x = unorm8(p & 255u);
    r.y = unorm8(p >> 8);
    return r;
}
SM_HD F4 unpack4(uint32_t p) { F4 r; r.x = unorm8(p & 255u); r.y = unorm8((p >> 8) & 255u); r.z = unorm8((p >> 16) & 255u); r.w = unorm8(p >> 24); return r; }
SM_HD F4 sample_rgba(const uint32_t* t, int w, int h, float tx, float ty)
{
    const float fx = floorf(tx), fy = floorf(ty);
    const float a = tx - fx, b = ty - fy;
    const int i0 = clampi((int)fx, w - 1), i1 = clampi((int)fx + 1, w - 1);
    const int j0 = clampi((int)fy, h - 1), j1 = clampi((int)fy + 1, h - 1);
    const F4 t00 = unpack4(t[(size_t)j0 * w + i0]), t10 = unpack4(a != 0.0f ? t[(size_t)j0 * w + i1] : 0u), t01 = unpack4(b != 0.0f ? t[(size_t)j1 * w + i0] : 0u),
             t11 = unpack4((a != 0.0f && b != 0.0f) ? t[(size_t)j1 * w + i1] : 0u);   // zero-weight taps are not fetched (see sample_rg)
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    F4 r;
    r.x = w00 * t00.x + w10 * t10.x + w01 * t01.x + w11 * t11.x;
    r.y = w00 * t00.y + w10 * t10.y + w01 * t01.y + w11 * t11.y;
    r.z = w00 * t00.z + w10 * t10.z + w01 * t01.z + w11 * t11.z;
    r.w = w00 * t00.w + w10 * t10.w + w01 * t01.w + w11 * t11.w;
    return r;
}
SM_HD float sample_r8(const uint8_t* t, int w, int h, float tx, float ty)
{
    const float fx = floorf(tx), fy = floorf(ty);
    const float a = tx - fx, b = ty - fy;
    const int i0 = clampi((int)fx, w - 1), i1 = clampi((int)fx + 1, w - 1);
    const int j0 = clampi((int)fy, h - 1), j1 = clampi((int)fy + 1, h - 1);
    const float w00 = (1.0f - a) * (1.0f - b), w10 = a * (1.0f - b), w01 = (1.0f - a) * b, w11 = a * b;
    const uint32_t p10 = a != 0.0f ? t[j0 * w + i1] : 0u, p01 = b != 0.0f ? t[j1 * w + i0] : 0u, p11 = (a != 0.0f && b != 0.0f) ? t[j1 * w + i1] : 0u;
    return w00 * unorm8(t[j0 * w + i0]) + w10 * unorm8(p10) + w01 * unorm8(p01) + w11 * unorm8(p11);
}

// ---- pass 1: luma edges (SMAA.h:689-741) ----------------------------------------------------------------------
SM_HD float luma_of(uint32_t rgba)
{
    return unorm8(rgba & 255u) * 0.2126f + unorm8((rgba >> 8) & 255u) * 0.7152f + unorm8((rgba >> 16) & 255u) * 0.0722f;
}
// The edge decision from the six luma deltas of SMAA.h:709-737: d = |L - left|, |L - top| (the pixel's own two candidate edges), the
// deltas to the right / bottom neighbours and the left / top neighbours' own deltas (left-left, top-top). The shader multiplies step()
// results; with every factor in {0, 1} that product is the conjunction below -- the same decisions, a third of the instructions. Note
// that each delta is ALSO some neighbour's own delta (|a - b| = |b - a| exactly): the dense kernel computes one horizontal and one
// vertical delta per pixel and hands them along. Returns bit 0 = red (left edge), bit 1 = green (top edge).
SM_HD uint32_t edge_bits(float threshold, float dx, float dy, float dx_right, float dy_bottom, float dx_left, float dy_top)
{
    bool ex = !(dx < threshold), ey = !(dy < threshold);                       // step(threshold, delta)
    if (!(ex || ey)) return 0u;                                                // discard
    const float fin = fmaxf(fmaxf(fmaxf(dx, dx_right), dx_left), fmaxf(fmaxf(dy, dy_bottom), dy_top));
    ex = ex && !(2.0f * dx < fin);                                             // step(finalDelta, 2 * delta): local contrast adaptation
    ey = ey && !(2.0f * dy < fin);
    return (ex ? 1u : 0u) | (ey ? 2u : 0u);
}
// Lumas: L centre, Ll / Lll one / two texels to the left, Lr right, Lt / Ltt one / two rows below in memory ("top" in the shader's
// texture space: offset (0,-1)), Lb the row above. Returns the RG8 texel (0 = discarded fragment).
SM_HD uint32_t edge_from_lumas(float threshold, float L, float Ll, float Lt, float Lr, float Lb, float Lll, float Ltt)
{
    const uint32_t e = edge_bits(threshold, fabsf(L - Ll), fabsf(L - Lt), fabsf(L - Lr), fabsf(L - Lb), fabsf(Ll - Lll), fabsf(Lt - Ltt));
    return ((e & 1u) ? 0x00ffu : 0u) | ((e & 2u) ? 0xff00u : 0u);
}

// ---- the orthogonal searches on bit planes ---------------------------------------------------------------------------------------
// SMAASearchXLeft / XRight / YUp / YDown (SMAA.h:1020-1077) walk a line of edges two texels at a time; every step is a 4-tap bilinear fetch
// and the walk of a long edge is up to max_steps dependent fetches in each of four directions -- most of pass 2's time. The edge texture
// holds only 0 and 255, so the dense kernel also writes it as two BIT PLANES (2 bits per pixel: bit 2k = red / left edge, bit 2k + 1 = green
// / top edge of the k-th pixel of a run) --
//   rows: plane_words(w) 64-bit words per row, 32 consecutive pixels of the ROW per word;
//   cols: one 16-bit word per column and block of 8 rows, [y >> 3][x]: 8 consecutive pixels of the COLUMN per word --
// and the NUMBER OF STEPS a search takes is computed from whole words: six loads for a horizontal search, 24 short ones for a vertical one,
// all independent, then a few dozen integer instructions, whatever the length of the edge. The search's float arithmetic is not touched:
// the loop's last fetch -- the one whose value goes into SMAASearchLength -- and everything after it run as before on the texture.
//
// Why the count is exact. At step k the loop fetches e = bilinear(edges) at (tx0 -/+ 2k, ty) and goes on while e.g > 0.8281 and e.r == 0
// (horizontal; vertical: r and g swapped). Positions are exact in texel space: the fractions are (0.75 | 0.25, 0.875) resp. (0.875,
// 0.75 | 0.25), the four weights are the dyadic products of those, the texels are 0 or 1, so the sums are exact. e.g > 0.8281 holds iff
// BOTH taps of the upper row (horizontal; weight 0.875 together) have their green bit -- any other combination sums to at most 0.78125
// --, e.r == 0 iff none of the four taps has its red bit. With ok[c] = G[y][c] & ~R[y][c] & ~R[y - 1][c] the step at column c goes on
// iff ok[c] & ok[c + 1]. The loop consumes n = min(max_steps, f + 1) fetches, f = the first step that fails. (Checked against the
// per-step loop on every edge pixel of the test patterns: tests/test_smaa_host.py; the host build can run both and compare.)
// Only for pixels whose whole search window lies inside the frame (no index is clamped); the others take the per-step loop.
struct SearchPlanes {
    const uint64_t* rows;    // h x plane_words(w); nullptr = no planes (host reference build): every search takes the per-step loop
    const uint16_t* cols;    // ((h + 7) / 8) x w
    int w, h;
    SM_HDM static int plane_words(int w) { return (w + 31) >> 5; }
    // first failing step among offsets o0, o0 +/- 2, ... (S of them) of the 96-position string `pair` (bit 2o = step at offset o goes on)
    SM_HDM static int first_fail(const uint64_t pair[3], int o0, int S, bool down)
    {
        const uint64_t M = 0x5555555555555555ull;
        const uint64_t par = (o0 & 1) ? 0x4444444444444444ull : 0x1111111111111111ull;
        const int o_lo = down ? o0 - 2 * (S - 1) : o0, o_hi = down ? o0 : o0 + 2 * (S - 1);
        uint64_t fail[3];
        for (int k = 0; k < 3; k++) {
            const int lo = o_lo - 32 * k, hi = o_hi - 32 * k;
            const uint64_t from = lo <= 0 ? ~0ull : (lo >= 32 ? 0ull : (~0ull << (2 * lo)));
            const uint64_t to = hi < 0 ? 0ull : (hi >= 31 ? ~0ull : ((1ull << (2 * hi + 2)) - 1ull));
            fail[k] = ~pair[k] & M & par & from & to;
        }
        if (down) {   // steps walk towards smaller offsets: the first failing one is the HIGHEST
            for (int k = 2; k >= 0; k--)
                if (fail[k] != 0ull) return (o0 - (32 * k + (63 - __builtin_clzll(fail[k])) / 2)) / 2;
        } else {
            for (int k = 0; k < 3; k++)
                if (fail[k] != 0ull) return ((32 * k + __builtin_ctzll(fail[k]) / 2) - o0) / 2;
        }
        return S;
    }
    SM_HDM static void pair_of(const uint64_t ok[3], uint64_t pair[3])
    {
        pair[0] = ok[0] & ((ok[0] >> 2) | (ok[1] << 62));
        pair[1] = ok[1] & ((ok[1] >> 2) | (ok[2] << 62));
        pair[2] = ok[2] & (ok[2] >> 2);
    }
    // fetches the loop of search_x consumes for pixel (x, y): dir < 0 left, > 0 right; -1 = no planes.
    // Round 3: windows that reach beyond the frame are counted too. The samplers clamp every tap's index on its own (CLAMP_TO_EDGE), so
    // beyond the frame the string of per-position conditions simply repeats its border position -- ok[c] = ok[0] for c < 0, ok[w - 1] for
    // c > w - 1, and row y - 1 of row 0 is row 0 -- and "the step at c goes on iff ok[c] & ok[c + 1]" holds as before. (Until then such
    // pixels -- everything within 2 max_steps + 2 of the frame's border -- walked their edges step by step: up to eight dependent round
    // trips per search, and the few waves that held them were the last to finish: 28 us for the slowest wave of the traced 4K frame
    // against 18 for the 95th percentile, tools/smaa_phase_times.py.)
    // (All loads of a count are issued first, from clamped indices, and the repetition is applied with selects afterwards: a branch between
    // the loads makes the compiler wait for each before the next is issued -- 24 dependent latencies instead of one, measured.)
    SM_HDM static uint64_t spread(uint64_t bit) { return (0ull - (bit & 1ull)) & 0x5555555555555555ull; }
    SM_HDM int count_x(int x, int y, int S, bool left) const
    {
        if (rows == nullptr) return -1;
        const int c_lo = left ? x - 1 - 2 * (S - 1) : x + 1;
        const int pw = plane_words(w), w0 = c_lo >> 5, qlast = (w - 1) >> 5, tail = (w - 1) & 31;     // (>> of a negative int: arithmetic)
        const int ym = y > 0 ? y - 1 : 0;
        const uint64_t M = 0x5555555555555555ull;
        const uint64_t beyond = tail != 31 ? M & (~0ull << (2 * ((tail + 1) & 31))) : 0ull;
        uint64_t q1[3], q0[3], ok[3], pair[3];
        for (int k = 0; k < 3; k++) {
            const int qi = w0 + k, q = qi < 0 ? 0 : (qi > qlast ? qlast : qi);
            q1[k] = rows[(size_t)y * pw + q];
            q0[k] = rows[(size_t)ym * pw + q];
        }
        const bool over = c_lo < 0 || c_lo + 2 * S + 1 > w - 1;     // the window hangs over the frame: rare, decided per wave after the loads
        if (SM_ANY(over)) {
            for (int k = 0; k < 3; k++) {
                const int qi = w0 + k;
                const uint64_t o = (q1[k] >> 1) & ~q1[k] & ~q0[k] & M;
                const uint64_t first = spread(o), last = spread(o >> (2 * tail));
                ok[k] = qi < 0 ? first : (qi > qlast ? last : (qi == qlast ? (o & ~beyond) | (last & beyond) : o));
            }
        } else {
            for (int k = 0; k < 3; k++) ok[k] = (q1[k] >> 1) & ~q1[k] & ~q0[k] & M;
        }
        pair_of(ok, pair);
        const int f = first_fail(pair, (left ? x - 1 : x + 1) - 32 * w0, S, left);
        return f + 1 < S ? f + 1 : S;
    }
    // the same for search_y: up = towards smaller y. A word of the column plane holds 8 rows: the replication is per 16-bit block.
    SM_HDM int count_y(int x, int y, int S, bool up) const
    {
        if (rows == nullptr) return -1;
        const int r_lo = up ? y - 1 - 2 * (S - 1) : y + 1;
        const int nb = (h + 7) >> 3, b0 = r_lo >> 3, btail = (h - 1) & 7;
        const int xm = x > 0 ? x - 1 : 0;
        const uint32_t beyond = btail != 7 ? 0x5555u & (0xffffu << (2 * ((btail + 1) & 7))) : 0u;
        uint32_t qa[12], qb[12];
        uint64_t ok[3], pair[3];
        for (int j = 0; j < 12; j++) {
            const int bi = b0 + j, blk = bi < 0 ? 0 : (bi > nb - 1 ? nb - 1 : bi);
            qa[j] = cols[(size_t)blk * w + xm];
            qb[j] = cols[(size_t)blk * w + x];
        }
        const bool over = r_lo < 0 || r_lo + 2 * S + 1 > h - 1;
        if (SM_ANY(over)) {
            for (int k = 0; k < 3; k++) {
                uint64_t o = 0ull;
                for (int m = 0; m < 4; m++) {
                    const int j = 4 * k + m, bi = b0 + j;
                    const uint32_t ob = qb[j] & ~(qa[j] >> 1) & ~(qb[j] >> 1) & 0x5555u;     // red of column x, no green in either column
                    const uint32_t first = (0u - (ob & 1u)) & 0x5555u, last = (0u - ((ob >> (2 * btail)) & 1u)) & 0x5555u;
                    const uint32_t r = bi < 0 ? first : (bi > nb - 1 ? last : (bi == nb - 1 ? (ob & ~beyond) | (last & beyond) : ob));
                    o |= (uint64_t)r << (16 * m);
                }
                ok[k] = o;
            }
        } else {
            const uint64_t M = 0x5555555555555555ull;
            for (int k = 0; k < 3; k++) {
                uint64_t a = 0ull, b = 0ull;
                for (int m = 0; m < 4; m++) { a |= (uint64_t)qa[4 * k + m] << (16 * m); b |= (uint64_t)qb[4 * k + m] << (16 * m); }
                ok[k] = b & ~(a >> 1) & ~(b >> 1) & M;
            }
        }
        pair_of(ok, pair);
        const int f = first_fail(pair, (up ? y - 1 : y + 1) - 8 * b0, S, up);
        return f + 1 < S ? f + 1 : S;
    }
};

// ---- pass 2: blending weights (SMAA.h:835-1243) --------------------------------------------------------------
template <class E>
struct BlendT {
    const Views& V;
    const Preset& P;
    const SearchPlanes& planes;   // bit planes of the edge texture for the orthogonal searches (rows == nullptr: per-step loops only)
    const E& src;                 // where single edge texels are fetched from (TexEdges / PlaneTex)

    SM_HDM F2 edges_at(float tx, float ty, int ox = 0, int oy = 0) const { return sample_edges(src, V.w, V.h, tx, ty, ox, oy); }

    SM_HDM static float decode1(float r) { return rintf(r * fabsf(5.0f * r - 3.75f)); }   // SMAADecodeDiagBilinearAccess, red channel

    // The search loops below fetch SEARCH_BATCH steps ahead before testing the first of them. A step's position does not depend on what the
    // previous step fetched, only whether the step is taken does, so the batch is evaluated in order from registers and abandoned at the
    // first failing test: the same values, tests and results as one fetch per iteration, with the chain of dependent memory latencies cut
    // by the batch size (a 32-step search of a long edge is 8 round trips to L2 instead of 32). Fetches past the stopping step are wasted
    // bandwidth only (positions are clamped, any texel is readable).
    enum { SEARCH_BATCH = 4 };

    // SMAASearchDiag1 / 2 (SMAA.h:861-892): steps of one texel along (dirx, diry); returns (steps, last weight); e = last edges
    SM_HDM F2 search_diag1(float tx, float ty, float dirx, float diry, F2& e) const
    {
        float n = -1.0f, wgt = 1.0f;
        const float last = (float)(P.max_steps_diag - 1);
        while (n < last && wgt > 0.9f) {
            F2 s[SEARCH_BATCH];
            float px = tx, py = ty;
            for (int k = 0; k < SEARCH_BATCH; k++) {
                px = 1.0f * dirx + px;
                py = 1.0f * diry + py;
                s[k] = edges_at(px, py);
            }
            for (int k = 0; k < SEARCH_BATCH && n < last && wgt > 0.9f; k++) {
                tx = 1.0f * dirx + tx;
                ty = 1.0f * diry + ty;
                n = 1.0f * 1.0f + n;
                e = s[k];
                wgt = e.x * 0.5f + e.y * 0.5f;
            }
        }
        return F2{n, wgt};
    }
    SM_HDM F2 search_diag2(float tx, float ty, float dirx, float diry, F2& e) const
    {
        float n = -1.0f, wgt = 1.0f;
        const float last = (float)(P.max_steps_diag - 1);
        tx += 0.25f;
        while (n < last && wgt > 0.9f) {
            F2 s[SEARCH_BATCH];
            float px = tx, py = ty;
            for (int k = 0; k < SEARCH_BATCH; k++) {
                px = 1.0f * dirx + px;
                py = 1.0f * diry + py;
                s[k] = edges_at(px, py);
            }
            for (int k = 0; k < SEARCH_BATCH && n < last && wgt > 0.9f; k++) {
                tx = 1.0f * dirx + tx;
                ty = 1.0f * diry + ty;
                n = 1.0f * 1.0f + n;
                e.x = decode1(s[k].x);
                e.y = rintf(s[k].y);
                wgt = e.x * 0.5f + e.y * 0.5f;
            }
        }
        return F2{n, wgt};
    }
    SM_HDM F2 area_diag(float d1, float d2, float e1, float e2) const   // SMAAAreaDiag, offset 0: diagonal half starts at column 80
    {
        return sample_rg(V.area, AREA_W, AREA_H, (20.0f * e1 + d1) + 80.0f, 20.0f * e2 + d2);
    }
    // SMAACalculateDiagWeights (SMAA.h:918-985) in two steps: the four diagonal searches -- independent of each other -- and the rest.
    // diag_search(k): k = 0 / 1 the first pair (towards (-1,+1) / (+1,-1)), k = 2 / 3 the second ((-1,-1) / (+1,+1)); returns (distance,
    // last weight) as the shader's d.x/d.z, d.y/d.w. The HIP kernel gives each search to its own lane (smaa_kernel.hip).
    SM_HDM F2 diag_search(int k, float X, float Y, F2 e) const
    {
        F2 end{0.0f, 0.0f};
        if (k == 0) {
            if (!(e.x > 0.0f)) return F2{0.0f, 0.0f};
            const F2 r = search_diag1(X, Y, -1.0f, 1.0f, end);
            return F2{r.x + ((end.y > 0.9f) ? 1.0f : 0.0f), r.y};
        }
        if (k == 1) return search_diag1(X, Y, 1.0f, -1.0f, end);
        if (k == 2) return search_diag2(X, Y, -1.0f, -1.0f, end);
        if (!(edges_at(X, Y, 1, 0).x > 0.0f)) return F2{0.0f, 0.0f};
        const F2 r = search_diag2(X, Y, 1.0f, 1.0f, end);
        return F2{r.x + ((end.y > 0.9f) ? 1.0f : 0.0f), r.y};
    }
    // The four diagonal searches of a pixel IN STEP (round 3): every round issues the next SEARCH_BATCH fetches of all searches that are
    // still running -- raw texel loads only, one per step for the first pair (their positions are texel centres), two for the second
    // (x + 0.25: weights 0.75 / 0.25, the row weights 1 / 0) -- and only then consumes them, each search in its own order with its own
    // stopping test: the same fetches, values and results as diag_search(0..3) one after the other, but the wave waits for ONE memory round
    // trip per round instead of one per search and round (a wave is as slow as its slowest lane: with four searches of up to four rounds
    // one after the other, the waves that held a long diagonal were the last of the kernel to finish -- tools/smaa_phase_times.py).
    // No branch sits between a load and the next load: the compiler waits for a load where its value is first used.
    // `pairs`: bit 0 = the first pair of searches (0, 1), bit 1 = the second (2, 3); a search that is left out issues no fetch and its
    // out[] entry is not meaningful (round 4: the device kernel gives each pair to a wave of its own).
    SM_HDM void diag_searches(float X, float Y, F2 e, F2 out[4], unsigned pairs = 3u) const
    {
        enum { DIAG_BATCH = SMAA_DIAG_BATCH };
        const float last = (float)(P.max_steps_diag - 1);
        const float dx[4] = {-1.0f, 1.0f, -1.0f, 1.0f}, dy[4] = {1.0f, -1.0f, -1.0f, 1.0f};
        float tx[4] = {X, X, X + 0.25f, X + 0.25f}, ty[4] = {Y, Y, Y, Y}, n[4] = {-1.0f, -1.0f, -1.0f, -1.0f}, wgt[4] = {1.0f, 1.0f, 1.0f, 1.0f};
        F2 end[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
        bool on[4];
        const bool p1 = (pairs & 1u) != 0u, p2 = (pairs & 2u) != 0u;
        on[0] = p1 && e.x > 0.0f;
        on[1] = p1;
        on[2] = p2;
        on[3] = p2 && edges_at(X, Y, 1, 0).x > 0.0f;
        const bool gate0 = on[0], gate3 = on[3];
        const int wm = V.w - 1, hm = V.h - 1;
        while (SM_ANY(on[0] || on[1] || on[2] || on[3])) {
            uint32_t r0[4][DIAG_BATCH], r1[2][DIAG_BATCH];     // r0: the texel at floor(position); r1: its right neighbour (second pair only)
            for (int s = 0; s < 4; s++) {
                if (!SM_ANY(on[s])) continue;                      // wave-uniform
                float px = tx[s], py = ty[s];
                for (int k = 0; k < DIAG_BATCH; k++) {
                    px = 1.0f * dx[s] + px;
                    py = 1.0f * dy[s] + py;
                    const int i = (int)floorf(px), j = clampi((int)floorf(py), hm);
                    r0[s][k] = src.raw(clampi(i, wm), j);
                    if (s >= 2) r1[s - 2][k] = src.raw(clampi(i + 1, wm), j);
                }
            }
            for (int s = 0; s < 4; s++) {
                if (!SM_ANY(on[s])) continue;
                for (int k = 0; k < DIAG_BATCH && on[s]; k++) {
                    tx[s] = 1.0f * dx[s] + tx[s];
                    ty[s] = 1.0f * dy[s] + ty[s];
                    n[s] = 1.0f * 1.0f + n[s];
                    F2 v;
                    if (s < 2) {       // texel centre: the bilinear sum is 1 * texel + three exact zeros
                        v.x = unorm8(r0[s][k] & 255u);
                        v.y = unorm8(r0[s][k] >> 8);
                    } else {           // a = 0.25, b = 0: w00 = 0.75, w10 = 0.25, the lower row's weights are exact zeros
                        const F2 q{0.75f * unorm8(r0[s][k] & 255u) + 0.25f * unorm8(r1[s - 2][k] & 255u), 0.75f * unorm8(r0[s][k] >> 8) + 0.25f * unorm8(r1[s - 2][k] >> 8)};
                        v.x = decode1(q.x);
                        v.y = rintf(q.y);
                    }
                    end[s] = v;
                    wgt[s] = v.x * 0.5f + v.y * 0.5f;
                    on[s] = n[s] < last && wgt[s] > 0.9f;
                }
            }
        }
        out[0] = gate0 ? F2{n[0] + ((end[0].y > 0.9f) ? 1.0f : 0.0f), wgt[0]} : F2{0.0f, 0.0f};
        out[1] = F2{n[1], wgt[1]};
        out[2] = F2{n[2], wgt[2]};
        out[3] = gate3 ? F2{n[3] + ((end[3].y > 0.9f) ? 1.0f : 0.0f), wgt[3]} : F2{0.0f, 0.0f};
    }
    // The same weights with every load of a stage in flight together (round 3): the seven crossing-edge texels of both diagonal pairs, then
    // both area texels. The positions are texel centres or lie a quarter texel beside / below one, so the taps are known: (0.75, 0.25)
    // over two columns, (0.25, 0.75) over two rows, single texels -- the bilinear sums of edges_at with their exact zeros left out (x + 0 = x
    // for the non-negative sums here). Should a position ever NOT have that form (it cannot: the distances are small whole numbers), the
    // wave takes diag_weights_from.
    SM_HDM F2 diag_weights_staged(float X, float Y, F2 s0r, F2 s1r, F2 s2r, F2 s3r, unsigned pairs = 3u) const
    {
        const int wm = V.w - 1, hm = V.h - 1;
        const float dx1 = s0r.x, dz1 = s0r.y, dy1 = s1r.x, dw1 = s1r.y, dx2 = s2r.x, dz2 = s2r.y, dy2 = s3r.x, dw2 = s3r.y;
        const bool on1 = (pairs & 1u) != 0u && dx1 + dy1 > 2.0f, on2 = (pairs & 2u) != 0u && dx2 + dy2 > 2.0f;
        const float t0x = (-dx1 + 0.25f) * 1.0f + X, t0y = dx1 * 1.0f + Y, t1x = dy1 * 1.0f + X, t1y = (-dy1 - 0.25f) * 1.0f + Y;
        const float ax = -dx2 * 1.0f + X, ay = -dx2 * 1.0f + Y, bx = dy2 * 1.0f + X, by = dy2 * 1.0f + Y;
        const float f0x = floorf(t0x), f0y = floorf(t0y), f1x = floorf(t1x), f1y = floorf(t1y);
        const bool form = (!on1 || (t0x - f0x == 0.25f && t0y == f0y && t1x == f1x && t1y - f1y == 0.75f)) &&
                          (!on2 || (ax == floorf(ax) && ay == floorf(ay) && bx == floorf(bx) && by == floorf(by)));
        if (SM_ANY(!form)) return diag_weights_from(X, Y, s0r, s1r, s2r, s3r, pairs);
        uint32_t a0 = 0u, a1 = 0u, b0 = 0u, b1 = 0u, c0 = 0u, c1 = 0u, c2 = 0u;
        if (on1) {
            const int j = clampi((int)f0y, hm), i = clampi((int)f1x + 1, wm);
            a0 = src.raw(clampi((int)f0x - 1, wm), j);            // edges_at(t0x, t0y, -1, 0): columns fx - 1, fx
            a1 = src.raw(clampi((int)f0x, wm), j);
            b0 = src.raw(i, clampi((int)f1y, hm));                // edges_at(t1x, t1y, 1, 0): rows fy, fy + 1
            b1 = src.raw(i, clampi((int)f1y + 1, hm));
        }
        if (on2) {
            c0 = src.raw(clampi((int)ax - 1, wm), clampi((int)ay, hm));
            c1 = src.raw(clampi((int)ax, wm), clampi((int)ay - 1, hm));
            c2 = src.raw(clampi((int)bx + 1, wm), clampi((int)by, hm));
        }
        float e11 = 0.0f, e12 = 0.0f, e21 = 0.0f, e22 = 0.0f;
        if (on1) {
            const F2 s0{0.75f * unorm8(a0 & 255u) + 0.25f * unorm8(a1 & 255u), 0.75f * unorm8(a0 >> 8) + 0.25f * unorm8(a1 >> 8)};
            const F2 s1{0.25f * unorm8(b0 & 255u) + 0.75f * unorm8(b1 & 255u), 0.25f * unorm8(b0 >> 8) + 0.75f * unorm8(b1 >> 8)};
            const float cy = decode1(s0.x), cx = rintf(s0.y), cw = decode1(s1.x), cz = rintf(s1.y);
            e11 = 2.0f * cx + cy;
            e12 = 2.0f * cz + cw;
            if (step_(0.9f, dz1) != 0.0f) e11 = 0.0f;
            if (step_(0.9f, dw1) != 0.0f) e12 = 0.0f;
        }
        if (on2) {
            const float cx = unorm8(c0 >> 8), cy = unorm8(c1 & 255u);
            e21 = 2.0f * cx + cy;
            e22 = 2.0f * unorm8(c2 >> 8) + unorm8(c2 & 255u);
            if (step_(0.9f, dz2) != 0.0f) e21 = 0.0f;
            if (step_(0.9f, dw2) != 0.0f) e22 = 0.0f;
        }
        // both area texels (SMAAAreaDiag): whole-numbered positions again, or the wave takes the general sampler
        const float u1 = (20.0f * e11 + dx1) + 80.0f, v1 = 20.0f * e12 + dy1, u2 = (20.0f * e21 + dx2) + 80.0f, v2 = 20.0f * e22 + dy2;
        const bool whole = (!on1 || (u1 == floorf(u1) && v1 == floorf(v1))) && (!on2 || (u2 == floorf(u2) && v2 == floorf(v2)));
        F2 wts{0.0f, 0.0f};
        if (SM_ANY(!whole)) {
            if (on1) { const F2 a = area_diag(dx1, dy1, e11, e12); wts.x += a.x; wts.y += a.y; }
            if (on2) { const F2 a = area_diag(dx2, dy2, e21, e22); wts.x += a.y; wts.y += a.x; }
            return wts;
        }
        uint32_t p1 = 0u, p2 = 0u;
        if (on1) p1 = V.area[(size_t)clampi((int)v1, AREA_H - 1) * AREA_W + clampi((int)u1, AREA_W - 1)];
        if (on2) p2 = V.area[(size_t)clampi((int)v2, AREA_H - 1) * AREA_W + clampi((int)u2, AREA_W - 1)];
        if (on1) { wts.x += unorm8(p1 & 255u); wts.y += unorm8(p1 >> 8); }
        if (on2) { wts.x += unorm8(p2 >> 8); wts.y += unorm8(p2 & 255u); }
        return wts;
    }
    SM_HDM F2 diag_weights_from(float X, float Y, F2 s0r, F2 s1r, F2 s2r, F2 s3r, unsigned pairs = 3u) const
    {
        F2 wts{0.0f, 0.0f};
        if (pairs & 1u) {
            const float dx = s0r.x, dz = s0r.y, dy = s1r.x, dw = s1r.y;
            if (dx + dy > 2.0f) {
                const F2 s0 = edges_at((-dx + 0.25f) * 1.0f + X, dx * 1.0f + Y, -1, 0), s1 = edges_at(dy * 1.0f + X, (-dy - 0.25f) * 1.0f + Y, 1, 0);
                // c.yxwz = decode(c.xyzw): decoded red of each fetch lands in c.y / c.w, rounded green in c.x / c.z
                const float cy = decode1(s0.x), cx = rintf(s0.y), cw = decode1(s1.x), cz = rintf(s1.y);
                float c1 = 2.0f * cx + cy, c2 = 2.0f * cz + cw;
                if (step_(0.9f, dz) != 0.0f) c1 = 0.0f;
                if (step_(0.9f, dw) != 0.0f) c2 = 0.0f;
                const F2 a = area_diag(dx, dy, c1, c2);
                wts.x += a.x;
                wts.y += a.y;
            }
        }
        if (pairs & 2u) {
            const float dx = s2r.x, dz = s2r.y, dy = s3r.x, dw = s3r.y;
            if (dx + dy > 2.0f) {
                const float ax = -dx * 1.0f + X, ay = -dx * 1.0f + Y, bx = dy * 1.0f + X, by = dy * 1.0f + Y;
                const float cx = edges_at(ax, ay, -1, 0).y, cy = edges_at(ax, ay, 0, -1).x;
                const F2 s = edges_at(bx, by, 1, 0);
                float c1 = 2.0f * cx + cy, c2 = 2.0f * s.y + s.x;
                if (step_(0.9f, dz) != 0.0f) c1 = 0.0f;
                if (step_(0.9f, dw) != 0.0f) c2 = 0.0f;
                const F2 a = area_diag(dx, dy, c1, c2);
                wts.x += a.y;
                wts.y += a.x;
            }
        }
        return wts;
    }

    // SMAASearchLength (SMAA.h:998-1015): texel (32 e.x + 66 offset, 32 - 32 e.y) of the 64 x 16 table
    SM_HDM float search_length(float ex, float ey, float offset) const
    {
        return sample_r8(V.search, SEARCH_W, SEARCH_H, 32.0f * ex + 66.0f * offset, -32.0f * ey + 32.0f);
    }
    // SMAASearchXLeft / XRight / YUp / YDown (SMAA.h:1020-1077): two texels per step from (tx, ty) towards `end`
    SM_HDM float search_x(float tx, float ty, float end, float dir, int x, int y) const
    {
        F2 e{0.0f, 1.0f};
        const float stepx = (dir * 2.0f) * 1.0f;
        const int n = planes.count_x(x, y, P.max_steps, dir < 0.0f);
        if (n > 0) {      // the loop below would consume n fetches: its last one, and where it leaves tx (stepx * k is exact: small dyadic numbers)
            e = edges_at(stepx * (float)(n - 1) + tx, ty);
            tx = stepx * (float)n + tx;
        } else
        while ((dir < 0.0f ? tx > end : tx < end) && e.y > 0.8281f && e.x == 0.0f) {
            F2 s[SEARCH_BATCH];
            float px = tx;
            for (int k = 0; k < SEARCH_BATCH; k++) {
                s[k] = edges_at(px, ty);
                px = stepx + px;
            }
            for (int k = 0; k < SEARCH_BATCH && (dir < 0.0f ? tx > end : tx < end) && e.y > 0.8281f && e.x == 0.0f; k++) {
                e = s[k];
                tx = stepx + tx;
            }
        }
        const float off = -(255.0f / 127.0f) * search_length(e.x, e.y, dir < 0.0f ? 0.0f : 0.5f) + 3.25f;
        return (-dir) * off + tx;
    }
    SM_HDM float search_y(float tx, float ty, float end, float dir, int x, int y) const
    {
        F2 e{1.0f, 0.0f};
        const float stepy = (dir * 2.0f) * 1.0f;
        const int n = planes.count_y(x, y, P.max_steps, dir < 0.0f);
        if (n > 0) {
            e = edges_at(tx, stepy * (float)(n - 1) + ty);
            ty = stepy * (float)n + ty;
        } else
        while ((dir < 0.0f ? ty > end : ty < end) && e.x > 0.8281f && e.y == 0.0f) {
            F2 s[SEARCH_BATCH];
            float py = ty;
            for (int k = 0; k < SEARCH_BATCH; k++) {
                s[k] = edges_at(tx, py);
                py = stepy + py;
            }
            for (int k = 0; k < SEARCH_BATCH && (dir < 0.0f ? ty > end : ty < end) && e.x > 0.8281f && e.y == 0.0f; k++) {
                e = s[k];
                ty = stepy + ty;
            }
        }
        const float off = -(255.0f / 127.0f) * search_length(e.y, e.x, dir < 0.0f ? 0.0f : 0.5f) + 3.25f;
        return (-dir) * off + ty;
    }
    SM_HDM F2 area(float d1, float d2, float e1, float e2) const   // SMAAArea (SMAA.h:1083-1095), offset 0
    {
        return sample_rg(V.area, AREA_W, AREA_H, 16.0f * rintf(4.0f * e1) + d1, 16.0f * rintf(4.0f * e2) + d2);
    }
    // SMAADetectHorizontalCornerPattern / Vertical (SMAA.h:1100-1140). horizontal: red of the rows above / two below the line at both
    // ends; vertical: green of the columns right / two left.
    SM_HDM void corners(F2& wts, float ax, float ay, float bx, float by, float d1, float d2, bool horizontal) const
    {
        if (P.corner_rounding < 0) return;
        const float lx = step_(d1, d2), ly = step_(d2, d1);
        const float keep = 1.0f - (float)P.corner_rounding / 100.0f;
        float rx = keep * lx, ry = keep * ly;
        rx /= lx + ly;
        ry /= lx + ly;
        float fx = 1.0f, fy = 1.0f;
        if (horizontal) {
            fx -= rx * edges_at(ax, ay, 0, 1).x;
            fx -= ry * edges_at(bx, by, 1, 1).x;
            fy -= rx * edges_at(ax, ay, 0, -2).x;
            fy -= ry * edges_at(bx, by, 1, -2).x;
        } else {
            fx -= rx * edges_at(ax, ay, 1, 0).y;
            fx -= ry * edges_at(bx, by, 1, 1).y;
            fy -= rx * edges_at(ax, ay, -2, 0).y;
            fy -= ry * edges_at(bx, by, -2, 1).y;
        }
        wts.x *= sat_(fx);
        wts.y *= sat_(fy);
    }

    // The four horizontal / vertical searches of SMAABlendingWeightCalculationPS (SMAA.h:1171-1179,1211-1219), k = 0 left, 1 right, 2 up,
    // 3 down: the texel-space coordinate the search ends at. Independent of each other, like the diagonal ones.
    SM_HDM float ortho_search(int k, float X, float Y) const
    {
        const float S = (float)P.max_steps;
        const int x = (int)X, y = (int)Y;
        if (k == 0) return search_x(X - 0.25f, Y - 0.125f, (-2.0f * S) * 1.0f + (X - 0.25f), -1.0f, x, y);
        if (k == 1) return search_x(X + 1.25f, Y - 0.125f, (2.0f * S) * 1.0f + (X + 1.25f), 1.0f, x, y);
        if (k == 2) return search_y(X - 0.125f, Y - 0.25f, (-2.0f * S) * 1.0f + (Y - 0.25f), -1.0f, x, y);
        return search_y(X - 0.125f, Y + 1.25f, (2.0f * S) * 1.0f + (Y + 1.25f), 1.0f, x, y);
    }
    SM_HDM F2 north_from(float X, float Y, float cx, float cz) const   // weights.rg from the ends of the two x searches
    {
        const float cy = Y - 0.25f;
        const float e1 = edges_at(cx, cy).x;
        const float d1 = fabsf(rintf(cx - X)), d2 = fabsf(rintf(cz - X));
        const float e2 = edges_at(cz, cy, 1, 0).x;
        F2 wgt = area(sqrtf(d1), sqrtf(d2), e1, e2);
        corners(wgt, cx, Y, cz, Y, d1, d2, true);
        return wgt;
    }
    SM_HDM F2 west_from(float X, float Y, float cy, float cz) const    // weights.ba from the ends of the two y searches
    {
        const float cx = X - 0.25f;
        const float e1 = edges_at(cx, cy).y;
        const float d1 = fabsf(rintf(cy - Y)), d2 = fabsf(rintf(cz - Y));
        const float e2 = edges_at(cx, cz, 0, 1).y;
        F2 wgt = area(sqrtf(d1), sqrtf(d2), e1, e2);
        corners(wgt, X, cy, X, cz, d1, d2, false);
        return wgt;
    }
    SM_HDM static uint32_t pack_weights(F4 w) { return to_unorm8(w.x) | (to_unorm8(w.y) << 8) | (to_unorm8(w.z) << 16) | (to_unorm8(w.w) << 24); }

    // SMAABlendingWeightCalculationPS (SMAA.h:1145-1243) for the pixel at (x, y) in three independent parts -- the diagonal weights, the
    // north edge's weights, the west edge's weights -- and the rule that picks among them (combine). One thread may run them one after the
    // other (weights(): the shader's order; host build, reference for the device kernel), or three WAVES may run one part each for the same
    // 64 pixels at the same time (smaa_kernel.hip, round 4): every part is a chain of ~6-10 dependent memory round trips, and a wave that
    // runs all three is as slow as their sum. The parts do not depend on each other; a part whose result combine() then discards was
    // computed from the same inputs it would have seen anyway (its inputs are the pass-1 textures only).
    SM_HDM F2 own_edges(int x, int y) const
    {
        const uint32_t own = src.raw(clampi(x, V.w - 1), clampi(y, V.h - 1));   // texel-centre fetch of the pixel's own edges
        return F2{unorm8(own & 255u), unorm8(own >> 8)};
    }
    SM_HDM bool has_diag_part(F2 e) const { return e.y > 0.0f && P.max_steps_diag > 0; }
    // SMAACalculateDiagWeights (SMAA.h:1157). With `pairs` = 1 or 2 only that pair of diagonals: the function's result is
    // (0 + a1.x) + a2.y, (0 + a1.y) + a2.x with a1 / a2 the area texels of the pairs that found a diagonal -- bytes / 255, never negative --
    // so part_diag(.., 1) + part_diag(.., 2), component by component, is the same float (x + 0 = x for x >= +0).
    SM_HDM F2 part_diag(float X, float Y, F2 e, unsigned pairs = 3u) const
    {
#if SMAA_DIAG_IN_STEP
        F2 ds[4] = {{0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}, {0.0f, 0.0f}};
        diag_searches(X, Y, e, ds, pairs);
        SMAA_PH(9);
        return diag_weights_staged(X, Y, ds[0], ds[1], ds[2], ds[3], pairs);
#else
        const F2 s0 = diag_search(0, X, Y, e), s1 = diag_search(1, X, Y, e), s2 = diag_search(2, X, Y, e), s3 = diag_search(3, X, Y, e);
        return diag_weights_from(X, Y, s0, s1, s2, s3, pairs);
#endif
    }
    SM_HDM F2 part_north(float X, float Y) const                 // SMAA.h:1165-1204
    {
        const float cx = ortho_search(0, X, Y);
        const float cz = ortho_search(1, X, Y);
        return north_from(X, Y, cx, cz);
    }
    SM_HDM F2 part_west(float X, float Y) const                  // SMAA.h:1211-1240
    {
        const float cy = ortho_search(2, X, Y);
        const float cz = ortho_search(3, X, Y);
        return west_from(X, Y, cy, cz);
    }
    // which parts the shader's control flow USES (SMAA.h:1152-1163, 1206-1209): the diagonal weights stand unless they cancel
    // (weights.r == -weights.g), in which case the north weights replace them; a standing diagonal also skips the west edge.
    SM_HDM static uint32_t combine(F2 e, bool diag_enabled, F2 dwt, F2 north, F2 west)
    {
        F4 out{0.0f, 0.0f, 0.0f, 0.0f};
        if (e.y > 0.0f) {
            bool hv = true;
            if (diag_enabled) {
                out.x = dwt.x;
                out.y = dwt.y;
                hv = (out.x == -out.y);
            }
            if (hv) {
                out.x = north.x;
                out.y = north.y;
            } else {
                e.x = 0.0f;
            }
        }
        if (e.x > 0.0f) {
            out.z = west.x;
            out.w = west.y;
        }
        return pack_weights(out);
    }
    SM_HDM uint32_t weights(int x, int y) const
    {
        const float X = (float)x, Y = (float)y;
        F4 out{0.0f, 0.0f, 0.0f, 0.0f};
        F2 e = own_edges(x, y);
        SMAA_PH(1);
        bool hv = e.y > 0.0f;
        if (e.y > 0.0f) {
            if (P.max_steps_diag > 0) {
                const F2 dwt = part_diag(X, Y, e);
                out.x = dwt.x;
                out.y = dwt.y;
                hv = (out.x == -out.y);
            }
        }
        SMAA_PH(2);
        if (e.y > 0.0f) {
            if (hv) {
                const F2 wgt = part_north(X, Y);
                out.x = wgt.x;
                out.y = wgt.y;
            } else {
                e.x = 0.0f;
            }
        }
        SMAA_PH(3);
        if (e.x > 0.0f) {
            const F2 wgt = part_west(X, Y);
            out.z = wgt.x;
            out.w = wgt.y;
        }
        SMAA_PH(4);
        return pack_weights(out);
    }
};
using Blend = BlendT<TexEdges>;

// ---- pass 3: neighbourhood blending (SMAA.h:1252-1300) ---------------------------------------------------------
// Returns true and the new RGBA8 texel when pixel (x, y) is blended; false when its four weights are all zero (the output is then the
// input texel, which the dense pass has already copied).
// MASKED (the HIP kernels): a weight texel counts only where the CURRENT frame has an edge pixel (`member`, the row bit plane) -- the weight
// texture is then never cleared: texels of earlier frames' edge pixels stay behind and are not looked at, exactly as if they were the zeros
// the reference's glClear + discard leave (GLWrapper.cpp:189-190).
template <bool MASKED = false>
SM_HD bool neighborhood(const Views& V, int x, int y, uint32_t& out, const PlaneTex* member = nullptr)
{
    const int xr = clampi(x + 1, V.w - 1), yt = clampi(y + 1, V.h - 1);
    uint32_t own = V.blend[(size_t)y * V.w + x], right = V.blend[(size_t)y * V.w + xr], top = V.blend[(size_t)yt * V.w + x];
    if (MASKED) {
        if (!member->any(x, y)) own = 0u;
        if (!member->any(xr, y)) right = 0u;
        if (!member->any(x, yt)) top = 0u;
    }
    const float ax = unorm8(right >> 24), ay = unorm8((top >> 8) & 255u), aw = unorm8(own & 255u), az = unorm8((own >> 16) & 255u);
    if (ax * 1.0f + ay * 1.0f + az * 1.0f + aw * 1.0f < 1e-5f) return false;
    const float X = (float)x, Y = (float)y;
    const bool horiz = max_(ax, az) > max_(ay, aw);
    float bx = 0.0f, by = ay, bz = 0.0f, bw = aw, wx = ay, wy = aw;
    if (horiz) { bx = ax; by = 0.0f; bz = az; bw = 0.0f; wx = ax; wy = az; }
    const float sum = wx * 1.0f + wy * 1.0f;
    wx /= sum;
    wy /= sum;
    const F4 s0 = sample_rgba(V.color, V.w, V.h, bx * 1.0f + X, by * 1.0f + Y), s1 = sample_rgba(V.color, V.w, V.h, bz * -1.0f + X, bw * -1.0f + Y);
    F4 c{wx * s0.x, wx * s0.y, wx * s0.z, wx * s0.w};
    c.x += wy * s1.x;
    c.y += wy * s1.y;
    c.z += wy * s1.z;
    c.w += wy * s1.w;
    out = to_unorm8(c.x) | (to_unorm8(c.y) << 8) | (to_unorm8(c.z) << 16) | (to_unorm8(c.w) << 24);
    return true;
}

}  // namespace smaa
