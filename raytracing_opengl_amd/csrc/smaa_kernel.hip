// smaa_kernel.hip -- the SMAA post-process as gfx950 kernels.
//
// The reference runs three full-screen fragment passes through three 8-bit render targets (GLWrapper.cpp:173-204): every pass
// reads and writes whole frames although edges -- the only pixels the second and third pass do anything for -- are a few per cent
// of a frame. Here the frame is touched densely ONCE and the rest is sparse:
//
//   smaa_edges_kernel    dense: reads the colour target (16 B per lane, 1 KiB per wave and row), copies it to the screen, detects luma
//                        edges (SMAA.h:689-741) on a four-row window of lumas held in registers, writes the edge texture as two BIT
//                        PLANES (2 bits per pixel -- the texels are 0 or 255: one byte per lane and row, 32 pixels of a row per 64-bit
//                        word; and 16 bits per column and strip, 8 pixels of a column; dense, 2 x 2 MB at 4K), brings the RG8 edge texture up to
//                        date (texels where this frame or the previous one has an edge) and appends the strip's edge pixels to a list --
//                        one atomic per 256 x 8 strip, none for the strips without an edge;
//   smaa_weights_roles_kernel  over the list: blending weights (SMAA.h:1145-1243) -> RGBA8 weight texel. The step counts of the four
//                        orthogonal searches come from whole plane words (smaa_device.h SearchPlanes), single edge texels from the RG8
//                        texture. (Round 2 walked every edge on the texture: up to 32 dependent 4-tap fetches per direction.) Round 4: four
//                        waves per 64 listed pixels, one independent part of the computation each (first / second pair of diagonals,
//                        north edge, west edge); round 3's one-thread-per-pixel kernel (smaa_weights_kernel) is the -DSMAA_ROLE_WAVES=0 build.
//   smaa_blend_kernel    over the list: neighbourhood blending (SMAA.h:1252-1300) of the listed pixel, its left and its lower
//                        neighbour -- the only pixels whose four weights can be non-zero -- overwriting their screen texels; a weight
//                        texel is looked at only where the row plane has an edge pixel.
//
// Nothing is cleared between frames (round 3; round 2 zeroed the previous frame's edge and weight texels through its list with a kernel
// of its own -- 5 us, the fixed cost of any sparse kernel here): the planes are rewritten densely every frame; the dense kernel keeps the
// previous frame's row plane and rewrites the RG8 edge texels wherever either frame has an edge, so that texture is always exact; and a
// weight texel that an earlier frame left behind at a pixel without an edge is never read -- the same textures, to a reader, as the
// reference's glClear(0) + discard produce (GLWrapper.cpp:177-178,189-190). smaa_expand_kernel masks the weight texture for
// rtx_read_pixels. (Reading the single edge texels from the row plane as well, so that the RG8 texture would not be needed at all, was
// measured: the weight kernel's diagonal searches then pay ~8 instructions more per tap, 26.9 -> 35.5 us at ULTRA. profiles/r03_smaa.txt) Algorithmic HBM traffic per frame: W*H*4 B read + W*H*4 B written, against 6 x W*H*4 B + 2 x W*H*2 B for three
// dense passes.
//
// The list is kept in SMAA_SEGMENTS independent segments, each with its own counter: a strip appends to segment (strip index mod
// SMAA_SEGMENTS), and the sparse kernels walk all segments (blockIdx.y = segment). One shared counter was measured to cost 16 us per 4K
// frame by itself -- ~3000 atomics on ONE address serialise in one L2 channel (profiles/r02_smaa_ablation.txt) -- with 64 addresses
// the append got 8 us cheaper, and with every counter in a 128-byte line of ITS OWN (SMAA_COUNT_STRIDE; round 3 -- 64 adjacent counters are
// two L2 lines, i.e. two atomic units) the atomics spread over the L2 channels. A segment's capacity is the pixel count of the strips that
// map to it, so it cannot overflow.
//
// Counters: two sets, used alternately. Frame f appends to set f & 1; smaa_weights of frame f zeroes set (f-1) & 1 for frame f+1 (the
// stats of frame f-1 were read from it until then).
// The sparse kernels treat the segments as one list again (SegmentedList below), so their work stays balanced.
#include "smaa_kernel.h"

#include <cstdlib>

#include <hip/hip_ext.h>

#ifdef SMAA_PHASE_TIMES
// diagnostic build (tools/smaa_phase_times.py): per wave of the last smaa_weights_kernel launch, s_memrealtime (10 ns ticks) at [0] kernel
// entry, [5] list prefix done, [6] list entry read, [1..4] the convergent points of smaa::BlendT::weights, [7] exit; [8] = 1 if the wave had a pixel
#include <hip/hip_runtime.h>
__device__ unsigned long long g_smaa_ph[8192][12];
#define SMAA_PH(k) do { g_smaa_ph[blockIdx.x * 4 + (threadIdx.x >> 6)][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
// the same for the dense kernel: [0] entry, [1] luma tables built (barrier), [2] first rows loaded, [3] rows done, [4] planes written, [5] exit
__device__ unsigned long long g_smaa_ep[8192][6];
#define SMAA_EP(k) do { g_smaa_ep[blockIdx.x * 4 + (threadIdx.x >> 6)][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
extern "C" __attribute__((visibility("default"))) int rtx_debug_smaa_edge_times(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_smaa_ep), sizeof(unsigned long long) * 8192 * 6) == hipSuccess ? 0 : 1;
}
extern "C" __attribute__((visibility("default"))) int rtx_debug_smaa_phase_times(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_smaa_ph), sizeof(unsigned long long) * 8192 * 12) == hipSuccess ? 0 : 1;
}
// the role-split weight kernel (tools/smaa_role_times.py): per wave [0] entry, [1] list prefix done, [2] list entry + own texel read, [3] the wave's
// part done, [4] past the barrier (every part of the workgroup done), [5] exit; [6] = 1 + role if some lane had a pixel, [7] = 1 if some lane ran the part
__device__ unsigned long long g_smaa_rp[8192][8];
#define SMAA_RP(k) do { g_smaa_rp[blockIdx.x * 4 + (threadIdx.x >> 6)][k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define SMAA_RP_SET(k, v) do { g_smaa_rp[blockIdx.x * 4 + (threadIdx.x >> 6)][k] = (v); } while (0)
extern "C" __attribute__((visibility("default"))) int rtx_debug_smaa_role_times(unsigned long long* out)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_smaa_rp), sizeof(unsigned long long) * 8192 * 8) == hipSuccess ? 0 : 1;
}
#endif

#ifndef SMAA_EP
#define SMAA_EP(k)
#endif
#ifndef SMAA_RP
#define SMAA_RP(k)
#define SMAA_RP_SET(k, v)
#endif

#include "smaa_device.h"

namespace {

constexpr int STRIP_W = 256;     // pixels per wave and row: 64 lanes x 4 pixels = one 1 KiB row segment per load
#ifndef SMAA_ABL
#define SMAA_ABL 0   /* timing ablations only (tools/ab_smaa_ablate.sh): 1 = no edge arithmetic, 2 = (round 4's strip-border loads; gone), 4 = no cross-lane moves, 8 = no LDS luma tables, 16 = no append, 32 = rows above the first are not loaded, 64 = no screen copy, 128 = the append's atomic is not issued (entries overwrite each other), 256 = no plane / texel stores, 512 = the weight kernel only walks its list */
#endif
#ifndef SMAA_STORES_IN_LOOP
#define SMAA_STORES_IN_LOOP 1  /* a row's plane byte and its RG8 edge texels are stored in the row's own iteration of the strip walk, under the row stream, instead
                                  of sixteen stores per wave behind the walk -- where every wave of the frame arrives at the same time (round 5) */
#endif
#ifndef SMAA_SCALAR_ATOMIC
#define SMAA_SCALAR_ATOMIC 0  /* the append reserves its list entries with a SCALAR atomic (s_atomic_add, returns under lgkmcnt): a vector atomic's return value is
                                 counted by vmcnt, in order behind every store the wave has issued -- an appending wave waited for its eight row stores to drain (round 5) */
#endif
#ifndef SMAA_EARLY_ATOMIC
#define SMAA_EARLY_ATOMIC (!SMAA_SCALAR_ATOMIC)   /* the append's atomic is issued before the plane and texel stores (round 4: traced ULTRA edges 25.2 -> 23.9 us, other presets +-0) */
#endif
#ifndef SMAA_ROW_DIST
#define SMAA_ROW_DIST 1       /* a row is requested this many iterations of the strip walk before the one that needs it */
#endif
#ifndef SMAA_XCD_BANDS
#define SMAA_XCD_BANDS 1
#endif
#ifndef SMAA_LINEAR_STRIPS
#define SMAA_LINEAR_STRIPS 1
#endif
#ifndef SMAA_STRIP_H
#define SMAA_STRIP_H 8
#endif
constexpr int STRIP_H_DEFAULT = SMAA_STRIP_H;   // rows per wave: H + 3 rows are read for H rows of output
constexpr int WAVES_PER_WG = 4;  // four strips side by side per workgroup

__device__ __forceinline__ uint32_t load_px(const uint32_t* color, int w, int h, int x, int y)
{
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);    // CLAMP_TO_EDGE
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    return color[(size_t)y * w + x];
}

struct Row4 { float l[4]; };   // lumas of one lane's four pixels in one row

// The dense pass. One wave owns a strip of STRIP_W x STRIP_H pixels and walks it bottom to top with a four-row window of lumas in
// registers (rows y-2, y-1, y, y+1 of SMAA.h:689-741's "top-top", "top", centre, "bottom"); the three horizontal neighbours a lane
// needs come from the adjacent lanes by cross-lane moves, the strip's outermost columns by two extra loads. Every row is copied to the
// screen as it passes. Edge bits are kept in two registers per lane for the whole strip; at the end the wave reserves its list slots with
// ONE atomic (none at all for the strips without an edge -- most of a frame), ranks its pixels with a scan of the lanes' counts and writes list + edge texels.
// VEC (frame widths that are multiples of four, chosen by smaa_launch): a lane's four pixels are ONE 16-byte load and one 16-byte store, and a
// lane is either wholly inside the frame or wholly outside. The two forms are separate instantiations on purpose: as two branches of one
// kernel the compiler merged their tails into four 4-byte loads per lane and row for BOTH (round 5: the ISA of round 4's kernel held no
// 16-byte load at all -- four times the cache look-ups per row).
template <int STRIP_H, bool VEC>
__global__ __launch_bounds__(64 * WAVES_PER_WG) void smaa_edges_kernel(SmaaBuffers b, float threshold, unsigned cur)
{
    static_assert(STRIP_H * 4 * 2 <= 128, "edge bits of a strip live in two 64-bit registers per lane");
    // Per-channel luma terms, unorm8(byte) * weight, as three 256-entry tables in LDS: the same float products the arithmetic contract
    // spells out (smaa::luma_of), computed once per workgroup instead of per pixel -- a pixel's luma is three look-ups and two adds. The
    // dense pass is VALU-bound without this (25 us of luma arithmetic at 4K against 10 us of memory time, profiles/r02_smaa_ablation.txt).
    __shared__ float lut[3][256];
    SMAA_EP(0);
    for (int k = 1; k < 6; k++) SMAA_EP(k);    // (waves that leave early: all stamps = entry)
    // (the tables are built further down, after the strip's first rows have been requested. Measured per wave, tools/smaa_edge_times.py:
    // every wave of the frame starts within a microsecond of the others and the first rows take 4 - 5 us to arrive whether the 1.1 us of
    // table building come before or after the requests -- the opening burst of 4 050 x 4 KB is what the waves wait for)
#if SMAA_ABL & 8
    auto luma = [&](uint32_t rgba) { return (float)(rgba & 255u) + (float)((rgba >> 8) & 255u) * 0.5f; };
#else
    auto luma = [&](uint32_t rgba) { return lut[0][rgba & 255u] + lut[1][(rgba >> 8) & 255u] + lut[2][(rgba >> 16) & 255u]; };
#endif
    const int w = b.w, h = b.h;
    const uint32_t* __restrict__ color = b.color;      // the colour target and the screen are different allocations: let the loads of
    uint32_t* __restrict__ screen = b.screen;          // the next rows move above the stores of this one
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#if SMAA_LINEAR_STRIPS
    // strips are numbered row by row and dealt to the workgroups four at a time: 4 050 strips of a 4K frame are 1 013 workgroups -- at most
    // four on each of the 256 CUs. (A grid of 4 x 270 workgroups of four strips side by side, the last of each row holding three, was 1 080:
    // a fifth workgroup on 56 CUs.)
#if SMAA_XCD_BANDS
    // workgroup g runs on XCD g % 8 (each XCD has its own L2): give every XCD one contiguous band of the frame, so that a strip and the
    // strips above and below it -- which read its first / last rows as their halo -- share an L2
    const int per_xcd = ((int)gridDim.x + 7) / 8, wg = ((int)blockIdx.x % 8) * per_xcd + (int)blockIdx.x / 8;
    const int strips_x = (w + STRIP_W - 1) / STRIP_W, strip_id = wg * WAVES_PER_WG + wave;
#else
    const int strips_x = (w + STRIP_W - 1) / STRIP_W, strip_id = (int)blockIdx.x * WAVES_PER_WG + wave;
#endif
    const bool no_strip = strip_id >= strips_x * ((h + STRIP_H - 1) / STRIP_H);   // wave-uniform; such a wave leaves after the barrier below
    const int x0 = (strip_id % strips_x) * STRIP_W, y0 = no_strip ? 0 : (strip_id / strips_x) * STRIP_H;
#else
    const int x0 = (blockIdx.x * WAVES_PER_WG + wave) * STRIP_W, y0 = blockIdx.y * STRIP_H;
    const bool no_strip = x0 >= w;
#endif
    const int px = x0 + lane * 4;
    const bool vec_ok = VEC && px < w;                                         // (w % 4 == 0: px < w <=> px + 3 < w)
    const int pxl = VEC ? (px < w ? px : w - 4) : px;                          // lanes right of the frame load its last four pixels and keep nothing
    // what the RG8 edge texture still holds for this lane's pixels: the previous resolve's row plane (requested first, needed last). All
    // STRIP_H loads are issued back to back from clamped addresses and masked afterwards: guarded one by one, each was followed by a wait
    // for itself -- eight cache round trips in a row before the strip's first row was asked for (round 5, seen in the ISA).
    const int pw8 = smaa::SearchPlanes::plane_words(w) * 8;                    // bytes per row-plane row
    const int byte_x = px >> 2;
    unsigned long long pbits[2] = {0, 0};
    {
        const uint8_t* const pplane = reinterpret_cast<const uint8_t*>(b.bits_prev);
        const int bxc = byte_x < pw8 ? byte_x : pw8 - 1;
        uint8_t pb[STRIP_H];
#pragma unroll
        for (int r = 0; r < STRIP_H; r++) pb[r] = pplane[(size_t)(y0 + r < h ? y0 + r : h - 1) * pw8 + bxc];
#pragma unroll
        for (int r = 0; r < STRIP_H; r++)
            if (y0 + r < h && byte_x < pw8) pbits[r >> 3] |= (unsigned long long)pb[r] << ((r & 7) * 8);
    }

    auto load_row = [&](int y, uint32_t c[4]) {                               // clamped in y; x clamped per pixel on the scalar path
        const int yc = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        if constexpr (VEC) {
            const uint4 v = *reinterpret_cast<const uint4*>(color + (size_t)yc * w + pxl);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) c[k] = load_px(color, w, h, px + k, yc);
        }
    };
    // The strip's outermost columns of a row -- what lane 0 needs to its left (x0 - 2, x0 - 1) and the strip's last lane to its right
    // (min(x0 + 256, w - 1): CLAMP_TO_EDGE) -- are ONE more load per row, issued with the row: lane 0 fetches x0 - 1, lane 1 x0 - 2, every other lane
    // the right-hand texel; the row's iteration takes their lumas from lanes 0, 1 and 2 (v_readlane). Round 4 fetched them inside
    // `if (lane == 0)` / `if (last lane)` in the iteration that needed them: three dependent cache round trips per row.
    const bool right_lane = lane == 63 || px + 4 >= w;
    const int border_x = lane == 0 ? x0 - 1 : (lane == 1 ? x0 - 2 : x0 + STRIP_W);
    auto load_border = [&](int y) {
        const int yc = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        return load_px(color, w, h, border_x, yc);
    };
    auto lumas = [&](const uint32_t c[4]) {
        Row4 r;
#pragma unroll
        for (int k = 0; k < 4; k++) r.l[k] = luma(c[k]);
        return r;
    };

    // window: lumas of rows y-1 (Lt) and y (Lc), vertical deltas |row - row below| of rows y-1 (dyt) and y (dyc)
    // The strip's rows travel through a ring of SMAA_ROW_DIST + 2 register rows: while row y is worked on, rows y + 1 ... y + 1 + SMAA_ROW_DIST are
    // held or in flight (the loop below is unrolled, every ring index is a constant).
    constexpr int RING = SMAA_ROW_DIST + 2;
    uint32_t c0[4], c1[4], ring[RING][4];
    uint32_t ering[RING];                                                      // the border texels travel with their rows
#pragma unroll
    for (int j = 0; j < RING; j++) ering[j] = 0;
    load_row(y0 - 2, c0);
    load_row(y0 - 1, c1);
#pragma unroll
    for (int j = 0; j <= SMAA_ROW_DIST; j++) {
        load_row(y0 + j, ring[j]);
        if (j < STRIP_H) ering[j] = load_border(y0 + j);
    }
    {
        // one entry of each table per thread, the weights as immediates: indexed from an array they were three loads from constant memory in a
        // loop, each followed by a wait for EVERY load in flight -- the strip's first rows included (round 5, seen in the ISA)
        static_assert(64 * WAVES_PER_WG == 256, "one table entry per thread");
        const float u = smaa::unorm8((uint32_t)threadIdx.x);
        lut[0][threadIdx.x] = u * 0.2126f;
        lut[1][threadIdx.x] = u * 0.7152f;
        lut[2][threadIdx.x] = u * 0.0722f;
    }
    __syncthreads();
    SMAA_EP(1);
    if (no_strip) return;
    const Row4 Ltt = lumas(c0);
    Row4 Lt = lumas(c1);
    Row4 Lc = lumas(ring[0]);
    Row4 dyt, dyc;
#pragma unroll
    for (int k = 0; k < 4; k++) { dyt.l[k] = fabsf(Lt.l[k] - Ltt.l[k]); dyc.l[k] = fabsf(Lc.l[k] - Lt.l[k]); }
    SMAA_EP(2);
    // a lane's four RG8 edge texels of a row as ONE 8-byte store (wherever this frame or the previous one has an edge among them)
    auto store_texels = [&](int y, unsigned bits8) {
        uint16_t tx[4];
#pragma unroll
        for (int k = 0; k < 4; k++) tx[k] = (uint16_t)((((bits8 >> (2 * k)) & 1u) ? 0x00ffu : 0u) | (((bits8 >> (2 * k)) & 2u) ? 0xff00u : 0u));
        uint16_t* const dst = b.edges + (size_t)y * w + px;
        if (vec_ok) {
            *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)tx[0] | ((unsigned)tx[1] << 16), (unsigned)tx[2] | ((unsigned)tx[3] << 16));
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (px + k < w) dst[k] = tx[k];
        }
    };
    unsigned long long ebits[2] = {0, 0};                                      // worked on. 2 bits (R, G) per pixel: bit (row * 4 + k) * 2
    unsigned valid = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) valid |= (px + k < w) ? (3u << (2 * k)) : 0u;
#pragma unroll
    for (int r = 0; r < STRIP_H; r++) {
        const int y = y0 + r;
        if (y >= h) break;                                                     // wave-uniform
        uint32_t* const cc = ring[r % RING];
        uint32_t* const cb = ring[(r + 1) % RING];
        const uint32_t ec = ering[r % RING];
        if (r + 1 + SMAA_ROW_DIST <= STRIP_H) {
            uint32_t* const cn = ring[(r + 1 + SMAA_ROW_DIST) % RING];
#if SMAA_ABL & 32
            for (int k = 0; k < 4; k++) cn[k] = cb[k] ^ (uint32_t)r;
#else
            load_row(y + 1 + SMAA_ROW_DIST, cn);
            if (r + 1 + SMAA_ROW_DIST < STRIP_H) ering[(r + 1 + SMAA_ROW_DIST) % RING] = load_border(y + 1 + SMAA_ROW_DIST);
#endif
        }
        const Row4 Lb = lumas(cb);
        // the dense copy: pass 3 for every pixel without weights (streamed: nothing reads these lines again before the sparse passes)
#if SMAA_ABL & 64
        if (threshold < -1.0e30f)                                             // (timing ablation: the screen copy is never stored)
#endif
        if constexpr (VEC) {
            if (vec_ok) {
                typedef uint32_t v4u __attribute__((ext_vector_type(4)));
                const v4u v = {cc[0], cc[1], cc[2], cc[3]};
                __builtin_nontemporal_store(v, reinterpret_cast<v4u*>(screen + (size_t)y * w + px));
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (px + k < w) screen[(size_t)y * w + px + k] = cc[k];
        }
        // horizontal neighbours of this row: two to the left of pixel 0, one to the right of pixel 3
#if SMAA_ABL & 4
        float left1 = Lc.l[3], left2 = Lc.l[2], right = Lc.l[0];
#else
        // lumas of the strip's border texels: lane 0 holds x0 - 1, lane 1 x0 - 2, lane 2 the right-hand one (wave-uniform after the read)
        const float le = luma(ec);
        const int sl1 = __builtin_amdgcn_readlane(__float_as_int(le), 0), sl2 = __builtin_amdgcn_readlane(__float_as_int(le), 1);
        const float sr = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(le), 2));
        // the neighbouring lanes' pixels by whole-wave DPP shifts (one VALU move each; __shfl_up / __shfl_down are ds_bpermute: an LDS round
        // trip). wave_shr:1 = lane i reads lane i - 1 and lane 0 keeps `old` -- which is the border texel it needs.
        const float left1 = __int_as_float(__builtin_amdgcn_update_dpp(sl1, __float_as_int(Lc.l[3]), 0x138, 0xf, 0xf, false));
        const float left2 = __int_as_float(__builtin_amdgcn_update_dpp(sl2, __float_as_int(Lc.l[2]), 0x138, 0xf, 0xf, false));
        const float rnext = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(sr), __float_as_int(Lc.l[0]), 0x130, 0xf, 0xf, false));
        const float right = right_lane ? sr : rnext;
#endif
        const float row[7] = {left2, left1, Lc.l[0], Lc.l[1], Lc.l[2], Lc.l[3], right};
        float dx[6];                                                           // dx[j] = |row[j+1] - row[j]|: pixel k's own delta is dx[k+1]
#pragma unroll
        for (int j = 0; j < 6; j++) dx[j] = fabsf(row[j + 1] - row[j]);
        Row4 dyb;                                                              // the row above's own vertical delta = this row's "bottom" delta
        unsigned bits = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            dyb.l[k] = fabsf(Lb.l[k] - Lc.l[k]);
#if !(SMAA_ABL & 1)
            bits |= smaa::edge_bits(threshold, dx[k + 1], dyc.l[k], dx[k + 2], dyb.l[k], dx[k], dyt.l[k]) << (2 * k);
#endif
        }
        ebits[r >> 3] |= (unsigned long long)(bits & valid) << ((r & 7) * 8);
#if SMAA_STORES_IN_LOOP && !(SMAA_ABL & 256)
        {
            const unsigned bits8 = bits & valid, old8 = (unsigned)(pbits[r >> 3] >> ((r & 7) * 8)) & 0xffu;
            if (byte_x < pw8) reinterpret_cast<uint8_t*>(b.bits)[(size_t)y * pw8 + byte_x] = (uint8_t)bits8;
            if ((bits8 | old8) != 0u) store_texels(y, bits8);
        }
#endif
        Lt = Lc;
        Lc = Lb;
        dyt = dyc;
        dyc = dyb;
    }
    SMAA_EP(3);
    // append: per-slot ballots rank the pixels; one atomic reserves the strip's entries
#if SMAA_ABL & 16
    if (threshold > -1.0e30f) { if (ebits[0] == 0x123456789abcdefull) screen[0] = 1; return; }   // keep the arithmetic alive, skip the append
#endif
#if SMAA_EARLY_ATOMIC
    // The strip's list entries are reserved HERE, before the plane and texel stores below: the atomic's round trip (the wave needs its
    // return value for the very last thing it does) then runs beside those stores instead of behind them.
    const bool has_edges = __ballot((ebits[0] | ebits[1]) != 0) != 0;
    const unsigned long long any_e[2] = {(ebits[0] | (ebits[0] >> 1)) & 0x5555555555555555ull, (ebits[1] | (ebits[1] >> 1)) & 0x5555555555555555ull};
    const unsigned mine_e = (unsigned)__popcll(any_e[0]) + (unsigned)__popcll(any_e[1]);
    unsigned incl_e = mine_e, base_e = 0;
    if (has_edges) {
        for (int off = 1; off < 64; off <<= 1) {
            const unsigned u = __shfl_up(incl_e, off, 64);
            if (lane >= off) incl_e += u;
        }
#if SMAA_LINEAR_STRIPS
        const unsigned strip_e = (unsigned)strip_id;
#else
        const unsigned strip_e = (blockIdx.y * gridDim.x + blockIdx.x) * WAVES_PER_WG + wave;
#endif
        if (lane == 63) base_e = atomicAdd(b.count + cur * SMAA_COUNT_SET + (strip_e % SMAA_SEGMENTS) * SMAA_COUNT_STRIDE, incl_e);
    }
#endif
    // the bit planes (dense: edge-free strips write their zeros too, so the planes need no clearing). Rows: this lane's four pixels of a
    // row are one byte, 64 lanes = 64 consecutive bytes. Columns: per 8-row block and column 16 bits, this lane's four columns = 8 bytes.
#if SMAA_ABL & 256
    if (threshold < -1.0e30f)
#endif
    {
        uint8_t* const plane = reinterpret_cast<uint8_t*>(b.bits);
#if !SMAA_STORES_IN_LOOP
        if (byte_x < pw8) {
#pragma unroll
            for (int r = 0; r < STRIP_H; r++)
                if (y0 + r < h) plane[(size_t)(y0 + r) * pw8 + byte_x] = (uint8_t)(ebits[r >> 3] >> ((r & 7) * 8));
        }
#else
        (void)plane;
#endif
#pragma unroll
        for (int blk = 0; blk < (STRIP_H + 7) / 8; blk++) {
            if (y0 + blk * 8 >= h) break;                                      // wave-uniform
            const unsigned long long eb = ebits[blk];                          // (STRIP_H = 4: the upper half of the block stays zero here,
            uint16_t col[4];                                                   //  see smaa_launch: the column plane needs strips of 8 or 16 rows)
#pragma unroll
            for (int k = 0; k < 4; k++) {
                unsigned v = 0;
#pragma unroll
                for (int r = 0; r < 8; r++) v |= (unsigned)((eb >> (r * 8 + 2 * k)) & 3ull) << (2 * r);
                col[k] = (uint16_t)v;
            }
            uint16_t* const dst = b.cbits + (size_t)((y0 >> 3) + blk) * w + px;
            if (vec_ok) {
                *reinterpret_cast<uint2*>(dst) = make_uint2((unsigned)col[0] | ((unsigned)col[1] << 16), (unsigned)col[2] | ((unsigned)col[3] << 16));
            } else {
#pragma unroll
                for (int k = 0; k < 4; k++)
                    if (px + k < w) dst[k] = col[k];
            }
        }
    }
    // The RG8 edge texels: a lane's four pixels of a row as ONE 8-byte store wherever this frame or the previous one has an edge among them
    // -- this frame's edges go in, the previous frame's come out, and the texture is exact without a clearing pass (round 2 zeroed the
    // previous list's texels with a kernel of its own: 5 us, the fixed cost of any sparse kernel here).
    SMAA_EP(4);
    SMAA_EP(5);
#if !SMAA_STORES_IN_LOOP
    if (__ballot((ebits[0] | ebits[1] | pbits[0] | pbits[1]) != 0) == 0) return;   // wave-uniform: most strips leave here
#pragma unroll
    for (int r = 0; r < STRIP_H; r++) {
        const unsigned bits8 = (unsigned)(ebits[r >> 3] >> ((r & 7) * 8)) & 0xffu, old8 = (unsigned)(pbits[r >> 3] >> ((r & 7) * 8)) & 0xffu;
        if (__ballot((bits8 | old8) != 0u) == 0) continue;                     // wave-uniform
        if ((bits8 | old8) != 0u) store_texels(y0 + r, bits8);
    }
#endif
    if (__ballot((ebits[0] | ebits[1]) != 0) == 0) return;                     // wave-uniform: nothing to append
    const unsigned long long any[2] = {(ebits[0] | (ebits[0] >> 1)) & 0x5555555555555555ull, (ebits[1] | (ebits[1] >> 1)) & 0x5555555555555555ull};
    // The list: an inclusive scan of the lanes' pixel counts ranks them (six cross-lane steps instead of a ballot per pixel slot), the last
    // lane reserves the strip's entries with ONE atomic, and every lane writes its own pixels one after the other (the list's order is free).
    const unsigned mine = (unsigned)__popcll(any[0]) + (unsigned)__popcll(any[1]);   // bit 2s of any[] set <=> pixel slot s has an edge
    unsigned incl = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned u = __shfl_up(incl, off, 64);
        if (lane >= off) incl += u;
    }
#if SMAA_LINEAR_STRIPS
    const unsigned strip = (unsigned)strip_id;
#else
    const unsigned strip = (blockIdx.y * gridDim.x + blockIdx.x) * WAVES_PER_WG + wave;
#endif
    const unsigned seg = strip % SMAA_SEGMENTS;
    unsigned base = 0;
#if SMAA_SCALAR_ATOMIC
    {
        // lane 63's inclusive sum is the strip's total; the counter's address is wave-uniform
        uint32_t* const cnt = b.count + cur * SMAA_COUNT_SET + (unsigned)__builtin_amdgcn_readfirstlane((int)seg) * SMAA_COUNT_STRIDE;
        base = (unsigned)__builtin_amdgcn_readlane((int)incl, 63);
#if !(SMAA_ABL & 128)
        asm volatile("s_nop 4\n\ts_atomic_add %0, %1, 0x0 glc\n\ts_waitcnt lgkmcnt(0)" : "+s"(base) : "s"(cnt) : "memory");
#endif
    }
#elif SMAA_EARLY_ATOMIC
    base = __shfl(base_e, 63, 64);
#else
    if (lane == 63) base = atomicAdd(b.count + cur * SMAA_COUNT_SET + seg * SMAA_COUNT_STRIDE, incl);   // lane 63's inclusive sum is the total
    base = __shfl(base, 63, 64);
#endif
    uint32_t* out = b.list + (size_t)seg * b.segment_capacity + base + (incl - mine);
    for (int half = 0; half < (STRIP_H * 4 + 31) / 32; half++) {
        unsigned long long m = any[half];
        while (m != 0ull) {                                                    // this lane's own slots (divergent, a handful at most)
            const int k = __builtin_ctzll(m), s = half * 32 + (k >> 1);
            m &= m - 1ull;
            // an entry: the pixel's index, and in bits 30 / 31 its own two edge bits (red = left edge, green = top edge), which the weight
            // kernel would otherwise fetch with a load that depends on this one (round 4; frames below 2^30 pixels: smaa_alloc)
            *out++ = (uint32_t)((size_t)(y0 + (s >> 2)) * w + px + (s & 3)) | ((uint32_t)((ebits[half] >> k) & 3ull) << 30);
        }
    }
    SMAA_EP(5);
}

// The sparse kernels see the segments as ONE list: the first wave of every workgroup scans the 64 counts (one per lane) into LDS, and
// a flat index is mapped to (segment, entry) by a six-step search of that prefix. Work is then balanced over the whole grid whatever
// the segments' individual lengths.
struct SegmentedList {
    unsigned prefix[SMAA_SEGMENTS + 1];
    unsigned wave_sum[4];
    __device__ __forceinline__ unsigned load(const uint32_t* counts)    // returns the total; all 256 threads must call it
    {
        static_assert(SMAA_SEGMENTS == 64 || SMAA_SEGMENTS == 256, "one count per lane of the first wave, or one per thread of the workgroup");
        if (SMAA_SEGMENTS == 64) {
            if (threadIdx.x < 64) {
                unsigned v = counts[threadIdx.x * SMAA_COUNT_STRIDE];
                for (int off = 1; off < 64; off <<= 1) {
                    const unsigned u = __shfl_up(v, off, 64);
                    if ((int)threadIdx.x >= off) v += u;
                }
                prefix[threadIdx.x + 1] = v;
                if (threadIdx.x == 0) prefix[0] = 0;
            }
            __syncthreads();
        } else {
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            unsigned v = counts[threadIdx.x * SMAA_COUNT_STRIDE];
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned u = __shfl_up(v, off, 64);
                if (lane >= off) v += u;
            }
            if (lane == 63) wave_sum[wave] = v;
            __syncthreads();
            unsigned add = 0;
            for (int k = 0; k < wave; k++) add += wave_sum[k];
            prefix[threadIdx.x + 1] = v + add;
            if (threadIdx.x == 0) prefix[0] = 0;
            __syncthreads();
        }
        return prefix[SMAA_SEGMENTS];
    }
    __device__ __forceinline__ unsigned locate(unsigned i, unsigned& within) const   // flat index -> segment, index within it
    {
        unsigned lo = 0;                                                  // largest seg with prefix[seg] <= i
#pragma unroll
        for (int bit = SMAA_SEGMENTS / 2; bit > 0; bit >>= 1)
            if (prefix[lo + bit] <= i) lo += bit;
        within = i - prefix[lo];
        return lo;
    }
    static constexpr uint32_t PIXEL_MASK = 0x3fffffffu;   // entry = pixel index | own edge bits << 30 (red, green)
    __device__ __forceinline__ uint32_t entry(const SmaaBuffers& b, unsigned i) const
    {
        unsigned k;
        const unsigned seg = locate(i, k);
        return b.list[(size_t)seg * b.segment_capacity + k];
    }
};

// For rtx_read_pixels only: the RG8 edge texture and the weight texture as the reference's passes would have left them, from the row plane
// (dense, one thread per pixel): edges = the plane's two bits as 0 / 255 bytes; weights = the stored texel where the plane has an edge
// pixel, zero elsewhere (which also wipes what earlier frames left behind).
__global__ __launch_bounds__(256) void smaa_expand_kernel(SmaaBuffers b)
{
    const size_t n = (size_t)b.w * b.h;
    const smaa::PlaneTex plane{b.bits, smaa::SearchPlanes::plane_words(b.w)};
    for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < n; p += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(p / (size_t)b.w), x = (int)(p - (size_t)y * b.w);
        b.edges[p] = (uint16_t)plane.raw(x, y);
        if (!plane.any(x, y)) b.blend[p] = 0u;
    }
}

// Blending weights, one thread per listed pixel. The four orthogonal searches of a pixel do not walk their edge: the number of steps comes
// from whole words of the two bit planes (smaa::SearchPlanes -- six / 24 independent loads and a few dozen integer instructions whatever
// the edge's length), and only the search's LAST fetch, the one SMAASearchLength reads, is made on the texture. Tried and dropped on the way
// (profiles/r03_smaa.txt): one wave per strip with the plane around the strip staged in LDS -- byte-exact, 5x SLOWER (256 us): a strip's
// pixels then run one after the other in one wave at two waves per SIMD, and the strips' work differs by three orders of magnitude; every
// edge fetch from the planes through a register window -- byte-exact, 2.5x slower (116 us): 25 instructions per tap instead of a load, and
// at one wave per SIMD the kernel is bound by what one wave issues; and, in round 2, four lanes per pixel on the texture path.
__global__ __launch_bounds__(256) void smaa_weights_kernel(SmaaBuffers b, int preset, unsigned cur)
{
    __shared__ SegmentedList L;
#ifdef SMAA_PHASE_TIMES
    SMAA_PH(0);
    g_smaa_ph[blockIdx.x * 4 + (threadIdx.x >> 6)][8] = 0ull;
#endif
    const unsigned n = L.load(b.count + cur * SMAA_COUNT_SET);
#ifdef SMAA_PHASE_TIMES
    SMAA_PH(5);
#endif
    if (blockIdx.x == 0 && threadIdx.x < SMAA_SEGMENTS) b.count[(cur ^ 1u) * SMAA_COUNT_SET + threadIdx.x * SMAA_COUNT_STRIDE] = 0;   // free for the next frame's appends
    const smaa::Preset P = smaa::preset_of(preset);
    const smaa::Views V{b.w, b.h, b.color, b.edges, b.blend, b.area, b.search};
    const smaa::SearchPlanes planes{b.bits, b.cbits, b.w, b.h};
    const smaa::TexEdges src{b.edges, b.w};
    const smaa::Blend B{V, P, planes, src};
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = L.entry(b, i) & SegmentedList::PIXEL_MASK;
        const int y = (int)(p / (uint32_t)b.w), x = (int)(p - (uint32_t)y * (uint32_t)b.w);
#ifdef SMAA_PHASE_TIMES
        g_smaa_ph[blockIdx.x * 4 + (threadIdx.x >> 6)][8] = 1ull;
        if (p == 0xffffffffu) return;    // (keeps p live up to here)
        SMAA_PH(6);
#endif
#if SMAA_ABL & 512
        b.blend[p] = (uint32_t)(x ^ y) | 1u;   // timing ablation: list walk + store only
#else
        b.blend[p] = B.weights(x, y);
#endif
    }
#ifdef SMAA_PHASE_TIMES
    SMAA_PH(7);
#endif
}

// The same weights with the three independent parts of a pixel's computation -- diagonal, north edge, west edge (smaa_device.h part_*) -- on
// WAVES of a workgroup at the same time (round 4; four waves: the diagonal part once per pair of diagonals, whose two results add up to it exactly). A listed pixel's weights are a chain of dependent memory round trips (step counts
// from the planes -> the search's last fetch -> the search table -> the crossing edges -> the area table, twice per edge, after up to four
// rounds of diagonal fetches): 14 us for the median wave and 22 for the slowest when one wave walks all of it, on a GPU that is otherwise idle
// -- the list of a traced 4K frame fills 1 200 waves. Each wave of a workgroup takes the same 64 list entries and ONE part; the north and west
// waves leave their two floats in LDS, the diagonal wave applies the shader's selection rule (smaa::Blend::combine) and stores the texel.
// A part the rule then discards was computed from the pass-1 textures like any other, so the bytes are those of weights().
#ifndef SMAA_ROLE_WAVES
#define SMAA_ROLE_WAVES 1
#endif
#ifndef SMAA_ROLE_GRID
#define SMAA_ROLE_GRID 2048   /* x 3 waves = 6 144 = six per SIMD: every workgroup resident at once (1 024 / 4 096: traced ULTRA 20.2 / 17.0 against 16.5 us) */
#endif
#ifndef SMAA_ROLE_PLANETEX
#define SMAA_ROLE_PLANETEX 0   /* A/B: 1 = every role reads single edge texels from the row bit plane instead of the RG8 texture, 2 = the diagonal roles only */
#endif
#ifndef SMAA_ROLE_MAX_PIXELS
#define SMAA_ROLE_MAX_PIXELS (SMAA_ROLE_GRID * 64u)   /* what the grid takes in ONE pass */
#endif
__global__ __launch_bounds__(256) void smaa_weights_roles_kernel(SmaaBuffers b, int preset, unsigned cur)
{
    __shared__ SegmentedList L;
    __shared__ smaa::F2 part[3][64];
    SMAA_RP(0);
    SMAA_RP_SET(6, 0ull);
    SMAA_RP_SET(7, 0ull);
    const unsigned n = L.load(b.count + cur * SMAA_COUNT_SET);
    SMAA_RP(1);
    if (blockIdx.x == 0 && threadIdx.x < SMAA_SEGMENTS) b.count[(cur ^ 1u) * SMAA_COUNT_SET + threadIdx.x * SMAA_COUNT_STRIDE] = 0;   // free for the next frame's appends
    const smaa::Preset P = smaa::preset_of(preset);
    const smaa::Views V{b.w, b.h, b.color, b.edges, b.blend, b.area, b.search};
    const smaa::SearchPlanes planes{b.bits, b.cbits, b.w, b.h};
    const smaa::TexEdges src{b.edges, b.w};
#if SMAA_ROLE_PLANETEX == 1
    const smaa::PlaneTex psrc{b.bits, smaa::SearchPlanes::plane_words(b.w)};
    const smaa::BlendT<smaa::PlaneTex> B{V, P, planes, psrc};
    const smaa::BlendT<smaa::PlaneTex>& BD = B;
#elif SMAA_ROLE_PLANETEX == 2
    const smaa::Blend B{V, P, planes, src};
    const smaa::PlaneTex psrc{b.bits, smaa::SearchPlanes::plane_words(b.w)};
    const smaa::BlendT<smaa::PlaneTex> BD{V, P, planes, psrc};
#else
    const smaa::Blend B{V, P, planes, src};
    const smaa::Blend& BD = B;
#endif
    if (n > SMAA_ROLE_MAX_PIXELS) {
        // A long list (a frame full of edges: 6 % of the pixels in the synthetic pattern) keeps every SIMD busy whatever the order, and then
        // the parts the selection rule discards are plain extra work (pattern ULTRA 76 -> 96 us with the roles): one thread per pixel, all
        // parts in the shader's order, as in rounds 2-3.
        for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
            const uint32_t p = L.entry(b, i) & SegmentedList::PIXEL_MASK;
            const int y = (int)(p / (uint32_t)b.w), x = (int)(p - (uint32_t)y * (uint32_t)b.w);
            b.blend[p] = B.weights(x, y);
        }
        return;
    }
    const unsigned role = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    for (unsigned base = blockIdx.x * 64u; base < n; base += gridDim.x * 64u) {   // workgroup-uniform
        const unsigned i = base + lane;
        const bool valid = i < n;
        uint32_t p = 0u;
        int x = 0, y = 0;
        smaa::F2 e{0.0f, 0.0f};
        if (valid) {
            const uint32_t raw = L.entry(b, i);
            p = raw & SegmentedList::PIXEL_MASK;
            y = (int)(p / (uint32_t)b.w);
            x = (int)(p - (uint32_t)y * (uint32_t)b.w);
            e = smaa::F2{(raw & 0x40000000u) ? 1.0f : 0.0f, (raw & 0x80000000u) ? 1.0f : 0.0f};   // = own_edges(x, y): the texels are 0 or 255
        }
        const float X = (float)x, Y = (float)y;
        smaa::F2 r{0.0f, 0.0f};
#ifdef SMAA_PHASE_TIMES
        if (e.x == 123.0f) return;       // (keeps the own-texel fetch in front of the stamp)
        SMAA_RP(2);
        SMAA_RP_SET(6, 1ull + role);
        if (__any(role == 1u ? (valid && e.y > 0.0f) : role == 2u ? (valid && e.x > 0.0f) : (valid && B.has_diag_part(e)))) SMAA_RP_SET(7, 1ull);
#endif
        if (role == 1u) {
            if (valid && e.y > 0.0f) r = B.part_north(X, Y);
            part[0][lane] = r;
        } else if (role == 2u) {
            if (valid && e.x > 0.0f) r = B.part_west(X, Y);
            part[1][lane] = r;
        } else if (role == 3u) {
            if (valid && B.has_diag_part(e)) r = BD.part_diag(X, Y, e, 2u);
            part[2][lane] = r;
        } else {
            if (valid && B.has_diag_part(e)) r = BD.part_diag(X, Y, e, 1u);
        }
#ifdef SMAA_PHASE_TIMES
        if (r.x == 123.0f) return;
        SMAA_RP(3);
#endif
        __syncthreads();
        SMAA_RP(4);
        if (role == 0u && valid) b.blend[p] = smaa::Blend::combine(e, P.max_steps_diag > 0, smaa::F2{r.x + part[2][lane].x, r.y + part[2][lane].y}, part[0][lane], part[1][lane]);
        __syncthreads();
    }
    SMAA_RP(5);
}

__global__ __launch_bounds__(256) void smaa_blend_kernel(SmaaBuffers b, unsigned cur)
{
    __shared__ SegmentedList L;
    const unsigned n = L.load(b.count + cur * SMAA_COUNT_SET);
    const smaa::Views V{b.w, b.h, b.color, b.edges, b.blend, b.area, b.search};
    const smaa::PlaneTex member{b.bits, smaa::SearchPlanes::plane_words(b.w)};
    // three candidates per listed pixel: itself, its left and its lower neighbour (a pixel's weights come from its own weight texel,
    // its right neighbour's alpha and its upper neighbour's green; only listed pixels have non-zero weight texels). A pixel reached
    // twice gets the same bytes twice.
    const unsigned items = n * 3u;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
        const unsigned which = i / n;                                           // all "itself" first, then the neighbours: coalesced list reads
        const uint32_t p = L.entry(b, i - which * n) & SegmentedList::PIXEL_MASK;
        int y = (int)(p / (uint32_t)b.w), x = (int)(p - (uint32_t)y * (uint32_t)b.w);
        if (which == 1) x -= 1;
        if (which == 2) y -= 1;
        if (x < 0 || y < 0) continue;
        uint32_t out;
        if (smaa::neighborhood<true>(V, x, y, out, &member)) b.screen[(size_t)y * b.w + x] = out;
    }
}

}  // namespace

int smaa_strip_count(int w, int h, int strip_h)
{
    const int gx = (w + STRIP_W * WAVES_PER_WG - 1) / (STRIP_W * WAVES_PER_WG), gy = (h + strip_h - 1) / strip_h;
    return gx * gy * WAVES_PER_WG;
}
static int strip_rows()
{
    static const int v = [] { const char* e = getenv("RTX_SMAA_STRIP_H"); const int x = e ? atoi(e) : STRIP_H_DEFAULT; return (x == 8 || x == 16) ? x : STRIP_H_DEFAULT; }();   // A/B knob (the column bit plane is written per block of 8 rows)
    return v;
}
size_t smaa_segment_capacity(int w, int h)
{
    const int strip_h = strip_rows();
    const int strips = smaa_strip_count(w, h, strip_h);
    return (size_t)((strips + SMAA_SEGMENTS - 1) / SMAA_SEGMENTS) * (size_t)(STRIP_W * strip_h);
}

size_t smaa_plane_bytes(int w, int h) { return (size_t)smaa::SearchPlanes::plane_words(w) * 8u * (size_t)h; }
size_t smaa_col_plane_bytes(int w, int h) { return (size_t)((h + 7) / 8) * (size_t)w * 2u + 16u; }

hipError_t smaa_expand(const SmaaBuffers& b, hipStream_t stream)
{
    hipLaunchKernelGGL(smaa_expand_kernel, dim3(2048), dim3(256), 0, stream, b);
    return hipGetLastError();
}

hipError_t smaa_launch(const SmaaBuffers& b, int preset, unsigned frame, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    const unsigned cur = frame & 1u;
    const dim3 sparse(1024);                                                    // grid-stride over the device-side total of the segment counts
    const int strip_h = strip_rows();
#if SMAA_LINEAR_STRIPS
#if SMAA_XCD_BANDS
    const dim3 grid((((((b.w + STRIP_W - 1) / STRIP_W) * ((b.h + strip_h - 1) / strip_h) + WAVES_PER_WG - 1) / WAVES_PER_WG) + 7) / 8 * 8);
#else
    const dim3 grid((((b.w + STRIP_W - 1) / STRIP_W) * ((b.h + strip_h - 1) / strip_h) + WAVES_PER_WG - 1) / WAVES_PER_WG);
#endif
#else
    const dim3 grid((b.w + STRIP_W * WAVES_PER_WG - 1) / (STRIP_W * WAVES_PER_WG), (b.h + strip_h - 1) / strip_h);
#endif
    const float thr = smaa::preset_of(preset).threshold;
    const bool vec = (b.w & 3) == 0 && b.w >= 4;
    const dim3 wg(64 * WAVES_PER_WG);
    if (strip_h == 8 && vec) hipExtLaunchKernelGGL((smaa_edges_kernel<8, true>), grid, wg, 0, stream, ev_start, nullptr, 0, b, thr, cur);
    else if (strip_h == 8) hipExtLaunchKernelGGL((smaa_edges_kernel<8, false>), grid, wg, 0, stream, ev_start, nullptr, 0, b, thr, cur);
    else if (vec) hipExtLaunchKernelGGL((smaa_edges_kernel<16, true>), grid, wg, 0, stream, ev_start, nullptr, 0, b, thr, cur);
    else hipExtLaunchKernelGGL((smaa_edges_kernel<16, false>), grid, wg, 0, stream, ev_start, nullptr, 0, b, thr, cur);
#if SMAA_ROLE_WAVES
    hipLaunchKernelGGL(smaa_weights_roles_kernel, dim3(SMAA_ROLE_GRID), dim3(256), 0, stream, b, preset, cur);
#else
    hipLaunchKernelGGL(smaa_weights_kernel, sparse, dim3(256), 0, stream, b, preset, cur);
#endif
    hipExtLaunchKernelGGL(smaa_blend_kernel, sparse, dim3(256), 0, stream, nullptr, ev_stop, 0, b, cur);
    return hipGetLastError();
}
