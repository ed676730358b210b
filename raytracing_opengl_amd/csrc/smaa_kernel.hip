// smaa_kernel.hip -- the SMAA post-process as gfx950 kernels.
//
// The reference runs three full-screen fragment passes through three 8-bit render targets (GLWrapper.cpp:173-204): every pass
// reads and writes whole frames although edges -- the only pixels the second and third pass do anything for -- are a few per cent
// of a frame. Here the frame is touched densely ONCE and the rest is sparse:
//
//   smaa_edges_kernel    dense: reads the colour target (16 B per lane, 1 KiB per wave and row), copies it to the screen, detects luma
//                        edges (SMAA.h:689-741) on a four-row window of lumas held in registers and, for edge pixels only, writes the
//                        RG8 edge texel and appends the pixel to a list -- one atomic per 256 x 8 strip (ballot ranks), none for the
//                        strips without an edge;
//   smaa_weights_kernel  over the list: blending weights (SMAA.h:1145-1243) -> RGBA8 weight texel of that pixel;
//   smaa_blend_kernel    over the list: neighbourhood blending (SMAA.h:1252-1300) of the listed pixel, its left and its lower
//                        neighbour -- the only pixels whose four weights can be non-zero -- overwriting their screen texels;
//   smaa_clear_kernel    over the PREVIOUS frame's list: zeroes the edge and weight texels it wrote.
//
// Invariant: the edge and weight textures are zero everywhere except at the pixels of the current list (allocated zeroed, cleared
// through the list before the next frame's pass 1), so the sparse passes see exactly the textures the reference's dense passes
// would have produced (glClear(0) + discard, GLWrapper.cpp:177-178,189-190). Algorithmic HBM traffic per frame: W*H*4 B read +
// W*H*4 B written, against 6 x W*H*4 B + 2 x W*H*2 B for three dense passes.
//
// Counters: two, used alternately. Frame f appends to count[f & 1]; smaa_clear of frame f walks the list with count[(f-1) & 1]
// entries; smaa_weights of frame f, the first kernel after which nobody needs it any more, zeroes count[(f-1) & 1] for frame f+1.
#include "smaa_kernel.h"

#include "smaa_device.h"

namespace {

constexpr int STRIP_W = 256;     // pixels per wave and row: 64 lanes x 4 pixels = one 1 KiB row segment per load
constexpr int STRIP_H = 8;       // rows per wave: 8 + 3 halo rows are read for 8 rows of output
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
// ONE atomic (none at all for the strips without an edge -- most of a frame), ranks its pixels with ballots and writes list + edge texels.
__global__ __launch_bounds__(64 * WAVES_PER_WG) void smaa_edges_kernel(SmaaBuffers b, float threshold, unsigned cur)
{
    const int w = b.w, h = b.h;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x0 = (blockIdx.x * WAVES_PER_WG + wave) * STRIP_W, y0 = blockIdx.y * STRIP_H;
    if (x0 >= w) return;                                                       // wave-uniform
    const int px = x0 + lane * 4;
    const bool vec_ok = ((w & 3) == 0) && (px + 3 < w);

    auto load_row = [&](int y, uint32_t c[4]) {                               // clamped in y; x clamped per pixel on the scalar path
        const int yc = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
        if (vec_ok) {
            const uint4 v = *reinterpret_cast<const uint4*>(b.color + (size_t)yc * w + px);
            c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++) c[k] = load_px(b.color, w, h, px + k, yc);
        }
    };
    auto lumas = [&](const uint32_t c[4]) {
        Row4 r;
#pragma unroll
        for (int k = 0; k < 4; k++) r.l[k] = smaa::luma_of(c[k]);
        return r;
    };

    uint32_t c[4];
    load_row(y0 - 2, c);
    Row4 Ltt = lumas(c);
    load_row(y0 - 1, c);
    Row4 Lt = lumas(c);
    uint32_t cc[4];
    load_row(y0, cc);
    Row4 Lc = lumas(cc);
    unsigned long long ebits = 0;                                              // 2 bits (R, G) per pixel: bit (row * 4 + k) * 2
    for (int r = 0; r < STRIP_H; r++) {
        const int y = y0 + r;
        if (y >= h) break;                                                     // wave-uniform
        uint32_t cb[4];
        load_row(y + 1, cb);
        const Row4 Lb = lumas(cb);
        // the dense copy: pass 3 for every pixel without weights
        if (vec_ok) {
            *reinterpret_cast<uint4*>(b.screen + (size_t)y * w + px) = make_uint4(cc[0], cc[1], cc[2], cc[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (px + k < w) b.screen[(size_t)y * w + px + k] = cc[k];
        }
        // horizontal neighbours of this row: two to the left of pixel 0, one to the right of pixel 3
        float left1 = __shfl_up(Lc.l[3], 1, 64), left2 = __shfl_up(Lc.l[2], 1, 64), right = __shfl_down(Lc.l[0], 1, 64);
        if (lane == 0) {
            left1 = smaa::luma_of(load_px(b.color, w, h, px - 1, y));
            left2 = smaa::luma_of(load_px(b.color, w, h, px - 2, y));
        }
        if (lane == 63 || px + 4 >= w) right = smaa::luma_of(load_px(b.color, w, h, px + 4, y));
        const float row[7] = {left2, left1, Lc.l[0], Lc.l[1], Lc.l[2], Lc.l[3], right};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            if (px + k < w) {
                const uint32_t e = smaa::edge_from_lumas(threshold, row[k + 2], row[k + 1], Lt.l[k], row[k + 3], Lb.l[k], row[k], Ltt.l[k]);
                const unsigned long long two = (unsigned long long)((e & 1u) | ((e >> 7) & 2u));   // RG8 texel 0x00ff / 0xff00 -> bits 0 / 1
                ebits |= two << ((r * 4 + k) * 2);
            }
        }
        Ltt = Lt;
        Lt = Lc;
        Lc = Lb;
#pragma unroll
        for (int k = 0; k < 4; k++) cc[k] = cb[k];
    }
    // append: per-slot ballots rank the pixels; one atomic reserves the strip's entries
    if (__ballot(ebits != 0) == 0) return;                                     // wave-uniform: most strips leave here
    unsigned total = 0;
    unsigned long long any = ebits | (ebits >> 1);                             // bit 2s set <=> pixel slot s has an edge
    {
        unsigned mine = (unsigned)__popcll(any & 0x5555555555555555ull);
        for (int off = 32; off > 0; off >>= 1) mine += __shfl_xor(mine, off, 64);
        total = mine;
    }
    unsigned base = 0;
    if (lane == 0) base = atomicAdd(b.count + cur, total);
    base = __shfl(base, 0, 64);
    unsigned before = 0;
    for (int s = 0; s < STRIP_H * 4; s++) {
        const bool has = (any >> (2 * s)) & 1ull;
        const unsigned long long bal = __ballot(has);
        if (bal == 0) continue;                                                // wave-uniform
        if (has) {
            const unsigned rank = before + (unsigned)__popcll(bal & ((1ull << lane) - 1ull));
            const int r = s >> 2, k = s & 3;
            const uint32_t p = (uint32_t)((size_t)(y0 + r) * w + px + k);
            const unsigned two = (unsigned)(ebits >> (2 * s)) & 3u;
            b.list[base + rank] = p;
            b.edges[p] = (uint16_t)(((two & 1u) ? 0x00ffu : 0u) | ((two & 2u) ? 0xff00u : 0u));
        }
        before += (unsigned)__popcll(bal);
    }
}

__global__ __launch_bounds__(256) void smaa_clear_kernel(SmaaBuffers b, unsigned prev)
{
    const unsigned n = b.count[prev];
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = b.list[i];
        b.edges[p] = 0;
        b.blend[p] = 0;
    }
}

__global__ __launch_bounds__(256) void smaa_weights_kernel(SmaaBuffers b, int preset, unsigned cur)
{
    const unsigned n = b.count[cur];
    if (blockIdx.x == 0 && threadIdx.x == 0) b.count[cur ^ 1u] = 0;            // free for the next frame's appends (see the header comment)
    const smaa::Preset P = smaa::preset_of(preset);
    const smaa::Views V{b.w, b.h, b.color, b.edges, b.blend, b.area, b.search};
    const smaa::Blend B{V, P};
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const uint32_t p = b.list[i];
        const int y = (int)(p / (uint32_t)b.w), x = (int)(p - (uint32_t)y * (uint32_t)b.w);
        b.blend[p] = B.weights(x, y);
    }
}

__global__ __launch_bounds__(256) void smaa_blend_kernel(SmaaBuffers b, unsigned cur)
{
    const unsigned n = b.count[cur];
    const smaa::Views V{b.w, b.h, b.color, b.edges, b.blend, b.area, b.search};
    // three candidates per listed pixel: itself, its left and its lower neighbour (a pixel's weights come from its own weight texel,
    // its right neighbour's alpha and its upper neighbour's green; only listed pixels have non-zero weight texels). A pixel reached
    // twice gets the same bytes twice.
    const unsigned items = n * 3u;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < items; i += gridDim.x * blockDim.x) {
        const unsigned which = i / n;                                           // all "itself" first, then the neighbours: coalesced list reads
        const uint32_t p = b.list[i - which * n];
        int y = (int)(p / (uint32_t)b.w), x = (int)(p - (uint32_t)y * (uint32_t)b.w);
        if (which == 1) x -= 1;
        if (which == 2) y -= 1;
        if (x < 0 || y < 0) continue;
        uint32_t out;
        if (smaa::neighborhood(V, x, y, out)) b.screen[(size_t)y * b.w + x] = out;
    }
}

}  // namespace

hipError_t smaa_launch(const SmaaBuffers& b, int preset, unsigned frame, hipStream_t stream)
{
    const unsigned cur = frame & 1u, prev = cur ^ 1u;
    const int sparse_blocks = 1024;                                             // grid-stride over a device-side count
    hipLaunchKernelGGL(smaa_clear_kernel, dim3(sparse_blocks), dim3(256), 0, stream, b, prev);
    const dim3 grid((b.w + STRIP_W * WAVES_PER_WG - 1) / (STRIP_W * WAVES_PER_WG), (b.h + STRIP_H - 1) / STRIP_H);
    hipLaunchKernelGGL(smaa_edges_kernel, grid, dim3(64 * WAVES_PER_WG), 0, stream, b, smaa::preset_of(preset).threshold, cur);
    hipLaunchKernelGGL(smaa_weights_kernel, dim3(sparse_blocks), dim3(256), 0, stream, b, preset, cur);
    hipLaunchKernelGGL(smaa_blend_kernel, dim3(sparse_blocks), dim3(256), 0, stream, b, cur);
    return hipGetLastError();
}
