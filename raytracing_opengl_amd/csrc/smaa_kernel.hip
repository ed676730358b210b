// smaa_kernel.hip -- the SMAA post-process as gfx950 kernels.
//
// The reference runs three full-screen fragment passes through three 8-bit render targets (GLWrapper.cpp:173-204): every pass
// reads and writes whole frames although edges -- the only pixels the second and third pass do anything for -- are a few per cent
// of a frame. Here the frame is touched densely ONCE and the rest is sparse:
//
//   smaa_edges_kernel    dense: reads the colour target (16 B per lane), copies it to the screen, detects luma edges on an LDS
//                        tile of lumas (SMAA.h:689-741) and, for edge pixels only, writes the RG8 edge texel and appends the pixel
//                        to a list -- one atomic per wave (ballot + prefix count), none for the waves without an edge;
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

constexpr int TILE_W = 64, TILE_H = 16;          // pixels per workgroup: 16 lanes x 4 pixels wide, 16 rows
constexpr int LDS_W = TILE_W + 3, LDS_H = TILE_H + 3;   // + 2 left / lower, + 1 right / upper

__device__ __forceinline__ uint32_t load_px(const uint32_t* color, int w, int h, int x, int y)
{
    x = x < 0 ? 0 : (x > w - 1 ? w - 1 : x);    // CLAMP_TO_EDGE
    y = y < 0 ? 0 : (y > h - 1 ? h - 1 : y);
    return color[(size_t)y * w + x];
}

__global__ __launch_bounds__(256) void smaa_edges_kernel(SmaaBuffers b, float threshold, unsigned cur)
{
    __shared__ float L[LDS_H][LDS_W];
    const int w = b.w, h = b.h;
    const int x0 = blockIdx.x * TILE_W, y0 = blockIdx.y * TILE_H;
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int px = x0 + tx * 4, py = y0 + ty;
    uint32_t c[4];
    const bool row_ok = py < h;
    const bool vec = row_ok && (px + 3 < w) && ((w & 3) == 0);
    if (vec) {
        const uint4 v = *reinterpret_cast<const uint4*>(b.color + (size_t)py * w + px);
        c[0] = v.x; c[1] = v.y; c[2] = v.z; c[3] = v.w;
        *reinterpret_cast<uint4*>(b.screen + (size_t)py * w + px) = v;          // the dense copy: pass 3 for every pixel without weights
    } else {
#pragma unroll
        for (int k = 0; k < 4; k++) {
            c[k] = load_px(b.color, w, h, px + k, py);
            if (row_ok && px + k < w) b.screen[(size_t)py * w + px + k] = c[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 4; k++) L[ty + 2][tx * 4 + 2 + k] = smaa::luma_of(c[k]);
    // halo: rows y0-2, y0-1, y0+TILE_H over all LDS_W columns; columns x0-2, x0-1, x0+TILE_W over the tile rows (249 cells)
    {
        const int t = threadIdx.x;
        int lx = -1, ly = -1;
        if (t < 3 * LDS_W) {
            const int r = t / LDS_W;
            lx = t - r * LDS_W;
            ly = r < 2 ? r : LDS_H - 1;
        } else if (t < 3 * LDS_W + 3 * TILE_H) {
            const int u = t - 3 * LDS_W, r = u / 3, cidx = u - r * 3;
            ly = r + 2;
            lx = cidx < 2 ? cidx : LDS_W - 1;
        }
        if (lx >= 0) L[ly][lx] = smaa::luma_of(load_px(b.color, w, h, x0 + lx - 2, y0 + ly - 2));
    }
    __syncthreads();
    uint32_t e[4];
    unsigned mask = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int lx = tx * 4 + 2 + k, ly = ty + 2;
        const bool real = row_ok && (px + k < w);
        e[k] = real ? smaa::edge_from_lumas(threshold, L[ly][lx], L[ly][lx - 1], L[ly - 1][lx], L[ly][lx + 1], L[ly + 1][lx], L[ly][lx - 2], L[ly - 2][lx]) : 0u;
        if (e[k]) mask |= 1u << k;
    }
    // append the edge pixels of this wave: per-slot ballots give every lane its rank without a scan
    unsigned long long bal[4];
    unsigned total = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) { bal[k] = __ballot((mask >> k) & 1u); total += (unsigned)__popcll(bal[k]); }
    if (total == 0) return;                                                    // wave-uniform: most waves leave here
    unsigned base = 0;
    const int lane = threadIdx.x & 63;
    if (lane == 0) base = atomicAdd(b.count + cur, total);
    base = __shfl(base, 0, 64);
    unsigned before = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if ((mask >> k) & 1u) {
            const unsigned rank = before + (unsigned)__popcll(bal[k] & ((1ull << lane) - 1ull));
            const uint32_t p = (uint32_t)((size_t)py * w + px + k);
            b.list[base + rank] = p;
            b.edges[p] = (uint16_t)e[k];
        }
        before += (unsigned)__popcll(bal[k]);
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
    const dim3 grid((b.w + TILE_W - 1) / TILE_W, (b.h + TILE_H - 1) / TILE_H);
    hipLaunchKernelGGL(smaa_edges_kernel, grid, dim3(256), 0, stream, b, smaa::preset_of(preset).threshold, cur);
    hipLaunchKernelGGL(smaa_weights_kernel, dim3(sparse_blocks), dim3(256), 0, stream, b, preset, cur);
    hipLaunchKernelGGL(smaa_blend_kernel, dim3(sparse_blocks), dim3(256), 0, stream, b, cur);
    return hipGetLastError();
}
