// rt_kernel.hip -- gfx950 kernels of the tracer and their launch function.
//
// Replaces glDrawArrays(GL_TRIANGLES,0,6) of the full-screen quad + the fragment program
// (reference GLWrapper.cpp:155-165, assets/shaders/quad.vert, rt.frag).
//
// Mapping (DESIGN.md "Kernel"): one pixel per lane; one wave64 = an 8x8 pixel tile whose 2x2
// pixel quads sit in 4 consecutive lanes; one 256-thread workgroup = four tiles side by side =
// 32x8 pixels; grid = ceil(W/32) x rows/8. Each lane stores one 16-byte RGBA32F pixel, so the 8
// lanes of a tile row write 128 contiguous bytes (and 4 tiles of a workgroup 512 B per image row).
// Scene tables are read with wave-uniform addresses: scalar (SMEM) loads from the DevScene blob,
// or LDS broadcast reads when the blob is staged per workgroup (LDS template flag).
#include <hip/hip_runtime.h>

#include "rt_device.h"
#include "rt_kernel.h"

#include <hip/hip_ext.h>

using namespace rtdev;

#ifdef RT_DK_STATS
extern "C" __attribute__((visibility("default"))) int rtx_debug_dk_stats(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dk), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[16] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_dk), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif
#ifdef RT_SCAN_STATS
extern "C" __attribute__((visibility("default"))) int rtx_debug_scan_stats(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_scan), sizeof(unsigned long long) * 32) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_scan), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif
#ifdef RT_WG_TIMES
// diagnostic build: per-workgroup start time and duration (s_memrealtime: 10 ns ticks, device-wide clock) of the last launch, to see which tiles are the
// long pole of a launch (tools/wg_times.py)
__device__ unsigned long long g_wg_start[1 << 16], g_wg_dur[1 << 16];
extern "C" __attribute__((visibility("default"))) int rtx_debug_wg_times(unsigned long long* start, unsigned long long* dur, int n)
{
    if (hipMemcpyFromSymbol(start, HIP_SYMBOL(g_wg_start), sizeof(unsigned long long) * n) != hipSuccess) return 1;
    if (hipMemcpyFromSymbol(dur, HIP_SYMBOL(g_wg_dur), sizeof(unsigned long long) * n) != hipSuccess) return 1;
    return 0;
}
#endif
#ifdef RT_PHASE_TIMERS
__device__ unsigned long long g_phase[PH_COUNT];
extern "C" __attribute__((visibility("default"))) int rtx_debug_phase_counters(unsigned long long* out, int reset)
{
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_phase), sizeof(unsigned long long) * PH_COUNT) != hipSuccess) return 1;
    if (reset) {
        unsigned long long z[PH_COUNT] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_phase), z, sizeof z) != hipSuccess) return 1;
    }
    return 0;
}
#endif

namespace {

__device__ __forceinline__ uint32_t pack_rgba8(f4 c)
{
    // GL fixed-point write-out: clamp to [0,1], scale by 255, round to nearest; NaN -> 0
    auto q = [](float v) -> uint32_t {
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        if (!(v == v)) v = 0.0f;
        return (uint32_t)(v * 255.0f + 0.5f);
    };
    return q(c.x) | (q(c.y) << 8) | (q(c.z) << 16) | (q(c.w) << 24);
}

#ifndef RT_WAVES_PER_EU
#define RT_WAVES_PER_EU 6
#endif
#ifndef RT_WPE_HEAVY
#define RT_WPE_HEAVY 6
#endif
#ifndef RT_CONST_SCENE
#define RT_CONST_SCENE 1
#endif
// Two register budgets of the same code (WPE = waves per SIMD the compiler must make room for; numbers for 4K frames,
// built with -mllvm -disable-machine-licm, see the Makefile):
//   WPE = RT_WAVES_PER_EU (6: 80 VGPRs, path state in LDS, 44 B of scratch per lane around the torus solver's register
//         peak) -- the default: 500 us on the default scene (5 waves, no scratch at all: 523 us; 4 waves: 585 us).
//   HEAVY (WPE = RT_WPE_HEAVY) -- the variant for scenes with many primitives: the candidate tables of rt_device.h (group culls, ray
//         pencils, slab tables) are compiled in. Round 1 ran it at 7 waves (72 VGPRs, 96 B): every ray walked long tables of scalar
//         loads and latency hiding was worth more than the spills (quadric-heavy 4K 2530 -> 2440 us). With the tables the scans are
//         short and carry more state: 6 waves (80 VGPRs, 80 B then, 64 B in the final kernel) 1297 / 2271 us (quadric / torus), 7 (128 B) 1327 / 2266, 5 (no
//         scratch) 1303 / 2295, 8 (188 B) 1476 / 2298. Chosen at launch from the primitive count (RTX_OPT_HIGH_OCCUPANCY).

constexpr bool ps_wide(int wpe) { return wpe <= 6; }   // 6 workgroups x 24 KB fit the CU's 160 KB of LDS, 7 do not

// HEAVY: the many-primitive variant -- its own register budget and the candidate tables of rt_device.h (group culls, ray pencils, slabs)
// SKYLOD: the mip-mapped sky box (GLWrapper::load_cubemap(faces, true)); instantiated for the scalar-load variants only
template <bool CULL, bool COUNT, bool LDS, int WPE, bool HEAVY = false, bool SKYLOD = false>
__global__ __launch_bounds__(256, WPE) void rt_trace_kernel(const RtLaunchParams p)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];

    // Workgroup -> image tile. The dispatcher places workgroup b on XCD b % 8 (observed, used for speed
    // only), and each XCD has its own L2: with the plain row-major order the eight neighbours of a
    // tile sit on eight different L2s. The optional remap (RTX_OPT_XCD_REMAP) hands each XCD whole
    // 4x4-workgroup super-tiles (128x32 px), dealt round-robin so that sky and object regions stay
    // balanced. Measured on the 4K default scene it does not pay (FETCH_SIZE 96.3 vs 97.5 MB, kernel
    // 0.939 vs 0.902 ms: the texture footprint of a tile is small and mostly served by the 256 MiB
    // Infinity Cache), so row-major order is the default.
    int bx, by;
    if (p.xcd_remap) {
        const int b = blockIdx.x;
        const int xcd = b & 7, j = b >> 3;
        const int st = j >> 4, o = j & 15;                 // super-tile index within this XCD, workgroup inside it
        const int g = st * 8 + xcd;                        // global super-tile index
        const int sx = g % p.st_nx, sy = g / p.st_nx;
        bx = sx * 4 + (o & 3);
        by = sy * 4 + (o >> 2);
        if (sy >= p.st_ny || bx >= p.grid_x || by >= p.grid_y) return;  // whole workgroup: no barrier follows
    } else {
        bx = blockIdx.x;
        by = blockIdx.y;
        // "Longest first" for small launches: the workgroup rows that show a torus are dispatched before the others.
        // Workgroups that run the quartic solver in several scans per pixel last ~20x the median (170 us against 9 us in the
        // 4K default frame, tools/wg_times.py) and sit in the middle rows; when a launch is one GPU's quarter or eighth of
        // the frame they ARE its tail (one rank's share of 8: 166-205 us in row order, 157-173 us hot rows first, ideal 74).
        // Not for large launches: there any re-ordering costs more than the tail it saves (593 -> 609-643 us for the whole
        // frame; neighbouring rows run the same code, mixed rows thrash the instruction cache). Which rows: rtx_capi.cpp.
        if (p.hot_rows > 0) {
            const int k = by;
            by = k < p.hot_rows ? p.hot_row0 + k : (k - p.hot_rows < p.hot_row0 ? k - p.hot_rows : k);
            // (Round 5 also ran these waves at s_setprio 3 -- a torus tile's wave is one serial chain of solver runs, and it shares its SIMD's issue
            // slots with five sky waves: no gain at 640x480 ... 4K nor for a rank's share of a frame, profiles/r05e_hot_rows_setprio_*.txt. One
            // thing learnt on the way: __builtin_amdgcn_s_setprio, or a volatile asm, ANYWHERE in the kernel -- even on a path never taken --
            // counts as a possible store, after which no load of the scene blob is "known unclobbered" and every scalar load of the kernel
            // becomes a vector load: s_load 222 -> 30, scratch 44 -> 92 B, the 4K frame 0.46 -> 0.66 ms
            // (profiles/r05d_setprio_builtin_clobbers_scalar_loads.txt). asm("s_setprio 3" : "+s"(by)) does not. Round 4's "s_setprio at 4K:
            // 680 us" was most likely that effect, not the priority.)
        }
    }
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
#ifdef RT_WG_TIMES
    const unsigned long long _wg_t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz, one clock for the whole device: start times of different XCDs compare
    const int _wg_id = (by * (int)gridDim.x + bx) & 0xffff;
    if (threadIdx.x == 0) { g_wg_start[_wg_id] = _wg_t0; g_wg_dur[_wg_id] = 0ull; }
#endif
    // 2x2 quads in consecutive lanes: bit0 = x&1, bit1 = y&1, bits 2-3 = quad column, bits 4-5 = quad row
    const int tx = ((lane >> 2) & 3) * 2 + (lane & 1);
    const int ty = ((lane >> 4) & 3) * 2 + ((lane >> 1) & 1);
    const int x = bx * 32 + wave * 8 + tx;
    const int row_local = by * 8 + ty;  // row inside this launch's packed band set
    // band_rows is a multiple of 8, so the eight rows of a workgroup lie in ONE band: the band index is
    // wave-uniform (scalar division), not a per-lane integer division
    const int band_j = (by * 8) / p.band_rows;
    const int y0_wg = (p.band_first + band_j * p.band_stride) * p.band_rows + (by * 8 - band_j * p.band_rows);
    const int y = y0_wg + ty;
    const bool real = (x < p.fb_w) && (y < p.fb_h) && (row_local < p.rows_local);
    // With quad-derivative LOD the pixels that complete a 2x2 quad beyond an odd-sized framebuffer run
    // as helper invocations (traced, never stored or counted), like a rasteriser's helper lanes.
    const bool helper = p.tex.lod != 0 && !real && (x < ((p.fb_w + 1) & ~1)) && (y < ((p.fb_h + 1) & ~1)) &&
                        (row_local < p.rows_local + (p.fb_h & 1));
    const bool alive = real || helper;

    const char* blob = p.scene;
    if (LDS) {
        // cooperative 16 B/lane copy of the whole DevScene blob
        const int n16 = p.scene_bytes >> 4;
        const float4* src = reinterpret_cast<const float4*>(p.scene);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int k = threadIdx.x; k < n16; k += 256) dst[k] = src[k];
        __syncthreads();
        blob = smem;
    }
    // The header (counts, camera, array offsets) is read from the blob with scalar loads where it is used.
    // A copy in the kernel arguments was measured slower: the compiler keeps all of it in SGPRs, runs out,
    // and parks the excess in VGPR lanes -- v_writelane/v_readlane are VALU issue slots, and VALU issue
    // is what bounds this kernel (4K default scene: 765 -> 741 us without the copy).
    // The scene blob, its header and the pencil masks are read-only for the whole launch: handed on as pointers into the CONSTANT address space
    // (cast there and back; the compiler's address-space inference sees through the round trip), every wave-uniform load of them is a scalar
    // load by definition. As plain global pointers they are scalar only where the compiler can prove that no store in the kernel may have
    // clobbered them -- a proof it gives up on beyond a hundred stores or at the first barrier (round 5: a pooled-solver variant's s_load count fell
    // from 305 to 42, profiles/r05s_torus_pool_ab.txt; most likely every "one harmless line made the kernel 40 % slower" of rounds 2 - 5 was this).
#if defined(__HIP_DEVICE_COMPILE__) && RT_CONST_SCENE
    typedef const __attribute__((address_space(4))) char* rt_const_ptr;
    rt_const_ptr scene_k = (rt_const_ptr)p.scene, masks_k = (rt_const_ptr)(const char*)p.pencil_masks;
    asm("" : "+s"(scene_k), "+s"(masks_k));   // (opaque: otherwise the cast there and back is folded away before the inference runs)
    const char* const scene_c = (const char*)scene_k;
    const uint32_t* const masks_c = (const uint32_t*)(const char*)masks_k;
    if (!LDS) blob = scene_c;
    const SceneView S = make_view(blob, reinterpret_cast<const DevSceneHeader*>(scene_c), masks_c);
#else
    const SceneView S = make_view(blob, reinterpret_cast<const DevSceneHeader*>(p.scene), p.pencil_masks);
#endif

    LaneCounters cnt = {};
#ifdef RT_PHASE_TIMERS
    const unsigned long long _k0 = clock64();
#endif
    // path state in LDS: dword columns of 256 lanes + one pad column for PathStore::fence -- 24 KB per workgroup in the WIDE
    // layout (up to 6 waves/SIMD), 20 KB otherwise (rt_device.h)
    constexpr bool WIDE = ps_wide(WPE);
    __shared__ float path_lds[(rtdev::path_slots(WIDE) + 1) * RT_PS_STRIDE];
    rtdev::PathStore path;
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass of this file sees the array-backed PathStore of the host build)
    path.base = path_lds + threadIdx.x;
    path.fence_slot = p.ps_fence_slot;
#endif
    const f4 px = trace_pixel<CULL, COUNT, WIDE, HEAVY, SKYLOD>(S, p.tex, path, alive, (float)x + 0.5f, (float)y + 0.5f, cnt);   // group culls: the many-primitive variant only

    // The pixel's coordinates are needed again only here. They are RE-DERIVED from the thread index
    // (laundered through an empty asm so the compiler cannot keep the first copy alive) instead of
    // occupying VGPRs -- or scratch spill slots -- for the whole trace.
    unsigned tid2 = threadIdx.x;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, wave2 = tid2 >> 6;
    const int x2 = bx * 32 + wave2 * 8 + ((lane2 >> 2) & 3) * 2 + (lane2 & 1);
    const int row2 = by * 8 + ((lane2 >> 4) & 3) * 2 + ((lane2 >> 1) & 1);
    const int y2 = y0_wg + ((lane2 >> 4) & 3) * 2 + ((lane2 >> 1) & 1);
    const bool real2 = (x2 < p.fb_w) && (y2 < p.fb_h) && (row2 < p.rows_local);
    if (!real2) cnt = LaneCounters{};
    if (real2) {
        const size_t idx = (size_t)row2 * (size_t)p.fb_w + (size_t)x2;
        if (p.out_f32) {
            typedef float v4f __attribute__((ext_vector_type(4)));
            const v4f v = {px.x, px.y, px.z, px.w};
            __builtin_nontemporal_store(v, reinterpret_cast<v4f*>(p.out_f32) + idx);
        }
        // The 8-bit target is what the SMAA resolve reads right behind this kernel (33 MB at 4K): an ordinary store leaves it in the last-level
        // cache for that reader -- resolve inside rtx_draw 56.6 -> 51.2 us, the trace itself unchanged (profiles/r04_u8_plain_store.txt); the
        // float target above (133 MB, no reader on this device) stays non-temporal.
        if (p.out_u8) p.out_u8[idx] = pack_rgba8(px);
    }
#ifdef RT_PHASE_TIMERS
    if (lane == 0) {
        for (int k = 0; k < PH_COUNT - 2; k++) atomicAdd(&g_phase[k], cnt.pc.acc[k]);
        atomicAdd(&g_phase[PH_COUNT - 2], (unsigned long long)clock64() - _k0);  // whole wave
        atomicAdd(&g_phase[PH_COUNT - 1], 1ull);                                // waves
    }
#endif
#ifdef RT_WG_TIMES
    if ((threadIdx.x & 63) == 0) atomicMax(&g_wg_dur[_wg_id], (unsigned long long)__builtin_amdgcn_s_memrealtime() - _wg_t0);
#endif
    if (COUNT) {
        uint32_t v[4] = {cnt.closest, cnt.shadow_ref, cnt.shadow_cast, cnt.torus_solves};
#pragma unroll
        for (int k = 0; k < 4; k++) {
            uint32_t s = v[k];
            for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
            if (lane2 == 0 && s) atomicAdd(reinterpret_cast<unsigned long long*>(p.counters) + k, (unsigned long long)s);
        }
    }
}

// Ray-pencil masks (rt_device.h pencil_cell_word): blockIdx.y = pencil, blockIdx.z = mask word, one thread per cell (+ the all-ones cell). Runs when the scene
// changes, in front of the first trace launch that uses it, on the same stream.
__global__ __launch_bounds__(256) void rt_pencil_build_kernel(const char* scene, uint32_t* masks)
{
    __shared__ PencilPrim prims[2 * RT_PENCIL_MAX_PRIMS];
    const SceneView S = make_view(scene);
    const DevPencil P = S.pencils()[blockIdx.y];
    if (P.kind == RT_PENCIL_OFF || blockIdx.x * 256u > P.cells) return;     // whole workgroup
    const int n = S.h->n_surface + S.h->n_torus;
    for (int k = threadIdx.x; k < n; k += 256) prims[k] = pencil_prim_at(S, P, k);
    __syncthreads();
    const uint32_t cell = blockIdx.x * 256u + threadIdx.x;
    if (cell > P.cells) return;
    const PencilCell C = pencil_cell_geometry(P, cell);
    masks[P.mask_off + (size_t)cell * S.h->pencil_stride + blockIdx.z] = pencil_cell_word(S, P, prims, C, cell, (int)blockIdx.z);
}

// device-side exhaustive check of unorm8 (result[0] = number of mismatching byte values)
__global__ void rt_selftest_kernel(int* result)
{
    const uint32_t b = threadIdx.x;
    const float ref = (float)b / 255.0f;
    if (unorm8(b) != ref) atomicAdd(result, 1);
}

}  // namespace

template <bool CULL, bool COUNT, bool LDS, int WPE = RT_WAVES_PER_EU, bool HEAVY = false, bool SKYLOD = false>
static hipError_t launch_variant(const RtLaunchParams& p_in, dim3 grid, size_t shmem, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    RtLaunchParams p = p_in;
    p.ps_fence_slot = rtdev::path_slots(ps_wide(WPE));   // the pad column behind this variant's path-state slots
    if (LDS && shmem > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&rt_trace_kernel<CULL, COUNT, LDS, WPE, HEAVY, SKYLOD>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem);
        if (e != hipSuccess) return e;
    }
    // With events: the launch's own begin / end timestamps land in them (hipExtLaunchKernel) -- no marker packets in front of and behind the
    // kernel, which cost the queue ~2.5 us each and keep consecutive launches from overlapping their ramp-down and ramp-up.
    if (ev_start && ev_stop) hipExtLaunchKernelGGL((rt_trace_kernel<CULL, COUNT, LDS, WPE, HEAVY, SKYLOD>), grid, dim3(256), (uint32_t)shmem, stream, ev_start, ev_stop, 0, p);
    else hipLaunchKernelGGL((rt_trace_kernel<CULL, COUNT, LDS, WPE, HEAVY, SKYLOD>), grid, dim3(256), shmem, stream, p);
    return hipGetLastError();
}

hipError_t rt_launch_trace(const RtLaunchParams& p_in, bool cull, bool count, bool lds, bool high_occupancy, hipStream_t stream, hipEvent_t ev_start, hipEvent_t ev_stop)
{
    RtLaunchParams p = p_in;
    dim3 grid((p.fb_w + 31) / 32, (p.rows_local + 7) / 8);
    if (grid.x == 0 || grid.y == 0) {   // nothing to trace (a rank without rows): the events still have to be recorded, their readers wait on them
        if (ev_start && hipEventRecord(ev_start, stream) != hipSuccess) return hipGetLastError();
        if (ev_stop && hipEventRecord(ev_stop, stream) != hipSuccess) return hipGetLastError();
        return hipSuccess;
    }
    p.grid_x = (int)grid.x;
    p.grid_y = (int)grid.y;
    if (p.xcd_remap) {
        p.st_nx = (p.grid_x + 3) / 4;
        p.st_ny = (p.grid_y + 3) / 4;
        const int n_st = p.st_nx * p.st_ny;
        grid = dim3((unsigned)(((n_st + 7) / 8) * 8 * 16), 1);   // every XCD gets the same number of 16-workgroup super-tiles
    }
    const size_t shmem = lds ? (size_t)((p.scene_bytes + 15) & ~15) : 0;
    const int sel = (cull ? 4 : 0) | (count ? 2 : 0) | (lds ? 1 : 0);
    if (p.tex.lod && p.tex.sky.levels > 1) {   // mip-mapped sky box: the SKYLOD instantiations (not built for the LDS-staged experiment)
        if (lds) return hipErrorNotSupported;
        if (high_occupancy && sel == 4) return launch_variant<true, false, false, RT_WPE_HEAVY, true, true>(p, grid, shmem, stream, ev_start, ev_stop);
        switch (sel) {
            case 0: return launch_variant<false, false, false, RT_WAVES_PER_EU, false, true>(p, grid, shmem, stream, ev_start, ev_stop);
            case 2: return launch_variant<false, true, false, RT_WAVES_PER_EU, false, true>(p, grid, shmem, stream, ev_start, ev_stop);
            case 4: return launch_variant<true, false, false, RT_WAVES_PER_EU, false, true>(p, grid, shmem, stream, ev_start, ev_stop);
            default: return launch_variant<true, true, false, RT_WAVES_PER_EU, false, true>(p, grid, shmem, stream, ev_start, ev_stop);
        }
    }
    if (high_occupancy && sel == 4) return launch_variant<true, false, false, RT_WPE_HEAVY, true>(p, grid, shmem, stream, ev_start, ev_stop);  // the product path only
    switch (sel) {
        case 0: return launch_variant<false, false, false>(p, grid, shmem, stream, ev_start, ev_stop);
        case 1: return launch_variant<false, false, true>(p, grid, shmem, stream, ev_start, ev_stop);
        case 2: return launch_variant<false, true, false>(p, grid, shmem, stream, ev_start, ev_stop);
        case 3: return launch_variant<false, true, true>(p, grid, shmem, stream, ev_start, ev_stop);
        case 4: return launch_variant<true, false, false>(p, grid, shmem, stream, ev_start, ev_stop);
        case 5: return launch_variant<true, false, true>(p, grid, shmem, stream, ev_start, ev_stop);
        case 6: return launch_variant<true, true, false>(p, grid, shmem, stream, ev_start, ev_stop);
        default: return launch_variant<true, true, true>(p, grid, shmem, stream, ev_start, ev_stop);
    }
}

hipError_t rt_launch_pencil_build(const char* d_scene, const DevSceneHeader& hdr, const DevPencil* pencils, uint32_t* d_masks, hipStream_t stream)
{
    const uint32_t records = hdr.n_pencil + (hdr.pencil_dir != 0xffffffffu ? 1u : 0u);   // the pencils, then the direction table
    uint32_t most = 0;
    for (uint32_t k = 0; k < records; k++)
        if (pencils[k].kind != RT_PENCIL_OFF && pencils[k].cells > most) most = pencils[k].cells;
    if (most == 0) return hipSuccess;
    hipLaunchKernelGGL(rt_pencil_build_kernel, dim3((most + 1 + 255) / 256, records, hdr.pencil_stride), dim3(256), 0, stream, d_scene, d_masks);
    return hipGetLastError();
}

hipError_t rt_launch_selftest(int* d_result, hipStream_t stream)
{
    hipLaunchKernelGGL(rt_selftest_kernel, dim3(1), dim3(256), 0, stream, d_result);
    return hipGetLastError();
}
