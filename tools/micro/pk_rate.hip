// Micro-benchmark (GPU box): issue cost of packed FP32 VALU instructions against scalar ones on gfx950.
// Every wave runs ITER iterations of 16 independent instructions of one kind; 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip && ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

#define REP16(X) X X X X X X X X X X X X X X X X
template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, float s)
{
    float a[16];
    v2f p[16];
    for (int i = 0; i < 16; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
    const v2f sv = {s, s};
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == 1) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sv));
            if (KIND == 2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == 3) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sv));
            if (KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(p[i]) : "v"(sv));
            if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s));
            if (KIND == 8) asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(a[i]) : "s20");
            if (KIND == 9) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]));
            if (KIND == 10) asm volatile("v_mul_f64 %0, %0, %0" : "+v"(p[i]));
            if (KIND == 11) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(a[i]) : "v"(s) : "s20", "s21");
            if (KIND == 12) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[i]), "v"(s) : "vcc");
            if (KIND == 13) asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(a[i]), "v"(s) : "s20", "s21");
            if (KIND == 14) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == 15) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i]) : "v"(s));
            if (KIND == 16) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(s) : "vcc");
            if (KIND == 17) asm volatile("v_div_scale_f32 %0, vcc, %0, %1, %0" : "+v"(a[i]) : "v"(s) : "vcc");
            if (KIND == 18) asm volatile("v_div_fmas_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s) : "vcc");
            if (KIND == 19) asm volatile("v_div_fixup_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(s));
            if (KIND == 20) asm volatile("v_writelane_b32 %0, s20, 3" : "+v"(a[i]) : : "s20");
            if (KIND == 21) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == 22) asm volatile("v_fmac_f32 %0, %1, %1" : "+v"(a[i]) : "v"(s));
            if (KIND == 23) asm volatile("s_and_b64 s[20:21], s[20:21], exec" : : : "s20", "s21", "scc");
        }
    }
    float r = 0;
    for (int i = 0; i < 16; i++) r += a[i] + p[i].x + p[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = r;
}
template <int KIND> void run(const char* name, float* d)
{
    const int iters = 20000, blocks = 256 * 4;  // 4 blocks of 4 waves per CU -> 4 waves per SIMD
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<KIND><<<blocks, 256>>>(d, 100, 1.0001f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256>>>(d, iters, 1.0001f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * 16 * 4;  // 4 waves per SIMD
    printf("%-14s %8.3f ms  %6.2f ns per wave-instruction per SIMD  (%.2f cycles at 2.4 GHz)\n", name, ms, ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
}
int main()
{
    float* d; hipMalloc(&d, 256 * 4 * 256 * sizeof(float));
    run<0>("v_mul_f32", d); run<1>("v_pk_mul_f32", d); run<2>("v_add_f32", d); run<3>("v_pk_add_f32", d);
    run<4>("v_fma_f32", d); run<5>("v_pk_fma_f32", d); run<6>("v_rcp_f32", d); run<7>("v_cndmask_b32", d);
    run<8>("v_readlane_b32", d); run<9>("v_sqrt_f32", d); run<10>("v_mul_f64", d);
    run<11>("v_cndmask_e64", d); run<12>("v_cmp vcc", d); run<13>("v_cmp_e64 sgpr", d); run<14>("v_max_f32", d);
    run<15>("v_mov_b32", d); run<16>("cmp+cndmask", d); run<17>("v_div_scale", d); run<18>("v_div_fmas", d);
    run<19>("v_div_fixup", d); run<20>("v_writelane", d); run<21>("v_add_u32", d); run<22>("v_fmac_f32", d); run<23>("s_and_b64", d);
    return 0;
}
