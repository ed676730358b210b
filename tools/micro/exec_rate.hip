// exec_rate.hip -- does a gfx950 SIMD issue a VALU instruction faster when only some lanes of the wave are enabled?
// (GPU box.) Question behind it (round 4, DESIGN.md section 5 "Torus-heavy"): the tail of a Durand-Kerner run keeps a whole wave busy
// for a handful of lanes; if the hardware skipped 16-lane quarters whose EXEC bits are all zero, compacting those lanes into one
// quarter would make the tail's sweeps cheaper without moving any work to another wave.
// Method: every wave runs `iters` x 32 independent instructions under an EXEC mask chosen by lane id; wall time by HIP events against
// the all-lanes run of the same kind (8 waves per SIMD: 2048 workgroups of 256).
//   hipcc --offload-arch=gfx950 -O3 -o exec_rate exec_rate.hip && ./exec_rate
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));
enum { K_FMA, K_ADD, K_PK_MUL, K_RCP, K_COUNT };
static const char* const kNames[K_COUNT] = {"v_fma_f32", "v_add_f32", "v_pk_mul_f32", "v_rcp_f32"};

template <int KIND>
__global__ __launch_bounds__(256) void stream(float* sink, int iters, float s, unsigned long long mask)
{
    float a[32];
    v2f p[32];
#pragma unroll
    for (int i = 0; i < 32; i++) { a[i] = threadIdx.x * 1e-3f + i; p[i].x = a[i]; p[i].y = a[i] + 0.5f; }
    const v2f sv = {s, s};
    if ((mask >> (threadIdx.x & 63)) & 1ull) {
        for (int it = 0; it < iters; it++) {
#pragma unroll
            for (int i = 0; i < 32; i++) {
                if (KIND == K_FMA) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == K_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
                if (KIND == K_PK_MUL) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(sv));
                if (KIND == K_RCP) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            }
        }
    }
    float r = 0;
#pragma unroll
    for (int i = 0; i < 32; i++) r += a[i] + p[i].x + p[i].y;
    sink[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int KIND>
static float run(float* d_sink, unsigned long long mask, int iters)
{
    const int blocks = 256 * 8;
    stream<KIND><<<blocks, 256>>>(d_sink, 200, 1.0001f, mask);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0);
        stream<KIND><<<blocks, 256>>>(d_sink, iters, 1.0001f, mask);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    return best;
}

template <int KIND>
static void kind(float* d_sink)
{
    const int iters = 2000;
    struct { const char* name; unsigned long long m; } masks[] = {
        {"all 64 lanes", ~0ull}, {"lanes 0-31", 0xffffffffull}, {"lanes 0-15", 0xffffull}, {"lanes 16-31", 0xffff0000ull}, {"lane 0", 1ull},
        {"lanes 0-15 + 48-63", 0xffff00000000ffffull}, {"every 4th lane", 0x1111111111111111ull}, {"lanes 32-63", 0xffffffff00000000ull}};
    float full = 0;
    for (auto& mk : masks) {
        const float ms = run<KIND>(d_sink, mk.m, iters);
        if (mk.m == ~0ull) full = ms;
        printf("%-14s %-20s %8.3f ms  = %.3f of the all-lanes run\n", kNames[KIND], mk.name, ms, ms / full);
    }
}

int main()
{
    float* d_sink;
    hipMalloc(&d_sink, 256 * 8 * 256 * sizeof(float));
    kind<K_FMA>(d_sink); kind<K_ADD>(d_sink); kind<K_PK_MUL>(d_sink); kind<K_RCP>(d_sink);
    return 0;
}
