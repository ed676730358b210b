// Micro-test (GPU box): accuracy of exp2(y*log2(x)) on the hardware transcendental units against double pow,
// for the argument range the shading code uses (x in [0,1], integer y).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
__global__ void k(const float* x, float y, float* fast, float* lib, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    fast[i] = __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x[i]));
    lib[i] = powf(x[i], y);
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> hx(n), hf(n), hl(n);
    for (int i = 0; i < n; i++) {
        // half the points uniformly in [0,1], half packed towards 1 (1 - 2^-k * u)
        if (i & 1) hx[i] = (float)i / n;
        else hx[i] = 1.0f - ldexpf((float)((i * 2654435761u) >> 8) / 16777216.0f, -((i >> 1) % 24));
    }
    hx[0] = 0.0f; hx[2] = 1.0f;
    float *dx, *df, *dl;
    hipMalloc(&dx, n * 4); hipMalloc(&df, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(dx, hx.data(), n * 4, hipMemcpyHostToDevice);
    const float ys[] = {1, 2, 5, 10, 50, 100, 200, 500, 1000, 5000};
    for (float y : ys) {
        k<<<n / 256, 256>>>(dx, y, df, dl, n);
        hipMemcpy(hf.data(), df, n * 4, hipMemcpyDeviceToHost);
        hipMemcpy(hl.data(), dl, n * 4, hipMemcpyDeviceToHost);
        double ef = 0, el = 0; float xf = 0;
        for (int i = 0; i < n; i++) {
            const double r = pow((double)hx[i], (double)y);
            const double a = fabs(hf[i] - r), b = fabs(hl[i] - r);
            if (a > ef) { ef = a; xf = hx[i]; }
            if (b > el) el = b;
        }
        printf("y=%6.0f  max abs err: exp2(y*log2 x) %.3e (at x=%.9g)   ocml powf %.3e\n", y, ef, xf, el);
    }
    return 0;
}
