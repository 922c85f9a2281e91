// Largest relative error of the raw hardware approximations v_rcp_f64 and v_sqrt_f64 (no refinement
// steps) against the correctly rounded results, over a dense sweep of mantissas and a spread of
// exponents.  usage: trans_accuracy [log2 of the samples per block of exponents, default 30]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>

__global__ void k_err(unsigned long long n, int exp_lo, int exp_hi, double *out) {
    double worst_rcp = 0, worst_sqrt = 0;
    const unsigned long long stride = (unsigned long long)gridDim.x * blockDim.x;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        // mantissa: i spread over [1, 4) so that both parities of the exponent are covered for sqrt
        const double m = 1.0 + 3.0 * ((double)i + 0.5) / (double)n;
        for (int e = exp_lo; e <= exp_hi; e += (exp_hi - exp_lo) / 4 + 1) {
            const double x = ldexp(m, e);
            const double r0 = __builtin_amdgcn_rcp(x), r1 = 1.0 / x;
            const double s0 = __builtin_amdgcn_sqrt(x), s1 = sqrt(x);
            worst_rcp = fmax(worst_rcp, fabs(r0 - r1) / r1);
            worst_sqrt = fmax(worst_sqrt, fabs(s0 - s1) / s1);
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        worst_rcp = fmax(worst_rcp, __shfl_xor(worst_rcp, off));
        worst_sqrt = fmax(worst_sqrt, __shfl_xor(worst_sqrt, off));
    }
    if ((threadIdx.x & 63) == 0) {
        double *slot = out + 2 * ((size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64);
        slot[0] = worst_rcp; slot[1] = worst_sqrt;
    }
}

int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 30;
    const unsigned long long n = 1ull << lg;
    const int blocks = 2048, threads = 256, waves = blocks * threads / 64;
    double *d;
    if (hipMalloc((void **)&d, sizeof(double) * 2 * waves) != hipSuccess) { printf("no device\n"); return 1; }
    double *h = (double *)malloc(sizeof(double) * 2 * waves);
    const int ranges[3][2] = {{-8, 8}, {-300, 300}, {-1000, 1000}};
    for (auto &rg : ranges) {
        hipLaunchKernelGGL(k_err, dim3(blocks), dim3(threads), 0, 0, n, rg[0], rg[1], d);
        if (hipMemcpy(h, d, sizeof(double) * 2 * waves, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
        double wr = 0, ws = 0;
        for (int i = 0; i < waves; ++i) { wr = fmax(wr, h[2 * i]); ws = fmax(ws, h[2 * i + 1]); }
        printf("exponents %d..%d, 2^%d mantissas: v_rcp_f64 worst relative error 2^%.2f, v_sqrt_f64 2^%.2f\n", rg[0], rg[1], lg, log2(wr), log2(ws));
    }
    return 0;
}
