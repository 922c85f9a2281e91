// division_check.hip -- brute-force check of rtd::div_by_shared against the IEEE division it stands
// for in the render kernel: u = (x + xi) / nx with x a pixel index, xi a 53-bit uniform in [0, 1)
// and nx an image size (main.cpp:130-133).  For every size from 1 to 8192 and a spread of larger
// ones up to 2^16 (and 2^32 - 1), numerators cover every pixel column's neighbourhood: random
// fractions, fractions of few bits, the largest below 1, and exact integers.
// Prints the number of mismatches (must be 0).   usage: division_check [numerators per size]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rt_device.hpp"

using namespace rtd;

__device__ unsigned long long mix(unsigned long long x) {
    x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ void k_check(const unsigned *sizes, unsigned n_sizes, unsigned per_size, unsigned long long *bad, double *example) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (unsigned long long)n_sizes * per_size) return;
    const unsigned nx = sizes[i / per_size];
    const double d = (double)nx, y = refined_rcp(d);
    unsigned long long h = mix(i);
    const unsigned x = (unsigned)(mix(h) % nx);
    double xi;
    switch (h % 6) {
    case 0: xi = 0.0; break;
    case 1: xi = 1.0 - 0x1p-53; break;
    case 2: xi = (double)((h >> 8) & 0xFFu) * 0x1p-8; break;                  // few bits
    case 3: xi = (double)((h >> 11) | 1ull) * 0x1p-53 * 0x1p-30; break;        // tiny
    default: xi = (double)(h >> 11) * 0x1p-53; break;
    }
    const double n = (double)x + xi;
    const double a = n / d, b = div_by_shared(n, d, y);
    if (!(a == b)) {
        if (atomicAdd(bad, 1ull) == 0) { example[0] = n; example[1] = d; example[2] = a; example[3] = b; }
    }
}

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

int main(int argc, char **argv) {
    const unsigned per_size = argc > 1 ? (unsigned)std::strtoul(argv[1], nullptr, 10) : 65536u;
    std::vector<unsigned> sizes;
    for (unsigned s = 1; s <= 8192; ++s) sizes.push_back(s);
    for (unsigned s = 8193; s < 65536; s += 37) sizes.push_back(s);
    for (unsigned s : {65535u, 65536u, 1u << 20, (1u << 24) + 1u, 0x7FFFFFFFu, 0xFFFFFFFFu}) sizes.push_back(s);
    unsigned *d_sizes; unsigned long long *d_bad; double *d_ex;
    CHECK(hipMalloc((void **)&d_sizes, sizes.size() * sizeof(unsigned)));
    CHECK(hipMalloc((void **)&d_bad, sizeof(unsigned long long)));
    CHECK(hipMalloc((void **)&d_ex, 4 * sizeof(double)));
    CHECK(hipMemcpy(d_sizes, sizes.data(), sizes.size() * sizeof(unsigned), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_bad, 0, sizeof(unsigned long long)));
    const unsigned long long total = (unsigned long long)sizes.size() * per_size;
    hipLaunchKernelGGL(k_check, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, 0, d_sizes, (unsigned)sizes.size(), per_size, d_bad, d_ex);
    CHECK(hipDeviceSynchronize());
    unsigned long long bad = 0; double ex[4];
    CHECK(hipMemcpy(&bad, d_bad, sizeof bad, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ex, d_ex, sizeof ex, hipMemcpyDeviceToHost));
    std::printf("divisions %llu mismatches %llu\n", total, bad);
    if (bad) std::printf("example: %.17g / %.17g = %.17g, shared reciprocal gives %.17g\n", ex[0], ex[1], ex[2], ex[3]);
    return bad ? 1 : 0;
}
