// sampler_dump.hip -- the per-path stream's unit-sphere sampler (StreamRng::in_unit_sphere), n points of the
// stream (seed, pixel, sample) per lane for `lanes` consecutive pixels, written as raw doubles to a file:
// tests compare them with the oracle's restatement bit for bit.
//   sampler_dump SEED PIXEL0 SAMPLE LANES N OUT
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "rt_device.hpp"

__global__ void k_dump(uint64_t seed, uint32_t pixel0, uint32_t sample, uint32_t lanes, uint32_t n, double *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lanes) return;
    rtd::StreamRng g;
    g.init(seed, pixel0 + i, sample);
    for (uint32_t k = 0; k < n; ++k) {
        double x, y, z;
        g.in_unit_sphere(x, y, z);
        double *o = out + ((size_t)i * n + k) * 3;
        o[0] = x; o[1] = y; o[2] = z;
    }
}

int main(int argc, char **argv) {
    if (argc != 7) { std::fprintf(stderr, "usage: sampler_dump SEED PIXEL0 SAMPLE LANES N OUT\n"); return 2; }
    const uint64_t seed = std::strtoull(argv[1], nullptr, 10);
    const uint32_t pixel0 = (uint32_t)std::strtoul(argv[2], nullptr, 10), sample = (uint32_t)std::strtoul(argv[3], nullptr, 10);
    const uint32_t lanes = (uint32_t)std::strtoul(argv[4], nullptr, 10), n = (uint32_t)std::strtoul(argv[5], nullptr, 10);
    const size_t count = (size_t)lanes * n * 3;
    double *d = nullptr;
    if (hipMalloc((void **)&d, count * sizeof(double)) != hipSuccess) { std::fprintf(stderr, "no device memory\n"); return 1; }
    hipLaunchKernelGGL(k_dump, dim3((lanes + 255) / 256), dim3(256), 0, 0, seed, pixel0, sample, lanes, n, d);
    std::vector<double> h(count);
    if (hipMemcpy(h.data(), d, count * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { std::fprintf(stderr, "kernel failed\n"); return 1; }
    std::FILE *f = std::fopen(argv[6], "wb");
    if (!f || std::fwrite(h.data(), sizeof(double), count, f) != count) { std::fprintf(stderr, "cannot write %s\n", argv[6]); return 1; }
    std::fclose(f);
    return 0;
}
