// select_rates.hip -- issue cost of selects, compares and lane-mask traffic on gfx950, as
// valu_rates.hip measures arithmetic: ns per wave-instruction per SIMD with 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 select_rates.hip -o select_rates && ./select_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)
constexpr int kIters = 2048;

#define KERNEL(NAME, BODY, PER_ITER)                                                                  \
    __global__ void NAME(unsigned *out) {                                                             \
        unsigned r[16];                                                                               \
        unsigned a = 0x9E3779B9u + threadIdx.x, b = 0xD2511F53u ^ (threadIdx.x * 77u);                \
        float fa = 1.0f + threadIdx.x, fb = 33.0f;                                                    \
        for (int i = 0; i < 16; ++i) r[i] = a + i;                                                    \
        for (int it = 0; it < kIters; ++it) {                                                         \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) { BODY; }                                  \
        }                                                                                             \
        unsigned s = 0;                                                                               \
        for (int i = 0; i < 16; ++i) s += r[i];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s + (unsigned)fa + (unsigned)fb;                 \
    }                                                                                                 \
    constexpr int NAME##_n = PER_ITER;

KERNEL(k_cndmask_vcc, asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a)), 1)
KERNEL(k_cndmask_sgpr, asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(a) : "s20", "s21"), 1)
KERNEL(k_cndmask_indep, { unsigned x; asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(x) : "v"(a), "v"(b)); r[i] = x; }, 1)
KERNEL(k_cmp_f32_vcc, asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(fa), "v"(fb) : "vcc"), 1)
KERNEL(k_cmp_f32_sgpr, asm volatile("v_cmp_lt_f32_e64 s[20:21], %0, %1" : : "v"(fa), "v"(fb) : "s20", "s21"), 1)
KERNEL(k_cmp_then_cndmask, asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc"), 2)
KERNEL(k_minmax_u32, { unsigned lo; asm volatile("v_min_u32 %0, %1, %2\n v_max_u32 %1, %1, %2" : "=&v"(lo), "+v"(r[i]) : "v"(a)); r[i] ^= lo; }, 3)
KERNEL(k_max_u32, asm volatile("v_max_u32 %0, %0, %1" : "+v"(r[i]) : "v"(a)), 1)
KERNEL(k_med3_u32, asm volatile("v_med3_u32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)), 1)
KERNEL(k_bfe, asm volatile("v_bfe_u32 %0, %0, 3, 9" : "+v"(r[i])), 1)
KERNEL(k_perm, asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b)), 1)
KERNEL(k_readlane, { unsigned x; asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(x) : "v"(r[i])); r[i] += x; }, 2)
KERNEL(k_ballot_pop, { unsigned long long m = __ballot(r[i] & 1); r[i] += (unsigned)__popcll(m); }, 1)

typedef void (*kern_t)(unsigned *);
struct entry { const char *name; kern_t fn; int per_iter; };

int main() {
    entry table[] = {
        {"v_cndmask_b32 vcc (dst = src0)", k_cndmask_vcc, 1}, {"v_cndmask_b32_e64 sgpr pair", k_cndmask_sgpr, 1},
        {"v_cndmask_b32 vcc, fresh dst + mov", k_cndmask_indep, 1},
        {"v_cmp_lt_f32 vcc", k_cmp_f32_vcc, 1}, {"v_cmp_lt_f32_e64 sgpr", k_cmp_f32_sgpr, 1},
        {"v_cmp_lt_u32 + v_cndmask (pair)", k_cmp_then_cndmask, 1}, {"v_min_u32 + v_max_u32 + v_xor (triple)", k_minmax_u32, 1},
        {"v_max_u32", k_max_u32, 1}, {"v_med3_u32", k_med3_u32, 1}, {"v_bfe_u32", k_bfe, 1}, {"v_perm_b32", k_perm, 1},
        {"v_readlane + v_add (pair)", k_readlane, 1}, {"ballot + popcount + add", k_ballot_pop, 1},
    };
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned *out;
    CHECK(hipMalloc(&out, sizeof(unsigned) * 1024 * 1024));
    std::printf("%-40s ns per loop body per SIMD at 1, 2, 4 waves per SIMD\n", "body");
    for (const entry &e : table) {
        std::printf("%-40s", e.name);
        for (int wps = 1; wps <= 4; wps *= 2) {
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a));
            CHECK(hipEventCreate(&b));
            hipLaunchKernelGGL(e.fn, dim3(cus), dim3(256 * wps), 0, 0, out);
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(e.fn, dim3(cus), dim3(256 * wps), 0, 0, out);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            std::printf("  %d: %7.3f", wps, ms * 1e6 / ((double)kIters * 16 * wps));
            CHECK(hipEventDestroy(a));
            CHECK(hipEventDestroy(b));
        }
        std::printf("\n");
    }
    return 0;
}
