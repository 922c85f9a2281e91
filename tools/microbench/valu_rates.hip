// valu_rates.hip -- issue cost of the instructions the render kernel is made of, on the
// GPU it runs on: cycles per wave-instruction with 1..4 waves per SIMD, and the shader
// clock under that load (cycle counter against the 100 MHz real-time counter).
//   hipcc --offload-arch=gfx950 -O2 valu_rates.hip -o valu_rates && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

constexpr int kIters = 2048;

// sixteen independent instances of one instruction per iteration
#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

#define KERNEL_F64(NAME, ASM)                                                                         \
    __global__ void NAME(double *out, long long *clk) {                                               \
        double r[16];                                                                                 \
        double a = 1.000001 + threadIdx.x * 1e-9, b = 0.999999;                                       \
        for (int i = 0; i < 16; ++i) r[i] = a + i;                                                    \
        long long t0 = clock64(), w0 = wall_clock64();                                                \
        for (int it = 0; it < kIters; ++it) {                                                         \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b)); \
        }                                                                                             \
        long long t1 = clock64(), w1 = wall_clock64();                                                \
        double s = 0;                                                                                 \
        for (int i = 0; i < 16; ++i) s += r[i];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }   \
    }

#define KERNEL_U32(NAME, ASM)                                                                         \
    __global__ void NAME(double *out, long long *clk) {                                               \
        unsigned r[16];                                                                               \
        unsigned a = 0x9E3779B9u + threadIdx.x, b = 0xD2511F53u;                                      \
        for (int i = 0; i < 16; ++i) r[i] = a + i;                                                    \
        long long t0 = clock64(), w0 = wall_clock64();                                                \
        for (int it = 0; it < kIters; ++it) {                                                         \
            _Pragma("unroll") for (int i = 0; i < 16; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(a), "v"(b)); \
        }                                                                                             \
        long long t1 = clock64(), w1 = wall_clock64();                                                \
        unsigned s = 0;                                                                               \
        for (int i = 0; i < 16; ++i) s += r[i];                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                               \
        if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }   \
    }

KERNEL_F64(k_add_f64, "v_add_f64 %0, %0, %1")
KERNEL_F64(k_mul_f64, "v_mul_f64 %0, %0, %2")
KERNEL_F64(k_fma_f64, "v_fma_f64 %0, %0, %2, %1")
KERNEL_F64(k_max_f64, "v_max_f64 %0, %0, %1")
KERNEL_F64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL_F64(k_sqrt_f64, "v_sqrt_f64 %0, %0")
KERNEL_F64(k_rsq_f64, "v_rsq_f64 %0, %0")
KERNEL_F64(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
KERNEL_F64(k_mov_b64, "v_mov_b64 %0, %1")
KERNEL_U32(k_fma_f32, "v_fma_f32 %0, %0, %2, %1")
KERNEL_U32(k_max3_f32, "v_max3_f32 %0, %0, %1, %2")
KERNEL_U32(k_max_f32, "v_max_f32 %0, %0, %1")
KERNEL_F64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %2, %1")
KERNEL_U32(k_add_u32, "v_add_u32 %0, %0, %1")
KERNEL_U32(k_xor_b32, "v_xor_b32 %0, %0, %1")
KERNEL_U32(k_lshl_b32, "v_lshlrev_b32 %0, 7, %0")
KERNEL_U32(k_alignbit, "v_alignbit_b32 %0, %0, %0, 25")
KERNEL_U32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %2")
KERNEL_U32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %2")
KERNEL_U32(k_mul_u24, "v_mul_u32_u24 %0, %0, %2")
KERNEL_U32(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
KERNEL_U32(k_add3, "v_add3_u32 %0, %0, %1, %2")
KERNEL_U32(k_lshl_add, "v_lshl_add_u32 %0, %0, 9, %1")
KERNEL_U32(k_cvt_f64_u32, "v_cvt_f32_u32 %0, %0")
KERNEL_U32(k_min_u32, "v_min_u32 %0, %0, %1")
KERNEL_U32(k_and_or, "v_and_or_b32 %0, %0, %1, %2")
KERNEL_U32(k_min3_f32, "v_min3_f32 %0, %0, %1, %2")

__global__ void k_cmp_f64(double *out, long long *clk) {
    double a = 1.000001 + threadIdx.x * 1e-9, b = 0.999999;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a), "v"(b) : "vcc");
    }
    long long t1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

template <int BYTES>
__global__ void k_lds_read(double *out, long long *clk) {
    __shared__ double buf[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) buf[i] = i;
    __syncthreads();
    // 120-byte record stride, as the render kernel's node records
    unsigned addr = (unsigned)(size_t)buf + (threadIdx.x * 37u % 500u) * (BYTES == 16 ? 112u : 120u);
    double r[16];
    for (int i = 0; i < 16; ++i) r[i] = 0;
    long long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (BYTES == 16) { __attribute__((ext_vector_type(4))) unsigned q; asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(addr), "n"((i & 3) * 16)); r[i] = q.x + q.w; }
            else if (BYTES == 8) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[i]) : "v"(addr), "n"(i * 8));
            else { unsigned x; asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(x) : "v"(addr), "n"(i * 8)); r[i] = x; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long long t1 = clock64(), w1 = wall_clock64();
    double s = 0;
    for (int i = 0; i < 16; ++i) s += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

typedef void (*kern_t)(double *, long long *);
struct entry { const char *name; kern_t fn; };

int main() {
    entry table[] = {
        {"v_add_f64", k_add_f64}, {"v_mul_f64", k_mul_f64}, {"v_fma_f64", k_fma_f64}, {"v_max_f64", k_max_f64},
        {"v_cmp_lt_f64", k_cmp_f64}, {"v_rcp_f64", k_rcp_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_rsq_f64", k_rsq_f64},
        {"v_div_fixup_f64", k_div_fixup_f64}, {"v_mov_b64", k_mov_b64},
        {"v_fma_f32", k_fma_f32}, {"v_max3_f32", k_max3_f32}, {"v_max_f32", k_max_f32}, {"v_pk_fma_f32", k_pk_fma_f32},
        {"v_add_u32", k_add_u32}, {"v_xor_b32", k_xor_b32}, {"v_lshlrev_b32", k_lshl_b32}, {"v_alignbit_b32", k_alignbit},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mul_u32_u24", k_mul_u24}, {"v_cndmask_b32", k_cndmask},
        {"v_min_u32", k_min_u32}, {"v_and_or_b32", k_and_or}, {"v_min3_f32", k_min3_f32},
        {"ds_read_b64", k_lds_read<8>}, {"ds_read_b32", k_lds_read<4>}, {"ds_read_b128", k_lds_read<16>},
    };
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    std::printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, cus, prop.clockRate);
    double *out;
    long long *clk;
    CHECK(hipMalloc(&out, sizeof(double) * 1024 * 1024));
    CHECK(hipMalloc(&clk, sizeof(long long) * 2 * 4096));
    std::printf("%-18s %s\n", "instruction", "waves/SIMD: cycles per wave-instruction per SIMD (counter ticks) | event-time ns per instr per SIMD | counter MHz");
    for (const entry &e : table) {
        std::printf("%-18s", e.name);
        for (int wps = 1; wps <= 4; ++wps) {
            const int block = 256 * wps;   // wps waves on each of the CU's 4 SIMDs
            hipEvent_t a, b;
            CHECK(hipEventCreate(&a));
            CHECK(hipEventCreate(&b));
            hipLaunchKernelGGL(e.fn, dim3(cus), dim3(block), 0, 0, out, clk);   // warm-up
            CHECK(hipEventRecord(a));
            hipLaunchKernelGGL(e.fn, dim3(cus), dim3(block), 0, 0, out, clk);
            CHECK(hipEventRecord(b));
            CHECK(hipEventSynchronize(b));
            float ms = 0;
            CHECK(hipEventElapsedTime(&ms, a, b));
            std::vector<long long> h(2 * cus);
            CHECK(hipMemcpy(h.data(), clk, sizeof(long long) * 2 * cus, hipMemcpyDeviceToHost));
            double ticks = 0, wall = 0;
            for (int i = 0; i < cus; ++i) { ticks += h[2 * i]; wall += h[2 * i + 1]; }
            ticks /= cus; wall /= cus;
            const double insts_per_simd = (double)kIters * 16 * wps;
            std::printf("  %d: %6.2f | %6.3f ns | %5.0f", wps, ticks / insts_per_simd, ms * 1e6 / insts_per_simd, ticks / (wall / 100.0));
            CHECK(hipEventDestroy(a));
            CHECK(hipEventDestroy(b));
        }
        std::printf("\n");
    }
    return 0;
}
