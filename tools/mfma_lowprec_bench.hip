// Inner-loop microbenchmark for the low-precision question: the direct 3x3 loop's operand pattern (per horizontal tap 3 weight
// fragments + 5 pixel-row fragments from LDS feed 9 MFMAs; 12-row x 32-output tile, three workgroups of 48 KiB per CU, no
// global traffic) issued with
//   f16   v_mfma_f32_32x32x16_f16          16 bytes per lane and operand, K = 16
//   i8    v_mfma_i32_32x32x32_i8           16 bytes per lane and operand, K = 32
//   fp8   v_mfma_f32_32x32x16_fp8_fp8       8 bytes per lane and operand, K = 16   (the gfx942 instruction)
//   fp8   v_mfma_scale_f32_32x32x64_f8f6f4 32 bytes per lane and operand, K = 64   (e4m3, unit scales)
// on random operands.  One JSON line per case: multiply-accumulates per second as TOP/s (2 * MACs), the LDS bytes read per
// MAC, and the shader clock the kernel held (s_memtime against the 100 MHz s_memrealtime), which on the power cap says
// what a MAC costs in energy.
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_lowprec_bench.hip -o tools/bin/mfma_lowprec_bench
//   ./mfma_lowprec_bench [seconds per case]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                  \
    do {                                                                          \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess) {                                                   \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

constexpr int kLds = 48 * 1024;

enum { F16 = 0, I8 = 1, FP8_K16 = 2, FP8_K64 = 3 };
template <int MODE> struct Mode;
template <> struct Mode<F16> { static constexpr int FB = 16, K = 16; };
template <> struct Mode<I8> { static constexpr int FB = 16, K = 32; };
template <> struct Mode<FP8_K16> { static constexpr int FB = 8, K = 16; };
template <> struct Mode<FP8_K64> { static constexpr int FB = 32, K = 64; };

template <int MODE>
struct Frag {
    u32x4 lo, hi;  // FB bytes per lane: lo alone (16), lo[0..1] (8), lo + hi (32)
};

template <int MODE>
__device__ __forceinline__ Frag<MODE> load_frag(const char* smem, int frag, int lane) {
    constexpr int FB = Mode<MODE>::FB;
    Frag<MODE> f;
    const char* a = smem + frag * 64 * FB + lane * FB;
    if constexpr (FB == 8) {
        const u32x2 v = *(const u32x2*)a;
        f.lo = u32x4{v[0], v[1], 0u, 0u};
    } else {
        f.lo = *(const u32x4*)a;
        if constexpr (FB == 32) f.hi = *(const u32x4*)(a + 16);
    }
    return f;
}

template <int MODE>
__device__ __forceinline__ void mma(f32x16& c, const Frag<MODE>& a, const Frag<MODE>& b) {
    if constexpr (MODE == F16) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a.lo), __builtin_bit_cast(f16x8, b.lo), c, 0, 0, 0);
    } else if constexpr (MODE == I8) {
        c = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a.lo), __builtin_bit_cast(i32x4, b.lo),
                                                                             __builtin_bit_cast(i32x16, c), 0, 0, 0));
    } else if constexpr (MODE == FP8_K16) {
        const long x = (long)a.lo[0] | ((long)a.lo[1] << 32), y = (long)b.lo[0] | ((long)b.lo[1] << 32);
        c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(x, y, c, 0, 0, 0);
    } else {
        const i32x8 x = {(int)a.lo[0], (int)a.lo[1], (int)a.lo[2], (int)a.lo[3], (int)a.hi[0], (int)a.hi[1], (int)a.hi[2], (int)a.hi[3]};
        const i32x8 y = {(int)b.lo[0], (int)b.lo[1], (int)b.lo[2], (int)b.lo[3], (int)b.hi[0], (int)b.hi[1], (int)b.hi[2], (int)b.hi[3]};
        // e4m3 x e4m3, both block scales 2^0 (E8M0 127)
        c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(x, y, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
}

template <int MODE>
__global__ __launch_bounds__(256, 3) void loop_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int iters,
                                                      unsigned long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kLds / 16; i += 256) ((uint4*)smem)[i] = src[(blockIdx.x * (kLds / 16) + i) & 0xFFFFF];
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    constexpr int FB = Mode<MODE>::FB;
    constexpr int frags = kLds / (64 * FB);
    f32x16 acc[3];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            Frag<MODE> wf[3], pf[5];
            const int base = (it + dx * 8) % frags;
#pragma unroll
            for (int k = 0; k < 3; ++k) wf[k] = load_frag<MODE>(smem, (base + k) % frags, lane);
#pragma unroll
            for (int k = 0; k < 5; ++k) pf[k] = load_frag<MODE>(smem, (base + 3 + k) % frags, lane);
#pragma unroll
            for (int rho = 0; rho < 5; ++rho)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = rho - dy;
                    if (r >= 0 && r < 3) mma<MODE>(acc[r], wf[dy], pf[rho]);
                }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[r][i];
    if (s == 123.456f) sink[blockIdx.x * 256 + tid] = s;
    if (clk && lane == 0) {
        const size_t o = ((size_t)blockIdx.x * 4 + (tid >> 6)) * 2;
        clk[o] = __builtin_amdgcn_s_memtime() - c0;
        clk[o + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int MODE>
static void run_case(const char* name, const uint4* src, float* sink, unsigned long long* clk, double seconds) {
    const int grid = 768, iters = 8000;
    auto k = loop_kernel<MODE>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double spent = 0;  // reach the steady power state first
    while (spent < seconds) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) k<<<grid, 256, kLds>>>(src, sink, iters, nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        spent += ms / 1e3;
    }
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<<<grid, 256, kLds>>>(src, sink, iters, i == reps - 1 ? clk : nullptr);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double macs = (double)reps * grid * 4 * iters * 27.0 * 32.0 * 32.0 * Mode<MODE>::K;
    const double lds_bytes = (double)reps * grid * 4 * iters * 24.0 * 64.0 * Mode<MODE>::FB;
    std::vector<unsigned long long> h((size_t)grid * 4 * 2);
    CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (size_t i = 0; i < h.size(); i += 2)
        if (h[i + 1]) ghz.push_back((double)h[i] / (double)h[i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("{\"case\": \"%s\", \"tops\": %.1f, \"x_f16_dense_peak_2500\": %.3f, \"mfma_per_s_G\": %.2f, \"lds_bytes_per_kmac\": %.2f, "
           "\"ms_per_launch\": %.3f, \"in_kernel_clock_ghz\": %.3f}\n",
           name, 2.0 * macs / (ms / 1e3) / 1e12, 2.0 * macs / (ms / 1e3) / 2.5e15,
           (double)reps * grid * 4 * iters * 27.0 / (ms / 1e3) / 1e9, lds_bytes / macs * 1e3, ms / reps,
           ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t n = 1 << 20;  // 16 MiB
    std::vector<uint8_t> h(n * 16);
    uint32_t s = 12345u;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        // random sign and mantissa, exponent field kept in the middle of every format that reads these bytes: as e4m3
        // 0.25 .. 4 (never NaN), as int8 -64 .. 63 without the smallest magnitudes, as f16 pairs normal numbers
        v = (uint8_t)(((s >> 24) & 0x87u) | (0x28u + (((s >> 16) & 3u) << 3)));
    }
    uint4* src;
    float* sink;
    unsigned long long* clk;
    CHECK(hipMalloc((void**)&src, n * 16));
    CHECK(hipMalloc((void**)&sink, 1024 * 256 * 4));
    CHECK(hipMalloc((void**)&clk, 1024 * 4 * 2 * 8));
    CHECK(hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice));
    for (int round = 0; round < 2; ++round) {
        run_case<F16>("f16 32x32x16 (16 B per lane)", src, sink, clk, seconds);
        run_case<I8>("int8 32x32x32 (16 B per lane)", src, sink, clk, seconds);
        run_case<FP8_K16>("fp8 e4m3 32x32x16_fp8_fp8 (8 B per lane)", src, sink, clk, seconds);
        run_case<FP8_K64>("fp8 e4m3 scale_32x32x64_f8f6f4 (32 B per lane)", src, sink, clk, seconds);
    }
    return 0;
}
