// MFMA shape against sustained rate on a power-limited chip: the inner loop of the 3x3 convolution (every
// operand fragment read from LDS with ds_read_b128, 0.89 fragment reads per 32x32x16 MFMA, three 4-wave
// workgroups of 48 KiB per CU) with v_mfma_f32_32x32x16 against the same FLOPs and the same LDS bytes issued as
// v_mfma_f32_16x16x32, f16 and bf16, on random operands.  No global traffic inside the loop.
//
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shape_bench.hip -o gpurun_out/mfma_shape_bench
//   ./mfma_shape_bench [seconds per case]         -> one JSON line per case (TFLOP/s, in-kernel clock)
//
// Second part ("staging" cases, 32x32x16 f16 only): the same loop with the real kernel's global -> LDS staging added
// step by step, to price its parts on the power cap: LDS-DMA pieces issued between the MFMAs (24 KiB per workgroup and
// stage: 15 KiB of "pixels" from a buffer that is L2-resident, Infinity-Cache-resident or HBM-sized, 9 KiB of "weights"
// that every workgroup shares), first free-running (no wait ever), then with the stage's s_waitcnt vmcnt(0) + s_barrier.
//
// Third part ("winograd" cases, f16): the inner loop a Winograd F(2x2,3x3) form of the same convolution would run, at
// equal ALGORITHMIC FLOPs.  One unit per wave = 32 output tiles of 2x2 pixels x 32 outputs x 16 input channels (one
// stage): 16 ds_read_b128 of the 4x4 input patch (a lane holds its tile's 16 pixels for its 8 channels), the input
// transform V = B^T d B in registers (32 packed-f16 vector adds of 8 channels = 128 v_pk_add_f16 per lane, every add
// rounding to f16), 16 ds_read_b128 of transformed-weight fragments and 16 MFMAs 32x32x16 into 16 accumulators (one per
// transform position: 256 accumulator registers, so one wave per SIMD and one workgroup per CU).  The direct loop
// spends 36 MFMAs on the same 128 pixels x 32 outputs x 16 channels.  Software pipeline: the MFMAs of unit k read V(k)
// while the adds transform unit k+1 in a second register array, two adds of 8 channels behind every MFMA; the register
// an MFMA has just consumed is refilled at once with a patch fragment of unit k+2.  The output transform (24 f32 adds
// per tile and output, once per convolution, not per stage) is not part of the loop.
//
// One "stage pair" per wave = the work of two 16-channel stages of a 3-row x 32-pixel x 32-output wave tile:
// 54 MFMAs 32x32x16 or 108 MFMAs 16x16x32 (the two-plane-stage form, K = 32), 48 fragments of 1 KiB either way.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                        \
    do {                                                                                \
        hipError_t e_ = (x);                                                            \
        if (e_ != hipSuccess) {                                                         \
            fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

constexpr int kLds = 48 * 1024;

template <bool BF>
__device__ __forceinline__ void mma32(f32x16& c, const u32x4& a, const u32x4& b) {
    if constexpr (BF)
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
template <bool BF>
__device__ __forceinline__ void mma16(f32x4& c, const u32x4& a, const u32x4& b) {
    if constexpr (BF)
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    else
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

// SHAPE 32: per horizontal tap and plane 3 weight fragments + 5 pixel-row fragments feed 9 MFMAs (3 rows x 3
// vertical taps).  SHAPE 16: per horizontal tap 6 weight fragments (3 vertical taps x 2 halves of the 32 outputs,
// K = 32 = both planes) + 10 pixel fragments (5 rows x 2 halves of the 32 pixels) feed 36 MFMAs.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

// STAGING 0: none.  1: DMA pieces issued, never waited for.  2: + s_waitcnt vmcnt(0) and s_barrier per stage.
// pix / pix_bytes: the buffer the pixel pieces stream through (each workgroup walks its own 15 KiB windows).
template <int SHAPE, bool BF, int STAGING = 0>
__global__ __launch_bounds__(256, 3) void loop_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int pairs,
                                                      unsigned long long* __restrict__ clk, const char* pix = nullptr,
                                                      unsigned long long pix_bytes = 0) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kLds / 16; i += 256) ((uint4*)smem)[i] = src[(blockIdx.x * (kLds / 16) + i) & 0xFFFFF];
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const int frags = kLds / 1024;  // 48 fragments of 1 KiB in the workgroup's LDS
    if constexpr (SHAPE == 32) {
        f32x16 acc[3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][i] = 0.f;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        const unsigned long long windows = STAGING ? pix_bytes / (16 * 1024) : 1;
        for (int it = 0; it < pairs; ++it) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {      // the two planes of the pair, one after the other
                __amdgpu_buffer_rsrc_t ra, rb;
                char* slot = smem + half * (24 * 1024);
                if constexpr (STAGING) {
                    if constexpr (STAGING == 2) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        __builtin_amdgcn_s_barrier();
                    }
                    const unsigned long long stage = (unsigned long long)it * 2 + half;
                    const unsigned long long win = ((unsigned long long)blockIdx.x * 7919ull + stage) % windows;
                    ra = make_rsrc(pix + win * (16 * 1024), 16 * 1024);              // this stage's pixel window
                    rb = make_rsrc((const char*)src + (stage % 12) * (9 * 1024), 9 * 1024);  // weights: shared by all
                }
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    if constexpr (STAGING) {
                        // two of the wave's six pieces per horizontal tap: pieces 0-3 of a wave are pixel pieces
                        // (15 of them per workgroup), 4-5 weight pieces (9 per workgroup)
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int q = dx * 2 + j;
                            if (q < 4) {
                                const int k = wave + 4 * q;
                                if (k < 15)
                                    __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(slot + k * 1024),
                                                                             16, k * 1024 + lane * 16, 0, 0, 0);
                            } else {
                                const int k = wave + 4 * (q - 4);
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(slot + (15 + k) * 1024),
                                                                         16, k * 1024 + lane * 16, 0, 0, 0);
                            }
                        }
                    }
                    u32x4 wf[3], pf[5];
                    const int base = (it + half * 24 + dx * 8) % frags;
#pragma unroll
                    for (int k = 0; k < 3; ++k) wf[k] = *(const u32x4*)(smem + ((base + k) % frags) * 1024 + lane * 16);
#pragma unroll
                    for (int k = 0; k < 5; ++k) pf[k] = *(const u32x4*)(smem + ((base + 3 + k) % frags) * 1024 + lane * 16);
#pragma unroll
                    for (int rho = 0; rho < 5; ++rho)
#pragma unroll
                        for (int dy = 0; dy < 3; ++dy) {
                            const int r = rho - dy;
                            if (r >= 0 && r < 3) mma32<BF>(acc[r], wf[dy], pf[rho]);
                        }
                }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[r][i];
        if (s == 123.456f) sink[blockIdx.x * 256 + tid] = s;
    } else {
        f32x4 acc[3][2][2];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[r][a][b][i] = 0.f;
        for (int it = 0; it < pairs; ++it) {
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                u32x4 wf[3][2], pf[5][2];
                const int base = (it + dx * 16) % frags;
#pragma unroll
                for (int k = 0; k < 6; ++k) wf[k >> 1][k & 1] = *(const u32x4*)(smem + ((base + k) % frags) * 1024 + lane * 16);
#pragma unroll
                for (int k = 0; k < 10; ++k) pf[k >> 1][k & 1] = *(const u32x4*)(smem + ((base + 6 + k) % frags) * 1024 + lane * 16);
#pragma unroll
                for (int rho = 0; rho < 5; ++rho)
#pragma unroll
                    for (int dy = 0; dy < 3; ++dy) {
                        const int r = rho - dy;
                        if (r >= 0 && r < 3) {
#pragma unroll
                            for (int a = 0; a < 2; ++a)
#pragma unroll
                                for (int b = 0; b < 2; ++b) mma16<BF>(acc[r][a][b], wf[dy][a], pf[rho][b]);
                        }
                    }
            }
        }
        float s = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int i = 0; i < 4; ++i) s += acc[r][a][b][i];
        if (s == 123.456f) sink[blockIdx.x * 256 + tid] = s;
    }
    if (clk && lane == 0) {
        const size_t o = ((size_t)blockIdx.x * 4 + (tid >> 6)) * 2;
        clk[o] = __builtin_amdgcn_s_memtime() - c0;
        clk[o + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

// a - b on 8 halves as four v_pk_add_f16 with the second operand negated (the compiler splits a vector subtraction
// of halves into scalar v_sub_f16)
__device__ __forceinline__ h8 sub8(const h8& a, const h8& b) {
    const u32x4 x = __builtin_bit_cast(u32x4, a), y = __builtin_bit_cast(u32x4, b);
    u32x4 r;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        unsigned t;
        asm("v_pk_add_f16 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(t) : "v"(x[i]), "v"(y[i]));
        r[i] = t;
    }
    return __builtin_bit_cast(h8, r);
}

// XFORM 1: the full loop.  XFORM 0: the same loads and MFMAs without the transform adds (what the matrix pipe and
// the LDS alone sustain in this form).
template <int XFORM>
__global__ __launch_bounds__(256, 1) void wino_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int units,
                                                      unsigned long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kLds / 16; i += 256) ((uint4*)smem)[i] = src[(blockIdx.x * (kLds / 16) + i) & 0xFFFFF];
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;
    auto frag = [&](int f) { return *(const h8*)(smem + (f & 31) * 1024 + lane * 16); };   // ring over the first 32 KiB
    h8 P[16], Q[16];   // one holds V of the unit being multiplied, the other the patch being transformed
    h8 wf[8];
#pragma unroll
    for (int k = 0; k < 16; ++k) { P[k] = frag(k); Q[k] = frag(16 + k); }
#pragma unroll
    for (int k = 0; k < 8; ++k) wf[k] = frag(32 + k);
    // One unit: MFMAs read V (and refill each consumed register with a patch fragment of the unit after next), the
    // adds turn X from patch d into V = B^T d B in place: steps 0-15 the four columns (B^T d), 16-31 the four rows.
    auto step = [&](h8 (&X)[16], const int st) {
        if constexpr (XFORM) {
            const int half = st >> 4, line = (st >> 2) & 3, op = st & 3;
            // element e of the line: column transform walks rows (stride 4) of column `line`, row transform walks
            // the columns (stride 1) of row `line`
            auto at = [&](int e) -> h8& { return half == 0 ? X[e * 4 + line] : X[line * 4 + e]; };
            // in place, order chosen so that each input is still intact when it is read:
            //   op0: x0 = x0 - x2 ; op1: x3 = x1 - x3 ; op2: tmp = x1 + x2, x2 = x2 - x1 (two adds) ; op3: x1 = tmp
            if (op == 0) at(0) = sub8(at(0), at(2));
            else if (op == 1) at(3) = sub8(at(1), at(3));
            else if (op == 2) { const h8 t = at(1) + at(2); at(2) = sub8(at(2), at(1)); at(1) = t; }
        }
    };
    auto unit = [&](h8 (&V)[16], h8 (&X)[16], const int par) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[k & 7], V[k], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            // constant LDS offsets, as in a real kernel (patch and weight fragments of a stage sit at fixed places):
            // no address arithmetic in the loop.  `par` = which of the two register arrays is V in this unit.
            V[k] = frag(par * 16 + k);                  // patch fragment of the unit after next
            wf[k & 7] = frag(par * 8 + 5 + k);          // weight fragment for the multiply eight slots ahead
            step(X, 2 * k);
            step(X, 2 * k + 1);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int it = 0; it < units; it += 2) {
        unit(P, Q, 0);
        unit(Q, P, 1);
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[k][i];
#pragma unroll
    for (int k = 0; k < 16; ++k) s += (float)P[k][0] + (float)Q[k][1];
    if (s == 123.456f) sink[blockIdx.x * 256 + tid] = s;
    if (clk && lane == 0) {
        const size_t o = ((size_t)blockIdx.x * 4 + (tid >> 6)) * 2;
        clk[o] = __builtin_amdgcn_s_memtime() - c0;
        clk[o + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

template <int XFORM>
static void run_wino(const char* name, const uint4* src, float* sink, unsigned long long* clk, double seconds) {
    const int grid = 768, units = 6000;
    auto k = wino_kernel<XFORM>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double spent = 0;
    while (spent < seconds) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) k<<<grid, 256, kLds>>>(src, sink, units, nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        spent += ms / 1e3;
    }
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<<<grid, 256, kLds>>>(src, sink, units, i == reps - 1 ? clk : nullptr);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // per wave and unit: 16 MFMAs issued; the direct form spends 36 on the same pixels, outputs and channels
    const double issued = (double)reps * grid * 4 * units * 16.0 * 32768.0;
    const double algorithmic = issued * 36.0 / 16.0;
    std::vector<unsigned long long> h((size_t)grid * 4 * 2);
    CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (size_t i = 0; i < h.size(); i += 2)
        if (h[i + 1]) ghz.push_back((double)h[i] / (double)h[i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("{\"case\": \"%s\", \"tflops_algorithmic\": %.1f, \"frac_of_2500_algorithmic\": %.3f, \"tflops_issued\": %.1f, "
           "\"ms_per_launch\": %.3f, \"in_kernel_clock_ghz\": %.3f}\n",
           name, algorithmic / (ms / 1e3) / 1e12, algorithmic / (ms / 1e3) / 2.5e15, issued / (ms / 1e3) / 1e12, ms / reps,
           ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
    fflush(stdout);
}

// Fourth part ("winograd1d" case, f16): Winograd F(2,3) along x only.  One unit per wave = 2 output rows x 32 tiles of
// two pixels (64 pixels wide) x 32 outputs x 16 input channels: 16 patch reads (4 rows x 4 columns), 16 vector adds
// (v = (d0 - d2, d1 + d2, d2 - d1, d1 - d3) per patch row: 64 v_pk_add_f16), 12 weight-fragment reads (4 positions x 3
// vertical taps) and 24 MFMAs into 8 accumulators (2 rows x 4 positions: 128 registers, two waves per SIMD); the direct
// form spends 36 MFMAs on the same work.  Two 4-wave workgroups of 64 KiB per CU.
constexpr int kLds1d = 64 * 1024;
__global__ __launch_bounds__(256, 2) void wino1d_kernel(const uint4* __restrict__ src, float* __restrict__ sink, int units,
                                                        unsigned long long* __restrict__ clk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < kLds1d / 16; i += 256) ((uint4*)smem)[i] = src[(blockIdx.x * (kLds1d / 16) + i) & 0xFFFFF];
    __syncthreads();
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    f32x16 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][k][i] = 0.f;
    auto frag = [&](int f) { return *(const h8*)(smem + (f & 63) * 1024 + lane * 16); };
    // Patch rows are walked one at a time (as the direct kernel reuses a pixel-row fragment for its vertical taps): row
    // rho feeds outputs r = rho - ky.  Two row buffers: the multiplies read one while the other is loaded and transformed.
    h8 A[4], B[4], wf[3][4];
    auto load_row = [&](h8 (&X)[4], const int f) {
#pragma unroll
        for (int j = 0; j < 4; ++j) X[j] = frag(f + j);
    };
    auto xform = [&](h8 (&X)[4], const int op) {
        if (op == 0) X[0] = sub8(X[0], X[2]);
        else if (op == 1) X[3] = sub8(X[1], X[3]);
        else if (op == 2) { const h8 t = X[1] + X[2]; X[2] = sub8(X[2], X[1]); X[1] = t; }
    };
    load_row(A, 0);
#pragma unroll
    for (int op = 0; op < 3; ++op) xform(A, op);
    auto row = [&](h8 (&V)[4], h8 (&X)[4], const int rho, const int par) {
        load_row(X, par * 16 + ((rho + 1) & 3) * 4 + 5);
        if (rho == 3) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) wf[ky][xi] = frag(32 + par * 12 + ky * 4 + xi);   // next unit's weights
        }
        __builtin_amdgcn_sched_barrier(0);
        int m = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int r = rho - ky;
            if (r < 0 || r > 1) continue;
#pragma unroll
            for (int xi = 0; xi < 4; ++xi) {
                acc[r][xi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ky][xi], V[xi], acc[r][xi], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (m >= 1 && m <= 3) xform(X, m - 1);
                ++m;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) wf[ky][xi] = frag(32 + ky * 4 + xi);
    for (int it = 0; it < units; it += 2) {
#pragma unroll
        for (int par = 0; par < 2; ++par) {
            row(A, B, 0, par);
            row(B, A, 1, par);
            row(A, B, 2, par);
            row(B, A, 3, par);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[r][k][i];
#pragma unroll
    for (int k = 0; k < 4; ++k) s += (float)A[k][0] + (float)B[k][1];
    if (s == 123.456f) sink[blockIdx.x * 256 + tid] = s;
    if (clk && lane == 0) {
        const size_t o = ((size_t)blockIdx.x * 4 + (tid >> 6)) * 2;
        clk[o] = __builtin_amdgcn_s_memtime() - c0;
        clk[o + 1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

static void run_wino1d(const char* name, const uint4* src, float* sink, unsigned long long* clk, double seconds) {
    const int grid = 1024, units = 6000;
    auto k = wino1d_kernel;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds1d));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    double spent = 0;
    while (spent < seconds) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) k<<<grid, 256, kLds1d>>>(src, sink, units, nullptr);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        spent += ms / 1e3;
    }
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<<<grid, 256, kLds1d>>>(src, sink, units, i == reps - 1 ? clk : nullptr);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double issued = (double)reps * grid * 4 * units * 24.0 * 32768.0;
    const double algorithmic = issued * 36.0 / 24.0;
    std::vector<unsigned long long> h((size_t)768 * 4 * 2);
    CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (size_t i = 0; i < h.size(); i += 2)
        if (h[i + 1]) ghz.push_back((double)h[i] / (double)h[i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("{\"case\": \"%s\", \"tflops_algorithmic\": %.1f, \"frac_of_2500_algorithmic\": %.3f, \"tflops_issued\": %.1f, "
           "\"ms_per_launch\": %.3f, \"in_kernel_clock_ghz\": %.3f}\n",
           name, algorithmic / (ms / 1e3) / 1e12, algorithmic / (ms / 1e3) / 2.5e15, issued / (ms / 1e3) / 1e12, ms / reps,
           ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
    fflush(stdout);
}

template <int SHAPE, bool BF, int STAGING = 0>
static void run_case(const char* name, const uint4* src, float* sink, unsigned long long* clk, double seconds,
                     const char* pix = nullptr, unsigned long long pix_bytes = 0) {
    const int grid = 768, pairs = 4000;
    auto k = loop_kernel<SHAPE, BF, STAGING>;
    CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // reach the steady power state first
    double spent = 0;
    while (spent < seconds) {
        CHECK(hipEventRecord(e0));
        for (int i = 0; i < 5; ++i) k<<<grid, 256, kLds>>>(src, sink, pairs, nullptr, pix, pix_bytes);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        spent += ms / 1e3;
    }
    CHECK(hipEventRecord(e0));
    const int reps = 10;
    for (int i = 0; i < reps; ++i) k<<<grid, 256, kLds>>>(src, sink, pairs, i == reps - 1 ? clk : nullptr, pix, pix_bytes);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    // FLOPs: per wave and pair 54 MFMAs of 32x32x16 (or their equivalent) = 54 * 32768
    const double flop = (double)reps * grid * 4 * pairs * 54.0 * 32768.0;
    std::vector<unsigned long long> h((size_t)grid * 4 * 2);
    CHECK(hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> ghz;
    for (size_t i = 0; i < h.size(); i += 2)
        if (h[i + 1]) ghz.push_back((double)h[i] / (double)h[i + 1] * 0.1);
    std::sort(ghz.begin(), ghz.end());
    printf("{\"case\": \"%s\", \"tflops\": %.1f, \"frac_of_2500\": %.3f, \"ms_per_launch\": %.3f, \"in_kernel_clock_ghz\": %.3f}\n",
           name, flop / (ms / 1e3) / 1e12, flop / (ms / 1e3) / 2.5e15, ms / reps, ghz.empty() ? 0.0 : ghz[ghz.size() / 2]);
    fflush(stdout);
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    const size_t n = 1 << 20;  // 16 MiB of random halves in [-1, 1)
    std::vector<uint16_t> h(n * 8);
    uint32_t s = 12345u;
    for (auto& v : h) {
        s = s * 1664525u + 1013904223u;
        // sign, exponent 0x38..0x3b (f16: 0.5 .. 1, bf16 reads the same bits as small normal numbers), random mantissa
        v = (uint16_t)(((s >> 16) & 0x8000u) | 0x3800u | ((s >> 8) & 0x03FFu));
    }
    uint4* src;
    float* sink;
    unsigned long long* clk;
    CHECK(hipMalloc((void**)&src, n * 16));
    CHECK(hipMalloc((void**)&sink, 1024 * 256 * 4));
    CHECK(hipMalloc((void**)&clk, 1024 * 4 * 2 * 8));
    CHECK(hipMemcpy(src, h.data(), n * 16, hipMemcpyHostToDevice));
    if (argc > 2 && !strcmp(argv[2], "winograd")) {
        // A/B on one device in one run: the direct loop, the Winograd loop, the Winograd loop without its adds
        for (int round = 0; round < 3; ++round) {
            run_case<32, false>("direct 3x3, f16 32x32x16, operands from LDS", src, sink, clk, seconds);
            run_wino<1>("winograd F(2x2,3x3), f16: 16 patch + 16 weight reads, 128 v_pk_add_f16, 16 MFMAs per unit", src, sink, clk, seconds);
            run_wino<0>("winograd F(2x2,3x3) without the transform adds (loads and MFMAs only)", src, sink, clk, seconds);
            run_wino1d("winograd F(2,3) along x: 16 patch + 12 weight reads, 64 v_pk_add_f16, 24 MFMAs per unit, two waves per SIMD", src, sink, clk, seconds);
        }
        return 0;
    }
    const bool staging_only = argc > 2 && !strcmp(argv[2], "staging");
    for (int round = 0; round < 2 && !staging_only; ++round) {
        run_case<32, false>("f16 32x32x16", src, sink, clk, seconds);
        run_case<16, false>("f16 16x16x32", src, sink, clk, seconds);
        run_case<32, true>("bf16 32x32x16", src, sink, clk, seconds);
        run_case<16, true>("bf16 16x16x32", src, sink, clk, seconds);
    }
    // staging: pixel windows streamed through 2 MiB (L2-resident), 128 MiB (Infinity Cache) and 2 GiB (HBM)
    char* pix;
    const unsigned long long big = 2ull << 30;
    CHECK(hipMalloc((void**)&pix, big));
    for (unsigned long long o = 0; o < big; o += n * 16) CHECK(hipMemcpy(pix + o, src, n * 16, hipMemcpyDeviceToDevice));
    for (int round = 0; round < 2; ++round) {
        run_case<32, false, 0>("f16 32x32x16 loop only", src, sink, clk, seconds);
        run_case<32, false, 1>("f16 32x32x16 + DMA free-running, pixels from 2 MiB (L2)", src, sink, clk, seconds, pix, 2ull << 20);
        run_case<32, false, 1>("f16 32x32x16 + DMA free-running, pixels from 128 MiB (Infinity Cache)", src, sink, clk, seconds, pix, 128ull << 20);
        run_case<32, false, 1>("f16 32x32x16 + DMA free-running, pixels from 2 GiB (HBM)", src, sink, clk, seconds, pix, big);
        run_case<32, false, 2>("f16 32x32x16 + DMA + wait and barrier per stage, pixels from 2 MiB (L2)", src, sink, clk, seconds, pix, 2ull << 20);
        run_case<32, false, 2>("f16 32x32x16 + DMA + wait and barrier per stage, pixels from 128 MiB (Infinity Cache)", src, sink, clk, seconds, pix, 128ull << 20);
        run_case<32, false, 2>("f16 32x32x16 + DMA + wait and barrier per stage, pixels from 2 GiB (HBM)", src, sink, clk, seconds, pix, big);
    }
    return 0;
}
