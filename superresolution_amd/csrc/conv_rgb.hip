// The RGB layer of the three networks (3x3 convolution with 3 outputs and the fused post-processing:
// realesrgan_model.py:97 conv_last, edsr_model.py:100-104 tail, rcan_model.py:140-143 tail) as a 1x1 GEMM over the
// pixel patch followed by a nine-tap gather.
//
// With 3 outputs a 32-wide MFMA tile of the direct kernel multiplies 29 columns of zeros.  Here the 32 columns of the
// matrix instruction are (tap, output) pairs: column 3 * tap + o holds W[o][.][tap], 27 of 32 columns in use.  One
// chain of cin / 16 MFMAs per 32 patch pixels gives Y[pixel][tap][o] = sum_c x[pixel][c] * W[o][c][tap], nine times
// fewer matrix instructions than a multiply per tap, and an output pixel is the sum of nine Y values of its
// neighbours, fetched from LDS.  Every patch pixel is multiplied exactly once, so the pixel fragments go from global
// memory straight into registers (16 bytes per lane, 1 KiB contiguous per wave and plane): no staging, no barrier
// before the one in front of the gather.  The weights are read from the direct kernel's packed image
// (srmi_pack_conv3x3), cin / 16 fragments per lane, once per workgroup.
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "devutil.h"
#include "conv_epilogue.h"  // make_rsrc, SRMI_OOB_OFFSET, kNW

namespace srmi {
namespace {

constexpr int kRows = 16;                          // output rows of a tile (32 pixels wide)
constexpr int kPH = kRows + 2, kPW = kPatchW;      // the patch: one pixel of halo on every side
constexpr int kPatch = kPH * kPW;                  // 612 pixels
constexpr int kBlocks = (kPatch + 31) / 32;        // MFMA column blocks of 32 patch pixels
[[maybe_unused]] constexpr int kBPW = kBlocks / kNW;  // blocks per wave (used in device code)
constexpr int kPPad = kBlocks * 32;
constexpr int kCols = 27;                          // (tap, output) pairs
constexpr int kLdsBytes = kCols * kPPad * 4;       // Y[column][patch pixel], f32
static_assert(kBlocks % kNW == 0, "every wave multiplies the same number of pixel blocks");
static_assert(kRows == 4 * kNW, "the gather gives every wave four output rows");

template <int DT>
__device__ __forceinline__ f32x16 mma16(const u32x4& a, const u32x4& b, const f32x16& c) {
    if constexpr (DT == SRMI_F16)
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// KC: input channels / 64.  wrow: rows per tap of the packed weight image (the cout tile it was packed for).
template <int DT, int KC>
__global__ __launch_bounds__(64 * kNW, 2) void conv_rgb_kernel(const ConvParams p, const int wrow) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* Y = (float*)smem;

    const int tid = threadIdx.x;
    const int lane = tid & 63, lx = lane & 31, lh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // consecutive workgroups of the launch order go to different XCDs (one L2 each): renumber so that an XCD
    // works on a contiguous run of tiles, whose halo rows and columns it then finds in its own L2
    int tile, n;
    {
        const unsigned gx = gridDim.x, total = gx * gridDim.y;
        const unsigned lin = blockIdx.x + gx * blockIdx.y;
        const unsigned xcd = lin & 7u, j = lin >> 3;
        const unsigned q = total >> 3, r = total & 7u;
        const unsigned id = xcd * q + (xcd < r ? xcd : r) + j;
        tile = (int)(id % gx);
        n = (int)(id / gx);
    }
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int x0 = tx * kTileW, y0 = ty * kRows;

    // ---- weights: column lx = 3 * tap + o, channels 8 * lh .. + 7 of every 16-channel plane ------------------
    u32x4 wf[KC * 4];
    {
        const int tap = lx / 3, o = lx - 3 * tap;
        const char* wp = p.wpk + ((size_t)tap * wrow + o) * 32 + lh * 16;
#pragma unroll
        for (int k = 0; k < KC * 4; ++k) {
            const u32x4 zero = {0u, 0u, 0u, 0u};
            wf[k] = lx < kCols ? *(const u32x4*)(wp + (size_t)k * 9 * wrow * 32) : zero;
        }
    }

    // ---- this wave's pixel blocks: per-lane byte offset of patch pixel (block * 32 + lx) in a plane ----------
    unsigned poff[kBPW];
#pragma unroll
    for (int b = 0; b < kBPW; ++b) {
        const int pp = (wave * kBPW + b) * 32 + lx;
        const int rho = pp / kPW, xi = pp - rho * kPW;
        const int gy = y0 - 1 + rho, gx = x0 - 1 + xi;
        const bool ok = pp < kPatch && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        poff[b] = ok ? (unsigned)(gy * p.w + gx) * 32u + (unsigned)lh * 16u : SRMI_OOB_OFFSET;  // outside: zeros
    }
    const char* in_img = p.in.ptr + (long long)n * p.in.img_stride_b;

    // A unit = one block x 64 channels: four 16-byte loads per lane, four MFMAs.  D units are in flight.
    constexpr int U = kBPW * KC, D = U < 6 ? U : 6;
    u32x4 fr[D][4];
    auto fetch = [&](const int u) {
        const int b = u / KC, kc = u % KC;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const auto rs = make_rsrc(in_img + (long long)(kc * 4 + j) * p.in.plane_stride_b, p.in_plane_bytes);
            fr[u % D][j] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, poff[b], 0, 0));
        }
    };
#pragma unroll
    for (int u = 0; u < D; ++u) fetch(u);
    f32x16 acc;
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int b = u / KC, kc = u % KC;
        if (kc == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc = mma16<DT>(wf[kc * 4 + j], fr[u % D][j], acc);
        if (u + D < U) fetch(u + D);
        if (kc == KC - 1) {
            // register i of a lane: column 8 * (i / 4) + 4 * lh + i % 4 of pixel lx
            float* yb = Y + (wave * kBPW + b) * 32 + lx;
#pragma unroll
            for (int i = 0; i < 12; ++i) yb[(8 * (i >> 2) + 4 * lh + (i & 3)) * kPPad] = acc[i];
            if (lh == 0) {
#pragma unroll
                for (int i = 12; i < 15; ++i) yb[(24 + (i & 3)) * kPPad] = acc[i];
            }
        }
    }
    __syncthreads();

    // ---- gather: out(y, x)[o] = sum over taps of Y[3 * tap + o][(y + dy) * 34 + x + dx] -------------------------
    const float b0 = p.bias[0], b1 = p.bias[1], b2 = p.bias[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int row = wave * 4 + 2 * t + lh;
        const int y = y0 + row, x = x0 + lx;
        float v[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* yp = Y + (row + dy) * kPW + lx + dx + (dy * 3 + dx) * 3 * kPPad;
#pragma unroll
                for (int o = 0; o < 3; ++o) v[o] += yp[o * kPPad];
            }
        v[0] += b0; v[1] += b1; v[2] += b2;
        if (y < p.h && x < p.w) {
            const long long pix = (long long)y * p.w + x;
            float rgb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                rgb[c] = p.tail_mat[3 * c] * v[0] + p.tail_mat[3 * c + 1] * v[1] + p.tail_mat[3 * c + 2] * v[2] + p.tail_bias[c];
            if (p.tail_mode == SRMI_TAIL_F32_NCHW) {
                float* o = (float*)p.tail_out + (long long)n * 3 * p.h * p.w + pix;
#pragma unroll
                for (int c = 0; c < 3; ++c) o[(long long)c * p.h * p.w] = rgb[c];
            } else {
                unsigned char* o = (unsigned char*)p.tail_out + ((long long)n * p.h * p.w + pix) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) o[c] = to_u8(rgb[c], p.tail_lo, p.tail_hi, p.tail_scale, p.tail_round);
            }
        }
    }
#endif
}

template <int DT, int KC>
int configure_one() {
    SRMI_HIP(hipFuncSetAttribute((const void*)conv_rgb_kernel<DT, KC>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    return 0;
}

template <int DT>
int launch_dt(const ConvParams& p, int wrow, dim3 grid, hipStream_t s) {
    switch (p.ncb / 4) {
        case 1: hipLaunchKernelGGL((conv_rgb_kernel<DT, 1>), grid, dim3(64 * kNW), kLdsBytes, s, p, wrow); break;
        case 2: hipLaunchKernelGGL((conv_rgb_kernel<DT, 2>), grid, dim3(64 * kNW), kLdsBytes, s, p, wrow); break;
        case 3: hipLaunchKernelGGL((conv_rgb_kernel<DT, 3>), grid, dim3(64 * kNW), kLdsBytes, s, p, wrow); break;
        case 4: hipLaunchKernelGGL((conv_rgb_kernel<DT, 4>), grid, dim3(64 * kNW), kLdsBytes, s, p, wrow); break;
        default: SRMI_FAIL("srmi_conv3x3: the RGB gather kernel takes 64, 128, 192 or 256 input channels");
    }
    SRMI_HIP(hipGetLastError());
    return 0;
}

}  // namespace
}  // namespace srmi

using namespace srmi;

// The dynamic-LDS limit is a per-device attribute of each function: set once per device.
int srmi_conv_rgb_prepare_device() {
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    SRMI_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (tracked && (configured.load(std::memory_order_acquire) >> dev & 1ull)) return 0;
    if (configure_one<SRMI_F16, 1>() || configure_one<SRMI_F16, 2>() || configure_one<SRMI_F16, 3>() || configure_one<SRMI_F16, 4>() ||
        configure_one<SRMI_BF16, 1>() || configure_one<SRMI_BF16, 2>() || configure_one<SRMI_BF16, 3>() || configure_one<SRMI_BF16, 4>())
        return -1;
    if (tracked) configured.fetch_or(1ull << dev, std::memory_order_release);
    return 0;
}

// Whether the descriptor's RGB layer can take this kernel (the caller has validated the descriptor).
bool srmi_conv_rgb_takes(const srmi_conv_desc_t* d) {
    return d->tail_mode != SRMI_TAIL_NONE && (d->dtype == SRMI_F16 || d->dtype == SRMI_BF16) && d->algo == SRMI_ALGO_DIRECT &&
           d->cout == 3 && d->cin % 64 == 0 && d->cin <= 256 && d->act == SRMI_ACT_NONE && !d->in_up2 && !d->in_gate.ptr &&
           !d->res1.ptr && !d->res2.ptr && !d->ca.sums && !d->chansum;
}

int srmi_conv_rgb_launch(const ConvParams& p, int cout_tile, hipStream_t s) {
    if (srmi_conv_rgb_prepare_device()) return -1;
    const int tiles_y = (p.h + kRows - 1) / kRows;
    ConvParams q = p;
    q.tiles_y = tiles_y;
    dim3 grid(q.tiles_x * tiles_y, p.n, 1);
    return p.in.dtype == SRMI_F16 ? launch_dt<SRMI_F16>(q, cout_tile, grid, s) : launch_dt<SRMI_BF16>(q, cout_tile, grid, s);
}
