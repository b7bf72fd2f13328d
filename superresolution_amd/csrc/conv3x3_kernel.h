// The direct 3x3 convolution kernel (template over compute type, rows per wave, output blocks, gated input, loop form) and its
// launch helpers.  Included by one translation unit per compute type (conv3x3_f16.hip, _bf16, _f32, _i8), so that the
// instantiations compile side by side; the host entry point is in conv3x3.hip.
//
// 3x3 convolution as implicit GEMM on the gfx950 matrix cores, with the networks' elementwise
// tails fused into the epilogue.
//
// Work decomposition
//   workgroup  = 256 threads (4 waves) -> a spatial tile of (4*MB) rows x 32 columns of one image
//                and a tile of 32*NB output channels.  LDS use is kept below 80 KiB so that at
//                least two workgroups share a CU: one workgroup's loads, barriers and epilogue
//                stores run under the other's MFMAs.
//   wave       = MB consecutive rows of the tile (one row of 32 pixels = one MFMA column block)
//                x NB blocks of 32 output channels: MB*NB accumulators of 32x32 (f32).
//   MFMA roles = A operand: weights (rows = output channel), B operand: pixels (columns), so a lane
//                ends up holding output channels of ONE pixel and the epilogue reads residuals and
//                writes results with 16-byte accesses.
//
// Data movement: the reduction over input channels runs in STAGES of 32 bytes of channels per
// pixel (16 halves or 8 floats = one plane of the activation layout, so the fetched cache lines
// are used in full).  Per stage both operands are copied
// global -> LDS by LDS-DMA (buffer_load ... lds, 16 B per lane, no VGPRs):
//   weights   : pre-swizzled on the host into exactly the LDS image, a linear copy;
//   pixels    : (4*MB+2) x 34 halo patch; the per-lane source offset is computed once per
//               workgroup, out-of-image lanes carry an offset beyond the buffer descriptor's range
//               and therefore read zeros (= the convolution's zero padding); nearest-x2
//               upsampling is a pure address map.
// Stages form a ring of SRMI_RING slots (2: with the 12-row tile a workgroup then needs under 53 KiB
// and three fit a CU, which beats a deeper ring at two per CU); the next stage's pieces are issued
// between the multiply steps of the current one, counted vmcnt waits and one raw s_barrier per stage.
// Reductions of 8 stages or more run a variant of the loop whose barrier sits behind the stage's
// last LDS read instead of its last MFMA (see the main loop).
// LDS rows are 32 bytes; the 16-byte chunk index is XORed with (row>>3)&1 so that the 16-lane
// groups of ds_read_b128 touch 16 different bank slots (conflict-free).
//
// The kernel is written in bytes: f16, bf16 and f32 differ only in the MFMA issued per pair of
// 16-byte chunks (one 32x32x16 half MFMA, or four 32x32x2 f32 MFMAs) and in the epilogue casts.
#pragma once
#include <stdlib.h>

#include <atomic>

#include "common.h"
#include "devutil.h"

// Timing ablations, in-kernel stamps and alternative schedules are diagnostic builds (tools/build_variant.sh
// passes -DSRMI_DIAG with them); the shipped library is built without any of them.
#if !defined(SRMI_DIAG) && (defined(SRMI_ABL_NO_DMA) || defined(SRMI_ABL_NO_DMA_A) || defined(SRMI_ABL_NO_DMA_B) || \
                            defined(SRMI_ABL_NO_LDS) || defined(SRMI_STAMPS) || defined(SRMI_BULK_ISSUE) ||          \
                            defined(SRMI_SINGLE_FRAG) || defined(SRMI_NO_XCD_MAP) || defined(SRMI_CLOCK) || defined(SRMI_STAMPS_EPI) || defined(SRMI_GAP_OVERRIDE) || defined(SRMI_NW) || defined(SRMI_RING) || defined(SRMI_ABL_DX_COPY))
#error "diagnostic kernel variants need -DSRMI_DIAG"
#endif

#include "conv_epilogue.h"

namespace srmi {

typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int DT>
__device__ __forceinline__ void mma_chunk(f32x16& c, const u32x4& wfrag, const u32x4& pfrag) {
    if constexpr (DT == SRMI_F16) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wfrag),
                                                   __builtin_bit_cast(f16x8, pfrag), c, 0, 0, 0);
    } else if constexpr (DT == SRMI_BF16) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wfrag),
                                                    __builtin_bit_cast(bf16x8, pfrag), c, 0, 0, 0);
    } else if constexpr (DT == SRMI_I8) {
        // int8 operands, exact int32 accumulation: one instruction multiplies the 32 input channels of a plane
        // (16 bytes per lane half).  The accumulator registers are shared with the float paths bit for bit.
        c = __builtin_bit_cast(f32x16, __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, wfrag),
                                                                             __builtin_bit_cast(i32x4, pfrag),
                                                                             __builtin_bit_cast(i32x16, c), 0, 0, 0));
    } else {
        // whole-vector casts first: bit-casting a single element of a vector reference makes
        // hipcc (ROCm 7.2) read element 0 for every j
        const f32x4 wv = __builtin_bit_cast(f32x4, wfrag);
        const f32x4 pv = __builtin_bit_cast(f32x4, pfrag);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(wv[j], pv[j], c, 0, 0, 0);
    }
}


// KT: taps per direction.  3: the 3x3 convolution.  2: one phase of a 3x3 convolution behind a nearest-x2 upsampling, which
// on the source grid is a 2x2 convolution (below).
template <int MB, int NB, int KT = 3>
struct ConvGeom {
    static constexpr int TH = kNW * MB, PH = TH + KT - 1, PW = kPatchW;
    static constexpr int NCH = PH * PW * 2;                // 16-byte chunks of one pixel-patch stage
    static constexpr int A_PIECES = (NCH + 63) / 64;       // 1-KiB LDS-DMA pieces
    static constexpr int A_BYTES = A_PIECES * 1024;
    static constexpr int B_BYTES = KT * KT * 32 * NB * 32;  // weight image of one stage
    static constexpr int B_PIECES = B_BYTES / 1024;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    // ring of stages: three deep (two stages in flight behind the one being multiplied) where two
    // workgroups still fit a CU's 160 KiB, else two deep
#ifndef SRMI_RING
#define SRMI_RING 2
#endif
    static constexpr int NS = (SRMI_RING == 3 && 3 * STAGE_BYTES <= 80 * 1024) ? 3 : 2;
    static constexpr int LDS_BYTES = NS * STAGE_BYTES;
    // the epilogue's bias and tile-sum scratch alias the stage buffers (written after the last stage), so
    // an 8-row tile needs exactly 40 KiB and four workgroups share a CU's 160 KiB, a 12-row tile 48 KiB (three)
    static constexpr int LDS_TOTAL = LDS_BYTES;
    static constexpr int BIAS_OFFSET = 8192;               // behind the tile sums (kNW * 32 * NB floats <= 8 KiB)
    static constexpr int GATE_BYTES = 2048;                // gated input: float[cin <= 256] behind the stage ring, then
                                                           // 1 KiB that absorbs the writes of pieces a wave does not own
    // waves per SIMD the register budget is held to: 4 for the 8-row tile (<= 128 registers), else 2
    static constexpr int MIN_WAVES = (MB == 2 && NB == 1 && kNW == 4) ? 4 : (kNW == 4 ? 2 : 4);
};

// CARRIED: the main loop's form (below), chosen by the host from the number of stages: one kernel holds one form, so the
// accumulators of the two loops never meet in copies (or spills) in front of the epilogue.
//
// KT == 2, the sub-pixel form of "nearest-x2 upsampling, then 3x3 convolution": output pixel (2y + py, 2x + px) only sees
// the source pixels (y + py - 1 .. y + py, x + px - 1 .. x + px), each through the sum of the taps that land on it, so
// the layer is four 2x2 convolutions on the SOURCE grid (4/9 of the multiplies), one per phase (py, px).  The host packs
// the summed weights phase-major (srmi_pack_conv3x3_up2); a workgroup's output-channel tile lies inside one phase, which
// shifts its patch origin, and the results leave through the pixel-shuffle store (ps_r == 2).
template <int DT, int MB, int NB, int GATE, bool CARRIED = false, int KT = 3>
__global__ __launch_bounds__(64 * kNW, (GATE ? 2 : ConvGeom<MB, NB, KT>::MIN_WAVES)) void conv3x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
    static_assert(KT == 3 || (KT == 2 && !GATE && !CARRIED), "the 2x2 form: plain input, classic loop");
    using G = ConvGeom<MB, NB, KT>;
    constexpr int TH = G::TH, PW = G::PW;
    constexpr int APW = (G::A_PIECES + kNW - 1) / kNW, BPW = (G::B_PIECES + kNW - 1) / kNW;  // pieces per wave

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Workgroups are dispatched round-robin over the 8 XCDs (each with its own L2) in linear
    // launch order.  Re-number them so that one XCD gets a contiguous run of the order
    // (output-channel tile fastest, then spatial tile, then image): neighbouring spatial tiles share
    // halo rows and the output-channel tiles of one spatial tile share the whole pixel patch in L2.
#ifndef SRMI_NO_XCD_MAP
    int tile, ct, n;
    {
        const unsigned gx = gridDim.x, gy = gridDim.y;
        const unsigned total = gx * gy * gridDim.z;
        const unsigned lin = blockIdx.x + gx * (blockIdx.y + gy * blockIdx.z);
        const unsigned xcd = lin & 7u, j = lin >> 3;
        const unsigned q = total >> 3, r = total & 7u;
        unsigned id = xcd * q + (xcd < r ? xcd : r) + j;
        ct = (int)(id % gy); id /= gy;
        tile = (int)(id % gx);
        n = (int)(id / gx);
    }
#else
    const int tile = blockIdx.x, ct = blockIdx.y, n = blockIdx.z;
#endif
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int x0 = tx * kTileW, y0 = ty * TH;
    const int nstage = p.ncb;  // one stage per input plane

    // ---- per-lane source offsets of the pixel patch (identical for every stage) ----------------
    // LDS chunk i holds logical chunk c = (i&1) ^ ((pixel>>3)&1) of patch pixel i>>1.
    unsigned aoff[APW];
    {
        // 2x2 form: the phase of this output-channel tile moves the patch origin from (y0 - 1, x0 - 1) to (y0 - 1 + py, x0 - 1 + px)
        const int phase = KT == 2 ? (ct * 32 * NB) / p.ps_c : 0;
        const int sy = phase >> 1, sx = phase & 1;
        const int srcW = p.in_up2 ? (p.w >> 1) : p.w;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int i = (wave + kNW * j) * 64 + lane;
            const int pp = i >> 1;
            const int c = (i & 1) ^ ((pp >> 3) & 1);
            const int rho = pp / PW, xi = pp - rho * PW;
            int gy = y0 - 1 + rho + sy, gx = x0 - 1 + xi + sx;
            const bool ok = (i < G::NCH) && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
            if (p.in_up2) { gy >>= 1; gx >>= 1; }
            aoff[j] = ok ? ((unsigned)(gy * srcW + gx) * 32u + (unsigned)c * 16u) : SRMI_OOB_OFFSET;
        }
    }
    const char* in_img = p.in.ptr + (long long)n * p.in.img_stride_b;
    const char* w_tile = p.wpk + (size_t)ct * nstage * G::B_BYTES;
    const unsigned plane_bytes = p.in_plane_bytes;

    // stage s = input plane s: 32 bytes of channels per pixel.  A wave's share of a
    // stage is PPW LDS-DMA pieces of 1 KiB: piece q < APW is pixel-patch piece (wave + 4q), the rest
    // are weight pieces.
    constexpr int PPW = APW + BPW;
    auto issue_piece = [&](const __amdgpu_buffer_rsrc_t ra, const __amdgpu_buffer_rsrc_t rb, char* abuf, const int q) {
#ifdef SRMI_ABL_NO_DMA  // timing ablation (wrong results): nothing is staged after the prologue
        return;
#endif
#ifdef SRMI_ABL_NO_DMA_A  // timing ablation: no pixel pieces
        if (q < APW) return;
#endif
#ifdef SRMI_ABL_NO_DMA_B  // timing ablation: no weight pieces
        if (q >= APW) return;
#endif
        if (q < APW) {
            const int k = wave + kNW * q;
            if (k < G::A_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(abuf + k * 1024),
                                                         16, aoff[q < APW ? q : 0], 0, 0, 0);
        } else {
            const int k = wave + kNW * (q - APW);
            if (k < G::B_PIECES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(abuf + G::A_BYTES + k * 1024),
                                                         16, k * 1024 + lane * 16, 0, 0, 0);
        }
    };
    auto pixel_rsrc = [&](int s) {
        if constexpr (GATE == 2)  // the two f32 planes that hold the stage's 16 channels
            return make_rsrc(in_img + (long long)(2 * s) * p.in.plane_stride_b, (unsigned)p.in.plane_stride_b + plane_bytes);
        else
            return make_rsrc(in_img + (long long)s * p.in.plane_stride_b, plane_bytes);
    };
    auto weight_rsrc = [&](int s) { return make_rsrc(w_tile + (size_t)s * G::B_BYTES, G::B_BYTES); };
    auto issue = [&](int s, int buf) {
        const auto ra = pixel_rsrc(s);
        const auto rb = weight_rsrc(s);
#pragma unroll
        for (int q = 0; q < PPW; ++q) issue_piece(ra, rb, smem + buf * G::STAGE_BYTES, q);
    };

    // ---- gated input (channel attention folded into the consumer) --------------------------------
    // The convolution reads  in + gate_alpha * (in_gate * gate[n][c])  instead of `in`: the pixel pieces of a
    // stage are fetched into registers (same per-lane offsets, out-of-image lanes read zeros from both
    // tensors), combined with the arithmetic of ca_apply_kernel and written to the stage slot with
    // ds_write; interior pixels of the tile are also stored to `in_sum`, the trunk the next block adds to.
    // GATE == 2: `in` and `in_sum` are f32 tensors (the trunk kept in single precision): a 16-byte chunk of a stage is
    // the whole 32-byte pixel of one f32 plane (plane 2s + chunk), fetched and stored as two 16-byte halves.
    static_assert(!GATE || (G::NS == 2 && (DT == SRMI_F16 || DT == SRMI_BF16)), "gated input: half types, two-slot ring");
    u32x4 gx[GATE ? APW : 1], gt[GATE ? APW : 1], gx1[GATE == 2 ? APW : 1];
    constexpr unsigned kOobF32 = 0xFFFFFF00u;  // out of range, and still so 16 bytes further
    // f32 trunk: per-lane offset of this lane's pixel in the two-plane window of a stage
    auto x_offset = [&](const int q) {
        const unsigned a = aoff[q];
        return a == SRMI_OOB_OFFSET ? kOobF32 : (a & ~31u) + ((a >> 4) & 1u) * (unsigned)p.in.plane_stride_b;
    };
    auto s_offset = [&](const int q) {
        const unsigned a = aoff[q];
        return (a & ~31u) + ((a >> 4) & 1u) * (unsigned)p.in_sum.plane_stride_b;
    };
    unsigned keep = 0;  // bit j: piece j of this lane is an in-image pixel of the tile's interior
    const float* gate_lds = (const float*)(smem + G::LDS_BYTES);
    const char* gate_img = nullptr;
    char* sum_img = nullptr;
    if constexpr (GATE) {
        gate_img = p.in_gate.ptr + (long long)n * p.in_gate.img_stride_b;
        sum_img = p.in_sum.ptr ? p.in_sum.ptr + (long long)n * p.in_sum.img_stride_b : nullptr;
#pragma unroll
        for (int j = 0; j < APW; ++j) {
            const int i = (wave + kNW * j) * 64 + lane;
            const int pp = i >> 1;
            const int rho = pp / PW, xi = pp - rho * PW;
            const bool interior = rho >= 1 && rho <= TH && xi >= 1 && xi <= kTileW;
            if (interior && aoff[j] != SRMI_OOB_OFFSET) keep |= 1u << j;
        }
    }
    auto gate_rsrc = [&](int s) { return make_rsrc(gate_img + (long long)s * p.in_gate.plane_stride_b, plane_bytes); };
    auto sum_rsrc = [&](int s) {
        if constexpr (GATE == 2)
            return make_rsrc(sum_img + (long long)(2 * s) * p.in_sum.plane_stride_b,
                             sum_img ? (unsigned)p.in_sum.plane_stride_b + plane_bytes : 0u);
        else
            return make_rsrc(sum_img + (long long)s * p.in_sum.plane_stride_b, sum_img ? plane_bytes : 0u);
    };
    auto gate_fetch = [&](const int q, const bool really, const __amdgpu_buffer_rsrc_t ra, const __amdgpu_buffer_rsrc_t rt) {
        if constexpr (GATE == 2) {
            const unsigned xo = really ? x_offset(q) : kOobF32;
            gx[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, xo, 0, 0));
            gx1[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, xo + 16u, 0, 0));
            gt[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, really ? aoff[q] : SRMI_OOB_OFFSET, 0, 0));
        } else if constexpr (GATE == 1) {
            const unsigned off = really ? aoff[q] : SRMI_OOB_OFFSET;
            gx[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 0));
            gt[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, off, 0, 0));
        }
    };
    // One STEP combines one channel of one piece (8 steps per piece, 8 * APW per stage).  The main loop runs
    // the steps of stage s+1 between the MFMAs of stage s, so the conversions hide under the matrix pipe;
    // the last step of a piece writes its 16 bytes to the stage slot (and to in_sum) and then requests the same
    // piece of stage s+2, which so has a whole stage to arrive.
    constexpr int kGateSteps = 8 * APW;
    const float gate_alpha = p.gate_alpha;
    f32x4 gg0 = {0.f, 0.f, 0.f, 0.f}, gg1 = {0.f, 0.f, 0.f, 0.f};
    u32x4 gpk = {0u, 0u, 0u, 0u};
    float gprev = 0.f;
    // The steps contain no branch: between the request of a piece and its use the compiler must be able to COUNT
    // the vector-memory operations issued (a conditional one in between makes it wait for all of them, the
    // weight pieces just requested included).  What must not happen is masked instead: a piece this wave does
    // not own reads with out-of-range offsets and writes to the spare KiB, a stage that does not exist is
    // requested with out-of-range offsets (zeros, no traffic) and stored with them (dropped).
    auto gate_step = [&](const int step, const int s, char* abuf, const __amdgpu_buffer_rsrc_t rx, const bool live,
                         const bool fetch_next, const __amdgpu_buffer_rsrc_t ra_next, const __amdgpu_buffer_rsrc_t rt_next) {
        if constexpr (GATE) {
            const int q = step >> 3, j = step & 7;
            const int k = wave + kNW * q;
            const bool mine = k < G::A_PIECES;
            if (j == 0) {
                const int i = k * 64 + lane;
                const int c = (i & 1) ^ ((i >> 4) & 1);  // logical chunk of this lane's LDS chunk
                const float* g = gate_lds + s * 16 + c * 8;
                gg0 = *(const f32x4*)g;
                gg1 = *(const f32x4*)(g + 4);
            }
            float xv, tv;
            if constexpr (DT == SRMI_F16) tv = (float)__builtin_bit_cast(f16x8, gt[q])[j];
            else tv = (float)__builtin_bit_cast(bf16x8, gt[q])[j];
            if constexpr (GATE == 2) xv = __uint_as_float(j < 4 ? gx[q][j & 3] : gx1[q][j & 3]);
            else if constexpr (DT == SRMI_F16) xv = (float)__builtin_bit_cast(f16x8, gx[q])[j];
            else xv = (float)__builtin_bit_cast(bf16x8, gx[q])[j];
            const float o = (tv * (j < 4 ? gg0[j & 3] : gg1[j & 3])) * gate_alpha + xv;
            if constexpr (GATE == 2) {  // the single-precision sum takes the place of x until it is stored
                if (j < 4) gx[q][j & 3] = __float_as_uint(o);
                else gx1[q][j & 3] = __float_as_uint(o);
            }
            if (j & 1) {
                if constexpr (DT == SRMI_F16) {
                    typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
                    const f16x2 h = {to_f16_sat(gprev), to_f16_sat(o)};
                    gpk[j >> 1] = __builtin_bit_cast(unsigned, h);
                } else {
                    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                    const bf16x2 h = {(__bf16)gprev, (__bf16)o};
                    gpk[j >> 1] = __builtin_bit_cast(unsigned, h);
                }
            } else {
                gprev = o;
            }
            if (j == 7) {
                char* dst = mine ? abuf + k * 1024 : smem + G::LDS_BYTES + 1024;
                *(u32x4*)(dst + lane * 16) = gpk;
                const bool store = live && ((keep >> q) & 1u);
                if constexpr (GATE == 2) {
                    const unsigned so = store ? s_offset(q) : kOobF32;
                    __builtin_amdgcn_raw_buffer_store_b128(gx[q], rx, so, 0, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(gx1[q], rx, so + 16u, 0, 0);
                } else {
                    __builtin_amdgcn_raw_buffer_store_b128(gpk, rx, store ? aoff[q] : SRMI_OOB_OFFSET, 0, 0);
                }
                gate_fetch(q, fetch_next, ra_next, rt_next);
            }
        }
    };
    // vector-memory operations of this wave that are younger than the weight pieces of a stage when that stage
    // starts: per pixel piece the in_sum store(s) and the loads of the next fetch
    constexpr int kGateYounger = (GATE == 2 ? 5 : 3) * APW;
    static_assert(!GATE || kGateYounger <= 63, "vmcnt counts up to 63");

    f32x16 acc[MB][NB];
#pragma unroll
    for (int r = 0; r < MB; ++r)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][b][i] = 0.f;

    const int lx = lane & 31, lh = lane >> 5;
    const int p0 = wave * MB * PW + lx;            // LDS pixel index of (first row of this wave, lx)
    const int bm = (lx >> 3) & 1;                  // weight-row swizzle term (row = 32*k + lx)

#ifdef SRMI_STAMPS
    // diagnostic build: lane 0 of every wave records s_memtime at each phase boundary into
    // [workgroup][wave][40]; slot 39 holds (XCC_ID << 32) | HW_ID
    unsigned long long* stamp_buf = (unsigned long long*)p.tail_out;
    int stamp_i = 0;
    const size_t stamp_base = (((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * kNW + wave) * 40;
    if (stamp_buf && lane == 0) {
        unsigned hw_, xcc_;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc_));
        stamp_buf[stamp_base + 39] = ((unsigned long long)xcc_ << 32) | hw_;
    }
#define STAMP()                                                                                   \
    do {                                                                                          \
        if (stamp_buf && lane == 0 && stamp_i < 39) {                                             \
            unsigned long long t_;                                                                \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
            stamp_buf[stamp_base + stamp_i++] = t_;                                               \
        }                                                                                         \
    } while (0)
#else
#define STAMP() do {} while (0)
#endif
    // LDS-DMA operations this wave issues per stage (its share of the 1-KiB pieces): the count a
    // counted vmcnt wait must leave outstanding to keep the NEXT stage in flight.
    int my_pieces = 0;
#pragma unroll
    for (int j = 0; j < APW; ++j) my_pieces += (wave + kNW * j < G::A_PIECES) ? 1 : 0;
#pragma unroll
    for (int j = 0; j < BPW; ++j) my_pieces += (wave + kNW * j < G::B_PIECES) ? 1 : 0;
    my_pieces = __builtin_amdgcn_readfirstlane(my_pieces);

#ifdef SRMI_CLOCK
    // diagnostic build: shader-clock and 100 MHz real-time counters at both ends of the workgroup's life,
    // [workgroup][wave][4] = {cycles0, real0, cycles1, real1}; held clock = d(cycles) / d(real) * 100 MHz
    unsigned long long* clock_buf = (unsigned long long*)p.tail_out;
    const size_t clock_base = (((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * kNW + wave) * 4;
    const unsigned long long clk_c0 = __builtin_amdgcn_s_memtime(), clk_r0 = __builtin_amdgcn_s_memrealtime();
#endif
    STAMP();
    if constexpr (GATE) {
        if (tid < p.ncb * 16) ((float*)(smem + G::LDS_BYTES))[tid] = p.gate[(long long)n * p.ncb * 16 + tid];
        {
            const auto ra = pixel_rsrc(0);
            const auto rt = gate_rsrc(0);
            const auto rb = weight_rsrc(0);
#pragma unroll
            for (int q = APW; q < PPW; ++q) issue_piece(ra, rb, smem, q);
#pragma unroll
            for (int q = 0; q < APW; ++q) gate_fetch(q, true, ra, rt);
        }
        __syncthreads();  // the gate vector is in LDS
        {
            const int s1 = nstage > 1 ? 1 : 0;
            const auto rx = sum_rsrc(0);
            const auto ra = pixel_rsrc(s1);
            const auto rt = gate_rsrc(s1);
#pragma unroll
            for (int st = 0; st < kGateSteps; ++st) gate_step(st, 0, smem, rx, true, nstage > 1, ra, rt);
        }
    } else {
        issue(0, 0);
    }
    // this thread's bias value waits in a register until the stage buffers are free
    const float bias_v = tid < 32 * NB ? p.bias[ct * 32 * NB + tid] : 0.f;
    float scale_v = 0.f;
    if constexpr (DT == SRMI_I8) scale_v = tid < 32 * NB ? p.wscale[ct * 32 * NB + tid] : 0.f;
    if (G::NS == 3 && nstage > 1) issue(1, 1);
    STAMP();

    // One fragment set = everything the MFMAs of one horizontal tap (dx) read from LDS.  Two sets
    // are alternated so that the ds_reads of step i+1 are in flight while step i multiplies.
    struct Frag {
        u32x4 wf[KT][NB];
        u32x4 pf[MB + KT - 1];
    };
    auto load_frag = [&](Frag& f, const char* Ab, const char* Bb, const int dx) {
#ifdef SRMI_ABL_NO_LDS  // timing ablation (wrong results): fragments are read once and reused
        if (dx != 0 || Ab != smem) return;
#endif
#pragma unroll
        for (int rho = 0; rho < MB + KT - 1; ++rho) {
            if (rho < KT) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
                    f.wf[rho][b] = *(const u32x4*)(Bb + ((rho * KT + dx) * 32 * NB + b * 32) * 32 + ((lh ^ bm) << 4));
            }
            const int pp = p0 + rho * PW + dx;
#ifdef SRMI_ABL_DX_COPY  // timing ablation (wrong results): only the first horizontal tap reads pixel fragments
            if (dx != 0) continue;
#endif
            f.pf[rho] = *(const u32x4*)(Ab + pp * 32 + (((lh ^ (pp >> 3)) & 1) << 4));
        }
    };
    // part 0: the MFMAs of the first patch row (they only need wf[0] and pf[0]); part 1: the rest.
    // `after(i)` runs after the i-th multiply step (one 16-byte chunk of channels) of the stage: the main loop issues the next stage's DMA
    // pieces there, one every few MFMAs, so that their issue cost hides under the matrix pipe.
    int mfma_i = 0;
    auto mma_frag = [&](const Frag& f, const int part, auto&& after) {
#pragma unroll
        for (int rho = 0; rho < MB + KT - 1; ++rho) {
            if ((rho == 0) != (part == 0)) continue;
#pragma unroll
            for (int dy = 0; dy < KT; ++dy) {
                const int r = rho - dy;
                if (r >= 0 && r < MB) {
#pragma unroll
                    for (int b = 0; b < NB; ++b) {
                        mma_chunk<DT>(acc[r][b], f.wf[dy][b], f.pf[rho]);
                        after(mfma_i++);
                    }
                }
            }
        }
    };
#ifdef SRMI_GAP_OVERRIDE  // diagnostic build: issue the next stage's pieces closer together, earlier in the stage
    constexpr int GAP = SRMI_GAP_OVERRIDE;
#else
    constexpr int GAP = (KT * KT * MB * NB) / PPW;  // multiply steps between two DMA pieces
#endif
    static_assert(GAP >= 1, "a stage needs at least as many multiply steps as DMA pieces per wave");

    // Two main loops; long reductions take the first (measured: +3 % on EDSR's 16-stage layers, +1 % on
    // the 8-12-stage layers of a dense block, but -4 % on 4-stage layers, which keep the second).
    static_assert(!CARRIED || (G::NS == 2 && !GATE), "the fragment-carried loop: two-slot ring, plain input");
    if constexpr (CARRIED) {
    // Fragment-carried pipeline (two-slot ring).  The barrier of a stage sits after the LAST LDS read
    // of the stage has returned, not after its last MFMA: the final third of the stage's MFMAs only
    // needs registers, so it runs behind the barrier, together with the first fragment reads of the
    // next stage and the DMA issue of the stage after that.  No wave ever waits for LDS with an
    // empty matrix pipe at a stage boundary.
    // both slots are free at the start: stage 1 is requested right behind stage 0, and only stage 0
    // has to have landed before the first fragment reads
    if (nstage > 1) {
        issue(1, 1);
        wait_vmcnt_le(my_pieces);
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    Frag fr0, fr1;
    load_frag(fr0, smem, smem + G::A_BYTES + lx * 32, 0);
    auto none = [](const int) {};
    auto run_stage = [&](Frag& fa, Frag& fb, const int s, const int cur) {
        const char* Ab = smem + cur * G::STAGE_BYTES;
        const char* Bb = Ab + G::A_BYTES + lx * 32;
        load_frag(fb, Ab, Bb, 1);
#ifdef SRMI_ABL_DX_COPY
#pragma unroll
        for (int rho = 0; rho < MB + 2; ++rho) fb.pf[rho] = fa.pf[rho];
#endif
        __builtin_amdgcn_sched_barrier(0);
        mma_frag(fa, 0, none);
        mma_frag(fa, 1, none);
        __builtin_amdgcn_sched_barrier(0);
        load_frag(fa, Ab, Bb, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag(fb, 0, none);
        mma_frag(fb, 1, none);
        __builtin_amdgcn_sched_barrier(0);
        // every LDS read of stage s has been issued: once they and this wave's share of stage s+1
        // have landed, the barrier tells everybody that slot `cur` may be overwritten and that the
        // other slot is complete
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const bool more = s + 2 < nstage;
        {
            // unconditional (after the last stage the values are simply unused): a branch here makes
            // the compiler's waitcnt bookkeeping pessimistic and the MFMAs below would wait for these reads
            const char* An = smem + (cur ^ 1) * G::STAGE_BYTES;
            load_frag(fb, An, An + G::A_BYTES + lx * 32, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        const int sn = more ? s + 2 : s;
        const auto ra = pixel_rsrc(sn);
        const auto rb = weight_rsrc(sn);
        char* nbuf = smem + cur * G::STAGE_BYTES;
        mfma_i = 0;
        auto after = [&](const int i) {
            if (i < PPW) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue_piece(ra, rb, nbuf, i);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        static_assert(3 * MB * NB >= PPW, "the last third of a stage must have a slot for every DMA piece");
        mma_frag(fa, 0, after);
        mma_frag(fa, 1, after);
        __builtin_amdgcn_sched_barrier(0);
    };
    for (int s = 0; s < nstage; s += 2) {
        run_stage(fr0, fr1, s, 0);
        if (s + 1 < nstage) run_stage(fr1, fr0, s + 1, 1);
    }
    } else {
    int cur = 0;  // ring slot of stage s
    for (int s = 0; s < nstage; ++s) {
        // stage s has landed once at most the younger stage's pieces are still outstanding
        if constexpr (GATE) {
            // the weight pieces of stage s have landed once only the younger operations are outstanding; this
            // wave's writes of the stage's pixels must be in LDS before the barrier
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kGateYounger) : "memory");
        } else if (G::NS == 3 && s + 1 < nstage) wait_vmcnt_le(my_pieces);
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        // one barrier per stage: everybody's share of stage s is in LDS, and everybody is done
        // reading the slot that the next issue overwrites
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        const int ahead = G::NS - 1;
        int slot = cur + ahead;
        if (slot >= G::NS) slot -= G::NS;
        const bool more = s + ahead < nstage;
        const int sn = more ? s + ahead : s;
        const auto ra = pixel_rsrc(sn);
        const auto rb = weight_rsrc(sn);
        // gated input: the steps below build stage s+1 and request stage s+2
        const bool more2 = s + 2 < nstage;
        const int sn2 = more2 ? s + 2 : s;
        const auto rx = sum_rsrc(GATE ? sn : 0);
        const auto ra2 = pixel_rsrc(GATE ? sn2 : 0);
        const auto rt2 = gate_rsrc(GATE ? sn2 : 0);
        char* nbuf = smem + slot * G::STAGE_BYTES;
#ifdef SRMI_BULK_ISSUE  // diagnostic build: the whole next stage is issued before the multiply steps
        if (more) issue(sn, slot);
#endif
        auto after = [&](const int i) {
            if constexpr (GATE) {
                // weight pieces first (the counted wait above relies on everything after them being younger),
                // then the steps, spread over the remaining MFMAs
                constexpr int NMF = 9 * MB * NB;
                constexpr int EPH = (kGateSteps + (NMF - BPW) - 1) / (NMF - BPW);  // steps per MFMA
                static_assert(NMF > BPW, "a stage needs more multiply steps than weight pieces per wave");
                __builtin_amdgcn_sched_barrier(0);
                if (i < BPW) {
                    // a wave without a piece in this round repeats its previous one (same bytes to the same
                    // place); after the last stage the other slot receives the current weights again, unused
                    const int kq = wave + kNW * i;
                    const int kk = kq < G::B_PIECES ? kq : kq - kNW;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(nbuf + G::A_BYTES + kk * 1024),
                                                             16, kk * 1024 + lane * 16, 0, 0, 0);
                } else {
#pragma unroll
                    for (int e = 0; e < EPH; ++e) {
                        const int st = (i - BPW) * EPH + e;
                        if (st < kGateSteps) gate_step(st, sn, nbuf, rx, more, more2, ra2, rt2);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                return;
            }
#ifndef SRMI_BULK_ISSUE
            if (i % GAP == GAP - 1 && i / GAP < PPW) {
                __builtin_amdgcn_sched_barrier(0);
                if (more) issue_piece(ra, rb, nbuf, i / GAP);
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
        };

        const char* Ab = smem + cur * G::STAGE_BYTES;
        const char* Bb = Ab + G::A_BYTES + lx * 32;
        // Per step: the first MFMAs of the current set make the compiler wait for that set alone
        // (nothing else is outstanding yet); then the next set's ds_reads are issued and run
        // under the remaining MFMAs, so the following step's wait finds them complete.
        mfma_i = 0;
        if constexpr (KT == 2) {
            // two horizontal taps: both fragment sets are requested up front, the second lands under the first's MFMAs
            Frag f0, f1;
            load_frag(f0, Ab, Bb, 0);
            __builtin_amdgcn_sched_barrier(0);
            mma_frag(f0, 0, after);
            __builtin_amdgcn_sched_barrier(0);
            load_frag(f1, Ab, Bb, 1);
            __builtin_amdgcn_sched_barrier(0);
            mma_frag(f0, 1, after);
            __builtin_amdgcn_sched_barrier(0);
            mma_frag(f1, 0, after);
            mma_frag(f1, 1, after);
            __builtin_amdgcn_sched_barrier(0);
        } else {
#ifdef SRMI_SINGLE_FRAG  // diagnostic build: one fragment set, reloaded per horizontal tap (fewer registers)
        Frag f0;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            load_frag(f0, Ab, Bb, dx);
            __builtin_amdgcn_sched_barrier(0);
            mma_frag(f0, 0, after);
            mma_frag(f0, 1, after);
            __builtin_amdgcn_sched_barrier(0);
        }
#else
        Frag f0, f1;
        load_frag(f0, Ab, Bb, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag(f0, 0, after);
        __builtin_amdgcn_sched_barrier(0);
        load_frag(f1, Ab, Bb, 1);
#ifdef SRMI_ABL_DX_COPY
#pragma unroll
        for (int rho = 0; rho < MB + 2; ++rho) f1.pf[rho] = f0.pf[rho];
#endif
        __builtin_amdgcn_sched_barrier(0);
        mma_frag(f0, 1, after);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag(f1, 0, after);
        __builtin_amdgcn_sched_barrier(0);
        load_frag(f0, Ab, Bb, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag(f1, 1, after);
        __builtin_amdgcn_sched_barrier(0);
        mma_frag(f0, 0, after);
        mma_frag(f0, 1, after);
        __builtin_amdgcn_sched_barrier(0);
#endif
        }
        STAMP();
        cur = cur + 1 == G::NS ? 0 : cur + 1;
    }
    }  // classic loop
    // the epilogue reuses the stage buffers (bias, channel sums): all waves must be out of the main loop
    // first (and the unused requests of a gated input's last stages must have landed)
    if constexpr (GATE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef SRMI_STAMPS_EPI
    STAMP();
#endif
    if (tid < 32 * NB) {
        ((float*)(smem + G::BIAS_OFFSET))[tid] = bias_v;
        if constexpr (DT == SRMI_I8) ((float*)(smem + G::BIAS_OFFSET))[32 * NB + tid] = scale_v;
    }
    __syncthreads();
#ifdef SRMI_STAMPS_EPI
    STAMP();
#endif

    // ---- epilogue --------------------------------------------------------------------------------
    // One uniform switch selects straight-line code specialised for the tensor types in use; the
    // combinations the networks need are compiled, anything else takes the generic path.
    const EpiCtx ctx{n, ct, tile, x0, y0, wave, lx, lh, tid};
    const float* lds_bias = (const float*)(smem + G::BIAS_OFFSET);
    float* lds_red = (float*)smem;
    const int od = p.out.ptr ? p.out.dtype : -1, rd = p.res1.ptr ? p.res1.dtype : -1;
    const int r2 = p.res2.ptr ? p.res2.dtype : -1, o2 = p.out2.ptr ? p.out2.dtype : -1;
    const bool plain_io = p.tail_mode == SRMI_TAIL_NONE && od == DT && (r2 == -1 || r2 == rd);
    if constexpr (DT == SRMI_I8) {
        // int8 dense blocks: grown channels go out as int8; the block's last conv adds the half-precision trunk
        // and writes the new trunk twice, quantised (next conv input) and in half precision (next residual)
        if (plain_io && rd == -1 && o2 == -1) {
            epilogue<DT, MB, NB, DT, -1, false, -1, 0, false>(p, ctx, acc, lds_bias, lds_red);
        } else if (plain_io && rd == SRMI_F16 && r2 == -1 && o2 == SRMI_F16) {
            epilogue<DT, MB, NB, DT, SRMI_F16, false, SRMI_F16, 0, false>(p, ctx, acc, lds_bias, lds_red);
        } else if (plain_io && rd == SRMI_F16 && r2 == SRMI_F16 && o2 == SRMI_F16) {
            epilogue<DT, MB, NB, DT, SRMI_F16, true, SRMI_F16, 0, false>(p, ctx, acc, lds_bias, lds_red);
        } else {
            epilogue<DT, MB, NB, kAnyType, kAnyType, true, kAnyType>(p, ctx, acc, lds_bias, lds_red);
        }
    } else {
    // (per-tile channel sums go with the plain arm or the generic one: the arms with residuals carry no sum registers)
    const bool sums = p.chansum != nullptr || p.ca_sums != nullptr;
    if (plain_io && rd == -1 && o2 == -1) {
        if (sums) epilogue<DT, MB, NB, DT, -1, false, -1, 0, true>(p, ctx, acc, lds_bias, lds_red);
        else epilogue<DT, MB, NB, DT, -1, false, -1, 0, false>(p, ctx, acc, lds_bias, lds_red);
    } else if (!sums && plain_io && rd == SRMI_F32 && r2 == -1 && o2 == -1) {
        epilogue<DT, MB, NB, DT, SRMI_F32, false, -1, 0, false>(p, ctx, acc, lds_bias, lds_red);
    } else if (!sums && plain_io && rd == SRMI_F32 && r2 == -1 && o2 == SRMI_F32) {
        epilogue<DT, MB, NB, DT, SRMI_F32, false, SRMI_F32, 0, false>(p, ctx, acc, lds_bias, lds_red);
    } else if (!sums && plain_io && rd == SRMI_F32 && r2 == SRMI_F32 && o2 == SRMI_F32) {
        epilogue<DT, MB, NB, DT, SRMI_F32, true, SRMI_F32, 0, false>(p, ctx, acc, lds_bias, lds_red);
    } else if (!sums && plain_io && rd == DT && r2 == -1 && o2 == -1) {
        epilogue<DT, MB, NB, DT, DT, false, -1, 0, false>(p, ctx, acc, lds_bias, lds_red);
    } else if (!sums && plain_io && rd == DT && r2 == -1 && o2 == DT) {
        epilogue<DT, MB, NB, DT, DT, false, DT, 0, false>(p, ctx, acc, lds_bias, lds_red);
    } else if (!sums && plain_io && rd == DT && r2 == DT && o2 == DT) {
        epilogue<DT, MB, NB, DT, DT, true, DT, 0, false>(p, ctx, acc, lds_bias, lds_red);
    } else {
        epilogue<DT, MB, NB, kAnyType, kAnyType, true, kAnyType>(p, ctx, acc, lds_bias, lds_red);
    }
    }
    STAMP();
#ifdef SRMI_CLOCK
    if (clock_buf && lane == 0 && p.tail_mode == SRMI_TAIL_NONE) {
        clock_buf[clock_base + 0] = clk_c0;
        clock_buf[clock_base + 1] = clk_r0;
        clock_buf[clock_base + 2] = __builtin_amdgcn_s_memtime();
        clock_buf[clock_base + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#endif
}

// ---- host launcher --------------------------------------------------------------------------------
// The dynamic-LDS limit is a per-device attribute of the function: set once on every device used.  Setting
// it twice is harmless, so concurrent first calls only need the bookkeeping itself to be atomic.
// (a plain input on the two-slot ring has both loop forms: two kernels)
template <int MB, int NB, int GATE, int KT = 3>
constexpr bool kHasCarried = !GATE && KT == 3 && ConvGeom<MB, NB, KT>::NS == 2;
constexpr int kCarriedMinStages = 8;   // stages from which the fragment-carried loop is the faster one

template <int DT, int MB, int NB, int GATE, int KT = 3>
static int configure() {
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    SRMI_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !(configured.load(std::memory_order_acquire) >> dev & 1ull)) {
        constexpr int lds = (int)ConvGeom<MB, NB, KT>::LDS_TOTAL + (GATE ? ConvGeom<MB, NB, KT>::GATE_BYTES : 0);
        SRMI_HIP(hipFuncSetAttribute((const void*)conv3x3_kernel<DT, MB, NB, GATE, false, KT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if constexpr (kHasCarried<MB, NB, GATE, KT>)
            SRMI_HIP(hipFuncSetAttribute((const void*)conv3x3_kernel<DT, MB, NB, GATE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        if (tracked) configured.fetch_or(1ull << dev, std::memory_order_release);
    }
    return 0;
}

template <int DT, int MB, int NB, int GATE = 0, int KT = 3>
static int launch(const ConvParams& p, hipStream_t stream) {
    constexpr size_t lds = ConvGeom<MB, NB, KT>::LDS_TOTAL + (GATE ? ConvGeom<MB, NB, KT>::GATE_BYTES : 0);
    if (configure<DT, MB, NB, GATE, KT>()) return -1;
    dim3 grid(p.tiles_x * p.tiles_y, p.cout / (32 * NB), p.n);
    if constexpr (kHasCarried<MB, NB, GATE, KT>) {
        if (p.ncb >= kCarriedMinStages) {
            hipLaunchKernelGGL((conv3x3_kernel<DT, MB, NB, GATE, true>), grid, dim3(64 * kNW), lds, stream, p);
            SRMI_HIP(hipGetLastError());
            return 0;
        }
    }
    hipLaunchKernelGGL((conv3x3_kernel<DT, MB, NB, GATE, false, KT>), grid, dim3(64 * kNW), lds, stream, p);
    SRMI_HIP(hipGetLastError());
    return 0;
}

// the 2x2 (sub-pixel) form: 8- and 12-row tiles, 32 or 64 outputs per workgroup
template <int DT>
static int launch_up2_dt(const ConvParams& p, int mb, int nb, hipStream_t s) {
    if (mb != 2 && mb != 3) SRMI_FAIL("srmi_conv3x3: the sub-pixel form needs rows_per_wave 2 or 3");
    if (nb == 1) return mb == 2 ? launch<DT, 2, 1, 0, 2>(p, s) : launch<DT, 3, 1, 0, 2>(p, s);
    return mb == 2 ? launch<DT, 2, 2, 0, 2>(p, s) : launch<DT, 3, 2, 0, 2>(p, s);
}

template <int DT>
static int configure_up2_dt() {
    return configure<DT, 2, 1, 0, 2>() || configure<DT, 3, 1, 0, 2>() || configure<DT, 2, 2, 0, 2>() || configure<DT, 3, 2, 0, 2>();
}

template <int DT>
static int launch_dt(const ConvParams& p, int mb, int nb, hipStream_t s) {
    if constexpr (DT == SRMI_I8) {  // dense-block shapes only: 32-wide output tiles, 8- or 12-row tiles
        if (nb != 1 || mb >= 4) SRMI_FAIL("srmi_conv3x3: int8 needs cout_tile 32 and rows_per_wave 2 or 3");
        return mb == 2 ? launch<DT, 2, 1>(p, s) : launch<DT, 3, 1>(p, s);
    }
    if constexpr (DT == SRMI_F16 || DT == SRMI_BF16) {
        if (p.in_gate.ptr) {  // validated by the caller: rows_per_wave 2 or 3 (2 with an f32 `in`)
            if (p.in.dtype == SRMI_F32) return nb == 1 ? launch<DT, 2, 1, 2>(p, s) : launch<DT, 2, 2, 2>(p, s);
            if (nb == 1) return mb == 2 ? launch<DT, 2, 1, 1>(p, s) : launch<DT, 3, 1, 1>(p, s);
            return mb == 2 ? launch<DT, 2, 2, 1>(p, s) : launch<DT, 3, 2, 1>(p, s);
        }
    }
    if (nb == 1) {
        if (mb == 2) return launch<DT, 2, 1>(p, s);
        if (mb == 3) return launch<DT, 3, 1>(p, s);
        return launch<DT, 4, 1>(p, s);
    }
    // 4 rows x 64 channels per wave does not fit the 256-register budget of two waves per SIMD
    if (mb == 2) return launch<DT, 2, 2>(p, s);
    return launch<DT, 3, 2>(p, s);
}

template <int DT>
static int configure_dt() {
    if constexpr (DT == SRMI_I8) return configure<DT, 2, 1, 0>() || configure<DT, 3, 1, 0>();
    if constexpr (DT == SRMI_F16 || DT == SRMI_BF16) {
        if (configure<DT, 2, 1, 1>() || configure<DT, 3, 1, 1>() || configure<DT, 2, 2, 1>() || configure<DT, 3, 2, 1>() ||
            configure<DT, 2, 1, 2>() || configure<DT, 2, 2, 2>())
            return -1;
    }
    return configure<DT, 2, 1, 0>() || configure<DT, 3, 1, 0>() || configure<DT, 4, 1, 0>() ||
           configure<DT, 2, 2, 0>() || configure<DT, 3, 2, 0>();
}

}  // namespace srmi
