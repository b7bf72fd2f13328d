// One residual channel-attention block body in ONE kernel (RCAB.forward, rcan_model.py:79-97, without its last
// elementwise line): y = conv2(relu(conv1(x'))) with the intermediate tile kept in LDS, so the 64-channel tensor between
// the two convolutions is neither written to nor read from memory (two of a block's six tensor passes).
//
//   x' = x + gate_alpha * (y_prev * gate[n][c])   formed while loading (the previous block's attention output, exactly
//        the arithmetic of srmi_conv3x3's gated input), interior pixels stored to in_sum on the way;
//   t  = relu(conv1(x') + b1)                      on the output tile plus a one-pixel ring (recomputed per tile), f16 in LDS,
//        zero outside the image (conv2's zero padding applies to t, not to x');
//   y  = conv2(t) + b2                             -> out, with the channel-attention tail of srmi_conv3x3 (pooling by
//        fixed-point atomics, squeeze / excite / sigmoid in the image's last workgroup).
//
// Work decomposition: workgroup = 4 waves = 8 rows x 32 columns x 64 output channels of one image.
//   conv1 runs on the 10 x 34 = 340 pixels of the ring-extended tile as ELEVEN blocks of 32 consecutive pixels in
//   row-major order of that region (an MFMA column is a pixel with its own LDS address, so a block may wrap around a row
//   end): 352 computed for 256 useful outputs, +37.5 % matrix work on conv1, +19 % on the block.  Wave w takes blocks
//   w, w + 4, w + 8 (the twelfth does not exist: computed on clamped addresses and dropped).
//   conv2 runs like the direct kernel with 2 rows per wave and both 32-channel halves, reading its pixel fragments from
//   the intermediate in LDS; only its weights are staged.
// LDS (78.5 KiB + 1.5 KiB = exactly half a CU's 160 KiB: two workgroups per CU):
//   region 1, 42.5 KiB: during conv1 the two-slot ring of conv1's weight stages (18 KiB each); then the intermediate
//             [channel plane 4][channel half 2][pixel 340][16 bytes];
//   region 2, 36 KiB:   during conv1 the two-slot ring of x' stages, [channel half 2][pixel 12 x 36][16 bytes], written by
//             the waves themselves (the gated input passes through registers); during conv2 the two-slot ring of conv2's
//             weight stages.
//   Pixel data are stored as planes of 16-byte chunks: neighbouring lanes read neighbouring chunks (conflict-free
//   ds_read_b128 without a swizzle) and every tap is a constant offset from a lane's base address.
// The order of the multiply-accumulates per output equals the direct kernel's (stage, horizontal tap, vertical tap), so
// conv1 and conv2 produce the bits the unfused pair produces; only the pooled sums see different tile shapes.
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <type_traits>

#include "conv_epilogue.h"

namespace srmi {

struct RcabGeom {
    static constexpr int TH = 8, TW = kTileW;
    static constexpr int XH = TH + 4, XW = TW + 4, NPX = XH * XW;            // x' patch 12 x 36 = 432 pixels
    static constexpr int XC_PIECES = (NPX + 63) / 64;                        // 7 pieces (1 KiB) per channel half
    static constexpr int XC_BYTES = XC_PIECES * 1024;
    static constexpr int X_PIECES = 2 * XC_PIECES, X_BYTES = X_PIECES * 1024;  // 14 KiB per stage
    static constexpr int T_H = TH + 2, T_W = TW + 2, NPT = T_H * T_W;        // intermediate 10 x 34 = 340 pixels
    static constexpr int TC_BYTES = NPT * 16, TP_BYTES = 2 * TC_BYTES;       // per channel half / per 16-channel plane
    static constexpr int NBLK = (NPT + 31) / 32;                             // 11 blocks of 32 pixels
    static constexpr int BLK_PER_WAVE = (NBLK + kNW - 1) / kNW;              // 3
    static constexpr int W_BYTES = 9 * 64 * 32, W_PIECES = W_BYTES / 1024;   // 18 KiB per weight stage
    static constexpr int APW = (X_PIECES + kNW - 1) / kNW;                   // 4 pixel pieces per wave (some waves own 3)
    static constexpr int BPW = (W_PIECES + kNW - 1) / kNW;                   // 5 weight pieces per wave (some own 4)
    static constexpr int MAX_PLANES = 4;                                     // 64 channels
    static constexpr int R1 = 0, R1_BYTES = MAX_PLANES * TP_BYTES;           // 43520
    static constexpr int R2 = R1 + R1_BYTES, R2_BYTES = 2 * W_BYTES;         // 36864
    static constexpr int GATE_OFF = R2 + R2_BYTES;                           // float[64]
    static constexpr int SPARE_OFF = GATE_OFF + 256;                         // 1 KiB absorbing writes of pieces a wave does not own
    static constexpr int BIAS1_OFF = SPARE_OFF + 1024;                       // float[64]: bias of the first convolution
    static constexpr int LDS_BYTES = BIAS1_OFF + 256;                        // 81920 = half a CU's LDS
};
static_assert(2 * RcabGeom::W_BYTES <= RcabGeom::R1_BYTES && 2 * RcabGeom::X_BYTES <= RcabGeom::R2_BYTES, "rings fit their regions");
static_assert(2 * RcabGeom::LDS_BYTES <= 160 * 1024, "two workgroups per CU");
static_assert(kNW == 4, "four waves per workgroup");

// XF32: `in` and `in_sum` are f32 tensors (the residual trunk kept in single precision): a 16-byte chunk of a stage is
// then the whole 32-byte pixel of one f32 plane (plane 2 s + chunk), fetched and stored as two 16-byte halves; the
// multiplies see its half-precision cast, in_sum the f32 value.
template <bool XF32>
__global__ __launch_bounds__(64 * kNW, 2) void rcab_fused_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = RcabGeom;
    constexpr int DT = SRMI_F16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lx = lane & 31, lh = lane >> 5;
    int tile, n;
    {
        const unsigned gx = gridDim.x;
        const unsigned total = gx * gridDim.z;
        const unsigned lin = blockIdx.x + gx * blockIdx.z;
        const unsigned xcd = lin & 7u, j = lin >> 3;
        const unsigned q = total >> 3, r = total & 7u;
        const unsigned id = xcd * q + (xcd < r ? xcd : r) + j;
        tile = (int)(id % gx);
        n = (int)(id / gx);
    }
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int x0 = tx * G::TW, y0 = ty * G::TH;
    const int nstage = p.ncb;   // 16-channel planes of x' (= planes of the intermediate)
#ifdef SRMI_STAMPS
    // diagnostic build: lane 0 of every wave records s_memtime at each phase boundary into [workgroup][wave][40];
    // slot 39 holds (XCC_ID << 32) | HW_ID
    unsigned long long* stamp_buf = (unsigned long long*)p.tail_out;
    int stamp_i = 0;
    const size_t stamp_base = (((size_t)blockIdx.z * gridDim.x + blockIdx.x) * kNW + wave) * 40;
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
    STAMP();

    // ---- gated loader: per-lane source offsets of this wave's x' pieces ---------------------------------
    // LDS chunk i of a stage = channel half i / 448, patch pixel i % 448 (row-major 12 x 36); piece k = chunks 64k..64k+63.
    // (conditions combined with & on values read once: short-circuit forms compile to a branch per term, each with its
    // own reload of the image size from the argument block)
    unsigned aoff[G::APW];
    unsigned keep = 0;   // bit j: piece j of this lane is an in-image pixel of the tile's interior (stored to in_sum)
    const unsigned img_h = (unsigned)p.h, img_w = (unsigned)p.w;
#pragma unroll
    for (int j = 0; j < G::APW; ++j) {
        const int k = wave + kNW * j;
        const int c = k / G::XC_PIECES;
        const int pp = (k - c * G::XC_PIECES) * 64 + lane;
        const int row = pp / G::XW, col = pp - row * G::XW;
        const int gy = y0 - 2 + row, gx = x0 - 2 + col;
        const bool ok = (k < G::X_PIECES) & (pp < G::NPX) & ((unsigned)gy < img_h) & ((unsigned)gx < img_w);
        aoff[j] = ok ? ((unsigned)gy * img_w + (unsigned)gx) * 32u + (unsigned)c * 16u : SRMI_OOB_OFFSET;
        const bool interior = ((unsigned)(row - 2) < (unsigned)G::TH) & ((unsigned)(col - 2) < (unsigned)G::TW);
        keep |= (ok & interior) ? 1u << j : 0u;
    }
    const char* in_img = p.in.ptr + (long long)n * p.in.img_stride_b;
    const char* gate_img = p.in_gate.ptr + (long long)n * p.in_gate.img_stride_b;
    char* sum_img = p.in_sum.ptr ? p.in_sum.ptr + (long long)n * p.in_sum.img_stride_b : nullptr;
    const unsigned plane_bytes = p.in_plane_bytes;
    auto pixel_rsrc = [&](int s) {
        if constexpr (XF32)  // a window over the two f32 planes that hold the stage's 16 channels
            return make_rsrc(in_img + (long long)(2 * s) * p.in.plane_stride_b, (unsigned)p.in.plane_stride_b + plane_bytes);
        else
            return make_rsrc(in_img + (long long)s * p.in.plane_stride_b, plane_bytes);
    };
    auto gate_rsrc = [&](int s) { return make_rsrc(gate_img + (long long)s * p.in_gate.plane_stride_b, plane_bytes); };
    auto sum_rsrc = [&](int s) {
        if constexpr (XF32)
            return make_rsrc(sum_img + (long long)(2 * s) * p.in_sum.plane_stride_b,
                             sum_img ? (unsigned)p.in_sum.plane_stride_b + plane_bytes : 0u);
        else
            return make_rsrc(sum_img + (long long)s * p.in_sum.plane_stride_b, sum_img ? plane_bytes : 0u);
    };
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
    auto w1_rsrc = [&](int s) { return make_rsrc(p.wpk + (size_t)s * G::W_BYTES, G::W_BYTES); };
    auto w2_rsrc = [&](int s) { return make_rsrc(p.wpk2 + (size_t)s * G::W_BYTES, G::W_BYTES); };
    // weight piece i of this wave into `slot`; a wave without a piece in its last round repeats its previous one
    auto issue_w = [&](const __amdgpu_buffer_rsrc_t rb, char* slot, const int i) {
        const int kq = wave + kNW * i;
        const int kk = kq < G::W_PIECES ? kq : kq - kNW;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(slot + kk * 1024), 16,
                                                 kk * 1024 + lane * 16, 0, 0, 0);
    };

    u32x4 gx[G::APW], gt[G::APW], gx1[XF32 ? G::APW : 1];
    auto gate_fetch = [&](const int q, const bool really, const __amdgpu_buffer_rsrc_t ra, const __amdgpu_buffer_rsrc_t rt) {
        const unsigned off = really ? aoff[q] : SRMI_OOB_OFFSET;
        if constexpr (XF32) {
            const unsigned xo = really ? x_offset(q) : kOobF32;
            gx[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, xo, 0, 0));
            gx1[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, xo + 16u, 0, 0));
        } else {
            gx[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, off, 0, 0));
        }
        gt[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rt, off, 0, 0));
    };
    // One STEP combines one channel of one piece (8 steps per piece); the last step of a piece writes its 16 bytes to
    // the stage slot (and to in_sum) and requests the same piece two stages on.  No branch inside (the counted
    // vmcnt waits need every vector-memory operation between a request and its use to be countable): what must not
    // happen is masked with out-of-range offsets and the spare KiB.
    constexpr int kGateSteps = 8 * G::APW;
    const float gate_alpha = p.gate_alpha;
    const float* gate_lds = (const float*)(smem + G::GATE_OFF);
    f32x4 gg0 = {0.f, 0.f, 0.f, 0.f}, gg1 = {0.f, 0.f, 0.f, 0.f};
    u32x4 gpk = {0u, 0u, 0u, 0u};
    float gprev = 0.f;
    auto gate_step = [&](const int step, const int s, char* xbuf, const __amdgpu_buffer_rsrc_t rx, const bool live,
                         const bool fetch_next, const __amdgpu_buffer_rsrc_t ra_next, const __amdgpu_buffer_rsrc_t rt_next) {
        const int q = step >> 3, j = step & 7;
        const int k = wave + kNW * q;
        const bool mine = k < G::X_PIECES;
        if (j == 0) {
            const int c = (k < G::XC_PIECES) ? 0 : 1;      // channel half of the whole piece
            const float* g = gate_lds + s * 16 + c * 8;
            gg0 = *(const f32x4*)g;
            gg1 = *(const f32x4*)(g + 4);
        }
        const float tv = (float)__builtin_bit_cast(f16x8, gt[q])[j];
        float xv;
        if constexpr (XF32) xv = __uint_as_float(j < 4 ? gx[q][j & 3] : gx1[q][j & 3]);
        else xv = (float)__builtin_bit_cast(f16x8, gx[q])[j];
        const float o = (tv * (j < 4 ? gg0[j & 3] : gg1[j & 3])) * gate_alpha + xv;
        if constexpr (XF32) {  // the single-precision sum takes the place of x until it is stored
            if (j < 4) gx[q][j & 3] = __float_as_uint(o);
            else gx1[q][j & 3] = __float_as_uint(o);
        }
        if (j & 1) {
            typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
            const f16x2 h = {to_f16_sat(gprev), to_f16_sat(o)};
            gpk[j >> 1] = __builtin_bit_cast(unsigned, h);
        } else {
            gprev = o;
        }
        if (j == 7) {
            char* dst = mine ? xbuf + k * 1024 : smem + G::SPARE_OFF;
            *(u32x4*)(dst + lane * 16) = gpk;
            const bool store = live && ((keep >> q) & 1u);
            if constexpr (XF32) {
                const unsigned so = store ? s_offset(q) : kOobF32;
                __builtin_amdgcn_raw_buffer_store_b128(gx[q], rx, so, 0, 0);
                __builtin_amdgcn_raw_buffer_store_b128(gx1[q], rx, so + 16u, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(gpk, rx, store ? aoff[q] : SRMI_OOB_OFFSET, 0, 0);
            }
            gate_fetch(q, fetch_next, ra_next, rt_next);
        }
    };
    constexpr int kGateYounger = (XF32 ? 5 : 3) * G::APW;   // per pixel piece: the in_sum store(s) and the loads of the next fetch

    // ---- conv1: per-lane LDS addresses of its three pixel blocks -------------------------------------------
    int xb[G::BLK_PER_WAVE];      // byte offset of patch pixel (r1, c1) = tap (0, 0) of the lane's pixel, within a stage slot
    int tpix[G::BLK_PER_WAVE];    // index of the lane's pixel in the intermediate, or -1
    bool tin[G::BLK_PER_WAVE];    // that pixel lies inside the image
#pragma unroll
    for (int b = 0; b < G::BLK_PER_WAVE; ++b) {
        const int pp1 = 32 * (wave + kNW * b) + lx;
        const int ppc = pp1 < G::NPT ? pp1 : G::NPT - 1;
        const int r1 = ppc / G::T_W, c1 = ppc - r1 * G::T_W;
        xb[b] = lh * G::XC_BYTES + (r1 * G::XW + c1) * 16;
        tpix[b] = pp1 < G::NPT ? pp1 : -1;
        const int gy = y0 - 1 + r1, gx_ = x0 - 1 + c1;
        tin[b] = ((unsigned)gy < img_h) & ((unsigned)gx_ < img_w);
    }
    const int bm = (lx >> 3) & 1;   // weight-row swizzle term of the packed weight image

    f32x16 acc1[G::BLK_PER_WAVE][2];
#pragma unroll
    for (int b = 0; b < G::BLK_PER_WAVE; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc1[b][h][i] = 0.f;

    // ---- prologue -------------------------------------------------------------------------------------------
    char* const w1slot0 = smem + G::R1;
    char* const xslot0 = smem + G::R2;
    // the gate vector is requested first and waited for alone: the weight pieces and the x' pieces behind it stay in flight
    const float gate_v = p.gate[(long long)n * nstage * 16 + (tid < nstage * 16 ? tid : 0)];
    const float bias1_v = p.bias[tid & 63];
    const float bias2_v = p.bias2[tid & 63];
    {
        const auto ra = pixel_rsrc(0);
        const auto rt = gate_rsrc(0);
        const auto rb = w1_rsrc(0);
#pragma unroll
        for (int i = 0; i < G::BPW; ++i) issue_w(rb, w1slot0, i);
#pragma unroll
        for (int q = 0; q < G::APW; ++q) gate_fetch(q, true, ra, rt);
    }
    STAMP();
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::BPW + (XF32 ? 3 : 2) * G::APW) : "memory");
    if (tid < nstage * 16) ((float*)(smem + G::GATE_OFF))[tid] = gate_v;
    if (tid < 64) ((float*)(smem + G::BIAS1_OFF))[tid] = bias1_v;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the gate vector is in LDS
    __builtin_amdgcn_sched_barrier(0);
    STAMP();
    {
        const int s1 = nstage > 1 ? 1 : 0;
        const auto rx = sum_rsrc(0);
        const auto ra = pixel_rsrc(s1);
        const auto rt = gate_rsrc(s1);
#pragma unroll
        for (int st = 0; st < kGateSteps; ++st) gate_step(st, 0, xslot0, rx, true, nstage > 1, ra, rt);
    }
    STAMP();

    // ---- conv1 main loop ------------------------------------------------------------------------------------
    struct FragA {
        u32x4 wf[2];
        u32x4 pf[G::BLK_PER_WAVE];
    };
    // LAST: the final stage prepares nothing (no weight pieces, no x' steps)
    auto conv1_stage = [&](const int s, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;
        // the weight pieces of stage s have landed once only the younger operations are outstanding; this wave's
        // writes of the stage's pixels must be in LDS before the barrier
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kGateYounger) : "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        const int cur = s & 1;
        const bool more = s + 1 < nstage, more2 = s + 2 < nstage;
        const int sn = more ? s + 1 : s, sn2 = more2 ? s + 2 : s;
        const auto rb = w1_rsrc(sn);
        const auto rx = sum_rsrc(sn);
        const auto ra2 = pixel_rsrc(sn2);
        const auto rt2 = gate_rsrc(sn2);
        char* nw = smem + G::R1 + (cur ^ 1) * G::W_BYTES;
        char* nx = smem + G::R2 + (cur ^ 1) * G::X_BYTES;
        const char* Wb = smem + G::R1 + cur * G::W_BYTES + lx * 32 + ((lh ^ bm) << 4);
        const char* Xb = smem + G::R2 + cur * G::X_BYTES;
        int mfma_i = 0;
        auto after = [&](const int i) {
            // weight pieces first (the counted wait relies on everything after them being younger), then one step of
            // the next stage's x' per multiply
            if constexpr (!LAST) {
                __builtin_amdgcn_sched_barrier(0);
                if (i < G::BPW) issue_w(rb, nw, i);
                else if (i - G::BPW < kGateSteps) gate_step(i - G::BPW, sn, nx, rx, more, more2, ra2, rt2);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        static_assert(9 * G::BLK_PER_WAVE * 2 >= G::BPW + kGateSteps, "a stage has a multiply for every piece and step");
        // taps in the direct kernel's order: horizontal tap outer, vertical tap inner
        auto load = [&](FragA& f, const int t) {
            const int dx = t / 3, dy = t - 3 * dx;
#pragma unroll
            for (int h = 0; h < 2; ++h) f.wf[h] = *(const u32x4*)(Wb + ((dy * 3 + dx) * 64 + h * 32) * 32);
#pragma unroll
            for (int b = 0; b < G::BLK_PER_WAVE; ++b) f.pf[b] = *(const u32x4*)(Xb + xb[b] + (dy * G::XW + dx) * 16);
        };
        auto mul = [&](const FragA& f) {
#pragma unroll
            for (int b = 0; b < G::BLK_PER_WAVE; ++b)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    acc1[b][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.wf[h]),
                                                                         __builtin_bit_cast(f16x8, f.pf[b]), acc1[b][h], 0, 0, 0);
                    after(mfma_i++);
                }
        };
        FragA f0, f1;
        load(f0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < 9; t += 2) {
            if (t + 1 < 9) load(f1, t + 1);
            __builtin_amdgcn_sched_barrier(0);
            mul(f0);
            __builtin_amdgcn_sched_barrier(0);
            if (t + 1 < 9) {
                if (t + 2 < 9) load(f0, t + 2);
                __builtin_amdgcn_sched_barrier(0);
                mul(f1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        STAMP();
    };
    for (int s = 0; s + 1 < nstage; ++s) conv1_stage(s, std::false_type());
    conv1_stage(nstage - 1, std::true_type());
    // Everybody is done with conv1's rings: region 2 takes conv2's first two weight stages, region 1 the intermediate.
    // (Of this wave's operations only in_sum stores and the out-of-range fetches of the stage before last can still be
    // outstanding: they touch no LDS.)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    STAMP();
    if (tid < 64) ((float*)(smem + G::GATE_OFF))[tid] = bias2_v;   // conv1 was the gate vector's last reader
    char* const w2slot0 = smem + G::R2;
    {
        const auto r0 = w2_rsrc(0);
#pragma unroll
        for (int i = 0; i < G::BPW; ++i) issue_w(r0, w2slot0, i);
        const auto r1 = w2_rsrc(nstage > 1 ? 1 : 0);
#pragma unroll
        for (int i = 0; i < G::BPW; ++i) issue_w(r1, w2slot0 + G::W_BYTES, i);
    }
    // ---- conv1 epilogue: t = relu(acc + b1) as f16 into the intermediate, zero outside the image ---------------
    {
        float b1r[2][2][8];
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* bp = (const float*)(smem + G::BIAS1_OFF) + h * 32 + 16 * q + 8 * lh;
                const f32x4 v0 = *(const f32x4*)bp, v1 = *(const f32x4*)(bp + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { b1r[h][q][j] = v0[j]; b1r[h][q][4 + j] = v1[j]; }
            }
        // relu and the saturating conversion in one median: med3(v, 0, 65504) = clamp(max(v, 0)) also for infinities and
        // NaN (the minimum of the other two); pairs of neighbouring channels convert together
        typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int b = 0; b < G::BLK_PER_WAVE; ++b) {
            u32x4 pk[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // half-wave exchange: afterwards the lane owns channels 16 q + 8 lh .. + 7 of the 32-channel half
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc1[b][h][8 * q + j]),
                                                                         __float_as_uint(acc1[b][h][8 * q + 4 + j]), false, false);
                        v[j] = __builtin_amdgcn_fmed3f(__uint_as_float(sw[0]) + b1r[h][q][j], 0.f, 65504.f);
                        v[4 + j] = __builtin_amdgcn_fmed3f(__uint_as_float(sw[1]) + b1r[h][q][4 + j], 0.f, 65504.f);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const f16x2 hh = {(_Float16)v[2 * i], (_Float16)v[2 * i + 1]};
                        pk[h][q][i] = tin[b] ? __builtin_bit_cast(unsigned, hh) : 0u;
                    }
                }
            if (tpix[b] >= 0) {
                char* const dst = smem + G::R1 + lh * G::TC_BYTES + tpix[b] * 16;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int q = 0; q < 2; ++q) *(u32x4*)(dst + (h * 2 + q) * G::TP_BYTES) = pk[h][q];
            }
        }
    }
    STAMP();
    // ---- conv2 ---------------------------------------------------------------------------------------------------
    f32x16 acc2[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc2[r][h][i] = 0.f;
    const int tb = lh * G::TC_BYTES + ((2 * wave) * G::T_W + lx) * 16;   // intermediate pixel (row 2 wave, column lx) = tap (0, 0)
    struct FragB {
        u32x4 wf[3][2];
        u32x4 pf[4];
    };
    for (int s = 0; s < nstage; ++s) {
        // stage s of the weights has landed (the stage behind it may still be in flight at s = 0); this wave's part of
        // the intermediate is written
        if (s == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G::BPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        const int cur = s & 1;
        if (s >= 1 && s + 1 < nstage) {
            const auto rb = w2_rsrc(s + 1);
#pragma unroll
            for (int i = 0; i < G::BPW; ++i) issue_w(rb, w2slot0 + (cur ^ 1) * G::W_BYTES, i);
        }
        const char* Wb = w2slot0 + cur * G::W_BYTES + lx * 32 + ((lh ^ bm) << 4);
        const char* Tb = smem + G::R1 + s * G::TP_BYTES + tb;
        auto load = [&](FragB& f, const int dx) {
#pragma unroll
            for (int rho = 0; rho < 4; ++rho) {
                if (rho < 3) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) f.wf[rho][h] = *(const u32x4*)(Wb + ((rho * 3 + dx) * 64 + h * 32) * 32);
                }
                f.pf[rho] = *(const u32x4*)(Tb + (rho * G::T_W + dx) * 16);
            }
        };
        auto mul = [&](const FragB& f) {
#pragma unroll
            for (int rho = 0; rho < 4; ++rho)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy) {
                    const int r = rho - dy;
                    if (r >= 0 && r < 2) {
#pragma unroll
                        for (int h = 0; h < 2; ++h)
                            acc2[r][h] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, f.wf[dy][h]),
                                                                                 __builtin_bit_cast(f16x8, f.pf[rho]), acc2[r][h], 0, 0, 0);
                    }
                }
        };
        FragB f0, f1;
        load(f0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load(f1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mul(f0);
        __builtin_amdgcn_sched_barrier(0);
        load(f0, 2);
        __builtin_amdgcn_sched_barrier(0);
        mul(f1);
        __builtin_amdgcn_sched_barrier(0);
        mul(f0);
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
    }
    // the epilogue's tile-sum scratch aliases region 2: everybody must be out of conv2 (its bias sits in the gate
    // vector's place since conv1 ended)
    __syncthreads();
    STAMP();
    const EpiCtx ctx{n, 0, tile, x0, y0, wave, lx, lh, tid};
    const float* lds_bias = (const float*)(smem + G::GATE_OFF);
    float* lds_red = (float*)(smem + G::R2);
    if (p.out.dtype == DT && p.ca_sums) epilogue<DT, 2, 2, DT, -1, false, -1, 0, true>(p, ctx, acc2, lds_bias, lds_red);
    else if (p.out.dtype == DT) epilogue<DT, 2, 2, DT, -1, false, -1, 0, false>(p, ctx, acc2, lds_bias, lds_red);
    else epilogue<DT, 2, 2, kAnyType, kAnyType, true, kAnyType>(p, ctx, acc2, lds_bias, lds_red);
#ifdef SRMI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the stores and atomics of the epilogue count as its time
    STAMP();
#endif
#undef STAMP
#endif
}

static int rcab_configure() {
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    SRMI_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !(configured.load(std::memory_order_acquire) >> dev & 1ull)) {
        SRMI_HIP(hipFuncSetAttribute((const void*)rcab_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RcabGeom::LDS_BYTES));
        SRMI_HIP(hipFuncSetAttribute((const void*)rcab_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)RcabGeom::LDS_BYTES));
        if (tracked) configured.fetch_or(1ull << dev, std::memory_order_release);
    }
    return 0;
}

}  // namespace srmi

using namespace srmi;

int srmi_rcab_prepare_device() { return rcab_configure(); }

extern "C" int srmi_rcab_fused(const srmi_rcab_desc_t* d, void* hip_stream) {
    if (!d) SRMI_FAIL("srmi_rcab_fused: null descriptor");
    if (d->dtype != SRMI_F16) SRMI_FAIL("srmi_rcab_fused: f16 compute only");
    if (d->c != 64) SRMI_FAIL("srmi_rcab_fused: the block must have 64 channels (got %d)", d->c);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) SRMI_FAIL("srmi_rcab_fused: bad shape n=%d h=%d w=%d", d->n, d->h, d->w);
    const bool xf32 = d->in.ptr && d->in.dtype == SRMI_F32;   // the residual trunk in single precision
    if (!d->in.ptr || (d->in.dtype != SRMI_F16 && !xf32)) SRMI_FAIL("srmi_rcab_fused: `in` must be an f16 tensor, or an f32 one with a gated input");
    if (!d->in_gate.ptr || d->in_gate.dtype != SRMI_F16 || !d->gate) SRMI_FAIL("srmi_rcab_fused: in_gate (f16) and gate are required "
                                                                                 "(pass in_gate = in with gate_alpha 0 for a plain input)");
    if (d->in_sum.ptr && (d->in_sum.dtype != d->in.dtype || d->in_sum.ptr == d->in.ptr || d->in_sum.ptr == d->in_gate.ptr))
        SRMI_FAIL("srmi_rcab_fused: in_sum must be a separate tensor of the type of `in`");
    if (xf32 && (long long)d->h * d->w * kPixBytes * 2 >= (1ll << 32)) SRMI_FAIL("srmi_rcab_fused: an f32 `in` needs planes below 2 GiB");
    if (!d->w1 || !d->b1 || !d->w2 || !d->b2) SRMI_FAIL("srmi_rcab_fused: missing weights or bias");
    if (!d->out.ptr || d->out.ptr == d->in_gate.ptr || d->out.ptr == d->in.ptr)
        SRMI_FAIL("srmi_rcab_fused: `out` must be a tensor of its own (other workgroups still read in / in_gate)");
    if (d->ca.sums && (!d->ca.count || !d->ca.w1 || !d->ca.b1 || !d->ca.w2 || !d->ca.b2 || !d->ca.scale || d->ca.cr <= 0 || d->ca.cr > 512))
        SRMI_FAIL("srmi_rcab_fused: incomplete channel-attention block");
    if ((long long)d->h * d->w * kPixBytes >= (1ll << 32)) SRMI_FAIL("srmi_rcab_fused: plane larger than 4 GiB");

    ConvParams p;
    memset(&p, 0, sizeof(p));
    p.n = d->n; p.h = d->h; p.w = d->w;
    p.ncb = d->c / 16;
    p.cout = p.cout_real = d->c;
    p.act = SRMI_ACT_NONE;
    p.ps_r = 1; p.ps_c = d->c;
    p.tiles_x = (d->w + RcabGeom::TW - 1) / RcabGeom::TW;
    p.tiles_y = (d->h + RcabGeom::TH - 1) / RcabGeom::TH;
    p.tail_mode = SRMI_TAIL_NONE;
    p.in = to_view(d->in);
    p.in_plane_bytes = (unsigned)((long long)d->h * d->w * kPixBytes);
    p.wpk = (const char*)d->w1; p.bias = d->b1;
    p.wpk2 = (const char*)d->w2; p.bias2 = d->b2;
    p.in_gate = to_view(d->in_gate); p.in_sum = to_view(d->in_sum);
    p.gate = d->gate; p.gate_alpha = d->gate_alpha;
    p.alpha = p.beta = 1.f;
    p.out = to_view(d->out);
    p.ca_sums = (unsigned long long*)d->ca.sums; p.ca_count = d->ca.count;
    p.ca_w1 = d->ca.w1; p.ca_b1 = d->ca.b1; p.ca_w2 = d->ca.w2; p.ca_b2 = d->ca.b2;
    p.ca_scale = d->ca.scale; p.ca_cr = d->ca.cr;
    p.ca_inv_hw = 1.f / ((float)d->h * (float)d->w);
#if defined(SRMI_DIAG) && defined(SRMI_STAMPS)
    p.tail_out = getenv("SRMI_STAMP_PTR") ? (void*)strtoull(getenv("SRMI_STAMP_PTR"), nullptr, 0) : nullptr;
#endif
    if (rcab_configure()) return -1;
    dim3 grid(p.tiles_x * p.tiles_y, 1, p.n);
    if (xf32) hipLaunchKernelGGL(rcab_fused_kernel<true>, grid, dim3(64 * kNW), RcabGeom::LDS_BYTES, (hipStream_t)hip_stream, p);
    else hipLaunchKernelGGL(rcab_fused_kernel<false>, grid, dim3(64 * kNW), RcabGeom::LDS_BYTES, (hipStream_t)hip_stream, p);
    SRMI_HIP(hipGetLastError());
    return 0;
}
