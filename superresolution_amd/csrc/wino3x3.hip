// 3x3 convolution as Winograd F(2x2,3x3) on the gfx950 matrix cores (f16 operands, f32 accumulation), with the same
// fused epilogue as the direct kernel (conv3x3_kernel.h).  16 multiplies per 2x2 output tile and input channel instead of 36:
// the matrix pipes do 4/9 of the direct kernel's work per algorithmic FLOP, paid for with packed-f16 vector adds.
//
//   Y = A^T [ sum_c (G g_c G^T) .* (B^T d_c B) ] A          d: 4x4 input patch of a tile, g: 3x3 filter, Y: 2x2 outputs
//
// Work decomposition
//   workgroup = 4 waves = 16 rows x 32 columns of one image x 32 output channels
//   wave      = 32 tiles of 2x2 pixels (2 tile rows x 16 tile columns = 4 x 32 pixels): tile index = MFMA column, so
//               lane l holds tile l % 32 for the 8 input channels 8 * (l / 32) .. + 7 of the stage
//   per stage (16 input channels = one plane of the activation layout) and wave:
//     16 ds_read_b128        the 4x4 patch of the lane's tile (one pixel's 8 channels each)
//     32 vector adds (f16x8) V = B^T d B in registers, every add rounding to f16 (128 v_pk_add_f16)
//     16 ds_read_b128        transformed-weight fragments U[xi] (xi = one of the 16 transform positions), packed on the host
//     16 MFMA 32x32x16       acc[xi] += U[xi] (32 outputs x 16 channels) * V[xi] (16 channels x 32 tiles)
//   The 16 accumulators take 256 registers: one wave per SIMD, one workgroup per CU.  The adds that transform stage s+1
//   are issued between the MFMAs of stage s (two or three behind every MFMA), in a second register array.
//   After the last stage: Y = A^T M A in f32 (24 adds per tile and output), then the shared epilogue on the four
//   pixels of the tile.
//
// Data movement: stages are copied global -> LDS by LDS-DMA into a ring of four slots, three stages ahead of the
// multiplies (nothing else runs on the CU that could hide a wait).  A slot holds
//   the pixel patch, 18 x 34 pixels, as two planes of 16-byte chunks [channel half][row][even columns | odd columns]:
//       the four patch columns of neighbouring tiles are then 16 bytes apart for neighbouring lanes (conflict-free
//       ds_read_b128 without a swizzle), and every fragment of a lane sits at a constant offset from one base address;
//       out-of-image pixels read zeros through the buffer descriptor's range check (the convolution's zero padding);
//   the transformed weights of the stage, [xi][channel half][output 32][16 bytes], a linear copy of the host's packing.
#include <atomic>

#include "conv_epilogue.h"

namespace srmi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct WinoGeom {
    static constexpr int TH = 16, PH = TH + 2, PW = kPatchW, HALF = PW / 2;   // 18 x 34 patch, 17 even + 17 odd columns
    static constexpr int NPIX = PH * PW;                   // 612
    static constexpr int PLANE_PIECES = (NPIX + 63) / 64;  // 10 LDS-DMA pieces (1 KiB) per channel half
    static constexpr int PLANE_BYTES = PLANE_PIECES * 1024;
    static constexpr int A_PIECES = 2 * PLANE_PIECES, A_BYTES = A_PIECES * 1024;   // 20 KiB
    static constexpr int B_BYTES = 16 * 32 * 32, B_PIECES = B_BYTES / 1024;        // 16 KiB
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;                           // 36 KiB
    static constexpr int NS = 4;
    static constexpr int LDS_BYTES = NS * STAGE_BYTES;                              // 144 KiB
    static constexpr int BIAS_OFFSET = 8192;
    static constexpr int APW = A_PIECES / kNW, BPW = B_PIECES / kNW, PPW = APW + BPW;  // pieces per wave: 5 + 4
};
static_assert(kNW == 4 && kTileW == 32 && (kPatchW % 2) == 0, "the Winograd kernel's geometry");
static_assert(WinoGeom::A_PIECES % kNW == 0 && WinoGeom::B_PIECES % kNW == 0, "every wave issues the same number of pieces");

// a - b on 8 halves: four v_pk_add_f16 with the second operand negated (a vector subtraction of halves is
// otherwise split into scalar v_sub_f16)
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

__global__ __launch_bounds__(64 * kNW, 1) void wino3x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = WinoGeom;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // workgroups renumbered so that one XCD (own L2) gets a contiguous run of (output tile, spatial tile, image)
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
    const int tx = tile % p.tiles_x, ty = tile / p.tiles_x;
    const int x0 = tx * kTileW, y0 = ty * G::TH;
    const int nstage = p.ncb;

    // ---- LDS-DMA source offsets of this wave's patch pieces (the same for every stage) ----------
    // LDS chunk q of the patch = channel half q / 640, slot pp = q % 640 = row * 34 + s; s < 17: patch column 2s,
    // else patch column 2 (s - 17) + 1.
    unsigned aoff[G::APW];
#pragma unroll
    for (int j = 0; j < G::APW; ++j) {
        const int k = wave + kNW * j;
        const int c = k / G::PLANE_PIECES;
        const int pp = (k - c * G::PLANE_PIECES) * 64 + lane;
        const int row = pp / G::PW, s = pp - row * G::PW;
        const int xi = s < G::HALF ? 2 * s : 2 * (s - G::HALF) + 1;
        const int gy = y0 - 1 + row, gx = x0 - 1 + xi;
        const bool ok = pp < G::NPIX && gy >= 0 && gy < p.h && gx >= 0 && gx < p.w;
        aoff[j] = ok ? ((unsigned)(gy * p.w + gx) * 32u + (unsigned)c * 16u) : SRMI_OOB_OFFSET;
    }
    const char* in_img = p.in.ptr + (long long)n * p.in.img_stride_b;
    const char* w_tile = p.wpk + (size_t)ct * nstage * G::B_BYTES;
    const unsigned plane_bytes = p.in_plane_bytes;
    auto slot_ptr = [&](const int s) { return smem + (s & (G::NS - 1)) * G::STAGE_BYTES; };
    // Piece q of this wave for stage s: q < APW pixel pieces, then weight pieces.  Stages beyond the last are
    // requested all the same, through descriptors of zero size: they fetch nothing and write zeros into a slot nobody
    // reads again, and in exchange the main loop has no branch and every stage has the same number of requests
    // outstanding (the counted waits rely on that).
    auto issue_piece = [&](const int s, const int q) {
        char* slot = slot_ptr(s);
        const bool live = s < nstage;
        if (q < G::APW) {
            const auto ra = make_rsrc(in_img + (long long)(live ? s : 0) * p.in.plane_stride_b, live ? plane_bytes : 0u);
            const int k = wave + kNW * q;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(slot + k * 1024), 16,
                                                     aoff[q < G::APW ? q : 0], 0, 0, 0);
        } else {
            const auto rb = make_rsrc(w_tile + (size_t)(live ? s : 0) * G::B_BYTES, live ? (unsigned)G::B_BYTES : 0u);
            const int k = wave + kNW * (q - G::APW);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(slot + G::A_BYTES + k * 1024),
                                                     16, k * 1024 + lane * 16, 0, 0, 0);
        }
    };

    // ---- per-lane LDS read addresses -----------------------------------------------------------------
    const int lx = lane & 31, lh = lane >> 5;
    const int tr = lx >> 4, tc = lx & 15;
    const int pbase = lh * G::PLANE_BYTES + ((wave * 4 + 2 * tr) * G::PW + tc) * 16;   // patch pixel (0, 0) of the lane's tile
    const int wbase = G::A_BYTES + lh * 512 + lx * 16;                                  // U[0] row of the lane
    // patch pixel (r, j) of the tile: r rows down, column parity j & 1 selects the even / odd half of the row
    auto ld_patch = [&](h8 (&X)[16], const char* slot) {
        // column by column, so that the first column transform can start while the rest is still arriving
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                X[r * 4 + j] = *(const h8*)(slot + pbase + (r * G::PW + (j & 1) * G::HALF + (j >> 1)) * 16);
    };
    auto ld_w = [&](const char* slot, const int xi) { return *(const h8*)(slot + wbase + xi * 1024); };

    // ---- input transform, one vector add per step ---------------------------------------------------------
    // X holds d[r][j] at X[4r + j] and is turned into V = B^T d B in place: steps 0-15 the four columns (B^T d:
    // t0 = d0 - d2, t1 = d1 + d2, t2 = d2 - d1, t3 = d1 - d3), steps 16-31 the four rows (the same map along j).
    h8 xf_tmp;
    auto xf_step = [&](h8 (&X)[16], const int st) {
        const int half = st >> 4, line = (st >> 2) & 3, op = st & 3;
        auto at = [&](const int e) -> h8& { return half == 0 ? X[e * 4 + line] : X[line * 4 + e]; };
        if (op == 0) at(0) = sub8(at(0), at(2));
        else if (op == 1) at(3) = sub8(at(1), at(3));
        else if (op == 2) xf_tmp = at(1) + at(2);
        else { at(2) = sub8(at(2), at(1)); at(1) = xf_tmp; }
    };

    f32x16 acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[k][i] = 0.f;

    // ---- prologue: three stages requested, the first one transformed before the first multiply ------------
    constexpr int kAhead = G::NS - 1;
    const float bias_v = tid < 32 ? p.bias[ct * 32 + tid] : 0.f;
#pragma unroll
    for (int s = 0; s < kAhead; ++s)
#pragma unroll
        for (int q = 0; q < G::PPW; ++q) issue_piece(s, q);
    // stage 0 (and the bias, requested before it) has landed once only the two younger stages are outstanding
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((kAhead - 1) * G::PPW) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    h8 P[16], Q[16], wf[8];
    ld_patch(P, slot_ptr(0));
#pragma unroll
    for (int k = 0; k < 8; ++k) wf[k] = ld_w(slot_ptr(0), k);
#pragma unroll
    for (int st = 0; st < 32; ++st) xf_step(P, st);

    // ---- main loop ----------------------------------------------------------------------------------------------
    // Stage s: the MFMAs read V (stage s, already transformed) and the weights of slot s; X receives the patch of
    // stage s + 1 and is transformed between the MFMAs; the weight fragments of the second half of the stage, then of
    // the first half of stage s + 1, refill the 8-deep fragment ring; stage s + 3 is requested piece by piece.
    // Every stage runs the same code, the last one included: the "next" patch it transforms is the zeros that the
    // zero-size requests wrote (never multiplied), which costs the last stage's spare vector slots and keeps the whole
    // reduction in one loop without a branch inside a stage.
    auto run_stage = [&](h8 (&V)[16], h8 (&X)[16], const int s) {
        // stage s + 1 has landed (its patch is read now) once only stage s + 2 is outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::PPW) : "memory");
        // everybody's share of stage s + 1 is in LDS, and everybody is done with slot s - 1, which stage s + 3 overwrites
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* slot_w = slot_ptr(s);
        const char* slot_n = slot_ptr(s + 1);
        ld_patch(X, slot_n);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[k & 7], V[k], acc[k], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (k < 8) wf[k] = ld_w(slot_w, k + 8);
            else wf[k - 8] = ld_w(slot_n, k - 8);
            if (k < G::PPW) issue_piece(s + kAhead, k);
            // 32 transform steps behind the MFMAs 1..15: three each behind the first two, then two each
            const int first = k == 0 ? 0 : (k <= 2 ? 3 * (k - 1) : 6 + 2 * (k - 3));
            const int count = k == 0 ? 0 : (k <= 2 ? 3 : 2);
#pragma unroll
            for (int e = 0; e < count; ++e) xf_step(X, first + e);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the stage count is even (cin a multiple of 32, checked by the host): one loop body = two stages, one exit
    for (int s = 0; s < nstage; s += 2) {
        run_stage(P, Q, s);
        run_stage(Q, P, s + 1);
    }
    // requests for stages beyond the last may still be writing zeros into the slots the epilogue reuses
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- output transform: Y = A^T M A, A^T = [[1, 1, 1, 0], [0, 1, -1, -1]] --------------------------------
    f32x16 o[4][1];
    {
        f32x16 s0[4], s1[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            s0[b] = acc[0 + b] + acc[4 + b] + acc[8 + b];
            s1[b] = acc[4 + b] - acc[8 + b] - acc[12 + b];
        }
        o[0][0] = s0[0] + s0[1] + s0[2];
        o[1][0] = s0[1] - s0[2] - s0[3];
        o[2][0] = s1[0] + s1[1] + s1[2];
        o[3][0] = s1[1] - s1[2] - s1[3];
    }
    // the epilogue reuses the stage buffers (bias, channel sums): all waves must be out of the main loop first
    __syncthreads();
    if (tid < 32) ((float*)(smem + G::BIAS_OFFSET))[tid] = bias_v;
    __syncthreads();

    const EpiCtx ctx{n, ct, tile, x0, y0, wave, lx, lh, tid};
    const float* lds_bias = (const float*)(smem + G::BIAS_OFFSET);
    float* lds_red = (float*)smem;
    constexpr int DT = SRMI_F16;
    const int od = p.out.ptr ? p.out.dtype : -1, rd = p.res1.ptr ? p.res1.dtype : -1;
    const int r2 = p.res2.ptr ? p.res2.dtype : -1, o2 = p.out2.ptr ? p.out2.dtype : -1;
    const bool plain_io = p.tail_mode == SRMI_TAIL_NONE && od == DT && (r2 == -1 || r2 == rd);
    if (plain_io && rd == -1 && o2 == -1) {
        epilogue<DT, 4, 1, DT, -1, false, -1, 1, false>(p, ctx, o, lds_bias, lds_red);
    } else if (plain_io && rd == SRMI_F32 && r2 == -1 && o2 == SRMI_F32) {
        epilogue<DT, 4, 1, DT, SRMI_F32, false, SRMI_F32, 1, false>(p, ctx, o, lds_bias, lds_red);
    } else if (plain_io && rd == SRMI_F32 && r2 == SRMI_F32 && o2 == SRMI_F32) {
        epilogue<DT, 4, 1, DT, SRMI_F32, true, SRMI_F32, 1, false>(p, ctx, o, lds_bias, lds_red);
    } else if (plain_io && rd == DT && r2 == -1 && o2 == -1) {
        epilogue<DT, 4, 1, DT, DT, false, -1, 1, false>(p, ctx, o, lds_bias, lds_red);
    } else if (plain_io && rd == DT && r2 == DT && o2 == DT) {
        epilogue<DT, 4, 1, DT, DT, true, DT, 1, false>(p, ctx, o, lds_bias, lds_red);
    } else {
        epilogue<DT, 4, 1, kAnyType, kAnyType, true, kAnyType, 1>(p, ctx, o, lds_bias, lds_red);
    }
#endif
}

static int wino_configure() {
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    SRMI_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !(configured.load(std::memory_order_acquire) >> dev & 1ull)) {
        SRMI_HIP(hipFuncSetAttribute((const void*)wino3x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)WinoGeom::LDS_BYTES));
        if (tracked) configured.fetch_or(1ull << dev, std::memory_order_release);
    }
    return 0;
}

}  // namespace srmi

using namespace srmi;

int srmi_wino3x3_prepare_device() { return wino_configure(); }

int srmi_wino3x3_tile_rows() { return WinoGeom::TH; }

// Called by srmi_conv3x3 (conv3x3.hip) for descriptors with algo == SRMI_ALGO_WINOGRAD, after the common checks.
int srmi_wino3x3_launch(const ConvParams& p, hipStream_t stream) {
    if (wino_configure()) return -1;
    dim3 grid(p.tiles_x * p.tiles_y, p.cout / 32, p.n);
    hipLaunchKernelGGL(wino3x3_kernel, grid, dim3(64 * kNW), WinoGeom::LDS_BYTES, stream, p);
    SRMI_HIP(hipGetLastError());
    return 0;
}
