// 3x3 convolution with Winograd F(2,3) along x on the gfx950 matrix cores (f16 operands, f32 accumulation), same fused
// epilogue as the direct kernel (conv3x3_kernel.h).  Per pair of neighbouring output pixels and vertical tap, 4 multiplies
// instead of 6: the matrix pipes do 2/3 of the direct kernel's work per algorithmic FLOP, paid for with 16 packed-f16
// vector adds of 8 channels per 24 MFMAs.
//
//   Y[r][2t .. 2t+1] = A^T sum_{ky, c} (G g[ky][.]) .* (B^T d[r + ky][2t-1 .. 2t+2])
//   B^T d = (d0 - d2, d1 + d2, d2 - d1, d1 - d3)      G g = (g0, (g0 + g1 + g2) / 2, (g0 - g1 + g2) / 2, g2)
//   A^T m = (m0 + m1 + m2, m1 - m2 - m3)
//
// Work decomposition
//   workgroup = 4 waves = 8 rows x 64 columns of one image x 32 output channels
//   wave      = 2 rows x 32 tiles of two pixels: tile column = MFMA column, lane l holds tile l % 32 for the 8 input
//               channels 8 * (l / 32) .. + 7 of the stage.  8 accumulators (2 rows x 4 transform positions) = 128
//               registers: two waves per SIMD, two workgroups per CU.
//   per stage (16 input channels) and wave: the four patch rows of its two output rows are walked one at a time (a
//   transformed row feeds the vertical taps of up to two output rows): 4 ds_read_b128 + 4 vector adds per row, 12
//   transformed-weight fragments per stage, 24 MFMAs.  The next row is loaded and transformed between the MFMAs of the
//   current one.
//
// Data movement: stages are copied global -> LDS by LDS-DMA into a two-slot ring, the next stage's pieces issued
// between the multiplies of the current one.  A slot holds the pixel patch, 10 x 66 pixels, as two planes of 16-byte
// chunks [channel half][row][even columns | odd columns] (the four patch columns of neighbouring tiles are 16 bytes
// apart for neighbouring lanes: conflict-free ds_read_b128 at constant offsets from one base address; out-of-image pixels
// read zeros through the buffer descriptor's range check), and the stage's transformed weights
// [vertical tap][position][channel half][output 32][16 bytes], a linear copy of the host's packing.
#include <atomic>

#include "conv_epilogue.h"

namespace srmi {

typedef _Float16 h8 __attribute__((ext_vector_type(8)));

struct Wino1dGeom {
    static constexpr int TH = 2 * kNW, TW = 64, PH = TH + 2, PW = TW + 2, HALF = PW / 2;   // 10 x 66 patch, 33 even + 33 odd columns
    static constexpr int NPIX = PH * PW;                    // 660
    static constexpr int PLANE_PIECES = (NPIX + 63) / 64;   // 11 LDS-DMA pieces (1 KiB) per channel half
    static constexpr int PLANE_BYTES = PLANE_PIECES * 1024;
    static constexpr int A_PIECES = 2 * PLANE_PIECES, A_BYTES = A_PIECES * 1024;   // 22 KiB
    static constexpr int B_BYTES = 12 * 32 * 32, B_PIECES = B_BYTES / 1024;        // 12 KiB
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;                           // 34 KiB
    static constexpr int NS = 2;
    static constexpr int LDS_BYTES = NS * STAGE_BYTES;                              // 68 KiB: two workgroups per CU
    static constexpr int BIAS_OFFSET = 8192;
    static constexpr int APW = (A_PIECES + kNW - 1) / kNW, BPW = B_PIECES / kNW, PPW = APW + BPW;   // pieces per wave: 6 (or 5) + 3
};
static_assert(kNW == 4 && Wino1dGeom::B_PIECES % kNW == 0, "the F(2,3) kernel's geometry");
static_assert(Wino1dGeom::PPW <= 24, "a stage has a multiply for every DMA piece of a wave");

__device__ __forceinline__ h8 sub8h(const h8& a, const h8& b) {
    // a - b on 8 halves: four v_pk_add_f16 with the second operand negated (a vector subtraction of halves is otherwise
    // split into scalar v_sub_f16)
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

__global__ __launch_bounds__(64 * kNW, 2) void wino1d3x3_kernel(const ConvParams p) {
#if defined(__HIP_DEVICE_COMPILE__)
    using G = Wino1dGeom;
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
    const int x0 = tx * G::TW, y0 = ty * G::TH;
    const int nstage = p.ncb;
#ifdef SRMI_STAMPS
    // diagnostic build: lane 0 of every wave records s_memtime at each phase boundary into [workgroup][wave][40];
    // slot 39 holds (XCC_ID << 32) | HW_ID
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
    STAMP();

    // ---- LDS-DMA source offsets of this wave's patch pieces (the same for every stage) ----------
    // LDS chunk q of the patch = channel half q / 704, slot pp = q % 704 = row * 66 + s; s < 33: patch column 2s, else
    // patch column 2 (s - 33) + 1.
    // (conditions combined with & on values read once: short-circuit forms compile to a branch per term, each with its
    // own reload of the image size from the argument block)
    unsigned aoff[G::APW];
    const unsigned img_h = (unsigned)p.h, img_w = (unsigned)p.w;
#pragma unroll
    for (int j = 0; j < G::APW; ++j) {
        const int k = wave + kNW * j;
        const int c = k / G::PLANE_PIECES;
        const int pp = (k - c * G::PLANE_PIECES) * 64 + lane;
        const int row = pp / G::PW, s = pp - row * G::PW;
        const int xi = s < G::HALF ? 2 * s : 2 * (s - G::HALF) + 1;
        const int gy = y0 - 1 + row, gx = x0 - 1 + xi;
        const bool ok = (k < G::A_PIECES) & (pp < G::NPIX) & ((unsigned)gy < img_h) & ((unsigned)gx < img_w);
        aoff[j] = ok ? ((unsigned)gy * img_w + (unsigned)gx) * 32u + (unsigned)c * 16u : SRMI_OOB_OFFSET;
    }
    const char* in_img = p.in.ptr + (long long)n * p.in.img_stride_b;
    const char* w_tile = p.wpk + (size_t)ct * nstage * G::B_BYTES;
    const unsigned plane_bytes = p.in_plane_bytes;
    // Piece q of this wave for stage s: q < APW pixel pieces, then weight pieces.  A stage beyond the last is requested
    // all the same through descriptors of zero size (fetches nothing, writes zeros into a slot nobody reads again), and a
    // wave without a pixel piece in its last round repeats its previous one: no branch in the main loop.
    auto issue_piece = [&](const int s, const int q) {
        char* slot = smem + (s & 1) * G::STAGE_BYTES;
        const bool live = s < nstage;
        if (q < G::APW) {
            const auto ra = make_rsrc(in_img + (long long)(live ? s : 0) * p.in.plane_stride_b, live ? plane_bytes : 0u);
            const int kq = wave + kNW * q;
            const bool mine = kq < G::A_PIECES;
            const int k = mine ? kq : kq - kNW;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (__attribute__((address_space(3))) void*)(slot + k * 1024), 16,
                                                     mine ? aoff[q] : aoff[q > 0 ? q - 1 : 0], 0, 0, 0);
        } else {
            const auto rb = make_rsrc(w_tile + (size_t)(live ? s : 0) * G::B_BYTES, live ? (unsigned)G::B_BYTES : 0u);
            const int k = wave + kNW * (q - G::APW);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (__attribute__((address_space(3))) void*)(slot + G::A_BYTES + k * 1024),
                                                     16, k * 1024 + lane * 16, 0, 0, 0);
        }
    };

    // ---- per-lane LDS read addresses -----------------------------------------------------------------
    const int lx = lane & 31, lh = lane >> 5;
    const int pbase = lh * G::PLANE_BYTES + ((2 * wave) * G::PW + lx) * 16;   // patch pixel (row 2 wave, column 2 lx) of the lane's tile
    const int wbase = G::A_BYTES + lh * 512 + lx * 16;
    auto ld_row = [&](h8 (&X)[4], const char* slot, const int rho) {
#pragma unroll
        for (int j = 0; j < 4; ++j) X[j] = *(const h8*)(slot + pbase + (rho * G::PW + (j & 1) * G::HALF + (j >> 1)) * 16);
    };
    auto ld_w = [&](h8 (&W)[4], const char* slot, const int ky) {
#pragma unroll
        for (int xi = 0; xi < 4; ++xi) W[xi] = *(const h8*)(slot + wbase + (ky * 4 + xi) * 1024);
    };
    // B^T d in place, one vector add per step (step 3 only renames): (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    h8 xf_tmp;
    auto xf_step = [&](h8 (&X)[4], const int op) {
        if (op == 0) X[0] = sub8h(X[0], X[2]);
        else if (op == 1) X[3] = sub8h(X[1], X[3]);
        else if (op == 2) xf_tmp = X[1] + X[2];
        else { X[2] = sub8h(X[2], X[1]); X[1] = xf_tmp; }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][k][i] = 0.f;

    const float bias_v = tid < 32 ? p.bias[ct * 32 + tid] : 0.f;
#pragma unroll
    for (int q = 0; q < G::PPW; ++q) issue_piece(0, q);
    STAMP();

    // ---- main loop: one stage = four patch rows; row rho feeds output rows r = rho - ky -------------------------
    // Rows 0 and 1 are requested at the stage's start; row rho + 2 is requested into the registers of row rho as soon as
    // its last multiply has issued, and transformed behind the last four multiplies of row rho + 1: every LDS read has
    // at least four multiplies to arrive before its first use, only row 0's read and transform are exposed per stage.
    // (A stage takes a lone wave 1.5 k cycles for 0.77 k of multiplies either way: a wave issues its vector adds, LDS
    // reads and multiplies one after the other, and it is the second wave of the SIMD that fills the matrix pipe.)
    // (Measured and not kept: the stage barrier behind the stage's last LDS read, with the next stage's first row loaded
    // and transformed under the last row's multiplies.  It asks for the next stage four multiplies earlier, and the wait
    // for its LDS-DMA costs more than the bubble it removes.)
    h8 Ra[4], Rb[4], wf[3][4];
    for (int s = 0; s < nstage; ++s) {
        // stage s has landed for everybody, and everybody is done reading the other slot
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        STAMP();
        const char* slot = smem + (s & 1) * G::STAGE_BYTES;
        ld_row(Ra, slot, 0);
        ld_w(wf[0], slot, 0);
        ld_row(Rb, slot, 1);
        ld_w(wf[1], slot, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int op = 0; op < 4; ++op) xf_step(Ra, op);
        __builtin_amdgcn_sched_barrier(0);
        int mi = 0;   // multiply index within the stage: the next stage's pieces go behind the first PPW of them
        auto row = [&](h8 (&V)[4], h8 (&X)[4], const int rho) {
            const int nm = (rho == 0 || rho == 3) ? 4 : 8;   // multiplies of this row
            int m = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int r = rho - ky;
                if (r < 0 || r > 1) continue;
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    acc[r][xi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[ky][xi], V[xi], acc[r][xi], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (mi < G::PPW) issue_piece(s + 1, mi);
                    if (rho < 3 && m >= nm - 4) xf_step(X, m - (nm - 4));
                    ++m;
                    ++mi;
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // this row's registers are free once its last multiply has issued
            if (rho < 2) ld_row(V, slot, rho + 2);
            if (rho == 0) ld_w(wf[2], slot, 2);
            __builtin_amdgcn_sched_barrier(0);
        };
        row(Ra, Rb, 0);
        row(Rb, Ra, 1);
        row(Ra, Rb, 2);
        row(Rb, Ra, 3);
        STAMP();
    }

    // ---- output transform: (m0 + m1 + m2, m1 - m2 - m3) per output row ----------------------------------------------
    f32x16 o[4][1];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        o[2 * r][0] = acc[r][0] + acc[r][1] + acc[r][2];
        o[2 * r + 1][0] = acc[r][1] - acc[r][2] - acc[r][3];
    }
    // the epilogue reuses the stage buffers (bias, channel sums): the zero-size requests behind the last stage must have
    // landed and all waves must be out of the main loop first
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid < 32) ((float*)(smem + G::BIAS_OFFSET))[tid] = bias_v;
    __syncthreads();
    STAMP();

    const EpiCtx ctx{n, ct, tile, x0, y0, wave, lx, lh, tid};
    const float* lds_bias = (const float*)(smem + G::BIAS_OFFSET);
    float* lds_red = (float*)smem;
    constexpr int DT = SRMI_F16;
    const int od = p.out.ptr ? p.out.dtype : -1, rd = p.res1.ptr ? p.res1.dtype : -1;
    const int r2 = p.res2.ptr ? p.res2.dtype : -1, o2 = p.out2.ptr ? p.out2.dtype : -1;
    const bool plain_io = p.tail_mode == SRMI_TAIL_NONE && od == DT && (r2 == -1 || r2 == rd);
    if (plain_io && rd == -1 && o2 == -1) {
        epilogue<DT, 4, 1, DT, -1, false, -1, 2, false>(p, ctx, o, lds_bias, lds_red);
    } else if (plain_io && rd == SRMI_F32 && r2 == -1 && o2 == SRMI_F32) {
        epilogue<DT, 4, 1, DT, SRMI_F32, false, SRMI_F32, 2, false>(p, ctx, o, lds_bias, lds_red);
    } else if (plain_io && rd == SRMI_F32 && r2 == SRMI_F32 && o2 == SRMI_F32) {
        epilogue<DT, 4, 1, DT, SRMI_F32, true, SRMI_F32, 2, false>(p, ctx, o, lds_bias, lds_red);
    } else if (plain_io && rd == DT && r2 == -1 && o2 == -1) {
        epilogue<DT, 4, 1, DT, DT, false, -1, 2, false>(p, ctx, o, lds_bias, lds_red);
    } else if (plain_io && rd == DT && r2 == DT && o2 == DT) {
        epilogue<DT, 4, 1, DT, DT, true, DT, 2, false>(p, ctx, o, lds_bias, lds_red);
    } else {
        epilogue<DT, 4, 1, kAnyType, kAnyType, true, kAnyType, 2>(p, ctx, o, lds_bias, lds_red);
    }
#ifdef SRMI_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the epilogue's stores count as its time
    STAMP();
#endif
#undef STAMP
#endif
}

static int wino1d_configure() {
    static std::atomic<unsigned long long> configured{0};
    int dev = 0;
    SRMI_HIP(hipGetDevice(&dev));
    const bool tracked = dev >= 0 && dev < 64;
    if (!tracked || !(configured.load(std::memory_order_acquire) >> dev & 1ull)) {
        SRMI_HIP(hipFuncSetAttribute((const void*)wino1d3x3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Wino1dGeom::LDS_BYTES));
        if (tracked) configured.fetch_or(1ull << dev, std::memory_order_release);
    }
    return 0;
}

}  // namespace srmi

using namespace srmi;

int srmi_wino1d3x3_prepare_device() { return wino1d_configure(); }

void srmi_wino1d3x3_tile(int* rows, int* cols) {
    *rows = Wino1dGeom::TH;
    *cols = Wino1dGeom::TW;
}

// Called by srmi_conv3x3 (conv3x3.hip) for descriptors with algo == SRMI_ALGO_WINOGRAD_X, after the common checks.
int srmi_wino1d3x3_launch(const ConvParams& p, hipStream_t stream) {
    if (wino1d_configure()) return -1;
    dim3 grid(p.tiles_x * p.tiles_y, p.cout / 32, p.n);
    hipLaunchKernelGGL(wino1d3x3_kernel, grid, dim3(64 * kNW), Wino1dGeom::LDS_BYTES, stream, p);
    SRMI_HIP(hipGetLastError());
    return 0;
}
