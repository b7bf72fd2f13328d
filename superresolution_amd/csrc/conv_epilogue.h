// Shared by the convolution kernels (conv3x3_kernel.h: direct implicit GEMM; wino3x3.hip: Winograd F(2x2,3x3)): buffer
// descriptors, the counted vmcnt wait and the fused epilogue (bias, activation, residuals, typed stores, pixel shuffle,
// channel sums / channel attention, RGB tail).  Both kernels leave their results in the accumulator layout of
// v_mfma_f32_32x32x16 with weights as the A operand and pixels as the B operand: a lane holds, for ONE pixel column,
// the output channels 8*(i/4) + 4*(lane/32) + i%4 of a 32-channel block in register i.
#pragma once
#include "common.h"
#include "devutil.h"

namespace srmi {

// waves per workgroup (diagnostic builds may override)
#ifndef SRMI_NW
#define SRMI_NW 4
#endif
constexpr int kNW = SRMI_NW;

#define SRMI_OOB_OFFSET 0xFFFFFFF0u  // beyond every descriptor's num_records -> reads 0

#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}
#endif

// ---- epilogue ------------------------------------------------------------------------------------
struct EpiCtx {
    int n, ct, tile, x0, y0, wave, lx, lh, tid;
};

constexpr int kAnyType = 99;  // dtype resolved at run time (generic path)

// 8 consecutive channels (c % 8 == 0) of one pixel; TYPE is a dtype code, kAnyType, or -1 (absent)
template <int TYPE>
__device__ __forceinline__ void ld8(const TView& t, int n, long long pix, int c, float (&v)[8]) {
    if constexpr (TYPE == kAnyType) {
        load8(t, n, pix, c, v);
    } else if constexpr (TYPE == SRMI_F32) {
        const char* a = t.ptr + (long long)n * t.img_stride_b + (long long)(c >> 3) * t.plane_stride_b + pix * 32 + (c & 7) * 4;
        const f32x4 x0 = *(const f32x4*)a, x1 = *(const f32x4*)(a + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = x0[j]; v[4 + j] = x1[j]; }
    } else if constexpr (TYPE == SRMI_I8) {
        load8(t, n, pix, c, v);
    } else {
        const char* a = t.ptr + (long long)n * t.img_stride_b + (long long)(c >> 4) * t.plane_stride_b + pix * 32 + (c & 15) * 2;
        if constexpr (TYPE == SRMI_F16) {
            const f16x8 x = *(const f16x8*)a;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
        } else {
            const bf16x8 x = *(const bf16x8*)a;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
        }
    }
}

template <int TYPE>
__device__ __forceinline__ void st8(const TView& t, int n, long long pix, int c, const float (&v)[8]) {
    if constexpr (TYPE == kAnyType) {
        store8(t, n, pix, c, v);
    } else if constexpr (TYPE == SRMI_F32) {
        char* a = t.ptr + (long long)n * t.img_stride_b + (long long)(c >> 3) * t.plane_stride_b + pix * 32 + (c & 7) * 4;
        const f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
        *(f32x4*)a = x0;
        *(f32x4*)(a + 16) = x1;
    } else if constexpr (TYPE == SRMI_I8) {
        char* a = t.ptr + (long long)n * t.img_stride_b + (long long)(c >> 5) * t.plane_stride_b + pix * 32 + (c & 31);
        *(u32x2*)a = quant8_i8(v, t.qscale);
    } else {
        char* a = t.ptr + (long long)n * t.img_stride_b + (long long)(c >> 4) * t.plane_stride_b + pix * 32 + (c & 15) * 2;
        if constexpr (TYPE == SRMI_F16) {
            f16x8 x;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = to_f16_sat(v[j]);
            *(f16x8*)a = x;
        } else {
            bf16x8 x;
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (__bf16)v[j];
            *(bf16x8*)a = x;
        }
    }
}

// Sum over the 32 lanes of each half wave, valid in lanes 31 and 63: six DPP adds on the vector pipe
// (row shifts build the 16-lane row sums in lane 15 of every row, row_bcast:15 carries rows 0 and 2
// into rows 1 and 3) instead of five cross-lane shuffles through the LDS crossbar.
template <int CTRL, int ROW_MASK, int BANK_MASK>
__device__ __forceinline__ float dpp_move(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, false));
}
__device__ __forceinline__ float half_wave_sum(float v) {
    float s = v + dpp_move<0x111, 0xf, 0xf>(v);      // row_shr:1
    s += dpp_move<0x112, 0xf, 0xf>(v);                // row_shr:2
    s += dpp_move<0x113, 0xf, 0xf>(v);                // row_shr:3  -> sums of 4 behind every lane
    s += dpp_move<0x114, 0xf, 0xe>(s);                // row_shr:4, banks 1-3
    s += dpp_move<0x118, 0xf, 0xc>(s);                // row_shr:8, banks 2-3 -> lane 15 of a row = row sum
    s += dpp_move<0x142, 0xa, 0xf>(s);                // row_bcast:15 into rows 1 and 3
    return s;
}

// After the half-wave exchange a lane owns, for its pixel, the 8 consecutive channels
// 16*q + 8*lh .. +7 of every 32-channel block (q = 0, 1): all tensor accesses are 16 or 32 bytes.
// WINO != 0: the accumulators come out of a Winograd kernel's output transform (MB == 4 pixels per lane).
//   1 (F(2x2,3x3)): "row" r of a lane is pixel (r >> 1, r & 1) of its 2x2 output tile; the wave's 32 tiles lie as 2 tile
//     rows x 16 tile columns;
//   2 (F(2,3) along x): "row" r is pixel (row r >> 1, column r & 1) of the lane's pair of two-pixel tiles in the wave's
//     two image rows; the wave's 32 tile columns span 64 pixels.
// SUMS: the per-tile channel sums (chansum / channel attention) are compiled in; callers that know there are none pass
// false and save the 16 NB registers of the running sums (a 64-output tile has none to spare).
template <int DT, int MB, int NB, int OUT, int RES, bool RES2, int OUT2, int WINO = 0, bool SUMS = true>
__device__ __forceinline__ void epilogue(const ConvParams& p, const EpiCtx& k, f32x16 (&acc)[MB][NB],
                                         const float* lds_bias, float* lds_red) {
    constexpr bool generic = OUT == kAnyType;
    static_assert(!WINO || MB == 4, "a 2x2 output tile has four pixels");
    // image coordinates of accumulator row r of this lane
    auto row_y = [&](const int r) {
        if constexpr (WINO == 1) return k.y0 + k.wave * 4 + 2 * (k.lx >> 4) + (r >> 1);
        else if constexpr (WINO == 2) return k.y0 + k.wave * 2 + (r >> 1);
        else return k.y0 + k.wave * MB + r;
    };
    auto row_x = [&](const int r) {
        if constexpr (WINO == 1) return k.x0 + 2 * (k.lx & 15) + (r & 1);
        else if constexpr (WINO == 2) return k.x0 + 2 * k.lx + (r & 1);
        else return k.x0 + k.lx;
    };
    const bool tail = generic && p.tail_mode != SRMI_TAIL_NONE;
    // The lane's bias values: kept in registers for 32-output tiles; a 64-output tile (96 accumulator registers) reads
    // them from LDS again for every row instead of holding 32 more registers through the whole epilogue.
    constexpr bool bias_per_row = NB > 1;
    float bias_r[NB][2][8];
    auto load_bias = [&](const int lane_half) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* bp = lds_bias + b * 32 + 16 * q + 8 * lane_half;
                const f32x4 b0 = *(const f32x4*)bp, b1 = *(const f32x4*)(bp + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { bias_r[b][q][j] = b0[j]; bias_r[b][q][4 + j] = b1[j]; }
            }
    };
    if constexpr (!bias_per_row) load_bias(k.lh);
    // int8 compute: per-output-channel dequantisation scales, laid out like the bias
    float scale_r[DT == SRMI_I8 ? NB : 1][2][8];
    if constexpr (DT == SRMI_I8) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const float* sp = lds_bias + 32 * NB + b * 32 + 16 * q + 8 * k.lh;
                const f32x4 s0 = *(const f32x4*)sp, s1 = *(const f32x4*)(sp + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { scale_r[b][q][j] = s0[j]; scale_r[b][q][4 + j] = s1[j]; }
            }
    }
    const bool want_sums = SUMS && (p.chansum != nullptr || p.ca_sums != nullptr);
    float csum[SUMS ? NB : 1][2][8];
    if constexpr (SUMS) {
        if (want_sums) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) csum[b][q][j] = 0.f;
        }
    }
    const int ow = p.w * p.ps_r;
    const bool has_res1 = generic ? p.res1.ptr != nullptr : RES != -1;
    const bool has_res2 = generic ? p.res2.ptr != nullptr : RES2;
    const bool has_out = generic ? p.out.ptr != nullptr : OUT != -1;
    const bool has_out2 = generic ? p.out2.ptr != nullptr : OUT2 != -1;
    // Half-precision residuals of all MB rows are requested before anything else, as raw 16-byte
    // loads: their latencies overlap each other and the bias / exchange work instead of being paid
    // once per row and residual.  (f32 residuals and 64-wide tiles would need twice the registers
    // and keep the per-row loads below.)
    constexpr bool prefetch_res = !generic && NB == 1 && (RES == SRMI_F16 || RES == SRMI_BF16);
    u32x4 raw1[prefetch_res ? MB : 1][NB][2], raw2[(prefetch_res && RES2) ? MB : 1][NB][2];
    if constexpr (prefetch_res) {
#pragma unroll
        for (int r = 0; r < MB; ++r) {
            const int y = row_y(r), x = row_x(r);
            const long long pix = ((y < p.h) && (x < p.w)) ? (long long)y * p.w + x : 0;
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int c = (k.ct * NB + b) * 32 + 16 * q + 8 * k.lh;
                    raw1[r][b][q] = *(const u32x4*)(p.res1.ptr + (long long)k.n * p.res1.img_stride_b +
                                                    (long long)(c >> 4) * p.res1.plane_stride_b + pix * 32 + (c & 15) * 2);
                    if constexpr (RES2)
                        raw2[r][b][q] = *(const u32x4*)(p.res2.ptr + (long long)k.n * p.res2.img_stride_b +
                                                        (long long)(c >> 4) * p.res2.plane_stride_b + pix * 32 + (c & 15) * 2);
                }
        }
    }
    auto unpack8 = [](const u32x4& raw, float (&o)[8]) {
        if constexpr (RES == SRMI_F16) {
            const f16x8 h = __builtin_bit_cast(f16x8, raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (float)h[j];
        } else {
            const bf16x8 h = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
            for (int j = 0; j < 8; ++j) o[j] = (float)h[j];
        }
    };
#pragma unroll
    for (int r = 0; r < MB; ++r) {
        const int y = row_y(r), x = row_x(r);
        const bool inside = (y < p.h) && (x < p.w);
        const long long pix = inside ? (long long)y * p.w + x : 0;   // dead lanes read pixel 0
        if constexpr (bias_per_row) {
            int lane_half = k.lh;
            asm volatile("" : "+v"(lane_half));   // opaque per row: these reads are not merged with the other rows'
            load_bias(lane_half);
        }
        float v[NB][2][8];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // half-wave exchange: lanes 0-31 keep group 2q and receive the partner's,
                    // lanes 32-63 receive the partner's group 2q+1 and keep their own
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(acc[r][b][8 * q + j]),
                                                                     __float_as_uint(acc[r][b][8 * q + 4 + j]),
                                                                     false, false);
                    if constexpr (DT == SRMI_I8) {
                        v[b][q][j] = (float)(int)sw[0] * scale_r[b][q][j] + bias_r[b][q][j];
                        v[b][q][4 + j] = (float)(int)sw[1] * scale_r[b][q][4 + j] + bias_r[b][q][4 + j];
                    } else {
                        v[b][q][j] = __uint_as_float(sw[0]) + bias_r[b][q][j];
                        v[b][q][4 + j] = __uint_as_float(sw[1]) + bias_r[b][q][4 + j];
                    }
                }
            }
        if (p.act == SRMI_ACT_RELU) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[b][q][j] = fmaxf(v[b][q][j], 0.f);
        } else if (p.act == SRMI_ACT_LRELU) {
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[b][q][j] = fmaxf(v[b][q][j], v[b][q][j] * 0.2f);  // slope < 1
        }
        if (has_res1) {
            float rr[NB][2][8];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if constexpr (prefetch_res) unpack8(raw1[r][b][q], rr[b][q]);
                    else ld8<RES>(p.res1, k.n, pix, (k.ct * NB + b) * 32 + 16 * q + 8 * k.lh, rr[b][q]);
                }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[b][q][j] = v[b][q][j] * p.alpha + rr[b][q][j];
        }
        if (has_res2) {
            float rr[NB][2][8];
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if constexpr (prefetch_res && RES2) unpack8(raw2[r][b][q], rr[b][q]);
                    else ld8<RES>(p.res2, k.n, pix, (k.ct * NB + b) * 32 + 16 * q + 8 * k.lh, rr[b][q]);
                }
#pragma unroll
            for (int b = 0; b < NB; ++b)
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[b][q][j] = v[b][q][j] * p.beta + rr[b][q][j];
        }
        if (inside && !tail) {
            if (has_out2) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        st8<OUT2>(p.out2, k.n, pix, (k.ct * NB + b) * 32 + 16 * q + 8 * k.lh, v[b][q]);
            }
            if (has_out) {
                const bool round8 = (p.out.flags & SRMI_TENSOR_ROUND8) != 0;
                if (round8) {
                    // 8 significant bits (round to nearest even), then the cast of the store below
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int j = 0; j < 8; ++j) v[b][q][j] = (float)(__bf16)v[b][q][j];
                }
                if (p.ps_r == 1) {
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int q = 0; q < 2; ++q)
                            st8<OUT>(p.out, k.n, pix, (k.ct * NB + b) * 32 + 16 * q + 8 * k.lh, v[b][q]);
                } else {
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int cq = (k.ct * NB + b) * 32 + 16 * q + 8 * k.lh;
                            const int ij = cq / p.ps_c, cc = cq - ij * p.ps_c;
                            const int oy = y * p.ps_r + ij / p.ps_r, ox = x * p.ps_r + ij % p.ps_r;
                            st8<OUT>(p.out, k.n, (long long)oy * ow + ox, cc, v[b][q]);
                        }
                }
            }
        }
        if constexpr (SUMS) {
            if (want_sums) {
#pragma unroll
                for (int b = 0; b < NB; ++b)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
#pragma unroll
                        for (int j = 0; j < 8; ++j) csum[b][q][j] += inside ? v[b][q][j] : 0.f;
            }
        }
        if (tail && inside && k.lh == 0 && k.ct == 0) {
            float rgb[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                rgb[c] = p.tail_mat[3 * c] * v[0][0][0] + p.tail_mat[3 * c + 1] * v[0][0][1] +
                         p.tail_mat[3 * c + 2] * v[0][0][2] + p.tail_bias[c];
            if (p.tail_mode == SRMI_TAIL_F32_NCHW) {
                float* o = (float*)p.tail_out + (long long)k.n * 3 * p.h * p.w + pix;
#pragma unroll
                for (int c = 0; c < 3; ++c) o[(long long)c * p.h * p.w] = rgb[c];
            } else {
                unsigned char* o = (unsigned char*)p.tail_out + ((long long)k.n * p.h * p.w + pix) * 3;
#pragma unroll
                for (int c = 0; c < 3; ++c) o[c] = to_u8(rgb[c], p.tail_lo, p.tail_hi, p.tail_scale, p.tail_round);
            }
        }
    }

    if constexpr (SUMS) {
    if (want_sums) {
        // deterministic tile sums: lanes -> wave (DPP adds inside each 32-lane half), waves -> LDS
        // (lds_red aliases the stage buffers, which nobody reads any more)
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float s = half_wave_sum(csum[b][q][j]);
                    if (k.lx == 31) lds_red[k.wave * 32 * NB + b * 32 + 16 * q + 8 * k.lh + j] = s;
                }
        __syncthreads();
        if (k.tid < 32 * NB) {
            float s = 0.f;
#pragma unroll
            for (int wv = 0; wv < kNW; wv += 4)
                s += (lds_red[wv * 32 * NB + k.tid] + lds_red[(wv + 1) * 32 * NB + k.tid]) +
                     (lds_red[(wv + 2) * 32 * NB + k.tid] + lds_red[(wv + 3) * 32 * NB + k.tid]);
            const int cq = k.ct * 32 * NB + k.tid;
            if (cq < p.cout_real) {
                if (p.chansum) p.chansum[((long long)k.n * (p.tiles_x * p.tiles_y) + k.tile) * p.cout_real + cq] = s;
                if (p.ca_sums) {
                    // 44.20 fixed point: the image total is the same whatever order the tiles arrive in.  The
                    // returned value is waited for: the add has then been performed where every other
                    // workgroup's atomics on this address are, before this workgroup reports its arrival.
                    // Neighbouring tiles finish together and atomics on one address run one after the other: tiles
                    // are spread over SRMI_CA_SLOTS accumulators per channel.
                    const unsigned long long before = __hip_atomic_fetch_add(
                        p.ca_sums + ((long long)k.n * SRMI_CA_SLOTS + (k.tile & (SRMI_CA_SLOTS - 1))) * p.cout_real + cq,
                        (unsigned long long)__float2ll_rn(s * 1048576.f), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    asm volatile("s_waitcnt vmcnt(0)" ::"v"(before) : "memory");
                }
            }
        }
        if (p.ca_sums) {
            // The workgroup that arrives last for its image finds every tile's contribution in sums[n][.]: it
            // turns them into the attention scale and leaves sums and count at zero for the next launch.
            // Only atomics carry data between workgroups here (no fence: a release at device scope would write
            // the whole L2 back): adds performed -> barrier -> arrival counted -> the last arrival exchanges the
            // totals out.
            int* flag = (int*)(lds_red + kNW * 32 * NB);
            __syncthreads();
            if (k.tid == 0) {
                const unsigned total = gridDim.x * gridDim.y;
                const unsigned old = __hip_atomic_fetch_add(p.ca_count + k.n, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *flag = old == total - 1;
            }
            __syncthreads();
            if (*flag) {
                float* mean = lds_red;                 // [cout_real]
                float* z = lds_red + p.cout_real;      // [cr]
                __syncthreads();
                for (int c = k.tid; c < p.cout_real; c += 64 * kNW) {
                    unsigned long long* slot0 = p.ca_sums + (long long)k.n * SRMI_CA_SLOTS * p.cout_real + c;
                    long long v[SRMI_CA_SLOTS];
#pragma unroll
                    for (int u = 0; u < SRMI_CA_SLOTS; ++u)
                        v[u] = (long long)__hip_atomic_exchange(slot0 + (long long)u * p.cout_real, 0ull, __ATOMIC_RELAXED,
                                                                __HIP_MEMORY_SCOPE_AGENT);
                    long long total = 0;
#pragma unroll
                    for (int u = 0; u < SRMI_CA_SLOTS; ++u) total += v[u];
                    mean[c] = (float)((double)total * (1.0 / 1048576.0)) * p.ca_inv_hw;
                }
                __syncthreads();
                for (int j = k.tid; j < p.ca_cr; j += 64 * kNW) {
                    float a = p.ca_b1[j];
                    for (int c = 0; c < p.cout_real; ++c) a += p.ca_w1[j * p.cout_real + c] * mean[c];
                    z[j] = fmaxf(a, 0.f);
                }
                __syncthreads();
                for (int c = k.tid; c < p.cout_real; c += 64 * kNW) {
                    float a = p.ca_b2[c];
                    for (int j = 0; j < p.ca_cr; ++j) a += p.ca_w2[c * p.ca_cr + j] * z[j];
                    p.ca_scale[(long long)k.n * p.cout_real + c] = 1.f / (1.f + expf(-a));
                }
                if (k.tid == 0) __hip_atomic_store(p.ca_count + k.n, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    }
}

// s_waitcnt vmcnt(n) for a wave-uniform run-time n (the instruction takes an immediate)
__device__ __forceinline__ void wait_vmcnt_le(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

}  // namespace srmi
