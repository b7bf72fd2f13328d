// Device-side helpers shared by the kernels: vector typedefs and 4-channel accessors on plane
// tensors.
#pragma once
#include "common.h"

namespace srmi {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

// f32 -> f16 with saturation: a value beyond the half range is stored as +-65504 instead of +-inf, so one
// large activation cannot turn the rest of the network into NaNs (in-range values round as a plain cast)
__device__ __forceinline__ _Float16 to_f16_sat(float v) {
    return (_Float16)__builtin_amdgcn_fmed3f(v, -65504.f, 65504.f);
}

// ---- 4-channel accessors on plane tensors (runtime dtype, wave-uniform branches) -----------------
// symmetric int8 quantisation: round to nearest even, clamp to +-127
__device__ __forceinline__ int quant_i8(float v, float qscale) {
    return (int)__builtin_amdgcn_fmed3f(rintf(v * qscale), -127.f, 127.f);
}

// 8 values -> 8 packed int8 (little endian: element 0 in the low byte of word 0)
__device__ __forceinline__ u32x2 quant8_i8(const float (&v)[8], float qscale) {
    u32x2 r;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        unsigned x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) x |= ((unsigned)quant_i8(v[4 * w + j], qscale) & 0xffu) << (8 * j);
        r[w] = x;
    }
    return r;
}

__device__ __forceinline__ char* chan_addr(const TView& t, int n, long long pix, int c) {
    char* base = t.ptr + (long long)n * t.img_stride_b;
    if (t.dtype == SRMI_F32) return base + (long long)(c >> 3) * t.plane_stride_b + pix * 32 + (c & 7) * 4;
    if (t.dtype == SRMI_I8) return base + (long long)(c >> 5) * t.plane_stride_b + pix * 32 + (c & 31);
    return base + (long long)(c >> 4) * t.plane_stride_b + pix * 32 + (c & 15) * 2;
}

__device__ __forceinline__ void load4(const TView& t, int n, long long pix, int c, float (&v)[4]) {
    const char* a = chan_addr(t, n, pix, c);
    if (t.dtype == SRMI_F32) {
        f32x4 x = *(const f32x4*)a;
        v[0] = x[0]; v[1] = x[1]; v[2] = x[2]; v[3] = x[3];
    } else if (t.dtype == SRMI_F16) {
        f16x4 x = *(const f16x4*)a;
        v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3];
    } else if (t.dtype == SRMI_I8) {
        const unsigned x = *(const unsigned*)a;
        const float s = 1.0f / t.qscale;
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (float)(int)(signed char)(x >> (8 * j)) * s;
    } else {
        bf16x4 x = *(const bf16x4*)a;
        v[0] = (float)x[0]; v[1] = (float)x[1]; v[2] = (float)x[2]; v[3] = (float)x[3];
    }
}

__device__ __forceinline__ void store4(const TView& t, int n, long long pix, int c, const float (&v)[4]) {
    char* a = chan_addr(t, n, pix, c);
    if (t.dtype == SRMI_F32) {
        f32x4 x = {v[0], v[1], v[2], v[3]};
        *(f32x4*)a = x;
    } else if (t.dtype == SRMI_F16) {
        f16x4 x = {to_f16_sat(v[0]), to_f16_sat(v[1]), to_f16_sat(v[2]), to_f16_sat(v[3])};
        *(f16x4*)a = x;
    } else if (t.dtype == SRMI_I8) {
        unsigned x = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) x |= ((unsigned)quant_i8(v[j], t.qscale) & 0xffu) << (8 * j);
        *(unsigned*)a = x;
    } else {
        bf16x4 x = {(__bf16)v[0], (__bf16)v[1], (__bf16)v[2], (__bf16)v[3]};
        *(bf16x4*)a = x;
    }
}

// ---- 8-channel accessors: channel c is a multiple of 8; 16 bytes for halves, 32 for floats ------
__device__ __forceinline__ void load8(const TView& t, int n, long long pix, int c, float (&v)[8]) {
    const char* a = chan_addr(t, n, pix, c);
    if (t.dtype == SRMI_F32) {
        const f32x4 x0 = *(const f32x4*)a, x1 = *(const f32x4*)(a + 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) { v[j] = x0[j]; v[4 + j] = x1[j]; }
    } else if (t.dtype == SRMI_F16) {
        const f16x8 x = *(const f16x8*)a;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    } else if (t.dtype == SRMI_I8) {
        const u32x2 x = *(const u32x2*)a;
        const float s = 1.0f / t.qscale;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)(int)(signed char)(x[j >> 2] >> (8 * (j & 3))) * s;
    } else {
        const bf16x8 x = *(const bf16x8*)a;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (float)x[j];
    }
}

__device__ __forceinline__ void store8(const TView& t, int n, long long pix, int c, const float (&v)[8]) {
    char* a = chan_addr(t, n, pix, c);
    if (t.dtype == SRMI_F32) {
        const f32x4 x0 = {v[0], v[1], v[2], v[3]}, x1 = {v[4], v[5], v[6], v[7]};
        *(f32x4*)a = x0;
        *(f32x4*)(a + 16) = x1;
    } else if (t.dtype == SRMI_F16) {
        f16x8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = to_f16_sat(v[j]);
        *(f16x8*)a = x;
    } else if (t.dtype == SRMI_I8) {
        *(u32x2*)a = quant8_i8(v, t.qscale);
    } else {
        bf16x8 x;
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (__bf16)v[j];
        *(bf16x8*)a = x;
    }
}

// clamp -> scale -> round to the u8 grid; separate roundings for the multiply and the +0.5 (no
// fused multiply-add), as the scalar host code of the runners computes them.
__device__ __forceinline__ unsigned char to_u8(float v, float lo, float hi, float scale, int mode) {
#pragma clang fp contract(off)
    float t = fminf(fmaxf(v, lo), hi) * scale;
    if (mode == SRMI_ROUND_HALF_EVEN) t = rintf(t);
    else if (mode == SRMI_ROUND_HALF_UP) t = floorf(t + 0.5f);
    else t = truncf(t);
    return (unsigned char)t;
}


}  // namespace srmi
