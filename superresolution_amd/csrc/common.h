// Internal declarations shared by the HIP sources of libsrmi355.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdarg.h>
#include <stdio.h>
#include <string>

#include "../../include/srmi355.h"

namespace srmi {

// ---- error handling ---------------------------------------------------------------------------
void set_error(const char* fmt, ...);
#define SRMI_FAIL(...)                 \
    do {                               \
        ::srmi::set_error(__VA_ARGS__); \
        return -1;                     \
    } while (0)
#define SRMI_HIP(expr)                                                                   \
    do {                                                                                 \
        hipError_t e__ = (expr);                                                         \
        if (e__ != hipSuccess) SRMI_FAIL("%s failed: %s", #expr, hipGetErrorString(e__)); \
    } while (0)

// ---- device-side tensor view --------------------------------------------------------------------
// Byte strides; a pixel of one plane is always kPixBytes bytes.
struct TView {
    char* ptr;
    long long img_stride_b;
    long long plane_stride_b;
    int dtype;
    int flags;  // SRMI_TENSOR_* bits
    float qscale;  // SRMI_I8: q = clamp(rint(value * qscale), -127, 127)
};

static inline int dtype_size(int dt) { return dt == SRMI_F32 ? 4 : dt == SRMI_I8 ? 1 : 2; }
// One plane holds kPixBytes bytes of channels per pixel (16 halves or 8 floats): one LDS-DMA stage of the
// conv kernel reads whole planes, so every fetched cache line is used in full.
constexpr int kPixBytes = 32;
static inline int dtype_cpb(int dt) { return dt == SRMI_F32 ? 8 : dt == SRMI_I8 ? 32 : 16; }  // channels per plane

static inline TView to_view(const srmi_tensor_t& t) {
    TView v;
    v.ptr = (char*)t.ptr;
    v.img_stride_b = (long long)t.img_stride * dtype_size(t.dtype);
    v.plane_stride_b = (long long)t.plane_stride * dtype_size(t.dtype);
    v.dtype = t.dtype;
    v.flags = t.flags;
    v.qscale = t.qscale;
    return v;
}

constexpr int kTileW = 32;      // pixels per MFMA column block = tile width
constexpr int kPatchW = kTileW + 2;

// ---- kernel parameter blocks --------------------------------------------------------------------
struct ConvParams {
    int n, h, w;          // conv (output) spatial size
    int ncb;              // input channel blocks (planes of the input dtype)
    int cout;             // padded to a multiple of the workgroup's cout tile
    int cout_real;
    int in_up2;
    int act;
    int ps_r, ps_c;       // pixel-shuffle factor and channels after the shuffle
    int tiles_x, tiles_y;
    int tail_mode, tail_round;
    unsigned in_plane_bytes;  // bytes of one plane of the input tensor (descriptor range)
    TView in;
    const char* wpk;
    const float* bias;
    const char* wpk2;     // fused block kernel: weights and bias of the second convolution
    const float* bias2;
    const float* wscale;  // SRMI_I8: per-output-channel dequantisation scale (packed order), else null
    // gated input: the conv reads in + gate_alpha * (in_gate * gate[n][c]) and stores it to in_sum (if set)
    TView in_gate, in_sum;
    const float* gate;
    float gate_alpha;
    float alpha, beta;
    TView res1, res2, out, out2;
    float* chansum;
    // channel attention finished by the last workgroup of an image (see srmi_conv_desc_t::ca)
    unsigned long long* ca_sums;
    unsigned* ca_count;
    const float *ca_w1, *ca_b1, *ca_w2, *ca_b2;
    float* ca_scale;
    int ca_cr;
    float ca_inv_hw;
    void* tail_out;
    float tail_mat[9];
    float tail_bias[3];
    float tail_lo, tail_hi, tail_scale;
};

}  // namespace srmi

// runners.cpp: weight-file loader and device choice shared by the runners and the executor
int srmi_pick_device(int* device);
int srmi_load_weight_file(const char* path, int want_arch, int device, srmi_model_t** out, srmi_model_desc_t* desc_out);

// conv3x3.hip: per-device kernel attributes of every instantiation (current device)
int srmi_conv3x3_prepare_device();
// rcab_fused.hip: per-device kernel attribute of the fused block kernel (current device)
int srmi_rcab_prepare_device();
// wino3x3.hip: the Winograd F(2x2,3x3) form of the convolution (srmi_conv_desc_t::algo == SRMI_ALGO_WINOGRAD)
int srmi_wino3x3_prepare_device();
int srmi_wino3x3_tile_rows();
// the direct kernel's instantiations, one translation unit per compute type (conv3x3_f16.hip, ...)
int srmi_conv3x3_launch_f16(const srmi::ConvParams& p, int mb, int nb, hipStream_t s);
int srmi_conv3x3_launch_bf16(const srmi::ConvParams& p, int mb, int nb, hipStream_t s);
int srmi_conv3x3_launch_f32(const srmi::ConvParams& p, int mb, int nb, hipStream_t s);
int srmi_conv3x3_launch_i8(const srmi::ConvParams& p, int mb, int nb, hipStream_t s);
int srmi_conv3x3_configure_f16();
int srmi_conv3x3_configure_bf16();
int srmi_conv3x3_configure_f32();
int srmi_conv3x3_configure_i8();
// the sub-pixel form of an upsampling convolution (algo == SRMI_ALGO_SUBPIXEL_UP2; conv3x3_up2_{f16,bf16}.hip)
int srmi_conv3x3_launch_up2_f16(const srmi::ConvParams& p, int mb, int nb, hipStream_t s);
int srmi_conv3x3_launch_up2_bf16(const srmi::ConvParams& p, int mb, int nb, hipStream_t s);
int srmi_conv3x3_configure_up2_f16();
int srmi_conv3x3_configure_up2_bf16();
int srmi_wino3x3_launch(const srmi::ConvParams& p, hipStream_t stream);
// conv_rgb.hip: the RGB layer (3 outputs) as one GEMM over (tap, output) columns and a nine-tap gather
int srmi_conv_rgb_prepare_device();
bool srmi_conv_rgb_takes(const srmi_conv_desc_t* d);
int srmi_conv_rgb_launch(const srmi::ConvParams& p, int cout_tile, hipStream_t stream);
// wino1d3x3.hip: the Winograd F(2,3)-along-x form (algo == SRMI_ALGO_WINOGRAD_X)
int srmi_wino1d3x3_prepare_device();
void srmi_wino1d3x3_tile(int* rows, int* cols);
int srmi_wino1d3x3_launch(const srmi::ConvParams& p, hipStream_t stream);
