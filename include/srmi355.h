/*
 * srmi355 — C ABI of the MI355X (gfx950) super-resolution inference library.
 *
 * Plain C: pointers, sizes and PODs only.  Every function returns 0 on success and -1 on
 * failure (the convention of the reference runners, edsr/mtk/cpp/jni/src/edsr.cpp:216-219);
 * srmi_last_error() returns the message of the calling thread's last failure.
 *
 * Three layers are exported:
 *   1. operator level  (srmi_conv3x3, srmi_conv_first, srmi_ca_*, srmi_pixel_shuffle, ...):
 *      one HIP kernel launch each, device pointers in / out.  These replace the torch ops the
 *      reference modules call (nn.Conv2d + activation + residual, nn.PixelShuffle,
 *      F.interpolate(nearest), CALayer) — see the citation on each declaration.
 *   2. model level (srmi_model_*): a whole-network executor for EDSR / RCAN / RRDBNet that takes
 *      the reference's state_dict tensors by name and runs forward(img) on a HIP stream.
 *   3. runner level (include/edsr.h, rcan.h, realesrgan.h): the reference's own extern "C"
 *      runner API (<m>_init / <m>_inference / <m>_inference_ex / <m>_release) on host buffers.
 *
 * Device tensor layout ("planes"): an activation tensor of C channels is stored as
 * [N][C/CPB][H][W][CPB] with CPB = 32 bytes / sizeof(element): 16 channels per plane for
 * f16/bf16, 8 for f32.  A pixel of one plane is therefore always 32 bytes.
 * Strides in this header are in ELEMENTS of the tensor's own dtype.
 */
#ifndef SRMI355_H_
#define SRMI355_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* element types */
#define SRMI_F32 0
#define SRMI_F16 1
#define SRMI_BF16 2
#define SRMI_I8 3  /* signed 8-bit, symmetric, value = q / qscale; 32 channels per plane (w8a8 path) */

/* activations fused into the conv epilogue */
#define SRMI_ACT_NONE 0
#define SRMI_ACT_RELU 1   /* nn.ReLU            (edsr_model.py:87, rcan_model.py:128) */
#define SRMI_ACT_LRELU 2  /* nn.LeakyReLU(0.2)  (realesrgan_model.py:23,94) */

/* float -> u8 rounding of the different reference runners */
#define SRMI_ROUND_HALF_EVEN 0 /* torch .round()      (realesrgan/mtk/python/test/test_pytorch.py:73-76) */
#define SRMI_ROUND_HALF_UP 1   /* (uint8)(v + 0.5f)   (realesrgan.cpp:124-126, edsr.cpp:116-118) */
#define SRMI_ROUND_TRUNC 2     /* astype(uint8)       (edsr/mtk/python/test/test_pytorch.py:60-61) */

/* network-input encodings accepted by srmi_conv_first / srmi_model_forward */
#define SRMI_IN_F32_NCHW 0
#define SRMI_IN_U8_NHWC 1

/* RGB tail encodings produced by the last conv */
#define SRMI_TAIL_NONE 0
#define SRMI_TAIL_F32_NCHW 1
#define SRMI_TAIL_U8_NHWC 2

const char* srmi_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU); never fails. */
int srmi_device_count(void);
/* Library build info string (arch, version). */
const char* srmi_version(void);

/* ------------------------------------------------------------------------------------------ *
 * 1. Operator level
 * ------------------------------------------------------------------------------------------ */

/* A view of a plane-layout device tensor. */
typedef struct {
    void* ptr;            /* device pointer to plane 0 of image 0 (NULL = absent) */
    int64_t img_stride;   /* elements between images */
    int64_t plane_stride; /* elements between planes */
    int32_t dtype;        /* SRMI_F32 / SRMI_F16 / SRMI_BF16 / SRMI_I8 */
    int32_t flags;        /* SRMI_TENSOR_* bits, 0 = none */
    float qscale;         /* SRMI_I8 tensors: stored q = clamp(rint(value * qscale), -127, 127) */
    int32_t reserved;
} srmi_tensor_t;

/* Honoured on `out` of srmi_conv3x3: values are rounded to 8 significant bits (bfloat16's precision,
 * round to nearest even) before they are stored in the tensor's own element type.  The matrix cores
 * draw measurably less power on operands with short significands, so tensors that tolerate it (the
 * grown channels x1..x4 of a dense block, realesrgan_model.py:27-30) are kept that way. */
#define SRMI_TENSOR_ROUND8 1
#define SRMI_ALGO_DIRECT 0
#define SRMI_ALGO_WINOGRAD 1     /* F(2x2,3x3) */
#define SRMI_ALGO_WINOGRAD_X 2   /* F(2,3) along x */
#define SRMI_ALGO_SUBPIXEL_UP2 3 /* an in_up2 convolution as four 2x2 convolutions on the source grid */
#define SRMI_CA_SLOTS 8   /* accumulators per image and channel of srmi_conv_desc_t::ca */

/*
 * 3x3, stride 1, zero-padding 1 convolution with fused epilogue; replaces
 * nn.Conv2d(cin, cout, 3, padding=1) (edsr_model.py:11-15, rcan_model.py:17-18,
 * realesrgan_model.py:18-22,73-92) plus the elementwise ops that follow it in the reference
 * forward passes:
 *
 *   v   = act(conv(in) + bias)
 *   v   = v * alpha + res1          (if res1.ptr)   ResBlock / RDB / RCAB-group residuals
 *   v   = v * beta  + res2          (if res2.ptr)   RRDB outer residual (realesrgan_model.py:49)
 *   out                  = v        (if out.ptr; cast to out.dtype, f16 saturating at +-65504; a view into
 *                                    a wider tensor gives the dense block's concat; PixelShuffle(ps_r) store)
 *   out2                 = v        (if out2.ptr; second copy, usually the f32 trunk)
 *   chansum[n][tile][c]  = sum over the tile's pixels of v   (if chansum; CALayer pooling)
 *   tail: rgb = tail_mat * v[0:3] + tail_bias -> f32 NCHW, or clamp/scale/round -> u8 NHWC
 *
 * `h`,`w` are the conv's spatial size.  With in_up2 the input tensor is (h/2, w/2) and is read
 * through F.interpolate(scale_factor=2, mode='nearest') (realesrgan_model.py:110,113).
 * With ps_r > 1 the `cout` output channels are stored as nn.PixelShuffle(ps_r) would
 * (edsr_model.py:57,67): out is (h*ps_r, w*ps_r) with cout/ps_r^2 channels.
 * With dtype SRMI_I8 (the reference's w8a8 deployment, realesrgan/rknn/python/convert.py:198-278) the input tensor
 * and the packed weights are int8 (weights: integer-valued floats handed to srmi_pack_conv3x3), accumulation is
 * exact in int32, and v = acc * wscale[o] + bias[o]; outputs and residuals may be of any element type.
 * cin must be a multiple of the plane width; cout a multiple of 32, or 3 with a tail.
 * Weights come from srmi_pack_conv3x3 for the same (dtype, cout_tile, ps_r).
 */
typedef struct {
    int32_t dtype;       /* compute element type: type of `in` and of the packed weights */
    int32_t n, h, w;
    int32_t cin, cout;
    int32_t in_up2;
    int32_t act;
    int32_t ps_r;        /* 1 (none), 2 or 3 */
    int32_t cout_tile;   /* 32 or 64: output channels per workgroup (must match the packing) */
    int32_t rows_per_wave; /* 0 = library default; 2, 3, or (cout_tile 32 only) 4 */
    int32_t tail_mode;   /* SRMI_TAIL_* */
    srmi_tensor_t in;
    const void* wpk;     /* packed weights (device) */
    const float* bias;   /* packed-order bias, float[cout padded to cout_tile] (device) */
    float alpha, beta;
    srmi_tensor_t res1, res2;
    srmi_tensor_t out, out2;
    float* chansum;      /* float[n][tiles][cout] or NULL */
    void* tail_out;
    float tail_mat[9];
    float tail_bias[3];
    float tail_lo, tail_hi, tail_scale;
    int32_t tail_round;
    /* dtype SRMI_I8 (int8 operands, int32 accumulation on the matrix cores): the accumulator of output channel o
     * is multiplied by wscale[o] (= weight scale * input activation scale, packed order like `bias`) before the
     * bias is added; NULL for the float types. */
    const float* wscale;
    /* Gated input (f16 / bf16 compute): with in_gate.ptr set the convolution reads
     *   x' = in + gate_alpha * (in_gate * gate[n][c])        gate: float[n][cin] (device)
     * instead of `in`, i.e. the output of a channel-attention block (RCAB.forward, rcan_model.py:94-97; `in` = the
     * block's input, in_gate = its body before the attention scale, gate = srmi_ca_scale's result), so the block's
     * last elementwise pass over the image is folded into the convolution that consumes it (the next block's
     * first convolution, or the group's last one, rcan_model.py:104-112).  x' has the bits srmi_ca_apply would
     * store; it is also written to in_sum (if in_sum.ptr; a tensor other than in / in_gate) for the next residual.
     * `in` and in_sum are of the compute dtype, or both SRMI_F32 (a trunk kept in single precision: the convolution
     * multiplies the half-precision cast of x', in_sum receives x' itself; rows_per_wave 2 only). */
    srmi_tensor_t in_gate;
    const float* gate;
    float gate_alpha;
    /* SRMI_ALGO_DIRECT (0): implicit GEMM over the nine taps.  SRMI_ALGO_WINOGRAD: Winograd F(2x2,3x3) — the same
     * convolution with 16 instead of 36 multiplies per 2x2 output tile and input channel; f16 compute only, weights
     * from srmi_pack_wino3x3, cin a multiple of 32, cout_tile 32, no in_up2 / gated input / pixel shuffle / tail / channel attention.  The
     * input transform adds and the transformed weights are rounded to f16, so results differ from the direct form in
     * the last bits (both stay within the half-precision parity bound of the networks).
     * SRMI_ALGO_WINOGRAD_X: Winograd F(2,3) along x only (4 instead of 6 multiplies per pair of output pixels and
     * vertical tap, weights from srmi_pack_wino1d3x3), same restrictions.
     * SRMI_ALGO_SUBPIXEL_UP2: for a convolution behind a nearest-x2 upsampling (in_up2 = 1; F.interpolate(scale_factor=2,
     * mode='nearest') then conv, realesrgan_model.py:115-116).  Output pixel (2y + py, 2x + px) sees only 2 x 2 source pixels,
     * each through the sum of the taps that land on it: the layer is evaluated as four 2x2 convolutions on the source grid,
     * one per phase (py, px) — the same function with 4/9 of the multiplies.  Weights from srmi_pack_conv3x3_up2 (tap
     * sums formed in f32, rounded once to the compute type); f16 / bf16 compute, h and w even, cout a multiple of
     * cout_tile, rows_per_wave 2 or 3, no residuals / second output / gated input / pixel shuffle / tail / channel sums. */
    int32_t algo;
    srmi_tensor_t in_sum;
    /* Channel attention finished inside the convolution (CALayer, rcan_model.py:61-76): with ca.sums set, every
     * workgroup adds its tile's channel sums, as 44.20 fixed point, to sums[n][tile % SRMI_CA_SLOTS][c] (integer
     * adds: the total does not depend on their order), and the workgroup that finishes an image last computes
     *   s = sigmoid(W2 relu(W1 (sums / (h*w)) + b1) + b2)  -> scale[n][c]
     * and clears sums[n] and count[n] again.  sums (uint64[n][SRMI_CA_SLOTS][cout]) and count (uint32[n]) must be zero before
     * the launch; w1: float[cr][cout], w2: float[cout][cr] (device).  Excludes `chansum`; cout <= 512. */
    struct {
        uint64_t* sums;
        uint32_t* count;
        const float *w1, *b1, *w2, *b2;
        float* scale;
        int32_t cr;
        int32_t reserved;
    } ca;
} srmi_conv_desc_t;

int srmi_conv3x3(const srmi_conv_desc_t* d, void* hip_stream);
/* Number of spatial tiles per image srmi_conv3x3 uses for (h, w, rows_per_wave): the middle
 * extent of `chansum`. */
int srmi_conv3x3_tiles(int h, int w, int rows_per_wave);

/* Bytes of the packed form of an OIHW [cout][cin][3][3] weight; bias_floats receives the padded
 * bias length. */
size_t srmi_pack_conv3x3_bytes(int dtype, int cout, int cin, int cout_tile, int* bias_floats);
/* Host-side repack: OIHW f32 -> the device image the conv kernel stages into LDS.
 * `bias` may be NULL (zeros). Output buffers are host memory. */
int srmi_pack_conv3x3(int dtype, int cout, int cin, int cout_tile, int ps_r,
                      const float* w_oihw, const float* bias, void* wpk_out, float* bias_out);

/* Sub-pixel form of the weights of an in_up2 convolution (srmi_conv_desc_t::algo == SRMI_ALGO_SUBPIXEL_UP2): for phase
 * (py, px) and source tap (a, b) the sum of w[ky][kx] over the ky with (py + ky - 1) >> 1 == py - 1 + a (arithmetic shift)
 * and the kx alike, laid out [cout tile][stage][tap 2x2][row of the tile][32 bytes] with packed output channel
 * phase * cout + o; the bias is repeated per phase (bias_floats = 4 * cout).  cout a multiple of cout_tile. */
size_t srmi_pack_conv3x3_up2_bytes(int dtype, int cout, int cin, int cout_tile, int* bias_floats);
int srmi_pack_conv3x3_up2(int dtype, int cout, int cin, int cout_tile, const float* w_oihw, const float* bias,
                          void* wpk_out, float* bias_out);

/* Winograd form of the weights (srmi_conv_desc_t::algo == SRMI_ALGO_WINOGRAD, f16): U = G g G^T per (output, input)
 * channel pair, computed in double precision and rounded once to f16, laid out as the kernel stages it:
 * [cout tile of 32][stage of 16 input channels][transform position 16][channel half 2][output 32][8 halves].
 * cin a multiple of 16, cout a multiple of 32 (padded rows are zero).  Bias as in srmi_pack_conv3x3 (cout_tile 32). */
size_t srmi_pack_wino3x3_bytes(int cout, int cin, int* bias_floats);
int srmi_pack_wino3x3(int cout, int cin, const float* w_oihw, const float* bias, void* wpk_out, float* bias_out);
/* The same for SRMI_ALGO_WINOGRAD_X: U[ky][xi] = sum_kx G[xi][kx] g[ky][kx], laid out
 * [cout tile of 32][stage of 16 input channels][vertical tap 3][position 4][channel half 2][output 32][8 halves]. */
size_t srmi_pack_wino1d3x3_bytes(int cout, int cin, int* bias_floats);
int srmi_pack_wino1d3x3(int cout, int cin, const float* w_oihw, const float* bias, void* wpk_out, float* bias_out);

/*
 * First layer: 3-channel input -> cout channels.  Fuses the network's pre-processing
 * (MeanShift sub_mean as a general 3x3 matrix + bias, edsr_model.py:18-26,115; or the runners'
 * u8 -> float conversion, edsr.cpp:21-41, realesrgan.cpp:28-48) into the loader: the affine map
 * is applied to in-bounds pixels only, so zero padding stays zero in the shifted domain.
 *   in_kind SRMI_IN_F32_NCHW: float [n][3][h][w];  SRMI_IN_U8_NHWC: uint8 [n][h][w][3].
 * w_kcio: float [3][3][3][cout] (ky, kx, cin, cout) on the device; bias float[cout].
 */
typedef struct {
    int32_t n, h, w, cout;
    int32_t in_kind;
    int32_t reserved;
    const void* in;
    float pre_mat[9];
    float pre_bias[3];
    const float* w_kcio;
    const float* bias;
    srmi_tensor_t out, out2;
} srmi_conv_first_desc_t;

int srmi_conv_first(const srmi_conv_first_desc_t* d, void* hip_stream);

/*
 * The body of a residual channel-attention block in one kernel (RCAB.forward, rcan_model.py:79-97):
 *   x'  = in + gate_alpha * (in_gate * gate[n][c])      the previous block's output, formed while loading exactly as
 *                                                       srmi_conv3x3's gated input does; stored to in_sum if given
 *   out = conv2(relu(conv1(x') + b1)) + b2             the tensor between the two convolutions stays on the chip
 * and, with `ca` set, the channel-attention tail of srmi_conv3x3 on `out` (pooling, squeeze / excite, sigmoid ->
 * ca.scale).  The block's last line, out * scale * res_scale + x', is again left to the consumer's gated input.
 * For a block whose input is a plain tensor pass in_gate = in and gate_alpha = 0 (gate must still point at finite
 * floats).  f16 tensors, c = 64; w1 / w2 from srmi_pack_conv3x3(SRMI_F16, 64, 64, 64, 1, ...).  `out` and in_sum must not
 * alias `in` or in_gate (other workgroups still read their halos).  With a gated input, `in` and in_sum may be f32
 * tensors (the residual trunk kept in single precision, planes below 2 GiB): x' is then formed and stored in f32 and
 * the convolution sees its f16 cast, as srmi_ca_apply's two outputs would be.
 */
typedef struct {
    int32_t dtype;        /* SRMI_F16 */
    int32_t n, h, w, c;
    float gate_alpha;
    srmi_tensor_t in, in_gate, in_sum;
    const float* gate;    /* float[n][c] (device) */
    const void* w1;       /* packed weights of the first convolution (device) */
    const float* b1;
    const void* w2;
    const float* b2;
    srmi_tensor_t out;
    struct {
        uint64_t* sums;
        uint32_t* count;
        const float *w1, *b1, *w2, *b2;
        float* scale;
        int32_t cr;
        int32_t reserved;
    } ca;                 /* as srmi_conv_desc_t::ca; sums == NULL: no attention tail */
} srmi_rcab_desc_t;

int srmi_rcab_fused(const srmi_rcab_desc_t* d, void* hip_stream);

/*
 * CALayer (rcan_model.py:61-76), split at its global dependency:
 *   srmi_ca_scale : mean = sum(chansum over tiles) / (h*w); z = relu(W1 mean + b1);
 *                   s = sigmoid(W2 z + b2)                      -> scale[n][c]
 *   srmi_ca_apply : out = x + res_scale * (y * s[c])   (RCAB.forward, rcan_model.py:94-97)
 * w1: float[cr][c], w2: float[c][cr] (the 1x1 conv weights, device pointers).
 */
int srmi_ca_scale(const float* chansum, int n, int tiles, int c, int cr, int hw,
                  const float* w1, const float* b1, const float* w2, const float* b2,
                  float* scale, void* hip_stream);
int srmi_ca_apply(const srmi_tensor_t* y, const srmi_tensor_t* x, const float* scale,
                  float res_scale, int n, int h, int w, int c,
                  const srmi_tensor_t* out, const srmi_tensor_t* out2, void* hip_stream);

/* Per-(n, c) sums over H*W of a plane tensor: AdaptiveAvgPool2d(1) numerator
 * (rcan_model.py:65,74) as a standalone wavefront reduction.  sums: float[n][c]. */
int srmi_channel_sums(const srmi_tensor_t* x, int n, int h, int w, int c, float* sums,
                      void* hip_stream);

/* *slot = max(*slot, max |x|) over c channels of a plane tensor; *slot must start at a non-negative value (0).
 * The activation statistic of the int8 calibration (the "normal" max algorithm of the reference's quantiser,
 * realesrgan/rknn/python/convert.py:215-230). */
int srmi_absmax(const srmi_tensor_t* x, int n, int h, int w, int c, float* slot, void* hip_stream);

/* Standalone coalesced permutes on plane tensors (same dtype in and out):
 * nn.PixelShuffle(r) (edsr_model.py:57,67; rcan_model.py:44,51) and
 * F.interpolate(scale_factor=2, mode='nearest') (realesrgan_model.py:110,113). */
int srmi_pixel_shuffle(const srmi_tensor_t* in, const srmi_tensor_t* out, int n, int h, int w,
                       int c_in, int r, void* hip_stream);
int srmi_nearest_up2(const srmi_tensor_t* in, const srmi_tensor_t* out, int n, int h, int w,
                     int c, void* hip_stream);

/* Layout conversion between float NCHW and plane tensors (test / interop helpers). */
int srmi_nchw_to_planes(const float* in, const srmi_tensor_t* out, int n, int c, int h, int w,
                        void* hip_stream);
int srmi_planes_to_nchw(const srmi_tensor_t* in, float* out, int n, int c, int h, int w,
                        void* hip_stream);

/* Bilinear u8 HWC resize of the runners (edsr.cpp:46-94): top-left aligned, x1/y1 clamped,
 * (uint8)(val + 0.5f).  Device pointers. */
int srmi_resize_bilinear_u8(const uint8_t* in, int in_w, int in_h, uint8_t* out, int out_w,
                            int out_h, void* hip_stream);

/* ------------------------------------------------------------------------------------------ *
 * 2. Model level
 * ------------------------------------------------------------------------------------------ */

#define SRMI_ARCH_EDSR 1    /* edsr_model.py:80-124 */
#define SRMI_ARCH_RCAN 2    /* rcan_model.py:119-159 */
#define SRMI_ARCH_RRDBNET 3 /* realesrgan_model.py:60-123 */

typedef struct {
    int32_t arch;
    int32_t dtype;        /* compute dtype SRMI_F32 / SRMI_F16 / SRMI_BF16 */
    int32_t scale;        /* 2, 3, 4, 8 (RRDBNet: 2 or 4) */
    int32_t n_feats;      /* EDSR 256, RCAN 64, RRDBNet num_feat 64 */
    int32_t n_blocks;     /* EDSR n_resblocks, RCAN n_resblocks per group, RRDBNet num_block */
    int32_t n_groups;     /* RCAN n_resgroups (else 0) */
    int32_t grow_ch;      /* RRDBNet num_grow_ch (else 0) */
    int32_t reduction;    /* RCAN reduction (else 0) */
    float res_scale;      /* EDSR 0.1, RCAN 1.0 */
    int32_t with_meanshift; /* 1: EDSR / RCAN full model (sub_mean/add_mean from the state_dict);
                               0: *Core variants (edsr_model.py:146, rcan_model.py:166) */
    int32_t trunk_f32;    /* half dtypes only: keep the residual trunk in f32 */
    int32_t device;       /* HIP device ordinal */
} srmi_model_desc_t;

typedef struct srmi_model srmi_model_t;

int srmi_model_create(const srmi_model_desc_t* desc, srmi_model_t** out);
void srmi_model_destroy(srmi_model_t* m);
/* Hand over one state_dict entry (host memory, float32, contiguous, the reference's key names
 * and shapes, i.e. the module trees of edsr_model.py / rcan_model.py / realesrgan_model.py).  Unknown keys or wrong shapes fail. */
int srmi_model_set_tensor(srmi_model_t* m, const char* key, const float* data,
                          const int64_t* shape, int ndim);
/* Number of state_dict keys the architecture expects, and the i-th key. */
int srmi_model_num_keys(const srmi_model_t* m);
const char* srmi_model_key(const srmi_model_t* m, int i);
/* Pack and upload the weights.  Fails if a key is missing (strict=True semantics). */
int srmi_model_finalize(srmi_model_t* m);

/* Calls on ONE model are serialised: each call is ordered behind the previous call on that model (an
 * event wait on the caller's stream), whatever streams or host threads they come from, because they
 * share the model's workspace.  Different models are independent.  Two exceptions:
 *   - srmi_model_forward_u8_noalloc takes no part in this ordering (it waits for and records nothing): all
 *     models on a GPU share that GPU's static array, and the CALLER keeps calls that use it from overlapping
 *     (one stream, or events of its own);
 *   - a call made while `hip_stream` is being captured into a HIP graph neither waits for nor records the
 *     model's ordering event (an event recorded outside a capture cannot be waited for inside it): the graph
 *     orders the call, the caller keeps other work on the model away from the graph's replays, and the
 *     workspace must exist before the capture (srmi_model_reserve, or one eager call of the same shape) —
 *     a captured call that would have to allocate fails with a message saying so.
 * forward(x): f32 NCHW [n][3][h][w] -> f32 NCHW [n][3][h*scale][w*scale], device pointers.
 * EDSR/RCAN: 0..rgb_range domain including MeanShift; RRDBNet: [0,1], no clamp. */
int srmi_model_forward_f32(srmi_model_t* m, const float* in, float* out, int n, int h, int w,
                           void* hip_stream);
/* Runner path: u8 NHWC in -> u8 NHWC out (device pointers) with the runners' pre/post fused:
 * RRDBNet p/255 ... clamp(v,0,1)*255 (realesrgan.cpp:28-48,110-130); EDSR/RCAN p - mean*range ...
 * clamp(v + mean*range, 0, 255) (edsr.cpp:21-41,101-122). */
int srmi_model_forward_u8(srmi_model_t* m, const uint8_t* in, uint8_t* out, int n, int h, int w,
                          int round_mode, void* hip_stream);
/* Allocates, ahead of the first call, everything a later srmi_model_forward_* of up to (n, h, w) needs
 * (workspace, lane streams, events), so that the first inference does not pay for it; the runners call
 * it from <m>_init, as the reference builds its executor and staging buffers in <m>_init
 * (realesrgan.cpp:137-190). */
int srmi_model_reserve(srmi_model_t* m, int n, int h, int w);
/* Frames in HOST memory -> frames in HOST memory (u8 NHWC both sides), synchronous: the shape of the
 * reference runners' API (realesrgan.cpp:217-257: host image in, host image out).  Frame i runs on internal
 * lane i % lanes: copy into the lane's pinned staging buffer, H2D, network and D2H on the lane's
 * stream; the calling thread copies finished frames out while the GPU works on the other lanes, so
 * PCIe transfers and host copies overlap compute.  srmi_model_reserve_host allocates the staging ahead
 * of time. */
int srmi_model_forward_u8_host(srmi_model_t* m, const uint8_t* host_in, uint8_t* host_out, int n, int h, int w,
                               int round_mode);
int srmi_model_reserve_host(srmi_model_t* m, int n, int h, int w);
/*
 * srmi_model_forward_u8 for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  No counterpart in the reference.  Same inputs, outputs and
 * results as srmi_model_forward_u8, but the activations live in a static array of
 * srmi_static_workspace_bytes() (64 MiB) that is part of the library's code object on every GPU, the
 * work runs on the caller's stream alone (no internal lanes), and the call creates no stream, event or
 * allocation.  Inputs whose workspace exceeds the array (e.g. RRDBNet x4 beyond about 100x100 pixels
 * per image) fail with a message naming both sizes.  All models on one GPU share that GPU's array:
 * calls that overlap in time on one GPU (different streams or threads) race; calls on one stream, or
 * ordered by events, are safe.  The array keeps the previous call's contents: the one region that must
 * start at zero (RCAN's channel sums and arrival counters) is cleared on hip_stream at the start of
 * every call, everything else is written before it is read.
 */
int srmi_model_forward_u8_noalloc(srmi_model_t* m, const uint8_t* in, uint8_t* out, int n, int h, int w,
                                  int round_mode, void* hip_stream);
size_t srmi_static_workspace_bytes(void);
/* u8 runner path of a Core (no-MeanShift) EDSR / RCAN model: subtract shift[c] (= mean*range)
 * before the first conv and add it back after the last one (edsr.cpp:36-37,116).  No effect on
 * full models (their MeanShift tensors are used) or RRDBNet. */
int srmi_model_set_rgb_shift(srmi_model_t* m, const float* shift);
/* Bytes of device workspace currently held by the model (grows with the largest n*h*w seen). */
size_t srmi_model_workspace_bytes(const srmi_model_t* m);
/* Images a forward call sweeps through each layer together (default 1: one image per
 * layer sweep keeps its activations in the 256 MiB Infinity Cache). */
int srmi_model_set_sweep(srmi_model_t* m, int images_per_sweep);
/* RCAN with a half-precision trunk: 1 (default) lets the convolution that consumes a channel-attention block's
 * output form it while loading (srmi_conv3x3's gated input) instead of a separate srmi_ca_apply pass over the
 * image; 0 keeps the separate pass.  Both give the same bytes. */
int srmi_model_set_fused_attention(srmi_model_t* m, int on);
/* RCAN and EDSR with f16 compute and 64 features (trunk in f16, or in f32 beside it): 1 runs both convolutions of every
 * block in one kernel (srmi_rcab_fused: the tensor between them never leaves the chip), 0 as two srmi_conv3x3 launches.
 * RCAN needs the fused attention above; in EDSR the block's residual add moves into the consumer's gated input (gate
 * vector of ones).  Both forms stay within the half-precision parity bound; see srmi_model_get_fused_rcab for the default. */
int srmi_model_set_fused_rcab(srmi_model_t* m, int on);
int srmi_model_get_fused_rcab(const srmi_model_t* m);
/* RRDBNet with f16 compute: 1 (default) runs conv_up1 and conv_up2, the convolutions behind the two nearest-x2
 * upsamplings (realesrgan_model.py:115-116), in their sub-pixel form (SRMI_ALGO_SUBPIXEL_UP2: the same function with 4/9
 * of the multiplies; the summed taps are rounded once to the compute type, so results differ from the nine-tap form in
 * the last bits), 0 as nine-tap convolutions that read their input through the x2 address map.  Other compute types always
 * run the nine-tap form (srmi_conv3x3 itself offers the sub-pixel form for bf16 too). */
int srmi_model_set_subpixel_up(srmi_model_t* m, int on);
int srmi_model_get_subpixel_up(const srmi_model_t* m);
/* Internal HIP streams (1..4, default 3) over which the sweeps of one forward call are spread.
 * The lanes fork from and join the caller's stream through events, so a call is ordered on the
 * caller's stream exactly as with one lane; a call with a single sweep uses the caller's stream
 * directly.  Images are independent (realesrgan_model.py:96-123 has no cross-image op), so the
 * results do not depend on this setting. */
int srmi_model_set_lanes(srmi_model_t* m, int lanes);
/* RRDBNet with f16 compute only (default on; call before srmi_model_finalize): the grown channels
 * x1..x4 of every dense block (realesrgan_model.py:27-30) and the weight columns that multiply them
 * are rounded to 8 significant bits, the trunk x and everything else keep f16's 11.  Operands with
 * short significands cost the matrix cores less power, the chip holds a higher clock, and the
 * result stays within the half-precision parity bound (u8 diff <= 1 LSB, PSNR >= 50 dB); 0 gives
 * plain f16 everywhere. */
int srmi_model_set_growth_round8(srmi_model_t* m, int on);
/* RRDBNet with f16 compute: `on` = SRMI_ALGO_WINOGRAD (1) runs the 15 convolutions of every RRDB
 * (realesrgan_model.py:13-49) in their Winograd F(2x2,3x3) form, SRMI_ALGO_WINOGRAD_X (2) in the F(2,3)-along-x form,
 * 0 in the direct form; srmi_model_get_winograd returns the current choice (the library's default included).  Takes effect at the next forward call; both
 * forms of the weights are packed at srmi_model_finalize.  Frames of 8 rows or fewer always run the direct form.
 * Both forms meet the half-precision parity bound; the direct form is the faster one on whole layers. */
int srmi_model_set_winograd(srmi_model_t* m, int on);
int srmi_model_get_winograd(const srmi_model_t* m);
/*
 * w8a8 mode of RRDBNet (f16 models): the reference's low-precision deployment (realesrgan/rknn/python/convert.py:198-278,
 * quantized_dtype w8a8 with a calibration set of [0,1] NCHW images) on the int8 matrix cores.
 *   srmi_model_calibrate_begin(m);  srmi_model_forward_f32 / _u8 on the calibration images (any number of calls);
 *   srmi_model_calibrate_end(m)     -> per-tensor activation scales (max |x| / 127) of every dense-block tensor, int8
 *                                      weights per output channel (activation scales folded in), w8a8 switched on.
 * In w8a8 mode the 15 convolutions of every RRDB (realesrgan_model.py:13-49) read int8 activations and weights and
 * accumulate in int32; the residual trunk stays in half precision beside its quantised copy, and the layers
 * outside the blocks (conv_first, conv_body, upsampling, conv_hr, conv_last) stay f16.  srmi_model_set_w8a8 switches
 * between the two forms of a calibrated model.
 */
int srmi_model_calibrate_begin(srmi_model_t* m);
int srmi_model_calibrate_end(srmi_model_t* m);
int srmi_model_set_w8a8(srmi_model_t* m, int on);
/* 1 while the int8 form is the one that runs, else 0 (also 0 for a calibrated model switched back to f16). */
int srmi_model_get_w8a8(const srmi_model_t* m);
/* Number of calibrated tensors and their max |x| (order: per dense block, trunk input then x1..x4). */
int srmi_model_calibration(const srmi_model_t* m, float* absmax, int capacity);
/* Install activation ranges recorded earlier (srmi_model_calibration) and switch w8a8 on: how a deployed model
 * skips the calibration run, as the reference ships the already quantised .rknn file.  The _ex form installs the
 * ranges and leaves the model in the form `enable_w8a8` names (0: calibrated, but running in f16). */
int srmi_model_set_calibration(srmi_model_t* m, const float* absmax, int n);
int srmi_model_set_calibration_ex(srmi_model_t* m, const float* absmax, int n, int enable_w8a8);

/* Algorithmic conv FLOPs (2*Cin*Cout*9 per output pixel of every conv) of one forward at
 * (h, w) input size. */
double srmi_model_flops(const srmi_model_t* m, int h, int w);

#ifdef __cplusplus
}
#endif
#endif /* SRMI355_H_ */
