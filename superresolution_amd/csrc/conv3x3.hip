// Host entry point of the direct 3x3 convolution (srmi_conv3x3): validation, tiling, dispatch to the kernel instantiations of the
// descriptor's compute type (conv3x3_kernel.h, one translation unit per type), to the Winograd kernels or, for the RGB layer,
// to conv_rgb.hip.
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "conv_epilogue.h"   // kNW

namespace srmi {

static int default_rows_per_wave() { return 3; }  // 12-row tiles: the fastest variant on every layer measured

}  // namespace srmi

using namespace srmi;

// Sets the kernel attributes of every instantiation on the current device, so that later launches on it
// (for instance under stream capture) consist of the launch alone.
int srmi_conv3x3_prepare_device() {
    return (srmi_conv3x3_configure_f16() || srmi_conv3x3_configure_bf16() || srmi_conv3x3_configure_f32() || srmi_conv3x3_configure_i8() ||
            srmi_conv3x3_configure_up2_f16() || srmi_conv3x3_configure_up2_bf16() ||
            srmi_wino3x3_prepare_device() || srmi_wino1d3x3_prepare_device() || srmi_conv_rgb_prepare_device()) ? -1 : 0;
}

extern "C" int srmi_conv3x3_tiles(int h, int w, int rows_per_wave) {
    const int mb = rows_per_wave ? rows_per_wave : default_rows_per_wave();
    const int th = kNW * mb;
    return ((w + kTileW - 1) / kTileW) * ((h + th - 1) / th);
}

extern "C" int srmi_conv3x3(const srmi_conv_desc_t* d, void* hip_stream) {
    if (!d) SRMI_FAIL("srmi_conv3x3: null descriptor");
    if (d->dtype != SRMI_F32 && d->dtype != SRMI_F16 && d->dtype != SRMI_BF16 && d->dtype != SRMI_I8)
        SRMI_FAIL("srmi_conv3x3: bad dtype %d", d->dtype);
    if (d->dtype == SRMI_I8) {
        if (!d->wscale) SRMI_FAIL("srmi_conv3x3: int8 compute needs wscale");
        if (d->tail_mode != SRMI_TAIL_NONE || d->ps_r > 1 || d->chansum || d->cout_tile != 32)
            SRMI_FAIL("srmi_conv3x3: int8 compute supports plain convolutions with cout_tile 32 only");
        if (d->out.ptr && d->out.dtype == SRMI_I8 && !(d->out.qscale > 0.f)) SRMI_FAIL("srmi_conv3x3: int8 output needs a positive qscale");
    }
    const int cpb = dtype_cpb(d->dtype);
    if (d->n <= 0 || d->h <= 0 || d->w <= 0) SRMI_FAIL("srmi_conv3x3: bad shape n=%d h=%d w=%d", d->n, d->h, d->w);
    if (d->cin <= 0 || d->cin % cpb) SRMI_FAIL("srmi_conv3x3: cin=%d is not a multiple of %d", d->cin, cpb);
    const bool gated_f32 = d->in_gate.ptr && d->in.ptr && d->in.dtype == SRMI_F32 && d->dtype != SRMI_F32;
    if (!d->in.ptr || (d->in.dtype != d->dtype && !gated_f32)) SRMI_FAIL("srmi_conv3x3: input tensor missing or of another dtype");
    if (!d->wpk || !d->bias) SRMI_FAIL("srmi_conv3x3: missing weights or bias");
    if (d->cout_tile != 32 && d->cout_tile != 64) SRMI_FAIL("srmi_conv3x3: cout_tile must be 32 or 64");
    const bool wino_x = d->algo == SRMI_ALGO_WINOGRAD_X;
    const bool wino = d->algo == SRMI_ALGO_WINOGRAD || wino_x;
    const bool subpix = d->algo == SRMI_ALGO_SUBPIXEL_UP2;
    if (d->algo != SRMI_ALGO_DIRECT && !wino && !subpix) SRMI_FAIL("srmi_conv3x3: unknown algo %d", d->algo);
    if (subpix) {
        if (d->dtype != SRMI_F16 && d->dtype != SRMI_BF16) SRMI_FAIL("srmi_conv3x3: the sub-pixel form computes in f16 or bf16");
        if (!d->in_up2) SRMI_FAIL("srmi_conv3x3: the sub-pixel form is the form of an in_up2 convolution");
        if (d->ps_r > 1 || d->res1.ptr || d->res2.ptr || d->out2.ptr || d->in_gate.ptr || d->tail_mode != SRMI_TAIL_NONE ||
            d->ca.sums || d->chansum || !d->out.ptr)
            SRMI_FAIL("srmi_conv3x3: the sub-pixel form writes one plane tensor and excludes residuals, a second output, gated "
                      "input, pixel shuffle, tail, channel sums and channel attention");
        if (d->cout <= 0 || d->cout % d->cout_tile) SRMI_FAIL("srmi_conv3x3: cout=%d is not a multiple of cout_tile=%d", d->cout, d->cout_tile);
    }
    if (wino) {
        if (d->dtype != SRMI_F16) SRMI_FAIL("srmi_conv3x3: the Winograd form computes in f16 only");
        if (d->cin % 32) SRMI_FAIL("srmi_conv3x3: the Winograd form needs cin to be a multiple of 32 (got %d)", d->cin);
        if (d->cout_tile != 32 || d->in_up2 || d->in_gate.ptr || d->ps_r > 1 || d->tail_mode != SRMI_TAIL_NONE || d->ca.sums || d->chansum)
            SRMI_FAIL("srmi_conv3x3: the Winograd form needs cout_tile 32 and excludes in_up2, gated input, pixel shuffle, tail, "
                      "channel sums and channel attention");
    }
    // the Winograd kernel has one tile shape: 16 rows (4 per wave)
    const int mb = wino_x ? 2 : wino ? srmi_wino3x3_tile_rows() / kNW : (d->rows_per_wave ? d->rows_per_wave : default_rows_per_wave());
    if (mb < 2 || mb > 4) SRMI_FAIL("srmi_conv3x3: rows_per_wave must be 2, 3 or 4");
    if (subpix && mb == 4) SRMI_FAIL("srmi_conv3x3: the sub-pixel form needs rows_per_wave 2 or 3");
    if (d->cout_tile == 64 && mb == 4) SRMI_FAIL("srmi_conv3x3: rows_per_wave 4 needs cout_tile 32");
    const bool tail = d->tail_mode != SRMI_TAIL_NONE;
    if (tail) {
        if (d->cout != 3 || !d->tail_out) SRMI_FAIL("srmi_conv3x3: an RGB tail needs cout == 3 and tail_out");
        if (d->out.ptr || d->out2.ptr || d->chansum || d->ps_r > 1)
            SRMI_FAIL("srmi_conv3x3: an RGB tail excludes plane outputs");
    } else {
        if (d->cout <= 0 || d->cout % d->cout_tile) SRMI_FAIL("srmi_conv3x3: cout=%d is not a multiple of cout_tile=%d", d->cout, d->cout_tile);
        if (!d->out.ptr && !d->out2.ptr) SRMI_FAIL("srmi_conv3x3: no output tensor");
    }
    const int r = d->ps_r <= 1 ? 1 : d->ps_r;
    if (r > 3) SRMI_FAIL("srmi_conv3x3: ps_r must be 1, 2 or 3");
    if (r > 1 && (d->cout % (r * r * 32) || d->out2.ptr || d->res1.ptr || d->res2.ptr))
        SRMI_FAIL("srmi_conv3x3: pixel-shuffle store needs cout %% (32*r*r) == 0 and no residual/out2");
    if (d->in_up2 && ((d->h | d->w) & 1)) SRMI_FAIL("srmi_conv3x3: in_up2 needs even h and w");
    if (d->in_gate.ptr) {
        if (d->dtype != SRMI_F16 && d->dtype != SRMI_BF16) SRMI_FAIL("srmi_conv3x3: a gated input needs f16 or bf16 compute");
        if (d->in_gate.dtype != d->dtype || !d->gate) SRMI_FAIL("srmi_conv3x3: in_gate must be of the compute dtype and come with a gate vector");
        if (d->in_up2 || d->cin > 256 || mb > 3) SRMI_FAIL("srmi_conv3x3: a gated input excludes in_up2, cin > 256 and rows_per_wave above 3");
        if (d->in_sum.ptr && (d->in_sum.dtype != d->in.dtype || d->in_sum.ptr == d->in.ptr || d->in_sum.ptr == d->in_gate.ptr))
            SRMI_FAIL("srmi_conv3x3: in_sum must be a separate tensor of the type of `in`");
        if (gated_f32 && (mb != 2 || (long long)d->h * d->w * kPixBytes * 2 >= (1ll << 32)))
            SRMI_FAIL("srmi_conv3x3: a gated f32 input needs rows_per_wave 2 and planes below 2 GiB");
    } else if (d->in_sum.ptr || d->gate) {
        SRMI_FAIL("srmi_conv3x3: in_sum / gate without in_gate");
    }
    if (d->ca.sums) {
        if (!d->ca.count || !d->ca.w1 || !d->ca.b1 || !d->ca.w2 || !d->ca.b2 || !d->ca.scale || d->ca.cr <= 0)
            SRMI_FAIL("srmi_conv3x3: incomplete channel-attention block");
        if (d->chansum || tail || r > 1 || d->cout > 512 || d->ca.cr > 512 || d->dtype == SRMI_I8)
            SRMI_FAIL("srmi_conv3x3: channel attention excludes chansum, tail, pixel shuffle, int8 and cout > 512");
    }
    if ((long long)d->h * d->w * kPixBytes * r * r >= (1ll << 32)) SRMI_FAIL("srmi_conv3x3: plane larger than 4 GiB");

    ConvParams p;
    p.n = d->n; p.h = d->h; p.w = d->w;
    p.ncb = d->cin / cpb;
    p.cout_real = d->cout;
    p.cout = (d->cout + d->cout_tile - 1) / d->cout_tile * d->cout_tile;
    p.in_up2 = d->in_up2 ? 1 : 0;
    p.act = d->act;
    p.ps_r = r;
    p.ps_c = d->cout / (r * r);
    p.tiles_x = (d->w + kTileW - 1) / kTileW;
    p.tiles_y = (d->h + kNW * mb - 1) / (kNW * mb);
    p.tail_mode = d->tail_mode; p.tail_round = d->tail_round;
    p.in = to_view(d->in);
    {
        const long long src_px = d->in_up2 ? (long long)(d->h / 2) * (d->w / 2) : (long long)d->h * d->w;
        p.in_plane_bytes = (unsigned)(src_px * kPixBytes);
    }
    p.wpk = (const char*)d->wpk; p.bias = d->bias;
    p.wscale = d->wscale;
    p.in_gate = to_view(d->in_gate); p.in_sum = to_view(d->in_sum);
    p.gate = d->gate; p.gate_alpha = d->gate_alpha;
    p.alpha = d->alpha; p.beta = d->beta;
    p.res1 = to_view(d->res1); p.res2 = to_view(d->res2);
    p.out = to_view(d->out); p.out2 = to_view(d->out2);
    p.chansum = d->chansum;
    p.ca_sums = (unsigned long long*)d->ca.sums; p.ca_count = d->ca.count;
    p.ca_w1 = d->ca.w1; p.ca_b1 = d->ca.b1; p.ca_w2 = d->ca.w2; p.ca_b2 = d->ca.b2;
    p.ca_scale = d->ca.scale; p.ca_cr = d->ca.cr;
    p.ca_inv_hw = 1.f / ((float)d->h * (float)d->w);
    p.tail_out = d->tail_out;
#ifdef SRMI_DIAG
    // timing ablations (wrong results): every input plane aliases plane 0 / every output plane aliases plane 0,
    // which removes most of the memory traffic and leaves the matrix work as it is
    if (getenv("SRMI_ABL_ALIAS_IN")) p.in.plane_stride_b = 0;
    if (getenv("SRMI_ABL_ALIAS_OUT")) { p.out.plane_stride_b = 0; p.out2.plane_stride_b = 0; }
#endif
#if defined(SRMI_DIAG) && (defined(SRMI_STAMPS) || defined(SRMI_CLOCK))
    if (d->tail_mode == SRMI_TAIL_NONE) {
        p.tail_out = getenv("SRMI_STAMP_PTR") ? (void*)strtoull(getenv("SRMI_STAMP_PTR"), nullptr, 0) : nullptr;
        // optional filter: only launches with this many input channels / outputs record
        if (getenv("SRMI_STAMP_CIN") && atoi(getenv("SRMI_STAMP_CIN")) != d->cin) p.tail_out = nullptr;
        if (getenv("SRMI_STAMP_COUT") && atoi(getenv("SRMI_STAMP_COUT")) != d->cout) p.tail_out = nullptr;
    }
#endif
    for (int i = 0; i < 9; ++i) p.tail_mat[i] = d->tail_mat[i];
    for (int i = 0; i < 3; ++i) p.tail_bias[i] = d->tail_bias[i];
    p.tail_lo = d->tail_lo; p.tail_hi = d->tail_hi; p.tail_scale = d->tail_scale;

    hipStream_t s = (hipStream_t)hip_stream;
    if (wino_x) {
        int th = 0, tw = 0;
        srmi_wino1d3x3_tile(&th, &tw);
        p.tiles_x = (d->w + tw - 1) / tw;
        p.tiles_y = (d->h + th - 1) / th;
        return srmi_wino1d3x3_launch(p, s);
    }
    if (wino) return srmi_wino3x3_launch(p, s);
    if (subpix) {
        // the kernel runs on the source grid: four phases of cout outputs each, stored through the pixel-shuffle map
        p.h = d->h / 2; p.w = d->w / 2;
        p.in_up2 = 0;
        p.cout = p.cout_real = 4 * d->cout;
        p.ps_r = 2; p.ps_c = d->cout;
        p.tiles_x = (p.w + kTileW - 1) / kTileW;
        p.tiles_y = (p.h + kNW * mb - 1) / (kNW * mb);
        const int nb2 = d->cout_tile / 32;
        return d->dtype == SRMI_F16 ? srmi_conv3x3_launch_up2_f16(p, mb, nb2, s) : srmi_conv3x3_launch_up2_bf16(p, mb, nb2, s);
    }
    // SRMI_RGB_GATHER=0 (diagnostic) keeps the RGB layer on the direct kernel
    static const bool rgb_gather = !(getenv("SRMI_RGB_GATHER") && atoi(getenv("SRMI_RGB_GATHER")) == 0);
    if (rgb_gather && srmi_conv_rgb_takes(d)) return srmi_conv_rgb_launch(p, d->cout_tile, s);
    const int nb = d->cout_tile / 32;
    if (d->dtype == SRMI_F16) return srmi_conv3x3_launch_f16(p, mb, nb, s);
    if (d->dtype == SRMI_BF16) return srmi_conv3x3_launch_bf16(p, mb, nb, s);
    if (d->dtype == SRMI_I8) return srmi_conv3x3_launch_i8(p, mb, nb, s);
    return srmi_conv3x3_launch_f32(p, mb, nb, s);
}
