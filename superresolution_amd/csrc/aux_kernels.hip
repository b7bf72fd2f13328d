// HBM-bound kernels around the convolutions: the 3-channel first layer with fused
// pre-processing, channel attention (pooling finish + scale/residual apply), standalone
// PixelShuffle / nearest-x2 permutes, float-NCHW <-> plane layout conversion and the runners'
// bilinear u8 resize.  All of them move 8-16 bytes per lane with consecutive lanes on
// consecutive addresses.
#include "common.h"
#include "devutil.h"

namespace srmi {

// ------------------------------------------------------------------------------------------------
// First layer (cin = 3): one thread = one pixel x 16 output channels.
// ------------------------------------------------------------------------------------------------
struct FirstParams {
    int n, h, w, cout, in_kind;
    const void* in;
    float pre_mat[9];
    float pre_bias[3];
    const float* wgt;   // [27][cout]
    const float* bias;  // [cout]
    TView out, out2;
};

__device__ __forceinline__ float pre_affine(const float* m, float b, const float (&raw)[3]) {
#pragma clang fp contract(off)
    // every product and sum individually rounded, in the order a 1x1 conv / the runners' scalar
    // code evaluates them
    return ((m[0] * raw[0] + m[1] * raw[1]) + m[2] * raw[2]) + b;
}

template <int IN_KIND>
__global__ __launch_bounds__(256) void conv_first_kernel(const FirstParams p) {
    // one thread = one pixel x 16 output channels (one half plane = 32 bytes of halves).  The channel
    // group is uniform over the workgroup, so the weights come through scalar loads; the 27 inputs
    // of the pixel are converted once and reused for the 16 outputs.  (8, 32 and 64 outputs per thread
    // are all slower: the kernel lives on having many waves to hide its scalar weight fetches.)
    const long long hw = (long long)p.h * p.w;
    const long long pixn = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pixn >= (long long)p.n * hw) return;
    const int cg = blockIdx.y;
    const int n = (int)(pixn / hw);
    const long long pix = pixn - (long long)n * hw;
    const int y = (int)(pix / p.w), x = (int)(pix - (long long)y * p.w);

    float v[27];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            // no branch per tap: a tap outside the image loads the nearest pixel inside it (always a valid
            // address) and is zeroed afterwards, so that all loads of a thread are in flight together
            const int yy = y + ky - 1, xx = x + kx - 1;
            const unsigned keep = (yy >= 0 && yy < p.h && xx >= 0 && xx < p.w) ? 0xffffffffu : 0u;
            const int yc = min(max(yy, 0), p.h - 1), xc = min(max(xx, 0), p.w - 1);
            float* vt = v + (ky * 3 + kx) * 3;
            float raw[3];
            if (IN_KIND == SRMI_IN_U8_NHWC) {
                const unsigned char* s = (const unsigned char*)p.in + ((long long)n * hw + (long long)yc * p.w + xc) * 3;
                raw[0] = (float)s[0]; raw[1] = (float)s[1]; raw[2] = (float)s[2];
            } else {
                const float* s = (const float*)p.in + (long long)n * 3 * hw + (long long)yc * p.w + xc;
                raw[0] = s[0]; raw[1] = s[hw]; raw[2] = s[2 * hw];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c)  // zeroed through a bit mask: a select here comes back as a branch per tap
                vt[c] = __uint_as_float(__float_as_uint(pre_affine(p.pre_mat + 3 * c, p.pre_bias[c], raw)) & keep);
        }
    }
    float acc[16];
    const float* bias = p.bias + cg * 16;
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = bias[j];
    const float* wg = p.wgt + cg * 16;
#pragma unroll
    for (int t = 0; t < 27; ++t) {
#pragma unroll
        for (int j = 0; j < 16; ++j) acc[j] += wg[(long long)t * p.cout + j] * v[t];
    }
    const float o0[8] = {acc[0], acc[1], acc[2], acc[3], acc[4], acc[5], acc[6], acc[7]};
    const float o1[8] = {acc[8], acc[9], acc[10], acc[11], acc[12], acc[13], acc[14], acc[15]};
    if (p.out.ptr) { store8(p.out, n, pix, cg * 16, o0); store8(p.out, n, pix, cg * 16 + 8, o1); }
    if (p.out2.ptr) { store8(p.out2, n, pix, cg * 16, o0); store8(p.out2, n, pix, cg * 16 + 8, o1); }
}

// ------------------------------------------------------------------------------------------------
// Channel attention
// ------------------------------------------------------------------------------------------------
// One workgroup per image: finish the pooling (fixed summation order -> bitwise reproducible),
// then the squeeze / excite 1x1 convolutions and the sigmoid.  The tile partials are summed in
// kCaParts interleaved groups per channel with sixteen independent loads in flight per thread:
// the kernel's time is the latency of that chain of global loads.
constexpr int kCaParts = 16;
constexpr int kCaThreads = 1024;
__global__ __launch_bounds__(kCaThreads) void ca_scale_kernel(const float* chansum, int tiles, int c, int cr,
                                                              float inv_hw, const float* w1, const float* b1,
                                                              const float* w2, const float* b2, float* scale) {
    extern __shared__ float sh[];  // [kCaParts][c] partials, [c] mean, [cr] z
    float* part = sh;
    float* mean = sh + kCaParts * c;
    float* z = mean + c;
    const int n = blockIdx.x;
    const float* src = chansum + (long long)n * tiles * c;
    for (int i = threadIdx.x; i < kCaParts * c; i += kCaThreads) {
        const int ch = i % c, q = i / c;
        float s = 0.f;
        for (int t = q; t < tiles; t += 16 * kCaParts) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int tt = t + u * kCaParts;
                v[u] = tt < tiles ? src[(long long)tt * c + ch] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u];
        }
        part[q * c + ch] = s;
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += kCaThreads) {
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < kCaParts; ++q) s += part[q * c + ch];
        mean[ch] = s * inv_hw;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cr; j += kCaThreads) {
        float s = b1[j];
        for (int ch = 0; ch < c; ++ch) s += w1[j * c + ch] * mean[ch];
        z[j] = fmaxf(s, 0.f);
    }
    __syncthreads();
    for (int ch = threadIdx.x; ch < c; ch += kCaThreads) {
        float s = b2[ch];
        for (int j = 0; j < cr; ++j) s += w2[ch * cr + j] * z[j];
        scale[(long long)n * c + ch] = 1.f / (1.f + expf(-s));
    }
}

// out = x + res_scale * (y * s[c]); one thread = one pixel x 8 channels (16-byte accesses on half
// tensors).  Consecutive threads walk the pixels of one plane half, so a wave reads and writes
// contiguous runs.
__global__ __launch_bounds__(256) void ca_apply_kernel(TView y, TView x, const float* scale, float res_scale,
                                                       int n_img, long long hw, int c, TView out, TView out2) {
    const int groups = c >> 3;
    const long long total = (long long)n_img * hw * groups;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        // index order: (n, group pair, pixel, group within the pair): the two 8-channel groups of a
        // 16-channel half plane sit next to each other in memory
        const int sub = (int)(idx & 1);
        long long t = idx >> 1;
        const long long pix = t % hw;
        t /= hw;
        const int pairs = groups >> 1;
        const int g = (int)(t % pairs) * 2 + sub;
        const int n = (int)(t / pairs);
        const int ch = g * 8;
        float yv[8], xv[8], o[8];
        load8(y, n, pix, ch, yv);
        load8(x, n, pix, ch, xv);
        const f32x4 s0 = *(const f32x4*)(scale + (long long)n * c + ch);
        const f32x4 s1 = *(const f32x4*)(scale + (long long)n * c + ch + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o[j] = (yv[j] * s0[j]) * res_scale + xv[j];
            o[4 + j] = (yv[4 + j] * s1[j]) * res_scale + xv[4 + j];
        }
        if (out.ptr) store8(out, n, pix, ch, o);
        if (out2.ptr) store8(out2, n, pix, ch, o);
    }
}

// Per-(n, c) sums over all pixels.  One workgroup = one 16-channel span x a segment of pixels; a thread
// reads 8 channels of a pixel (16 bytes on half tensors), so a wave reads contiguous runs.  Threads
// with the same 8-channel half are reduced with wavefront shuffles, the four waves through LDS, and
// each workgroup ends with one float atomic per channel.
__global__ __launch_bounds__(256) void channel_sums_kernel(TView x, long long hw, int c, float* sums, int seg) {
    const int n = blockIdx.z;
    const int sub = threadIdx.x & 1;
    const int ch = blockIdx.y * 16 + sub * 8;
    const long long p_begin = (long long)blockIdx.x * seg;
    const long long p_end = p_begin + seg < hw ? p_begin + seg : hw;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (ch < c) {
        // four independent 16-byte loads in flight per thread (the sum is over f32 values of a fixed set per thread,
        // added in a fixed order: reproducible)
        long long pix = p_begin + (threadIdx.x >> 1);
        for (; pix + 384 < p_end; pix += 512) {
            float v[4][8];
#pragma unroll
            for (int u = 0; u < 4; ++u) load8(x, n, pix + 128 * u, ch, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += v[u][j];
        }
        for (; pix < p_end; pix += 128) {
            float v[8];
            load8(x, n, pix, ch, v);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += v[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int m = 2; m < 64; m <<= 1) acc[j] += __shfl_xor(acc[j], m);
    }
    __shared__ float red[4][2][8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < 2) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave][lane][j] = acc[j];
    }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int sb = threadIdx.x >> 3, j = threadIdx.x & 7;
        const int cc = blockIdx.y * 16 + sb * 8 + j;
        if (cc < c) {
            const float s = (red[0][sb][j] + red[1][sb][j]) + (red[2][sb][j] + red[3][sb][j]);
            atomicAdd(sums + (long long)n * c + cc, s);
        }
    }
}

// max |x| over c channels of a plane tensor (calibration of the int8 path).  Non-negative floats order like their
// bit patterns, so one integer atomic max per workgroup folds the result into *slot (which accumulates over calls).
__global__ __launch_bounds__(256) void absmax_kernel(TView x, int n_img, long long hw, int c, unsigned* slot) {
    const int groups = c >> 3;
    const long long total = (long long)n_img * hw * groups;
    float m = 0.f;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int sub = (int)(idx & 1);
        long long t = idx >> 1;
        const long long pix = t % hw;
        t /= hw;
        const int pairs = groups >> 1;
        const int g = (int)(t % pairs) * 2 + sub;
        const int n = (int)(t / pairs);
        float v[8];
        load8(x, n, pix, g * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j) m = fmaxf(m, fabsf(v[j]));
    }
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float red[4];
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
        if (m == m) atomicMax(slot, __float_as_uint(m));
    }
}

// ------------------------------------------------------------------------------------------------
// Permutes.  One thread = one output pixel x one 16-byte chunk.
// ------------------------------------------------------------------------------------------------
template <typename E>
__global__ __launch_bounds__(256) void pixel_shuffle_kernel(TView in, TView out, int n_img, int h, int w,
                                                            int c_out, int r) {
    constexpr int EPC = 16 / (int)sizeof(E);      // elements per chunk
    constexpr int CPB = kPixBytes / (int)sizeof(E);
    const int chunks = c_out / EPC;
    const int oh = h * r, ow = w * r;
    const long long total = (long long)n_img * oh * ow * chunks;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ck = (int)(idx % chunks);
        long long t = idx / chunks;
        const int ox = (int)(t % ow); t /= ow;
        const int oy = (int)(t % oh);
        const int n = (int)(t / oh);
        const int y = oy / r, x = ox / r, ij = (oy - y * r) * r + (ox - x * r);
        const char* ibase = in.ptr + (long long)n * in.img_stride_b + ((long long)y * w + x) * kPixBytes;
        __attribute__((aligned(16))) E v[EPC];
#pragma unroll
        for (int e = 0; e < EPC; ++e) {
            const int ci = (ck * EPC + e) * r * r + ij;
            v[e] = *(const E*)(ibase + (long long)(ci / CPB) * in.plane_stride_b + (ci % CPB) * sizeof(E));
        }
        const int co = ck * EPC;
        char* o = out.ptr + (long long)n * out.img_stride_b + (long long)(co / CPB) * out.plane_stride_b +
                  ((long long)oy * ow + ox) * kPixBytes + (co % CPB) * sizeof(E);
        *(u32x4*)o = *(const u32x4*)v;
    }
}

// PixelShuffle(2) on 2-byte elements with 16-byte accesses only: one thread = one INPUT pixel x 64 input channels (four
// planes = eight 16-byte chunks) = one whole 16-channel output plane of the four output pixels (2y + i, 2x + j): input
// channel 4 * co + (2 i + j) -> output channel co (edsr_model.py:57; rcan_model.py:44).  The halves are regrouped in
// registers; a thread writes 64 contiguous bytes per output row, so the stores of a wave cover whole cache lines.
__global__ __launch_bounds__(256) void pixel_shuffle2_h16_kernel(TView in, TView out, int n_img, int h, int w, int c_out) {
    const int oplanes = c_out >> 4;
    const long long hw = (long long)h * w;
    const long long total = (long long)n_img * oplanes * hw;
    const int ow = 2 * w;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const long long pix = idx % hw;
        long long t = idx / hw;
        const int g = (int)(t % oplanes);
        const int n = (int)(t / oplanes);
        const int y = (int)(pix / w), x = (int)(pix - (long long)y * w);
        const char* ib = in.ptr + (long long)n * in.img_stride_b + (long long)(4 * g) * in.plane_stride_b + pix * kPixBytes;
        unsigned short v[64];
#pragma unroll
        for (int pl = 0; pl < 4; ++pl) {
            *(u32x4*)(v + 16 * pl) = *(const u32x4*)(ib + (long long)pl * in.plane_stride_b);
            *(u32x4*)(v + 16 * pl + 8) = *(const u32x4*)(ib + (long long)pl * in.plane_stride_b + 16);
        }
        char* ob = out.ptr + (long long)n * out.img_stride_b + (long long)g * out.plane_stride_b;
#pragma unroll
        for (int ij = 0; ij < 4; ++ij) {
            __attribute__((aligned(16))) unsigned short o[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) o[e] = v[4 * e + ij];
            char* d = ob + ((long long)(2 * y + (ij >> 1)) * ow + 2 * x + (ij & 1)) * kPixBytes;
            *(u32x4*)d = *(const u32x4*)o;
            *(u32x4*)(d + 16) = *(const u32x4*)(o + 8);
        }
    }
}

__global__ __launch_bounds__(256) void nearest_up2_kernel(TView in, TView out, int n_img, int h, int w, int planes) {
    // one thread = one 16-byte chunk of an input pixel -> the same chunk of its four output pixels: every input byte is
    // read once (the previous form read it four times, through the cache)
    const int ow = 2 * w;
    const long long total = (long long)n_img * planes * h * w * 2;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int ck = (int)(idx & 1);
        long long t = idx >> 1;
        const int x = (int)(t % w); t /= w;
        const int y = (int)(t % h); t /= h;
        const int pl = (int)(t % planes);
        const int n = (int)(t / planes);
        const u32x4 v = *(const u32x4*)(in.ptr + (long long)n * in.img_stride_b + (long long)pl * in.plane_stride_b +
                                        ((long long)y * w + x) * kPixBytes + ck * 16);
        char* d = out.ptr + (long long)n * out.img_stride_b + (long long)pl * out.plane_stride_b +
                  ((long long)(2 * y) * ow + 2 * x) * kPixBytes + ck * 16;
        *(u32x4*)d = v;
        *(u32x4*)(d + kPixBytes) = v;
        *(u32x4*)(d + (long long)ow * kPixBytes) = v;
        *(u32x4*)(d + (long long)ow * kPixBytes + kPixBytes) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// float NCHW <-> planes (any plane dtype).  One thread = one pixel x 8 channels (16 bytes of a half
// plane); blockIdx.y = (image, channel group), so no index is divided.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nchw_to_planes_kernel(const float* in, TView out, int c, int groups, long long hw) {
    const int n = blockIdx.y / groups, g = blockIdx.y - n * groups;
    const float* src = in + ((long long)n * c + g * 8) * hw;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < hw; pix += (long long)gridDim.x * 256) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = g * 8 + j < c ? src[j * hw + pix] : 0.f;
        store8(out, n, pix, g * 8, v);
    }
}

__global__ __launch_bounds__(256) void planes_to_nchw_kernel(TView in, float* out, int c, int groups, long long hw) {
    const int n = blockIdx.y / groups, g = blockIdx.y - n * groups;
    float* dst = out + ((long long)n * c + g * 8) * hw;
    for (long long pix = (long long)blockIdx.x * 256 + threadIdx.x; pix < hw; pix += (long long)gridDim.x * 256) {
        float v[8];
        load8(in, n, pix, g * 8, v);
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (g * 8 + j < c) dst[j * hw + pix] = v[j];
    }
}

// ------------------------------------------------------------------------------------------------
// Bilinear u8 resize of the runners; every float operation individually rounded (no FMA
// contraction) in the order the scalar host code evaluates them.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void resize_bilinear_u8_kernel(const unsigned char* in, int in_w, int in_h,
                                                                 unsigned char* out, int out_w, int out_h,
                                                                 float x_ratio, float y_ratio) {
#pragma clang fp contract(off)
    const long long total = (long long)out_w * out_h;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
        const int x = (int)(idx % out_w), y = (int)(idx / out_w);
        const float px = (float)x * x_ratio, py = (float)y * y_ratio;
        const int x0 = (int)px, y0 = (int)py;
        int x1 = x0 + 1, y1 = y0 + 1;
        if (x1 >= in_w) x1 = in_w - 1;
        if (y1 >= in_h) y1 = in_h - 1;
        const float dx = px - (float)x0, dy = py - (float)y0;
        const float w00 = (1.f - dx) * (1.f - dy);
        const float w01 = dx * (1.f - dy);
        const float w10 = (1.f - dx) * dy;
        const float w11 = dx * dy;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const float p00 = in[((long long)y0 * in_w + x0) * 3 + c];
            const float p01 = in[((long long)y0 * in_w + x1) * 3 + c];
            const float p10 = in[((long long)y1 * in_w + x0) * 3 + c];
            const float p11 = in[((long long)y1 * in_w + x1) * 3 + c];
            const float v = ((w00 * p00 + w01 * p01) + w10 * p10) + w11 * p11;
            out[idx * 3 + c] = (unsigned char)(v + 0.5f);
        }
    }
}

static inline int grid_for(long long total, int per_block = 256, int cap = 256 * 16) {
    long long g = (total + per_block - 1) / per_block;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace srmi

using namespace srmi;

extern "C" int srmi_conv_first(const srmi_conv_first_desc_t* d, void* hip_stream) {
    if (!d) SRMI_FAIL("srmi_conv_first: null descriptor");
    if (d->n <= 0 || d->h <= 0 || d->w <= 0 || d->cout <= 0 || d->cout % 32)
        SRMI_FAIL("srmi_conv_first: bad shape n=%d h=%d w=%d cout=%d", d->n, d->h, d->w, d->cout);
    if (!d->in || !d->w_kcio || !d->bias || (!d->out.ptr && !d->out2.ptr)) SRMI_FAIL("srmi_conv_first: null pointer");
    if (d->in_kind != SRMI_IN_F32_NCHW && d->in_kind != SRMI_IN_U8_NHWC) SRMI_FAIL("srmi_conv_first: bad in_kind");
    FirstParams p;
    p.n = d->n; p.h = d->h; p.w = d->w; p.cout = d->cout; p.in_kind = d->in_kind;
    p.in = d->in;
    for (int i = 0; i < 9; ++i) p.pre_mat[i] = d->pre_mat[i];
    for (int i = 0; i < 3; ++i) p.pre_bias[i] = d->pre_bias[i];
    p.wgt = d->w_kcio; p.bias = d->bias;
    p.out = to_view(d->out); p.out2 = to_view(d->out2);
    const long long total = (long long)d->n * d->h * d->w;
    if ((total + 255) / 256 > 0x7fffffffll) SRMI_FAIL("srmi_conv_first: problem too large");
    const dim3 grid((unsigned)((total + 255) / 256), (unsigned)(d->cout / 16));
    if (d->in_kind == SRMI_IN_U8_NHWC)
        hipLaunchKernelGGL(conv_first_kernel<SRMI_IN_U8_NHWC>, grid, dim3(256), 0, (hipStream_t)hip_stream, p);
    else
        hipLaunchKernelGGL(conv_first_kernel<SRMI_IN_F32_NCHW>, grid, dim3(256), 0, (hipStream_t)hip_stream, p);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_ca_scale(const float* chansum, int n, int tiles, int c, int cr, int hw,
                             const float* w1, const float* b1, const float* w2, const float* b2,
                             float* scale, void* hip_stream) {
    if (!chansum || !w1 || !b1 || !w2 || !b2 || !scale) SRMI_FAIL("srmi_ca_scale: null pointer");
    if (n <= 0 || tiles <= 0 || c <= 0 || cr <= 0 || hw <= 0 || c > 512) SRMI_FAIL("srmi_ca_scale: bad shape");
    const size_t lds = (size_t)((kCaParts + 1) * c + cr) * sizeof(float);
    hipLaunchKernelGGL(ca_scale_kernel, dim3(n), dim3(kCaThreads), lds, (hipStream_t)hip_stream, chansum, tiles, c, cr,
                       1.0f / (float)hw, w1, b1, w2, b2, scale);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_ca_apply(const srmi_tensor_t* y, const srmi_tensor_t* x, const float* scale,
                             float res_scale, int n, int h, int w, int c,
                             const srmi_tensor_t* out, const srmi_tensor_t* out2, void* hip_stream) {
    if (!y || !x || !y->ptr || !x->ptr || !scale || !out) SRMI_FAIL("srmi_ca_apply: null pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 16) SRMI_FAIL("srmi_ca_apply: bad shape (c must be a multiple of 16)");
    TView o2;
    o2.ptr = nullptr; o2.img_stride_b = o2.plane_stride_b = 0; o2.dtype = 0; o2.flags = 0;
    if (out2) o2 = to_view(*out2);
    const long long hw = (long long)h * w;
    hipLaunchKernelGGL(ca_apply_kernel, dim3(grid_for(n * hw * (c / 8), 256, 1 << 20)), dim3(256), 0, (hipStream_t)hip_stream,
                       to_view(*y), to_view(*x), scale, res_scale, n, hw, c, to_view(*out), o2);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_channel_sums(const srmi_tensor_t* x, int n, int h, int w, int c, float* sums, void* hip_stream) {
    if (!x || !x->ptr || !sums) SRMI_FAIL("srmi_channel_sums: null pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 8) SRMI_FAIL("srmi_channel_sums: bad shape (c must be a multiple of 8)");
    hipStream_t s = (hipStream_t)hip_stream;
    SRMI_HIP(hipMemsetAsync(sums, 0, (size_t)n * c * sizeof(float), s));
    const long long hw = (long long)h * w;
    // about 768 workgroups in all (three per CU; more only adds contention on the per-channel atomics),
    // segments a multiple of the 128 pixels a workgroup reads per step
    const int spans = (c + 15) / 16;
    long long seg = (hw * spans * n + 767) / 768;
    seg = (seg + 127) / 128 * 128;
    dim3 grid((unsigned)((hw + seg - 1) / seg), (unsigned)spans, (unsigned)n);
    hipLaunchKernelGGL(channel_sums_kernel, grid, dim3(256), 0, s, to_view(*x), hw, c, sums, (int)seg);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_absmax(const srmi_tensor_t* x, int n, int h, int w, int c, float* slot, void* hip_stream) {
    if (!x || !x->ptr || !slot) SRMI_FAIL("srmi_absmax: null pointer");
    if (n <= 0 || h <= 0 || w <= 0 || c <= 0 || c % 16) SRMI_FAIL("srmi_absmax: bad shape (c must be a multiple of 16)");
    const long long hw = (long long)h * w;
    hipLaunchKernelGGL(absmax_kernel, dim3(grid_for(n * hw * (c / 8), 256, 1024)), dim3(256), 0, (hipStream_t)hip_stream,
                       to_view(*x), n, hw, c, (unsigned*)slot);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_pixel_shuffle(const srmi_tensor_t* in, const srmi_tensor_t* out, int n, int h, int w,
                                  int c_in, int r, void* hip_stream) {
    if (!in || !out || !in->ptr || !out->ptr) SRMI_FAIL("srmi_pixel_shuffle: null pointer");
    if (in->dtype != out->dtype) SRMI_FAIL("srmi_pixel_shuffle: dtype mismatch");
    if (r < 2 || r > 4 || c_in % (r * r)) SRMI_FAIL("srmi_pixel_shuffle: bad r=%d / c_in=%d", r, c_in);
    const int c_out = c_in / (r * r);
    const int epc = 16 / dtype_size(in->dtype);
    if (c_out % epc) SRMI_FAIL("srmi_pixel_shuffle: c_out=%d is not a multiple of %d", c_out, epc);
    const long long total = (long long)n * h * r * w * r * (c_out / epc);
    hipStream_t s = (hipStream_t)hip_stream;
    if (in->dtype != SRMI_F32 && r == 2 && c_out % 16 == 0)
        hipLaunchKernelGGL(pixel_shuffle2_h16_kernel, dim3(grid_for((long long)n * (c_out / 16) * h * w, 256, 1 << 20)), dim3(256), 0, s,
                           to_view(*in), to_view(*out), n, h, w, c_out);
    else if (in->dtype == SRMI_F32)
        hipLaunchKernelGGL(pixel_shuffle_kernel<float>, dim3(grid_for(total)), dim3(256), 0, s, to_view(*in), to_view(*out), n, h, w, c_out, r);
    else
        hipLaunchKernelGGL(pixel_shuffle_kernel<unsigned short>, dim3(grid_for(total)), dim3(256), 0, s, to_view(*in), to_view(*out), n, h, w, c_out, r);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_nearest_up2(const srmi_tensor_t* in, const srmi_tensor_t* out, int n, int h, int w,
                                int c, void* hip_stream) {
    if (!in || !out || !in->ptr || !out->ptr) SRMI_FAIL("srmi_nearest_up2: null pointer");
    if (in->dtype != out->dtype) SRMI_FAIL("srmi_nearest_up2: dtype mismatch");
    const int cpb = dtype_cpb(in->dtype);
    if (c <= 0 || c % cpb) SRMI_FAIL("srmi_nearest_up2: c=%d is not a multiple of %d", c, cpb);
    const int planes = c / cpb;
    const long long total = (long long)n * planes * h * w * 2;
    hipLaunchKernelGGL(nearest_up2_kernel, dim3(grid_for(total, 256, 1 << 20)), dim3(256), 0, (hipStream_t)hip_stream,
                       to_view(*in), to_view(*out), n, h, w, planes);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_nchw_to_planes(const float* in, const srmi_tensor_t* out, int n, int c, int h, int w,
                                   void* hip_stream) {
    if (!in || !out || !out->ptr) SRMI_FAIL("srmi_nchw_to_planes: null pointer");
    const int cpb = dtype_cpb(out->dtype);
    const int c_pad = (c + cpb - 1) / cpb * cpb;
    const long long hw = (long long)h * w;
    // every plane holds a multiple of 8 channels, so groups of 8 cover the padded channel count exactly
    const long long blocks_y = (long long)n * (c_pad / 8);
    if (n <= 0 || c <= 0 || hw <= 0 || blocks_y > 65535) SRMI_FAIL("srmi_nchw_to_planes: bad shape n=%d c=%d h=%d w=%d", n, c, h, w);
    hipLaunchKernelGGL(nchw_to_planes_kernel, dim3(grid_for(hw, 256, 1024), (unsigned)blocks_y), dim3(256), 0,
                       (hipStream_t)hip_stream, in, to_view(*out), c, c_pad / 8, hw);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_planes_to_nchw(const srmi_tensor_t* in, float* out, int n, int c, int h, int w,
                                   void* hip_stream) {
    if (!in || !in->ptr || !out) SRMI_FAIL("srmi_planes_to_nchw: null pointer");
    const long long hw = (long long)h * w;
    const int groups = (c + 7) / 8;
    const long long blocks_y = (long long)n * groups;
    if (n <= 0 || c <= 0 || hw <= 0 || blocks_y > 65535) SRMI_FAIL("srmi_planes_to_nchw: bad shape n=%d c=%d h=%d w=%d", n, c, h, w);
    hipLaunchKernelGGL(planes_to_nchw_kernel, dim3(grid_for(hw, 256, 1024), (unsigned)blocks_y), dim3(256), 0,
                       (hipStream_t)hip_stream, to_view(*in), out, c, groups, hw);
    SRMI_HIP(hipGetLastError());
    return 0;
}

extern "C" int srmi_resize_bilinear_u8(const uint8_t* in, int in_w, int in_h, uint8_t* out, int out_w,
                                       int out_h, void* hip_stream) {
    if (!in || !out) SRMI_FAIL("srmi_resize_bilinear_u8: null pointer");
    if (in_w <= 0 || in_h <= 0 || out_w <= 0 || out_h <= 0) SRMI_FAIL("srmi_resize_bilinear_u8: bad size");
    const float xr = (float)in_w / (float)out_w, yr = (float)in_h / (float)out_h;
    hipLaunchKernelGGL(resize_bilinear_u8_kernel, dim3(grid_for((long long)out_w * out_h)), dim3(256), 0,
                       (hipStream_t)hip_stream, in, in_w, in_h, out, out_w, out_h, xr, yr);
    SRMI_HIP(hipGetLastError());
    return 0;
}
