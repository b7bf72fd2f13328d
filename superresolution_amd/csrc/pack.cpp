// Host-side repacking of OIHW float32 convolution weights into the LDS image the conv kernel
// stages with LDS-DMA.
//
// Packed layout: [cout tile][stage = (cin block, half)][tap = ky*3+kx][row = output channel within
// tile][32 bytes] where the 32 bytes are the stage's input channels (16 halves, 8 floats or 32 int8) as two
// 16-byte chunks, stored chunk index = logical chunk ^ ((row_in_image >> 3) & 1), row_in_image =
// tap * cout_tile + row.  With pixel-shuffle packing the output channels are permuted so that a
// 32-channel block holds one (i, j) sub-pixel position of 32 consecutive post-shuffle channels.
#include <math.h>
#include <string.h>

#include <vector>

#include "common.h"

namespace srmi {

static inline uint16_t f32_to_f16(float f) {
    // round-to-nearest-even, IEEE binary16 with subnormals
    uint32_t x;
    memcpy(&x, &f, 4);
    const uint32_t sign = (x >> 16) & 0x8000u;
    x &= 0x7FFFFFFFu;
    if (x >= 0x7F800000u) return (uint16_t)(sign | 0x7C00u | ((x > 0x7F800000u) ? 0x200u : 0u));
    if (x >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u);  // rounds to inf
    if (x < 0x33000001u) return (uint16_t)sign;               // rounds to zero
    int e = (int)(x >> 23) - 127;
    uint32_t m = (x & 0x7FFFFFu) | 0x800000u;
    int shift;
    uint32_t base;
    if (e < -14) {  // subnormal half
        shift = 13 + (-14 - e);
        base = 0;
    } else {
        shift = 13;
        base = (uint32_t)(e + 15) << 10;
        m &= 0x7FFFFFu;
    }
    uint32_t q = m >> shift;
    const uint32_t rem = m & ((1u << shift) - 1u);
    const uint32_t half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1u))) q++;
    return (uint16_t)(sign | (base + q));
}

static inline uint16_t f32_to_bf16(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    if ((x & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((x >> 16) | 0x40u);  // NaN stays NaN
    x += 0x7FFFu + ((x >> 16) & 1u);
    return (uint16_t)(x >> 16);
}

// packed output channel -> original output channel (or -1 for padding)
static inline int packed_to_orig(int q, int cout, int r) {
    if (q >= cout) return -1;
    if (r <= 1) return q;
    const int c_after = cout / (r * r);
    const int ij = q / c_after, c = q - ij * c_after;
    return c * r * r + ij;
}

}  // namespace srmi

using namespace srmi;

extern "C" size_t srmi_pack_conv3x3_bytes(int dtype, int cout, int cin, int cout_tile,
                                          int* bias_floats) {
    if (cout_tile != 32 && cout_tile != 64) return 0;
    const int cpb = dtype_cpb(dtype);
    if (cin <= 0 || cin % cpb || cout <= 0) return 0;
    const int ntile = (cout + cout_tile - 1) / cout_tile;
    if (bias_floats) *bias_floats = ntile * cout_tile;
    return (size_t)ntile * (cin / cpb) * 9 * cout_tile * kPixBytes;
}

extern "C" int srmi_pack_conv3x3(int dtype, int cout, int cin, int cout_tile, int ps_r,
                                 const float* w, const float* bias, void* wpk_out,
                                 float* bias_out) {
    int nbias = 0;
    const size_t bytes = srmi_pack_conv3x3_bytes(dtype, cout, cin, cout_tile, &nbias);
    if (!bytes) SRMI_FAIL("srmi_pack_conv3x3: bad shape dtype=%d cout=%d cin=%d cout_tile=%d", dtype, cout, cin, cout_tile);
    if (!w || !wpk_out || !bias_out) SRMI_FAIL("srmi_pack_conv3x3: null pointer");
    const int r = ps_r <= 1 ? 1 : ps_r;
    if (r > 1 && cout % (r * r * 32)) SRMI_FAIL("srmi_pack_conv3x3: cout=%d incompatible with pixel shuffle %d", cout, r);
    const int cpb = dtype_cpb(dtype), esz = dtype_size(dtype), epc = 16 / esz;  // elements per chunk
    const int nstage = cin / cpb, ntile = nbias / cout_tile;  // one stage per input plane
    char* out = (char*)wpk_out;
    for (int t = 0; t < ntile; ++t)
        for (int st = 0; st < nstage; ++st)
            for (int tap = 0; tap < 9; ++tap)
                for (int row = 0; row < cout_tile; ++row) {
                    const int o = packed_to_orig(t * cout_tile + row, cout, r);
                    const int img_row = tap * cout_tile + row;
                    char* dst_row = out + ((((size_t)t * nstage + st) * 9 + tap) * cout_tile + row) * 32;
                    for (int c = 0; c < 2; ++c) {
                        char* dst = dst_row + ((c ^ ((img_row >> 3) & 1)) << 4);
                        for (int e = 0; e < epc; ++e) {
                            const int ci = st * cpb + c * epc + e;
                            const float v = o < 0 ? 0.f : w[((size_t)o * cin + ci) * 9 + tap];
                            if (dtype == SRMI_F32) {
                                memcpy(dst + 4 * e, &v, 4);
                            } else if (dtype == SRMI_I8) {
                                // already quantised by the caller: integer-valued floats in [-127, 127]
                                const long q = lrintf(v);
                                dst[e] = (char)(signed char)(q < -127 ? -127 : q > 127 ? 127 : q);
                            } else {
                                const uint16_t hv = dtype == SRMI_F16 ? f32_to_f16(v) : f32_to_bf16(v);
                                memcpy(dst + 2 * e, &hv, 2);
                            }
                        }
                    }
                }
    for (int q = 0; q < nbias; ++q) {
        const int o = packed_to_orig(q, cout, r);
        bias_out[q] = (o < 0 || !bias) ? 0.f : bias[o];
    }
    return 0;
}

// Sub-pixel form of "nearest-x2 upsampling, then 3x3 convolution": row Y = 2y + py of the upsampled image reads source rows
// (Y + ky - 1) >> 1 = y - 1 + py + a with a in {0, 1}; taps that land on the same source row are added (f32), and
// likewise along x.  Packed output channel q = phase * cout + o, phase = 2 py + px (what the pixel-shuffle store expects).
extern "C" size_t srmi_pack_conv3x3_up2_bytes(int dtype, int cout, int cin, int cout_tile, int* bias_floats) {
    if (dtype != SRMI_F16 && dtype != SRMI_BF16) return 0;
    if (cin <= 0 || cin % 16 || cout <= 0 || (cout_tile != 32 && cout_tile != 64) || cout % cout_tile) return 0;
    if (bias_floats) *bias_floats = 4 * cout;
    return (size_t)(4 * cout / cout_tile) * (cin / 16) * 4 * cout_tile * 32;
}

extern "C" int srmi_pack_conv3x3_up2(int dtype, int cout, int cin, int cout_tile, const float* w, const float* bias,
                                     void* wpk_out, float* bias_out) {
    int nbias = 0;
    const size_t bytes = srmi_pack_conv3x3_up2_bytes(dtype, cout, cin, cout_tile, &nbias);
    if (!bytes) SRMI_FAIL("srmi_pack_conv3x3_up2: bad shape dtype=%d cout=%d cin=%d cout_tile=%d", dtype, cout, cin, cout_tile);
    if (!w || !wpk_out || !bias_out) SRMI_FAIL("srmi_pack_conv3x3_up2: null pointer");
    const int nstage = cin / 16, ntile = nbias / cout_tile;
    char* out = (char*)wpk_out;
    for (int t = 0; t < ntile; ++t)
        for (int st = 0; st < nstage; ++st)
            for (int tap = 0; tap < 4; ++tap)
                for (int row = 0; row < cout_tile; ++row) {
                    const int q = t * cout_tile + row, phase = q / cout, o = q - phase * cout;
                    const int py = phase >> 1, px = phase & 1, a = tap >> 1, b = tap & 1;
                    const int img_row = tap * cout_tile + row;
                    char* dst_row = out + ((((size_t)t * nstage + st) * 4 + tap) * cout_tile + row) * 32;
                    for (int c = 0; c < 2; ++c) {
                        char* dst = dst_row + ((c ^ ((img_row >> 3) & 1)) << 4);
                        for (int e = 0; e < 8; ++e) {
                            const int ci = st * 16 + c * 8 + e;
                            float v = 0.f;
                            for (int ky = 0; ky < 3; ++ky)
                                for (int kx = 0; kx < 3; ++kx) {
                                    // floor((p + k - 1) / 2) for p in {0, 1}: -1, 0, 0 (p = 0) and 0, 0, 1 (p = 1)
                                    const int ry = (py + ky + 1) / 2 - 1, rx = (px + kx + 1) / 2 - 1;
                                    if (ry == py - 1 + a && rx == px - 1 + b) v += w[((size_t)o * cin + ci) * 9 + ky * 3 + kx];
                                }
                            const uint16_t hv = dtype == SRMI_F16 ? f32_to_f16(v) : f32_to_bf16(v);
                            memcpy(dst + 2 * e, &hv, 2);
                        }
                    }
                }
    for (int q = 0; q < nbias; ++q) bias_out[q] = bias ? bias[q % cout] : 0.f;
    return 0;
}

// Winograd F(2x2,3x3): U = G g G^T, G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], per (output, input) channel pair.
extern "C" size_t srmi_pack_wino3x3_bytes(int cout, int cin, int* bias_floats) {
    if (cin <= 0 || cin % 16 || cout <= 0) return 0;
    const int ntile = (cout + 31) / 32;
    if (bias_floats) *bias_floats = ntile * 32;
    return (size_t)ntile * (cin / 16) * 16 * 32 * kPixBytes;
}

extern "C" int srmi_pack_wino3x3(int cout, int cin, const float* w, const float* bias, void* wpk_out, float* bias_out) {
    int nbias = 0;
    const size_t bytes = srmi_pack_wino3x3_bytes(cout, cin, &nbias);
    if (!bytes) SRMI_FAIL("srmi_pack_wino3x3: bad shape cout=%d cin=%d (cin must be a multiple of 16)", cout, cin);
    if (!w || !wpk_out || !bias_out) SRMI_FAIL("srmi_pack_wino3x3: null pointer");
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nstage = cin / 16, ntile = nbias / 32;
    char* out = (char*)wpk_out;
    for (int t = 0; t < ntile; ++t)
        for (int row = 0; row < 32; ++row) {
            const int o = t * 32 + row;
            for (int ci = 0; ci < cin; ++ci) {
                double U[4][4] = {};
                if (o < cout) {
                    const float* g = w + ((size_t)o * cin + ci) * 9;   // g[ky][kx]
                    double tmp[4][3];
                    for (int a = 0; a < 4; ++a)
                        for (int j = 0; j < 3; ++j) tmp[a][j] = G[a][0] * g[0 * 3 + j] + G[a][1] * g[1 * 3 + j] + G[a][2] * g[2 * 3 + j];
                    for (int a = 0; a < 4; ++a)
                        for (int b = 0; b < 4; ++b) U[a][b] = tmp[a][0] * G[b][0] + tmp[a][1] * G[b][1] + tmp[a][2] * G[b][2];
                }
                const int st = ci / 16, c = (ci % 16) / 8, e = ci % 8;
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        const int xi = a * 4 + b;
                        char* dst = out + ((((size_t)t * nstage + st) * 16 + xi) * 2 + c) * 512 + row * 16 + e * 2;
                        const uint16_t hv = f32_to_f16((float)U[a][b]);
                        memcpy(dst, &hv, 2);
                    }
            }
        }
    for (int q = 0; q < nbias; ++q) bias_out[q] = (q < cout && bias) ? bias[q] : 0.f;
    return 0;
}

// Winograd F(2,3) along x: U[ky][xi] = sum_kx G[xi][kx] g[ky][kx] per (output, input) channel pair.
extern "C" size_t srmi_pack_wino1d3x3_bytes(int cout, int cin, int* bias_floats) {
    if (cin <= 0 || cin % 16 || cout <= 0) return 0;
    const int ntile = (cout + 31) / 32;
    if (bias_floats) *bias_floats = ntile * 32;
    return (size_t)ntile * (cin / 16) * 12 * 32 * kPixBytes;
}

extern "C" int srmi_pack_wino1d3x3(int cout, int cin, const float* w, const float* bias, void* wpk_out, float* bias_out) {
    int nbias = 0;
    const size_t bytes = srmi_pack_wino1d3x3_bytes(cout, cin, &nbias);
    if (!bytes) SRMI_FAIL("srmi_pack_wino1d3x3: bad shape cout=%d cin=%d (cin must be a multiple of 16)", cout, cin);
    if (!w || !wpk_out || !bias_out) SRMI_FAIL("srmi_pack_wino1d3x3: null pointer");
    static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    const int nstage = cin / 16, ntile = nbias / 32;
    char* out = (char*)wpk_out;
    for (int t = 0; t < ntile; ++t)
        for (int row = 0; row < 32; ++row) {
            const int o = t * 32 + row;
            for (int ci = 0; ci < cin; ++ci) {
                const float* g = w + ((size_t)(o < cout ? o : 0) * cin + ci) * 9;   // g[ky][kx]
                const int st = ci / 16, c = (ci % 16) / 8, e = ci % 8;
                for (int ky = 0; ky < 3; ++ky)
                    for (int xi = 0; xi < 4; ++xi) {
                        const double u = o < cout ? G[xi][0] * g[ky * 3 + 0] + G[xi][1] * g[ky * 3 + 1] + G[xi][2] * g[ky * 3 + 2] : 0.0;
                        char* dst = out + ((((size_t)t * nstage + st) * 12 + ky * 4 + xi) * 2 + c) * 512 + row * 16 + e * 2;
                        const uint16_t hv = f32_to_f16((float)u);
                        memcpy(dst, &hv, 2);
                    }
            }
        }
    for (int q = 0; q < nbias; ++q) bias_out[q] = (q < cout && bias) ? bias[q] : 0.f;
    return 0;
}
