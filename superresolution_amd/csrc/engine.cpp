// Whole-network executor: builds the fused layer sequence of EDSR / RCAN / RRDBNet from the
// reference's state_dict and runs forward(img) as a chain of kernel launches on one HIP stream.
//
// Fused execution plans (one line = one launch):
//   RRDBNet (realesrgan_model.py:96-123)
//     conv_first(img) -> D[0:nf] and F
//     per RRDB, per RDB: conv1..4 write their grow_ch channels into slices of the dense buffer D
//       (this is torch.cat, realesrgan_model.py:27-30), conv5: *0.2 + x [(...)*0.2 + R on rdb3]
//       -> first nf channels of the OTHER dense buffer (ping-pong, so halo reads never race)
//     conv_body + F ; conv_up1 / conv_up2 read through a nearest-x2 address map ; conv_hr ;
//     conv_last -> RGB tail (f32 NCHW, or clamp*255 round -> u8)
//   EDSR (edsr_model.py:114-124)
//     head(img, MeanShift in the loader) ; 32 x [conv+ReLU ; conv*res_scale + x] ; conv + head ;
//     upsampler convs with PixelShuffle stores ; tail conv -> RGB tail with add_mean
//   RCAN (rcan_model.py:149-159)
//     as EDSR, with per RCAB: conv+ReLU ; conv (+ per-tile channel sums) ; squeeze/excite ;
//     x + res_scale * y * s ; and per group: conv + group input.
//
// Precision: activations are stored in the compute dtype; with trunk_f32 (half dtypes) every
// tensor that is only ever consumed as a residual is additionally kept in f32, so rounding to
// half happens once per convolution input and never accumulates along the residual chain.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <map>
#include <mutex>
#include <string>
#include <string_view>
#include <vector>

#include "common.h"

// Workspace of srmi_model_forward_u8_noalloc: one array per GPU, part of the library's code object, so
// the entry point never allocates.  Host code must not pass the array's name to a kernel or a memset
// (on the host the name is only a stand-in for the symbol): it uses the per-device address that
// hipGetSymbolAddress returned at set-up (static_workspace() below).
#define SRMI_STATIC_WS_BYTES (64ull << 20)
static __device__ __align__(256) unsigned char scratch[SRMI_STATIC_WS_BYTES];

namespace srmi {

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
    bool set = false;
};

struct ConvW {
    void* wpk_wino = nullptr;  // Winograd F(2x2,3x3) form of the same weights (RRDBNet dense blocks, f16), else null
    void* wpk_wino_x = nullptr;  // Winograd F(2,3)-along-x form
    void* wpk_up2 = nullptr;     // sub-pixel form (four 2x2 convolutions on the source grid) of a convolution behind a
    float* bias_up2 = nullptr;   // nearest-x2 upsampling (RRDBNet conv_up1 / conv_up2, f16), else null
    int cout_tile_up2 = 32;
    void* wpk = nullptr;
    float* bias = nullptr;
    float* wscale = nullptr;  // int8 convs: per-output-channel dequantisation scale
    int cout = 0, cin = 0, cout_tile = 0, ps_r = 1;
};

struct CaW {
    float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
};

struct Buf {
    char* ptr = nullptr;
    int dtype = 0, c = 0, h = 0, w = 0;
};

static inline int64_t numel(const std::vector<int64_t>& s) {
    int64_t n = 1;
    for (auto v : s) n *= v;
    return n;
}

}  // namespace srmi

using namespace srmi;

struct srmi_model {
    srmi_model_desc_t d;
    std::vector<std::string> keys;
    std::map<std::string, HostTensor> tensors;
    bool finalized = false;

    // transparent comparator: the per-launch lookups pass a string_view and allocate nothing on the host
    std::map<std::string, ConvW, std::less<>> convs;
    std::vector<CaW> cas;
    float* first_w = nullptr;  // [27][nf]
    float* first_b = nullptr;
    float sub_mat[9], sub_bias[3], add_mat[9], add_bias[3];
    float core_shift[3] = {0, 0, 0};
    bool has_core_shift = false;
    std::vector<void*> dev_allocs;

    char* ws = nullptr;
    size_t ws_bytes = 0;
    int sweep = 1;
    bool fuse_attention = true;  // RCAN, half-precision trunk: attention output formed by the consuming conv
    bool subpixel_up = true;     // RRDBNet, f16: conv_up1 / conv_up2 in their sub-pixel form (4/9 of the multiplies)
    bool fuse_rcab = true;       // RCAN / EDSR, f16 operands, 64 features: both convolutions of a block in one kernel (srmi_rcab_fused)
    int mb_override = 0;
    // RRDBNet, f16: the grown channels x1..x4 of every dense block (and the weights that multiply them)
    // carry 8 significant bits; the trunk keeps full f16 precision (see srmi_model_set_growth_round8)
    bool growth_round8 = true;
    // RRDBNet, f16: the dense blocks' convolutions run in their Winograd F(2x2,3x3) form (srmi_model_set_winograd).
    // Off by default: measured slower than the direct form on whole layers (its 256 accumulator registers allow one
    // wave per SIMD, so nothing on the CU hides a workgroup's first loads and its epilogue).
    int winograd = SRMI_ALGO_DIRECT;   // SRMI_ALGO_* of the dense blocks' convolutions

    // Independent sweeps of one call run on `lanes` internal streams that fork from and join the
    // caller's stream, so that one sweep's launch gaps, prologues and epilogues overlap another's
    // main loops.  Each lane owns a slice of the workspace.
    static constexpr int kMaxLanes = 4;
    int lanes = 3;
    hipStream_t lane_stream[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t lane_done[kMaxLanes] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t fork = nullptr;
    // Calls on one model share its workspace and lanes: `call_mutex` serialises their host side and every
    // call waits for `last_done` (recorded on the previous caller's stream), so calls issued from
    // different streams or threads run one after the other on the device instead of overwriting each
    // other's activations.
    std::mutex call_mutex;
    hipEvent_t last_done = nullptr;
    bool has_last_done = false;
    char* static_ws = nullptr;  // this device's copy of `scratch` (resolved in srmi_model_finalize)

    // w8a8 mode (RRDBNet): calibration statistics and the int8 form of the dense-block convs.
    // Slot 5*g + j holds max |x| of tensor j of dense block g (j = 0: the block's trunk input, 1..4: x1..x4);
    // block 3*num_block is the trunk behind the last block.
    bool calibrating = false, calibrated = false, w8a8 = false;
    float* calib_dev = nullptr;
    std::vector<float> absmax;
    std::map<std::string, ConvW, std::less<>> convs_q;
    std::vector<void*> dev_allocs_q;
    int calib_slots() const { return 5 * (3 * d.n_blocks + 1); }
    float act_scale(int g, int j) const {
        const float a = absmax[5 * g + j];
        return a > 0.f ? a / 127.f : 1.f;
    }

    // Host-buffer path (srmi_model_forward_u8_host): one slot per lane = pinned staging + device images
    struct HostSlot {
        uint8_t *pin_in = nullptr, *pin_out = nullptr, *dev_in = nullptr, *dev_out = nullptr;
        size_t in_cap = 0, out_cap = 0;
        hipEvent_t done = nullptr;
    };
    HostSlot slot[kMaxLanes];

    int cdt() const { return d.dtype; }
    int tdt() const { return (d.dtype == SRMI_F32 || d.trunk_f32) ? SRMI_F32 : d.dtype; }  // trunk dtype
    bool split_trunk() const { return tdt() != cdt(); }
};

namespace srmi {

static void expect(srmi_model* m, const std::string& key, std::vector<int64_t> shape) {
    m->keys.push_back(key);
    HostTensor t;
    t.shape = std::move(shape);
    m->tensors[key] = std::move(t);
}

static void expect_conv(srmi_model* m, const std::string& name, int cout, int cin, int k) {
    expect(m, name + ".weight", {cout, cin, k, k});
    expect(m, name + ".bias", {cout});
}

static int upsampler_stages(int scale, int* r) {
    if (scale == 3) { *r = 3; return 1; }
    *r = 2;
    int n = 0;
    while ((1 << n) < scale) ++n;
    return n;
}

static int build_keys(srmi_model* m) {
    const auto& d = m->d;
    const int nf = d.n_feats;
    if (d.arch == SRMI_ARCH_RRDBNET) {
        // registration order of RRDBNet.__init__ (realesrgan_model.py:67-94)
        const int gc = d.grow_ch;
        expect_conv(m, "conv_first", nf, 3, 3);
        for (int i = 0; i < d.n_blocks; ++i)
            for (int r = 1; r <= 3; ++r)
                for (int j = 1; j <= 5; ++j) {
                    char name[96];
                    snprintf(name, sizeof(name), "body.%d.rdb%d.conv%d", i, r, j);
                    expect_conv(m, name, j == 5 ? nf : gc, nf + (j - 1) * gc, 3);
                }
        expect_conv(m, "conv_body", nf, nf, 3);
        expect_conv(m, "conv_up1", nf, nf, 3);
        if (d.scale == 4) expect_conv(m, "conv_up2", nf, nf, 3);
        expect_conv(m, "conv_hr", nf, nf, 3);
        expect_conv(m, "conv_last", 3, nf, 3);
        return 0;
    }
    // EDSR.__init__ (edsr_model.py:82-112) / RCAN.__init__ (rcan_model.py:123-147)
    if (d.with_meanshift) {
        expect_conv(m, "sub_mean", 3, 3, 1);
        expect_conv(m, "add_mean", 3, 3, 1);
    }
    expect_conv(m, "head.0", nf, 3, 3);
    char name[128];
    if (d.arch == SRMI_ARCH_EDSR) {
        for (int i = 0; i < d.n_blocks; ++i) {
            snprintf(name, sizeof(name), "body.%d.body.0", i);
            expect_conv(m, name, nf, nf, 3);
            snprintf(name, sizeof(name), "body.%d.body.2", i);
            expect_conv(m, name, nf, nf, 3);
        }
        snprintf(name, sizeof(name), "body.%d", d.n_blocks);
        expect_conv(m, name, nf, nf, 3);
    } else {
        const int cr = nf / d.reduction;
        for (int g = 0; g < d.n_groups; ++g) {
            for (int b = 0; b < d.n_blocks; ++b) {
                snprintf(name, sizeof(name), "body.%d.body.%d.body.0", g, b);
                expect_conv(m, name, nf, nf, 3);
                snprintf(name, sizeof(name), "body.%d.body.%d.body.2", g, b);
                expect_conv(m, name, nf, nf, 3);
                snprintf(name, sizeof(name), "body.%d.body.%d.body.3.conv_du.0", g, b);
                expect_conv(m, name, cr, nf, 1);
                snprintf(name, sizeof(name), "body.%d.body.%d.body.3.conv_du.2", g, b);
                expect_conv(m, name, nf, cr, 1);
            }
            snprintf(name, sizeof(name), "body.%d.body.%d", g, d.n_blocks);
            expect_conv(m, name, nf, nf, 3);
        }
        snprintf(name, sizeof(name), "body.%d", d.n_groups);
        expect_conv(m, name, nf, nf, 3);
    }
    int r;
    const int stages = upsampler_stages(d.scale, &r);
    for (int s = 0; s < stages; ++s) {
        snprintf(name, sizeof(name), "tail.0.%d", 2 * s);
        expect_conv(m, name, r * r * nf, nf, 3);
    }
    expect_conv(m, "tail.1", 3, nf, 3);
    return 0;
}

static int dev_upload(srmi_model* m, const void* host, size_t bytes, void** out) {
    void* p = nullptr;
    SRMI_HIP(hipMalloc(&p, bytes));
    m->dev_allocs.push_back(p);
    SRMI_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
    *out = p;
    return 0;
}

static int upload_conv(srmi_model* m, const std::string& name, int ps_r) {
    const HostTensor& w = m->tensors[name + ".weight"];
    const HostTensor& b = m->tensors[name + ".bias"];
    ConvW cw;
    cw.cout = (int)w.shape[0];
    cw.cin = (int)w.shape[1];
    // 32-wide tiles keep a workgroup under 168 registers and 53 KiB of LDS, so three fit a CU.  In
    // RRDBNet, where four of five convs of a dense block have 32 outputs anyway, that also pays for
    // the 64-output convs; EDSR and RCAN (64/256 outputs throughout) run faster on 64-wide tiles,
    // which fetch the pixel patch once per 64 outputs.
    // (output counts that are not a multiple of 64, e.g. n_feats 96 or 32*9 for scale 3, also take 32)
    cw.cout_tile = (cw.cout <= 32 || cw.cout % 64 || m->d.arch == SRMI_ARCH_RRDBNET) ? 32 : 64;
#ifdef SRMI_DIAG  // diagnostic builds only: the shipped library takes no tiling from the environment
    if (getenv("SRMI_COUT_TILE")) cw.cout_tile = atoi(getenv("SRMI_COUT_TILE")) == 32 || cw.cout <= 32 ? 32 : 64;
    // 64-wide tiles for RRDBNet's convs outside the dense blocks only (conv_body, the upsampling and HR convs)
    if (getenv("SRMI_COUT_TILE_OUTER") && name.find(".rdb") == std::string::npos && cw.cout % 64 == 0) cw.cout_tile = 64;
#endif
    cw.ps_r = ps_r;
    std::vector<float> rounded;
    const float* wsrc = w.data.data();
    if (m->growth_round8 && m->d.arch == SRMI_ARCH_RRDBNET && m->cdt() == SRMI_F16 && name.find(".rdb") != std::string::npos) {
        // rounded to bfloat16 precision (nearest even): every weight of conv1..4, whose outputs are
        // themselves kept at 8 significant bits, and in conv5 the columns of the grown input
        // channels [n_feats, cin); conv5's trunk columns keep f16 precision
        const bool is_conv5 = name.size() >= 5 && name.compare(name.size() - 5, 5, "conv5") == 0;
        rounded = w.data;
        for (int o = 0; o < cw.cout; ++o)
            for (int ci = is_conv5 ? m->d.n_feats : 0; ci < cw.cin; ++ci)
                for (int t = 0; t < 9; ++t) {
                    float& v = rounded[((size_t)o * cw.cin + ci) * 9 + t];
                    uint32_t u;
                    memcpy(&u, &v, 4);
                    if ((u & 0x7f800000u) != 0x7f800000u) u = (u + 0x7fffu + ((u >> 16) & 1u)) & 0xffff0000u;
                    memcpy(&v, &u, 4);
                }
        wsrc = rounded.data();
    }
    int nbias = 0;
    const size_t bytes = srmi_pack_conv3x3_bytes(m->cdt(), cw.cout, cw.cin, cw.cout_tile, &nbias);
    if (!bytes) SRMI_FAIL("%s: unsupported conv shape [%d,%d,3,3] for this dtype", name.c_str(), cw.cout, cw.cin);
    std::vector<char> wp(bytes);
    std::vector<float> bp(nbias);
    if (srmi_pack_conv3x3(m->cdt(), cw.cout, cw.cin, cw.cout_tile, ps_r, wsrc, b.data.data(), wp.data(), bp.data()))
        return -1;
    if (dev_upload(m, wp.data(), bytes, &cw.wpk)) return -1;
    if (dev_upload(m, bp.data(), nbias * sizeof(float), (void**)&cw.bias)) return -1;
    // the dense blocks' convolutions also in their Winograd form (from the unrounded weights: the transformed
    // weights are rounded once, to f16); same bias vector (32-wide tiles in both forms)
    if (m->d.arch == SRMI_ARCH_RRDBNET && m->cdt() == SRMI_F16 && name.find(".rdb") != std::string::npos &&
        cw.cout_tile == 32 && cw.cin % 32 == 0) {
        int nb2 = 0;
        const size_t wb = srmi_pack_wino3x3_bytes(cw.cout, cw.cin, &nb2);
        if (wb && nb2 == nbias) {
            std::vector<char> up(wb);
            std::vector<float> b2(nb2);
            if (srmi_pack_wino3x3(cw.cout, cw.cin, w.data.data(), b.data.data(), up.data(), b2.data())) return -1;
            if (dev_upload(m, up.data(), wb, &cw.wpk_wino)) return -1;
        }
        const size_t xb = srmi_pack_wino1d3x3_bytes(cw.cout, cw.cin, &nb2);
        if (xb && nb2 == nbias) {
            std::vector<char> up(xb);
            std::vector<float> b2(nb2);
            if (srmi_pack_wino1d3x3(cw.cout, cw.cin, w.data.data(), b.data.data(), up.data(), b2.data())) return -1;
            if (dev_upload(m, up.data(), xb, &cw.wpk_wino_x)) return -1;
        }
    }
    // the convolutions behind RRDBNet's nearest-x2 upsamplings also in their sub-pixel form (from the unrounded weights:
    // the tap sums are rounded once)
    // (f16 only: with bf16's 8-bit significands the extra rounding of the sums costs the network 1 dB)
    if (m->d.arch == SRMI_ARCH_RRDBNET && m->cdt() == SRMI_F16 &&
        (name == "conv_up1" || name == "conv_up2") && cw.cout % cw.cout_tile == 0) {
        int nb2 = 0;
        // one phase of 64 outputs per workgroup where the width allows (measured: 183 against 207 us for conv_up2 at 2040x1356)
        cw.cout_tile_up2 = cw.cout % 64 == 0 ? 64 : cw.cout_tile;
        const size_t ub = srmi_pack_conv3x3_up2_bytes(m->cdt(), cw.cout, cw.cin, cw.cout_tile_up2, &nb2);
        if (ub) {
            std::vector<char> up(ub);
            std::vector<float> b2(nb2);
            if (srmi_pack_conv3x3_up2(m->cdt(), cw.cout, cw.cin, cw.cout_tile_up2, w.data.data(), b.data.data(), up.data(), b2.data())) return -1;
            if (dev_upload(m, up.data(), ub, &cw.wpk_up2)) return -1;
            if (dev_upload(m, b2.data(), nb2 * sizeof(float), (void**)&cw.bias_up2)) return -1;
        }
    }
    m->convs[name] = cw;
    return 0;
}

static int upload_first(srmi_model* m, const std::string& name) {
    const HostTensor& w = m->tensors[name + ".weight"];  // [nf][3][3][3]
    const HostTensor& b = m->tensors[name + ".bias"];
    const int nf = (int)w.shape[0];
    std::vector<float> t((size_t)27 * nf);
    for (int o = 0; o < nf; ++o)
        for (int c = 0; c < 3; ++c)
            for (int k = 0; k < 9; ++k) t[(size_t)(k * 3 + c) * nf + o] = w.data[((size_t)o * 3 + c) * 9 + k];
    if (dev_upload(m, t.data(), t.size() * sizeof(float), (void**)&m->first_w)) return -1;
    return dev_upload(m, b.data.data(), nf * sizeof(float), (void**)&m->first_b);
}

// ---- workspace arena ---------------------------------------------------------------------------
struct Arena {
    size_t off = 0;
    char* base = nullptr;
    int ns = 1;
    Buf take(int c, int h, int w, int dtype) {
        Buf b;
        b.dtype = dtype; b.c = c; b.h = h; b.w = w;
        const size_t planes = (size_t)(c + dtype_cpb(dtype) - 1) / dtype_cpb(dtype);
        b.ptr = base ? base + off : nullptr;
        off += ((size_t)ns * planes * h * w * kPixBytes + 255) & ~(size_t)255;
        return b;
    }
    float* take_floats(size_t n) {
        float* p = base ? (float*)(base + off) : nullptr;
        off += (n * sizeof(float) + 255) & ~(size_t)255;
        return p;
    }
};

static srmi_tensor_t tv(const Buf& b, int c0 = 0) {
    srmi_tensor_t t;
    memset(&t, 0, sizeof(t));
    if (!b.ptr) return t;
    const int cpb = dtype_cpb(b.dtype);
    const int64_t plane = (int64_t)b.h * b.w * cpb;
    t.ptr = b.ptr + (size_t)(c0 / cpb) * plane * dtype_size(b.dtype);
    t.plane_stride = plane;
    t.img_stride = plane * ((b.c + cpb - 1) / cpb);
    t.dtype = b.dtype;
    return t;
}

static int pick_rows_per_wave(const srmi_model* m, int h) {
    if (m->mb_override) return m->mb_override;
    // Three rows per wave (12-row tiles) is the fastest variant on every layer of the three networks
    // when kernels of several lanes share the CUs; only images shorter than one such tile use the
    // 8-row tile.
    return h <= 8 ? 2 : 3;
}

// Rows per wave of a convolution with a gated input: its pixel pieces pass through registers, which the 12-row
// variant of the 64-channel tile does not have to spare.
static int gate_rows_per_wave(const srmi_model* m, int h) {
    if (m->mb_override) return m->mb_override > 3 ? 3 : m->mb_override;
#ifdef SRMI_DIAG
    if (getenv("SRMI_GATE_MB")) return atoi(getenv("SRMI_GATE_MB"));
#endif
    (void)h;
    return 2;
}

struct Exec {
    srmi_model* m;
    hipStream_t stream;
    int ns;  // images in this sweep

    // input of a conv that consumes a channel-attention block's output without materialising it first
    struct Gate {
        srmi_tensor_t body;   // the block's body before the attention scale
        const float* scale;   // float[ns][c]
        float alpha;          // res_scale
        srmi_tensor_t sum;    // where the block's output is stored on the way (ptr NULL: nowhere)
    };

    // channel attention finished by the conv that produces the block's body (its last workgroup per image)
    struct Attention {
        const CaW* w;
        int cr;
        uint64_t* sums;    // [ns][SRMI_CA_SLOTS][c], zero between launches
        uint32_t* count;   // [ns], zero between launches
        float* scale;      // [ns][c]
    };

    int conv(const char* name, const srmi_tensor_t& in, int h, int w, int act, bool up2,
             float alpha, const srmi_tensor_t* res1, float beta, const srmi_tensor_t* res2,
             const srmi_tensor_t* out, const srmi_tensor_t* out2, float* chansum = nullptr, int force_mb = 0,
             const Gate* gate = nullptr, const Attention* att = nullptr) {
        auto it = m->convs.find(std::string_view(name));
        if (it == m->convs.end()) SRMI_FAIL("internal: conv '%s' not uploaded", name);
        const ConvW& cw = it->second;
        srmi_conv_desc_t d;
        memset(&d, 0, sizeof(d));
        d.dtype = m->cdt();
        d.n = ns; d.h = h; d.w = w;
        d.cin = cw.cin; d.cout = cw.cout;
        d.in_up2 = up2; d.act = act; d.ps_r = cw.ps_r; d.cout_tile = cw.cout_tile;
        d.rows_per_wave = force_mb ? force_mb : pick_rows_per_wave(m, h);
        d.in = in;
        d.wpk = cw.wpk; d.bias = cw.bias;
        // Winograd form where it exists and applies (not while calibrating: the recorded ranges belong to the direct form)
        void* const wino_w = m->winograd == SRMI_ALGO_WINOGRAD ? cw.wpk_wino : m->winograd == SRMI_ALGO_WINOGRAD_X ? cw.wpk_wino_x : nullptr;
        if (wino_w && !up2 && !gate && !att && !chansum && h > 8 && !m->calibrating) {
            d.algo = m->winograd;
            d.wpk = wino_w;
            d.rows_per_wave = 0;
        }
        // an upsampling convolution in its sub-pixel form (not while calibrating, as above)
        if (up2 && cw.wpk_up2 && m->subpixel_up && !res1 && !res2 && !out2 && !gate && !att && !chansum && !m->calibrating &&
            d.rows_per_wave <= 3) {
            d.algo = SRMI_ALGO_SUBPIXEL_UP2;
            d.wpk = cw.wpk_up2;
            d.bias = cw.bias_up2;
            d.cout_tile = cw.cout_tile_up2;
        }
        d.alpha = alpha; d.beta = beta;
        if (res1) d.res1 = *res1;
        if (res2) d.res2 = *res2;
        if (out) d.out = *out;
        if (out2) d.out2 = *out2;
        d.chansum = chansum;
        if (att) {
            d.ca.sums = att->sums; d.ca.count = att->count; d.ca.scale = att->scale; d.ca.cr = att->cr;
            d.ca.w1 = att->w->w1; d.ca.b1 = att->w->b1; d.ca.w2 = att->w->w2; d.ca.b2 = att->w->b2;
        }
        if (gate) {
            d.in_gate = gate->body; d.gate = gate->scale; d.gate_alpha = gate->alpha; d.in_sum = gate->sum;
        }
        return srmi_conv3x3(&d, stream);
    }

    // both convolutions of a residual channel-attention block in one launch (srmi_rcab_fused); gate == nullptr: the
    // block's input is the plain tensor `in`
    int rcab(const char* name1, const char* name2, const srmi_tensor_t& in, int h, int w, const Gate* gate,
             const float* finite_floats, const srmi_tensor_t& out, const Attention* att) {
        auto i1 = m->convs.find(std::string_view(name1));
        auto i2 = m->convs.find(std::string_view(name2));
        if (i1 == m->convs.end() || i2 == m->convs.end()) SRMI_FAIL("internal: convs '%s' / '%s' not uploaded", name1, name2);
        srmi_rcab_desc_t r;
        memset(&r, 0, sizeof(r));
        r.dtype = m->cdt();
        r.n = ns; r.h = h; r.w = w; r.c = i1->second.cout;
        r.in = in;
        if (gate) {
            r.in_gate = gate->body; r.gate = gate->scale; r.gate_alpha = gate->alpha; r.in_sum = gate->sum;
        } else {
            r.in_gate = in; r.gate = finite_floats; r.gate_alpha = 0.f;
        }
        r.w1 = i1->second.wpk; r.b1 = i1->second.bias;
        r.w2 = i2->second.wpk; r.b2 = i2->second.bias;
        r.out = out;
        if (att) {
            r.ca.sums = att->sums; r.ca.count = att->count; r.ca.scale = att->scale; r.ca.cr = att->cr;
            r.ca.w1 = att->w->w1; r.ca.b1 = att->w->b1; r.ca.w2 = att->w->w2; r.ca.b2 = att->w->b2;
        }
        return srmi_rcab_fused(&r, stream);
    }

    // int8 form of a dense-block conv (w8a8 mode)
    int conv_q(const char* name, const srmi_tensor_t& in, int h, int w, int act, float alpha,
               const srmi_tensor_t* res1, float beta, const srmi_tensor_t* res2, const srmi_tensor_t* out,
               const srmi_tensor_t* out2) {
        auto it = m->convs_q.find(std::string_view(name));
        if (it == m->convs_q.end()) SRMI_FAIL("internal: int8 conv '%s' not prepared", name);
        const ConvW& cw = it->second;
        srmi_conv_desc_t d;
        memset(&d, 0, sizeof(d));
        d.dtype = SRMI_I8;
        d.n = ns; d.h = h; d.w = w;
        d.cin = cw.cin; d.cout = cw.cout;
        d.act = act; d.ps_r = 1; d.cout_tile = cw.cout_tile;
        d.rows_per_wave = pick_rows_per_wave(m, h);
        d.in = in;
        d.wpk = cw.wpk; d.bias = cw.bias; d.wscale = cw.wscale;
        d.alpha = alpha; d.beta = beta;
        if (res1) d.res1 = *res1;
        if (res2) d.res2 = *res2;
        if (out) d.out = *out;
        if (out2) d.out2 = *out2;
        return srmi_conv3x3(&d, stream);
    }

    // calibration: fold max |x| of a freshly written tensor into its slot
    int observe(const srmi_tensor_t& t, int h, int w, int c, int slot) {
        if (!m->calibrating) return 0;
        return srmi_absmax(&t, ns, h, w, c, m->calib_dev + slot, stream);
    }

    int tail(const char* name, const srmi_tensor_t& in, int h, int w, int mode, void* dst,
             const float* mat, const float* bias, float lo, float hi, float scale, int round) {
        auto it = m->convs.find(std::string_view(name));
        if (it == m->convs.end()) SRMI_FAIL("internal: conv '%s' not uploaded", name);
        const ConvW& cw = it->second;
        srmi_conv_desc_t d;
        memset(&d, 0, sizeof(d));
        d.dtype = m->cdt();
        d.n = ns; d.h = h; d.w = w;
        d.cin = cw.cin; d.cout = 3;
        d.ps_r = 1; d.cout_tile = cw.cout_tile;
        d.rows_per_wave = pick_rows_per_wave(m, h);
        d.in = in;
        d.wpk = cw.wpk; d.bias = cw.bias;
        d.alpha = d.beta = 1.f;
        d.tail_mode = mode;
        d.tail_out = dst;
        for (int i = 0; i < 9; ++i) d.tail_mat[i] = mat[i];
        for (int i = 0; i < 3; ++i) d.tail_bias[i] = bias[i];
        d.tail_lo = lo; d.tail_hi = hi; d.tail_scale = scale; d.tail_round = round;
        return srmi_conv3x3(&d, stream);
    }

    int first(const void* in, int in_kind, int h, int w, const float* mat, const float* bias,
              const srmi_tensor_t* out, const srmi_tensor_t* out2) {
        srmi_conv_first_desc_t d;
        memset(&d, 0, sizeof(d));
        d.n = ns; d.h = h; d.w = w; d.cout = m->d.n_feats;
        d.in_kind = in_kind;
        d.in = in;
        for (int i = 0; i < 9; ++i) d.pre_mat[i] = mat[i];
        for (int i = 0; i < 3; ++i) d.pre_bias[i] = bias[i];
        d.w_kcio = m->first_w; d.bias = m->first_b;
        if (out) d.out = *out;
        if (out2) d.out2 = *out2;
        return srmi_conv_first(&d, stream);
    }
};

struct IoSpec {
    int in_kind;            // SRMI_IN_*
    const void* in;
    int tail_mode;          // SRMI_TAIL_*
    void* out;
    float pre_mat[9], pre_bias[3];
    float post_mat[9], post_bias[3];
    float lo, hi, scale;
    int round;
};

static const float kIdentity[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};

// The plan functions are written once and run in two modes: with a null arena base they only
// measure the workspace; with a real base they launch.
static int run_rrdbnet(srmi_model* m, Arena& A, Exec* ex, const IoSpec* io, int h, int w) {
    const auto& d = m->d;
    const int nf = d.n_feats, gc = d.grow_ch, T = m->cdt(), TR = m->tdt();
    const bool split = m->split_trunk();
    Buf D[2] = {A.take(nf + 4 * gc, h, w, T), A.take(nf + 4 * gc, h, w, T)};
    Buf R[2] = {A.take(nf, h, w, TR), A.take(nf, h, w, TR)};
    Buf X[2];
    if (split) { X[0] = A.take(nf, h, w, TR); X[1] = A.take(nf, h, w, TR); }
    Buf F = A.take(nf, h, w, TR);
    const int up = d.scale == 4 ? 2 : 1;
    Buf U1 = A.take(nf, 2 * h, 2 * w, T);
    Buf U2, U3;
    const int hh = h * d.scale, ww = w * d.scale;
    if (up == 2) U2 = A.take(nf, hh, ww, T);
    U3 = A.take(nf, hh, ww, T);
    if (!ex) return 0;
#ifdef SRMI_DIAG
    // timing ablation (wrong results): both dense buffers are one, a lane touches 66 MB less
    if (getenv("SRMI_ABL_ONE_DENSE")) D[1] = D[0];
#endif

    // conv_first -> D0[0:nf] (conv input of the first RDB) and F (kept for the long skip);
    // F doubles as R of the first RRDB.
    {
        srmi_tensor_t o = tv(D[0]), o2 = tv(F);
        if (ex->first(io->in, io->in_kind, h, w, io->pre_mat, io->pre_bias, &o, &o2)) return -1;
        if (ex->observe(o, h, w, nf, 0)) return -1;
    }
    int k = 0;  // dense buffer holding the current trunk in channels [0, nf)
    char name[96];
    for (int i = 0; i < d.n_blocks; ++i) {
        const Buf& Rin = i == 0 ? F : R[i & 1];
        for (int r = 1; r <= 3; ++r) {
            const srmi_tensor_t in = tv(D[k]);
            for (int j = 1; j <= 4; ++j) {
                snprintf(name, sizeof(name), "body.%d.rdb%d.conv%d", i, r, j);
                srmi_tensor_t o = tv(D[k], nf + (j - 1) * gc);
                if (m->growth_round8 && T == SRMI_F16) o.flags = SRMI_TENSOR_ROUND8;
                if (ex->conv(name, in, h, w, SRMI_ACT_LRELU, false, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
                if (ex->observe(o, h, w, gc, 5 * (3 * i + r - 1) + j)) return -1;
            }
            snprintf(name, sizeof(name), "body.%d.rdb%d.conv5", i, r);
            // residual x of this RDB: the RRDB input for rdb1, else the previous RDB's output
            srmi_tensor_t res1;
            if (r == 1) res1 = tv(Rin);
            else res1 = split ? tv(X[r & 1]) : tv(D[k]);
            const srmi_tensor_t o = tv(D[k ^ 1]);
            if (r < 3) {
                srmi_tensor_t o2 = split ? tv(X[(r + 1) & 1]) : srmi_tensor_t{};
                if (ex->conv(name, in, h, w, SRMI_ACT_NONE, false, 0.2f, &res1, 1.f, nullptr, &o, split ? &o2 : nullptr)) return -1;
            } else {
                const srmi_tensor_t res2 = tv(Rin);
                const srmi_tensor_t o2 = tv(R[(i + 1) & 1]);
                if (ex->conv(name, in, h, w, SRMI_ACT_NONE, false, 0.2f, &res1, 0.2f, &res2, &o, &o2)) return -1;
            }
            if (ex->observe(o, h, w, nf, 5 * (3 * i + r))) return -1;
            k ^= 1;
        }
    }
    // conv_body + F -> feat, stored in the now-free other dense buffer
    Buf FE = D[k ^ 1];
    FE.c = nf;
    {
        const srmi_tensor_t in = tv(D[k]), res = tv(F), o = tv(FE);
        if (ex->conv("conv_body", in, h, w, SRMI_ACT_NONE, false, 1.f, &res, 1.f, nullptr, &o, nullptr)) return -1;
    }
    {
        const srmi_tensor_t in = tv(FE), o = tv(U1);
        if (ex->conv("conv_up1", in, 2 * h, 2 * w, SRMI_ACT_LRELU, true, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
    }
    Buf last = U1;
    if (up == 2) {
        const srmi_tensor_t in = tv(U1), o = tv(U2);
        if (ex->conv("conv_up2", in, hh, ww, SRMI_ACT_LRELU, true, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
        last = U2;
    }
    {
        const srmi_tensor_t in = tv(last), o = tv(U3);
        if (ex->conv("conv_hr", in, hh, ww, SRMI_ACT_LRELU, false, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
    }
    return ex->tail("conv_last", tv(U3), hh, ww, io->tail_mode, io->out, io->post_mat, io->post_bias,
                    io->lo, io->hi, io->scale, io->round);
}

// RRDBNet in w8a8 mode: the dense blocks on the int8 matrix cores.  The dense buffer holds int8 planes of 32
// channels (x in planes 0-1, x1..x4 in planes 2..5, each tensor with its calibrated scale); every block's last
// conv adds the half-precision trunk and writes the new trunk twice: quantised into the other dense buffer (the
// next block's conv input) and in f16 (the next residual), so quantisation noise never accumulates along the
// residual chain.  Everything outside the blocks runs as in the f16 plan.
static int run_rrdbnet_w8a8(srmi_model* m, Arena& A, Exec* ex, const IoSpec* io, int h, int w) {
    const auto& d = m->d;
    const int nf = d.n_feats, gc = d.grow_ch, TH = SRMI_F16;
    Buf D[2] = {A.take(nf + 4 * gc, h, w, SRMI_I8), A.take(nf + 4 * gc, h, w, SRMI_I8)};
    Buf R[2] = {A.take(nf, h, w, TH), A.take(nf, h, w, TH)};
    Buf X[2] = {A.take(nf, h, w, TH), A.take(nf, h, w, TH)};
    Buf F = A.take(nf, h, w, TH);
    Buf FE = A.take(nf, h, w, TH);
    const int up = d.scale == 4 ? 2 : 1;
    Buf U1 = A.take(nf, 2 * h, 2 * w, TH);
    Buf U2, U3;
    const int hh = h * d.scale, ww = w * d.scale;
    if (up == 2) U2 = A.take(nf, hh, ww, TH);
    U3 = A.take(nf, hh, ww, TH);
    if (!ex) return 0;
    auto qt = [&](const Buf& b, int c0, int g, int j) {
        srmi_tensor_t t = tv(b, c0);
        t.qscale = 1.0f / m->act_scale(g, j);
        return t;
    };
    {
        srmi_tensor_t o = qt(D[0], 0, 0, 0), o2 = tv(F);
        if (ex->first(io->in, io->in_kind, h, w, io->pre_mat, io->pre_bias, &o, &o2)) return -1;
    }
    int k = 0;
    char name[96];
    for (int i = 0; i < d.n_blocks; ++i) {
        const Buf& Rin = i == 0 ? F : R[i & 1];
        for (int r = 1; r <= 3; ++r) {
            const int g = 3 * i + r - 1;
            const srmi_tensor_t in = qt(D[k], 0, g, 0);
            for (int j = 1; j <= 4; ++j) {
                snprintf(name, sizeof(name), "body.%d.rdb%d.conv%d", i, r, j);
                const srmi_tensor_t o = qt(D[k], nf + (j - 1) * gc, g, j);
                if (ex->conv_q(name, in, h, w, SRMI_ACT_LRELU, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
            }
            snprintf(name, sizeof(name), "body.%d.rdb%d.conv5", i, r);
            const srmi_tensor_t res1 = r == 1 ? tv(Rin) : tv(X[r & 1]);
            const srmi_tensor_t o = qt(D[k ^ 1], 0, g + 1, 0);
            if (r < 3) {
                const srmi_tensor_t o2 = tv(X[(r + 1) & 1]);
                if (ex->conv_q(name, in, h, w, SRMI_ACT_NONE, 0.2f, &res1, 1.f, nullptr, &o, &o2)) return -1;
            } else {
                const srmi_tensor_t res2 = tv(Rin), o2 = tv(R[(i + 1) & 1]);
                if (ex->conv_q(name, in, h, w, SRMI_ACT_NONE, 0.2f, &res1, 0.2f, &res2, &o, &o2)) return -1;
            }
            k ^= 1;
        }
    }
    {
        const srmi_tensor_t in = tv(R[d.n_blocks & 1]), res = tv(F), o = tv(FE);
        if (ex->conv("conv_body", in, h, w, SRMI_ACT_NONE, false, 1.f, &res, 1.f, nullptr, &o, nullptr)) return -1;
    }
    {
        const srmi_tensor_t in = tv(FE), o = tv(U1);
        if (ex->conv("conv_up1", in, 2 * h, 2 * w, SRMI_ACT_LRELU, true, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
    }
    Buf last = U1;
    if (up == 2) {
        const srmi_tensor_t in = tv(U1), o = tv(U2);
        if (ex->conv("conv_up2", in, hh, ww, SRMI_ACT_LRELU, true, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
        last = U2;
    }
    {
        const srmi_tensor_t in = tv(last), o = tv(U3);
        if (ex->conv("conv_hr", in, hh, ww, SRMI_ACT_LRELU, false, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
    }
    return ex->tail("conv_last", tv(U3), hh, ww, io->tail_mode, io->out, io->post_mat, io->post_bias,
                    io->lo, io->hi, io->scale, io->round);
}

// Int8 weights of the dense blocks from the calibrated activation scales: the scale of each input tensor is
// folded into its weight columns, then every output channel is quantised symmetrically to [-127, 127]
// (per-channel weight scales, the usual w8a8 recipe; realesrgan/rknn/python/convert.py:198-230 leaves this to the
// vendor quantiser).
static int prepare_w8a8(srmi_model* m) {
    const auto& d = m->d;
    const int nf = d.n_feats, gc = d.grow_ch;
    for (void* p : m->dev_allocs_q) (void)hipFree(p);
    m->dev_allocs_q.clear();
    m->convs_q.clear();
    auto upload = [&](const void* host, size_t bytes, void** out) -> int {
        void* p = nullptr;
        SRMI_HIP(hipMalloc(&p, bytes));
        m->dev_allocs_q.push_back(p);
        SRMI_HIP(hipMemcpy(p, host, bytes, hipMemcpyHostToDevice));
        *out = p;
        return 0;
    };
    char name[96];
    for (int i = 0; i < d.n_blocks; ++i)
        for (int r = 1; r <= 3; ++r)
            for (int j = 1; j <= 5; ++j) {
                snprintf(name, sizeof(name), "body.%d.rdb%d.conv%d", i, r, j);
                const HostTensor& w = m->tensors[std::string(name) + ".weight"];
                const HostTensor& b = m->tensors[std::string(name) + ".bias"];
                const int g = 3 * i + r - 1, cout = (int)w.shape[0], cin = (int)w.shape[1];
                std::vector<float> wq(w.data.size()), ws(cout);
                for (int o = 0; o < cout; ++o) {
                    float amax = 0.f;
                    for (int c = 0; c < cin; ++c) {
                        const float sa = m->act_scale(g, c < nf ? 0 : (c - nf) / gc + 1);
                        for (int t = 0; t < 9; ++t) {
                            const float v = w.data[((size_t)o * cin + c) * 9 + t] * sa;
                            wq[((size_t)o * cin + c) * 9 + t] = v;
                            amax = fmaxf(amax, fabsf(v));
                        }
                    }
                    const float sw = amax > 0.f ? amax / 127.f : 1.f;
                    ws[o] = sw;
                    for (size_t e = (size_t)o * cin * 9; e < (size_t)(o + 1) * cin * 9; ++e) wq[e] = rintf(wq[e] / sw);
                }
                ConvW cw;
                cw.cout = cout; cw.cin = cin; cw.cout_tile = 32; cw.ps_r = 1;
                int nbias = 0;
                const size_t bytes = srmi_pack_conv3x3_bytes(SRMI_I8, cout, cin, 32, &nbias);
                if (!bytes) SRMI_FAIL("%s: shape [%d,%d,3,3] does not fit the int8 path (channels in multiples of 32)", name, cout, cin);
                std::vector<char> wp(bytes);
                std::vector<float> bp(nbias), sp(nbias, 0.f);
                if (srmi_pack_conv3x3(SRMI_I8, cout, cin, 32, 1, wq.data(), b.data.data(), wp.data(), bp.data())) return -1;
                for (int o = 0; o < cout; ++o) sp[o] = ws[o];
                if (upload(wp.data(), bytes, &cw.wpk)) return -1;
                if (upload(bp.data(), nbias * sizeof(float), (void**)&cw.bias)) return -1;
                if (upload(sp.data(), nbias * sizeof(float), (void**)&cw.wscale)) return -1;
                m->convs_q[name] = cw;
            }
    return 0;
}

// Upsampler + tail conv shared by EDSR and RCAN (edsr_model.py:50-77,105-108).
static int run_tail_stages(srmi_model* m, Arena& A, Exec* ex, const IoSpec* io, Buf cur, int h, int w) {
    const auto& d = m->d;
    int r;
    const int stages = upsampler_stages(d.scale, &r);
    char name[64];
    int ch = h, cw = w;
    for (int s = 0; s < stages; ++s) {
        Buf nxt = A.take(d.n_feats, ch * r, cw * r, m->cdt());
        if (ex) {
            snprintf(name, sizeof(name), "tail.0.%d", 2 * s);
            const srmi_tensor_t in = tv(cur), o = tv(nxt);
            if (ex->conv(name, in, ch, cw, SRMI_ACT_NONE, false, 1.f, nullptr, 1.f, nullptr, &o, nullptr)) return -1;
        }
        cur = nxt;
        ch *= r; cw *= r;
    }
    if (!ex) return 0;
    return ex->tail("tail.1", tv(cur), ch, cw, io->tail_mode, io->out, io->post_mat, io->post_bias,
                    io->lo, io->hi, io->scale, io->round);
}

static int run_edsr(srmi_model* m, Arena& A, Exec* ex, const IoSpec* io, int h, int w) {
    const auto& d = m->d;
    const int nf = d.n_feats, T = m->cdt(), TR = m->tdt();
    const bool split = m->split_trunk();
    // 64 features, f16 operands (the EDSR-baseline family): conv -> ReLU -> conv of a ResBlock (edsr_model.py:44-47) as
    // one kernel with the tensor between them in LDS (srmi_rcab_fused without its attention tail).  A block's output
    // x + res_scale * y is then never written by a pass of its own: the next block (or body.N) forms it while loading,
    // with a gate vector of ones, and stores it to the next trunk buffer on the way.
    const bool one_kernel = m->fuse_rcab && T == SRMI_F16 && (!split || TR == SRMI_F32) && nf == 64 && !m->mb_override &&
                            d.n_blocks > 0;
    Buf XT[2] = {A.take(nf, h, w, T), A.take(nf, h, w, T)};
    Buf XR[2];
    if (split) { XR[0] = A.take(nf, h, w, TR); XR[1] = A.take(nf, h, w, TR); }
    Buf Tm = A.take(nf, h, w, T);
    Buf H = A.take(nf, h, w, TR);
    Buf Yb[2];
    float* ones = nullptr;
    if (one_kernel) {
        Yb[0] = A.take(nf, h, w, T);
        Yb[1] = A.take(nf, h, w, T);
        ones = A.take_floats((size_t)A.ns * nf);
    }
    if (ex && one_kernel) {
        SRMI_HIP(hipMemsetD32Async((hipDeviceptr_t)ones, 0x3f800000, (size_t)A.ns * nf, ex->stream));   // 1.0f
        srmi_tensor_t o = tv(XT[0]), o2 = tv(H);
        if (ex->first(io->in, io->in_kind, h, w, io->pre_mat, io->pre_bias, &o, &o2)) return -1;
        char name[64], name2[64];
        int k = 0, kr = 0, yk = 0;
        bool trunk_in_xr = false;
        // the trunk before the pending block: the compute-dtype copy, or (split) the f32 one (the head's output at first)
        auto trunk_view = [&]() { return !split ? tv(XT[k]) : (trunk_in_xr ? tv(XR[kr]) : tv(H)); };
        for (int i = 0; i < d.n_blocks; ++i) {
            snprintf(name, sizeof(name), "body.%d.body.0", i);
            snprintf(name2, sizeof(name2), "body.%d.body.2", i);
            const bool pending = i > 0;
            const srmi_tensor_t in = pending ? trunk_view() : tv(XT[k]);
            const Exec::Gate gate{tv(Yb[yk]), ones, d.res_scale, split ? tv(XR[kr ^ 1]) : tv(XT[k ^ 1])};
            if (ex->rcab(name, name2, in, h, w, pending ? &gate : nullptr, ones, tv(Yb[yk ^ 1]), nullptr)) return -1;
            if (pending) {
                if (split) { kr ^= 1; trunk_in_xr = true; }
                else k ^= 1;
            }
            yk ^= 1;
        }
        // body.N consumes the last block's output the same way and adds the head's output (edsr_model.py:119-121)
        snprintf(name, sizeof(name), "body.%d", d.n_blocks);
        const srmi_tensor_t in = trunk_view(), res = tv(H), o1 = tv(XT[k ^ 1]);
        const Exec::Gate gate{tv(Yb[yk]), ones, d.res_scale, srmi_tensor_t{}};
        if (ex->conv(name, in, h, w, SRMI_ACT_NONE, false, 1.f, &res, 1.f, nullptr, &o1, nullptr, nullptr,
                     gate_rows_per_wave(m, h), &gate)) return -1;
        return run_tail_stages(m, A, ex, io, XT[k ^ 1], h, w);
    }
    if (ex) {
        srmi_tensor_t o = tv(XT[0]), o2 = tv(H);
        if (ex->first(io->in, io->in_kind, h, w, io->pre_mat, io->pre_bias, &o, &o2)) return -1;
        char name[64];
        int k = 0;
        for (int i = 0; i < d.n_blocks; ++i) {
            snprintf(name, sizeof(name), "body.%d.body.0", i);
            const srmi_tensor_t in = tv(XT[k]), t = tv(Tm);
            if (ex->conv(name, in, h, w, SRMI_ACT_RELU, false, 1.f, nullptr, 1.f, nullptr, &t, nullptr)) return -1;
            snprintf(name, sizeof(name), "body.%d.body.2", i);
            // residual: trunk value before this block (edsr_model.py:44-47)
            srmi_tensor_t res = split ? (i == 0 ? tv(H) : tv(XR[(i - 1) & 1])) : tv(XT[k]);
            const srmi_tensor_t o1 = tv(XT[k ^ 1]);
            srmi_tensor_t o2b = split ? tv(XR[i & 1]) : srmi_tensor_t{};
            if (ex->conv(name, t, h, w, SRMI_ACT_NONE, false, d.res_scale, &res, 1.f, nullptr, &o1, split ? &o2b : nullptr)) return -1;
            k ^= 1;
        }
        snprintf(name, sizeof(name), "body.%d", d.n_blocks);
        const srmi_tensor_t in = tv(XT[k]), res = tv(H), o1 = tv(XT[k ^ 1]);
        if (ex->conv(name, in, h, w, SRMI_ACT_NONE, false, 1.f, &res, 1.f, nullptr, &o1, nullptr)) return -1;
        return run_tail_stages(m, A, ex, io, XT[k ^ 1], h, w);
    }
    return run_tail_stages(m, A, nullptr, io, XT[0], h, w);
}

static int run_rcan(srmi_model* m, Arena& A, Exec* ex, const IoSpec* io, int h, int w) {
    const auto& d = m->d;
    const int nf = d.n_feats, T = m->cdt(), TR = m->tdt();
    const bool split = m->split_trunk();
    const int cr = nf / d.reduction;
    Buf XT[2] = {A.take(nf, h, w, T), A.take(nf, h, w, T)};
    Buf XR[2];
    if (split) { XR[0] = A.take(nf, h, w, TR); XR[1] = A.take(nf, h, w, TR); }
    Buf Tm = A.take(nf, h, w, T);
    // A block's output x + res_scale * (y * s) is never written by a pass of its own (see below): the convolution that
    // consumes it forms it while loading.
    const bool fused = m->fuse_attention && (T == SRMI_F16 || T == SRMI_BF16) && nf <= 256 &&
                       (!split || gate_rows_per_wave(m, h) == 2);
    // ... and with f16 operands and 64 features both convolutions of a block run in one kernel (the trunk in f16, or in
    // f32 beside the f16 operands), which reads the previous block's body while it writes its own: two body buffers
    const bool rcab = fused && m->fuse_rcab && T == SRMI_F16 && (!split || TR == SRMI_F32) && nf == 64 && !m->mb_override;
    Buf Yb[2] = {A.take(nf, h, w, T), rcab ? A.take(nf, h, w, T) : Buf()};
    int yk = 0;   // Yb[yk] holds the latest block body
    Buf H = A.take(nf, h, w, TR);
    Buf GI[2] = {A.take(nf, h, w, TR), A.take(nf, h, w, TR)};
    const int ns = A.ns;
    // per-image channel sums (64-bit fixed point) and arrival counters of the attention blocks: zero whenever no
    // conv is running on them (cleared here once, then by the workgroup that consumes them)
    uint64_t* ca_sums = (uint64_t*)A.take_floats((size_t)ns * SRMI_CA_SLOTS * nf * 2);
    uint32_t* ca_count = (uint32_t*)A.take_floats((size_t)ns);
    float* scale = A.take_floats((size_t)ns * nf);
    if (ex) {
        // (the attention scales too: a block with a plain input passes them as its gate vector with a factor of zero,
        // which wants finite numbers)
        SRMI_HIP(hipMemsetAsync(ca_sums, 0, (size_t)((char*)(scale + (size_t)ns * nf) - (char*)ca_sums), ex->stream));
        srmi_tensor_t o = tv(XT[0]), o2 = tv(H);
        if (ex->first(io->in, io->in_kind, h, w, io->pre_mat, io->pre_bias, &o, &o2)) return -1;
        char name[96];
        int k = 0;    // XT[k] holds the trunk in the compute dtype
        int kr = 0;   // XR[kr] holds it in the trunk dtype (split mode), when trunk_in_xr
        bool trunk_in_xr = false;
        int ca_index = 0;
        // (fused:) the convolution that consumes a block's output (the next block's first, or the group's last) forms
        // it while loading and, where a later residual needs it, stores it on the way (in the trunk's own precision:
        // half, or f32 when split).
        for (int g = 0; g < d.n_groups; ++g) {
            const Buf& Gin = g == 0 ? H : GI[g & 1];
            bool pending = false;  // trunk + res_scale * (Y * scale) is the block's output, not yet stored
            // where the trunk before the pending block lives: the compute-dtype copy, or (split) the f32 one
            auto trunk_view = [&]() { return !split ? tv(XT[k]) : (trunk_in_xr ? tv(XR[kr]) : tv(Gin)); };
            for (int b = 0; b < d.n_blocks; ++b) {
                snprintf(name, sizeof(name), "body.%d.body.%d.body.0", g, b);
                if (rcab) {
                    char name2[96];
                    snprintf(name2, sizeof(name2), "body.%d.body.%d.body.2", g, b);
                    const Exec::Attention att{&m->cas[ca_index++], cr, ca_sums, ca_count, scale};
                    // a group's first block reads the trunk's compute-dtype copy; the others form their input from the
                    // trunk (in its own precision) and the previous block's body, and store it on the way
                    const srmi_tensor_t in = pending ? trunk_view() : tv(XT[k]);
                    const Exec::Gate gate{tv(Yb[yk]), scale, d.res_scale, split ? tv(XR[kr ^ 1]) : tv(XT[k ^ 1])};
                    if (ex->rcab(name, name2, in, h, w, pending ? &gate : nullptr, scale, tv(Yb[yk ^ 1]), &att)) return -1;
                    if (pending) {
                        if (split) { kr ^= 1; trunk_in_xr = true; }
                        else k ^= 1;
                    }
                    yk ^= 1;
                    pending = true;
                    continue;
                }
                const srmi_tensor_t t = tv(Tm), y = tv(Yb[yk]);
                if (pending) {
                    const srmi_tensor_t in = trunk_view();
                    const Exec::Gate gate{y, scale, d.res_scale, split ? tv(XR[kr ^ 1]) : tv(XT[k ^ 1])};
                    if (ex->conv(name, in, h, w, SRMI_ACT_RELU, false, 1.f, nullptr, 1.f, nullptr, &t, nullptr, nullptr,
                                 gate_rows_per_wave(m, h), &gate)) return -1;
                    if (split) { kr ^= 1; trunk_in_xr = true; }
                    else k ^= 1;
                    pending = false;
                } else {
                    const srmi_tensor_t in = tv(XT[k]);
                    if (ex->conv(name, in, h, w, SRMI_ACT_RELU, false, 1.f, nullptr, 1.f, nullptr, &t, nullptr)) return -1;
                }
                snprintf(name, sizeof(name), "body.%d.body.%d.body.2", g, b);
                // body.2 also pools its output and, in its last workgroup, runs the squeeze / excite layers
                const Exec::Attention att{&m->cas[ca_index++], cr, ca_sums, ca_count, scale};
                if (ex->conv(name, t, h, w, SRMI_ACT_NONE, false, 1.f, nullptr, 1.f, nullptr, &y, nullptr, nullptr, 0, nullptr, &att)) return -1;
                if (fused) {
                    pending = true;
                    continue;
                }
                // x + res_scale * (y * s)   (rcan_model.py:94-97)
                const srmi_tensor_t xres = trunk_view();
                const srmi_tensor_t o1 = tv(XT[k ^ 1]);
                srmi_tensor_t o2b = split ? tv(XR[kr ^ 1]) : srmi_tensor_t{};
                if (srmi_ca_apply(&y, &xres, scale, d.res_scale, ns, h, w, nf, &o1, split ? &o2b : nullptr, ex->stream)) return -1;
                k ^= 1;
                if (split) { kr ^= 1; trunk_in_xr = true; }
            }
            // group tail conv + group input (rcan_model.py:109-112); its output is the next group's input
            snprintf(name, sizeof(name), "body.%d.body.%d", g, d.n_blocks);
            const srmi_tensor_t res = tv(Gin), o1 = tv(XT[k ^ 1]), og = tv(GI[(g + 1) & 1]);
            if (pending) {
                const srmi_tensor_t in = trunk_view();
                const Exec::Gate gate{tv(Yb[yk]), scale, d.res_scale, srmi_tensor_t{}};
                if (ex->conv(name, in, h, w, SRMI_ACT_NONE, false, 1.f, &res, 1.f, nullptr, &o1, &og, nullptr,
                             gate_rows_per_wave(m, h), &gate)) return -1;
            } else {
                const srmi_tensor_t in = tv(XT[k]);
                if (ex->conv(name, in, h, w, SRMI_ACT_NONE, false, 1.f, &res, 1.f, nullptr, &o1, &og)) return -1;
            }
            k ^= 1;
            trunk_in_xr = false;  // the trunk in trunk dtype is now GI[(g+1)&1]
        }
        snprintf(name, sizeof(name), "body.%d", d.n_groups);
        const srmi_tensor_t in = tv(XT[k]), res = tv(H), o1 = tv(XT[k ^ 1]);
        if (ex->conv(name, in, h, w, SRMI_ACT_NONE, false, 1.f, &res, 1.f, nullptr, &o1, nullptr)) return -1;
        return run_tail_stages(m, A, ex, io, XT[k ^ 1], h, w);
    }
    return run_tail_stages(m, A, nullptr, io, XT[0], h, w);
}

static int run_plan(srmi_model* m, Arena& A, Exec* ex, const IoSpec* io, int h, int w) {
    switch (m->d.arch) {
        case SRMI_ARCH_RRDBNET:
            return (m->w8a8 && !m->calibrating) ? run_rrdbnet_w8a8(m, A, ex, io, h, w) : run_rrdbnet(m, A, ex, io, h, w);
        case SRMI_ARCH_EDSR: return run_edsr(m, A, ex, io, h, w);
        case SRMI_ARCH_RCAN: return run_rcan(m, A, ex, io, h, w);
    }
    SRMI_FAIL("unknown architecture %d", m->d.arch);
}

// Address of this device's copy of `scratch` (the device must be current).  Every GPU holds its own copy
// at its own address, so the pointer is looked up once per device and kept.
static int static_workspace(int device, char** out) {
    static std::mutex mu;
    static char* ptr[64] = {};
    if (device < 0 || device >= 64) SRMI_FAIL("static workspace: device %d out of range", device);
    std::lock_guard<std::mutex> lock(mu);
    if (!ptr[device]) {
        void* p = nullptr;
        SRMI_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) SRMI_FAIL("static workspace: hipGetSymbolAddress returned a null address");
        ptr[device] = (char*)p;
    }
    *out = ptr[device];
    return 0;
}

// RAII: make `device` current for a call and restore the caller's device afterwards
struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int enter(int device) {
        SRMI_HIP(hipGetDevice(&prev));
        if (prev != device) {
            SRMI_HIP(hipSetDevice(device));
            switched = true;
        }
        return 0;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// Largest workspace a forward of (n, h, w) needs with the model's current sweep and lane settings.
static int plan_bytes(srmi_model* m, int n, int h, int w, int max_lanes, int* ns_out, int* lanes_out, size_t* lane_bytes) {
    const int ns = n < m->sweep ? n : m->sweep;
    Arena probe;
    probe.ns = ns;
    IoSpec io;
    memset(&io, 0, sizeof(io));
    if (run_plan(m, probe, nullptr, &io, h, w)) return -1;
    const int chunks = (n + ns - 1) / ns;
    *ns_out = ns;
    *lanes_out = chunks < max_lanes ? chunks : max_lanes;
    *lane_bytes = (probe.off + 255) & ~(size_t)255;
    return 0;
}

// Grows the model's own workspace and creates its lane streams / events (the only allocations of the
// forward path; srmi_model_reserve runs them ahead of time).  Caller holds call_mutex.
static int ensure_resources(srmi_model* m, size_t bytes, int lanes, bool lane_streams_always = false) {
    if (bytes > m->ws_bytes) {
        // wait for work that may still use the old buffer first
        SRMI_HIP(hipDeviceSynchronize());
        if (m->ws) SRMI_HIP(hipFree(m->ws));
        m->ws = nullptr;
        m->ws_bytes = 0;
        SRMI_HIP(hipMalloc((void**)&m->ws, bytes));
        m->ws_bytes = bytes;
    }
    if (!m->last_done) SRMI_HIP(hipEventCreateWithFlags(&m->last_done, hipEventDisableTiming));
    if (lanes > 1 || lane_streams_always) {
        if (!m->fork) SRMI_HIP(hipEventCreateWithFlags(&m->fork, hipEventDisableTiming));
        for (int l = 0; l < lanes; ++l) {
            if (!m->lane_stream[l]) SRMI_HIP(hipStreamCreateWithFlags(&m->lane_stream[l], hipStreamNonBlocking));
            if (!m->lane_done[l]) SRMI_HIP(hipEventCreateWithFlags(&m->lane_done[l], hipEventDisableTiming));
        }
    }
    return 0;
}

static bool hip_failed(hipError_t e, const char* what) {
    if (e == hipSuccess) return false;
    set_error("%s failed: %s", what, hipGetErrorString(e));
    return true;
}

// static_ws: run in the per-device static array on the caller's stream only, without any allocation,
// stream or event creation (srmi_model_forward_u8_noalloc); otherwise in the model's own workspace.
static int forward(srmi_model* m, IoSpec io, int n, int h, int w, hipStream_t stream, bool static_ws = false) {
    if (!m) SRMI_FAIL("forward: null model");
    if (!m->finalized) SRMI_FAIL("forward: srmi_model_finalize has not been called");
    if (!io.in || !io.out) SRMI_FAIL("forward: null input or output pointer");
    if (n <= 0 || h <= 0 || w <= 0) SRMI_FAIL("forward: bad shape n=%d h=%d w=%d", n, h, w);
    DeviceGuard guard;
    if (guard.enter(m->d.device)) return -1;
    std::lock_guard<std::mutex> lock(m->call_mutex);
    int ns = 1, lanes = 1;
    size_t lane_bytes = 0;
    if (plan_bytes(m, n, h, w, static_ws ? 1 : m->lanes, &ns, &lanes, &lane_bytes)) return -1;
    char* ws = nullptr;
    // A call that is being captured into a HIP graph must not wait for, or record, events that live outside the
    // capture (`last_done` was recorded by an earlier eager call): such a call is ordered by the graph alone, and
    // the caller keeps other work on this model away from the graph's replays.
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (stream && hipStreamIsCapturing(stream, &capture) != hipSuccess) {
        (void)hipGetLastError();
        capture = hipStreamCaptureStatusNone;
    }
    const bool capturing = capture != hipStreamCaptureStatusNone;
    if (static_ws) {
        if (!m->static_ws) SRMI_FAIL("forward: the static workspace of device %d was not resolved", m->d.device);
        if (lane_bytes > SRMI_STATIC_WS_BYTES)
            SRMI_FAIL("srmi_model_forward_u8_noalloc: a %dx%d input needs %zu bytes of workspace, the static workspace "
                      "holds %llu (use srmi_model_forward_u8, or smaller tiles)", w, h, lane_bytes,
                      (unsigned long long)SRMI_STATIC_WS_BYTES);
        ws = m->static_ws;
    } else {
        if (capturing && (lane_bytes * lanes > m->ws_bytes || !m->last_done || (lanes > 1 && !m->lane_stream[lanes - 1])))
            SRMI_FAIL("forward: the stream is being captured and this call would have to allocate its workspace or lanes: "
                      "call srmi_model_reserve(n, h, w) (or run one eager forward of this shape) before the capture");
        if (ensure_resources(m, lane_bytes * lanes, lanes)) return -1;
        ws = m->ws;
        // order this call behind the previous one on this model, whatever stream that ran on
        if (m->has_last_done && !capturing) SRMI_HIP(hipStreamWaitEvent(stream, m->last_done, 0));
    }
    // Lanes fork from the caller's stream and join it again on EVERY way out of this function, failed HIP calls
    // included: the caller's stream is never left unordered against work already issued on a lane.
    struct LaneJoin {
        srmi_model* m;
        hipStream_t stream;
        int forked = 0;
        ~LaneJoin() {
            for (int l = 0; l < forked; ++l) {
                if (hipEventRecord(m->lane_done[l], m->lane_stream[l]) == hipSuccess)
                    (void)hipStreamWaitEvent(stream, m->lane_done[l], 0);
            }
        }
    } join{m, stream};
    if (lanes > 1) {
        SRMI_HIP(hipEventRecord(m->fork, stream));
        for (int l = 0; l < lanes; ++l) {
            SRMI_HIP(hipStreamWaitEvent(m->lane_stream[l], m->fork, 0));
            join.forked = l + 1;
        }
    }
    const size_t in_img = io.in_kind == SRMI_IN_U8_NHWC ? (size_t)h * w * 3 : (size_t)h * w * 3 * sizeof(float);
    const size_t hh = (size_t)h * m->d.scale, ww = (size_t)w * m->d.scale;
    const size_t out_img = io.tail_mode == SRMI_TAIL_U8_NHWC ? hh * ww * 3 : hh * ww * 3 * sizeof(float);
    const char* in0 = (const char*)io.in;
    char* out0 = (char*)io.out;
    int rc = 0;
    for (int i = 0, chunk = 0; i < n && !rc; i += ns, ++chunk) {
        const int lane = chunk % lanes;
        Exec ex;
        ex.m = m;
        ex.stream = lanes > 1 ? m->lane_stream[lane] : stream;
        ex.ns = (n - i) < ns ? (n - i) : ns;
        Arena A;
        A.base = ws + (size_t)lane * lane_bytes;
        A.ns = ns;
        io.in = in0 + (size_t)i * in_img;
        io.out = out0 + (size_t)i * out_img;
        rc = run_plan(m, A, &ex, &io, h, w);
    }
    if (lanes > 1) {
        // join now (the guard then has nothing left to do), so that `last_done` below lies behind the lanes
        const int forked = join.forked;
        join.forked = 0;
        for (int l = 0; l < forked; ++l) {
            if (hipEventRecord(m->lane_done[l], m->lane_stream[l]) != hipSuccess ||
                hipStreamWaitEvent(stream, m->lane_done[l], 0) != hipSuccess) {
                (void)hipGetLastError();
                set_error("forward: joining lane %d failed", l);
                rc = -1;
            }
        }
    }
    if (!static_ws && !capturing) {
        SRMI_HIP(hipEventRecord(m->last_done, stream));
        m->has_last_done = true;
    }
    return rc;
}

// Pre/post-processing of the runners' u8 path for this model.
static void u8_io(const srmi_model* m, const uint8_t* in, uint8_t* out, int round_mode, IoSpec* iop) {
    IoSpec& io = *iop;
    memset(&io, 0, sizeof(io));
    io.in_kind = SRMI_IN_U8_NHWC;
    io.in = in;
    io.tail_mode = SRMI_TAIL_U8_NHWC;
    io.out = out;
    io.round = round_mode;
    memcpy(io.pre_mat, kIdentity, sizeof(io.pre_mat));
    memcpy(io.post_mat, kIdentity, sizeof(io.post_mat));
    if (m->d.arch == SRMI_ARCH_RRDBNET) {
        // realesrgan.cpp:35,44 (p * (1/255)) and :124-126 (clamp(v,0,1)*255)
        const float s = 1.0f / 255.0f;
        io.pre_mat[0] = io.pre_mat[4] = io.pre_mat[8] = s;
        io.lo = 0.f; io.hi = 1.f; io.scale = 255.f;
    } else {
        // edsr.cpp:36-37 (p - mean*range), :116-118 (clamp(v + mean*range, 0, 255)).  With a full
        // model the shift comes from the checkpoint's MeanShift tensors, with a core model the
        // caller's tensors are used as they are (identity, no shift).
        if (m->d.with_meanshift) {
            memcpy(io.pre_mat, m->sub_mat, sizeof(io.pre_mat));
            memcpy(io.post_mat, m->add_mat, sizeof(io.post_mat));
            for (int i = 0; i < 3; ++i) { io.pre_bias[i] = m->sub_bias[i]; io.post_bias[i] = m->add_bias[i]; }
        } else if (m->has_core_shift) {
            for (int i = 0; i < 3; ++i) { io.pre_bias[i] = -m->core_shift[i]; io.post_bias[i] = m->core_shift[i]; }
        }
        io.lo = 0.f; io.hi = 255.f; io.scale = 1.f;
    }
}

// ---- host-buffer path ----------------------------------------------------------------------------
static int ensure_host_slots(srmi_model* m, int slots, size_t in_bytes, size_t out_bytes) {
    for (int l = 0; l < slots; ++l) {
        auto& s = m->slot[l];
        if (!s.done) SRMI_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        if (in_bytes > s.in_cap) {
            if (s.pin_in) SRMI_HIP(hipHostFree(s.pin_in));
            if (s.dev_in) SRMI_HIP(hipFree(s.dev_in));
            s.pin_in = s.dev_in = nullptr;
            s.in_cap = 0;
            SRMI_HIP(hipHostMalloc((void**)&s.pin_in, in_bytes, hipHostMallocDefault));
            SRMI_HIP(hipMalloc((void**)&s.dev_in, in_bytes));
            s.in_cap = in_bytes;
        }
        if (out_bytes > s.out_cap) {
            if (s.pin_out) SRMI_HIP(hipHostFree(s.pin_out));
            if (s.dev_out) SRMI_HIP(hipFree(s.dev_out));
            s.pin_out = s.dev_out = nullptr;
            s.out_cap = 0;
            SRMI_HIP(hipHostMalloc((void**)&s.pin_out, out_bytes, hipHostMallocDefault));
            SRMI_HIP(hipMalloc((void**)&s.dev_out, out_bytes));
            s.out_cap = out_bytes;
        }
    }
    return 0;
}

static void free_host_slots(srmi_model* m) {
    for (auto& s : m->slot) {
        if (s.pin_in) (void)hipHostFree(s.pin_in);
        if (s.pin_out) (void)hipHostFree(s.pin_out);
        if (s.dev_in) (void)hipFree(s.dev_in);
        if (s.dev_out) (void)hipFree(s.dev_out);
        if (s.done) (void)hipEventDestroy(s.done);
        s = srmi_model::HostSlot();
    }
}

// Frames in host memory -> frames in host memory, synchronous.  Frame i runs on lane i % lanes:
// memcpy into the lane's pinned buffer, then on the lane's stream H2D -> network -> D2H into the lane's
// pinned output; while the GPU works on the other lanes' frames the calling thread copies finished
// frames out of the pinned buffers.  Copies therefore overlap compute on both sides.
static int forward_u8_host(srmi_model* m, const uint8_t* in, uint8_t* out, int n, int h, int w, int round_mode) {
    if (!m->finalized) SRMI_FAIL("srmi_model_forward_u8_host: srmi_model_finalize has not been called");
    if (!in || !out) SRMI_FAIL("srmi_model_forward_u8_host: null input or output pointer");
    if (n <= 0 || h <= 0 || w <= 0) SRMI_FAIL("srmi_model_forward_u8_host: bad shape n=%d h=%d w=%d", n, h, w);
    DeviceGuard guard;
    if (guard.enter(m->d.device)) return -1;
    std::lock_guard<std::mutex> lock(m->call_mutex);
    const int lanes = n < m->lanes ? n : m->lanes;
    // one frame per sweep on this path: a lane holds exactly one frame
    Arena probe;
    probe.ns = 1;
    IoSpec io;
    u8_io(m, nullptr, nullptr, round_mode, &io);
    if (run_plan(m, probe, nullptr, &io, h, w)) return -1;
    const size_t lane_bytes = (probe.off + 255) & ~(size_t)255;
    if (ensure_resources(m, lane_bytes * lanes, lanes, true)) return -1;
    const size_t in_bytes = (size_t)h * w * 3;
    const size_t out_bytes = in_bytes * m->d.scale * m->d.scale;
    if (ensure_host_slots(m, lanes, in_bytes, out_bytes)) return -1;
    if (m->has_last_done) SRMI_HIP(hipEventSynchronize(m->last_done));
    int rc = 0;
    auto drain = [&](int frame) -> int {
        auto& s = m->slot[frame % lanes];
        SRMI_HIP(hipEventSynchronize(s.done));
        memcpy(out + (size_t)frame * out_bytes, s.pin_out, out_bytes);
        return 0;
    };
    int issued = 0;
    for (int i = 0; i < n && !rc; ++i) {
        const int l = i % lanes;
        auto& s = m->slot[l];
        if (i >= lanes && drain(i - lanes)) { rc = -1; break; }
        memcpy(s.pin_in, in + (size_t)i * in_bytes, in_bytes);
        hipStream_t st = m->lane_stream[l];
        // failures inside the loop leave through the drain below: frames already issued still read and write
        // the pinned buffers, and the next call would overwrite them
        if (hip_failed(hipMemcpyAsync(s.dev_in, s.pin_in, in_bytes, hipMemcpyHostToDevice, st), "hipMemcpyAsync (input frame)")) { rc = -1; break; }
        Exec ex;
        ex.m = m;
        ex.stream = st;
        ex.ns = 1;
        Arena A;
        A.base = m->ws + (size_t)l * lane_bytes;
        A.ns = 1;
        io.in = s.dev_in;
        io.out = s.dev_out;
        rc = run_plan(m, A, &ex, &io, h, w);
        if (rc) break;
        if (hip_failed(hipMemcpyAsync(s.pin_out, s.dev_out, out_bytes, hipMemcpyDeviceToHost, st), "hipMemcpyAsync (output frame)") ||
            hip_failed(hipEventRecord(s.done, st), "hipEventRecord")) {
            // the frame's completion cannot be waited for through its event: wait for the whole device instead
            (void)hipDeviceSynchronize();
            rc = -1;
            break;
        }
        issued = i + 1;
    }
    // frames issued - lanes .. issued - 1 are still in their slots
    for (int f = issued > lanes ? issued - lanes : 0; f < issued; ++f)
        if (drain(f)) rc = -1;
    if (rc) (void)hipDeviceSynchronize();
    return rc;
}

}  // namespace srmi

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
extern "C" int srmi_model_create(const srmi_model_desc_t* desc, srmi_model_t** out) {
    if (!desc || !out) SRMI_FAIL("srmi_model_create: null argument");
    const srmi_model_desc_t& d = *desc;
    if (d.dtype != SRMI_F32 && d.dtype != SRMI_F16 && d.dtype != SRMI_BF16) SRMI_FAIL("srmi_model_create: bad dtype %d", d.dtype);
    if (d.n_feats <= 0 || d.n_feats % 32) SRMI_FAIL("srmi_model_create: n_feats=%d must be a positive multiple of 32", d.n_feats);
    if (d.n_blocks <= 0) SRMI_FAIL("srmi_model_create: n_blocks=%d", d.n_blocks);
    if (d.arch == SRMI_ARCH_RRDBNET) {
        // realesrgan_model.py:82-88
        if (d.scale != 2 && d.scale != 4) SRMI_FAIL("Unsupported scale: %d. Supported scales: 2, 4", d.scale);
        if (d.grow_ch <= 0 || d.grow_ch % 32) SRMI_FAIL("srmi_model_create: num_grow_ch=%d must be a positive multiple of 32", d.grow_ch);
    } else if (d.arch == SRMI_ARCH_EDSR || d.arch == SRMI_ARCH_RCAN) {
        // edsr_model.py:54-75: powers of two, or 3
        const bool pow2 = d.scale >= 2 && (d.scale & (d.scale - 1)) == 0;
        if (!pow2 && d.scale != 3) SRMI_FAIL("scale %d is not supported", d.scale);
        if (d.arch == SRMI_ARCH_RCAN) {
            if (d.n_groups <= 0) SRMI_FAIL("srmi_model_create: n_resgroups=%d", d.n_groups);
            if (d.reduction <= 0 || d.n_feats % d.reduction) SRMI_FAIL("srmi_model_create: reduction=%d does not divide n_feats", d.reduction);
        }
    } else {
        SRMI_FAIL("srmi_model_create: unknown architecture %d", d.arch);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        (void)hipGetLastError();
        SRMI_FAIL("srmi_model_create: no HIP device available (this library has no CPU path)");
    }
    if (d.device < 0 || d.device >= ndev) SRMI_FAIL("srmi_model_create: device %d out of range (%d devices)", d.device, ndev);
    srmi_model* m = new srmi_model();
    m->d = d;
    if (m->d.dtype == SRMI_F32) m->d.trunk_f32 = 0;
    for (int i = 0; i < 9; ++i) m->sub_mat[i] = m->add_mat[i] = kIdentity[i];
    for (int i = 0; i < 3; ++i) m->sub_bias[i] = m->add_bias[i] = 0.f;
#ifdef SRMI_DIAG
    const char* e = getenv("SRMI_ROWS_PER_WAVE");
    if (e && atoi(e) >= 2 && atoi(e) <= 4) m->mb_override = atoi(e);
#endif
    build_keys(m);
    *out = m;
    return 0;
}

extern "C" void srmi_model_destroy(srmi_model_t* m) {
    if (!m) return;
    (void)hipSetDevice(m->d.device);
    (void)hipDeviceSynchronize();
    for (void* p : m->dev_allocs) (void)hipFree(p);
    if (m->ws) (void)hipFree(m->ws);
    for (int l = 0; l < srmi_model::kMaxLanes; ++l) {
        if (m->lane_done[l]) (void)hipEventDestroy(m->lane_done[l]);
        if (m->lane_stream[l]) (void)hipStreamDestroy(m->lane_stream[l]);
    }
    if (m->fork) (void)hipEventDestroy(m->fork);
    if (m->last_done) (void)hipEventDestroy(m->last_done);
    free_host_slots(m);
    for (void* p : m->dev_allocs_q) (void)hipFree(p);
    if (m->calib_dev) (void)hipFree(m->calib_dev);
    delete m;
}

extern "C" int srmi_model_num_keys(const srmi_model_t* m) { return m ? (int)m->keys.size() : 0; }

extern "C" const char* srmi_model_key(const srmi_model_t* m, int i) {
    if (!m || i < 0 || i >= (int)m->keys.size()) return nullptr;
    return m->keys[i].c_str();
}

extern "C" int srmi_model_set_tensor(srmi_model_t* m, const char* key, const float* data,
                                     const int64_t* shape, int ndim) {
    if (!m || !key || !data || !shape) SRMI_FAIL("srmi_model_set_tensor: null argument");
    auto it = m->tensors.find(key);
    if (it == m->tensors.end()) SRMI_FAIL("unexpected key \"%s\" in state_dict", key);
    HostTensor& t = it->second;
    std::vector<int64_t> s(shape, shape + ndim);
    if (s != t.shape) {
        std::string got, want;
        for (auto v : s) got += std::to_string(v) + ",";
        for (auto v : t.shape) want += std::to_string(v) + ",";
        SRMI_FAIL("size mismatch for %s: checkpoint shape (%s) vs model shape (%s)", key, got.c_str(), want.c_str());
    }
    t.data.assign(data, data + numel(s));
    t.set = true;
    m->finalized = false;
    return 0;
}

extern "C" int srmi_model_finalize(srmi_model_t* m) {
    if (!m) SRMI_FAIL("srmi_model_finalize: null model");
    for (const auto& k : m->keys)
        if (!m->tensors[k].set) SRMI_FAIL("missing key \"%s\" in state_dict", k.c_str());
    SRMI_HIP(hipSetDevice(m->d.device));
    SRMI_HIP(hipDeviceSynchronize());
    for (void* p : m->dev_allocs) (void)hipFree(p);
    m->dev_allocs.clear();
    m->convs.clear();
    m->cas.clear();
    // new weights: an earlier calibration no longer applies
    m->calibrated = m->w8a8 = m->calibrating = false;
    const auto& d = m->d;
    if (d.arch == SRMI_ARCH_RRDBNET) {
        if (upload_first(m, "conv_first")) return -1;
        for (const auto& k : m->keys) {
            if (k.size() < 7 || k.compare(k.size() - 7, 7, ".weight")) continue;
            const std::string name = k.substr(0, k.size() - 7);
            if (name == "conv_first") continue;
            if (upload_conv(m, name, 1)) return -1;
        }
    } else {
        if (d.with_meanshift) {
            // MeanShift is a 1x1 conv stored in the checkpoint: general 3x3 matrix + bias
            memcpy(m->sub_mat, m->tensors["sub_mean.weight"].data.data(), 9 * sizeof(float));
            memcpy(m->sub_bias, m->tensors["sub_mean.bias"].data.data(), 3 * sizeof(float));
            memcpy(m->add_mat, m->tensors["add_mean.weight"].data.data(), 9 * sizeof(float));
            memcpy(m->add_bias, m->tensors["add_mean.bias"].data.data(), 3 * sizeof(float));
        }
        if (upload_first(m, "head.0")) return -1;
        int r;
        upsampler_stages(d.scale, &r);
        for (const auto& k : m->keys) {
            if (k.size() < 7 || k.compare(k.size() - 7, 7, ".weight")) continue;
            const std::string name = k.substr(0, k.size() - 7);
            if (name == "head.0" || name == "sub_mean" || name == "add_mean") continue;
            if (name.find("conv_du") != std::string::npos) continue;
            const bool is_up = name.compare(0, 7, "tail.0.") == 0;
            if (upload_conv(m, name, is_up ? r : 1)) return -1;
        }
        if (d.arch == SRMI_ARCH_RCAN) {
            char name[128];
            for (int g = 0; g < d.n_groups; ++g)
                for (int b = 0; b < d.n_blocks; ++b) {
                    CaW ca;
                    snprintf(name, sizeof(name), "body.%d.body.%d.body.3.conv_du.", g, b);
                    const std::string p = name;
                    const HostTensor& w1 = m->tensors[p + "0.weight"];
                    const HostTensor& b1 = m->tensors[p + "0.bias"];
                    const HostTensor& w2 = m->tensors[p + "2.weight"];
                    const HostTensor& b2 = m->tensors[p + "2.bias"];
                    if (dev_upload(m, w1.data.data(), w1.data.size() * 4, (void**)&ca.w1)) return -1;
                    if (dev_upload(m, b1.data.data(), b1.data.size() * 4, (void**)&ca.b1)) return -1;
                    if (dev_upload(m, w2.data.data(), w2.data.size() * 4, (void**)&ca.w2)) return -1;
                    if (dev_upload(m, b2.data.data(), b2.data.size() * 4, (void**)&ca.b2)) return -1;
                    m->cas.push_back(ca);
                }
        }
    }
    // set-up for the allocation-free entry point: this device's address of the static workspace, and
    // the per-device kernel attributes, so that nothing but launches is left for the call itself
    if (static_workspace(d.device, &m->static_ws)) return -1;
    if (srmi_conv3x3_prepare_device() || srmi_rcab_prepare_device()) return -1;
    m->finalized = true;
    return 0;
}

extern "C" int srmi_model_forward_f32(srmi_model_t* m, const float* in, float* out, int n, int h, int w,
                                      void* hip_stream) {
    if (!m) SRMI_FAIL("srmi_model_forward_f32: null model");
    IoSpec io;
    memset(&io, 0, sizeof(io));
    io.in_kind = SRMI_IN_F32_NCHW;
    io.in = in;
    io.tail_mode = SRMI_TAIL_F32_NCHW;
    io.out = out;
    const bool ms = m->d.arch != SRMI_ARCH_RRDBNET && m->d.with_meanshift;
    memcpy(io.pre_mat, ms ? m->sub_mat : kIdentity, sizeof(io.pre_mat));
    memcpy(io.post_mat, ms ? m->add_mat : kIdentity, sizeof(io.post_mat));
    for (int i = 0; i < 3; ++i) {
        io.pre_bias[i] = ms ? m->sub_bias[i] : 0.f;
        io.post_bias[i] = ms ? m->add_bias[i] : 0.f;
    }
    return forward(m, io, n, h, w, (hipStream_t)hip_stream);
}

extern "C" int srmi_model_forward_u8(srmi_model_t* m, const uint8_t* in, uint8_t* out, int n, int h, int w,
                                     int round_mode, void* hip_stream) {
    if (!m) SRMI_FAIL("srmi_model_forward_u8: null model");
    if (round_mode < 0 || round_mode > 2) SRMI_FAIL("srmi_model_forward_u8: bad round_mode %d", round_mode);
    IoSpec io;
    u8_io(m, in, out, round_mode, &io);
    return forward(m, io, n, h, w, (hipStream_t)hip_stream);
}

extern "C" int srmi_model_forward_u8_noalloc(srmi_model_t* m, const uint8_t* in, uint8_t* out, int n, int h, int w,
                                             int round_mode, void* hip_stream) {
    if (!m) SRMI_FAIL("srmi_model_forward_u8_noalloc: null model");
    if (round_mode < 0 || round_mode > 2) SRMI_FAIL("srmi_model_forward_u8_noalloc: bad round_mode %d", round_mode);
    IoSpec io;
    u8_io(m, in, out, round_mode, &io);
    return forward(m, io, n, h, w, (hipStream_t)hip_stream, /*static_ws=*/true);
}

extern "C" int srmi_model_forward_u8_host(srmi_model_t* m, const uint8_t* host_in, uint8_t* host_out, int n, int h, int w,
                                          int round_mode) {
    if (!m) SRMI_FAIL("srmi_model_forward_u8_host: null model");
    if (round_mode < 0 || round_mode > 2) SRMI_FAIL("srmi_model_forward_u8_host: bad round_mode %d", round_mode);
    return forward_u8_host(m, host_in, host_out, n, h, w, round_mode);
}

extern "C" int srmi_model_reserve_host(srmi_model_t* m, int n, int h, int w) {
    if (!m) SRMI_FAIL("srmi_model_reserve_host: null model");
    if (!m->finalized) SRMI_FAIL("srmi_model_reserve_host: srmi_model_finalize has not been called");
    if (n <= 0 || h <= 0 || w <= 0) SRMI_FAIL("srmi_model_reserve_host: bad shape n=%d h=%d w=%d", n, h, w);
    DeviceGuard guard;
    if (guard.enter(m->d.device)) return -1;
    std::lock_guard<std::mutex> lock(m->call_mutex);
    const int lanes = n < m->lanes ? n : m->lanes;
    Arena probe;
    probe.ns = 1;
    IoSpec io;
    memset(&io, 0, sizeof(io));
    if (run_plan(m, probe, nullptr, &io, h, w)) return -1;
    const size_t lane_bytes = (probe.off + 255) & ~(size_t)255;
    if (ensure_resources(m, lane_bytes * lanes, lanes, true)) return -1;
    const size_t in_bytes = (size_t)h * w * 3;
    return ensure_host_slots(m, lanes, in_bytes, in_bytes * m->d.scale * m->d.scale);
}

extern "C" size_t srmi_static_workspace_bytes(void) { return (size_t)SRMI_STATIC_WS_BYTES; }

extern "C" int srmi_model_reserve(srmi_model_t* m, int n, int h, int w) {
    if (!m) SRMI_FAIL("srmi_model_reserve: null model");
    if (!m->finalized) SRMI_FAIL("srmi_model_reserve: srmi_model_finalize has not been called");
    if (n <= 0 || h <= 0 || w <= 0) SRMI_FAIL("srmi_model_reserve: bad shape n=%d h=%d w=%d", n, h, w);
    DeviceGuard guard;
    if (guard.enter(m->d.device)) return -1;
    std::lock_guard<std::mutex> lock(m->call_mutex);
    int ns = 1, lanes = 1;
    size_t lane_bytes = 0;
    if (plan_bytes(m, n, h, w, m->lanes, &ns, &lanes, &lane_bytes)) return -1;
    return ensure_resources(m, lane_bytes * lanes, lanes);
}

// Core (no-MeanShift) models on the u8 runner path: p - shift before the first conv, + shift after
// the last (edsr.cpp:36-37,116); full models take the shift from their MeanShift tensors instead.
extern "C" int srmi_model_set_rgb_shift(srmi_model_t* m, const float* shift) {
    if (!m || !shift) SRMI_FAIL("srmi_model_set_rgb_shift: null argument");
    if (m->d.arch == SRMI_ARCH_RRDBNET || m->d.with_meanshift) return 0;
    for (int i = 0; i < 3; ++i) { m->core_shift[i] = shift[i]; }
    m->has_core_shift = true;
    return 0;
}

extern "C" size_t srmi_model_workspace_bytes(const srmi_model_t* m) { return m ? m->ws_bytes : 0; }

extern "C" int srmi_model_set_fused_attention(srmi_model_t* m, int on) {
    if (!m) SRMI_FAIL("srmi_model_set_fused_attention: null model");
    m->fuse_attention = on != 0;
    return 0;
}

extern "C" int srmi_model_set_fused_rcab(srmi_model_t* m, int on) {
    if (!m) SRMI_FAIL("srmi_model_set_fused_rcab: null model");
    std::lock_guard<std::mutex> lock(m->call_mutex);
    m->fuse_rcab = on != 0;
    return 0;
}

extern "C" int srmi_model_get_fused_rcab(const srmi_model_t* m) { return (m && m->fuse_rcab) ? 1 : 0; }

extern "C" int srmi_model_set_subpixel_up(srmi_model_t* m, int on) {
    if (!m) SRMI_FAIL("srmi_model_set_subpixel_up: null model");
    std::lock_guard<std::mutex> lock(m->call_mutex);
    m->subpixel_up = on != 0;
    return 0;
}

extern "C" int srmi_model_get_subpixel_up(const srmi_model_t* m) { return (m && m->subpixel_up) ? 1 : 0; }

extern "C" int srmi_model_set_sweep(srmi_model_t* m, int images_per_sweep) {
    if (!m || images_per_sweep < 1) SRMI_FAIL("srmi_model_set_sweep: bad argument");
    m->sweep = images_per_sweep;
    return 0;
}

extern "C" int srmi_model_set_growth_round8(srmi_model_t* m, int on) {
    if (!m) SRMI_FAIL("srmi_model_set_growth_round8: null model");
    if (m->finalized) SRMI_FAIL("srmi_model_set_growth_round8: call before srmi_model_finalize (the weights are packed there)");
    m->growth_round8 = on != 0;
    return 0;
}

extern "C" int srmi_model_set_winograd(srmi_model_t* m, int on) {
    if (!m) SRMI_FAIL("srmi_model_set_winograd: null model");
    std::lock_guard<std::mutex> lock(m->call_mutex);
    if (on != SRMI_ALGO_DIRECT && on != SRMI_ALGO_WINOGRAD && on != SRMI_ALGO_WINOGRAD_X)
        SRMI_FAIL("srmi_model_set_winograd: %d is none of SRMI_ALGO_DIRECT / _WINOGRAD / _WINOGRAD_X", on);
    m->winograd = on;
    return 0;
}

extern "C" int srmi_model_get_winograd(const srmi_model_t* m) { return m ? m->winograd : 0; }

extern "C" int srmi_model_calibrate_begin(srmi_model_t* m) {
    if (!m) SRMI_FAIL("srmi_model_calibrate_begin: null model");
    if (!m->finalized) SRMI_FAIL("srmi_model_calibrate_begin: srmi_model_finalize has not been called");
    if (m->d.arch != SRMI_ARCH_RRDBNET || m->d.dtype != SRMI_F16)
        SRMI_FAIL("srmi_model_calibrate_begin: the w8a8 mode exists for RRDBNet models with f16 compute");
    if (m->d.n_feats % 32 || m->d.grow_ch % 32) SRMI_FAIL("srmi_model_calibrate_begin: channel counts must be multiples of 32");
    DeviceGuard guard;
    if (guard.enter(m->d.device)) return -1;
    std::lock_guard<std::mutex> lock(m->call_mutex);
    const size_t bytes = (size_t)m->calib_slots() * sizeof(float);
    if (!m->calib_dev) SRMI_HIP(hipMalloc((void**)&m->calib_dev, bytes));
    SRMI_HIP(hipDeviceSynchronize());
    SRMI_HIP(hipMemset(m->calib_dev, 0, bytes));
    m->calibrating = true;
    return 0;
}

extern "C" int srmi_model_calibrate_end(srmi_model_t* m) {
    if (!m) SRMI_FAIL("srmi_model_calibrate_end: null model");
    if (!m->calibrating) SRMI_FAIL("srmi_model_calibrate_end: srmi_model_calibrate_begin has not been called");
    DeviceGuard guard;
    if (guard.enter(m->d.device)) return -1;
    std::lock_guard<std::mutex> lock(m->call_mutex);
    SRMI_HIP(hipDeviceSynchronize());
    m->absmax.assign(m->calib_slots(), 0.f);
    SRMI_HIP(hipMemcpy(m->absmax.data(), m->calib_dev, m->absmax.size() * sizeof(float), hipMemcpyDeviceToHost));
    m->calibrating = false;
    if (!(m->absmax[0] > 0.f)) SRMI_FAIL("srmi_model_calibrate_end: no forward pass ran between begin and end");
    if (prepare_w8a8(m)) return -1;
    m->calibrated = true;
    m->w8a8 = true;
    return 0;
}

extern "C" int srmi_model_set_calibration(srmi_model_t* m, const float* absmax, int n) {
    return srmi_model_set_calibration_ex(m, absmax, n, 1);
}

extern "C" int srmi_model_get_w8a8(const srmi_model_t* m) { return (m && m->w8a8) ? 1 : 0; }

extern "C" int srmi_model_set_calibration_ex(srmi_model_t* m, const float* absmax, int n, int enable_w8a8) {
    if (!m || !absmax) SRMI_FAIL("srmi_model_set_calibration: null argument");
    if (!m->finalized) SRMI_FAIL("srmi_model_set_calibration: srmi_model_finalize has not been called");
    if (m->d.arch != SRMI_ARCH_RRDBNET || m->d.dtype != SRMI_F16)
        SRMI_FAIL("srmi_model_set_calibration: the w8a8 mode exists for RRDBNet models with f16 compute");
    if (m->d.n_feats % 32 || m->d.grow_ch % 32) SRMI_FAIL("srmi_model_set_calibration: channel counts must be multiples of 32");
    if (n != m->calib_slots()) SRMI_FAIL("srmi_model_set_calibration: %d values given, this network has %d calibrated tensors", n, m->calib_slots());
    for (int i = 0; i < n; ++i)
        if (!(absmax[i] >= 0.f) || absmax[i] > 3.0e38f) SRMI_FAIL("srmi_model_set_calibration: value %d is not a finite non-negative number", i);
    DeviceGuard guard;
    if (guard.enter(m->d.device)) return -1;
    std::lock_guard<std::mutex> lock(m->call_mutex);
    SRMI_HIP(hipDeviceSynchronize());
    m->absmax.assign(absmax, absmax + n);
    m->calibrating = false;
    if (prepare_w8a8(m)) return -1;
    m->calibrated = true;
    m->w8a8 = enable_w8a8 != 0;
    return 0;
}

extern "C" int srmi_model_set_w8a8(srmi_model_t* m, int on) {
    if (!m) SRMI_FAIL("srmi_model_set_w8a8: null model");
    if (on && !m->calibrated) SRMI_FAIL("srmi_model_set_w8a8: the model has not been calibrated (srmi_model_calibrate_begin / _end)");
    std::lock_guard<std::mutex> lock(m->call_mutex);
    m->w8a8 = on != 0;
    return 0;
}

extern "C" int srmi_model_calibration(const srmi_model_t* m, float* absmax, int capacity) {
    if (!m || !m->calibrated) return 0;
    const int n = (int)m->absmax.size();
    if (absmax)
        for (int i = 0; i < n && i < capacity; ++i) absmax[i] = m->absmax[i];
    return n;
}

extern "C" int srmi_model_set_lanes(srmi_model_t* m, int lanes) {
    if (!m || lanes < 1 || lanes > srmi_model::kMaxLanes)
        SRMI_FAIL("srmi_model_set_lanes: lanes must be 1..%d", srmi_model::kMaxLanes);
    m->lanes = lanes;
    return 0;
}

extern "C" double srmi_model_flops(const srmi_model_t* m, int h, int w) {
    if (!m) return 0.0;
    const auto& d = m->d;
    const double px = (double)h * w, nf = d.n_feats;
    auto cv = [](double cin, double cout, double pixels) { return 2.0 * cin * cout * 9.0 * pixels; };
    double f = cv(3, nf, px);
    if (d.arch == SRMI_ARCH_RRDBNET) {
        const double gc = d.grow_ch;
        double rdb = 0;
        for (int j = 0; j < 4; ++j) rdb += cv(nf + j * gc, gc, px);
        rdb += cv(nf + 4 * gc, nf, px);
        f += 3.0 * d.n_blocks * rdb + cv(nf, nf, px);
        f += cv(nf, nf, 4 * px);
        double hr = 4 * px;
        if (d.scale == 4) { f += cv(nf, nf, 16 * px); hr = 16 * px; }
        f += cv(nf, nf, hr) + cv(nf, 3, hr);
        return f;
    }
    if (d.arch == SRMI_ARCH_EDSR) f += (2.0 * d.n_blocks + 1) * cv(nf, nf, px);
    else f += (d.n_groups * (2.0 * d.n_blocks + 1) + 1) * cv(nf, nf, px);
    int r;
    const int stages = upsampler_stages(d.scale, &r);
    double p = px;
    for (int s = 0; s < stages; ++s) { f += cv(nf, r * r * nf, p); p *= r * r; }
    f += cv(nf, 3, p);
    return f;
}
