// The reference runners' extern "C" API (<model>_init / _inference / _inference_ex / _release,
// include/edsr.h, rcan.h, realesrgan.h) on top of the model executor: host u8 buffers in and out,
// fixed model input size, bilinear resize of other sizes, malloc()ed result.
//
// Weight file ("SRMIW001"), little endian, written by superresolution_amd/export.py:
//   char magic[8]; int32 arch, scale, n_feats, n_blocks, n_groups, grow_ch, reduction,
//   with_meanshift; float res_scale; int32 n_tensors;
//   per tensor: int32 name_len; char name[name_len]; int32 ndim; int64 dims[ndim]; float data[].
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "../../include/edsr.h"
#include "../../include/rcan.h"
#include "../../include/realesrgan.h"
#include "common.h"

extern "C" int srmi_model_set_rgb_shift(srmi_model_t* m, const float* mean_times_range);

namespace srmi {

struct Runner {
    srmi_model_t* model = nullptr;
    hipStream_t stream = nullptr;
    uint8_t *d_in = nullptr, *d_resized = nullptr, *d_out = nullptr;
    size_t in_cap = 0;
    int in_w = 0, in_h = 0, scale = 0;
};

static int env_dtype() {
    const char* e = getenv("SRMI_DTYPE");
    if (!e) return SRMI_F16;
    if (!strcmp(e, "f32") || !strcmp(e, "fp32")) return SRMI_F32;
    if (!strcmp(e, "bf16")) return SRMI_BF16;
    return SRMI_F16;
}

static bool read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

// Loads a weight file into a new model; `want_arch` guards against feeding e.g. an EDSR file to
// rcan_init.
static int load_model(const char* path, int want_arch, srmi_model_t** out, srmi_model_desc_t* desc_out) {
    FILE* f = fopen(path, "rb");
    if (!f) SRMI_FAIL("cannot open weight file %s", path);
    char magic[8];
    int32_t hdr[8];
    float res_scale;
    int32_t nt;
    if (!read_exact(f, magic, 8) || memcmp(magic, "SRMIW001", 8) || !read_exact(f, hdr, sizeof(hdr)) ||
        !read_exact(f, &res_scale, 4) || !read_exact(f, &nt, 4)) {
        fclose(f);
        SRMI_FAIL("%s is not an SRMIW001 weight file", path);
    }
    srmi_model_desc_t d;
    memset(&d, 0, sizeof(d));
    d.arch = hdr[0]; d.scale = hdr[1]; d.n_feats = hdr[2]; d.n_blocks = hdr[3];
    d.n_groups = hdr[4]; d.grow_ch = hdr[5]; d.reduction = hdr[6]; d.with_meanshift = hdr[7];
    d.res_scale = res_scale;
    d.dtype = env_dtype();
    d.trunk_f32 = 1;
    d.device = 0;
    if (d.arch != want_arch) {
        fclose(f);
        SRMI_FAIL("%s holds architecture %d, expected %d", path, d.arch, want_arch);
    }
    srmi_model_t* m = nullptr;
    if (srmi_model_create(&d, &m)) { fclose(f); return -1; }
    std::vector<float> data;
    for (int i = 0; i < nt; ++i) {
        int32_t nl, nd;
        if (!read_exact(f, &nl, 4) || nl <= 0 || nl > 4096) goto bad;
        {
            std::string name(nl, '\0');
            if (!read_exact(f, &name[0], nl) || !read_exact(f, &nd, 4) || nd < 0 || nd > 8) goto bad;
            int64_t dims[8];
            if (nd && !read_exact(f, dims, sizeof(int64_t) * nd)) goto bad;
            int64_t n = 1;
            for (int k = 0; k < nd; ++k) {
                if (dims[k] < 0 || dims[k] > (1ll << 31)) goto bad;
                n *= dims[k];
                if (n > (1ll << 31)) goto bad;
            }
            data.resize((size_t)n);
            if (!read_exact(f, data.data(), sizeof(float) * (size_t)n)) goto bad;
            if (srmi_model_set_tensor(m, name.c_str(), data.data(), dims, nd)) { fclose(f); srmi_model_destroy(m); return -1; }
        }
    }
    fclose(f);
    if (srmi_model_finalize(m)) { srmi_model_destroy(m); return -1; }
    *out = m;
    *desc_out = d;
    return 0;
bad:
    fclose(f);
    srmi_model_destroy(m);
    SRMI_FAIL("%s is truncated or corrupt", path);
}

static int runner_create(const char* path, int arch, int in_w, int in_h, Runner** out, int* scale) {
    srmi_model_t* m = nullptr;
    srmi_model_desc_t d;
    if (load_model(path, arch, &m, &d)) return -1;
    Runner* r = new Runner();
    r->model = m;
    r->in_w = in_w; r->in_h = in_h; r->scale = d.scale;
    const size_t in_bytes = (size_t)in_w * in_h * 3, out_bytes = in_bytes * d.scale * d.scale;
    if (hipStreamCreate(&r->stream) != hipSuccess || hipMalloc((void**)&r->d_resized, in_bytes) != hipSuccess ||
        hipMalloc((void**)&r->d_out, out_bytes) != hipSuccess) {
        set_error("runner: device allocation failed");
        srmi_model_destroy(m);
        delete r;
        return -1;
    }
    *out = r;
    *scale = d.scale;
    return 0;
}

static void runner_destroy(Runner* r) {
    if (!r) return;
    (void)hipStreamSynchronize(r->stream);
    if (r->d_in) (void)hipFree(r->d_in);
    (void)hipFree(r->d_resized);
    (void)hipFree(r->d_out);
    srmi_model_destroy(r->model);
    (void)hipStreamDestroy(r->stream);
    delete r;
}

// u8 HWC host image (w x h) -> malloc()ed u8 HWC host image of the model's output size.
static int runner_run(Runner* r, const unsigned char* rgb, int w, int h, unsigned char** out) {
    const size_t in_bytes = (size_t)w * h * 3;
    const uint8_t* d_src;
    if (w == r->in_w && h == r->in_h) {
        SRMI_HIP(hipMemcpyAsync(r->d_resized, rgb, in_bytes, hipMemcpyHostToDevice, r->stream));
        d_src = r->d_resized;
    } else {
        if (in_bytes > r->in_cap) {
            if (r->d_in) SRMI_HIP(hipFree(r->d_in));
            r->d_in = nullptr; r->in_cap = 0;
            SRMI_HIP(hipMalloc((void**)&r->d_in, in_bytes));
            r->in_cap = in_bytes;
        }
        SRMI_HIP(hipMemcpyAsync(r->d_in, rgb, in_bytes, hipMemcpyHostToDevice, r->stream));
        if (srmi_resize_bilinear_u8(r->d_in, w, h, r->d_resized, r->in_w, r->in_h, r->stream)) return -1;
        d_src = r->d_resized;
    }
    if (srmi_model_forward_u8(r->model, d_src, r->d_out, 1, r->in_h, r->in_w, SRMI_ROUND_HALF_UP, r->stream)) return -1;
    const size_t out_bytes = (size_t)r->in_w * r->in_h * 3 * r->scale * r->scale;
    unsigned char* host = (unsigned char*)malloc(out_bytes);
    if (!host) SRMI_FAIL("runner: cannot allocate %zu bytes for the output image", out_bytes);
    if (hipMemcpyAsync(host, r->d_out, out_bytes, hipMemcpyDeviceToHost, r->stream) != hipSuccess ||
        hipStreamSynchronize(r->stream) != hipSuccess) {
        free(host);
        SRMI_FAIL("runner: device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));
    }
    *out = host;
    return 0;
}

static void log_error(const char* what) { fprintf(stderr, "[ERROR] %s: %s\n", what, srmi_last_error()); }

}  // namespace srmi

using namespace srmi;

// ---- Real-ESRGAN ------------------------------------------------------------------------------------
extern "C" int realesrgan_init(const char* model_path, realesrgan_context_t* ctx) {
    if (!model_path || !ctx) { fprintf(stderr, "[ERROR] Invalid parameters\n"); return -1; }
    memset(ctx, 0, sizeof(*ctx));
    ctx->input_height = 339; ctx->input_width = 510; ctx->input_channels = 3;
    Runner* r = nullptr;
    int scale = 0;
    if (runner_create(model_path, SRMI_ARCH_RRDBNET, ctx->input_width, ctx->input_height, &r, &scale)) {
        log_error("realesrgan_init");
        return -1;
    }
    ctx->scale_factor = scale;
    ctx->output_height = ctx->input_height * scale; ctx->output_width = ctx->input_width * scale;
    ctx->output_channels = 3;
    ctx->input_size = (size_t)ctx->input_height * ctx->input_width * 3 * sizeof(float);
    ctx->output_size = (size_t)ctx->output_height * ctx->output_width * 3 * sizeof(float);
    ctx->neuron_executor = r;
    ctx->initialized = 1;
    return 0;
}

extern "C" int realesrgan_release(realesrgan_context_t* ctx) {
    if (!ctx || !ctx->initialized) return -1;
    runner_destroy((Runner*)ctx->neuron_executor);
    ctx->neuron_executor = nullptr;
    ctx->initialized = 0;
    return 0;
}

extern "C" int realesrgan_inference(realesrgan_context_t* ctx, const unsigned char* input_rgb, unsigned char** output_rgb) {
    if (!ctx || !ctx->initialized || !input_rgb || !output_rgb) { fprintf(stderr, "[ERROR] Invalid parameters\n"); return -1; }
    if (runner_run((Runner*)ctx->neuron_executor, input_rgb, ctx->input_width, ctx->input_height, output_rgb)) {
        log_error("realesrgan_inference");
        return -1;
    }
    return 0;
}

extern "C" int realesrgan_inference_ex(realesrgan_context_t* ctx, const unsigned char* input_rgb, int input_width,
                                       int input_height, unsigned char** output_rgb, int* output_width, int* output_height) {
    if (!ctx || !ctx->initialized || !input_rgb || !output_rgb || !output_width || !output_height) {
        fprintf(stderr, "[ERROR] Invalid parameters\n");
        return -1;
    }
    if (input_width <= 0 || input_height <= 0) { fprintf(stderr, "[ERROR] Invalid input size\n"); return -1; }
    if (runner_run((Runner*)ctx->neuron_executor, input_rgb, input_width, input_height, output_rgb)) {
        log_error("realesrgan_inference_ex");
        return -1;
    }
    *output_width = ctx->output_width;
    *output_height = ctx->output_height;
    return 0;
}

// ---- EDSR / RCAN (identical apart from the fixed size) -------------------------------------------------
#define SRMI_MEANSHIFT_RUNNER(NAME, ARCH, IN_W, IN_H)                                                          \
    extern "C" int NAME##_init(const char* model_path, NAME##_context_t* ctx) {                                \
        if (!model_path || !ctx) { fprintf(stderr, "[ERROR] Invalid parameters\n"); return -1; }              \
        memset(ctx, 0, sizeof(*ctx));                                                                          \
        Runner* r = nullptr;                                                                                   \
        int scale = 0;                                                                                         \
        if (runner_create(model_path, ARCH, IN_W, IN_H, &r, &scale)) { log_error(#NAME "_init"); return -1; } \
        ctx->neuron_executor = r;                                                                              \
        ctx->input_width = IN_W; ctx->input_height = IN_H; ctx->input_channels = 3;                            \
        ctx->output_width = IN_W * scale; ctx->output_height = IN_H * scale; ctx->output_channels = 3;         \
        ctx->scale_factor = scale;                                                                             \
        ctx->rgb_mean[0] = 0.4488f; ctx->rgb_mean[1] = 0.4371f; ctx->rgb_mean[2] = 0.4040f;                    \
        ctx->rgb_range = 255.0f;                                                                               \
        const float shift[3] = {ctx->rgb_mean[0] * ctx->rgb_range, ctx->rgb_mean[1] * ctx->rgb_range,          \
                                ctx->rgb_mean[2] * ctx->rgb_range};                                            \
        srmi_model_set_rgb_shift(r->model, shift);                                                             \
        ctx->initialized = 1;                                                                                  \
        return 0;                                                                                              \
    }                                                                                                          \
    extern "C" int NAME##_release(NAME##_context_t* ctx) {                                                     \
        if (!ctx || !ctx->initialized) return 0;                                                               \
        runner_destroy((Runner*)ctx->neuron_executor);                                                         \
        ctx->neuron_executor = nullptr;                                                                        \
        ctx->initialized = 0;                                                                                  \
        return 0;                                                                                              \
    }                                                                                                          \
    extern "C" int NAME##_inference_ex(NAME##_context_t* ctx, const unsigned char* input_rgb, int input_width, \
                                       int input_height, unsigned char** output_rgb, int* output_width,       \
                                       int* output_height) {                                                   \
        if (!ctx || !ctx->initialized || !input_rgb || !output_rgb) {                                          \
            fprintf(stderr, "[ERROR] Invalid parameters\n");                                                  \
            return -1;                                                                                         \
        }                                                                                                      \
        if (input_width <= 0 || input_height <= 0) { fprintf(stderr, "[ERROR] Invalid input size\n"); return -1; } \
        if (runner_run((Runner*)ctx->neuron_executor, input_rgb, input_width, input_height, output_rgb)) {     \
            log_error(#NAME "_inference_ex");                                                                  \
            return -1;                                                                                         \
        }                                                                                                      \
        if (output_width) *output_width = ctx->output_width;                                                   \
        if (output_height) *output_height = ctx->output_height;                                                \
        return 0;                                                                                              \
    }                                                                                                          \
    extern "C" int NAME##_inference(NAME##_context_t* ctx, const unsigned char* input_rgb, unsigned char** output_rgb) { \
        if (!ctx) { fprintf(stderr, "[ERROR] Invalid parameters\n"); return -1; }                             \
        return NAME##_inference_ex(ctx, input_rgb, ctx->input_width, ctx->input_height, output_rgb, nullptr, nullptr); \
    }

SRMI_MEANSHIFT_RUNNER(edsr, SRMI_ARCH_EDSR, 256, 256)
SRMI_MEANSHIFT_RUNNER(rcan, SRMI_ARCH_RCAN, 510, 339)
