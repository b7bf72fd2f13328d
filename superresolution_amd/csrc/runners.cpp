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
#include "../../include/sr.h"
#include "common.h"


namespace srmi {

struct Runner {
    srmi_model_t* model = nullptr;
    hipStream_t stream = nullptr;
    uint8_t *d_in = nullptr, *d_resized = nullptr, *d_out = nullptr;
    size_t in_cap = 0;
    int in_w = 0, in_h = 0, scale = 0, device = 0;
};

// Input of another size: upload, bilinear resize on the device (edsr.cpp:46-94), network, download.
static int runner_run_resized(Runner* r, const unsigned char* rgb, int w, int h, unsigned char* host, size_t out_bytes) {
    int prev = 0;
    SRMI_HIP(hipGetDevice(&prev));
    SRMI_HIP(hipSetDevice(r->device));
    int rc = -1;
    do {
        const size_t in_bytes = (size_t)w * h * 3;
        if (in_bytes > r->in_cap) {
            if (r->d_in && hipFree(r->d_in) != hipSuccess) break;
            r->d_in = nullptr; r->in_cap = 0;
            if (hipMalloc((void**)&r->d_in, in_bytes) != hipSuccess) { set_error("runner: cannot allocate the input image on the device"); break; }
            r->in_cap = in_bytes;
        }
        if (hipMemcpyAsync(r->d_in, rgb, in_bytes, hipMemcpyHostToDevice, r->stream) != hipSuccess) { set_error("runner: host -> device copy failed"); break; }
        if (srmi_resize_bilinear_u8(r->d_in, w, h, r->d_resized, r->in_w, r->in_h, r->stream)) break;
        if (srmi_model_forward_u8(r->model, r->d_resized, r->d_out, 1, r->in_h, r->in_w, SRMI_ROUND_HALF_UP, r->stream)) break;
        if (hipMemcpyAsync(host, r->d_out, out_bytes, hipMemcpyDeviceToHost, r->stream) != hipSuccess ||
            hipStreamSynchronize(r->stream) != hipSuccess) {
            set_error("runner: device -> host copy failed: %s", hipGetErrorString(hipGetLastError()));
            break;
        }
        rc = 0;
    } while (0);
    (void)hipSetDevice(prev);
    return rc;
}

static int env_dtype() {
    const char* e = getenv("SRMI_DTYPE");
    if (!e) return SRMI_F16;
    if (!strcmp(e, "f32") || !strcmp(e, "fp32")) return SRMI_F32;
    if (!strcmp(e, "bf16")) return SRMI_BF16;
    return SRMI_F16;
}

static bool read_exact(FILE* f, void* dst, size_t n) { return fread(dst, 1, n, f) == n; }

static int load_model_body(FILE* f, long file_size, const char* path, int want_arch, int device, srmi_model_t** out,
                           srmi_model_desc_t* desc_out) {
    char magic[8];
    int32_t hdr[8];
    float res_scale;
    int32_t nt;
    if (!read_exact(f, magic, 8) || memcmp(magic, "SRMIW001", 8) || !read_exact(f, hdr, sizeof(hdr)) ||
        !read_exact(f, &res_scale, 4) || !read_exact(f, &nt, 4))
        SRMI_FAIL("%s is not an SRMIW001 weight file", path);
    srmi_model_desc_t d;
    memset(&d, 0, sizeof(d));
    d.arch = hdr[0]; d.scale = hdr[1]; d.n_feats = hdr[2]; d.n_blocks = hdr[3];
    d.n_groups = hdr[4]; d.grow_ch = hdr[5]; d.reduction = hdr[6]; d.with_meanshift = hdr[7];
    d.res_scale = res_scale;
    d.dtype = env_dtype();
    // SRMI_TRUNK_F32=0: the residual trunk in the compute type instead of an f32 copy beside it (faster)
    const char* tf = getenv("SRMI_TRUNK_F32");
    d.trunk_f32 = !(tf && !strcmp(tf, "0"));
    d.device = device;
    if (want_arch && d.arch != want_arch) SRMI_FAIL("%s holds architecture %d, expected %d", path, d.arch, want_arch);
    srmi_model_t* m = nullptr;
    if (srmi_model_create(&d, &m)) return -1;
    // SRMI_GROWTH_ROUND8=0: plain f16 in RRDBNet's dense blocks (srmi_model_set_growth_round8)
    const char* g = getenv("SRMI_GROWTH_ROUND8");
    if (g && !strcmp(g, "0") && srmi_model_set_growth_round8(m, 0)) { srmi_model_destroy(m); return -1; }
    std::vector<float> data;
    bool ok = nt >= 0;
    for (int i = 0; i < nt && ok; ++i) {
        int32_t nl = 0, nd = 0;
        ok = read_exact(f, &nl, 4) && nl > 0 && nl <= 4096;
        if (!ok) break;
        std::string name(nl, '\0');
        ok = read_exact(f, &name[0], nl) && read_exact(f, &nd, 4) && nd >= 0 && nd <= 8;
        if (!ok) break;
        int64_t dims[8];
        ok = !nd || read_exact(f, dims, sizeof(int64_t) * nd);
        if (!ok) break;
        // a tensor cannot hold more floats than the file has bytes left for
        const long left = file_size - ftell(f);
        int64_t n = 1;
        for (int k = 0; k < nd && ok; ++k) {
            ok = dims[k] >= 0 && dims[k] <= (int64_t)left;
            if (ok) { n *= dims[k]; ok = n <= (int64_t)left / 4; }
        }
        if (!ok) break;
        data.resize((size_t)n);
        ok = read_exact(f, data.data(), sizeof(float) * (size_t)n);
        if (!ok) break;
        if (srmi_model_set_tensor(m, name.c_str(), data.data(), dims, nd)) { srmi_model_destroy(m); return -1; }
    }
    if (!ok) {
        srmi_model_destroy(m);
        SRMI_FAIL("%s is truncated or corrupt", path);
    }
    if (srmi_model_finalize(m)) { srmi_model_destroy(m); return -1; }
    // optional trailer: activation ranges of a model calibrated for the int8 mode; SRMI_W8A8=0 ignores them
    char tag[8];
    int32_t nc = 0;
    if (read_exact(f, tag, 8) && !memcmp(tag, "SRMIQ001", 8)) {
        const char* q = getenv("SRMI_W8A8");
        if (!read_exact(f, &nc, 4) || nc <= 0 || (long)nc * 4 > file_size - ftell(f)) {
            srmi_model_destroy(m);
            SRMI_FAIL("%s: corrupt calibration trailer", path);
        }
        std::vector<float> cal((size_t)nc);
        if (!read_exact(f, cal.data(), sizeof(float) * (size_t)nc)) {
            srmi_model_destroy(m);
            SRMI_FAIL("%s: truncated calibration trailer", path);
        }
        if (!(q && !strcmp(q, "0")) && d.dtype == SRMI_F16 && srmi_model_set_calibration(m, cal.data(), nc)) {
            srmi_model_destroy(m);
            return -1;
        }
    }
    // which arithmetic this context runs in (a calibrated weight file switches the dense blocks to int8)
    {
        const char* g8 = getenv("SRMI_GROWTH_ROUND8");
        const bool round8 = d.arch == SRMI_ARCH_RRDBNET && d.dtype == SRMI_F16 && !(g8 && !strcmp(g8, "0"));
        fprintf(stderr, "[INFO] %s: precision %s%s%s, %s residual trunk, device %d\n", path,
                d.dtype == SRMI_F32 ? "f32" : d.dtype == SRMI_BF16 ? "bf16" : "f16",
                srmi_model_get_w8a8(m) ? " + w8a8 dense blocks (int8, calibrated ranges from the weight file)" : "",
                (round8 && !srmi_model_get_w8a8(m)) ? " (dense-block grown channels at 8 significant bits)" : "",
                (d.trunk_f32 || d.dtype == SRMI_F32) ? "f32" : d.dtype == SRMI_BF16 ? "bf16" : "f16", device);
    }
    *out = m;
    *desc_out = d;
    return 0;
}

}  // namespace srmi

// Device a new runner context or executor binds to: $SRMI_DEVICE when set, else the calling thread's
// current HIP device (so hipSetDevice(k) before <m>_init puts the context on GPU k).
int srmi_pick_device(int* device) {
    const char* e = getenv("SRMI_DEVICE");
    if (e && *e) {
        *device = atoi(e);
        return 0;
    }
    SRMI_HIP(hipGetDevice(device));
    return 0;
}

// Loads a weight file into a new model on `device`; `want_arch` (0 = any) guards against feeding e.g. an
// EDSR file to rcan_init.  No exception leaves this function.
int srmi_load_weight_file(const char* path, int want_arch, int device, srmi_model_t** out, srmi_model_desc_t* desc_out) {
    FILE* f = fopen(path, "rb");
    if (!f) SRMI_FAIL("cannot open weight file %s", path);
    long size = 0;
    if (fseek(f, 0, SEEK_END) == 0) size = ftell(f);
    rewind(f);
    int rc = -1;
    try {
        rc = srmi::load_model_body(f, size, path, want_arch, device, out, desc_out);
    } catch (const std::exception& e) {
        srmi::set_error("%s: %s", path, e.what());
    } catch (...) {
        srmi::set_error("%s: unknown failure while loading", path);
    }
    fclose(f);
    return rc;
}

namespace srmi {

static void runner_destroy(Runner* r) {
    if (!r) return;
    if (r->stream) (void)hipStreamSynchronize(r->stream);
    if (r->d_in) (void)hipFree(r->d_in);
    if (r->d_resized) (void)hipFree(r->d_resized);
    if (r->d_out) (void)hipFree(r->d_out);
    if (r->model) srmi_model_destroy(r->model);
    if (r->stream) (void)hipStreamDestroy(r->stream);
    delete r;
}

static int runner_create(const char* path, int arch, int in_w, int in_h, Runner** out, int* scale) {
    int device = 0;
    if (srmi_pick_device(&device)) return -1;
    srmi_model_t* m = nullptr;
    srmi_model_desc_t d;
    if (srmi_load_weight_file(path, arch, device, &m, &d)) return -1;
    Runner* r = new Runner();
    r->model = m;
    r->device = device;
    r->in_w = in_w; r->in_h = in_h; r->scale = d.scale;
    int prev = 0;
    (void)hipGetDevice(&prev);
    const size_t in_bytes = (size_t)in_w * in_h * 3, out_bytes = in_bytes * d.scale * d.scale;
    // everything an inference needs is set up here, as the reference builds its executor and buffers in
    // <m>_init (realesrgan.cpp:137-190): the first inference pays for no allocation
    const bool ok = hipSetDevice(device) == hipSuccess && hipStreamCreate(&r->stream) == hipSuccess &&
                    hipMalloc((void**)&r->d_resized, in_bytes) == hipSuccess &&
                    hipMalloc((void**)&r->d_out, out_bytes) == hipSuccess;
    if (!ok) set_error("runner: device allocation failed: %s", hipGetErrorString(hipGetLastError()));
    const int rc = ok ? srmi_model_reserve_host(m, 1, in_h, in_w) : -1;
    (void)hipSetDevice(prev);
    if (rc) {
        runner_destroy(r);
        return -1;
    }
    *out = r;
    *scale = d.scale;
    return 0;
}

// u8 HWC host image (w x h) -> malloc()ed u8 HWC host image of the model's output size.
static int runner_run(Runner* r, const unsigned char* rgb, int w, int h, unsigned char** out) {
    const size_t out_bytes = (size_t)r->in_w * r->in_h * 3 * r->scale * r->scale;
    unsigned char* host = (unsigned char*)malloc(out_bytes);
    if (!host) SRMI_FAIL("runner: cannot allocate %zu bytes for the output image", out_bytes);
    int rc;
    if (w == r->in_w && h == r->in_h) {
        // pinned staging, copies on the lane's stream (srmi_model_forward_u8_host)
        rc = srmi_model_forward_u8_host(r->model, rgb, host, 1, h, w, SRMI_ROUND_HALF_UP);
    } else {
        rc = runner_run_resized(r, rgb, w, h, host, out_bytes);
    }
    if (rc) {
        free(host);
        return -1;
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

// ---- generic runner (include/sr.h): any of the three networks, native input size ---------------------------
extern "C" int sr_init(const char* model_path, sr_context_t* ctx) {
    if (!model_path || !ctx) { fprintf(stderr, "[ERROR] Invalid parameters\n"); return -1; }
    memset(ctx, 0, sizeof(*ctx));
    int device = 0;
    srmi_model_t* m = nullptr;
    srmi_model_desc_t d;
    if (srmi_pick_device(&device) || srmi_load_weight_file(model_path, 0, device, &m, &d)) {
        log_error("sr_init");
        return -1;
    }
    if (d.arch != SRMI_ARCH_RRDBNET && !d.with_meanshift) {
        // a core EDSR / RCAN file: the runner subtracts and adds the dataset mean itself (edsr.cpp:36-37,116)
        const float shift[3] = {0.4488f * 255.0f, 0.4371f * 255.0f, 0.4040f * 255.0f};
        if (srmi_model_set_rgb_shift(m, shift)) {
            srmi_model_destroy(m);
            log_error("sr_init");
            return -1;
        }
    }
    ctx->executor = m;
    ctx->arch = d.arch;
    ctx->scale_factor = d.scale;
    ctx->channels = 3;
    ctx->initialized = 1;
    return 0;
}

extern "C" int sr_release(sr_context_t* ctx) {
    if (!ctx || !ctx->initialized) return 0;
    srmi_model_destroy((srmi_model_t*)ctx->executor);
    ctx->executor = nullptr;
    ctx->initialized = 0;
    return 0;
}

extern "C" int sr_reserve(sr_context_t* ctx, int input_width, int input_height) {
    if (!ctx || !ctx->initialized) { fprintf(stderr, "[ERROR] Invalid parameters\n"); return -1; }
    if (input_width <= 0 || input_height <= 0) { fprintf(stderr, "[ERROR] Invalid input size\n"); return -1; }
    if (srmi_model_reserve_host((srmi_model_t*)ctx->executor, 1, input_height, input_width)) {
        log_error("sr_reserve");
        return -1;
    }
    return 0;
}

extern "C" int sr_inference(sr_context_t* ctx, const unsigned char* input_rgb, int input_width, int input_height,
                            unsigned char** output_rgb, int* output_width, int* output_height) {
    if (!ctx || !ctx->initialized || !input_rgb || !output_rgb) { fprintf(stderr, "[ERROR] Invalid parameters\n"); return -1; }
    if (input_width <= 0 || input_height <= 0) { fprintf(stderr, "[ERROR] Invalid input size\n"); return -1; }
    const size_t s = (size_t)ctx->scale_factor;
    const size_t out_bytes = (size_t)input_width * s * (size_t)input_height * s * 3;
    unsigned char* host = (unsigned char*)malloc(out_bytes);
    if (!host) {
        set_error("sr_inference: cannot allocate %zu bytes for the output image", out_bytes);
        log_error("sr_inference");
        return -1;
    }
    if (srmi_model_forward_u8_host((srmi_model_t*)ctx->executor, input_rgb, host, 1, input_height, input_width,
                                   SRMI_ROUND_HALF_UP)) {
        free(host);
        log_error("sr_inference");
        return -1;
    }
    *output_rgb = host;
    if (output_width) *output_width = input_width * ctx->scale_factor;
    if (output_height) *output_height = input_height * ctx->scale_factor;
    return 0;
}

extern "C" const char* sr_last_error(void) { return srmi_last_error(); }
