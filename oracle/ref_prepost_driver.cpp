// Test infrastructure only (never linked into or loaded by the product): exposes the reference runners'
// file-local pre-processing, post-processing and bilinear resize functions to the tests.  The reference
// source file named by REF_RUNNER_SRC is compiled from where it lies (this file includes it, so its
// `static` functions are visible here) together with the reference's own neuron_executor.cpp; nothing of
// the vendor runtime is needed because no executor is ever initialised.  Built by oracle/Makefile into
// oracle/_ref/ when /root/reference is present.
//
//   REF_RUNNER_SRC  absolute path of edsr.cpp / rcan.cpp / realesrgan.cpp
//   REF_MEANSHIFT   1 for EDSR / RCAN (mean and range arguments), 0 for Real-ESRGAN
#include REF_RUNNER_SRC

#include <cstdlib>
#include <cstring>

extern "C" {

// u8 HWC -> f32 NCHW (edsr.cpp:21-41, rcan.cpp:21-41, realesrgan.cpp:28-48)
void ref_preprocess(const unsigned char* in, float* out, int h, int w, const float* mean, float range) {
#if REF_MEANSHIFT
    preprocess_image(in, out, h, w, mean, range);
#else
    (void)mean; (void)range;
    preprocess_image(in, out, h, w);
#endif
}

// f32 NCHW -> u8 HWC (edsr.cpp:101-122, rcan.cpp:101-122, realesrgan.cpp:110-130)
void ref_postprocess(const float* in, unsigned char* out, int h, int w, const float* mean, float range) {
#if REF_MEANSHIFT
    postprocess_image(in, out, h, w, mean, range);
#else
    (void)mean; (void)range;
    postprocess_image(in, out, h, w);
#endif
}

// bilinear u8 resize (edsr.cpp:46-94, rcan.cpp:46-94, realesrgan.cpp:53-101); returns 0 on success
int ref_resize(const unsigned char* in, int in_w, int in_h, unsigned char* out, int out_w, int out_h) {
    unsigned char* r = resize_image(in, in_w, in_h, out_w, out_h);
    if (!r) return -1;
    memcpy(out, r, (size_t)out_w * out_h * 3);
    free(r);
    return 0;
}

}  // extern "C"
