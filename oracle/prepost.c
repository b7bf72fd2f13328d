/*
 * CPU oracle for the byte-level pre/post-processing of the reference's MTK runners, restated in
 * plain C.  TEST INFRASTRUCTURE ONLY: tests/ and bench.py's cpu_baseline leg may load this; the
 * product library never links it.  Build: make -C oracle  (gcc -O2 -ffp-contract=off, so every
 * float operation is rounded individually, as in the numpy restatement of the same functions).
 *
 *   or_preprocess_unit      realesrgan/mtk/cpp/jni/src/realesrgan.cpp:28-48
 *   or_preprocess_meanshift edsr/mtk/cpp/jni/src/edsr.cpp:21-41
 *   or_postprocess_unit     realesrgan.cpp:110-130
 *   or_postprocess_range    edsr.cpp:101-122
 *   or_resize_bilinear      edsr.cpp:46-94 (identical in rcan.cpp and realesrgan.cpp:53-101)
 *
 * Pinning: tests/golden/prepost_cpp_ref.npz holds outputs of the reference runners' own C++ functions
 * (compiled from the reference's sources by `make -C oracle ref`, run by tools/make_golden_prepost_cpp.py);
 * tests/test_oracle.py checks these restatements against them bit for bit.
 */
#include <stdint.h>

void or_preprocess_unit(const uint8_t* rgb, float* out, int h, int w) {
    const float scale = 1.0f / 255.0f;
    for (int c = 0; c < 3; c++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) out[(c * h + y) * w + x] = (float)rgb[(y * w + x) * 3 + c] * scale;
}

void or_preprocess_meanshift(const uint8_t* rgb, float* out, int h, int w, const float* mean, float range) {
    for (int c = 0; c < 3; c++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) out[(c * h + y) * w + x] = (float)rgb[(y * w + x) * 3 + c] - mean[c] * range;
}

void or_postprocess_unit(const float* in, uint8_t* rgb, int h, int w) {
    for (int c = 0; c < 3; c++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float v = in[(c * h + y) * w + x];
                v = (v < 0.0f) ? 0.0f : (v > 1.0f) ? 1.0f : v;
                rgb[(y * w + x) * 3 + c] = (uint8_t)(v * 255.0f + 0.5f);
            }
}

void or_postprocess_range(const float* in, uint8_t* rgb, int h, int w, const float* mean, float range) {
    for (int c = 0; c < 3; c++)
        for (int y = 0; y < h; y++)
            for (int x = 0; x < w; x++) {
                float v = in[(c * h + y) * w + x] + mean[c] * range;
                v = v < 0.0f ? 0.0f : (v > 255.0f ? 255.0f : v);
                rgb[(y * w + x) * 3 + c] = (uint8_t)(v + 0.5f);
            }
}

void or_resize_bilinear(const uint8_t* in, int in_w, int in_h, uint8_t* out, int out_w, int out_h) {
    const float x_ratio = (float)in_w / out_w;
    const float y_ratio = (float)in_h / out_h;
    for (int y = 0; y < out_h; y++)
        for (int x = 0; x < out_w; x++) {
            const float px = x * x_ratio, py = y * y_ratio;
            const int x0 = (int)px, y0 = (int)py;
            int x1 = x0 + 1, y1 = y0 + 1;
            if (x1 >= in_w) x1 = in_w - 1;
            if (y1 >= in_h) y1 = in_h - 1;
            const float dx = px - x0, dy = py - y0;
            for (int c = 0; c < 3; c++) {
                const float p00 = in[(y0 * in_w + x0) * 3 + c], p01 = in[(y0 * in_w + x1) * 3 + c];
                const float p10 = in[(y1 * in_w + x0) * 3 + c], p11 = in[(y1 * in_w + x1) * 3 + c];
                const float val = (1 - dx) * (1 - dy) * p00 + dx * (1 - dy) * p01 + (1 - dx) * dy * p10 + dx * dy * p11;
                out[(y * out_w + x) * 3 + c] = (uint8_t)(val + 0.5f);
            }
        }
}
