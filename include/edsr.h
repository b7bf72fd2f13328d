/*
 * Runner API of the EDSR demo, backed by the MI355X engine.
 * Drop-in for the reference header edsr/mtk/cpp/jni/src/edsr.h:20-92: same context struct layout, same
 * four functions, same ownership (the callee malloc()s *output_rgb, the caller free()s it) and
 * return codes (0 / -1).  `model_path` names a weight file written by
 * superresolution_amd.export.save_weights() instead of a .dla blob; `neuron_executor` holds the
 * engine handle.  Buffers are host memory, u8 HWC RGB.  Fixed model size: 256 x 256 -> 1024 x 1024.
 */
#ifndef SRMI_EDSR_H_
#define SRMI_EDSR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    void* neuron_executor; /* engine handle (opaque) */
    int input_width;
    int input_height;
    int input_channels;
    int output_width;
    int output_height;
    int output_channels;
    int scale_factor;
    float rgb_mean[3];     /* 0.4488, 0.4371, 0.4040 (edsr.cpp:169-171) */
    float rgb_range;       /* 255 */
    size_t input_size;
    size_t output_size;
    int initialized;
} edsr_context_t;

/* edsr.cpp:126-182. */
int edsr_init(const char* model_path, edsr_context_t* ctx);
/* edsr.cpp:184-200: returns 0 on an uninitialised context. */
int edsr_release(edsr_context_t* ctx);
/* edsr.cpp:202-207. */
int edsr_inference(edsr_context_t* ctx, const unsigned char* input_rgb, unsigned char** output_rgb);
/* edsr.cpp:209-300: bilinear resize when the size differs; p - mean*range -> net ->
 * clamp(v + mean*range, 0, 255) + 0.5; output_width / output_height may be NULL. */
int edsr_inference_ex(edsr_context_t* ctx, const unsigned char* input_rgb, int input_width, int input_height,
                      unsigned char** output_rgb, int* output_width, int* output_height);

#ifdef __cplusplus
}
#endif
#endif
