/*
 * Runner API of the Real-ESRGAN demo, backed by the MI355X engine.
 * Drop-in for the reference header realesrgan/mtk/cpp/jni/src/realesrgan.h:20-88: same context
 * struct layout, same four functions, same ownership (the callee malloc()s *output_rgb, the
 * caller free()s it) and return codes (0 / -1).  `model_path` names a weight file written by
 * superresolution_amd.export.save_weights() instead of a .dla blob; `neuron_executor` holds the
 * engine handle.  Buffers are host memory, u8 HWC RGB.
 */
#ifndef SRMI_REALESRGAN_H_
#define SRMI_REALESRGAN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    void* neuron_executor; /* engine handle (opaque) */
    int input_width;       /* 510 */
    int input_height;      /* 339 */
    int input_channels;
    int output_width;      /* 2040 */
    int output_height;     /* 1356 */
    int output_channels;
    int scale_factor;
    size_t input_size;     /* bytes of the float32 NCHW input the reference runner staged */
    size_t output_size;
    int initialized;
} realesrgan_context_t;

/* realesrgan.cpp:137-191: fixed 339x510 input, x4. */
int realesrgan_init(const char* model_path, realesrgan_context_t* ctx);
/* realesrgan.cpp:196-212: -1 on an uninitialised context. */
int realesrgan_release(realesrgan_context_t* ctx);
/* realesrgan.cpp:217-257: input must be input_width x input_height; p/255 -> net -> clamp(v,0,1)*255+0.5. */
int realesrgan_inference(realesrgan_context_t* ctx, const unsigned char* input_rgb, unsigned char** output_rgb);
/* realesrgan.cpp:262-305: bilinear resize to the model size first when the size differs; all
 * pointer arguments are required. */
int realesrgan_inference_ex(realesrgan_context_t* ctx, const unsigned char* input_rgb, int input_width,
                            int input_height, unsigned char** output_rgb, int* output_width,
                            int* output_height);

#ifdef __cplusplus
}
#endif
#endif
