/*
 * Generic runner API: one set of entry points for all three networks, any input size, no resize.
 *
 * The reference has one runner per network with a fixed model input size (edsr.h:52-92 256x256,
 * rcan.h 510x339, realesrgan.h:48-88 510x339) because its .dla blobs are compiled for one shape, and
 * resizes every other input to it (edsr.cpp:278-297).  This engine takes any size, so sr_* is those four
 * functions folded into three: the architecture and scale come from the weight file's header, an input of
 * w x h gives an output of (w*scale) x (h*scale).  Same conventions as the per-network runners: host
 * memory, u8 HWC RGB, the callee malloc()s *output_rgb and the caller free()s it, 0 / -1 return codes,
 * pre/post-processing of the network's own runner (edsr.cpp:21-41,101-122; realesrgan.cpp:28-48,110-130,
 * rounding (uint8)(v + 0.5f)).  A context binds to $SRMI_DEVICE or the calling thread's current HIP device
 * at sr_init and is not re-entrant; different contexts are independent.
 */
#ifndef SRMI_SR_H_
#define SRMI_SR_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    void* executor;     /* engine handle (opaque) */
    int arch;           /* 1 EDSR, 2 RCAN, 3 RRDBNet (Real-ESRGAN): SRMI_ARCH_* of srmi355.h */
    int scale_factor;   /* 2, 3, 4 or 8 */
    int channels;       /* 3 */
    int initialized;
} sr_context_t;

/* Loads a weight file written by superresolution_amd.export.save_weights() (any of the three networks). */
int sr_init(const char* model_path, sr_context_t* ctx);
/* 0 on an uninitialised or NULL context, as edsr_release (edsr.cpp:185-187). */
int sr_release(sr_context_t* ctx);
/* input_rgb: input_width x input_height x 3 bytes.  output_width / output_height may be NULL. */
int sr_inference(sr_context_t* ctx, const unsigned char* input_rgb, int input_width, int input_height,
                 unsigned char** output_rgb, int* output_width, int* output_height);
/* Allocates everything inferences of up to input_width x input_height need (workspace, pinned staging), so that
 * the first sr_inference pays for no allocation; optional. */
int sr_reserve(sr_context_t* ctx, int input_width, int input_height);
/* Message of the calling thread's last failed call into the library (same string as srmi_last_error). */
const char* sr_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
