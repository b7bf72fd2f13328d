// C ABI behind include/mi355_executor.h: the backend seam of the reference runners (NeuronExecutor,
// neuron_executor.h:85-117) on the model executor.  Run() = pinned staging -> H2D -> srmi_model_forward_f32
// -> D2H, on a stream of its own; dense float32 NCHW host buffers on both sides, as NeuronExecutor::Run
// copies them (neuron_executor.cpp:456-494).
#include <string.h>

#include "../../include/mi355_executor.h"
#include "common.h"

struct srmi_executor {
    srmi_model_t* model = nullptr;
    int device = 0, n = 0, h = 0, w = 0, scale = 0;
    size_t in_bytes = 0, out_bytes = 0;
    hipStream_t stream = nullptr;
    float *pin_in = nullptr, *pin_out = nullptr, *dev_in = nullptr, *dev_out = nullptr;
};

namespace {
struct DeviceScope {
    int prev = 0;
    explicit DeviceScope(int device) {
        (void)hipGetDevice(&prev);
        (void)hipSetDevice(device);
    }
    ~DeviceScope() { (void)hipSetDevice(prev); }
};
}  // namespace

extern "C" void srmi_executor_destroy(srmi_executor_t* ex) {
    if (!ex) return;
    DeviceScope scope(ex->device);
    if (ex->stream) (void)hipStreamSynchronize(ex->stream);
    if (ex->pin_in) (void)hipHostFree(ex->pin_in);
    if (ex->pin_out) (void)hipHostFree(ex->pin_out);
    if (ex->dev_in) (void)hipFree(ex->dev_in);
    if (ex->dev_out) (void)hipFree(ex->dev_out);
    if (ex->model) srmi_model_destroy(ex->model);
    if (ex->stream) (void)hipStreamDestroy(ex->stream);
    delete ex;
}

extern "C" int srmi_executor_create(const char* model_path, const uint32_t* input_shape, int input_rank,
                                    const uint32_t* output_shape, int output_rank, srmi_executor_t** out) {
    if (!model_path || !input_shape || !output_shape || !out) SRMI_FAIL("srmi_executor_create: null argument");
    if (input_rank != 4 || output_rank != 4) SRMI_FAIL("srmi_executor_create: shapes must be [N,3,H,W]");
    if (input_shape[1] != 3 || output_shape[1] != 3 || input_shape[0] == 0 || input_shape[0] != output_shape[0] ||
        input_shape[2] == 0 || input_shape[3] == 0)
        SRMI_FAIL("srmi_executor_create: expected [N,3,H,W] -> [N,3,H*scale,W*scale]");
    int device = 0;
    if (srmi_pick_device(&device)) return -1;
    srmi_model_t* m = nullptr;
    srmi_model_desc_t d;
    if (srmi_load_weight_file(model_path, 0, device, &m, &d)) return -1;
    srmi_executor* ex = new srmi_executor();
    ex->model = m;
    ex->device = device;
    ex->n = (int)input_shape[0]; ex->h = (int)input_shape[2]; ex->w = (int)input_shape[3];
    ex->scale = d.scale;
    if (output_shape[2] != input_shape[2] * (uint32_t)d.scale || output_shape[3] != input_shape[3] * (uint32_t)d.scale) {
        srmi::set_error("srmi_executor_create: output shape %ux%u does not match input %ux%u at the model's scale %d",
                        output_shape[2], output_shape[3], input_shape[2], input_shape[3], d.scale);
        srmi_executor_destroy(ex);
        return -1;
    }
    ex->in_bytes = (size_t)ex->n * 3 * ex->h * ex->w * sizeof(float);
    ex->out_bytes = ex->in_bytes * d.scale * d.scale;
    DeviceScope scope(device);
    const bool ok = hipStreamCreate(&ex->stream) == hipSuccess &&
                    hipHostMalloc((void**)&ex->pin_in, ex->in_bytes, hipHostMallocDefault) == hipSuccess &&
                    hipHostMalloc((void**)&ex->pin_out, ex->out_bytes, hipHostMallocDefault) == hipSuccess &&
                    hipMalloc((void**)&ex->dev_in, ex->in_bytes) == hipSuccess &&
                    hipMalloc((void**)&ex->dev_out, ex->out_bytes) == hipSuccess;
    if (!ok) srmi::set_error("srmi_executor_create: buffer allocation failed: %s", hipGetErrorString(hipGetLastError()));
    if (!ok || srmi_model_reserve(m, ex->n, ex->h, ex->w)) {
        srmi_executor_destroy(ex);
        return -1;
    }
    *out = ex;
    return 0;
}

extern "C" int srmi_executor_run(srmi_executor_t* ex, const void* const* inputs, size_t n_inputs, void* const* outputs,
                                 size_t n_outputs) {
    if (!ex) SRMI_FAIL("srmi_executor_run: executor not initialised");
    // neuron_executor.cpp:439-447: the counts must match the model's
    if (n_inputs != 1 || n_outputs != 1 || !inputs || !outputs || !inputs[0] || !outputs[0])
        SRMI_FAIL("srmi_executor_run: expected exactly one input and one output buffer");
    DeviceScope scope(ex->device);
    memcpy(ex->pin_in, inputs[0], ex->in_bytes);
    SRMI_HIP(hipMemcpyAsync(ex->dev_in, ex->pin_in, ex->in_bytes, hipMemcpyHostToDevice, ex->stream));
    if (srmi_model_forward_f32(ex->model, ex->dev_in, ex->dev_out, ex->n, ex->h, ex->w, ex->stream)) return -1;
    SRMI_HIP(hipMemcpyAsync(ex->pin_out, ex->dev_out, ex->out_bytes, hipMemcpyDeviceToHost, ex->stream));
    SRMI_HIP(hipStreamSynchronize(ex->stream));
    memcpy(outputs[0], ex->pin_out, ex->out_bytes);
    return 0;
}

extern "C" size_t srmi_executor_input_size(const srmi_executor_t* ex, size_t index) {
    return (ex && index == 0) ? ex->in_bytes : 0;
}

extern "C" size_t srmi_executor_output_size(const srmi_executor_t* ex, size_t index) {
    return (ex && index == 0) ? ex->out_bytes : 0;
}
