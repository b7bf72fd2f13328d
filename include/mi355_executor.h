/*
 * Mi355Executor — the MI355X stand-in for the reference runners' backend seam, NeuronExecutor
 * (realesrgan/mtk/cpp/jni/src/mtk_npu/neuron_executor.h:85-117): constructed from a model file and the
 * input / output tensor shapes, Initialize() once, then Run(inputs, outputs) on dense float32 NCHW host
 * buffers.  A runner written against NeuronExecutor (edsr.cpp:148-157, realesrgan.cpp:154-163,241) keeps its
 * own pre- and post-processing and swaps only the type name and the model file.
 *
 * The model file is an SRMIW001 weight file (superresolution_amd/export.py) instead of a .dla blob.  As the
 * .dla holds the network without MeanShift, EDSR / RCAN files for this seam are written from the *Core
 * models (edsr_model.py:146, rcan_model.py:166); a file of a full model also works and then includes the
 * MeanShift layers, exactly as forward() of the full PyTorch module does.
 *
 * The class is header-only C++ over a plain C ABI (srmi_executor_*), so no C++ object crosses the
 * library boundary and the including translation unit needs neither HIP headers nor hipcc.
 */
#ifndef MI355_EXECUTOR_H_
#define MI355_EXECUTOR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct srmi_executor srmi_executor_t;

/* NeuronExecutor::NeuronExecutor + Initialize (neuron_executor.cpp:396-431): loads the weight file onto the
 * GPU ($SRMI_DEVICE, else the current HIP device), checks the shapes ([1,3,H,W] -> [1,3,H*scale,W*scale],
 * batch N > 1 allowed) and allocates the staging buffers.  0 / -1, message in srmi_last_error(). */
int srmi_executor_create(const char* model_path, const uint32_t* input_shape, int input_rank,
                         const uint32_t* output_shape, int output_rank, srmi_executor_t** out);
void srmi_executor_destroy(srmi_executor_t* ex);
/* NeuronExecutor::Run (neuron_executor.cpp:433-496): one float32 NCHW host buffer in, one out; synchronous. */
int srmi_executor_run(srmi_executor_t* ex, const void* const* inputs, size_t n_inputs, void* const* outputs,
                      size_t n_outputs);
/* GetInputSize / GetOutputSize: bytes of tensor `index` (0 when out of range). */
size_t srmi_executor_input_size(const srmi_executor_t* ex, size_t index);
size_t srmi_executor_output_size(const srmi_executor_t* ex, size_t index);

#ifdef __cplusplus
}  /* extern "C" */

#include <string>
#include <vector>

class Mi355Executor {
public:
    Mi355Executor(const std::string& model_path, const std::vector<std::vector<uint32_t>>& input_shapes,
                  const std::vector<std::vector<uint32_t>>& output_shapes, const std::string& name = "Mi355Executor")
        : model_path_(model_path), name_(name), input_shapes_(input_shapes), output_shapes_(output_shapes) {}
    ~Mi355Executor() { srmi_executor_destroy(ex_); }
    Mi355Executor(const Mi355Executor&) = delete;
    Mi355Executor& operator=(const Mi355Executor&) = delete;

    bool Initialize() {
        if (ex_) return true;
        if (input_shapes_.size() != 1 || output_shapes_.size() != 1) return false;
        return srmi_executor_create(model_path_.c_str(), input_shapes_[0].data(), (int)input_shapes_[0].size(),
                                    output_shapes_[0].data(), (int)output_shapes_[0].size(), &ex_) == 0;
    }
    bool Run(const std::vector<const void*>& inputs, std::vector<void*>& outputs) {
        if (!ex_) return false;
        return srmi_executor_run(ex_, inputs.data(), inputs.size(), outputs.data(), outputs.size()) == 0;
    }
    /* Bytes of tensor `index` as dense float32, from the constructor's shapes, so that they are valid
     * before Initialize() as in the reference (neuron_executor.cpp:498-520); 0 when out of range. */
    size_t GetInputSize(size_t index) const { return index < input_shapes_.size() ? Bytes(input_shapes_[index]) : 0; }
    size_t GetOutputSize(size_t index) const { return index < output_shapes_.size() ? Bytes(output_shapes_[index]) : 0; }
    size_t GetNumInputs() const { return input_shapes_.size(); }
    size_t GetNumOutputs() const { return output_shapes_.size(); }
    bool IsInitialized() const { return ex_ != nullptr; }

private:
    static size_t Bytes(const std::vector<uint32_t>& shape) {
        size_t n = sizeof(float);
        for (uint32_t d : shape) n *= d;
        return n;
    }
    std::string model_path_, name_;
    std::vector<std::vector<uint32_t>> input_shapes_, output_shapes_;
    srmi_executor_t* ex_ = nullptr;
};
#endif /* __cplusplus */
#endif /* MI355_EXECUTOR_H_ */
