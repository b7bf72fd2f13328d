// Drives Mi355Executor the way the reference runners drive NeuronExecutor
// (realesrgan/mtk/cpp/jni/src/realesrgan.cpp:154-163 construct + Initialize, :236-244 Run on float NCHW buffers).
// Plain C++17, built with g++: the translation unit sees only include/mi355_executor.h.
//   executor_demo model.srmi N H W scale in.f32 out.f32
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "mi355_executor.h"

extern "C" const char* srmi_last_error(void);

static int fail(const char* what) {
    std::fprintf(stderr, "[ERROR] %s: %s\n", what, srmi_last_error());
    return 1;
}

int main(int argc, char** argv) {
    if (argc != 8) {
        std::fprintf(stderr, "usage: %s model.srmi N H W scale in.f32 out.f32\n", argv[0]);
        return 2;
    }
    const uint32_t n = std::atoi(argv[2]), h = std::atoi(argv[3]), w = std::atoi(argv[4]), s = std::atoi(argv[5]);
    std::vector<std::vector<uint32_t>> in_shapes = {{n, 3, h, w}};
    std::vector<std::vector<uint32_t>> out_shapes = {{n, 3, h * s, w * s}};

    // a wrong output shape must be refused at Initialize
    {
        Mi355Executor bad(argv[1], in_shapes, {{n, 3, h * s + 1, w * s}}, "bad");
        if (bad.Initialize() || bad.IsInitialized()) { std::fprintf(stderr, "wrong output shape was accepted\n"); return 1; }
        std::vector<const void*> i0 = {nullptr};
        std::vector<void*> o0 = {nullptr};
        if (bad.Run(i0, o0)) { std::fprintf(stderr, "Run succeeded on an uninitialised executor\n"); return 1; }
    }

    auto ex = std::make_unique<Mi355Executor>(argv[1], in_shapes, out_shapes, "executor_demo");
    if (!ex->Initialize()) return fail("Initialize");
    if (!ex->IsInitialized() || ex->GetNumInputs() != 1 || ex->GetNumOutputs() != 1) return 1;
    const size_t in_bytes = (size_t)n * 3 * h * w * sizeof(float), out_bytes = in_bytes * s * s;
    if (ex->GetInputSize(0) != in_bytes || ex->GetOutputSize(0) != out_bytes || ex->GetInputSize(1) != 0) {
        std::fprintf(stderr, "size getters disagree with the shapes\n");
        return 1;
    }
    std::vector<float> in(in_bytes / sizeof(float)), out(out_bytes / sizeof(float));
    FILE* f = std::fopen(argv[6], "rb");
    if (!f || std::fread(in.data(), 1, in_bytes, f) != in_bytes) { std::fprintf(stderr, "cannot read %s\n", argv[6]); return 1; }
    std::fclose(f);
    for (int rep = 0; rep < 2; ++rep) {  // the executor is reusable
        std::vector<const void*> inputs = {in.data()};
        std::vector<void*> outputs = {out.data()};
        if (!ex->Run(inputs, outputs)) return fail("Run");
    }
    // two buffers where the model has one input: refused (neuron_executor.cpp:439-447)
    {
        std::vector<const void*> inputs = {in.data(), in.data()};
        std::vector<void*> outputs = {out.data()};
        if (ex->Run(inputs, outputs)) { std::fprintf(stderr, "Run accepted two inputs\n"); return 1; }
    }
    f = std::fopen(argv[7], "wb");
    if (!f || std::fwrite(out.data(), 1, out_bytes, f) != out_bytes) { std::fprintf(stderr, "cannot write %s\n", argv[7]); return 1; }
    std::fclose(f);
    std::printf("ok %zu -> %zu bytes\n", in_bytes, out_bytes);
    return 0;
}
