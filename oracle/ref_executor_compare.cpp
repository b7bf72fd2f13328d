// Test infrastructure only: the reference's NeuronExecutor (its own neuron_executor.cpp, compiled from where it
// lies) and Mi355Executor side by side, on everything that needs neither the vendor runtime nor a GPU: counts,
// tensor sizes before Initialize(), Run() before Initialize(), Initialize() on a missing model file.
// Built and run by tests/test_runner_errors_vs_reference.py into oracle/_ref/ (build container only).
#include <cstdio>
#include <vector>

#include "mi355_executor.h"
#include "mtk_npu/neuron_executor.h"

template <class A, class B>
static int compare(A& ref, B& mine, const char* what) {
    int bad = 0;
#define SAME(expr)                                                                                          \
    do {                                                                                                    \
        auto a = ref.expr;                                                                                  \
        auto b = mine.expr;                                                                                 \
        if (!(a == b)) { printf("%s: %s differs (%lld vs %lld)\n", what, #expr, (long long)a, (long long)b); ++bad; } \
    } while (0)
    SAME(GetNumInputs());
    SAME(GetNumOutputs());
    SAME(GetInputSize(0));
    SAME(GetOutputSize(0));
    SAME(GetInputSize(1));
    SAME(GetOutputSize(7));
    SAME(IsInitialized());
    std::vector<float> in(16), out(16);
    std::vector<const void*> ins = {in.data()};
    std::vector<void*> outs = {out.data()};
    SAME(Run(ins, outs));          // not initialised: false
    SAME(Initialize());            // missing model file: false
    SAME(IsInitialized());
    SAME(Run(ins, outs));
#undef SAME
    return bad;
}

int main() {
    int bad = 0;
    struct Case { const char* name; std::vector<uint32_t> in, out; };
    const Case cases[] = {{"edsr", {1, 3, 256, 256}, {1, 3, 1024, 1024}},
                          {"rcan", {1, 3, 339, 510}, {1, 3, 1356, 2040}},
                          {"realesrgan", {1, 3, 339, 510}, {1, 3, 1356, 2040}}};
    for (const Case& c : cases) {
        NeuronExecutor ref("/nonexistent/model.dla", {c.in}, {c.out}, c.name);
        Mi355Executor mine("/nonexistent/model.bin", {c.in}, {c.out}, c.name);
        bad += compare(ref, mine, c.name);
    }
    printf(bad ? "executor semantics DIFFER\n" : "executor semantics match\n");
    return bad ? 1 : 0;
}
