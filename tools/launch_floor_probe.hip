// Where does the ~5 us floor of a small kernel come from?  Dependent chains of (a) empty kernels, (b) kernels whose
// workgroups do one cold 16-byte load per thread and a store, (c) the same from an L2-hot buffer; 640 and 320
// workgroups of 256 threads, plain stream order and hipGraph replay.  Prints ns per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void empty_kernel(float*) {}
__global__ void touch_kernel(const f32x4* __restrict__ in, f32x4* __restrict__ out, size_t stride4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    out[i] = in[i + stride4];
}
template <typename F> static float time_chain(F launch, int n, hipStream_t s, bool graph) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    float ms = 0;
    if (!graph) {
        for (int i = 0; i < 20; ++i) launch(i);
        (void)hipEventRecord(e0, s);
        for (int i = 0; i < n; ++i) launch(i);
        (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
    } else {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) launch(i);
        (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        (void)hipEventRecord(e0, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
    }
    return ms * 1e6f / n;
}
int main() {
    hipStream_t s; (void)hipStreamCreate(&s);
    const size_t big = (size_t)1 << 30;                       // 1 GiB: every launch reads a different 2.6 MB window (cold)
    f32x4 *in, *out; (void)hipMalloc(&in, big); (void)hipMalloc(&out, 640 * 256 * 16);
    (void)hipMemset(in, 0, big);
    const int n = 400;
    for (int wgs : {320, 640}) {
        for (int graph = 0; graph < 2; ++graph) {
            const float e = time_chain([&](int) { hipLaunchKernelGGL(empty_kernel, dim3(wgs), dim3(256), 0, s, (float*)out); }, n, s, graph);
            const float cold = time_chain([&](int i) { hipLaunchKernelGGL(touch_kernel, dim3(wgs), dim3(256), 0, s, in, out, (size_t)(i % 300) * 640 * 256); }, n, s, graph);
            const float hot = time_chain([&](int) { hipLaunchKernelGGL(touch_kernel, dim3(wgs), dim3(256), 0, s, in, out, (size_t)0); }, n, s, graph);
            printf("%3d workgroups, %s: empty %.0f ns, one cold load + store %.0f ns, one L2-hot load + store %.0f ns per dependent launch\n",
                   wgs, graph ? "hipGraph replay" : "stream launches", e, cold, hot);
        }
    }
    return 0;
}
