// (tools/launch_floor_probe.hip is the round-2 sibling: 320 / 640 workgroups, cold and hot loads, stream order against graph replay.)
// What does ONE dependent launch cost in a replayed hipGraph, whatever the kernel does?  (B = 1 in <= 2.0 ms.)
// Chains of 1000 dependent kernel nodes, replayed; ns per node:
//   empty        one workgroup, no memory access
//   touch W      W workgroups of 256 threads; each thread reads one float4 the PREVIOUS node wrote (another workgroup's, so the
//                line comes from another CU / XCD) and writes one: the least a phase of a decode step does
//   gemv W       the same plus a 512-deep dot product per thread against a weight row that stays in L2 (2 KB per thread):
//                a stand-in for a 5-row product's own work
// A decode step at B = 1 is a sequence of phases in which every output needs ALL outputs of the phase before (projection ->
// attention over all heads -> projection -> LayerNorm over the whole row -> ...): one launch (or one grid barrier) per phase.
//   hipcc -O3 --offload-arch=gfx950 tools/phase_floor_probe.hip -o tools/phase_floor_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void empty_kernel() {}

__global__ __launch_bounds__(256) void touch_kernel(const float4* __restrict__ in, float4* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = (i + 4096 + 17) % n;                       // another workgroup's element
    float4 v = in[j];
    v.x += 1.f;
    out[i] = v;
}

__global__ __launch_bounds__(256) void gemv_kernel(const float4* __restrict__ in, const float4* __restrict__ w, float4* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int j = (i + 4096 + 17) % n;
    float4 v = in[j];
    const float4* row = w + (size_t)(i % 2048) * 128;        // 2 KB per thread, 4 MB in all: L2-resident after the first node
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
#pragma unroll 8
    for (int k = 0; k < 128; ++k) {
        const float4 a = row[k];
        acc0 = fmaf(a.x, v.x, acc0); acc1 = fmaf(a.y, v.y, acc1); acc2 = fmaf(a.z, v.z, acc2); acc3 = fmaf(a.w, v.w, acc3);
    }
    out[i] = float4{acc0, acc1, acc2, acc3};
}

int main() {
    const int n_nodes = 1000, max_wg = 256, n = max_wg * 256;
    hipStream_t s; (void)hipStreamCreate(&s);
    float4 *a, *b, *w;
    (void)hipMalloc(&a, n * sizeof(float4)); (void)hipMalloc(&b, n * sizeof(float4)); (void)hipMalloc(&w, 2048 * 128 * sizeof(float4));
    (void)hipMemset(a, 0, n * sizeof(float4)); (void)hipMemset(b, 0, n * sizeof(float4)); (void)hipMemset(w, 0, 2048 * 128 * sizeof(float4));
    auto chain = [&](int kind, int wgs) {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n_nodes; ++i) {
            float4* in = (i & 1) ? b : a; float4* out = (i & 1) ? a : b;
            if (kind == 0) hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s);
            else if (kind == 1) hipLaunchKernelGGL(touch_kernel, dim3(wgs), dim3(256), 0, s, in, out, wgs * 256);
            else hipLaunchKernelGGL(gemv_kernel, dim3(wgs), dim3(256), 0, s, in, w, out, wgs * 256);
        }
        (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        return best * 1e6f / n_nodes;
    };
    printf("dependent kernel nodes in a replayed hipGraph, ns per node (best of 5 replays of 1000 nodes)\n");
    printf("  empty kernel (1 workgroup)            %6.0f\n", chain(0, 1));
    for (int wgs : {8, 32, 64, 256}) printf("  touch, %3d workgroups                 %6.0f\n", wgs, chain(1, wgs));
    for (int wgs : {8, 32, 64, 256}) printf("  touch + 512-deep dot, %3d workgroups  %6.0f\n", wgs, chain(2, wgs));
    return 0;
}
