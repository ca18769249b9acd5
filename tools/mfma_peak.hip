// Bare v_mfma_f32_32x32x2_f32 issue-rate probe: what fp32 MFMA rate does this chip sustain under load?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters, float a0, float b0) {
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    float a = a0 + threadIdx.x * 1e-3f, b = b0 - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, a, acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, b, acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wgs : {256, 512, 1024}) {
        const int iters = 20000;
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, out, iters, 0.37f, 0.61f);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)wgs * 4 * iters * 4 * 4096.0;
            printf("wgs=%d iters=%d: %.3f ms  %.1f TFLOP/s\n", wgs, iters, ms, flops / ms / 1e9);
        }
    }
    return 0;
}
