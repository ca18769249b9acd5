// Does the sustained fp32 MFMA rate depend on the instruction shape?  32x32x2 vs 16x16x4, random operands, long launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void loop32(const float* in, float* out, int iters) {
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[(blockIdx.x * 256 + threadIdx.x) * 8 + j];
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], v[3], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[4], v[5], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[6], v[7], a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void loop16(const float* in, float* out, int iters) {
    f32x4 a[8];
    for (int t = 0; t < 8; ++t) a[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[(blockIdx.x * 256 + threadIdx.x) * 8 + j];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 8; ++t) a[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(v[t], v[(t + 3) & 7], a[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) s += a[t][0] + a[t][1] + a[t][2] + a[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int wgs = 1024, n = wgs * 256 * 8;
    float *in, *out, *h = (float*)malloc(n * 4);
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, wgs * 256 * 4);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        for (int shape = 0; shape < 2; ++shape) {
            const int iters = 40000;
            (void)hipEventRecord(e0);
            if (shape == 0) hipLaunchKernelGGL(loop32, dim3(wgs), dim3(256), 0, 0, in, out, iters);
            else hipLaunchKernelGGL(loop16, dim3(wgs), dim3(256), 0, 0, in, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const double flops = shape == 0 ? (double)wgs * 4 * iters * 4 * 4096.0 : (double)wgs * 4 * iters * 8 * 2048.0;
            printf("%s: %.3f ms  %.1f TFLOP/s\n", shape == 0 ? "32x32x2" : "16x16x4", ms, flops / ms / 1e9);
        }
    }
    return 0;
}
