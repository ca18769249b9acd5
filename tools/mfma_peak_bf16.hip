// bf16 MFMA sustained issue-rate probe (v_mfma_f32_32x32x16_bf16), long launches, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void mfma_loop(const bf16x8* in, float* out, int iters) {
    f32x16 acc0, acc1, acc2, acc3;
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; acc2[r] = 0.f; acc3[r] = 0.f; }
    bf16x8 v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[(blockIdx.x * 256 + threadIdx.x) * 8 + j];
    for (int i = 0; i < iters; ++i) {
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[0], v[1], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[2], v[3], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[4], v[5], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[6], v[7], acc3, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[7], v[2], acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[5], v[0], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[3], v[6], acc2, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(v[1], v[4], acc3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r] + acc2[r] + acc3[r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    const int wgs = 1024, n = wgs * 256 * 8 * 8;
    short* h = (short*)malloc(n * 2);
    bf16x8* in; float* out;
    (void)hipMalloc(&in, n * 2); (void)hipMalloc(&out, wgs * 256 * 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        for (int i = 0; i < n; ++i) { float f = mode == 0 ? 0.f : (float)rand() / RAND_MAX * 2.f - 1.f; unsigned u; memcpy(&u, &f, 4); h[i] = (short)(u >> 16); }
        (void)hipMemcpy(in, h, n * 2, hipMemcpyHostToDevice);
        const int iters = 40000;
        for (int rep = 0; rep < 4; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, in, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            double flops = (double)wgs * 4 * iters * 8 * 32768.0;
            printf("%s: %.3f ms  %.1f TFLOP/s bf16  (= %.1f TFLOP/s fp32-equivalent at 6 products)\n", mode == 0 ? "zeros" : "random", ms, flops / ms / 1e9, flops / ms / 1e9 / 6);
        }
    }
    return 0;
}
