// What makes an fp32 MFMA stream fall from ~153 to ~124 TFLOP/s?  Variants of a bare 32x32x2 loop:
//   A: 4 MFMAs / iteration, operand pairs in adjacent registers      (the 153 TF case)
//   B: 8 MFMAs / iteration, same 4 accumulators used twice, pairs (v0,v1)...(v6,v7) then crossed pairs
//   C: 8 MFMAs / iteration, 8 accumulators, adjacent pairs
//   D: 4 MFMAs / iteration, crossed (non-adjacent) pairs
//   E: 8 MFMAs / iteration, 4 accumulators used twice, adjacent pairs both times
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define M(acc, a, b) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0)
template <int VARIANT>
__global__ __launch_bounds__(256) void loop(const float* in, float* out, int iters) {
    f32x16 c[8];
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + j];
    for (int i = 0; i < iters; ++i) {
        if (VARIANT == 0) { M(c[0], v[0], v[1]); M(c[1], v[2], v[3]); M(c[2], v[4], v[5]); M(c[3], v[6], v[7]); }
        if (VARIANT == 1) { M(c[0], v[0], v[1]); M(c[1], v[2], v[3]); M(c[2], v[4], v[5]); M(c[3], v[6], v[7]);
                            M(c[0], v[7], v[2]); M(c[1], v[5], v[0]); M(c[2], v[3], v[6]); M(c[3], v[1], v[4]); }
        if (VARIANT == 2) { M(c[0], v[0], v[1]); M(c[1], v[2], v[3]); M(c[2], v[4], v[5]); M(c[3], v[6], v[7]);
                            M(c[4], v[8], v[9]); M(c[5], v[10], v[11]); M(c[6], v[12], v[13]); M(c[7], v[14], v[15]); }
        if (VARIANT == 3) { M(c[0], v[7], v[2]); M(c[1], v[5], v[0]); M(c[2], v[3], v[6]); M(c[3], v[1], v[4]); }
        if (VARIANT == 4) { M(c[0], v[0], v[1]); M(c[1], v[2], v[3]); M(c[2], v[4], v[5]); M(c[3], v[6], v[7]);
                            M(c[0], v[8], v[9]); M(c[1], v[10], v[11]); M(c[2], v[12], v[13]); M(c[3], v[14], v[15]); }
    }
    float s = 0.f;
    for (int t = 0; t < 8; ++t) for (int r = 0; r < 16; ++r) s += c[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int V> void run(const char* name, const float* in, float* out, int per_iter) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int wgs = 1024, iters = 160000 / per_iter;
    for (int rep = 0; rep < 2; ++rep) {
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(loop<V>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        printf("%s: %.3f ms  %.1f TFLOP/s\n", name, ms, (double)wgs * 4 * iters * per_iter * 4096.0 / ms / 1e9);
    }
}
int main() {
    const int n = 1024 * 256 * 16;
    float *in, *out, *h = (float*)malloc(n * 4);
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, 1024 * 256 * 4);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    run<0>("A 4/iter adjacent        ", in, out, 4);
    run<1>("B 8/iter 4 acc crossed   ", in, out, 8);
    run<2>("C 8/iter 8 acc adjacent  ", in, out, 8);
    run<3>("D 4/iter crossed         ", in, out, 4);
    run<4>("E 8/iter 4 acc adjacent  ", in, out, 8);
    run<0>("A again                  ", in, out, 4);
    return 0;
}
