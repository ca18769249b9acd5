// fp32 MFMA (32x32x2) throughput vs independent accumulator chains per wave (NACC) and waves per SIMD (grid = 256*W
// workgroups of 4 waves).  Long launches, random operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void loop(const float* in, float* out, int iters) {
    f32x16 c[NACC];
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) c[t][r] = 0.f;
    float v[2 * NACC];
    for (int j = 0; j < 2 * NACC; ++j) v[j] = in[(blockIdx.x * 256 + threadIdx.x) * 16 + (j & 15)];
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) c[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2 * t], v[2 * t + 1], c[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) for (int r = 0; r < 16; ++r) s += c[t][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC> void run(const float* in, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int W : {1, 2, 4, 8}) {
        if (NACC * 16 * W > 480) continue;                   // would not be co-resident
        const int wgs = 256 * W, iters = 320000 / (NACC * W);
        float best = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(loop<NACC>, dim3(wgs), dim3(256), 0, 0, in, out, iters);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const float tf = (double)wgs * 4 * iters * NACC * 4096.0 / ms / 1e9;
            if (rep > 0 && tf > best) best = tf;
        }
        printf("chains/wave=%d waves/SIMD=%d (chains/SIMD=%2d): %.1f TFLOP/s\n", NACC, W, NACC * W, best);
    }
}
int main() {
    const int n = 2048 * 256 * 16;
    float *in, *out, *h = (float*)malloc(n * 4);
    (void)hipMalloc(&in, n * 4); (void)hipMalloc(&out, 2048 * 256 * 4);
    for (int i = 0; i < n; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice);
    run<1>(in, out); run<2>(in, out); run<4>(in, out); run<8>(in, out); run<16>(in, out);
    return 0;
}
