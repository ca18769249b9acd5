// Wave-wide all-reduce without LDS: four DPP steps inside each row of 16 lanes (quad_perm xor 1, xor 2, row_half_mirror,
// row_mirror) and the gfx950 row exchanges v_permlane16_swap / v_permlane32_swap, against the __shfl_xor butterfly
// (ds_bpermute_b32 through the LDS crossbar).  Checks the results and times a dependent chain of each.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// hipcc (ROCm 7.2) folds the two results of a swap builtin into one value when both inputs are the same value
// (v_add v1, v1, v1 after the swap): the empty asm statements keep inputs and outputs distinct.
template <bool kRows32> __device__ __forceinline__ void swap_rows(float v, float& x, float& y) {
    int a = __builtin_bit_cast(int, v), b = a;
    asm volatile("" : "+v"(b));
    int x0, x1;
    if (kRows32) { auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false); x0 = r[0]; x1 = r[1]; }
    else { auto r = __builtin_amdgcn_permlane16_swap(a, b, false, false); x0 = r[0]; x1 = r[1]; }
    asm volatile("" : "+v"(x0), "+v"(x1));
    x = __builtin_bit_cast(float, x0); y = __builtin_bit_cast(float, x1);
}
__device__ __forceinline__ float swap16(float v) { float x, y; swap_rows<false>(v, x, y); return x + y; }
__device__ __forceinline__ float swap32(float v) { float x, y; swap_rows<true>(v, x, y); return x + y; }
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp<0xB1>(v); v += dpp<0x4E>(v); v += dpp<0x141>(v); v += dpp<0x140>(v);
    return swap32(swap16(v));
}
__device__ __forceinline__ float wave_sum_shfl(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
template <int MODE> __global__ void chain(const float* in, float* out, int iters) {
    float v = in[threadIdx.x];
    for (int i = 0; i < iters; ++i) v = (MODE ? wave_sum_dpp(v) : wave_sum_shfl(v)) * (1.0f / 64.0f) + in[(threadIdx.x + i) & 63] * 1e-3f;
    out[blockIdx.x * 64 + threadIdx.x] = v;
}
__global__ void once(const float* in, float* a, float* b) {
    a[threadIdx.x] = wave_sum_dpp(in[threadIdx.x]);
    b[threadIdx.x] = wave_sum_shfl(in[threadIdx.x]);
}
int main() {
    float h[64], *d, *a, *b;
    for (int i = 0; i < 64; ++i) h[i] = (float)(i * i % 17) + 0.25f * i;
    double ref = 0; for (int i = 0; i < 64; ++i) ref += h[i];
    (void)hipMalloc(&d, 256); (void)hipMalloc(&a, 1024 * 256); (void)hipMalloc(&b, 256);
    (void)hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(once, dim3(1), dim3(64), 0, 0, d, a, b);
    float ra[64], rb[64]; (void)hipMemcpy(ra, a, 256, hipMemcpyDeviceToHost); (void)hipMemcpy(rb, b, 256, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < 64; ++i) if (ra[i] != ra[0] || ra[i] != (float)ref) ++bad;
    printf("exact sum %.3f  dpp %.3f  shfl %.3f  lanes differing %d\n", ref, ra[0], rb[0], bad);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 2; ++mode) {
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            if (mode) hipLaunchKernelGGL(chain<1>, dim3(1024), dim3(64), 0, 0, d, a, 2000); else hipLaunchKernelGGL(chain<0>, dim3(1024), dim3(64), 0, 0, d, a, 2000);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
        }
        printf("%s: %.1f ns per dependent wave all-reduce\n", mode ? "dpp + permlane swaps" : "__shfl_xor butterfly ", ms * 1e6 / 2000);
    }
    return bad != 0;
}
