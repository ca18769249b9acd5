// Bandwidth floor of the selection kernel's access pattern: R rows of V floats (row stride LD), each row reduced
// (max, then sum of exp) by one workgroup of T threads holding the row in registers.  Variants: threads per row,
// rows per workgroup (sequential, software-pipelined: the next row's loads are issued before the current row's
// reductions).   hipcc -O3 --offload-arch=gfx950 tools/rowscan_probe.hip -o tools/rowscan_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float wave_max(float v) { for (int o = 32; o; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64)); return v; }
__device__ __forceinline__ float wave_sum(float v) { for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o, 64); return v; }

template <int T, int PT, int MINW>
__global__ __launch_bounds__(T, MINW) void scan_rows(const float* __restrict__ x, int ld, int V, int rows_per_wg, float* __restrict__ out) {
    constexpr int W = T / 64;
    __shared__ float red[2][W];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    f32x4 cur[PT], nxt[PT];
    const int row0 = blockIdx.x * rows_per_wg;
    auto load = [&](int row, f32x4 (&v)[PT]) {
        const float* xr = x + (size_t)row * ld;
#pragma unroll
        for (int j = 0; j < PT; ++j) v[j] = *reinterpret_cast<const f32x4*>(xr + min(4 * (tid + j * T), (V - 1) & ~3));
    };
    load(row0, cur);
    for (int i = 0; i < rows_per_wg; ++i) {
        if (i + 1 < rows_per_wg) load(row0 + i + 1, nxt);
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < PT; ++j) {
            const int c0 = 4 * (tid + j * T);
#pragma unroll
            for (int e = 0; e < 4; ++e) { if (c0 + e >= V) cur[j][e] = -INFINITY; m = fmaxf(m, cur[j][e]); }
        }
        m = wave_max(m);
        if (lane == 0) red[0][wave] = m;
        __syncthreads();
        m = red[0][0];
#pragma unroll
        for (int w = 1; w < W; ++w) m = fmaxf(m, red[0][w]);
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < PT; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __expf(cur[j][e] - m);
        s = wave_sum(s);
        if (lane == 0) red[1][wave] = s;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < W; ++w) tot += red[1][w];
        if (tid == 0) out[row0 + i] = m + logf(tot);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < PT; ++j) cur[j] = nxt[j];
    }
}

template <int T, int PT, int MINW>
float run(const float* x, int ld, int V, int rows, int rpw, float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((scan_rows<T, PT, MINW>), dim3(rows / rpw), dim3(T), 0, 0, x, ld, V, rpw, out);
    hipEventRecord(e0);
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((scan_rows<T, PT, MINW>), dim3(rows / rpw), dim3(T), 0, 0, x, ld, V, rpw, out);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms / iters * 1e3f;
}

int main() {
    const int rows = 1280, V = 10201, ld = 10204, iters = 50;
    float *x, *out, *big;
    hipMalloc(&x, (size_t)rows * ld * 4); hipMalloc(&out, rows * 4);
    std::vector<float> h((size_t)rows * ld);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 2000) / 100.f - 10.f;
    hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    const double mb = (double)rows * V * 4 / 1e6;
#define R(T, PT, MW, RPW) { float us = run<T, PT, MW>(x, ld, V, rows, RPW, out, iters); \
    printf("threads %4d  float4/thread %2d  minwaves %d  rows/wg %d : %6.2f us  %.2f TB/s\n", T, PT, MW, RPW, us, mb / us / 1e6 * 1e6 / 1e6); }
    R(256, 10, 1, 1) R(256, 10, 5, 1) R(512, 5, 1, 1) R(1024, 3, 1, 1) R(256, 10, 1, 5) R(256, 10, 1, 2) R(512, 5, 1, 5) R(128, 20, 1, 1) R(64, 40, 1, 1)
    return 0;
}
