// Prototype: fp32 MFMA GEMM with an explicit two-group ("ping-pong") schedule.
//
// The product kernel (csrc/gemm.hip) runs 4 waves per workgroup, one per SIMD, and relies on a second co-resident
// workgroup to fill the matrix pipe while a wave waits for LDS / memory / barriers; measured, its main loop stops at
// 119-131 TFLOP/s of the 153 a bare MFMA stream reaches.  Here a workgroup has 8 waves = 2 per SIMD in two groups
// that alternate by construction: while group A issues the 64 MFMAs of K tile kt from fragment registers, group B
// does all of its memory work (LDS write of tile kt+1 from staging registers, global loads of tile kt+2, LDS
// fragment reads of tile kt), then they swap.  One s_barrier per half period; LDS double-buffered.
//
//   tile 256 x 128 x 32, wave tile 64 x 64 (2 x 2 v_mfma_f32_32x32x2_f32 accumulators), LDS 2 x 54 KB.
//   C[M,N] = A[M,K] . W[N,K]^T, M % 256 == N % 128 == K % 32 == 0 (prototype: no tails, no epilogue options).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 256, BN = 128, BK = 32, LDT = BK + 4;
constexpr int kTileFloats = (BM + BN) * LDT;
constexpr int kLoads = (BM + BN) * (BK / 4) / 512;     // float4 per thread per K tile = 6

__global__ __launch_bounds__(512) void gemm_pingpong(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                                     int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = wave >> 2, w4 = wave & 3;
    const int wm = group * 2 + (w4 >> 1), wn = w4 & 1;
    const int tiles_m = M / BM;
    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x % tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt = K / BK;

    // loader: element i of this thread is tile row (tid + 512 i) / 8, 16-byte column (tid + 512 i) % 8
    const float* src[kLoads];
    int dst[kLoads];
#pragma unroll
    for (int i = 0; i < kLoads; ++i) {
        const int idx = tid + i * 512, row = idx >> 3, kq = idx & 7;
        src[i] = row < BM ? A + (size_t)(m0 + row) * K + kq * 4 : W + (size_t)(n0 + row - BM) * K + kq * 4;
        dst[i] = row * LDT + kq * 4;
    }
    f32x4 st[kLoads];
    auto g_load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < kLoads; ++i) st[i] = *reinterpret_cast<const f32x4*>(src[i] + kt * BK);
    };
    auto l_write = [&](int kt) {
        float* buf = lds + (kt & 1) * kTileFloats;
#pragma unroll
        for (int i = 0; i < kLoads; ++i) *reinterpret_cast<f32x4*>(buf + dst[i]) = st[i];
    };
    const int frag_row = lane & 31, frag_k = (lane >> 5) * 4;
    f32x4 fa[2][4], fb[2][4];
    auto l_read = [&](int kt) {
        const float* buf = lds + (kt & 1) * kTileFloats;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i][kk] = *reinterpret_cast<const f32x4*>(buf + (wm * 64 + i * 32 + frag_row) * LDT + kk * 8 + frag_k);
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[j][kk] = *reinterpret_cast<const f32x4*>(buf + (BM + wn * 64 + j * 32 + frag_row) * LDT + kk * 8 + frag_k);
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    auto mma = [&]() {
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][kk][s], fb[j][kk][s], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
    // memory half-period before the MFMAs of tile t: stage tile t+1 into LDS, start the loads of tile t+2, fetch
    // this wave's fragments of tile t
    auto mem = [&](int t) {
#ifndef ABLATE_MEM                                   // -DABLATE_MEM: fragments of tile 0 re-used for every tile (results wrong)
        if (t + 1 < nkt) l_write(t + 1);
        if (t + 2 < nkt) g_load(t + 2);
        l_read(t);
#elif ABLATE_MEM == 2                                // LDS fragment reads only
        l_read(t & 1);
#elif ABLATE_MEM == 3                                // global loads + LDS writes only
        if (t + 1 < nkt) l_write(t + 1);
        if (t + 2 < nkt) g_load(t + 2);
#else
        if (t == 0) l_read(0);
#endif
    };

    g_load(0);
    l_write(0);
    if (nkt > 1) g_load(1);
    __syncthreads();
    if (group == 0) mem(0);                        // group A enters the loop with its fragments of tile 0
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        if (group == 0) {
            mma();                                 // tile kt
            __syncthreads();
            if (kt + 1 < nkt) mem(kt + 1);
            __syncthreads();
        } else {
            mem(kt);
            __syncthreads();
            mma();                                 // tile kt
            __syncthreads();
        }
    }
    const int half = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                C[(size_t)(m0 + wm * 64 + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * N + n0 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
}

__global__ void gemm_ref(const float* A, const float* W, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(size_t)m * K + k], W[(size_t)n * K + k], s);
    C[(size_t)m * N + n] = s;
}

int main() {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t ldsb = 2 * (size_t)kTileFloats * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pingpong), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    {
        const int M = 512, N = 384, K = 544;
        float *A, *W, *C, *R;
        float* h = (float*)malloc((size_t)(M + N) * K * 4);
        for (size_t i = 0; i < (size_t)(M + N) * K; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
        (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, (size_t)N * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4); (void)hipMalloc(&R, (size_t)M * N * 4);
        (void)hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, h + (size_t)M * K, (size_t)N * K * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(gemm_pingpong, dim3((M / BM) * (N / BN)), dim3(512), ldsb, 0, A, W, C, M, N, K);
        hipLaunchKernelGGL(gemm_ref, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, W, R, M, N, K);
        float* hc = (float*)malloc((size_t)M * N * 4); float* hr = (float*)malloc((size_t)M * N * 4);
        (void)hipMemcpy(hc, C, (size_t)M * N * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hr, R, (size_t)M * N * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (size_t i = 0; i < (size_t)M * N; ++i) { maxerr = fmax(maxerr, fabs((double)hc[i] - hr[i])); maxref = fmax(maxref, fabs((double)hr[i])); }
        printf("check %dx%dx%d: max |err| %.3e (max |ref| %.2f) %s\n", M, N, K, maxerr, maxref, maxerr < 1e-4 * maxref ? "OK" : "MISMATCH");
#ifndef ABLATE_MEM
        if (!(maxerr < 1e-4 * maxref)) return 1;
#endif
    }
    for (int K : {512, 2048}) {
        const int M = 8192, N = 8192;
        float *A, *W, *C; float* h = (float*)malloc((size_t)M * K * 4);
        for (size_t i = 0; i < (size_t)M * K; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
        (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, (size_t)N * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
        (void)hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, h, (size_t)N * K * 4, hipMemcpyHostToDevice);
        float best = 0;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0);
            for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(gemm_pingpong, dim3((M / BM) * (N / BN)), dim3(512), ldsb, 0, A, W, C, M, N, K);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const float tf = 4 * 2.0 * M * N * K / ms / 1e9;
            if (rep && tf > best) best = tf;
        }
        printf("gemm_pingpong 8192x8192x%d: %.1f TFLOP/s\n", K, best);
        (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); free(h);
    }
    return 0;
}
