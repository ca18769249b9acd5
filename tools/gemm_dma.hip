// Prototype: 128x128x32 fp32 MFMA GEMM tile loop fed by LDS-DMA (global_load_lds_dwordx4) with a 3-deep LDS ring,
// counted vmcnt and a raw s_barrier, XOR-swizzled LDS image (swizzle applied on the per-lane SOURCE address,
// the LDS destination stays lane-linear).  Compared against the register-staged loop for speed and results.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int kTileFloats = (BM + BN) * BK;          // one ring slot: A image then B image, rows of 32 floats, no padding
#ifndef SLOTS
#define SLOTS 3
#endif
constexpr int kSlots = SLOTS;      // ring depth: SLOTS - 1 tiles in flight; 3 slots = 96 KB (1 workgroup per CU), 2 slots = 64 KB (2 per CU)

__device__ __forceinline__ void glds16(const float* gsrc, float* lds_dst) {
    __builtin_amdgcn_global_load_lds(gsrc, (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

__global__ __launch_bounds__(256) void gemm_dma(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_m = M / BM;
    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x % tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nkt = K / BK;

    // this lane's part of a tile fill: 4 A pieces + 4 B pieces per wave, a piece = 8 rows x 128 B = 1 KiB
    const int prow = lane >> 3, pchunk = lane & 7;
    auto issue = [&](int kt) {
        float* slot = lds + (kt % kSlots) * kTileFloats;
        const int k0 = kt * BK;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = wave * 32 + i * 8 + prow;
            const int chunk = pchunk ^ ((row >> 1) & 7);                     // logical k chunk stored at this physical place
            glds16(A + (size_t)(m0 + row) * K + k0 + 4 * chunk, slot + (wave * 32 + i * 8) * BK);
            glds16(W + (size_t)(n0 + row) * K + k0 + 4 * chunk, slot + BM * BK + (wave * 32 + i * 8) * BK);
        }
    };

    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int s = 0; s < kSlots - 1 && s < nkt; ++s) issue(s);
    const int frow = lane & 31, half = lane >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        if (kSlots > 2 && kt + 1 < nkt) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");   // tile kt landed, tile kt+1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + kSlots - 1 < nkt) issue(kt + kSlots - 1);
        const float* slot = lds + (kt % kSlots) * kTileFloats;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int ra = wm * 64 + i * 32 + frow, rb = wn * 64 + i * 32 + frow;
                a[i] = *reinterpret_cast<const f32x4*>(slot + ra * BK + 4 * ((2 * kk + half) ^ ((ra >> 1) & 7)));
                b[i] = *reinterpret_cast<const f32x4*>(slot + BM * BK + rb * BK + 4 * ((2 * kk + half) ^ ((rb >> 1) & 7)));
            }
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    }
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r)
        C[(size_t)(m0 + wm * 64 + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * N + n0 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
}

// reference: one thread per output element (slow, exact fp32 fma chain in k order)
__global__ void gemm_ref(const float* A, const float* W, float* C, int M, int N, int K) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    float s = 0.f;
    for (int k = 0; k < K; ++k) s = fmaf(A[(size_t)m * K + k], W[(size_t)n * K + k], s);
    C[(size_t)m * N + n] = s;
}

int main() {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t ldsb = (size_t)kSlots * kTileFloats * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_dma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    {   // correctness on a small problem
        const int M = 256, N = 384, K = 512;
        float *A, *W, *C, *R;
        float* h = (float*)malloc((size_t)(M + N) * K * 4);
        for (size_t i = 0; i < (size_t)(M + N) * K; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
        (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, (size_t)N * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4); (void)hipMalloc(&R, (size_t)M * N * 4);
        (void)hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, h + (size_t)M * K, (size_t)N * K * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(gemm_dma, dim3((M / BM) * (N / BN)), dim3(256), ldsb, 0, A, W, C, M, N, K);
        hipLaunchKernelGGL(gemm_ref, dim3((N + 255) / 256, M), dim3(256), 0, 0, A, W, R, M, N, K);
        float* hc = (float*)malloc((size_t)M * N * 4); float* hr = (float*)malloc((size_t)M * N * 4);
        (void)hipMemcpy(hc, C, (size_t)M * N * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hr, R, (size_t)M * N * 4, hipMemcpyDeviceToHost);
        double maxerr = 0, maxref = 0;
        for (size_t i = 0; i < (size_t)M * N; ++i) { maxerr = fmax(maxerr, fabs((double)hc[i] - hr[i])); maxref = fmax(maxref, fabs((double)hr[i])); }
        printf("check %dx%dx%d: max |err| %.3e (max |ref| %.2f) %s\n", M, N, K, maxerr, maxref, maxerr < 1e-4 * maxref ? "OK" : "MISMATCH");
        if (!(maxerr < 1e-4 * maxref)) return 1;
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
            for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(gemm_dma, dim3((M / BM) * (N / BN)), dim3(256), ldsb, 0, A, W, C, M, N, K);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            const float tf = 4 * 2.0 * M * N * K / ms / 1e9;
            if (rep && tf > best) best = tf;
        }
        printf("gemm_dma 8192x8192x%d: %.1f TFLOP/s\n", K, best);
        (void)hipFree(A); (void)hipFree(W); (void)hipFree(C); free(h);
    }
    return 0;
}
