// Where does the 128x128x32 fp32 MFMA GEMM tile loop lose time?  Ablations of the production loop structure
// (openviic_amd/csrc/gemm.hip) on M = N = 8192, K = 512 (4096 workgroups):
//   bit 0: no global loads after the prologue (stage registers reused)
//   bit 1: no LDS writes / barrier after the prologue
//   bit 2: no LDS reads in the MFMA block (fragments stay in registers)
// Results are garbage by construction; only the time matters.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int BM = 128, BN = 128, BK = 32, LDT = 36;

template <int ABL>
__global__ __launch_bounds__(256) void gemm(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C, int M, int N, int K) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_m = M / BM;
    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x % tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    constexpr int kBuf = (BM + BN) * LDT;
    f32x4 sa[4], sb[4];
    // bit 6: raw buffer loads -- per-lane byte offset fixed for the whole K loop, the K position is a scalar offset
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, (int)((size_t)M * K * 4 > 0x7fffffffu ? 0x7fffffff : (size_t)M * K * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, (int)((size_t)N * K * 4 > 0x7fffffffu ? 0x7fffffff : (size_t)N * K * 4), 0x00020000);
    int voa[4], vob[4];
    for (int i = 0; i < 4; ++i) {
        const int row = (tid >> 3) + i * 32;
        voa[i] = ((m0 + row) * K + (tid & 7) * 4) * 4;
        vob[i] = ((n0 + row) * K + (tid & 7) * 4) * 4;
    }
    auto load = [&](int kt) {
        if (ABL & 64) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa[i], kt * BK * 4, 0));
                sb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, vob[i], kt * BK * 4, 0));
            }
            return;
        }
        const int kq = tid & 7, k = kt * BK + kq * 4;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + i * 32;
            sa[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(m0 + row) * K + k);
            sb[i] = *reinterpret_cast<const f32x4*>(W + (size_t)(n0 + row) * K + k);
        }
    };
    auto store = [&](int buf) {
        const int kq = tid & 7;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = (tid >> 3) + i * 32;
            *reinterpret_cast<f32x4*>(lds + buf * kBuf + row * LDT + kq * 4) = sa[i];
            *reinterpret_cast<f32x4*>(lds + buf * kBuf + (BM + row) * LDT + kq * 4) = sb[i];
        }
    };
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if (ABL & 8) {      // stagger the co-resident workgroup by half a K tile of MFMA time
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        if (hwid & 1) __builtin_amdgcn_s_sleep(32);
    }
    load(0); store(0); __syncthreads();
    const int nkt = K / BK;
    const int frow = lane & 31, fk = (lane >> 5) * 4;
    f32x4 a[2], b[2];
    a[0] = sa[0]; a[1] = sa[1]; b[0] = sb[0]; b[1] = sb[1];
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = (ABL & 2) ? 0 : (kt & 1);
        if (!(ABL & 1) && kt + 1 < nkt) load(kt + 1);
        const float* ab = lds + buf * kBuf + (wm * 64 + frow) * LDT + fk;
        const float* bb = lds + buf * kBuf + (BM + wn * 64 + frow) * LDT + fk;
        if (ABL & 32) {
            // fragment reads of k-group kk+1 are issued before the MFMA cluster of k-group kk (register ping-pong)
            f32x4 fa[2][2], fb[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { fa[0][i] = *reinterpret_cast<const f32x4*>(ab + i * 32 * LDT); fb[0][i] = *reinterpret_cast<const f32x4*>(bb + i * 32 * LDT); }
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                if (kk + 1 < 4) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        fa[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(ab + i * 32 * LDT + (kk + 1) * 8);
                        fb[(kk + 1) & 1][i] = *reinterpret_cast<const f32x4*>(bb + i * 32 * LDT + (kk + 1) * 8);
                    }
                }
                if (ABL & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk & 1][i][s], fb[kk & 1][j][s], acc[i][j], 0, 0, 0);
                if (ABL & 16) __builtin_amdgcn_s_setprio(0);
            }
        } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            if (!(ABL & 4)) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { a[i] = *reinterpret_cast<const f32x4*>(ab + i * 32 * LDT + kk * 8); b[i] = *reinterpret_cast<const f32x4*>(bb + i * 32 * LDT + kk * 8); }
            }
            if (ABL & 16) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
            if (ABL & 16) __builtin_amdgcn_s_setprio(0);
        }
        }
        if (!(ABL & 2)) {
            if (kt + 1 < nkt) store(buf ^ 1);
            __syncthreads();
        }
    }
    const int half = lane >> 5;
    for (int j = 0; j < 2; ++j) for (int i = 0; i < 2; ++i) for (int r = 0; r < 16; ++r)
        C[(size_t)(m0 + wm * 64 + i * 32 + 4 * half + (r & 3) + 8 * (r >> 2)) * N + n0 + wn * 64 + j * 32 + (lane & 31)] = acc[i][j][r];
}
template <int ABL> void run(const float* A, const float* W, float* C, int M, int N, int K) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const size_t ldsb = 2 * (BM + BN) * LDT * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm<ABL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb);
    float best = 0;
    for (int rep = 0; rep < 4; ++rep) {
        (void)hipEventRecord(e0);
        for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(gemm<ABL>, dim3((M / BM) * (N / BN)), dim3(256), ldsb, 0, A, W, C, M, N, K);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const float tf = 4 * 2.0 * M * N * K / ms / 1e9;
        if (rep && tf > best) best = tf;
    }
    printf("ablation %d (%s%s%s%s%s): %.1f TFLOP/s\n", ABL, ABL & 1 ? "no-gload " : "", ABL & 2 ? "no-ldswrite/barrier " : "", ABL & 4 ? "no-ldsread " : "", ABL & 8 ? "stagger " : "", ABL & 16 ? "setprio " : "", best); if (ABL & 32) printf("   (with fragment ping-pong)\n"); if (ABL & 64) printf("   (buffer loads, scalar k offset)\n");
}
int main() {
    const int M = 8192, N = 8192, K = 512;
    float *A, *W, *C, *h = (float*)malloc((size_t)M * K * 4);
    (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, (size_t)N * K * 4); (void)hipMalloc(&C, (size_t)M * N * 4);
    for (size_t i = 0; i < (size_t)M * K; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice); (void)hipMemcpy(W, h, (size_t)N * K * 4, hipMemcpyHostToDevice);
    run<16>(A, W, C, M, N, K); run<80>(A, W, C, M, N, K); run<16>(A, W, C, M, N, K); run<80>(A, W, C, M, N, K); run<16>(A, W, C, M, N, K); run<80>(A, W, C, M, N, K);
    return 0;
}
