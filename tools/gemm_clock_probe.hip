// Is the fp32 MFMA GEMM loop held back by the clock (DVFS) or by its own issue stream?
// MI355X_MICROARCH.md, 'DVFS give-back' item 6: in-kernel clock = d(s_memtime) / d(s_memrealtime) x 100 MHz, stamped once
// around the main loop after >= 2 s of back-to-back launches on random data; median over workgroups.
//   mode 0: bare v_mfma_f32_32x32x2_f32 stream (4 accumulators, operands in registers), one wave per SIMD
//   mode 1: the production 128x128x32 loop structure (raw buffer loads -> registers -> one LDS buffer, two barriers per K
//           tile, ds_read_b128 fragments, setprio around the MFMA cluster), M = N = 8192, K = 2048
//   mode 2: the 64x64x32 four-chain structure of the decode products (one 32x32 tile per wave, 4 accumulator sets)
// Prints TFLOP/s (wall), MFMA cycles per SIMD / loop cycles (issue efficiency in shader cycles) and the clock.
// Diagnostic build only: the stamps go to a buffer of their own, no output value depends on them.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Stamp { unsigned long long cycles, real; };

__global__ __launch_bounds__(256) void bare(const float* in, float* out, Stamp* st, int iters) {
    f32x16 a0, a1, a2, a3;
    for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; a2[r] = 0.f; a3[r] = 0.f; }
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[(blockIdx.x * 256 + threadIdx.x) * 8 + j];
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[0], v[1], a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[2], v[3], a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[4], v[5], a2, 0, 0, 0);
        a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(v[6], v[7], a3, 0, 0, 0);
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

// BM x BN tile, 2 x 2 waves, NC accumulator sets per wave (chains), BK = 32, one LDS buffer
template <int BM, int BN, int NC>
__global__ __launch_bounds__(256) void gemm(const float* __restrict__ A, const float* __restrict__ W, float* __restrict__ C,
                                            Stamp* st, int M, int N, int K) {
    constexpr int BK = 32, LDT = 36, TM = BM / 64, TN = BN / 64, LA = BM * 8 / 256, LB = BN * 8 / 256;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int tiles_m = M / BM;
    const int tile_n = blockIdx.x / tiles_m, tile_m = blockIdx.x % tiles_m;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, M * K * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, N * K * 4, 0x00020000);
    f32x4 sa[LA], sb[LB];
    int voa[LA], vob[LB];
    for (int i = 0; i < LA; ++i) voa[i] = ((m0 + (tid >> 3) + i * 32) * K + (tid & 7) * 4) * 4;
    for (int i = 0; i < LB; ++i) vob[i] = ((n0 + (tid >> 3) + i * 32) * K + (tid & 7) * 4) * 4;
    auto load = [&](int kt) {
#pragma unroll
        for (int i = 0; i < LA; ++i) sa[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(ra, voa[i], kt * BK * 4, 0));
#pragma unroll
        for (int i = 0; i < LB; ++i) sb[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rw, vob[i], kt * BK * 4, 0));
    };
    auto store = [&]() {
#pragma unroll
        for (int i = 0; i < LA; ++i) *reinterpret_cast<f32x4*>(lds + ((tid >> 3) + i * 32) * LDT + (tid & 7) * 4) = sa[i];
#pragma unroll
        for (int i = 0; i < LB; ++i) *reinterpret_cast<f32x4*>(lds + (BM + (tid >> 3) + i * 32) * LDT + (tid & 7) * 4) = sb[i];
    };
    f32x16 acc[NC][TM][TN];
    for (int c = 0; c < NC; ++c) for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;
    load(0); store(); __syncthreads();
    const float* a_base = lds + (wm * (BM / 2) + (lane & 31)) * LDT + (lane >> 5) * 4;
    const float* b_base = lds + (BM + wn * (BN / 2) + (lane & 31)) * LDT + (lane >> 5) * 4;
    const int nkt = K / BK;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int kt = 0; kt < nkt; ++kt) {
        if (kt + 1 < nkt) load(kt + 1);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LDT + g * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LDT + g * 8);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[g % NC][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[g % NC][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
        if (kt + 1 < nkt) { __syncthreads(); store(); }
        __syncthreads();
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    for (int c = 1; c < NC; ++c) for (int i = 0; i < TM; ++i) for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[0][i][j][r] += acc[c][i][j][r];
    for (int i = 0; i < TM; ++i)
        for (int j = 0; j < TN; ++j)
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                C[(size_t)row * N + n0 + wn * (BN / 2) + j * 32 + (lane & 31)] = acc[0][i][j][r];
            }
    if (tid == 0) st[blockIdx.x] = Stamp{c1 - c0, r1 - r0};
}

static double median_clock(Stamp* dst, int n, double* cycles) {
    std::vector<Stamp> h(n);
    (void)hipMemcpy(h.data(), dst, n * sizeof(Stamp), hipMemcpyDeviceToHost);
    std::vector<double> clk, cyc;
    for (const Stamp& s : h) if (s.real) { clk.push_back((double)s.cycles / (double)s.real * 0.1); cyc.push_back((double)s.cycles); }
    std::sort(clk.begin(), clk.end()); std::sort(cyc.begin(), cyc.end());
    *cycles = cyc[cyc.size() / 2];
    return clk[clk.size() / 2];      // GHz
}

int main(int argc, char** argv) {
    const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
    // optional shape (multiples of 128 / 128 / 32): gemm_clock_probe.bin <seconds> M N K
    const int M = argc > 4 ? atoi(argv[2]) : 8192, N = argc > 4 ? atoi(argv[3]) : 8192, K = argc > 4 ? atoi(argv[4]) : 2048;
    float *A, *W, *C, *h = (float*)malloc((size_t)M * K * 4);
    Stamp* st;
    // the bare-MFMA mode reads 1024 x 256 x 8 floats of A and writes 1024 x 256 floats of C whatever the shape
    const size_t a_floats = std::max((size_t)M * K, (size_t)1024 * 256 * 8), c_floats = std::max((size_t)M * N, (size_t)1024 * 256);
    (void)hipMalloc(&A, a_floats * 4); (void)hipMalloc(&W, (size_t)N * K * 4); (void)hipMalloc(&C, c_floats * 4);
    (void)hipMemset(A, 0, a_floats * 4);
    (void)hipMalloc(&st, 65536 * sizeof(Stamp));
    for (size_t i = 0; i < (size_t)M * K; ++i) h[i] = (float)rand() / RAND_MAX * 2.f - 1.f;
    (void)hipMemcpy(A, h, (size_t)M * K * 4, hipMemcpyHostToDevice);
    for (size_t off = 0; off < (size_t)N * K; off += (size_t)M * K)
        (void)hipMemcpy(W + off, h, std::min((size_t)M * K, (size_t)N * K - off) * 4, hipMemcpyHostToDevice);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm<128, 128, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
        const int wgs = mode == 0 ? 1024 : (mode == 1 ? (M / 128) * (N / 128) : (M / 64) * (N / 64));
        const int iters = 16384;
        const double flop = mode == 0 ? (double)wgs * 4 * iters * 4 * 4096.0 : 2.0 * M * (double)N * K;
        auto launch = [&]() {
            if (mode == 0) hipLaunchKernelGGL(bare, dim3(wgs), dim3(256), 0, 0, A, C, st, iters);
            else if (mode == 1) hipLaunchKernelGGL((gemm<128, 128, 1>), dim3(wgs), dim3(256), 4 * 256 * 36, 0, A, W, C, st, M, N, K);
            else hipLaunchKernelGGL((gemm<64, 64, 4>), dim3(wgs), dim3(256), 4 * 128 * 36, 0, A, W, C, st, M, N, K);
        };
        launch(); (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float one; (void)hipEventElapsedTime(&one, e0, e1);
        const int reps = std::max(3, (int)(seconds * 1e3 / one));
        (void)hipEventRecord(e0);
        for (int i = 0; i < reps; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double cyc;
        const double ghz = median_clock(st, wgs, &cyc);
        // MFMA issue cycles one wave spends in the stamped loop (64 per v_mfma_f32_32x32x2_f32)
        const double mfma_cycles = mode == 0 ? iters * 4 * 64.0 : (mode == 1 ? (K / 2.0) * 4 * 64.0 : (K / 2.0) * 64.0);
        printf("%dx%dx%d mode %d (%s): %d launches, %.3f ms each, %.1f TFLOP/s wall; in-kernel clock %.3f GHz; loop %.0f cycles, own MFMA %.0f "
               "cycles = %.3f of the loop (x waves sharing the SIMD)\n", M, N, K, mode,
               mode == 0 ? "bare MFMA" : (mode == 1 ? "128x128x32 one chain" : "64x64x32 four chains"), reps, ms / reps,
               flop * reps / ms / 1e9, ghz, cyc, mfma_cycles, mfma_cycles / cyc);
    }
    return 0;
}
