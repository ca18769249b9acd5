// What is the floor of the AddNorm LayerNorm launch (1280 rows x 512, two K-slice partials + bias + residual in, one row out)?
// A dependent chain alternating a PRODUCER (writes the two partial tiles, as the K-split GEMM before the LayerNorm does -- the
// LayerNorm then reads lines another XCD has just written) and a LayerNorm variant; hipGraph replay; ns per LayerNorm launch =
// (chain with LayerNorm) - (chain of producers alone).
//   V0  the product's kernel: one wave per row, two-pass moments, __shfl_xor butterflies
//   V1  the same with DPP + v_permlane reductions (common.h: wave_sum_dpp)
//   V2  one pass: sum and sum of squares reduced together (DPP), var = E[x^2] - mean^2
//   V3  no reductions at all (mean = 0, rstd = 1): what the loads + store alone cost
//   V4  V1 with two rows per wave (half a wave per row; 160 workgroups)
//   hipcc -O3 --offload-arch=gfx950 tools/ln_floor_probe.hip -o tools/ln_floor_probe.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../openviic_amd/csrc/common.h"

template <int V>
__global__ __launch_bounds__(256) void ln_kernel(const float* __restrict__ x, long part_stride, const float* __restrict__ bias,
                                                 const float* __restrict__ residual, const float* __restrict__ gamma,
                                                 const float* __restrict__ beta, float eps, float* __restrict__ y, int rows, int d) {
    constexpr int kVecs = 2, kParts = 2;
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = d >> 2;
    float* yrow = y + (size_t)row * d;
    int col[kVecs];
#pragma unroll
    for (int i = 0; i < kVecs; ++i) col[i] = min(lane + i * 64, nvec - 1);
    f32x4 part[kParts][kVecs], bv[kVecs], rv[kVecs], gv[kVecs], bev[kVecs];
#pragma unroll
    for (int s = 0; s < kParts; ++s)
#pragma unroll
        for (int i = 0; i < kVecs; ++i) part[s][i] = reinterpret_cast<const f32x4*>(x + s * part_stride + (size_t)row * d)[col[i]];
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
        bv[i] = reinterpret_cast<const f32x4*>(bias)[col[i]];
        rv[i] = reinterpret_cast<const f32x4*>(residual + (size_t)row * d)[col[i]];
        gv[i] = reinterpret_cast<const f32x4*>(gamma)[col[i]];
        bev[i] = reinterpret_cast<const f32x4*>(beta)[col[i]];
    }
    f32x4 v[kVecs];
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
        v[i] = ((part[0][i] + part[1][i]) + bv[i]) + rv[i];
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
        if (V == 2) sq += (v[i][0] * v[i][0] + v[i][1] * v[i][1]) + (v[i][2] * v[i][2] + v[i][3] * v[i][3]);
    }
    float mean, rstd;
    if (V == 0) {
        mean = wave_sum(sum) / (float)d;
        sq = 0.f;
#pragma unroll
        for (int i = 0; i < kVecs; ++i) { const f32x4 t = v[i] - mean; sq += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]); }
        rstd = 1.0f / sqrtf(wave_sum(sq) / (float)d + eps);
    } else if (V == 1) {
        mean = wave_sum_dpp(sum) / (float)d;
        sq = 0.f;
#pragma unroll
        for (int i = 0; i < kVecs; ++i) { const f32x4 t = v[i] - mean; sq += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]); }
        rstd = 1.0f / sqrtf(wave_sum_dpp(sq) / (float)d + eps);
    } else if (V == 2) {
        const float s1 = wave_sum_dpp(sum), s2 = wave_sum_dpp(sq);     // independent: the two reductions overlap
        mean = s1 / (float)d;
        rstd = 1.0f / sqrtf(fmaxf(s2 / (float)d - mean * mean, 0.f) + eps);
    } else {
        mean = 0.f * sum; rstd = 1.f + eps;
    }
#pragma unroll
    for (int i = 0; i < kVecs; ++i) reinterpret_cast<f32x4*>(yrow)[lane + i * 64] = (v[i] - mean) * rstd * gv[i] + bev[i];
}

// two rows per wave: lanes 0..31 row 2w, lanes 32..63 row 2w + 1; each lane holds 4 float4 of its row
__global__ __launch_bounds__(256) void ln_two_rows(const float* __restrict__ x, long part_stride, const float* __restrict__ bias,
                                                   const float* __restrict__ residual, const float* __restrict__ gamma,
                                                   const float* __restrict__ beta, float eps, float* __restrict__ y, int rows, int d) {
    constexpr int kVecs = 4, kParts = 2;
    const int lane = threadIdx.x & 63, hl = lane & 31;
    const int row = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2 + (lane >> 5);
    if (row >= rows) return;
    float* yrow = y + (size_t)row * d;
    f32x4 part[kParts][kVecs], bv[kVecs], rv[kVecs], gv[kVecs], bev[kVecs];
#pragma unroll
    for (int s = 0; s < kParts; ++s)
#pragma unroll
        for (int i = 0; i < kVecs; ++i) part[s][i] = reinterpret_cast<const f32x4*>(x + s * part_stride + (size_t)row * d)[hl + i * 32];
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
        bv[i] = reinterpret_cast<const f32x4*>(bias)[hl + i * 32];
        rv[i] = reinterpret_cast<const f32x4*>(residual + (size_t)row * d)[hl + i * 32];
        gv[i] = reinterpret_cast<const f32x4*>(gamma)[hl + i * 32];
        bev[i] = reinterpret_cast<const f32x4*>(beta)[hl + i * 32];
    }
    f32x4 v[kVecs];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
        v[i] = ((part[0][i] + part[1][i]) + bv[i]) + rv[i];
        sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = half_wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) { const f32x4 t = v[i] - mean; sq += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]); }
    const float rstd = 1.0f / sqrtf(half_wave_sum(sq) / (float)d + eps);
#pragma unroll
    for (int i = 0; i < kVecs; ++i) reinterpret_cast<f32x4*>(yrow)[hl + i * 32] = (v[i] - mean) * rstd * gv[i] + bev[i];
}

// the K-split GEMM's stand-in: every workgroup writes a 32 x 32 tile of each partial (scattered over the XCDs as the GEMM's are)
__global__ __launch_bounds__(256) void producer(float* __restrict__ parts, long part_stride, int rows, int d, float seed) {
    const int tile = blockIdx.x, tm = tile / (d / 32), tn = tile % (d / 32), s = blockIdx.y;
    const int r = tm * 32 + (threadIdx.x >> 3), c = tn * 32 + (threadIdx.x & 7) * 4;
    if (r < rows) *reinterpret_cast<f32x4*>(parts + s * part_stride + (size_t)r * d + c) = f32x4{seed + r, seed + c, seed, 1.f};
}

int main() {
    const int rows = 1280, d = 512, n = 300;
    hipStream_t s; (void)hipStreamCreate(&s);
    float *parts, *bias, *res, *gamma, *beta, *y;
    const long stride = (long)rows * d;
    (void)hipMalloc(&parts, 2 * stride * 4); (void)hipMalloc(&res, stride * 4); (void)hipMalloc(&y, stride * 4);
    (void)hipMalloc(&bias, d * 4); (void)hipMalloc(&gamma, d * 4); (void)hipMalloc(&beta, d * 4);
    (void)hipMemset(parts, 0, 2 * stride * 4); (void)hipMemset(res, 0, stride * 4); (void)hipMemset(bias, 0, d * 4);
    (void)hipMemset(gamma, 0, d * 4); (void)hipMemset(beta, 0, d * 4);
    auto chain = [&](int variant, int nrows) {
        hipGraph_t g; hipGraphExec_t ge;
        (void)hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
        for (int i = 0; i < n; ++i) {
            // the residual of launch i is the output of launch i - 1 (y and res swap): a true dependent chain
            float* in_res = (i & 1) ? y : res; float* out = (i & 1) ? res : y;
            hipLaunchKernelGGL(producer, dim3(((nrows + 31) / 32) * (d / 32), 2), dim3(256), 0, s, parts, stride, nrows, d, (float)i);
            const dim3 grid((nrows + 3) / 4), block(256);
#define LN(V) hipLaunchKernelGGL(ln_kernel<V>, grid, block, 0, s, parts, stride, bias, in_res, gamma, beta, 1e-5f, out, nrows, d)
            if (variant == 0) LN(0); else if (variant == 1) LN(1); else if (variant == 2) LN(2); else if (variant == 3) LN(3);
            else if (variant == 4) hipLaunchKernelGGL(ln_two_rows, dim3((nrows + 7) / 8), block, 0, s, parts, stride, bias, in_res, gamma, beta, 1e-5f, out, nrows, d);
#undef LN
        }
        (void)hipStreamEndCapture(s, &g); (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        (void)hipGraphLaunch(ge, s); (void)hipStreamSynchronize(s);
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            (void)hipEventRecord(e0, s); (void)hipGraphLaunch(ge, s); (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        return best * 1e6f / n;
    };
    const char* names[] = {"V0 product (shfl butterflies, two passes)", "V1 DPP reductions, two passes", "V2 DPP, one pass (E[x^2] - mean^2)",
                           "V3 no reductions (loads + store only)", "V4 two rows per wave (half-wave DPP)", "producer alone"};
    for (int nrows : {1280, 160}) {
        const float base = chain(5, nrows);
        printf("%d rows: producer alone %.0f ns per launch\n", nrows, base);
        for (int v = 0; v < 5; ++v) printf("  %-44s %.0f ns per LayerNorm launch (chain pair %.0f)\n", names[v], chain(v, nrows) - base, chain(v, nrows));
    }
    return 0;
}
