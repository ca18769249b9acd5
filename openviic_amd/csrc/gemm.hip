// fp32 GEMM on the CDNA4 matrix cores:  C[M,N] = act([A1|A2] . W^T + bias) + R
//
// Every nn.Linear of the captioning path lands here (reference call sites: include/ovc.h,
// ovc_linear).  Activations A are [M,K] and weights W are [N,K] (PyTorch Linear layout), so both
// operands are K-contiguous and share one LDS image: tile[row][k] with rows padded to 36 floats.
//
// Tiling for 64-wide wavefronts / MFMA (not a warp-shaped port):
//   * v_mfma_f32_32x32x2_f32: one wave owns TM x TN accumulator tiles of 32x32 (16 VGPRs each).
//     A operand lane l holds A[row = l&31][k = l>>5], B operand W[col = l&31][k = l>>5].
//   * Each lane reads 4 consecutive k of its row with ONE ds_read_b128 (k = 8*kk + 4*(l>>5) + s)
//     and feeds them to 4 consecutive MFMAs (s = 0..3).  Both operands use the same k
//     permutation, so the sum over k is complete; only the fp32 summation order differs from a
//     sequential dot product.
//   * Row stride 36 floats: 36*r mod 64 hits 16 distinct 4-bank slots for any 16 rows that are
//     distinct mod 16, which is exactly what each ds_read_b128 lane group contains -> no bank
//     conflicts.
//   * 256 threads = 4 waves = one wave per SIMD; LDS double-buffered with one barrier per
//     32-deep K tile; the next tile's global loads are issued before the MFMA block and written
//     to LDS after it (issue-early / write-late).
//   * blockIdx is remapped so that the 8 XCDs each walk a contiguous range of N tiles with all
//     M tiles of a given N tile adjacent: a weight tile is fetched from HBM once per XCD and
//     re-used from that XCD's L2 by the other M tiles.
//   * Segments: N may be split into up to 8 equal segments with their own weight / bias /
//     output pointers (fused q|k|v projections writing straight into the K/V caches).
#include "common.h"

namespace {

constexpr int BK = 32;        // K depth of one LDS tile
constexpr int LDT = BK + 4;   // padded LDS row stride (floats)

template <int BM, int BN, int WM, int WN>
struct TileConfig {
    static constexpr int kThreads = 256;
    static constexpr int kWaveM = BM / WM;         // rows per wave
    static constexpr int kWaveN = BN / WN;         // cols per wave
    static constexpr int TM = kWaveM / 32;
    static constexpr int TN = kWaveN / 32;
    static constexpr int kLoadA = BM * (BK / 4) / kThreads;   // float4 per thread per tile
    static constexpr int kLoadB = BN * (BK / 4) / kThreads;
    static constexpr int kLdsFloats = 2 * (BM + BN) * LDT;
    static_assert(WM * WN == 4, "four waves per workgroup");
    static_assert(kWaveM % 32 == 0 && kWaveN % 32 == 0, "wave tile must be a multiple of 32x32");
    static_assert(kLoadA >= 1 && kLoadB >= 1, "tile too small for 256 loader threads");
};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Bijective "contiguous chunk per XCD" remap (blocks b and b+8 share an XCD).
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void gemm_f32_mfma(GemmArgs p, int tiles_m, int tiles_n_per_seg) {
    using Cfg = TileConfig<BM, BN, WM, WN>;
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tile_n_all = tile / tiles_m;           // M fastest: neighbours share the weight tile
    const int tile_m = tile - tile_n_all * tiles_m;
    const int seg = tile_n_all / tiles_n_per_seg;
    const int tile_n = tile_n_all - seg * tiles_n_per_seg;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const float* __restrict__ W = p.seg[seg].W;
    const int K = p.K1 + p.K2;
    const int nkt = (K + BK - 1) / BK;

    constexpr int kBufFloats = (BM + BN) * LDT;   // one buffer: A tile then B tile

    f32x4 stage_a[Cfg::kLoadA], stage_b[Cfg::kLoadB];

    // Branch-free tile loads: a runtime "load or zero" choice per element makes hipcc branch around
    // every load and wait vmcnt(0) in between (serialised L2 round trips).  Instead the address is
    // clamped in-bounds and the load is unconditional; a K tail (never present in the model shapes) is
    // zeroed when the registers are written to LDS.  Which of A1 / A2 a K tile comes from is
    // wave-uniform because K1 is a multiple of BK whenever K2 > 0.
    const bool k_tail = (K % BK) != 0 || (p.K1 % BK) != 0;   // uniform; false for every real shape
    bool a_ok = true, w_ok = true;
    auto load_tile = [&](int kt) {
        const int kq = tid & 7;
        const int k0 = kt * BK;
        const bool second = k0 >= p.K1;                       // uniform
        const float* __restrict__ Ab = second ? p.A2 : p.A1;
        const int lda = second ? p.lda2 : p.lda1;
        const int klim = second ? p.K2 : p.K1;
        const int ka = (second ? k0 - p.K1 : k0) + kq * 4;
        const int kw = k0 + kq * 4;
        a_ok = ka < klim;
        w_ok = kw < K;
        const int kac = min(ka, klim - 4);
        const int kwc = min(kw, K - 4);
#pragma unroll
        for (int i = 0; i < Cfg::kLoadA; ++i) {
            const int row = (tid >> 3) + i * 32;
            const int gm = min(m0 + row, p.M - 1);
            stage_a[i] = *reinterpret_cast<const f32x4*>(Ab + (size_t)gm * lda + kac);
        }
#pragma unroll
        for (int i = 0; i < Cfg::kLoadB; ++i) {
            const int row = (tid >> 3) + i * 32;
            const int gn = min(n0 + row, p.seg_n - 1);
            stage_b[i] = *reinterpret_cast<const f32x4*>(W + (size_t)gn * K + kwc);
        }
    };
    // The loaded registers are first touched here, after the MFMA block of the previous tile, so the
    // global-load latency hides under the matrix work (issue early / write late).
    auto store_tile = [&](int buf) {
        const int kq = tid & 7;
        if (k_tail) {
#pragma unroll
            for (int i = 0; i < Cfg::kLoadA; ++i)
                if (!a_ok) stage_a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < Cfg::kLoadB; ++i)
                if (!w_ok) stage_b[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < Cfg::kLoadA; ++i) {
            const int row = (tid >> 3) + i * 32;
            *reinterpret_cast<f32x4*>(lds + buf * kBufFloats + row * LDT + kq * 4) = stage_a[i];
        }
#pragma unroll
        for (int i = 0; i < Cfg::kLoadB; ++i) {
            const int row = (tid >> 3) + i * 32;
            *reinterpret_cast<f32x4*>(lds + buf * kBufFloats + (BM + row) * LDT + kq * 4) = stage_b[i];
        }
    };

    f32x16 acc[Cfg::TM][Cfg::TN];
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;

    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tile(kt + 1);

        const float* a_base = lds + buf * kBufFloats + (wm * Cfg::kWaveM + frag_row) * LDT + frag_k;
        const float* b_base = lds + buf * kBufFloats + (BM + wn * Cfg::kWaveN + frag_row) * LDT + frag_k;
#pragma unroll
        for (int kk = 0; kk < BK / 8; ++kk) {
            f32x4 a[Cfg::TM], b[Cfg::TN];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LDT + kk * 8);
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LDT + kk * 8);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[i][j], 0, 0, 0);
        }

        if (kt + 1 < nkt) store_tile(buf ^ 1);
        __syncthreads();
    }

    // Epilogue.  D layout of the 32x32 tile: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    const float* __restrict__ bias = p.seg[seg].bias;
    float* __restrict__ C = p.seg[seg].C;
    const int half = lane >> 5;
    const bool has_res = p.R != nullptr;                          // uniform
    const bool interior = m0 + BM <= p.M && n0 + BN <= p.seg_n;   // uniform: no bounds checks needed
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
        const int n = n0 + wn * Cfg::kWaveN + j * 32 + (lane & 31);
        const int nc = min(n, p.seg_n - 1);
        const float bv = bias ? bias[nc] : 0.f;
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i) {
            const int mbase = m0 + wm * Cfg::kWaveM + i * 32 + 4 * half;
            float out[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv;
                out[r] = p.act == 1 ? fmaxf(v, 0.f) : v;
            }
            if (has_res) {
                float res[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int mc = min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1);
                    res[r] = p.R[(size_t)mc * p.ldr + nc];         // unconditional, clamped in-bounds
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) out[r] += res[r];
            }
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) C[(size_t)(mbase + (r & 3) + 8 * (r >> 2)) * p.ldc + n] = out[r];
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = mbase + (r & 3) + 8 * (r >> 2);
                    if (m < p.M && n < p.seg_n) C[(size_t)m * p.ldc + n] = out[r];
                }
            }
        }
    }
}

template <int BM, int BN, int WM, int WN>
int launch_config(const GemmArgs& a, hipStream_t stream) {
    using Cfg = TileConfig<BM, BN, WM, WN>;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tiles_n = (a.seg_n + BN - 1) / BN;
    const int grid = tiles_m * tiles_n * a.nseg;
    const size_t lds_bytes = sizeof(float) * Cfg::kLdsFloats;
    static bool attr_set = false;   // raise the dynamic-LDS cap once per process (idempotent)
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_mfma<BM, BN, WM, WN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, WM, WN>), dim3(grid), dim3(256), lds_bytes, stream, a, tiles_m, tiles_n);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

// Estimated cost (in units of 32x32x32 MFMA blocks per SIMD) of a tiling on 256 CUs: rounds of
// workgroups times the MFMA work one wave does per workgroup.
template <int BM, int BN>
double tiling_cost(const GemmArgs& a) {
    const long tiles = (long)((a.M + BM - 1) / BM) * ((a.seg_n + BN - 1) / BN) * a.nseg;
    const long resident = 256L * ((BM + BN) * LDT * 8 <= 40 * 1024 ? 2 : 1);
    const long rounds = (tiles + resident - 1) / resident;
    const double per_wg = (double)(BM / 32) * (BN / 32) / 4.0 + 0.35;   // + fixed per-tile overhead
    return rounds * per_wg * ((BM + BN) * LDT * 8 <= 40 * 1024 ? 2.0 : 1.0);
}

}  // namespace

int ovc_gemm_launch(const GemmArgs& a, hipStream_t stream) {
    const int K = a.K1 + a.K2;
    if (a.M <= 0 || a.seg_n <= 0 || a.nseg <= 0 || a.nseg > OVC_MAX_SEGMENTS || K <= 0) return OVC_EINVAL;
    if ((a.K1 & 3) || (a.K2 & 3) || (a.lda1 & 3) || (a.K2 && (a.lda2 & 3))) return OVC_EINVAL;
    if (!ovc_aligned16(a.A1) || (a.K2 && !ovc_aligned16(a.A2))) return OVC_EINVAL;
    if (a.R && a.nseg != 1) return OVC_EINVAL;
    if (a.K2 > 0 && (a.K1 % BK)) return OVC_EINVAL;      // the A1|A2 seam must fall on a K-tile boundary
    for (int s = 0; s < a.nseg; ++s)
        if (!a.seg[s].W || !a.seg[s].C || !ovc_aligned16(a.seg[s].W)) return OVC_EINVAL;

    // A tile may not straddle two segments.
    const bool ok128 = a.nseg == 1 || a.seg_n % 128 == 0;
    const bool ok64 = a.nseg == 1 || a.seg_n % 64 == 0;
    if (!ok64) return OVC_EINVAL;

    double best = 1e300;
    int pick = 0;
    auto consider = [&](int id, double cost, bool allowed) {
        if (allowed && cost < best) { best = cost; pick = id; }
    };
    consider(0, tiling_cost<128, 128>(a), ok128);
    consider(1, tiling_cost<64, 128>(a), ok128);
    consider(2, tiling_cost<128, 64>(a), true);
    consider(3, tiling_cost<64, 64>(a), true);
    switch (pick) {
        case 0: return launch_config<128, 128, 2, 2>(a, stream);
        case 1: return launch_config<64, 128, 2, 2>(a, stream);
        case 2: return launch_config<128, 64, 2, 2>(a, stream);
        default: return launch_config<64, 64, 2, 2>(a, stream);
    }
}

extern "C" int ovc_linear(const float* x, int ldx, const float* x2, int ldx2, int K1, int K2,
                          const float* W, const float* bias, const float* residual, int ldr,
                          float* y, int ldy, int M, int N, int act, ovc_stream stream) {
    if (!x || !W || !y || (K2 > 0 && !x2)) return OVC_EINVAL;
    GemmArgs a{};
    a.A1 = x; a.lda1 = ldx; a.K1 = K1;
    a.A2 = K2 > 0 ? x2 : nullptr; a.lda2 = ldx2; a.K2 = K2 > 0 ? K2 : 0;
    a.M = M; a.seg_n = N; a.nseg = 1; a.ldc = ldy;
    a.R = residual; a.ldr = ldr; a.act = act;
    a.seg[0] = GemmSegment{W, bias, y};
    return ovc_gemm_launch(a, ovc_hip_stream(stream));
}
