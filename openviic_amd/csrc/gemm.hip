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
//   * 256 threads = 4 waves = one wave per SIMD; one LDS tile (OVC_F32_BUFS); the next tile's global loads are
//     issued before the MFMA block and written to LDS after it, behind a barrier (issue-early / write-late).
//   * blockIdx is remapped so that each of the 8 XCDs (private L2s) works on a contiguous part of the tile
//     grid: super-rows of 8 M tiles swept over N when the A panel exceeds L2, otherwise a 2-D split
//     pm x 8/pm chosen on the host to minimise the operand bytes the XCDs pull from the Infinity Cache.
//   * Segments: N may be split into up to 8 equal segments with their own weight / bias /
//     output pointers (fused q|k|v projections writing straight into the K/V caches).
//   * K split across workgroups (gridDim.y slices writing raw partial products that the consuming
//     LayerNorm sums in order) for the M = B*k products back to d_model, whose 32x32 tiles are bound by
//     the CU's L2 port rather than by the matrix cores.
//   * K-order classes (GemmArgs::kchains).  fp32 addition is not associative, and beam search takes
//     decisions on fp32 comparisons, so the ORDER in which a product sums over K must not depend on which
//     tiling a timing picked, on the batch size or on the box.  Every tiling belongs to one of two
//     classes, and all tilings of a class produce bit-identical results:
//       kchains = 1  one fmaf chain over k in the order of the 8-deep groups (k = 8g + {0,4,1,5,2,6,3,7});
//       kchains = 4  four chains, chain c = the 8-deep groups with g mod 4 == c, summed ((c0+c1)+c2)+c3.
//                    A workgroup may give the chains to 4, 2 or 1 waves of the same output tile (WK = 4 / 2 /
//                    1 with NC = 1 / 2 / 4 accumulator sets per wave): small-M products get 4x / 2x the
//                    wave-level parallelism of one chain without changing a single bit of the result.
//     The class is chosen by the CALL SITE (engine: 1 for the M = B*N encoder-side products, 4 for the
//     M = B*k decode-step products), never by the tuner; ovc_gemm_tune only ranks tilings inside it.
//   * Seventeen tilings (128x128 ... 32x32; K tile 16, 32 or 64); ovc_gemm_tune measures the ones of the requested
//     class per shape, a cost model covers shapes that were never measured.
#include <atomic>
#include <type_traits>
#include <mutex>
#include <vector>

#include <hip/hip_ext.h>

#include "common.h"

namespace {

// K depth of one LDS tile is a template parameter BK (32 or 64); rows are padded to BK + 4 floats:
// 36 r mod 64 and 68 r mod 64 both hit 16 distinct 4-bank slots for 16 rows distinct mod 16.

// WM x WN x WK waves: the workgroup tile is split WM x WN over the output and, for small tiles, WK ways
// over the K chains *inside* the workgroup (wave wk owns chains wk, wk + WK, ...: NC accumulator sets; the
// chains are summed in chain order through LDS at the end).  With M = B*k = 1280 decode rows an output
// of 1280 x 512 is only 640 MFMA tiles for 1024 SIMDs; giving the four chains of a 32x32 tile to four waves
// turns that into 2560 wave-sized tasks without any cross-workgroup reduction.
// NC * WK = number of chains of the K-order class: 1 (WK = 1, NC = 1) or 4.
// LDS tile buffers.  1 (default): one buffer -- the next K tile waits in registers and is written after a second barrier.
// Half the LDS per workgroup lets twice as many workgroups share a CU, which is worth more than the barrier: same box,
// alternating runs (tools/gemm_bench.py), best tiling per shape, 2 buffers -> 1: feature projection 247 -> 225 us, encoder
// o-proj 66.0 -> 60.4, encoder FFN-1 227 -> 211, vocabulary 123 -> 117, decode o-proj 12.0 -> 11.2, q|k|v 23.6 -> 22.7, FFN 28.7 ->
// 27.7; single-stream batch 15.26 -> 14.95 ms, four streams +-1 %.  2: double-buffered, one barrier per K tile (round 1).
#ifndef OVC_F32_BUFS
#define OVC_F32_BUFS 1
#endif
template <int BM, int BN, int WM, int WN, int WK, int BK, int NC>
struct TileConfig {
    static constexpr int LDT = BK + 4;             // padded LDS row stride (floats)
    static constexpr int kThreads = 256;
    static constexpr int kWaveM = BM / WM;         // rows per wave
    static constexpr int kWaveN = BN / WN;         // cols per wave
    static constexpr int TM = kWaveM / 32;
    static constexpr int TN = kWaveN / 32;
    static constexpr int kLoadA = BM * (BK / 4) / kThreads;   // float4 per thread per tile
    static constexpr int kLoadB = BN * (BK / 4) / kThreads;
    static constexpr int kChains = NC * WK;                   // K-order class of this instance
    static constexpr int kBufs = OVC_F32_BUFS;                // LDS tile buffers (2: one barrier per K tile)
    static constexpr int kTileFloats = kBufs * (BM + BN) * LDT;
    static constexpr int kRedFloats = (WK - 1) * NC * BM * BN;    // chain reduction scratch (re-uses the tile buffers)
    static constexpr int kLdsFloats = kTileFloats > kRedFloats ? kTileFloats : kRedFloats;
    static_assert(WM * WN * WK == 4, "four waves per workgroup");
    static_assert(kChains == 1 || kChains == 4, "K-order classes: one chain or four");
    static_assert(BK % 8 == 0 && (kChains == 1 || BK % 32 == 0), "a K tile holds whole 8-deep groups (four chains: whole 4-group periods)");
    static_assert(kWaveM % 32 == 0 && kWaveN % 32 == 0, "wave tile must be a multiple of 32x32");
    static_assert(kLoadA >= 1 && kLoadB >= 1, "tile too small for 256 loader threads");

};

__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    // Bijective "contiguous chunk per XCD" remap (blocks b and b+8 share an XCD).
    const int xcd = bid & 7, q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// Which output tile a workgroup owns.  Grouped order inside each XCD's contiguous chunk: super-rows of `group_m` M tiles are
// swept over all N tiles, M fastest inside the group (the group's A tiles stay in that XCD's L2 for the whole sweep and each
// weight tile is fetched once per super-row; group_m == tiles_m when all of A fits in L2 anyway) -- or, xcd_pm > 0, a 2-D
// split of the tile grid over the 8 XCDs (xcd_pm x 8/xcd_pm sub-rectangles, M fastest inside): XCD L2s are private, so with a
// split along N only every XCD pulls all of A from the Infinity Cache (8 x 2.6 MB for the 1280-row decode products against
// 3.6 MB of operands); the host picks the split with the least total operand traffic (tile_order).  Needs
// tiles_m % xcd_pm == 0 and tiles_n_all % (8 / xcd_pm) == 0.
__device__ __forceinline__ void tile_coords(int tiles_m, int group_m, int xcd_pm, int& tile_m, int& tile_n_all) {
    const int nwg = gridDim.x;
    const int tile = xcd_remap(blockIdx.x, nwg);
    const int tiles_n_all = nwg / tiles_m;
    if (xcd_pm > 0) {
        const int per_chunk = nwg >> 3, chunk = tile / per_chunk, j = tile - chunk * per_chunk;
        const int sub_m = tiles_m / xcd_pm, cm = chunk % xcd_pm, cn = chunk / xcd_pm;
        const int sub_n = tiles_n_all / (8 / xcd_pm);
        tile_m = cm * sub_m + j % sub_m;
        tile_n_all = cn * sub_n + j / sub_m;
    } else {
        const int group_size = group_m * tiles_n_all;
        const int group = tile / group_size;
        const int first_m = group * group_m;
        const int gm = min(group_m, tiles_m - first_m);
        const int in_group = tile - group * group_size;
        tile_n_all = in_group / gm;
        tile_m = first_m + (in_group - tile_n_all * gm);
    }
}

// The same map with every quotient precomputed on the host (round 4).  The kernel above spends 11 integer divisions (20
// instructions and a dependent v_rcp each) and five DEPENDENT rounds of scalar loads -- kernel arguments, the dispatch packet
// for gridDim, the dynamically indexed segment table -- before its first operand load is issued: 415 instructions and ~2 us of
// an M = 1280 product's ~11.  Here the launch hands over a TileMap (divisors with their multiply-high magics, gridDim, the K
// slice length) as the FIRST kernel argument, and the kernel fetches it together with every GemmArgs field the first tile's loads
// need in ONE scalar round trip.  magic = floor(2^32 / d) + 1: mulhi(x, magic) is x / d or x / d + 1 for every x < 2^32 (exact
// while x * d < 2^32, which a grid of 65 537 tiles in one group already violates), so fast_div corrects it once: exact for any grid.
struct TileMap {
    int nwg, tiles_m, tiles_n_all, tiles_n_per_seg;
    int xcd_pm, pm_shift, sub_m, sub_n;
    unsigned sub_m_magic, seg_magic;
    int group_m, group_size, gm_last;
    unsigned group_size_magic, group_m_magic, gm_last_magic;
    int kslice;
};

__device__ __forceinline__ int fast_div(int x, unsigned magic, int d) {
    if (d == 1) return x;
    const unsigned q = __umulhi((unsigned)x, magic);             // never low, at most one high
    return (int)(q * (unsigned)d > (unsigned)x ? q - 1 : q);
}

__device__ __forceinline__ void tile_coords_fast(const TileMap& t, int bid, int& tile_m, int& tile_n_all) {
    const int xcd = bid & 7, q = t.nwg >> 3, r = t.nwg & 7;
    if (t.xcd_pm > 0) {
        // 2-D split: nwg is a multiple of 8 here (tile_order), so the XCD's chunk is xcd itself and j = bid >> 3
        const int j = bid >> 3;
        const int cm = xcd & (t.xcd_pm - 1), cn = xcd >> t.pm_shift;
        const int jq = fast_div(j, t.sub_m_magic, t.sub_m);
        tile_m = cm * t.sub_m + (j - jq * t.sub_m);
        tile_n_all = cn * t.sub_n + jq;
    } else {
        const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
        const int tile = base + (bid >> 3);
        const int group = fast_div(tile, t.group_size_magic, t.group_size);
        const int first_m = group * t.group_m;
        const int in_group = tile - group * t.group_size;
        const bool last = first_m + t.group_m > t.tiles_m;              // the partial last super-row
        const int gm = last ? t.gm_last : t.group_m;
        tile_n_all = last ? fast_div(in_group, t.gm_last_magic, t.gm_last) : fast_div(in_group, t.group_m_magic, t.group_m);
        tile_m = first_m + (in_group - tile_n_all * gm);
    }
}

// Epilogue of a wave's TM x TN accumulator tiles: + bias, activation, + residual, store (or, K-split, the raw partial product
// of slice blockIdx.y).  D layout of a 32x32 tile: col = lane & 31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).  Branch-free on buffer
// descriptors: one per-lane byte offset (first row of the lane's 16, its column) and a scalar row offset per accumulator
// register; rows past M fall outside the descriptor and are dropped by the hardware, columns past seg_n get an
// out-of-range offset.  (m0, n0) = first row / column of the wave's tiles.
template <int TM, int TN>
__device__ __forceinline__ void store_wave_tiles(const GemmArgs& p, const float* __restrict__ bias, float* __restrict__ Cseg,
                                                 const f32x16 (&acc)[TM][TN], int m0, int n0, int lane) {
    float* __restrict__ C = Cseg + (size_t)blockIdx.y * p.part_stride;
    const int half = lane >> 5;
    const bool has_res = p.R != nullptr;                          // uniform
    const __amdgpu_buffer_rsrc_t rsrc_c = __builtin_amdgcn_make_buffer_rsrc(C, 0, p.M * p.ldc * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(has_res ? p.R : p.A1), 0,
                                                                             has_res && p.res_mod == 0 ? p.M * p.ldr * 4 : 0, 0x00020000);
    constexpr int kOutOfRange = 0x7ffffff0;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + j * 32 + (lane & 31);
        const bool n_ok = n < p.seg_n;
        const float bv = bias ? bias[min(n, p.seg_n - 1)] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mbase = m0 + i * 32 + 4 * half;
            float out[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][j][r] + bv;
                out[r] = p.act == 1 ? fmaxf(v, 0.f) : v;
            }
            if (has_res) {
                float res[16];
                if (p.res_mod == 0) {
                    const int voff_r = n_ok ? (mbase * p.ldr + n) * 4 : kOutOfRange;
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        res[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_r, voff_r, ((r & 3) + 8 * (r >> 2)) * p.ldr * 4, 0));
                } else {                                           // residual broadcast over stacked row blocks
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int mc = min(mbase + (r & 3) + 8 * (r >> 2), p.M - 1) % p.res_mod;
                        res[r] = p.R[(size_t)mc * p.ldr + min(n, p.seg_n - 1)];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) out[r] += res[r];
            }
            const int voff_c = n_ok ? (mbase * p.ldc + n) * 4 : kOutOfRange;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, out[r]), rsrc_c, voff_c, ((r & 3) + 8 * (r >> 2)) * p.ldc * 4, 0);
            if (p.stats_t) {
                // Transposed vocabulary product: this lane's 16 registers are 16 words (rows) of ONE beam row (column n), and the
                // lane 32 further on holds the block's other 16.  Block maximum and sum exp(y - maximum) are in-register
                // reductions in register order plus one half-wave exchange each -- ~70 vector instructions per tile where the
                // row-major form (below) needs ~370.  The 32-row blocks are global (m0 is a multiple of 32) and the order is
                // fixed: every tiling leaves the same bits.
                const int mrow = m0 + i * 32 + 4 * half;
                float v[16];
                float bm = -INFINITY;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    v[r] = mrow + (r & 3) + 8 * (r >> 2) < p.M ? out[r] : -INFINITY;
                    bm = fmaxf(bm, v[r]);
                }
                float x, y;
                ovc_swap_rows<true>(bm, x, y);
                bm = fmaxf(x, y);
                float bs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) bs += __expf(v[r] - bm);
                ovc_swap_rows<true>(bs, x, y);
                bs = x + y;
                if (lane < 32 && n_ok && m0 + i * 32 < p.M)
                    *reinterpret_cast<f32x2*>(p.stats_t + 2 * ((size_t)n * p.stats_ld + ((m0 + i * 32) >> 5))) = f32x2{bm, bs};
            }
            if (p.stats) {
                // Log-softmax pieces of this 32 x 32 tile (vocabulary projection): a register holds one output row per lane
                // half with the 32 columns on the lanes, so the row's block maximum and sum exp(y - maximum) are two
                // half-wave reductions per register.  Blocks are the global 32-column blocks (n0 is a multiple of 32) and
                // the reduction order is fixed, so every tiling of the class leaves the same bits.  Lane l < 16 of each half
                // then keeps register l's pair and ONE store instruction writes the tile's 32 (row, block) entries of the
                // row-major table (8 bytes each: the reader, one workgroup per image, then streams whole rows).
                float bm = 0.f, bs = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = n_ok ? out[r] : -INFINITY;
                    const float m = half_wave_max(v);
                    const float sum = half_wave_sum(__expf(v - m));
                    if ((lane & 15) == r) { bm = m; bs = sum; }
                }
                const int rr = lane & 15, row = mbase + (rr & 3) + 8 * (rr >> 2);
                if ((lane & 31) < 16 && row < p.M && n0 + j * 32 < p.seg_n)
                    *reinterpret_cast<f32x2*>(p.stats + 2 * ((size_t)row * p.stats_ld + ((n0 + j * 32) >> 5))) = f32x2{bm, bs};
            }
        }
    }
}

#include "gemm_split.h"   // gemm_split_mfma: the opt-in split-precision classes (bf16 planes on the 16-bit matrix path)
#include "gemm_rows16.h"  // gemm_rows16_f32: the four-chain class for products of up to 112 rows (16-row tiles, v_mfma_f32_16x16x4_f32)

// Registers are capped where residency matters: the M = 1280 decode products come as 1280 workgroups of 32x64 tiles,
// five per CU -- with more than 96 registers only four are resident and the fifth runs as a second round (35 vs 28 us).
template <int BM, int BN, int BK, int NC>
constexpr int min_waves_per_simd() { return BM * BN <= 32 * 64 ? 5 : (BM * BN <= 64 * 64 ? (NC == 4 && BK == 64 ? 3 : 4) : 1); }

template <int BM, int BN, int WM, int WN, int WK, int BK, int NC>
__global__ __launch_bounds__(256, (min_waves_per_simd<BM, BN, BK, NC>())) void gemm_f32_mfma(TileMap tmap, GemmArgs p) {
    using Cfg = TileConfig<BM, BN, WM, WN, WK, BK, NC>;
    constexpr int LDT = Cfg::LDT;
    constexpr int kVecPerRow = BK / 4;               // float4 per tile row
    constexpr int kRowsPerPass = 256 / kVecPerRow;   // tile rows covered by one pass of the 256 loader threads
    extern __shared__ __attribute__((aligned(16))) float lds[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wk = wave / (WM * WN);
    const int wm = (wave / WN) % WM, wn = wave % WN;

    // ---- ONE scalar round trip: the tile map and every argument the first tile's loads (and the epilogue's addresses) depend on.
    //      The empty asm pins them in SGPRs here, so hipcc issues all the s_loads back to back in front of it instead of
    //      sinking each next to its first use behind the previous one's wait (five dependent rounds before round 4). ----
    const TileMap t = tmap;
    const float* const a1_ptr = p.A1;
    const float* const w0_ptr = p.seg[0].W;
    const float* const bias0_ptr = p.seg[0].bias;
    float* const c0_ptr = p.seg[0].C;
    const float* const a2_shared = p.A2;
    uint8_t* const zero_rows_ptr = p.zero_rows_out;
    const int arg_lda1 = p.lda1, arg_lda2 = p.lda2, arg_K1 = p.K1, arg_K2 = p.K2, arg_M = p.M, arg_seg_n = p.seg_n, arg_nseg = p.nseg;
    // (one statement: one s_waitcnt; "what the epilogue addresses with" rides along -- otherwise one more round trip behind the K loop)
    asm volatile("" ::"s"(t.nwg), "s"(t.tiles_m), "s"(t.tiles_n_all), "s"(t.tiles_n_per_seg), "s"(t.xcd_pm), "s"(t.pm_shift), "s"(t.sub_m),
                 "s"(t.sub_n), "s"(t.sub_m_magic), "s"(t.seg_magic), "s"(t.group_m), "s"(t.group_size), "s"(t.gm_last),
                 "s"(t.group_size_magic), "s"(t.group_m_magic), "s"(t.gm_last_magic), "s"(t.kslice), "s"(a1_ptr), "s"(w0_ptr),
                 "s"(bias0_ptr), "s"(c0_ptr), "s"(arg_lda1), "s"(arg_K1), "s"(arg_K2), "s"(arg_M), "s"(arg_seg_n), "s"(arg_nseg),
                 "s"(a2_shared), "s"(zero_rows_ptr), "s"(arg_lda2), "s"(p.R), "s"(p.ldc), "s"(p.ldr), "s"(p.res_mod), "s"(p.act),
                 "s"(p.part_stride), "s"(p.stats), "s"(p.stats_t), "s"(p.stats_ld));

    int tile_m, tile_n_all;
    tile_coords_fast(t, (int)blockIdx.x, tile_m, tile_n_all);
    const int seg = arg_nseg == 1 ? 0 : fast_div(tile_n_all, t.seg_magic, t.tiles_n_per_seg);
    const int tile_n = tile_n_all - seg * t.tiles_n_per_seg;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    // single-segment products (every decode-step product but q|k|v): the segment's pointers came with the first round trip.
    // (Fetching the first THREE segments' pointers that way as well -- q|k|v -- measured slower: 18.4k against 18.6k captions/s
    // on one stream, 24.52k against 24.63k on four; the longer first batch delays everything else.)
    const float* __restrict__ W = arg_nseg == 1 ? w0_ptr : p.seg[seg].W;
    const float* __restrict__ seg_bias = arg_nseg == 1 ? bias0_ptr : p.seg[seg].bias;
    float* __restrict__ seg_C = arg_nseg == 1 ? c0_ptr : p.seg[seg].C;
    const int K = arg_K1 + arg_K2;
    // Cross-workgroup K split (gridDim.y slices): this workgroup covers [kbase, kbase + K / gridDim.y) and writes a
    // raw partial tile.  It halves / quarters the operand bytes a CU pulls through its L2 port for the M = 1280
    // decode shapes, whose 32x32 tiles are bound by that port rather than by the matrix cores.
    const int kslice = t.kslice;
    const int kbase = (int)blockIdx.y * kslice;
    const int nkt = (kslice + BK - 1) / BK;

    constexpr int kBufFloats = (BM + BN) * LDT;   // one buffer: A tile then B tile

    f32x4 stage_a[Cfg::kLoadA], stage_b[Cfg::kLoadB];

    // Branch-free tile loads: a runtime "load or zero" choice per element makes hipcc branch around
    // every load and wait vmcnt(0) in between (serialised L2 round trips).  Instead the address is
    // clamped in-bounds and the load is unconditional; a K tail (never present in the model shapes) is
    // zeroed when the registers are written to LDS.  Which of A1 / A2 a K tile comes from is
    // wave-uniform because K1 is a multiple of BK whenever K2 > 0.
    const bool k_tail = (K % BK) != 0 || (p.K1 % BK) != 0;   // uniform; false for every real shape
    bool a_ok = true, w_ok = true;

    // Tile loads are raw buffer loads: the per-lane byte offset (row * ld + 4-float column group) is fixed for
    // the whole K loop and the K position travels in the instruction's scalar offset, so the loop spends no
    // vector instructions on addresses (+4..5 % on the 128x128 loop, tools/gemm_ablation.hip).  Rows past M / N
    // fall outside the descriptor's range and read as zero; a K tail is zeroed when the registers go to LDS.
    const __amdgpu_buffer_rsrc_t rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a1_ptr), 0, arg_M * arg_lda1 * 4, 0x00020000);
    // per-segment second block (meshed level gates); the segment table is only consulted when there IS a second block
    const float* a2 = arg_K2 ? (p.seg[seg].A2 ? p.seg[seg].A2 : a2_shared) : a1_ptr;
    const __amdgpu_buffer_rsrc_t rsrc_a2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a2), 0,
                                                                              arg_K2 ? arg_M * arg_lda2 * 4 : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, arg_seg_n * K * 4, 0x00020000);
    int off_a1[Cfg::kLoadA], off_a2[Cfg::kLoadA], off_w[Cfg::kLoadB];
    {
        const int kq = tid % kVecPerRow;
#pragma unroll
        for (int i = 0; i < Cfg::kLoadA; ++i) {
            const int row = m0 + tid / kVecPerRow + i * kRowsPerPass;
            off_a1[i] = (row * arg_lda1 + kq * 4) * 4;
            off_a2[i] = (row * arg_lda2 + kq * 4) * 4;
        }
#pragma unroll
        for (int i = 0; i < Cfg::kLoadB; ++i) off_w[i] = ((n0 + tid / kVecPerRow + i * kRowsPerPass) * K + kq * 4) * 4;
    }
    auto load_tile = [&](int kt) {
        const int k0 = kbase + kt * BK;
        const bool second = k0 >= p.K1;                       // uniform: which of A1 | A2 this K tile comes from
        if (k_tail) {
            const int kq = tid % kVecPerRow;
            a_ok = (second ? k0 - p.K1 : k0) + kq * 4 < (second ? p.K2 : p.K1);
            w_ok = k0 + kq * 4 < K;
        }
        if (!second) {
#pragma unroll
            for (int i = 0; i < Cfg::kLoadA; ++i)
                stage_a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a1, off_a1[i], k0 * 4, 0));
        } else {
#pragma unroll
            for (int i = 0; i < Cfg::kLoadA; ++i)
                stage_a[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a2, off_a2[i], (k0 - p.K1) * 4, 0));
        }
#pragma unroll
        for (int i = 0; i < Cfg::kLoadB; ++i)
            stage_b[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off_w[i], k0 * 4, 0));
    };
    // The loaded registers are first touched here, after the MFMA block of the previous tile, so the
    // global-load latency hides under the matrix work (issue early / write late).
    // K1 fold (GemmArgs::zero_rows_out): the workgroups of the first column tile add up the A rows they stage anyway
    const bool row_sums = zero_rows_ptr != nullptr && tile_n_all == 0;        // uniform
    float rs[Cfg::kLoadA];
#pragma unroll
    for (int i = 0; i < Cfg::kLoadA; ++i) rs[i] = 0.f;
    auto store_tile = [&](int buf) {
        const int kq = tid % kVecPerRow;
        if (k_tail) {
#pragma unroll
            for (int i = 0; i < Cfg::kLoadA; ++i)
                if (!a_ok) stage_a[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < Cfg::kLoadB; ++i)
                if (!w_ok) stage_b[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (row_sums) {
#pragma unroll
            for (int i = 0; i < Cfg::kLoadA; ++i) rs[i] += (stage_a[i][0] + stage_a[i][1]) + (stage_a[i][2] + stage_a[i][3]);
        }
#pragma unroll
        for (int i = 0; i < Cfg::kLoadA; ++i) {
            const int row = tid / kVecPerRow + i * kRowsPerPass;
            *reinterpret_cast<f32x4*>(lds + buf * kBufFloats + row * LDT + kq * 4) = stage_a[i];
        }
#pragma unroll
        for (int i = 0; i < Cfg::kLoadB; ++i) {
            const int row = tid / kVecPerRow + i * kRowsPerPass;
            *reinterpret_cast<f32x4*>(lds + buf * kBufFloats + (BM + row) * LDT + kq * 4) = stage_b[i];
        }
    };

    f32x16 acc[NC][Cfg::TM][Cfg::TN];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][i][j][r] = 0.f;

    const int frag_row = lane & 31;
    const int frag_k = (lane >> 5) * 4;

    // the MFMA block of one K tile (LDS buffer `buf`)
    auto compute_tile = [&](int buf) {
        const float* a_base = lds + buf * kBufFloats + (wm * Cfg::kWaveM + frag_row) * LDT + frag_k;
        const float* b_base = lds + buf * kBufFloats + (BM + wn * Cfg::kWaveN + frag_row) * LDT + frag_k;
        // 8-deep k-groups of this K tile, in order.  One chain: every group goes to accumulator set 0.  Four chains:
        // group kk belongs to chain kk & 3 (K tiles and K slices start on multiples of 32, so this is the global
        // group index mod 4); this wave owns chains wk + WK * c, kept in set c.
        constexpr int kGroupsPerWave = Cfg::kChains == 1 ? BK / 8 : (BK / 32) * NC;
#pragma unroll
        for (int g = 0; g < kGroupsPerWave; ++g) {
            const int set = Cfg::kChains == 1 ? 0 : g % NC;
            const int kk = Cfg::kChains == 1 ? g : 4 * (g / NC) + wk + WK * (g % NC);
            f32x4 a[Cfg::TM], b[Cfg::TN];
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
                a[i] = *reinterpret_cast<const f32x4*>(a_base + i * 32 * LDT + kk * 8);
#pragma unroll
            for (int j = 0; j < Cfg::TN; ++j)
                b[j] = *reinterpret_cast<const f32x4*>(b_base + j * 32 * LDT + kk * 8);
            // Raised priority around the MFMA cluster: hipcc then keeps the cluster contiguous instead of
            // threading the next tile's loads / address arithmetic through it (+8..20 % on the 128x128 loop,
            // tools/gemm_ablation.hip).
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                    for (int j = 0; j < Cfg::TN; ++j)
                        acc[set][i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][s], b[j][s], acc[set][i][j], 0, 0, 0);
            __builtin_amdgcn_s_setprio(0);
        }
    };

    // Prefetch distance two for the tiles of up to 64 x 64 (round 3): the tile AFTER next is requested before this tile's MFMA block,
    // into a second set of staging registers, so that a K iteration no longer has to cover a whole L2 round trip -- with 16 K
    // tiles or fewer per output tile and an MFMA block of a fraction of a microsecond these loops were chains of load latencies
    // (steady state 0.94 us per 64-deep K tile where the matrix pipe needs 0.53).  Same instructions in the same order on every
    // accumulator: the bits do not change.  Measured: cross-q 12.9 -> 11.6 us, output projection 11.5 -> 10.6, q|k|v 22.6 -> 21.7,
    // FFN 27.6 -> 26.6, encoder q|k|v 160 -> 155, vocabulary^T 109.9 -> 106.3; captions/s +1.3 % (four streams), +2.7 % (one).
    // Not for the larger tiles (their second register set costs a resident workgroup: 128 x 64 160.9 -> 166.4 us) nor for the
    // K-tile-64 instances with chains in two waves (96-register cap: 13 spilled).  With two LDS buffers on top (one barrier per
    // K tile instead of two) every shape got slower again (encoder q|k|v 154.6 -> 163.8 us): LDS residency beats the barrier.
#ifdef OVC_NO_PF2                   // A/B builds only: tools/ab_bench.sh against a library compiled with -DOVC_NO_PF2
    constexpr bool kPF2 = false;
#else
    constexpr bool kPF2 = BM * BN <= 64 * 64 && Cfg::kBufs == 1 && !(WK == 2 && BK == 64);
#endif
    constexpr int kDepth = 2;       // staging register sets = tiles requested ahead (3 and 4 measured: slower on every shape --
                                    // 32x32 cross-q 11.9 / 12.6 / 12.5 us, 64x64 vocabulary^T 105.7 / 110.2 / 113.3 -- even without spills)
    bool done = false;
    if constexpr (kPF2) {
        if (!k_tail && p.K2 == 0 && !row_sums && nkt >= kDepth) {
            f32x4 sa[kDepth][Cfg::kLoadA], sb[kDepth][Cfg::kLoadB];
            auto load_set = [&](int kt, auto set_tag) {
                constexpr int S = decltype(set_tag)::value;
                const int k0 = kbase + kt * BK;
#pragma unroll
                for (int i = 0; i < Cfg::kLoadA; ++i)
                    sa[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a1, off_a1[i], k0 * 4, 0));
#pragma unroll
                for (int i = 0; i < Cfg::kLoadB; ++i)
                    sb[S][i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, off_w[i], k0 * 4, 0));
            };
            auto store_set = [&](auto set_tag) {
                constexpr int S = decltype(set_tag)::value;
                const int kq = tid % kVecPerRow;
#pragma unroll
                for (int i = 0; i < Cfg::kLoadA; ++i)
                    *reinterpret_cast<f32x4*>(lds + (tid / kVecPerRow + i * kRowsPerPass) * LDT + kq * 4) = sa[S][i];
#pragma unroll
                for (int i = 0; i < Cfg::kLoadB; ++i)
                    *reinterpret_cast<f32x4*>(lds + (BM + tid / kVecPerRow + i * kRowsPerPass) * LDT + kq * 4) = sb[S][i];
            };
            // tile j waits in register set j % kDepth.  One step: tile kt is in LDS and its set is free -> request tile kt + kDepth
            // into it, run tile kt's MFMA block, move tile kt + 1 (requested kDepth - 1 blocks ago) to LDS.
            auto step = [&](int kt, auto cur_tag) {
                constexpr int C = decltype(cur_tag)::value;
                if (kt + kDepth < nkt) load_set(kt + kDepth, cur_tag);
                compute_tile(0);
                if (kt + 1 < nkt) {
                    __syncthreads();
                    store_set(std::integral_constant<int, (C + 1) % kDepth>{});
                    __syncthreads();
                }
            };
            load_set(0, std::integral_constant<int, 0>{});
            if constexpr (kDepth > 1) load_set(1, std::integral_constant<int, 1 % kDepth>{});
            if constexpr (kDepth > 2) load_set(2, std::integral_constant<int, 2 % kDepth>{});
            if constexpr (kDepth > 3) load_set(3, std::integral_constant<int, 3 % kDepth>{});
            store_set(std::integral_constant<int, 0>{});
            __syncthreads();
            for (int kt = 0; kt < nkt; kt += kDepth) {
                step(kt, std::integral_constant<int, 0>{});
                if constexpr (kDepth > 1) { if (kt + 1 < nkt) step(kt + 1, std::integral_constant<int, 1 % kDepth>{}); }
                if constexpr (kDepth > 2) { if (kt + 2 < nkt) step(kt + 2, std::integral_constant<int, 2 % kDepth>{}); }
                if constexpr (kDepth > 3) { if (kt + 3 < nkt) step(kt + 3, std::integral_constant<int, 3 % kDepth>{}); }
            }
            __syncthreads();
            done = true;
        }
    }
    if (!done) {
        load_tile(0);
        store_tile(0);
        __syncthreads();
        for (int kt = 0; kt < nkt; ++kt) {
            const int buf = Cfg::kBufs == 2 ? (kt & 1) : 0;
            if (kt + 1 < nkt) load_tile(kt + 1);
            compute_tile(buf);
            if (kt + 1 < nkt) {
                if (Cfg::kBufs == 1) __syncthreads();
                store_tile(Cfg::kBufs == 2 ? (buf ^ 1) : 0);
            }
            __syncthreads();
        }
    }

    if (row_sums) {
        // the kVecPerRow consecutive lanes that staged one row hold its partial sums: butterfly inside that group
#pragma unroll
        for (int i = 0; i < Cfg::kLoadA; ++i) {
            float v = rs[i];
#pragma unroll
            for (int off = 1; off < kVecPerRow; off <<= 1) v += __shfl_xor(v, off, 64);
            const int row = m0 + tid / kVecPerRow + i * kRowsPerPass;
            if (tid % kVecPerRow == 0 && row < arg_M) zero_rows_ptr[row] = (v == 0.f) ? 1 : 0;
        }
    }

    // Chain reduction, always in chain order ((c0 + c1) + c2) + c3 whatever the wave layout: chain c lives in wave
    // c % WK, accumulator set c / WK.  Waves wk > 0 park their sets in LDS (the tile buffers are free after the last
    // barrier), wave wk == 0 adds everything up and runs the epilogue.
    if (Cfg::kChains > 1) {
        float* red = lds;
        const int wtile = wm * WN + wn;
        auto red_index = [&](int w, int c, int i, int j, int r) {
            return ((((((w - 1) * NC + c) * (WM * WN) + wtile) * Cfg::TM + i) * Cfg::TN + j) * 16 + r) * 64 + lane;
        };
        if (WK > 1) {
            if (wk > 0) {
#pragma unroll
                for (int c = 0; c < NC; ++c)
#pragma unroll
                    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                        for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                            for (int r = 0; r < 16; ++r) red[red_index(wk, c, i, j, r)] = acc[c][i][j][r];
            }
            __syncthreads();
            if (wk > 0) return;
        }
#pragma unroll
        for (int chain = 1; chain < Cfg::kChains; ++chain) {
            constexpr int kW = WK;
            const int w = chain % kW, c = chain / kW;          // compile-time after unrolling
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
                for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        acc[0][i][j][r] += w == 0 ? acc[c][i][j][r] : red[red_index(w, c, i, j, r)];
        }
    }

    store_wave_tiles<Cfg::TM, Cfg::TN>(p, seg_bias, seg_C, acc[0], m0 + wm * Cfg::kWaveM, n0 + wn * Cfg::kWaveN, lane);
}

// tile order shared by both kernels: super-rows of M tiles when the A panel exceeds an L2, else the 2-D XCD split with
// the least Infinity-Cache traffic  pn * |A| + pm * |W|
void tile_order(const GemmArgs& a, int tiles_m, int tiles_n, int* group_m, int* xcd_pm) {
    const size_t a_bytes = sizeof(float) * (size_t)a.M * (a.K1 + a.K2);
    *group_m = a_bytes <= (size_t)3 << 20 ? tiles_m : (tiles_m < 8 ? tiles_m : 8);
    *xcd_pm = 0;
    const int slices = a.ksplit > 1 ? a.ksplit : 1;
    const size_t w_bytes = sizeof(float) * (size_t)a.seg_n * a.nseg * (a.K1 + a.K2);
    const int tiles_n_all = tiles_n * a.nseg;
    if (a_bytes / slices <= (size_t)3 << 20) {
        size_t best = 8 * a_bytes + w_bytes;
        for (int pm = 2; pm <= 8; pm *= 2) {
            if (tiles_m % pm || tiles_n_all % (8 / pm)) continue;
            const size_t cost = (size_t)(8 / pm) * a_bytes + (size_t)pm * w_bytes;
            if (cost < best) { best = cost; *xcd_pm = pm; }
        }
    } else if (slices == 1 && w_bytes <= (size_t)3 << 20) {
        // A exceeds an L2 but W fits one and an XCD's share of A does (the transposed vocabulary product: A = fc, 21 MB,
        // W = the 2.6 MB of decoder outputs): split M over the XCDs -- inside a sub-rectangle M runs fastest, so the XCD's A
        // panel is re-used for every N tile and A is read ONCE overall: pn |A| + pm |W| = 42 MB where super-rows read 74 MB
        size_t best = ~(size_t)0;
        for (int pm = 8; pm >= 2; pm /= 2) {
            if (tiles_m % pm || tiles_n_all % (8 / pm) || a_bytes / pm > (size_t)3 << 20) continue;
            const size_t cost = (size_t)(8 / pm) * a_bytes + (size_t)pm * w_bytes;
            if (cost < best) { best = cost; *xcd_pm = pm; }
        }
    }
}

// Host side of tile_coords_fast: the divisors of the tile map with their multiply-high magics.
TileMap make_tile_map(int nwg, int tiles_m, int tiles_n_per_seg, int group_m, int xcd_pm, int kslice) {
    auto magic = [](int d) { return d > 1 ? (unsigned)(((uint64_t)1 << 32) / (uint64_t)d + 1) : 0u; };
    TileMap t{};
    t.nwg = nwg; t.tiles_m = tiles_m; t.tiles_n_all = nwg / tiles_m; t.tiles_n_per_seg = tiles_n_per_seg;
    t.xcd_pm = xcd_pm; t.kslice = kslice;
    t.seg_magic = magic(tiles_n_per_seg);
    if (xcd_pm > 0) {
        t.pm_shift = xcd_pm == 8 ? 3 : (xcd_pm == 4 ? 2 : (xcd_pm == 2 ? 1 : 0));
        t.sub_m = tiles_m / xcd_pm; t.sub_n = t.tiles_n_all / (8 / xcd_pm);
        t.sub_m_magic = magic(t.sub_m);
        t.group_m = tiles_m; t.group_size = nwg; t.gm_last = tiles_m;       // unused on this path; kept valid
    } else {
        t.group_m = group_m; t.group_size = group_m * t.tiles_n_all;
        const int rest = tiles_m % group_m;
        t.gm_last = rest ? rest : group_m;
        t.sub_m = 1; t.sub_n = 1;
    }
    t.group_size_magic = magic(t.group_size); t.group_m_magic = magic(t.group_m); t.gm_last_magic = magic(t.gm_last);
    return t;
}

template <int BM, int BN, int WM, int WN, int WK, int BK, int NC>
int launch_config(const GemmArgs& a, hipStream_t stream, const GemmLaunchOpts& opts) {
    using Cfg = TileConfig<BM, BN, WM, WN, WK, BK, NC>;
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tiles_n = (a.seg_n + BN - 1) / BN;
    const int grid = tiles_m * tiles_n * a.nseg;
    const size_t lds_bytes = sizeof(float) * Cfg::kLdsFloats;
    static std::once_flag attr_once;   // raise the dynamic-LDS cap once per process
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f32_mfma<BM, BN, WM, WN, WK, BK, NC>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    });
    int group_m, xcd_pm;
    tile_order(a, tiles_m, tiles_n, &group_m, &xcd_pm);
    const int slices = a.ksplit > 1 ? a.ksplit : 1;         // K slices run one after the other (gridDim.y is the slow index)
    const TileMap map = make_tile_map(grid, tiles_m, tiles_n, group_m, xcd_pm, (a.K1 + a.K2) / slices);
    // opts.copies (tuner only): gridDim.z identical copies of the product in one launch (the kernel ignores
    // blockIdx.z), a proxy for "this many batches in flight" that needs no extra streams.
    const dim3 grid3(grid, slices, opts.copies > 1 ? opts.copies : 1);
    if (opts.start && opts.stop)     // kernel-scoped events: the dispatch packet's own begin / end timestamps
        hipExtLaunchKernelGGL((gemm_f32_mfma<BM, BN, WM, WN, WK, BK, NC>), grid3, dim3(256), (uint32_t)lds_bytes, stream,
                              opts.start, opts.stop, 0, map, a);
    else
        hipLaunchKernelGGL((gemm_f32_mfma<BM, BN, WM, WN, WK, BK, NC>), grid3, dim3(256), lds_bytes, stream, map, a);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

template <int BM, int BN, int WM, int WN, int BK, int MODE, bool WD = false>
int launch_split_config(const GemmArgs& a, hipStream_t stream, const GemmLaunchOpts& opts) {
    using Cfg = SplitConfig<BM, BN, WM, WN, BK, MODE, WD>;
    if constexpr (!WD) {             // pre-cut weights on every segment: the instance that reads them straight from memory
        bool planes = (a.K1 + a.K2) % BK == 0 && a.K1 % BK == 0;   // whole K tiles only: the planes have no zero tail to read
        for (int s = 0; s < a.nseg; ++s) planes = planes && a.seg[s].Wp != nullptr;
        if (planes) return launch_split_config<BM, BN, WM, WN, BK, MODE, true>(a, stream, opts);
    }
    const int tiles_m = (a.M + BM - 1) / BM;
    const int tiles_n = (a.seg_n + BN - 1) / BN;
    const int grid = tiles_m * tiles_n * a.nseg;
    const size_t lds_bytes = Cfg::kLdsBytes;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split_mfma<BM, BN, WM, WN, BK, MODE, WD>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    });
    int group_m, xcd_pm;
    tile_order(a, tiles_m, tiles_n, &group_m, &xcd_pm);
    const dim3 grid3(grid, a.ksplit > 1 ? a.ksplit : 1, opts.copies > 1 ? opts.copies : 1);
    if (opts.start && opts.stop)
        hipExtLaunchKernelGGL((gemm_split_mfma<BM, BN, WM, WN, BK, MODE, WD>), grid3, dim3(256), (uint32_t)lds_bytes, stream,
                              opts.start, opts.stop, 0, a, tiles_m, tiles_n, group_m, xcd_pm);
    else
        hipLaunchKernelGGL((gemm_split_mfma<BM, BN, WM, WN, BK, MODE, WD>), grid3, dim3(256), lds_bytes, stream, a, tiles_m, tiles_n, group_m, xcd_pm);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

template <int NB>
int launch_rows16_config(const GemmArgs& a, hipStream_t stream, const GemmLaunchOpts& opts) {
    const int tiles_m = (a.M + 15) / 16;
    const int tiles_n = (a.seg_n + 16 * NB - 1) / (16 * NB);
    const size_t lds_bytes = sizeof(float) * 4 * (16 + 16 * NB) * kRows16Ldt;
    static std::once_flag attr_once;
    std::call_once(attr_once, [&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_rows16_f32<NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    });
    const int kslice = a.K1 / (a.ksplit > 1 ? a.ksplit : 1);
    const dim3 grid3(tiles_n * a.nseg, tiles_m * (opts.copies > 1 ? opts.copies : 1), a.ksplit > 1 ? a.ksplit : 1);
    if (opts.start && opts.stop)
        hipExtLaunchKernelGGL((gemm_rows16_f32<NB>), grid3, dim3(256), (uint32_t)lds_bytes, stream, opts.start, opts.stop, 0, a, tiles_m, tiles_n, kslice);
    else
        hipLaunchKernelGGL((gemm_rows16_f32<NB>), grid3, dim3(256), lds_bytes, stream, a, tiles_m, tiles_n, kslice);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

// bm, bn, wm, wn, wk, bk, nc, planes.  fp32 tilings (planes = 0): chains = wk * nc is the K-order class the instance
// belongs to; split-precision tilings (planes = the kernel's MODE: 3 = three bf16 planes, 4 = two fp16 planes): class
// 100 + MODE (kSplitClass).
struct TilingInfo { int bm, bn, wm, wn, wk, bk, nc, planes; };
constexpr int kSplitClass = 100;
#define OVC_TILINGS(X)                                                                                   \
    /* one chain (kchains = 1) */                                                                        \
    X(0, 128, 128, 2, 2, 1, 32, 1) X(1, 64, 128, 2, 2, 1, 32, 1) X(2, 128, 64, 2, 2, 1, 32, 1)          \
    X(3, 64, 64, 2, 2, 1, 32, 1) X(4, 64, 64, 2, 2, 1, 64, 1) X(5, 64, 128, 2, 2, 1, 64, 1)             \
    /* four chains (kchains = 4): in one wave, in two, in four */                                        \
    X(6, 64, 128, 2, 2, 1, 32, 4) X(7, 64, 64, 2, 2, 1, 32, 4) X(8, 64, 64, 2, 2, 1, 64, 4)             \
    X(9, 32, 64, 1, 2, 2, 32, 2) X(10, 64, 32, 2, 1, 2, 32, 2) X(11, 32, 64, 1, 2, 2, 64, 2)            \
    X(12, 64, 32, 2, 1, 2, 64, 2) X(13, 32, 32, 1, 1, 4, 32, 1) X(14, 32, 32, 1, 1, 4, 64, 1)          \
    /* one chain, shallow K tiles: half the LDS per workgroup -> twice the workgroups per CU, whose prologues and */ \
    /* epilogues then overlap other workgroups' MFMA (the fixed cost per round of workgroups drops by a third)   */ \
    X(15, 64, 64, 2, 2, 1, 16, 1) X(16, 128, 128, 2, 2, 1, 16, 1)
//  (K tiles of 128 were tried for the small tilings: slower -- 15.6 vs 12.2 us on 1280x512x512)
// split-precision tilings: id, bm, bn, wm, wn, bk, planes (one chain each; ids follow the fp32 ones)
#define OVC_SPLIT_SHAPES(X, first, planes)                                                                \
    X(first + 0, 128, 128, 2, 2, 32, planes) X(first + 1, 64, 128, 2, 2, 32, planes) X(first + 2, 128, 64, 2, 2, 32, planes) \
    X(first + 3, 64, 64, 2, 2, 32, planes) X(first + 4, 32, 128, 1, 4, 32, planes)
// deep K tiles for the M = B*k decode products: their MFMA block per K tile is a fraction of a microsecond, so a K loop is
// a chain of load latencies and fewer, larger tiles halve it
#define OVC_SPLIT_DEEP(X, first, planes) X(first + 0, 64, 64, 2, 2, 64, planes) X(first + 1, 64, 128, 2, 2, 64, planes)
#define OVC_SPLIT_TILINGS(X) OVC_SPLIT_SHAPES(X, 17, 3) OVC_SPLIT_SHAPES(X, 22, 4) OVC_SPLIT_DEEP(X, 27, 3) OVC_SPLIT_DEEP(X, 29, 4)
// small-rows instances of the four-chain class (gemm_rows16.h): id, 16-column blocks per workgroup.  Ids follow the others so that
// a tuning cache written by an earlier build keeps its meaning.
#define OVC_ROWS16_TILINGS(X) X(31, 1) X(32, 2)
constexpr int kRows16MaxM = 112;         // B <= 22 at beam 5 (a B = 25 search was no faster with them: 5.03 against 4.95 ms; B = 20: 4.78 against 4.97).  From ~80 rows on only the narrow products still win on these instances
                                         // (tools/gemm_bench.py small, rows 64 / 80 / 128: 1280-column-tile grids lose to the 32 x 32 tiles); the tuner ranks
#define OVC_TILING_INFO(id, bm, bn, wm, wn, wk, bk, nc) {bm, bn, wm, wn, wk, bk, nc, 0},
#define OVC_SPLIT_INFO(id, bm, bn, wm, wn, bk, planes) {bm, bn, wm, wn, 1, bk, 1, planes},
#define OVC_ROWS16_INFO(id, nb) {16, 16 * nb, 1, 1, 4, 32, 1, 0},
constexpr TilingInfo kTilings[] = {OVC_TILINGS(OVC_TILING_INFO) OVC_SPLIT_TILINGS(OVC_SPLIT_INFO) OVC_ROWS16_TILINGS(OVC_ROWS16_INFO)};
#undef OVC_TILING_INFO
#undef OVC_SPLIT_INFO
#undef OVC_ROWS16_INFO
constexpr int kNumTilings = sizeof(kTilings) / sizeof(kTilings[0]);
static_assert(kNumTilings == 33, "tiling ids: 17 fp32 + 14 split-precision (modes 3 and 4) + 2 small-rows fp32");
inline int tiling_chains(int t) { return kTilings[t].planes ? kSplitClass + kTilings[t].planes : kTilings[t].wk * kTilings[t].nc; }
inline bool class_ok(int c) { return c == 1 || c == 4 || c == kSplitClass + 3 || c == kSplitClass + 4; }

// One device per process (include/ovc.h): the library keeps per-process state that belongs to a device -- kernel attributes
// raised once (dynamic LDS caps), the graph-capture stream, captured graphs, tuning tables.  The first launching call binds
// the library to the device that is current then; a call made with another device current is refused, loudly.
std::atomic<int> g_bound_device{-1};

// Debug hook (ovc_debug_force_gemm_tiling): applies to every launch that does not carry its own
// GemmLaunchOpts::forced_tiling and whose class matches; not for use while other threads decode.
std::atomic<int> g_forced_tiling{-1};

// Shapes measured by ovc_gemm_tune: (M, seg_n, nseg, K, kchains, ksplit) -> fastest tiling of that class on this
// device.  Guarded by g_tuned_mutex (the engine may be driven from several host threads).
struct TunedShape { int M, seg_n, nseg, K, kchains, ksplit, objective, tiling; };
std::vector<TunedShape> g_tuned;
std::mutex g_tuned_mutex;
std::atomic<long> g_tune_calls{0};      // measurements actually run (tests assert that bucketed shapes re-use entries)

// Exact entry, or (near = true) the entry of the same product whose M (or, with M equal, whose column count) is closest within a factor of two: the best
// tiling moves slowly with M, and every tiling of a class gives the same bits, so borrowing a neighbour's choice
// costs at most a little speed.  Batches whose region count varies (M = B*N) then never wait for a tuning run.
// `objective` = how many identical products co-ran when the entry was measured (ovc_gemm_tune_objective): a caller that
// keeps several batches in flight looks up the table measured that way, one that decodes alone the isolated one.
int tuned_lookup(int M, int seg_n, int nseg, int K, int kchains, int ksplit, int objective, bool near) {
    std::lock_guard<std::mutex> lock(g_tuned_mutex);
    int best = -1;
    double best_ratio = 2.0;
    for (const TunedShape& t : g_tuned) {
        if (t.nseg != nseg || t.K != K || t.kchains != kchains || t.ksplit != ksplit || t.objective != objective) continue;
        if (t.M == M && t.seg_n == seg_n) return t.tiling;
        if (!near) continue;
        // a neighbour along ONE axis: the same columns with M within a factor of two, or -- single-segment products -- the same
        // rows with the column count within a factor of two.  The second form is the transposed vocabulary product's (M = V
        // words, seg_n = B * width beam rows): the batch size moves its COLUMNS, and the ragged last batch of a prediction loop
        // must not wait for a stream-synchronising measurement of the decode step's heaviest product.
        double ratio;
        if (t.seg_n == seg_n) ratio = t.M > M ? (double)t.M / M : (double)M / t.M;
        else if (t.M == M && nseg == 1) ratio = t.seg_n > seg_n ? (double)t.seg_n / seg_n : (double)seg_n / t.seg_n;
        else continue;
        if (ratio <= best_ratio) { best_ratio = ratio; best = t.tiling; }
    }
    return best;
}

// Predicted time of a tiling in MFMA-issue units (one unit = one v_mfma_f32_32x32x2_f32 slot of a
// SIMD): every SIMD of a CU executes the waves of the workgroups resident on that CU, so the critical
// path is ceil(workgroups / 256 CUs) * (MFMAs one wave issues per workgroup), plus a per-workgroup
// overhead (prologue load, epilogue, barriers) that smaller tiles pay more often, plus a bandwidth
// term for tiles whose operand traffic per FLOP is high.
double tiling_cost(const GemmArgs& a, const TilingInfo& t) {
    const int K = a.K1 + a.K2;
    const long wgs = (long)((a.M + t.bm - 1) / t.bm) * ((a.seg_n + t.bn - 1) / t.bn) * a.nseg;
    const double per_cu = (double)((wgs + 255) / 256);                       // workgroups on the busiest CU
    // split precision: planes (planes + 1) / 2 bf16 MFMAs of 32 cycles per 16-deep step against one fp32 MFMA of 64 per 2
    const int products = t.planes == 4 ? 3 : t.planes * (t.planes + 1) / 2;      // mode 4: two fp16 planes
    const double k_units = t.planes ? K / 16.0 * products * 0.5 : K / 2.0;
    // 16-row instances: a wave runs its chain's K / 4 on 16x16x4 instructions of half a slot each, per 16-column block
    const double mfma_per_wave = t.bm == 16 ? (t.bn / 16) * (K / 16.0) * 0.5 : (double)(t.bm / 32) * (t.bn / 32) * k_units / 4.0;
    const double overhead = (t.bm == 16 ? 16.0 : 48.0) + (t.bk >= 64 && !t.planes ? 1e9 : 0.0);    // fp32 BK=64 tilings: only when measured or forced
    const double bytes_per_flop = 2.0 * (t.bm + t.bn) / (double)(t.bm * t.bn);   // operand floats per MAC
    const double bw_penalty = 1.0 + 4.0 * bytes_per_flop;                    // 128x128: 1.06, 32x32: 1.5
    return per_cu * (mfma_per_wave * bw_penalty + overhead);
}

int args_chains(const GemmArgs& a) { return a.kchains > kSplitClass ? a.kchains : (a.kchains == 4 ? 4 : 1); }

// A tiling fits a problem when it belongs to the problem's K-order class, no tile straddles two N segments and the
// A1|A2 seam / the K slices fall on K-tile boundaries.
bool tiling_fits(const GemmArgs& a, int t) {
    if (tiling_chains(t) != args_chains(a)) return false;
    if (a.nseg > 1 && a.seg_n % kTilings[t].bn) return false;
    if (a.K2 > 0 && a.K1 % kTilings[t].bk) return false;
    if (a.ksplit > 1 && (a.K1 / a.ksplit) % kTilings[t].bk) return false;     // a slice is a whole number of K tiles
    if (kTilings[t].bm == 16 && (a.M > kRows16MaxM || a.K2 || a.R || a.stats || a.stats_t || a.zero_rows_out)) return false;   // gemm_rows16.h
    return true;
}

}  // namespace

extern "C" size_t ovc_split_weight_bytes(int N, int K, int mode);

int ovc_device_guard() {
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess) { (void)hipGetLastError(); return OVC_ELAUNCH; }
    int bound = g_bound_device.load(std::memory_order_relaxed);
    if (bound < 0 && g_bound_device.compare_exchange_strong(bound, dev)) return OVC_OK;
    return bound == dev ? OVC_OK : OVC_EDEVICE;
}

extern "C" int ovc_bound_device(void) { return g_bound_device.load(); }

extern "C" int ovc_debug_rebind_device(int device) {
    g_bound_device.store(device < 0 ? -1 : device);
    return OVC_OK;
}

// Forget every measured tiling (tests: a measurement-count assertion must not depend on what earlier tests left in the table).
extern "C" int ovc_debug_clear_tuning(void) {
    std::lock_guard<std::mutex> lock(g_tuned_mutex);
    g_tuned.clear();
    return OVC_OK;
}

extern "C" int ovc_debug_force_gemm_tiling(int tiling) {
    if (tiling < -1 || tiling >= kNumTilings) return OVC_EINVAL;
    g_forced_tiling.store(tiling);
    return OVC_OK;
}

const char* ovc_gemm_tiling_name(int tiling) {
#define OVC_TILING_NAME(id, bm, bn, wm, wn, wk, bk, nc) "gemm_f32_mfma<" #bm ", " #bn ", " #wm ", " #wn ", " #wk ", " #bk ", " #nc ">",
#define OVC_SPLIT_NAME(id, bm, bn, wm, wn, bk, planes) "gemm_split_mfma<" #bm ", " #bn ", " #wm ", " #wn ", " #bk ", " #planes ">",
#define OVC_ROWS16_NAME(id, nb) "gemm_rows16_f32<" #nb ">",
    static const char* names[] = {OVC_TILINGS(OVC_TILING_NAME) OVC_SPLIT_TILINGS(OVC_SPLIT_NAME) OVC_ROWS16_TILINGS(OVC_ROWS16_NAME)};
#undef OVC_TILING_NAME
#undef OVC_SPLIT_NAME
#undef OVC_ROWS16_NAME
    return tiling >= 0 && tiling < kNumTilings ? names[tiling] : "";
}

int ovc_gemm_tiling_class(int tiling) { return tiling >= 0 && tiling < kNumTilings ? tiling_chains(tiling) : 0; }

int ovc_gemm_pick_tiling(const GemmArgs& a, const GemmLaunchOpts& opts) {
    if (opts.forced_tiling >= 0) return opts.forced_tiling < kNumTilings && tiling_fits(a, opts.forced_tiling) ? opts.forced_tiling : -1;
    const int forced = g_forced_tiling.load();
    if (forced >= 0 && tiling_fits(a, forced)) return forced;
    const int objective = a.objective > 1 ? a.objective : 1;
    int tuned = tuned_lookup(a.M, a.seg_n, a.nseg, a.K1 + a.K2, args_chains(a), a.ksplit > 1 ? a.ksplit : 1, objective, true);
    if (tuned < 0 && objective > 1)     // nothing measured under that load: the isolated entry is still better than the cost model
        tuned = tuned_lookup(a.M, a.seg_n, a.nseg, a.K1 + a.K2, args_chains(a), a.ksplit > 1 ? a.ksplit : 1, 1, true);
    if (tuned >= 0 && tiling_fits(a, tuned)) return tuned;
    double best = 1e300;
    int pick = -1;
    for (int i = 0; i < kNumTilings; ++i) {
        if (!tiling_fits(a, i)) continue;
        const double c = tiling_cost(a, kTilings[i]);
        if (c < best) { best = c; pick = i; }
    }
    return pick;
}

int ovc_gemm_launch(const GemmArgs& a, hipStream_t stream, const GemmLaunchOpts& opts) {
    if (const int rc = ovc_device_guard()) return rc;
    const int K = a.K1 + a.K2;
    if (a.M <= 0 || a.seg_n <= 0 || a.nseg <= 0 || a.nseg > OVC_MAX_SEGMENTS || K <= 0) return OVC_EINVAL;
    if (a.kchains != 0 && !class_ok(a.kchains)) return OVC_EINVAL;
    if ((a.K1 & 3) || (a.K2 & 3) || (a.lda1 & 3) || (a.K2 && (a.lda2 & 3))) return OVC_EINVAL;
    if (!ovc_aligned16(a.A1) || (a.K2 && !a.A2 && !a.seg[0].A2) || (a.A2 && !ovc_aligned16(a.A2))) return OVC_EINVAL;
    for (int s = 0; s < a.nseg; ++s)
        if (a.seg[s].A2 && (!a.K2 || !ovc_aligned16(a.seg[s].A2))) return OVC_EINVAL;
    if (a.K2 && !a.A2)                                   // no shared second block: every segment brings its own
        for (int s = 0; s < a.nseg; ++s)
            if (!a.seg[s].A2) return OVC_EINVAL;
    if (a.R && a.nseg != 1) return OVC_EINVAL;
    if (a.zero_rows_out && (a.K2 || a.ksplit > 1 || a.kchains > kSplitClass)) return OVC_EINVAL;
    if (a.stats && (a.nseg != 1 || a.ksplit > 1 || !ovc_aligned16(a.stats) || a.stats_ld < (a.seg_n + 31) / 32)) return OVC_EINVAL;
    if (a.stats_t && (a.stats || a.seg[0].bias || a.nseg != 1 || a.ksplit > 1 || !ovc_aligned16(a.stats_t) || a.stats_ld < (a.M + 31) / 32)) return OVC_EINVAL;
    if (a.K2 > 0 && (a.K1 % 32)) return OVC_EINVAL;      // the A1|A2 seam must fall on a K-tile boundary
    if (a.ksplit > 1) {                                  // raw partial products: see GemmArgs::ksplit
        if (a.ksplit > kMaxKSplit || a.nseg != 1 || a.K2 || a.R || a.act || a.seg[0].bias) return OVC_EINVAL;
        if (a.K1 % (a.ksplit * 32) || a.part_stride < (long)a.M * a.ldc) return OVC_EINVAL;
    }
    for (int s = 0; s < a.nseg; ++s)
        if (!a.seg[s].W || !a.seg[s].C || !ovc_aligned16(a.seg[s].W)) return OVC_EINVAL;
    if (a.nseg > 1 && a.seg_n % 64) return OVC_EINVAL;   // a tile may not straddle two segments
    // buffer descriptors address 32-bit byte ranges (rows past the end must stay representable)
    const long kMaxBytes = 0x7fffffffL - (1L << 20);
    if ((long)(a.M + 256) * a.lda1 * 4 > kMaxBytes || (a.K2 && (long)(a.M + 256) * a.lda2 * 4 > kMaxBytes) ||
        (long)(a.seg_n + 256) * K * 4 > kMaxBytes || (long)(a.M + 256) * a.ldc * 4 > kMaxBytes ||
        (a.R && (long)(a.M + 256) * a.ldr * 4 > kMaxBytes)) return OVC_EINVAL;

    const int pick = ovc_gemm_pick_tiling(a, opts);
    switch (pick) {
#define OVC_TILING_CASE(id, bm, bn, wm, wn, wk, bk, nc) case id: return launch_config<bm, bn, wm, wn, wk, bk, nc>(a, stream, opts);
#define OVC_SPLIT_CASE(id, bm, bn, wm, wn, bk, planes) case id: return launch_split_config<bm, bn, wm, wn, bk, planes>(a, stream, opts);
#define OVC_ROWS16_CASE(id, nb) case id: return launch_rows16_config<nb>(a, stream, opts);
        OVC_TILINGS(OVC_TILING_CASE)
        OVC_SPLIT_TILINGS(OVC_SPLIT_CASE)
        OVC_ROWS16_TILINGS(OVC_ROWS16_CASE)
#undef OVC_TILING_CASE
#undef OVC_SPLIT_CASE
#undef OVC_ROWS16_CASE
        default: return OVC_EINVAL;
    }
}

// Measure every tiling of one K-order class on one GEMM shape and remember the fastest (process-wide).  Synchronises
// the stream: call it at set-up time, never inside a captured or latency-sensitive region.  The choice changes speed
// only: all tilings of a class produce the same bits.
extern "C" int ovc_gemm_tune(int M, int seg_n, int nseg, int K, int kchains, int ksplit, int objective, int epilogue, void* scratch,
                             size_t scratch_bytes, ovc_stream stream) {
    if (objective < 1 || objective > 8 || epilogue < 0 || epilogue > 2) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;
    if (M <= 0 || seg_n <= 0 || nseg <= 0 || nseg > OVC_MAX_SEGMENTS || K <= 0 || (K & 3)) return OVC_EINVAL;
    if (!class_ok(kchains) || ksplit < 1 || ksplit > kMaxKSplit || (ksplit > 1 && (nseg != 1 || K % (ksplit * 32)))) return OVC_EINVAL;
    const size_t na = (size_t)M * K, nw = (size_t)seg_n * nseg * K, nc = (size_t)M * seg_n * nseg;
    if (!scratch || !ovc_aligned16(scratch)) return OVC_EWORKSPACE;
    if (nseg > 1 && seg_n % 64) return OVC_EINVAL;
    float* A = reinterpret_cast<float*>(scratch);
    float* W = A + ((na + 3) & ~(size_t)3);
    float* C = W + ((nw + 3) & ~(size_t)3);
    if ((size_t)(C - A) + (size_t)ksplit * nc > scratch_bytes / sizeof(float)) return OVC_EWORKSPACE;
    if (tuned_lookup(M, seg_n, nseg, K, kchains, ksplit, objective, false) >= 0) return OVC_OK;
    GemmArgs a{};
    a.A1 = A; a.lda1 = K; a.K1 = K; a.M = M; a.seg_n = seg_n; a.nseg = nseg; a.ldc = seg_n; a.kchains = kchains;
    if (ksplit > 1) { a.ksplit = ksplit; a.part_stride = (long)nc; }
    for (int s = 0; s < nseg; ++s) a.seg[s] = GemmSegment{W + (size_t)s * seg_n * K, nullptr, C + (size_t)s * M * seg_n, nullptr, nullptr};
    // split-precision classes: when the scratch buffer has room behind the outputs, rank the instances that read pre-cut
    // weight planes (what the engine runs when the host supplies ovc_lin::planes); any bit pattern will do for a timing
    if (kchains > kSplitClass && (K & 15) == 0) {
        const size_t plane_floats = (ovc_split_weight_bytes(seg_n, K, kchains - kSplitClass) + 3) / 4;
        float* P = C + (((size_t)ksplit * nc + 3) & ~(size_t)3);
        if ((size_t)(P - A) + (size_t)nseg * plane_floats <= scratch_bytes / sizeof(float))
            for (int s = 0; s < nseg; ++s) a.seg[s].Wp = P + (size_t)s * plane_floats;
    }
    // Wide single-segment products of the decode class (the engine has one: the vocabulary projection) run with the
    // log-softmax epilogue (GemmArgs::stats); rank their tilings WITH it when the scratch buffer has room behind the outputs
    // (its cost depends on the tiling's register budget: the K-tile-64 instance loses 5 us to it, the K-tile-32 one 4)
    // A product that runs with the log-softmax epilogue (GemmArgs::stats / stats_t: the engine's vocabulary projection) is
    // ranked WITH it when the scratch buffer has room behind the outputs: its cost depends on the tiling's register budget
    // (the K-tile-64 instance loses 5 us to the row-major form, the K-tile-32 one 4).  epilogue: 1 = stats, 2 = stats_t.
    if ((epilogue == 1 || epilogue == 2) && nseg == 1 && ksplit == 1) {
        const bool transposed = epilogue == 2;
        const int ld = (((transposed ? M : seg_n) + 31) / 32 + 1) & ~1;
        const size_t entries = (size_t)(transposed ? seg_n : M) * ld;
        float* S = C + ((nc + 3) & ~(size_t)3);
        if ((size_t)(S - A) + 2 * entries <= scratch_bytes / sizeof(float)) { (transposed ? a.stats_t : a.stats) = S; a.stats_ld = ld; }
    }
    hipStream_t st = ovc_hip_stream(stream);
    hipEvent_t e0, e1;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return OVC_ELAUNCH;
    int rc = OVC_OK, best = -1;
    float best_ms = 1e30f;
    GemmLaunchOpts opts{};
    opts.copies = objective;
    for (int t = 0; t < kNumTilings && rc == OVC_OK; ++t) {
        if (!tiling_fits(a, t)) continue;
        opts.forced_tiling = t;
        for (int i = 0; i < 2 && rc == OVC_OK; ++i) rc = ovc_gemm_launch(a, st, opts);
        // the faster of three groups of six back-to-back launches: one group alone mis-ranks tilings that are 2-3 % apart
        // (round 3: the vocabulary product got the K-tile-64 instance on some boxes, 5 us slower than the K-tile-32 one)
        for (int group = 0; group < 3 && rc == OVC_OK; ++group) {
            (void)hipEventRecord(e0, st);
            for (int i = 0; i < 6 && rc == OVC_OK; ++i) rc = ovc_gemm_launch(a, st, opts);
            (void)hipEventRecord(e1, st);
            if (hipEventSynchronize(e1) != hipSuccess) rc = OVC_ELAUNCH;
            float ms = 0.f;
            if (rc == OVC_OK && hipEventElapsedTime(&ms, e0, e1) == hipSuccess && ms < best_ms) { best_ms = ms; best = t; }
        }
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    g_tune_calls.fetch_add(1);
    if (rc != OVC_OK) return rc;
    if (best >= 0) {
        std::lock_guard<std::mutex> lock(g_tuned_mutex);
        g_tuned.push_back(TunedShape{M, seg_n, nseg, K, kchains, ksplit, objective, best});
    }
    return OVC_OK;
}

extern "C" long ovc_gemm_tune_calls(void) { return g_tune_calls.load(); }

// Remembered tiling of a shape: -1 = nothing usable.  near != 0 also accepts the entry of the same product with the
// closest M within a factor of two (what the launch path itself falls back to).
extern "C" int ovc_gemm_tuned_get(int M, int seg_n, int nseg, int K, int kchains, int ksplit, int objective, int near) {
    return tuned_lookup(M, seg_n, nseg, K, class_ok(kchains) ? kchains : 1, ksplit > 1 ? ksplit : 1, objective > 1 ? objective : 1, near != 0);
}

extern "C" int ovc_gemm_tuned_set(int M, int seg_n, int nseg, int K, int kchains, int ksplit, int objective, int tiling) {
    if (objective < 1 || objective > 8) return OVC_EINVAL;
    if (tiling < 0 || tiling >= kNumTilings || !class_ok(kchains) || ksplit < 1 || ksplit > kMaxKSplit) return OVC_EINVAL;
    if (M <= 0 || seg_n <= 0 || nseg <= 0 || nseg > OVC_MAX_SEGMENTS || K <= 0) return OVC_EINVAL;
    GemmArgs a{};
    a.M = M; a.seg_n = seg_n; a.nseg = nseg; a.K1 = K; a.kchains = kchains; a.ksplit = ksplit;
    if (ksplit > 1 && (nseg != 1 || K % (ksplit * 32))) return OVC_EINVAL;
    if (!tiling_fits(a, tiling)) return OVC_EINVAL;           // wrong class, tile straddling a segment, slice not a whole K tile
    const TunedShape entry{M, seg_n, nseg, K, kchains, ksplit, objective, tiling};
    std::lock_guard<std::mutex> lock(g_tuned_mutex);
    for (TunedShape& t : g_tuned)
        if (t.M == M && t.seg_n == seg_n && t.nseg == nseg && t.K == K && t.kchains == kchains && t.ksplit == ksplit && t.objective == objective) { t = entry; return OVC_OK; }
    g_tuned.push_back(entry);
    return OVC_OK;
}

// Tuning helper: `iters` back-to-back launches of one GEMM on `stream` (no host work in between).
extern "C" int ovc_debug_repeat_linear(const float* x, int K, const float* W, const float* bias, float* y, int M, int N,
                                       int iters, ovc_stream stream) {
    GemmArgs a{};
    a.A1 = x; a.lda1 = K; a.K1 = K; a.M = M; a.seg_n = N; a.nseg = 1; a.ldc = N; a.act = 0;
    const int forced = g_forced_tiling.load();
    a.kchains = forced >= 0 ? tiling_chains(forced) : 1;      // follow a forced tiling's class
    a.seg[0] = GemmSegment{W, bias, y, nullptr};
    for (int i = 0; i < iters; ++i) {
        const int rc = ovc_gemm_launch(a, ovc_hip_stream(stream), GemmLaunchOpts{});
        if (rc != OVC_OK) return rc;
    }
    return OVC_OK;
}

namespace {
// W [N, K] fp32 -> 16-bit planes in MFMA-fragment order (gemm_split.h, WD): block j = 32 rows of W, step s = 16 columns;
// lane l of a wave holds row 32 j + (l & 31), columns 16 s + 8 (l >> 5) .. + 7.  Rows past N are zero.
template <int MODE>
__global__ __launch_bounds__(64) void split_weight_kernel(const float* __restrict__ W, int N, int K, u32x4* __restrict__ out) {
    constexpr int P = split_planes(MODE);
    const int lane = threadIdx.x, j = blockIdx.y, s = blockIdx.x, steps = K >> 4;
    const int row = 32 * j + (lane & 31), col = 16 * s + 8 * (lane >> 5);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = row < N ? W[(size_t)row * K + col + e] : 0.f;
    unsigned int pk[4][P];
#pragma unroll
    for (int q = 0; q < 4; ++q) split_pair<MODE>(v[2 * q], v[2 * q + 1], pk[q]);
#pragma unroll
    for (int pl = 0; pl < P; ++pl)
        out[(((size_t)j * steps + s) * P + pl) * 64 + lane] = u32x4{pk[0][pl], pk[1][pl], pk[2][pl], pk[3][pl]};
}
}  // namespace

// Bytes of the planes of a [N, K] weight in split-precision mode `mode` (3 or 4); 0 = invalid.
extern "C" size_t ovc_split_weight_bytes(int N, int K, int mode) {
    if (N <= 0 || K <= 0 || (K & 15) || (mode != 3 && mode != 4)) return 0;
    return (size_t)((N + 31) / 32) * (K >> 4) * split_planes(mode) * 64 * 16;
}

extern "C" int ovc_split_weight(const float* W, int N, int K, int mode, void* planes, ovc_stream stream) {
    if (!W || !planes || !ovc_split_weight_bytes(N, K, mode) || !ovc_aligned16(planes)) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    const dim3 grid(K >> 4, (N + 31) / 32);
    u32x4* out = reinterpret_cast<u32x4*>(planes);
    if (mode == 3) hipLaunchKernelGGL(split_weight_kernel<3>, grid, dim3(64), 0, ovc_hip_stream(stream), W, N, K, out);
    else hipLaunchKernelGGL(split_weight_kernel<4>, grid, dim3(64), 0, ovc_hip_stream(stream), W, N, K, out);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

// ovc_debug_linear_tiling with pre-cut weights (`planes` from ovc_split_weight, mode of the tiling's class).
extern "C" int ovc_debug_linear_planes(const float* x, int K, const float* W, const void* planes, const float* bias, float* y,
                                       int M, int N, int tiling, int ksplit, int iters, ovc_stream stream) {
    if (tiling < 0 || tiling >= kNumTilings || !kTilings[tiling].planes || !x || !W || !planes || !y || iters < 1) return OVC_EINVAL;
    GemmArgs a{};
    a.A1 = x; a.lda1 = K; a.K1 = K; a.M = M; a.seg_n = N; a.nseg = 1; a.ldc = N; a.kchains = tiling_chains(tiling);
    a.seg[0] = GemmSegment{W, ksplit > 1 ? nullptr : bias, y, nullptr, planes};
    if (ksplit > 1) { a.ksplit = ksplit; a.part_stride = (long)M * N; }
    GemmLaunchOpts opts{};
    opts.forced_tiling = tiling;
    for (int i = 0; i < iters; ++i) {
        const int rc = ovc_gemm_launch(a, ovc_hip_stream(stream), opts);
        if (rc != OVC_OK) return rc;
    }
    return OVC_OK;
}

// y = x W^T + bias computed by one named tiling (its class follows from the tiling): the parity tests use it to
// show that all tilings of a class give the same bits.  ksplit > 1: y receives `ksplit` raw partial products
// [ksplit][M][N] (no bias).
extern "C" int ovc_debug_linear_tiling(const float* x, int K, const float* W, const float* bias, float* y, int M, int N,
                                       int tiling, int ksplit, int iters, ovc_stream stream) {
    if (tiling < 0 || tiling >= kNumTilings || !x || !W || !y || iters < 1) return OVC_EINVAL;
    GemmArgs a{};
    a.A1 = x; a.lda1 = K; a.K1 = K; a.M = M; a.seg_n = N; a.nseg = 1; a.ldc = N; a.kchains = tiling_chains(tiling);
    a.seg[0] = GemmSegment{W, ksplit > 1 ? nullptr : bias, y, nullptr};
    if (ksplit > 1) { a.ksplit = ksplit; a.part_stride = (long)M * N; }
    GemmLaunchOpts opts{};
    opts.forced_tiling = tiling;
    for (int i = 0; i < iters; ++i) {                    // back to back, no host work in between (tools/gemm_bench.py)
        const int rc = ovc_gemm_launch(a, ovc_hip_stream(stream), opts);
        if (rc != OVC_OK) return rc;
    }
    return OVC_OK;
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
    a.kchains = 1;                                       // operator level: one chain, whatever the shape
    a.seg[0] = GemmSegment{W, bias, y, nullptr};
    return ovc_gemm_launch(a, ovc_hip_stream(stream), GemmLaunchOpts{});
}
