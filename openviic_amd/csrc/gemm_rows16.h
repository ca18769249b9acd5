// gemm_rows16_f32 -- the four-chain K-order class for products with a handful of rows (round 4; included by gemm.hip).
//
// The reference's own prediction loop decodes ONE image at a time (trainers/base_trainer.py:75-80): M = B * beam = 5 rows per
// decode-step product, 40 at B = 8.  The 32 x 32 instances of gemm_f32_mfma spend 6.6 us on such a launch whatever the tiling
//  while a dependent launch that reads its predecessor's output and writes costs 1.75 us
// (tools/phase_floor_probe.hip): the kernel's own latency -- a cooperative tile load, two barriers per K tile, a 64-MFMA chain of
// v_mfma_f32_32x32x2_f32 per wave of which 27 rows in 32 are padding -- is what a 5-row product pays for.  This instance:
//
//   * 16-row tiles on v_mfma_f32_16x16x4_f32: the SAME dependent fma chain per output element (k ascending, four k per
//     instruction instead of two), so the bits are those of every other instance of the class, in a quarter of the matrix time
//     (32 cycles per four k against 2 x 64);
//   * one wave per chain, as in the 32 x 32 instances (chain c = the 8-deep k groups g with g % 4 == c, summed ((c0+c1)+c2)+c3
//     through LDS at the end), but each wave fetches ITS OWN quarter of A and W -- the k groups of its chain -- with 16-byte loads
//     issued all at once, turns them into MFMA operand order through a wave-private LDS region (a lane group kq supplies ONE k of
//     every instruction: a transpose of what 16-byte loads deliver) and runs its whole chain: ONE global round trip and no workgroup barrier
//     before the chain reduction, whatever K is (512 k per pass);
//   * bias / ReLU / raw K-slice partials / N segments as in gemm_f32_mfma's epilogue; no residual, no second input block, no
//     statistics (tiling_fits keeps such products on the 32 x 32 instances).
//
// NB = 16-column blocks per workgroup (the workgroup's tile is 16 x 16 NB).
#pragma once

constexpr int kRows16Ldt = 132;          // LDS row stride (floats): 128 k of one chain + 4 (operand reads: two-way bank conflicts at most)
constexpr int kRows16Chunk = 512;        // k per pass: 16 periods of four 8-deep groups, one group per chain

// Grid: x = column tile over all segments, y = row tile (times the tuner's co-running copies), z = K slice.  The LDS regions allow
// one workgroup (NB = 2) or two (NB = 1) per CU, so a wave may hold every load of a pass in registers: amdgpu_waves_per_eu says so
// (without it hipcc budgets for eight waves per SIMD and serialises the loads behind 78 registers).
template <int NB>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) void gemm_rows16_f32(GemmArgs p, int tiles_m, int tiles_n_per_seg, int kslice) {
    constexpr int BN = 16 * NB;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, chain = threadIdx.x >> 6;
    float* a_lds = lds + (size_t)chain * (16 + BN) * kRows16Ldt;        // this wave's private region
    float* w_lds = a_lds + 16 * kRows16Ldt;

    // every scalar the loads and the epilogue need, fetched in one round trip (gemm.hip: the prologue of gemm_f32_mfma)
    const float* a_ptr = p.A1;
    const int lda = p.lda1, M = p.M, seg_n = p.seg_n, K = p.K1, ldc = p.ldc, act = p.act, nseg = p.nseg;
    const long part_stride = p.part_stride;
    const float* w0_ptr = p.seg[0].W;
    const float* bias0_ptr = p.seg[0].bias;
    float* c0_ptr = p.seg[0].C;
    asm volatile("" :: "s"(a_ptr), "s"(lda), "s"(M), "s"(seg_n), "s"(K), "s"(ldc), "s"(act), "s"(nseg), "s"(part_stride), "s"(w0_ptr),
                 "s"(bias0_ptr), "s"(c0_ptr), "s"(tiles_m), "s"(tiles_n_per_seg), "s"(kslice));

    int tile_m = blockIdx.y, tile_n = blockIdx.x, seg = 0;
    while (tile_m >= tiles_m) tile_m -= tiles_m;                       // a co-running copy (tuner only)
    while (tile_n >= tiles_n_per_seg) { tile_n -= tiles_n_per_seg; ++seg; }    // at most OVC_MAX_SEGMENTS - 1 steps: no division
    const int m0 = tile_m * 16, n0 = tile_n * BN;
    const int kbase = (int)blockIdx.z * kslice, kend = kbase + kslice;
    const bool partial = gridDim.z > 1;
    const float* W = nseg == 1 ? w0_ptr : p.seg[seg].W;
    const float* bias = partial ? nullptr : (nseg == 1 ? bias0_ptr : p.seg[seg].bias);
    float* C = partial ? c0_ptr + (size_t)blockIdx.z * part_stride : (nseg == 1 ? c0_ptr : p.seg[seg].C);

    const int r = lane & 15, kq = lane >> 4;
    // the bias of this lane's columns: asked for now, used after the chain reduction (an always-valid address when there is none)
    float bias_v[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) bias_v[nb] = (bias ? bias : W)[min(n0 + 16 * nb + r, seg_n - 1)];

    // Operand loads are raw buffer loads: rows past M / columns past seg_n fall outside the descriptors and read as zero, and a
    // piece past the end of the K range gets an offset outside every descriptor -- nothing to clamp, nothing to select.
    // Slot q of a lane is piece (lane & 31) of row 2 q + (lane >> 5); piece i = floats 4 i .. 4 i + 3 of the chain's k sequence,
    // i.e. k = k0 + 32 (i >> 1) + 8 chain + 4 (i & 1).
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a_ptr), 0, M * lda * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, seg_n * K * 4, 0x00020000);
    const int piece = lane & 31, prow = lane >> 5;
    const int koff = 32 * (piece >> 1) + 8 * chain + 4 * (piece & 1);
    const int voff_a = ((m0 + prow) * lda + koff) * 4, voff_w = ((n0 + prow) * K + koff) * 4;
    float* a_dst = a_lds + prow * kRows16Ldt + 4 * piece;
    float* w_dst = w_lds + prow * kRows16Ldt + 4 * piece;
    // The class's order INSIDE an 8-deep group is k = 8 g + {0, 4, 1, 5, 2, 6, 3, 7} (gemm.hip: a 32x32x2 instruction takes float e
    // of the lower half-wave's 16 bytes and float e of the upper one's): the group's first 16x16x4 instruction gets offsets
    // {0, 4, 1, 5} from its lane groups kq = 0..3, the second {2, 6, 3, 7} -- lane (r, kq) reads 4 (kq & 1) + (kq >> 1) and the
    // float two further on (one ds_read2_b32).
    const float* ap = a_lds + r * kRows16Ldt + 4 * (kq & 1) + (kq >> 1);
    const float* wp = w_lds + r * kRows16Ldt + 4 * (kq & 1) + (kq >> 1);

    f32x4 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = kbase; k0 < kend; k0 += kRows16Chunk) {
        const int outside = (int)0x80000000;                 // beyond any descriptor (their ranges are below 2 GB)
        const bool k_ok = k0 + koff < kend;
        const int va = k_ok ? voff_a : outside, vw = k_ok ? voff_w : outside;
        f32x4 av[8], wv[8 * NB];
#pragma unroll
        for (int q = 0; q < 8; ++q)                          // every load first: one round trip
            av[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, va, (2 * q * lda + k0) * 4, 0));
#pragma unroll
        for (int q = 0; q < 8 * NB; ++q)
            wv[q] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, vw, (2 * q * K + k0) * 4, 0));
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<f32x4*>(a_dst + 2 * q * kRows16Ldt) = av[q];
#pragma unroll
        for (int q = 0; q < 8 * NB; ++q) *reinterpret_cast<f32x4*>(w_dst + 2 * q * kRows16Ldt) = wv[q];
        // the region is this wave's own and a wave's LDS instructions execute in order: no workgroup barrier, only keep the
        // compiler from moving the operand reads above the writes
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

        // all 16 groups of the pass are in LDS (zeros past the end of the range).  Blocks of four groups, software-pipelined by
        // hand: the operand reads of block b + 1 are issued before the matrix instructions of block b (sched_barrier keeps hipcc
        // from sinking them back next to their users, where every group would wait out an LDS round trip); blocks wholly past
        // the range are skipped.
        const int groups = (min(kRows16Chunk, kend - k0) + 31) / 32;       // this chain's 8-deep groups in the pass
        f32x2 a2[2][4], b2[2][NB][4];
        auto read_block = [&](int blk, int buf) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int g = 4 * blk + u;
                a2[buf][u] = f32x2{ap[8 * g], ap[8 * g + 2]};
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) b2[buf][nb][u] = f32x2{wp[nb * 16 * kRows16Ldt + 8 * g], wp[nb * 16 * kRows16Ldt + 8 * g + 2]};
            }
        };
        read_block(0, 0);
#pragma unroll
        for (int blk = 0; blk < 4; ++blk) {
            if (4 * blk >= groups) break;                    // wave-uniform
            if (blk < 3) read_block(blk + 1, (blk + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[blk & 1][u][0], b2[blk & 1][nb][u][0], acc[nb], 0, 0, 0);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_16x16x4f32(a2[blk & 1][u][1], b2[blk & 1][nb][u][1], acc[nb], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the next pass overwrites what these reads used
        __builtin_amdgcn_wave_barrier();
    }

    // ---- chains summed in chain order through LDS: ((c0 + c1) + c2) + c3 ---------------------------------------------------
    float* red = lds;                                        // [3][NB][64] float4, over chain 0's region (its reads are done: barrier)
    __syncthreads();
    if (chain > 0) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) *reinterpret_cast<f32x4*>(red + ((size_t)((chain - 1) * NB + nb) * 64 + lane) * 4) = acc[nb];
    }
    __syncthreads();
    if (chain > 0) return;
#pragma unroll
    for (int c = 1; c < 4; ++c)
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) acc[nb] += *reinterpret_cast<const f32x4*>(red + ((size_t)((c - 1) * NB + nb) * 64 + lane) * 4);

    // ---- epilogue: lane (col = lane & 15, rows 4 (lane >> 4) + v) ------------------------------------------------------------
    const bool relu = !partial && act == 1;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int col = n0 + 16 * nb + r;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            const int row = m0 + 4 * kq + v;
            float y = acc[nb][v];
            if (bias) y += bias_v[nb];
            if (relu) y = fmaxf(y, 0.f);
            if (col < seg_n && row < M) C[(size_t)row * ldc + col] = y;
        }
    }
}
