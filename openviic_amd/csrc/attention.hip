// Attention kernels.
//
//  attention_mfma_kernel   general scaled-dot-product attention on projected heads, used by the
//                          encoder (nq = nk = regions, optional geometry bias / memory slots), the
//                          teacher-forced decoder and the DLCT-style cross form (nq != nk, per-query
//                          mask).  Q.K^T and P.V run on the matrix cores (v_mfma_f32_32x32x2_f32),
//                          the softmax row reduce on wavefront shuffles.
//  decode_self_attention   one query row per beam against its own history through the ancestor
//                          table (no cache re-ordering), nq = 1.
//  decode_cross_attention  the k beams of one image share that image's projected encoder K/V: one wave per
//                          (image, head), everything in registers, both contractions on v_mfma_f32_16x16x4_f32
//                          (d_k in {16, 32, 64}); head sizes 4 and 8 take an LDS-staged VALU kernel.
//
// Reference call sites: models/modules/attentions.py:51-55, :102-111, :171-183.
#include <cstdlib>
#include <mutex>

#include "common.h"

namespace {

constexpr int kQTile = 64;        // query rows per workgroup
constexpr int kLdQK = 68;         // LDS row stride (floats) of the Q / K / V images: 64 + 4
                                  // (68 r mod 64 = 4 r: conflict-free ds_read_b128 per lane group)

struct AttnArgs {
    const float* q; const float* k; const float* v; float* out;
    int b, nq, nk, h, dk, dv;
    const uint8_t* mask; long mask_sb, mask_sq;
    const float* geometry;
    const float* mem_k; const float* mem_v; int m; float mem_scale_k, mem_scale_v;
    int nkp;                       // nk + m rounded up to a multiple of 32
    int qtiles;
};

__global__ __launch_bounds__(256) void attention_mfma_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = blockIdx.x % p.qtiles;
    const int hd = (blockIdx.x / p.qtiles) % p.h;
    const int b = blockIdx.x / (p.qtiles * p.h);
    const int q0 = qt * kQTile;
    const int nkt = p.nk + p.m;            // real keys + memory slots
    const int lds_s = p.nkp + 4;           // row stride of the score / probability image

    float* Qs = lds;                       // [64][68]
    float* Ks = Qs + kQTile * kLdQK;       // [nkp][68]
    float* Vs = Ks + p.nkp * kLdQK;        // [nkp][68]
    float* Ss = Vs + p.nkp * kLdQK;        // [64][nkp+4]

    // ---- stage Q, K, V (zero-filled outside the valid region) -------------------------------
    {
        const int c4 = tid & 15, r0 = tid >> 4;           // 16 float4 per 64-wide row, 16 rows per pass
        const int col = c4 * 4;
        for (int r = r0; r < kQTile; r += 16) {
            f32x4 val = {0.f, 0.f, 0.f, 0.f};
            if (q0 + r < p.nq && col < p.dk)
                val = *reinterpret_cast<const f32x4*>(p.q + ((size_t)b * p.nq + q0 + r) * (p.h * p.dk) + hd * p.dk + col);
            *reinterpret_cast<f32x4*>(Qs + r * kLdQK + col) = val;
        }
        for (int r = r0; r < p.nkp; r += 16) {
            f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
            if (r < p.nk) {
                if (col < p.dk) kv = *reinterpret_cast<const f32x4*>(p.k + ((size_t)b * p.nk + r) * (p.h * p.dk) + hd * p.dk + col);
                if (col < p.dv) vv = *reinterpret_cast<const f32x4*>(p.v + ((size_t)b * p.nk + r) * (p.h * p.dv) + hd * p.dv + col);
            } else if (r < nkt) {
                const int mr = r - p.nk;
                if (col < p.dk) kv = *reinterpret_cast<const f32x4*>(p.mem_k + (size_t)mr * (p.h * p.dk) + hd * p.dk + col) * p.mem_scale_k;
                if (col < p.dv) vv = *reinterpret_cast<const f32x4*>(p.mem_v + (size_t)mr * (p.h * p.dv) + hd * p.dv + col) * p.mem_scale_v;
            }
            *reinterpret_cast<f32x4*>(Ks + r * kLdQK + col) = kv;
            *reinterpret_cast<f32x4*>(Vs + r * kLdQK + col) = vv;
        }
    }
    __syncthreads();

    const int frow = lane & 31, half = lane >> 5;
    // ---- S = Q K^T / sqrt(dk), masked, geometry-biased -> LDS ----------------------------------
    {
        const int ktiles = p.nkp >> 5;
        const int ksteps = (p.dk + 7) >> 3;
        const float inv_scale = sqrtf((float)p.dk);
        for (int tile = wave; tile < 2 * ktiles; tile += 4) {
            const int tq = tile / ktiles, tk = tile - tq * ktiles;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = Qs + (tq * 32 + frow) * kLdQK + half * 4;
            const float* bp = Ks + (tk * 32 + frow) * kLdQK + half * 4;
            for (int kk = 0; kk < ksteps; ++kk) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kk * 8);
                const f32x4 bb = *reinterpret_cast<const f32x4*>(bp + kk * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bb[s], acc, 0, 0, 0);
            }
            const int kj = tk * 32 + frow;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int qi = tq * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                const int gq = q0 + qi;
                float s = acc[r] / inv_scale;
                if (kj >= nkt) {
                    s = -INFINITY;
                } else if (kj < p.nk && gq < p.nq) {
                    if (p.mask && p.mask[(size_t)b * p.mask_sb + (size_t)gq * p.mask_sq + kj]) s = -INFINITY;
                    if (p.geometry)
                        s = logf(fmaxf(p.geometry[(((size_t)b * p.h + hd) * p.nq + gq) * p.nk + kj], 1e-6f)) + s;
                }
                Ss[qi * lds_s + kj] = s;
            }
        }
    }
    __syncthreads();

    // ---- row softmax: 4 lanes per row, wavefront-shuffle reduce ------------------------------------
    {
        const int row = tid >> 2, part = tid & 3;
        float* srow = Ss + row * lds_s;
        float mx = -INFINITY;
        for (int j = part; j < p.nkp; j += 4) mx = fmaxf(mx, srow[j]);
        mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
        float sum = 0.f;
        for (int j = part; j < p.nkp; j += 4) {
            const float e = expf(srow[j] - mx);
            srow[j] = e;
            sum += e;
        }
        sum += __shfl_xor(sum, 1, 64);
        sum += __shfl_xor(sum, 2, 64);
        for (int j = part; j < p.nkp; j += 4) srow[j] = srow[j] / sum;
    }
    __syncthreads();

    // ---- O = P V ------------------------------------------------------------------------------------------
    {
        const int vtiles = (p.dv + 31) >> 5;
        const int ksteps = p.nkp >> 3;
        for (int tile = wave; tile < 2 * vtiles; tile += 4) {
            const int tq = tile / vtiles, tn = tile - tq * vtiles;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const float* ap = Ss + (tq * 32 + frow) * lds_s + half * 4;
            const float* bp = Vs + (half * 4) * kLdQK + tn * 32 + frow;
            for (int kk = 0; kk < ksteps; ++kk) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(ap + kk * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s], bp[(kk * 8 + s) * kLdQK], acc, 0, 0, 0);
            }
            const int col = tn * 32 + frow;
            if (col < p.dv) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int gq = q0 + tq * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (gq < p.nq) p.out[((size_t)b * p.nq + gq) * (p.h * p.dv) + hd * p.dv + col] = acc[r];
                }
            }
        }
    }
}


// -------------------------------------------------------------------------------------------------
// attention_regs_kernel: the same operator with the scores kept in accumulator registers.
//
// One workgroup per (image, head); wave w owns queries 32 w .. 32 w + 31 and ALL keys.  K and V of the (image,
// head) go to LDS once with fully coalesced 256-byte row reads and are shared by the waves; each wave's Q fragment
// comes straight from global memory into registers.
//   S^T[key][q] = K Q^T  (v_mfma_f32_32x32x2_f32, A = K rows from LDS, B = Q rows): the accumulator of key tile tk
//       has the QUERY on the lane (q = lane & 31) and 16 keys on its registers (key = 32 tk + (r&3) + 8 (r>>2) +
//       4 (lane>>5)), so scale, mask, geometry bias and the softmax over keys are register arithmetic plus ONE
//       __shfl_xor(32) per reduction -- no score image in LDS, no scalar LDS stores (the general kernel above
//       spends most of its time there: 0.13 of the MFMA peak).
//   O^T[dv][q] = V^T P^T: the probabilities are used where they are, as the B operand: MFMA step (tk, r) contracts
//       over the two keys base_r and base_r + 4 that the two lane halves hold in register r; the A operand
//       V[key][dv = lane & 31] is a conflict-free 128-byte ds_read_b32 per half.
//   The output tile goes through LDS once (the K image is free by then) so that rows leave as whole 256-byte lines.
// NKT = key tiles of 32 (real keys + memory slots, <= 6); KG = 8-deep d groups of Q.K (dk <= 8 KG);
// DVT = 32-wide tiles of d_v.  Numerics: scores and probabilities as in the general kernel (same scale, mask,
// geometry order, expf, division); only the order of the softmax sum and of the P.V sum over keys differs.
// -------------------------------------------------------------------------------------------------
template <int NKT, int KG, int DVT>
__global__ __launch_bounds__(256) void attention_regs_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hd = blockIdx.x % p.h, b = blockIdx.x / p.h;
    const int nkt = p.nk + p.m;
    constexpr int kRows = NKT * 32;
    const int k_rows = max(kRows, ((p.nq + 31) >> 5) * 32);           // the K image doubles as the output staging area
    float* Ks = lds;                          // [k_rows][68]
    float* Vs = Ks + k_rows * kLdQK;          // [kRows][68]

    // ---- K, V -> LDS (zero-filled outside the valid region), Q fragment -> registers ------------------------------
    // All 256 threads stage (waves beyond the query tiles only help here), and all of a thread's global loads are
    // issued before the first LDS store: a load -> store loop would pay one memory round trip per pass.
    {
        const int c4 = tid & 15, col = c4 * 4, r0 = tid >> 4;
        constexpr int kPasses = kRows / 16;
        f32x4 kv[kPasses], vv[kPasses];
#pragma unroll
        for (int i = 0; i < kPasses; ++i) {
            const int r = r0 + 16 * i;
            const bool real = r < p.nk, slot = !real && r < nkt;
            const int mr = min(max(r - p.nk, 0), max(p.m - 1, 0));
            // one address per matrix, always valid: a real key row, a memory slot, or (padding) row 0 -- zeroed below
            const float* ks = slot ? p.mem_k + (size_t)mr * (p.h * p.dk) + hd * p.dk
                                   : p.k + ((size_t)b * p.nk + (real ? r : 0)) * (p.h * p.dk) + hd * p.dk;
            const float* vs = slot ? p.mem_v + (size_t)mr * (p.h * p.dv) + hd * p.dv
                                   : p.v + ((size_t)b * p.nk + (real ? r : 0)) * (p.h * p.dv) + hd * p.dv;
            kv[i] = *reinterpret_cast<const f32x4*>(ks + min(col, p.dk - 4));
            vv[i] = *reinterpret_cast<const f32x4*>(vs + min(col, p.dv - 4));
        }
#pragma unroll
        for (int i = 0; i < kPasses; ++i) {
            const int r = r0 + 16 * i;
            const bool real = r < p.nk, slot = !real && r < nkt;
            f32x4 kx = kv[i], vx = vv[i];
            if (slot) { kx = kx * p.mem_scale_k; vx = vx * p.mem_scale_v; }
            if ((!real && !slot) || col >= p.dk) kx = f32x4{0.f, 0.f, 0.f, 0.f};
            if ((!real && !slot) || col >= p.dv) vx = f32x4{0.f, 0.f, 0.f, 0.f};
            *reinterpret_cast<f32x4*>(Ks + r * kLdQK + col) = kx;
            *reinterpret_cast<f32x4*>(Vs + r * kLdQK + col) = vx;
        }
    }
    const int qwaves = (p.nq + 31) >> 5;                        // waves that own queries; the others only staged
    const int qi = lane & 31, half = lane >> 5;
    const int gq = wave * 32 + qi;                              // this lane's query
    const bool q_ok = gq < p.nq;
    f32x4 qf[KG];
    {
        const float* qrow = p.q + ((size_t)b * p.nq + min(gq, p.nq - 1)) * (p.h * p.dk) + hd * p.dk;
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
            const int d = 8 * kk + 4 * half;
            qf[kk] = *reinterpret_cast<const f32x4*>(qrow + min(d, p.dk - 4));
            if (!q_ok || d >= p.dk) qf[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    __syncthreads();

    // ---- S^T = K Q^T ---------------------------------------------------------------------------------------------------
    f32x16 st[NKT];
#pragma unroll
    for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[tk][r] = 0.f;
    if (wave < qwaves) {
#pragma unroll
    for (int kk = 0; kk < KG; ++kk) {
        f32x4 kf[NKT];
#pragma unroll
        for (int tk = 0; tk < NKT; ++tk) kf[tk] = *reinterpret_cast<const f32x4*>(Ks + (tk * 32 + qi) * kLdQK + 8 * kk + 4 * half);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tk = 0; tk < NKT; ++tk) st[tk] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tk][s], qf[kk][s], st[tk], 0, 0, 0);
    }
    }
    __syncthreads();                                            // every wave is done with the K image
    if (wave >= qwaves) return;

    // ---- scale, mask, geometry bias; softmax over the keys of this lane's query -------------------------------------------
    const float inv_scale = sqrtf((float)p.dk);
    const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.mask_sb + (size_t)min(gq, p.nq - 1) * p.mask_sq : nullptr;
    const float* grow = p.geometry ? p.geometry + (((size_t)b * p.h + hd) * p.nq + min(gq, p.nq - 1)) * p.nk : nullptr;
    float mx = -INFINITY;
#pragma unroll
    for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kj = tk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float s = st[tk][r] / inv_scale;
            if (kj >= nkt) {
                s = -INFINITY;
            } else if (kj < p.nk) {
                if (mrow && mrow[kj]) s = -INFINITY;
                if (grow) s = logf(fmaxf(grow[kj], 1e-6f)) + s;
            }
            st[tk][r] = s;
            mx = fmaxf(mx, s);
        }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = expf(st[tk][r] - mx);
            st[tk][r] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 32, 64);
#pragma unroll
    for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) st[tk][r] = st[tk][r] / sum;

    // ---- O^T = V^T P^T ---------------------------------------------------------------------------------------------------
    f32x16 ot[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;
#pragma unroll
    for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float* vrow = Vs + (tk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * kLdQK + qi;     // this half's key of step (tk, r)
#pragma unroll
            for (int t = 0; t < DVT; ++t) ot[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[t * 32], st[tk][r], ot[t], 0, 0, 0);
        }

    // ---- output: accumulator (dv on registers, query on the lane) -> LDS [query][dv] -> whole rows to memory ---------------
    float* Os = Ks + wave * 32 * kLdQK;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(Os + qi * kLdQK + t * 32 + 8 * j + 4 * half) =
                f32x4{ot[t][4 * j], ot[t][4 * j + 1], ot[t][4 * j + 2], ot[t][4 * j + 3]};
    // (each wave reads back only what it wrote itself: no workgroup barrier, the compiler's lgkmcnt wait orders it)
    {
        const int c4 = lane & 15, col = c4 * 4;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 4 + (lane >> 4);
            const int oq = wave * 32 + row;
            if (oq < p.nq && col < p.dv)
                *reinterpret_cast<f32x4*>(p.out + ((size_t)b * p.nq + oq) * (p.h * p.dv) + hd * p.dv + col) =
                    *reinterpret_cast<const f32x4*>(Os + row * kLdQK + col);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// attention_tiled_kernel: the same operator without a limit on the number of keys or queries (round 4; shapes beyond the
// register instances: more than 192 keys, or more than 128 keys for more than 128 queries).  The reference has
// none (attentions.py:44-58 works on any nk, :158-185 appends its 40 memory slots to any nk): bottom-up feature sets carry up
// to 100 regions per image (+ 40 slots = 140 keys), grid features 14 x 14 = 196 cells, the DLCT form regions + cells.
//
// One workgroup per (image, head, 128 queries), a wave per 32 queries -- the register layout of attention_regs_kernel
// (S^T = K Q^T with the query on the lane, O^T = V^T P^T with the probabilities as the B operand) -- but the keys pass
// through LDS in tiles of 128, in ascending order, under an online softmax: per query a running maximum M and a running sum
// L of exp(s - M); a tile whose maximum raises M rescales L and the output accumulators by exp(M_old - M_new), which is a
// per-lane scalar here because a lane's accumulator registers all belong to ITS query.  The tile order is fixed, so a result
// depends on nothing but the operands (no timing, no batch size).  Fully masked rows end with L = 0 and give 0 / 0 = NaN,
// as the reference's softmax over a row of -inf does.  Numerics vs the kernels above: the division by the softmax sum
// happens once at the end instead of per probability (~1 ulp per output); shapes with nk + m <= 192 and nq <= 128 never come
// here (they have register instances), so nothing that ran before round 4 changes a bit.
// KG = 8-deep d groups of Q.K (dk <= 8 KG); DVT = 32-wide tiles of d_v.
// -------------------------------------------------------------------------------------------------
template <int KG, int DVT>
__global__ __launch_bounds__(256) void attention_tiled_kernel(AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NKT = 4, kRows = NKT * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qt = blockIdx.x % p.qtiles;
    const int hd = (blockIdx.x / p.qtiles) % p.h;
    const int b = blockIdx.x / (p.qtiles * p.h);
    const int nkt = p.nk + p.m;
    float* Ks = lds;                          // [128][68]; doubles as the output staging area at the end
    float* Vs = Ks + kRows * kLdQK;           // [128][68]

    const int qi = lane & 31, half = lane >> 5;
    const int gq = qt * kRows + wave * 32 + qi;                 // this lane's query
    const bool q_ok = gq < p.nq;
    const bool wave_live = qt * kRows + wave * 32 < p.nq;       // wave-uniform: waves past the last query only stage
    f32x4 qf[KG];
    {
        const float* qrow = p.q + ((size_t)b * p.nq + min(gq, p.nq - 1)) * (p.h * p.dk) + hd * p.dk;
#pragma unroll
        for (int kk = 0; kk < KG; ++kk) {
            const int d = 8 * kk + 4 * half;
            qf[kk] = *reinterpret_cast<const f32x4*>(qrow + min(d, p.dk - 4));
            if (!q_ok || d >= p.dk) qf[kk] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const float inv_scale = sqrtf((float)p.dk);
    const uint8_t* mrow = p.mask ? p.mask + (size_t)b * p.mask_sb + (size_t)min(gq, p.nq - 1) * p.mask_sq : nullptr;
    const float* grow = p.geometry ? p.geometry + (((size_t)b * p.h + hd) * p.nq + min(gq, p.nq - 1)) * p.nk : nullptr;

    float m_run = -INFINITY, l_run = 0.f;
    f32x16 ot[DVT];
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[t][r] = 0.f;

    for (int k0 = 0; k0 < nkt; k0 += kRows) {
        // ---- this tile's K, V rows -> LDS (zero-filled past the last key); every load issued before the first LDS store ------
        {
            const int c4 = tid & 15, col = c4 * 4, r0 = tid >> 4;
            constexpr int kPasses = kRows / 16;
            f32x4 kv[kPasses], vv[kPasses];
#pragma unroll
            for (int i = 0; i < kPasses; ++i) {
                const int kr = k0 + r0 + 16 * i;
                const bool real = kr < p.nk, slot = !real && kr < nkt;
                const int mr = min(max(kr - p.nk, 0), max(p.m - 1, 0));
                const float* ks = slot ? p.mem_k + (size_t)mr * (p.h * p.dk) + hd * p.dk
                                       : p.k + ((size_t)b * p.nk + (real ? kr : 0)) * (p.h * p.dk) + hd * p.dk;
                const float* vs = slot ? p.mem_v + (size_t)mr * (p.h * p.dv) + hd * p.dv
                                       : p.v + ((size_t)b * p.nk + (real ? kr : 0)) * (p.h * p.dv) + hd * p.dv;
                kv[i] = *reinterpret_cast<const f32x4*>(ks + min(col, p.dk - 4));
                vv[i] = *reinterpret_cast<const f32x4*>(vs + min(col, p.dv - 4));
            }
#pragma unroll
            for (int i = 0; i < kPasses; ++i) {
                const int r = r0 + 16 * i, kr = k0 + r;
                const bool real = kr < p.nk, slot = !real && kr < nkt;
                f32x4 kx = kv[i], vx = vv[i];
                if (slot) { kx = kx * p.mem_scale_k; vx = vx * p.mem_scale_v; }
                if ((!real && !slot) || col >= p.dk) kx = f32x4{0.f, 0.f, 0.f, 0.f};
                if ((!real && !slot) || col >= p.dv) vx = f32x4{0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4*>(Ks + r * kLdQK + col) = kx;
                *reinterpret_cast<f32x4*>(Vs + r * kLdQK + col) = vx;
            }
        }
        __syncthreads();

        if (wave_live) {
            // ---- S^T = K Q^T for the tile's 128 keys ---------------------------------------------------------------------------
            f32x16 st[NKT];
#pragma unroll
            for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[tk][r] = 0.f;
#pragma unroll
            for (int kk = 0; kk < KG; ++kk) {
                f32x4 kf[NKT];
#pragma unroll
                for (int tk = 0; tk < NKT; ++tk) kf[tk] = *reinterpret_cast<const f32x4*>(Ks + (tk * 32 + qi) * kLdQK + 8 * kk + 4 * half);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tk = 0; tk < NKT; ++tk) st[tk] = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[tk][s], qf[kk][s], st[tk], 0, 0, 0);
            }
            // ---- scale, mask, geometry bias; the tile's maximum for this lane's query ----------------------------------------
            float mx = -INFINITY;
#pragma unroll
            for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kj = k0 + tk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    float s = st[tk][r] / inv_scale;
                    if (kj >= nkt) {
                        s = -INFINITY;
                    } else if (kj < p.nk) {
                        if (mrow && mrow[kj]) s = -INFINITY;
                        if (grow) s = logf(fmaxf(grow[kj], 1e-6f)) + s;
                    }
                    st[tk][r] = s;
                    mx = fmaxf(mx, s);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const bool none = m_new == -INFINITY;                    // every key so far masked: nothing to add, nothing to rescale
            const float alpha = none ? 1.f : expf(m_run - m_new);    // m_run = -inf, m_new finite: 0 (L and O are 0 anyway)
            float sum = 0.f;
#pragma unroll
            for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float e = none ? 0.f : expf(st[tk][r] - m_new);
                    st[tk][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 32, 64);
            l_run = l_run * alpha + sum;
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < DVT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) ot[t][r] *= alpha;
            // ---- O^T += V^T E^T -------------------------------------------------------------------------------------------------
#pragma unroll
            for (int tk = 0; tk < NKT; ++tk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float* vrow = Vs + (tk * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * kLdQK + qi;
#pragma unroll
                    for (int t = 0; t < DVT; ++t) ot[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(vrow[t * 32], st[tk][r], ot[t], 0, 0, 0);
                }
        }
        __syncthreads();                                             // every wave is done with this tile's images
    }
    if (!wave_live) return;

    // ---- normalise; accumulator (dv on registers, query on the lane) -> LDS [query][dv] -> whole rows to memory -----------------
    float* Os = Ks + wave * 32 * kLdQK;
#pragma unroll
    for (int t = 0; t < DVT; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(Os + qi * kLdQK + t * 32 + 8 * j + 4 * half) =
                f32x4{ot[t][4 * j] / l_run, ot[t][4 * j + 1] / l_run, ot[t][4 * j + 2] / l_run, ot[t][4 * j + 3] / l_run};
    {
        const int c4 = lane & 15, col = c4 * 4;
#pragma unroll
        for (int pass = 0; pass < 8; ++pass) {
            const int row = pass * 4 + (lane >> 4);
            const int oq = qt * kRows + wave * 32 + row;
            if (oq < p.nq && col < p.dv)
                *reinterpret_cast<f32x4*>(p.out + ((size_t)b * p.nq + oq) * (p.h * p.dv) + hd * p.dv + col) =
                    *reinterpret_cast<const f32x4*>(Os + row * kLdQK + col);
        }
    }
}

}  // namespace

extern "C" int ovc_attention(const float* q, const float* k, const float* v, int b, int nq, int nk, int h,
                             int dk, int dv, const uint8_t* mask, long mask_sb, long mask_sq,
                             const float* geometry, const float* mem_k, const float* mem_v, int m,
                             float mem_scale_k, float mem_scale_v, float* out, ovc_stream stream) {
    if (!q || !k || !v || !out || b <= 0 || nq <= 0 || nk <= 0 || h <= 0) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;      // kernel attributes below are raised once per process
    if (dk <= 0 || dv <= 0 || (dk & 3) || (dv & 3) || dk > 64 || dv > 64) return OVC_EINVAL;
    if (m < 0 || (m > 0 && (!mem_k || !mem_v))) return OVC_EINVAL;
    if (!ovc_aligned16(q) || !ovc_aligned16(k) || !ovc_aligned16(v)) return OVC_EINVAL;
    if (m > 0 && (!ovc_aligned16(mem_k) || !ovc_aligned16(mem_v))) return OVC_EINVAL;
    AttnArgs p{};
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.b = b; p.nq = nq; p.nk = nk; p.h = h; p.dk = dk; p.dv = dv;
    p.mask = mask; p.mask_sb = mask_sb; p.mask_sq = mask_sq;
    p.geometry = geometry;
    p.mem_k = mem_k; p.mem_v = mem_v; p.m = m; p.mem_scale_k = mem_scale_k; p.mem_scale_v = mem_scale_v;
    p.nkp = ((nk + m + 31) / 32) * 32;
    p.qtiles = (nq + kQTile - 1) / kQTile;
    // Register-resident kernel: one workgroup per (image, head), a wave per 32 queries (nq <= 128), scores never leave
    // the accumulators.  Head sizes up to 64, key tiles up to 4 x 32: everything the path uses.
    {
        const int nkt = (nk + m + 31) / 32, waves = (nq + 31) / 32, hmax = dk > dv ? dk : dv;
        // up to 192 keys (six 32-key tiles: 128 regions + 40 memory slots and a bit) for up to 128 queries; instances 5 and 6 exist
        // since round 4 so that the shipped meshed-memory configuration (MEMORY: 40) stays on this kernel for every N <= 128
        if (nkt <= 6 && waves <= 4 && !OVC_HOOK_ENV("OVC_ATTENTION_GENERAL")) {
            const int k_rows = nkt * 32 > waves * 32 ? nkt * 32 : waves * 32;
            const size_t bytes = sizeof(float) * (size_t)(k_rows + nkt * 32) * kLdQK;
            const dim3 grid(b * h), block(256);
#define OVC_ATT(NKT, KG, DVT)                                                                                         \
    do {                                                                                                              \
        static std::once_flag once;                                                                                   \
        std::call_once(once, [] {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_regs_kernel<NKT, KG, DVT>),            \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 112 * 1024);                        \
        });                                                                                                           \
        hipLaunchKernelGGL((attention_regs_kernel<NKT, KG, DVT>), grid, block, bytes, ovc_hip_stream(stream), p);     \
    } while (0)
#define OVC_ATT_H(NKT)                                                                                                \
    do {                                                                                                              \
        if (hmax <= 16) OVC_ATT(NKT, 2, 1); else if (hmax <= 32) OVC_ATT(NKT, 4, 1); else OVC_ATT(NKT, 8, 2);          \
    } while (0)
            if (nkt == 1) OVC_ATT_H(1); else if (nkt == 2) OVC_ATT_H(2); else if (nkt == 3) OVC_ATT_H(3); else if (nkt == 4) OVC_ATT_H(4);
            else if (nkt == 5) OVC_ATT_H(5); else OVC_ATT_H(6);
#undef OVC_ATT_H
#undef OVC_ATT
            OVC_RETURN_IF_LAUNCH_FAILED();
            return OVC_OK;
        }
    }
    // More than 192 keys (real + memory slots), or more than 128 keys for more than 128 queries: the key-tiled kernel with an
    // online softmax, any nq and nk.
    if (nk + m > 128) {
        const int hmax = dk > dv ? dk : dv;
        p.qtiles = (nq + 127) / 128;
        const size_t bytes = sizeof(float) * (size_t)(2 * 128) * kLdQK;
        const dim3 grid(b * h * p.qtiles), block(256);
#define OVC_ATT_TILED(KG, DVT)                                                                                        \
    do {                                                                                                              \
        static std::once_flag once;                                                                                   \
        std::call_once(once, [] {                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_tiled_kernel<KG, DVT>),                \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);                         \
        });                                                                                                           \
        hipLaunchKernelGGL((attention_tiled_kernel<KG, DVT>), grid, block, bytes, ovc_hip_stream(stream), p);         \
    } while (0)
        if (hmax <= 16) OVC_ATT_TILED(2, 1); else if (hmax <= 32) OVC_ATT_TILED(4, 1); else OVC_ATT_TILED(8, 2);
#undef OVC_ATT_TILED
        OVC_RETURN_IF_LAUNCH_FAILED();
        return OVC_OK;
    }
    // nq > 128 with at most 128 keys: the LDS-score kernel over 64-query tiles
    const size_t lds_bytes = sizeof(float) * ((size_t)kQTile * kLdQK + 2 * (size_t)p.nkp * kLdQK + (size_t)kQTile * (p.nkp + 4));
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(attention_mfma_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    });
    hipLaunchKernelGGL(attention_mfma_kernel, dim3(b * h * p.qtiles), dim3(256), lds_bytes, ovc_hip_stream(stream), p);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

// =================================================================================================
// Decode-time attention (engine only)
// =================================================================================================

// One workgroup per beam row, all heads at once.
//   phase 0  the row's ancestor slots and pad flags for positions 0..t go to LDS (breaks the dependent
//            anc -> K load chain: every K/V load below is independent and can be in flight together)
//   phase 1  wave w takes keys w, w+4, ...: a key row [h*dk] is read fully coalesced (float4 per lane,
//            256 floats per instruction), multiplied with the matching q registers and reduced inside
//            each head's dk/4-lane group by shuffles -> sc[head][key]
//   phase 2  softmax over the t+1 keys of each head (one wave per head, lane = key)
//   phase 3  out = P V: thread = (float4 column, key group); groups are combined through LDS
constexpr int kSelfMaxHeads = 32;

// CH = float4-per-lane instructions per key row (h*dk / 256 rounded up): compile-time so that the loads of
// several keys can be issued back to back with clamped (always valid) addresses and no branches.
template <int CH>
__global__ __launch_bounds__(256) void decode_self_attention_kernel(DecodeSelfArgs p) {
    __shared__ int slots[64];
    __shared__ uint8_t pads[64];
    __shared__ float sc[kSelfMaxHeads][64];
    __shared__ __attribute__((aligned(16))) float red[256 * 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x, t = p.t;
    const int hk = p.h * p.dk;                       // == h * dv (checked on the host)

    if (tid <= t) {
        const int slot = tid == t ? r : p.anc[(size_t)r * p.anc_ld + tid];
        slots[tid] = slot;
        pads[tid] = p.padflag[(size_t)tid * p.pad_ld + slot];
    }
    int ecol[CH];
    bool evalid[CH];
    f32x4 q4[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int e = c * 256 + lane * 4;
        evalid[c] = e < hk;
        ecol[c] = min(e, hk - 4);
        q4[c] = *reinterpret_cast<const f32x4*>(p.q + (size_t)r * p.ldq + ecol[c]);
        if (!evalid[c]) q4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    __syncthreads();

    const int group = p.dk >> 2;                     // lanes per head: a power of two <= 16
    const float scale_div = sqrtf((float)p.dk);
    const int niter = (t - wave + 4) >> 2;           // keys wave, wave+4, ... <= t
    for (int i0 = 0; i0 < niter; i0 += 4) {
        f32x4 k4[4][CH];
        int jj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {                // four keys in flight per wave
            jj[u] = min(wave + 4 * (i0 + u), t);
            const float* krow = p.kcache + (size_t)jj[u] * p.pos_stride + (size_t)slots[jj[u]] * p.ldkv;
#pragma unroll
            for (int c = 0; c < CH; ++c) k4[u][c] = *reinterpret_cast<const f32x4*>(krow + ecol[c]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool live = i0 + u < niter;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                float part = (q4[c][0] * k4[u][c][0] + q4[c][1] * k4[u][c][1]) + (q4[c][2] * k4[u][c][2] + q4[c][3] * k4[u][c][3]);
                for (int off = 1; off < group; off <<= 1) part += __shfl_xor(part, off, 64);
                if (live && evalid[c] && (lane & (group - 1)) == 0)
                    sc[(c * 256 + lane * 4) / p.dk][jj[u]] = pads[jj[u]] ? -INFINITY : part / scale_div;
            }
        }
    }
    __syncthreads();

    for (int hd = wave; hd < p.h; hd += 4) {
        const float s = lane <= t ? sc[hd][lane] : -INFINITY;
        const float mx = wave_max(s);
        const float e = lane <= t ? expf(s - mx) : 0.f;
        const float sum = wave_sum(e);
        if (lane <= t) sc[hd][lane] = e / sum;
    }
    __syncthreads();

    const int cols = hk >> 2;                        // float4 columns of the output row (<= 256)
    const int groups = 256 / cols;                   // key groups working in parallel
    const int col = tid % cols, g = tid / cols;
    const int hd = (col * 4) / p.dv;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // keys g, g+groups, ... <= t (0 when g > t); threads past the last whole key group (256 % cols != 0) idle
    const int nkeys = g < groups ? (t - g + groups) / groups : 0;
    for (int i0 = 0; i0 < nkeys; i0 += 4) {
        f32x4 v4[4];
        int jj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            jj[u] = min(g + groups * (i0 + u), t);
            v4[u] = *reinterpret_cast<const f32x4*>(p.vcache + (size_t)jj[u] * p.pos_stride + (size_t)slots[jj[u]] * p.ldkv + col * 4);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u < nkeys) acc += v4[u] * sc[hd][jj[u]];
    }
    if (g > 0) *reinterpret_cast<f32x4*>(red + (size_t)tid * 4) = acc;
    __syncthreads();
    if (g == 0) {
        for (int gg = 1; gg < groups; ++gg) acc += *reinterpret_cast<const f32x4*>(red + (size_t)(gg * cols + col) * 4);
        *reinterpret_cast<f32x4*>(p.out + (size_t)r * p.ldo + col * 4) = acc;
    }
}

// Per-image form with ancestor de-duplication (round 3).  The k beams of an image descend from one another: at step t
// their histories name, per position j, only a few DISTINCT cache rows (measured on the BASELINE workload: 0.43 of the
// k (t + 1) rows the per-row kernel above reads -- one copy per beam).  The rows a position can name are the image's own
// width_j slots of that position's cache block (width_0 = 1, width_j = k), so the union is found with a k-bit mask per
// position.  One workgroup per (image, 4 heads), one wave per head, no LDS for data -- the structure of the
// cross-attention kernel below with a gathered key list:
//   wave 0   lane j <= t ORs the local slots of the image's beams at position j into a mask, a wave scan turns the
//            population counts into list offsets, and every (position, slot) that some beam names becomes one key
//            n -> (j, l) of an LDS list (<= 16 NT entries), with its <pad> flag and each beam's slot per position
//   S^T[key][beam] = K Q^T  on v_mfma_f32_16x16x4_f32, the beams as the 16-wide N dimension: every distinct key row is
//            read once and scored against all beams; a (key, beam) pair counts only if the beam's history names that
//            slot at that position (sl[beam][j] == l) -- the others get -inf, i.e. probability exactly 0
//   softmax over the keys held in accumulator registers, O^T = V^T P^T with the probabilities as the B operand.
// Key tiles past the end of the list are skipped with wave-uniform branches.  NT = key tiles the instance can hold
// (worst case k (t + 1) keys, chosen by the host), SB = d_k / 16.
template <int NT, int SB>
__global__ __launch_bounds__(256) void decode_self_attention_mfma_kernel(DecodeSelfArgs p) {
    __shared__ unsigned short keyinfo[NT * 16];          // (position << 3) | local slot
    __shared__ uint8_t keypad[NT * 16];
    __shared__ uint8_t sl[OVC_MAX_BEAM][64];             // local slot of beam i at position j
    __shared__ int nkeys_shared;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, t = p.t, W = p.width;
    const int hd = min((int)blockIdx.y * 4 + wave, p.h - 1);
    const bool live = (int)blockIdx.y * 4 + wave < p.h;          // surplus waves redo the last head, store nothing
    const int r = lane & 15, kq = lane >> 4;

    // this wave's query fragments (independent of the key list: in flight while wave 0 builds it)
    const float* qg = p.q + (size_t)(b * W + min(r, W - 1)) * p.ldq + hd * p.dk;
    f32x4 qf[SB];
#pragma unroll
    for (int S = 0; S < SB; ++S) qf[S] = *reinterpret_cast<const f32x4*>(qg + 16 * S + 4 * kq);

    if (wave == 0) {
        const int j = lane;
        const int wj = j == 0 ? 1 : W;                        // slots of position j's cache block that belong to this image
        int slot[OVC_MAX_BEAM];
        uint8_t pad[OVC_MAX_BEAM];
        unsigned mask = 0;
        if (j <= t) {
#pragma unroll
            for (int i = 0; i < OVC_MAX_BEAM; ++i)             // all loads first: ancestor slots and the block's <pad> flags
                slot[i] = i < W ? (j == t ? i : p.anc[(size_t)(b * W + i) * p.anc_ld + j] - b * wj) : 0;
#pragma unroll
            for (int l = 0; l < OVC_MAX_BEAM; ++l) pad[l] = l < wj ? p.padflag[(size_t)j * p.pad_ld + b * wj + l] : 0;
#pragma unroll
            for (int i = 0; i < OVC_MAX_BEAM; ++i)
                if (i < W) {
                    const int s = min(max(slot[i], 0), wj - 1);   // a corrupt table can never index outside the image's block
                    sl[i][j] = (uint8_t)s;
                    mask |= 1u << s;
                }
        }
        const int cnt = __popc(mask);
        int incl = cnt;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int up = __shfl_up(incl, off, 64);
            if (lane >= off) incl += up;
        }
        int n = incl - cnt;
#pragma unroll
        for (int l = 0; l < OVC_MAX_BEAM; ++l)
            if (mask & (1u << l)) {
                if (n < NT * 16) { keyinfo[n] = (unsigned short)((j << 3) | l); keypad[n] = pad[l]; }
                ++n;
            }
        if (lane == 63) nkeys_shared = min(incl, NT * 16);
    }
    __syncthreads();
    const int nkeys = nkeys_shared;

    // ---- K fragments of the listed keys: tile T holds keys 16 T .. 16 T + 15, lane r loads key 16 T + r -------------
    f32x4 kf[NT][SB];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        if (16 * T < nkeys) {                                  // wave-uniform
            const int info = keyinfo[min(16 * T + r, nkeys - 1)];
            const int j = info >> 3, l = info & 7;
            const float* krow = p.kcache + (size_t)j * p.pos_stride + (size_t)(b * (j == 0 ? 1 : W) + l) * p.ldkv + hd * p.dk + 4 * kq;
#pragma unroll
            for (int S = 0; S < SB; ++S) kf[T][S] = *reinterpret_cast<const f32x4*>(krow + 16 * S);
        }
    }
    if (r >= W) {
#pragma unroll
        for (int S = 0; S < SB; ++S) qf[S] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    f32x4 st[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) st[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        if (16 * T < nkeys) {
#pragma unroll
            for (int S = 0; S < SB; ++S)
#pragma unroll
                for (int e = 0; e < 4; ++e) st[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[T][S][e], qf[S][e], st[T], 0, 0, 0);
        }
    }

    // ---- V fragments (in flight during the softmax): lane (r, kq), register g <-> key 16 T + 4 kq + g, columns 4 r .. ----
    const int vc = 4 * min(r, (p.dv >> 2) - 1);
    f32x4 vf[NT][4];
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        if (16 * T < nkeys) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int info = keyinfo[min(16 * T + 4 * kq + g, nkeys - 1)];
                const int j = info >> 3, l = info & 7;
                vf[T][g] = *reinterpret_cast<const f32x4*>(p.vcache + (size_t)j * p.pos_stride +
                                                           (size_t)(b * (j == 0 ? 1 : W) + l) * p.ldkv + hd * p.dv + vc);
            }
        }
    }

    // ---- scale, validity, softmax over the keys of this lane's beam column -------------------------------------------
    const float scale_div = sqrtf((float)p.dk);
    const int beam = min(r, W - 1);
    float mx = -INFINITY;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int key = 16 * T + 4 * kq + g;
            float s = -INFINITY;
            if (key < nkeys) {
                const int info = keyinfo[key];
                if (!keypad[key] && sl[beam][info >> 3] == (info & 7)) s = st[T][g] / scale_div;
            }
            st[T][g] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float e = expf(st[T][g] - mx);
            st[T][g] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) st[T][g] = st[T][g] / sum;

    // ---- O^T = V^T P^T -----------------------------------------------------------------------------------------------
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        if (16 * T < nkeys) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[T][g][e], st[T][g], acc[e], 0, 0, 0);
        }
    }
    if (live && r < W) {
        float* orow = p.out + (size_t)(b * W + r) * p.ldo + hd * p.dv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dvb = 16 * kq + 4 * g;
            if (dvb < p.dv) *reinterpret_cast<f32x4*>(orow + dvb) = f32x4{acc[0][g], acc[1][g], acc[2][g], acc[3][g]};
        }
    }
}

int ovc_decode_self_attention(const DecodeSelfArgs& p, int rows, hipStream_t stream) {
    const int hk = p.h * p.dk;
    if (p.t < 0 || p.t >= 64 || p.h <= 0 || p.h > kSelfMaxHeads) return OVC_EINVAL;
    if (p.dk != p.dv || (p.dk & (p.dk - 1)) || p.dk < 4 || p.dk > 64) return OVC_EINVAL;   // dk in {4,8,16,32,64}
    if (hk > 1024) return OVC_EINVAL;
    // per-image kernel with ancestor de-duplication: the image's rows in one workgroup, at most 112 listed keys
    static const bool per_row = OVC_HOOK_ENV("OVC_SELF_ATTENTION_ROWS") != nullptr;     // A/B switch: the round-1 per-row kernel
    const int W = p.width;
    if (!per_row && W >= 1 && W <= OVC_MAX_BEAM && rows % W == 0 && p.dk >= 16 && (p.t == 0 ? 1 : W * (p.t + 1)) <= 112) {
        const int worst = p.t == 0 ? 1 : W * (p.t + 1), tiles = (worst + 15) / 16;
        const dim3 grid(rows / W, (p.h + 3) / 4), block(256);
#define OVC_SELF(NT, SB) hipLaunchKernelGGL((decode_self_attention_mfma_kernel<NT, SB>), grid, block, 0, stream, p)
#define OVC_SELF_NT(SB)                                                                                             \
    do {                                                                                                            \
        if (tiles <= 1) OVC_SELF(1, SB); else if (tiles <= 2) OVC_SELF(2, SB); else if (tiles <= 4) OVC_SELF(4, SB); \
        else OVC_SELF(7, SB);                                                                                       \
    } while (0)
        if (p.dk == 64) OVC_SELF_NT(4); else if (p.dk == 32) OVC_SELF_NT(2); else OVC_SELF_NT(1);
#undef OVC_SELF_NT
#undef OVC_SELF
        OVC_RETURN_IF_LAUNCH_FAILED();
        return OVC_OK;
    }
    if (hk <= 256) hipLaunchKernelGGL(decode_self_attention_kernel<1>, dim3(rows), dim3(256), 0, stream, p);
    else if (hk <= 512) hipLaunchKernelGGL(decode_self_attention_kernel<2>, dim3(rows), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(decode_self_attention_kernel<4>, dim3(rows), dim3(256), 0, stream, p);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

// Decode cross-attention: the k beams of an image share the image's projected encoder K/V.
//
// decode_cross_attention_mfma_kernel -- one wave per (image, head, level), no LDS, both contractions on
// the matrix cores (v_mfma_f32_16x16x4_f32) with the beams as the 16-wide N dimension:
//   S^T[key][beam] = K Q^T : A = K rows (16 keys per tile), B = Q^T.  Lane (r = lane & 15, kq = lane >> 4)
//       loads float4 K[key r][16 S + 4 kq ..] / Q[beam r][16 S + 4 kq ..] and feeds element e to MFMA
//       (S, e); both operands use the same d <-> (S, kq, e) assignment, so every d is summed once.
//   softmax over keys: the accumulator has the beam on the lane (col = lane & 15) and the keys on
//       registers (row = 4 kq + reg): reduce over registers, then __shfl_xor 16 / 32 across the four kq lanes.
//   O^T[dv][beam] = V^T P^T : the probabilities stay in the accumulator registers and are used directly as
//       the B operand (lane (beam, kq), register reg  <->  key 16 T + 4 kq + reg); A = V^T is loaded as
//       float4 V[key][4 r ..] (a fully coalesced 256-byte head slice per key) and element e' feeds MFMA e',
//       whose output row r is dv = 4 r + e'.  Four MFMA results give each lane float4s of consecutive dv.
// K and V are read from HBM exactly once with 16-byte loads; the kernel is bound by that stream
// (52 MB per layer-step at B = 256).
// NT = key tiles (16 keys each), SB = d_k / 16: compile-time so that every load below is unconditional
// (hipcc branches around a guarded load and waits vmcnt(0) after it, serialising the whole K/V stream).
template <int NT, int SB>
__global__ __launch_bounds__(256) void decode_cross_attention_mfma_kernel(DecodeCrossArgs p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, hd = min((int)blockIdx.y * 4 + wave, p.heads - 1), lvl = blockIdx.z;
    const bool live = (int)blockIdx.y * 4 + wave < p.heads;      // surplus waves redo the last head, store nothing
    const int N = p.n, W = p.width;
    const int r = lane & 15, kq = lane >> 4;
    const float* kg = p.kx + (size_t)lvl * p.level_stride + (size_t)b * N * p.ldkv + hd * p.dk;
    const float* vg = p.vx + (size_t)lvl * p.level_stride + (size_t)b * N * p.ldkv + hd * p.dv;
    const float* qg = p.q + (size_t)(b * W + min(r, W - 1)) * p.ldq + hd * p.dk;

    // ---- all loads of the score phase, back to back -------------------------------------------------
    f32x4 qf[SB], kf[NT][SB];
#pragma unroll
    for (int S = 0; S < SB; ++S) qf[S] = *reinterpret_cast<const f32x4*>(qg + 16 * S + 4 * kq);
#pragma unroll
    for (int T = 0; T < NT; ++T) {
        const float* krow = kg + (size_t)min(16 * T + r, N - 1) * p.ldkv + 4 * kq;
#pragma unroll
        for (int S = 0; S < SB; ++S) kf[T][S] = *reinterpret_cast<const f32x4*>(krow + 16 * S);
    }
    // key mask bytes of this lane's keys (16 T + 4 kq + g); a dummy all-zero row when there is no mask
    uint8_t mk[NT][4];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) mk[T][g] = 0;
    if (p.encmask) {                                   // uniform: one branch around all the byte loads
        const uint8_t* mrow = p.encmask + (size_t)b * N;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int g = 0; g < 4; ++g) mk[T][g] = mrow[min(16 * T + 4 * kq + g, N - 1)];
    }
    if (r >= W) {
#pragma unroll
        for (int S = 0; S < SB; ++S) qf[S] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // NT independent accumulation chains, issued round-robin: a 16x16x4 MFMA can issue every 32 cycles but its result is
    // only available to a dependent one after 40, so one chain at a time would stall on every instruction.  The sum
    // over d inside each key tile keeps its order (S, then e).
    f32x4 st[NT];
#pragma unroll
    for (int T = 0; T < NT; ++T) st[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int S = 0; S < SB; ++S)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int T = 0; T < NT; ++T) st[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[T][S][e], qf[S][e], st[T], 0, 0, 0);

    // ---- V loads are issued before the softmax arithmetic so that they are in flight meanwhile ------------------
    const int vc = 4 * min(r, (p.dv >> 2) - 1);
    f32x4 vf[NT][4];
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            vf[T][g] = *reinterpret_cast<const f32x4*>(vg + (size_t)min(16 * T + 4 * kq + g, N - 1) * p.ldkv + vc);

    // scale, mask, softmax over the keys of this lane's beam column
    const float scale_div = sqrtf((float)p.dk);
    float mx = -INFINITY;
#pragma unroll
    for (int T = 0; T < NT; ++T) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int key = 16 * T + 4 * kq + g;
            float s = st[T][g] / scale_div;
            if (key >= N || mk[T][g]) s = -INFINITY;
            st[T][g] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float e = expf(st[T][g] - mx);
            st[T][g] = e;
            sum += e;
        }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g) st[T][g] = st[T][g] / sum;

    // O^T = V^T P^T (columns of V beyond d_v contribute to output rows that are never stored)
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[T][g][e], st[T][g], acc[e], 0, 0, 0);

    // acc[e][g] is O[beam = column][dv = 4 (4 kq + g) + e] where row 4 kq + g of the MFMA is the V column
    // group loaded by lane r' = 4 kq + g: four consecutive dv per (lane, g)
    if (live && r < W) {
        float* orow = p.out + (size_t)lvl * p.out_level_stride + (size_t)(b * W + r) * p.ldo + hd * p.dv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dvb = 16 * kq + 4 * g;
            if (dvb < p.dv) *reinterpret_cast<f32x4*>(orow + dvb) = f32x4{acc[0][g], acc[1][g], acc[2][g], acc[3][g]};
        }
    }
}

// The same kernel for more than 128 regions (round 4; the reference has no limit): the keys pass through the wave in chunks
// of 64 (four 16-key tiles), in ascending order, under an online softmax -- a running maximum and sum per beam column, the
// output accumulators rescaled by exp(M_old - M_new) when a chunk raises the maximum (a per-lane scalar: a lane's accumulator
// registers all belong to ITS beam).  Fixed chunk order: results depend on the operands only.  N <= 128 never comes here.
template <int SB>
__global__ __launch_bounds__(256) void decode_cross_attention_tiled_kernel(DecodeCrossArgs p) {
    constexpr int NT = 4;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int b = blockIdx.x, hd = min((int)blockIdx.y * 4 + wave, p.heads - 1), lvl = blockIdx.z;
    const bool live = (int)blockIdx.y * 4 + wave < p.heads;
    const int N = p.n, W = p.width;
    const int r = lane & 15, kq = lane >> 4;
    const float* kg = p.kx + (size_t)lvl * p.level_stride + (size_t)b * N * p.ldkv + hd * p.dk;
    const float* vg = p.vx + (size_t)lvl * p.level_stride + (size_t)b * N * p.ldkv + hd * p.dv;
    const float* qg = p.q + (size_t)(b * W + min(r, W - 1)) * p.ldq + hd * p.dk;
    const uint8_t* mrow = p.encmask ? p.encmask + (size_t)b * N : nullptr;

    f32x4 qf[SB];
#pragma unroll
    for (int S = 0; S < SB; ++S) qf[S] = *reinterpret_cast<const f32x4*>(qg + 16 * S + 4 * kq);
    if (r >= W) {
#pragma unroll
        for (int S = 0; S < SB; ++S) qf[S] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float scale_div = sqrtf((float)p.dk);
    const int vc = 4 * min(r, (p.dv >> 2) - 1);
    float m_run = -INFINITY, l_run = 0.f;
    f32x4 acc[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = f32x4{0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < N; k0 += 16 * NT) {
        f32x4 kf[NT][SB];
#pragma unroll
        for (int T = 0; T < NT; ++T) {
            const float* krow = kg + (size_t)min(k0 + 16 * T + r, N - 1) * p.ldkv + 4 * kq;
#pragma unroll
            for (int S = 0; S < SB; ++S) kf[T][S] = *reinterpret_cast<const f32x4*>(krow + 16 * S);
        }
        uint8_t mk[NT][4];
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int g = 0; g < 4; ++g) mk[T][g] = 0;
        if (mrow) {
#pragma unroll
            for (int T = 0; T < NT; ++T)
#pragma unroll
                for (int g = 0; g < 4; ++g) mk[T][g] = mrow[min(k0 + 16 * T + 4 * kq + g, N - 1)];
        }
        f32x4 st[NT];
#pragma unroll
        for (int T = 0; T < NT; ++T) st[T] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int S = 0; S < SB; ++S)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int T = 0; T < NT; ++T) st[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[T][S][e], qf[S][e], st[T], 0, 0, 0);
        f32x4 vf[NT][4];
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                vf[T][g] = *reinterpret_cast<const f32x4*>(vg + (size_t)min(k0 + 16 * T + 4 * kq + g, N - 1) * p.ldkv + vc);

        float mx = -INFINITY;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int key = k0 + 16 * T + 4 * kq + g;
                float s = st[T][g] / scale_div;
                if (key >= N || mk[T][g]) s = -INFINITY;
                st[T][g] = s;
                mx = fmaxf(mx, s);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const bool none = m_new == -INFINITY;
        const float alpha = none ? 1.f : expf(m_run - m_new);
        float sum = 0.f;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float e = none ? 0.f : expf(st[T][g] - m_new);
                st[T][g] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        l_run = l_run * alpha + sum;
        m_run = m_new;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] * alpha;
#pragma unroll
        for (int T = 0; T < NT; ++T)
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[e] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[T][g][e], st[T][g], acc[e], 0, 0, 0);
    }
    if (live && r < W) {
        float* orow = p.out + (size_t)lvl * p.out_level_stride + (size_t)(b * W + r) * p.ldo + hd * p.dv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int dvb = 16 * kq + 4 * g;
            if (dvb < p.dv)
                *reinterpret_cast<f32x4*>(orow + dvb) = f32x4{acc[0][g] / l_run, acc[1][g] / l_run, acc[2][g] / l_run, acc[3][g] / l_run};
        }
    }
}

// Head sizes 4 and 8 (not multiples of the 16-deep MFMA k block): VALU dots on LDS-staged rows.  The scores of all N keys
// stay in LDS; K and then V of the (image, head) pass through in chunks of 128 rows, ascending, so any region count fits
// (round 4) and the sums keep the key order of the one-shot form.
// Exercised by tests/test_engine_gpu.py::test_unusual_dimensions_against_oracle (d_k = 8 and d_k = 4 cases).
constexpr int kCrossChunk = 128;
__global__ __launch_bounds__(256) void decode_cross_attention_lds_kernel(DecodeCrossArgs p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x, hd = blockIdx.y, lvl = blockIdx.z;
    const int N = p.n, W = p.width;
    float* Xs = lds;                         // [128][68]: a chunk of K rows, later of V rows
    float* qs = Xs + kCrossChunk * kLdQK;    // [W][64]
    float* sc = qs + W * 64;                 // [W][N]

    const float* kg = p.kx + (size_t)lvl * p.level_stride + (size_t)b * N * p.ldkv;
    const float* vg = p.vx + (size_t)lvl * p.level_stride + (size_t)b * N * p.ldkv;
    const int c4 = tid & 15, r0 = tid >> 4, col = c4 * 4;
    for (int idx = tid; idx < W * 16; idx += 256) {
        const int i = idx >> 4, cc = (idx & 15) * 4;
        f32x4 qv = {0.f, 0.f, 0.f, 0.f};
        if (cc < p.dk) qv = *reinterpret_cast<const f32x4*>(p.q + (size_t)(b * W + i) * p.ldq + hd * p.dk + cc);
        *reinterpret_cast<f32x4*>(qs + i * 64 + cc) = qv;
    }
    const float scale_div = sqrtf((float)p.dk);
    const int k4n = (p.dk + 3) >> 2;
    for (int k0 = 0; k0 < N; k0 += kCrossChunk) {
        const int nc = min(kCrossChunk, N - k0);
        for (int r = r0; r < nc; r += 16) {
            f32x4 kv = {0.f, 0.f, 0.f, 0.f};
            if (col < p.dk) kv = *reinterpret_cast<const f32x4*>(kg + (size_t)(k0 + r) * p.ldkv + hd * p.dk + col);
            *reinterpret_cast<f32x4*>(Xs + r * kLdQK + col) = kv;
        }
        __syncthreads();
        for (int idx = tid; idx < W * nc; idx += 256) {
            const int i = idx / nc, j = idx - i * nc;
            const f32x4* kr = reinterpret_cast<const f32x4*>(Xs + j * kLdQK);
            const f32x4* qr = reinterpret_cast<const f32x4*>(qs + i * 64);
            float acc = 0.f;
            for (int c = 0; c < k4n; ++c) {
                const f32x4 a = qr[c], kk = kr[c];
                acc += (a[0] * kk[0] + a[1] * kk[1]) + (a[2] * kk[2] + a[3] * kk[3]);
            }
            float s = acc / scale_div;
            if (p.encmask && p.encmask[(size_t)b * N + k0 + j]) s = -INFINITY;
            sc[i * N + k0 + j] = s;
        }
        __syncthreads();
    }
    for (int i = wave; i < W; i += 4) {
        float mx = -INFINITY;
        for (int j = lane; j < N; j += 64) mx = fmaxf(mx, sc[i * N + j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < N; j += 64) {
            const float e = expf(sc[i * N + j] - mx);
            sc[i * N + j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        for (int j = lane; j < N; j += 64) sc[i * N + j] = sc[i * N + j] / sum;
    }
    // out[i][d] = sum_j P[i][j] V[j][d], j ascending; W * d_v <= 512 outputs: at most two per thread
    const int nout = W * p.dv;
    float acc[2] = {0.f, 0.f};
    for (int k0 = 0; k0 < N; k0 += kCrossChunk) {
        const int nc = min(kCrossChunk, N - k0);
        __syncthreads();                                   // the probabilities are complete / the previous chunk is consumed
        for (int r = r0; r < nc; r += 16) {
            f32x4 vv = {0.f, 0.f, 0.f, 0.f};
            if (col < p.dv) vv = *reinterpret_cast<const f32x4*>(vg + (size_t)(k0 + r) * p.ldkv + hd * p.dv + col);
            *reinterpret_cast<f32x4*>(Xs + r * kLdQK + col) = vv;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int idx = tid + 256 * u;
            if (idx < nout) {
                const int i = idx / p.dv, d = idx - i * p.dv;
                for (int j = 0; j < nc; ++j) acc[u] += sc[i * N + k0 + j] * Xs[j * kLdQK + d];
            }
        }
    }
    float* og = p.out + (size_t)lvl * p.out_level_stride;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = tid + 256 * u;
        if (idx < nout) {
            const int i = idx / p.dv, d = idx - i * p.dv;
            og[(size_t)(b * W + i) * p.ldo + hd * p.dv + d] = acc[u];
        }
    }
}

int ovc_decode_cross_attention(const DecodeCrossArgs& p, int B, int h, int levels, hipStream_t stream) {
    if (p.n <= 0 || p.n > OVC_MAX_REGIONS || p.width <= 0 || p.width > OVC_MAX_BEAM) return OVC_EINVAL;
    if (p.dk > 64 || p.dv > 64 || (p.dk & 3) || (p.dv & 3) || p.heads != h) return OVC_EINVAL;
    if (p.dk == 16 || p.dk == 32 || p.dk == 64) {
        const dim3 grid(B, (h + 3) / 4, levels), block(256);
        if (p.n > 128) {        // more regions than the register-resident instances hold: key chunks + online softmax
            if (p.dk == 64) hipLaunchKernelGGL(decode_cross_attention_tiled_kernel<4>, grid, block, 0, stream, p);
            else if (p.dk == 32) hipLaunchKernelGGL(decode_cross_attention_tiled_kernel<2>, grid, block, 0, stream, p);
            else hipLaunchKernelGGL(decode_cross_attention_tiled_kernel<1>, grid, block, 0, stream, p);
            OVC_RETURN_IF_LAUNCH_FAILED();
            return OVC_OK;
        }
        const bool small = p.n <= 64;
#define OVC_CROSS(NT, SB) hipLaunchKernelGGL((decode_cross_attention_mfma_kernel<NT, SB>), grid, block, 0, stream, p)
        if (p.dk == 64) { if (small) OVC_CROSS(4, 4); else OVC_CROSS(8, 4); }
        else if (p.dk == 32) { if (small) OVC_CROSS(4, 2); else OVC_CROSS(8, 2); }
        else { if (small) OVC_CROSS(4, 1); else OVC_CROSS(8, 1); }
#undef OVC_CROSS
        OVC_RETURN_IF_LAUNCH_FAILED();
        return OVC_OK;
    }
    const size_t lds_bytes = sizeof(float) * ((size_t)kCrossChunk * kLdQK + (size_t)p.width * 64 + (size_t)p.width * p.n);
    static std::once_flag attr_once;
    std::call_once(attr_once, [] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(decode_cross_attention_lds_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    });
    hipLaunchKernelGGL(decode_cross_attention_lds_kernel, dim3(B, h, levels), dim3(256), lds_bytes, stream, p);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}
