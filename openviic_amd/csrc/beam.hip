// Beam-search selection and bookkeeping on the device.
//
// Reference semantics (models/modules/beam_search.py:41-118), restated:
//   * candidates of image b are the width*V values  running[b,i] + logp[b,i,w]  (i = live beam,
//     w = word); a frozen beam (alive = 0, it has emitted <eos>) offers word 0 at its running score
//     and -999 for every other word;
//   * the k best candidates are taken in descending order, ties broken by the lower flat index
//     (torch.sort on CPU is stable);
//   * beam = idx / V, word = idx % V; every per-beam quantity follows the selected beam.
//
// The reference materialises log_softmax over [B*k, V] and then fully sorts [B, k*V]; here the
// log-sum-exp, the candidate scores and a k-way partial selection are fused in one pass per beam row,
// and the image's winners are the k best of its rows' k best.
#include "common.h"

namespace {

constexpr int kSelThreads = 256;
constexpr int kMaxK = OVC_MAX_BEAM;
constexpr int kSurvivorCap = 1024;   // LDS list of candidates that can still reach the row's top k

struct Cand { float v; int idx; };

__device__ __forceinline__ bool better(float v, int idx, float bv, int bidx) {
    return v > bv || (v == bv && idx < bidx);
}

__device__ __forceinline__ Cand wave_best(Cand c) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(c.v, off, 64);
        const int oi = __shfl_xor(c.idx, off, 64);
        if (better(ov, oi, c.v, c.idx)) { c.v = ov; c.idx = oi; }
    }
    return c;
}

// The k-th largest (with multiplicity) of the 64 lane values of a wave, exactly: a binary search over the bits of the
// order-preserving integer image of a float, one ballot + population count per bit and no cross-lane data movement
// (k rounds of a 6-step shuffle arg-max cost ~10x the latency).  NaN lanes count as the smallest value.
__device__ __forceinline__ float wave_kth_largest(float v, int k) {
    const unsigned bits = __float_as_uint(v);
    const unsigned key = (v != v) ? 0u : ((bits & 0x80000000u) ? ~bits : (bits | 0x80000000u));   // monotone in v
    unsigned t = 0;
#pragma unroll
    for (int bit = 31; bit >= 0; --bit) {
        const unsigned cand = t | (1u << bit);
        if (__popcll(__ballot(key >= cand)) >= k) t = cand;
    }
    if (t == 0u) return -INFINITY;                       // fewer than k comparable values
    const unsigned back = (t & 0x80000000u) ? (t & 0x7fffffffu) : ~t;
    return __uint_as_float(back);
}

// One workgroup (256 threads) per beam row -- B*width workgroups, several resident per CU, so one row's
// loads overlap another row's reductions (a single 1024-thread workgroup per image ran the same phases in
// lock-step on every CU: 30 us for 52 MB at B=256; per-row workgroups stream it at HBM rate).  The row's V
// logits are read once into registers (all loads in flight together); log-sum-exp, candidate scores and the
// row's k best candidates come from those registers.  Selection is threshold based: the k-th best of a wave's
// lane maxima bounds the row's k-th best from below, so only the few candidates at or above the largest such
// bound are collected (LDS list) and ranked by one wave.  The image's k winners are the k best of its rows'
// candidates (merged by beam_update_kernel / beam_merge_kernel with the same order: score, then flat index).
// kMasked: the caller wants the masked log-probabilities of every word written out (return_probs); the hot path does
// not, and then neither the stores nor their address arithmetic exist in the instruction stream.
template <int kPerThread, int kVec, bool kMasked>
__global__ __launch_bounds__(kSelThreads, (kVec == 4 && kPerThread <= 10 && !kMasked ? 5 : 1)) void beam_row_select_kernel(BeamSelectArgs p) {
    constexpr int kElems = kPerThread * kVec;      // logits per thread; element (j, e) is column kVec*(tid + j*256) + e
    constexpr int kWaves = kSelThreads / 64;
    __shared__ float red[kWaves];
    __shared__ float thr[kWaves];
    __shared__ int count;
    __shared__ float surv_v[kSurvivorCap];
    __shared__ int surv_i[kSurvivorCap];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x;
    const int W = p.width, V = p.V, k = p.k;
    const int i = row % W;                          // beam of this row inside its image
    const float run = p.running[row];
    const float alive = p.alive ? p.alive[row] : 1.0f;
    const bool live = alive != 0.0f;                // uniform over the workgroup
    float* cand_v = p.cand_v + (size_t)row * k;
    int* cand_i = p.cand_i + (size_t)row * k;

    if (!live && !kMasked) {
        // A frozen beam (it has emitted <eos>) offers word 0 at its running score and -999 for every other word
        // (beam_search.py:52-55): its k best are words 0..k-1, whatever the logits are.
        if (tid < k) {
            cand_v[tid] = tid == 0 ? run : (tid < V ? -999.0f : -INFINITY);
            cand_i[tid] = tid < V ? i * V + tid : 0x7fffffff;
        }
        if (tid == 0 && p.row_max_out) { p.row_max_out[row] = 0.f; p.row_lsum_out[row] = 0.f; }   // lp is multiplied by alive = 0
        return;
    }

    const float* x = p.logits + (size_t)row * p.ld;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.ld * 4, 0x00020000);
    // Column of element (j, e) = cbase + 4 * 256 * j + e.  The index lives in ONE register: every section below adds its
    // compile-time offsets on the fly, and an opaque copy per section keeps hipcc from computing all kElems indices once and
    // holding them across the kernel (round 2: 40 index registers + 40 compare masks -> 4 spills at five waves per SIMD).
    int cbase = kVec * tid;
    const int jfull = V / (kVec * kSelThreads);     // vectors j < jfull lie below V for every thread: no tail mask (uniform)
    float xv[kElems];
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        // unconditional loads from clamped (always valid) addresses; the tail is masked afterwards, so that all of
        // a thread's loads are in flight together (a guarded load costs a vmcnt(0) each)
        if (kVec == 4) {
            // raw buffer loads: one 32-bit lane offset for all of the thread's loads, the column block in the scalar
            // offset, the row's end in the descriptor (reads past it return 0 and are masked below) -- no per-load
            // 64-bit address registers, which is what keeps this kernel at five waves per SIMD
            const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, tid * 16, j * kSelThreads * 16, 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) xv[j * 4 + e] = v[e];
        } else {
            xv[j] = x[min(cbase + j * kSelThreads, V - 1)];
        }
    }
    asm volatile("" : "+v"(cbase));
#pragma unroll
    for (int j = 0; j < kPerThread; ++j) {
        if (j >= jfull) {                            // wave-uniform: only the vectors that can reach past V pay for the test
#pragma unroll
            for (int e = 0; e < kVec; ++e)
                if (cbase + j * kVec * kSelThreads + e >= V) xv[j * kVec + e] = -INFINITY;
        }
    }

    // ---- log-sum-exp: (x - max) - log(sum exp(x - max)), as ATen's log_softmax ------------------------------
    float mx = 0.f, ls = 0.f;
    if (!p.is_logp) {
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < kElems; ++j) m = fmaxf(m, xv[j]);
        m = wave_max(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) m = fmaxf(m, red[w]);
        mx = m;
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < kElems; ++j) sum += __expf(xv[j] - m);   // v_exp_f32 path (|rel err| ~2e-7 per term); exp(-inf) = 0 for the tail
        sum = wave_sum(sum);
        __syncthreads();
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        float tot = 0.f;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) tot += red[w];
        ls = logf(tot);
    }
    if (tid == 0 && p.row_max_out) { p.row_max_out[row] = mx; p.row_lsum_out[row] = ls; }

    // ---- candidate scores (kept in the logit registers) and each lane's best ------------------------------------
    // seq_mask * candidate + frozen * (1 - seq_mask) (beam_search.py:52-55) with seq_mask in {0, 1}: a live beam's
    // score is exactly run + lp (x + 0 == x), a frozen beam's exactly `frozen`.  A thread visits flat indices in
    // increasing order, hence a strict > keeps the lower index on ties.
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    float* mrow = kMasked ? p.masked_logp + (size_t)row * V : nullptr;
    asm volatile("" : "+v"(cbase));
#pragma unroll
    for (int j = 0; j < kElems; ++j) {
        const int c = cbase + (j / kVec) * kVec * kSelThreads + (j % kVec);
        float cand = -INFINITY;
        if (c < V) {
            const float lp = (xv[j] - mx) - ls;
            if (kMasked) mrow[c] = lp * alive;
            cand = live ? run + lp : (c == 0 ? run : -999.0f);
            if (cand > bv) { bv = cand; bi = c; }
        }
        xv[j] = cand;
    }
    bi = bi == 0x7fffffff ? bi : i * V + bi;

    // ---- a lower bound on the row's k-th best: the k-th best of one wave's lane maxima (k distinct candidates
    //      are >= it), tightened by taking the largest such bound over the waves ---------------------------------------
    {
        const float kth = wave_kth_largest(bv, k);
        if (lane == 0) thr[wave] = kth;
        if (tid == 0) count = 0;
    }
    __syncthreads();
    float T = thr[0];
#pragma unroll
    for (int w = 1; w < kWaves; ++w) T = fmaxf(T, thr[w]);

    // ---- survivors (score >= T; usually a few dozen) are appended to an LDS list -------------------------------------
    asm volatile("" : "+v"(cbase));
#pragma unroll
    for (int j = 0; j < kElems; ++j) {
        if (xv[j] >= T) {                            // tail elements hold -inf and T is a real candidate's score (or -inf: then checked)
            const int c = cbase + (j / kVec) * kVec * kSelThreads + (j % kVec);
            if (c < V) {
                const int pos = atomicAdd(&count, 1);
                if (pos < kSurvivorCap) { surv_v[pos] = xv[j]; surv_i[pos] = i * V + c; }
            }
        }
    }
    __syncthreads();
    const int nsurv = count;
    if (nsurv > kSurvivorCap) {
        // Massive ties (a live beam fed <pad> yields a uniform row: V equal scores; a frozen row when all
        // log-probs are wanted).  Rare, so it is written for few registers rather than speed -- the register peak
        // of this kernel decides whether all B*k workgroups are resident at once: k rounds of a block-wide argmax
        // over the candidates that come after the previous pick in the (score desc, index asc) order.
        // The candidates are recomputed from the logits in memory with the arithmetic of the register pass (same operands,
        // same operations: the same bits), so that this path keeps none of the kElems registers or their indices alive.
        float pv = INFINITY;
        int pi = -1;
        for (int round = 0; round < k; ++round) {
            Cand c; c.v = -INFINITY; c.idx = 0x7fffffff;
            for (int col = tid; col < V; col += kSelThreads) {
                const float cand = live ? run + ((x[col] - mx) - ls) : (col == 0 ? run : -999.0f);
                const int idx = i * V + col;
                const bool after = cand < pv || (cand == pv && idx > pi);
                if (after && better(cand, idx, c.v, c.idx)) { c.v = cand; c.idx = idx; }
            }
            c = wave_best(c);
            __syncthreads();                   // the previous round's (or the survivor list's) readers are done
            if (lane == 0) { surv_v[wave] = c.v; surv_i[wave] = c.idx; }
            __syncthreads();
            pv = surv_v[0]; pi = surv_i[0];
#pragma unroll
            for (int w = 1; w < kWaves; ++w)
                if (better(surv_v[w], surv_i[w], pv, pi)) { pv = surv_v[w]; pi = surv_i[w]; }
            if (tid == 0) { cand_v[round] = pv; cand_i[round] = pi; }
        }
        return;
    }
    // ---- rank the survivors: a survivor's rank is the number of survivors that beat it in the (score desc, flat
    //      index asc) order -- a strict total order, so ranks are unique and ranks 0..k-1 are the row's k best.  Every
    //      thread ranks its share against the whole list with broadcast LDS reads; no shuffles, no sorted lists.
    for (int e = tid; e < nsurv; e += kSelThreads) {
        const float v = surv_v[e];
        const int idx = surv_i[e];
        int rank = 0;
        for (int o = 0; o < nsurv; ++o) rank += better(surv_v[o], surv_i[o], v, idx) ? 1 : 0;
        if (rank < k) { cand_v[rank] = v; cand_i[rank] = idx; }
    }
    if (tid >= nsurv && tid < k) { cand_v[tid] = -INFINITY; cand_i[tid] = 0x7fffffff; }     // fewer than k candidates exist
}

// Any vocabulary size: the row is streamed from memory instead of held in registers -- one pass for the maximum,
// one for the sum of exponentials (and the masked log-probs), then k rounds of a block-wide argmax over the
// candidates that come after the previous pick in the (score desc, index asc) order.  k + 2 passes over the row:
// used only for vocabularies beyond the register-resident instances above (V > 16384; 64 * 256 without 16-byte rows).
__global__ __launch_bounds__(kSelThreads) void beam_row_select_streaming_kernel(BeamSelectArgs p) {
    constexpr int kWaves = kSelThreads / 64;
    __shared__ float red[kWaves];
    __shared__ float pick_v[kWaves];
    __shared__ int pick_i[kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x, W = p.width, V = p.V, k = p.k, i = row % W;
    const float run = p.running[row];
    const float alive = p.alive ? p.alive[row] : 1.0f;
    const bool live = alive != 0.0f;
    const float* x = p.logits + (size_t)row * p.ld;
    float* cand_v = p.cand_v + (size_t)row * k;
    int* cand_i = p.cand_i + (size_t)row * k;
    float mx = 0.f, ls = 0.f;
    if (!p.is_logp) {
        float m = -INFINITY;
        for (int c = tid; c < V; c += kSelThreads) m = fmaxf(m, x[c]);
        m = wave_max(m);
        if (lane == 0) red[wave] = m;
        __syncthreads();
        m = red[0];
        for (int w = 1; w < kWaves; ++w) m = fmaxf(m, red[w]);
        mx = m;
        float sum = 0.f;
        for (int c = tid; c < V; c += kSelThreads) sum += __expf(x[c] - m);
        sum = wave_sum(sum);
        __syncthreads();
        if (lane == 0) red[wave] = sum;
        __syncthreads();
        float tot = 0.f;
        for (int w = 0; w < kWaves; ++w) tot += red[w];
        ls = logf(tot);
    }
    if (tid == 0 && p.row_max_out) { p.row_max_out[row] = live ? mx : 0.f; p.row_lsum_out[row] = live ? ls : 0.f; }
    if (p.masked_logp) {
        float* mrow = p.masked_logp + (size_t)row * V;
        for (int c = tid; c < V; c += kSelThreads) mrow[c] = ((x[c] - mx) - ls) * alive;
    }
    float pv = INFINITY;
    int pi = -1;
    for (int round = 0; round < k; ++round) {
        Cand c; c.v = -INFINITY; c.idx = 0x7fffffff;
        for (int col = tid; col < V; col += kSelThreads) {
            const float cand = live ? run + ((x[col] - mx) - ls) : (col == 0 ? run : -999.0f);
            const int idx = i * V + col;
            const bool after = cand < pv || (cand == pv && idx > pi);
            if (after && better(cand, idx, c.v, c.idx)) { c.v = cand; c.idx = idx; }
        }
        c = wave_best(c);
        __syncthreads();
        if (lane == 0) { pick_v[wave] = c.v; pick_i[wave] = c.idx; }
        __syncthreads();
        pv = pick_v[0]; pi = pick_i[0];
        for (int w = 1; w < kWaves; ++w)
            if (better(pick_v[w], pick_i[w], pv, pi)) { pv = pick_v[w]; pi = pick_i[w]; }
        if (tid == 0) { cand_v[round] = pv; cand_i[round] = pi; }
    }
}

// The k best of an image's width*k row candidates, in order; lane r < k of the (single) wave returns the r-th.
__device__ __forceinline__ Cand merge_row_candidates(const float* cand_v, const int* cand_i, int b, int W, int k, int lane) {
    Cand c; c.v = -INFINITY; c.idx = 0x7fffffff;
    if (lane < W * k) { c.v = cand_v[(size_t)b * W * k + lane]; c.idx = cand_i[(size_t)b * W * k + lane]; }
    Cand mine = c;
    for (int round = 0; round < k; ++round) {
        const Cand w = wave_best(c);
        if (lane == round) mine = w;
        if (c.idx == w.idx) { c.v = -INFINITY; c.idx = 0x7fffffff; }      // flat indices are unique
    }
    return mine;
}

__global__ __launch_bounds__(64) void beam_merge_kernel(BeamSelectArgs p) {
    const Cand best = merge_row_candidates(p.cand_v, p.cand_i, blockIdx.x, p.width, p.k, threadIdx.x);
    if ((int)threadIdx.x < p.k) {
        // no winner at all (NaN scores: an image without valid regions): an in-range index, as in beam_update_kernel
        const int idx = (unsigned)best.idx < (unsigned)(p.width * p.V) ? best.idx : (int)threadIdx.x;
        p.chosen[blockIdx.x * p.k + threadIdx.x] = (int64_t)idx;
        p.score[blockIdx.x * p.k + threadIdx.x] = best.v;
    }
}

// Per-image bookkeeping after a selection: histories, per-token log-probs, ancestor table, alive
// flags and next input tokens follow the selected beams (beam_search.py:58-81 and the state
// re-ordering of :19-34,61 expressed as an ancestor-slot table instead of cache gathers).
// 256 threads per image: wave 0 merges the candidates and writes the per-beam scalars, then all four waves move the
// histories and build the next step's input rows with every load issued before the first store (the single-wave version
// walked ten dependent load -> store round trips for the embedding rows alone: 9.7 us).
// Everything that follows the choice of the image's k winners (parent[j], word[j] in LDS, visible to all threads): the
// histories / per-token log-probs / ancestor slots follow the selected beams, and the next step's input rows are built.
template <int kThreads>
__device__ __forceinline__ void beam_follow_winners(const BeamUpdateArgs& p, int b, int tid, const int* parent, const int* word) {
    const int k = p.k, W = p.width, T = p.T, t = p.t;
    for (int idx = tid; idx < k * t; idx += kThreads) {
        const int j = idx / t, pos = idx - j * t;
        const size_t src = ((size_t)b * W + parent[j]) * T + pos, dst = ((size_t)b * k + j) * T + pos;
        const int32_t hv = p.hist_in[src];
        const float lv = p.lp_in[src];
        const int32_t av = p.anc_in[src];
        p.hist_out[dst] = hv;
        p.lp_out[dst] = lv;
        p.anc_out[dst] = av;
    }
    // decoders.py:95-112 for the next step: token embedding + position t + 2 (running_seq counts from 1 and the
    // next step is t + 1), and the <pad> flag of each row
    if (p.next_x) {
        const int nvec = p.d_model >> 2, total = k * nvec;
        const f32x4* __restrict__ pos = reinterpret_cast<const f32x4*>(p.pos_emb + (size_t)(t + 2) * p.d_model);
        constexpr int kUnroll = 1024 / kThreads;           // 1024 float4 in flight per pass: k * d_model <= 4096 in one pass
        for (int base = tid; base < total; base += kThreads * kUnroll) {
            f32x4 ev[kUnroll], pv[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int idx = min(base + u * kThreads, total - 1);
                const int j = idx / nvec, c = idx - j * nvec;
                ev[u] = reinterpret_cast<const f32x4*>(p.word_emb + (size_t)word[j] * p.d_model)[c];
                pv[u] = pos[c];
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const int idx = base + u * kThreads;
                if (idx < total) {
                    const int j = idx / nvec, c = idx - j * nvec;
                    reinterpret_cast<f32x4*>(p.next_x + ((size_t)b * k + j) * p.d_model)[c] = ev[u] + pv[u];
                }
            }
        }
        if (tid < k) p.next_padflag[b * k + tid] = word[tid] == p.pad ? 1 : 0;
    }
}

// The per-beam scalars of winner `slot` (flat index f = beam * V + word, score v) of image b.
// row_max / row_lsum: the log-softmax pieces of the image's rows, indexed by the beam inside the image.
__device__ __forceinline__ void beam_record_winner(const BeamUpdateArgs& p, int b, int slot, int best_idx, float best_v,
                                                   const float* row_max, const float* row_lsum, int* parent, int* word) {
    const int k = p.k, W = p.width, V = p.V, T = p.T, t = p.t;
    // An image without a single valid region (all-zero features, e.g. the padding images of a ragged last shard)
    // has every key masked: its logits are NaN, no candidate compares greater than anything and no winner
    // exists.  The reference returns arbitrary in-range words for it; here slot j takes word j of beam 0, so
    // that every index derived from it stays in range.
    const int f = (unsigned)best_idx < (unsigned)(W * V) ? best_idx : slot;
    const int par = f / V, wd = f - par * V;
    parent[slot] = par; word[slot] = wd;
    const float alive = p.alive_in[b * W + par];
    const float x = p.logits[((size_t)b * W + par) * p.ld + wd];
    const float lp = ((x - row_max[par]) - row_lsum[par]) * alive;
    p.running_out[b * k + slot] = best_v;
    p.alive_out[b * k + slot] = alive * (wd != p.eos ? 1.0f : 0.0f);
    p.hist_out[((size_t)b * k + slot) * T + t] = wd;
    p.lp_out[((size_t)b * k + slot) * T + t] = lp;
    p.next_tok[b * k + slot] = wd;
    p.anc_out[((size_t)b * k + slot) * T + t] = b * W + par;
}

__global__ __launch_bounds__(256) void beam_update_kernel(BeamUpdateArgs p) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ int parent[kMaxK], word[kMaxK];
    if (tid < 64) {
        const Cand best = merge_row_candidates(p.cand_v, p.cand_i, b, p.width, p.k, tid);
        if (tid < p.k) beam_record_winner(p, b, tid, best.idx, best.v, p.row_max + b * p.width, p.row_lsum + b * p.width, parent, word);
        if (p.alive_count) {          // early exit: beams of this image that go on (a slot without a valid winner counts as ended)
            const bool on = tid < p.k && (unsigned)best.idx < (unsigned)(p.width * p.V) && p.alive_out[b * p.k + tid] != 0.0f;
            const int cnt = __popcll(__ballot(on));
            if (tid == 0 && cnt) atomicAdd(p.alive_count + p.t, cnt);
        }
    }
    __syncthreads();
    beam_follow_winners<256>(p, b, tid, parent, word);
}

// ---------------------------------------------------------------------------------------------------------------------
// Selection WITHOUT reading the logits back (round 3).  The vocabulary GEMM's epilogue leaves, per beam row and 32-column
// block, the block's maximum and sum exp(x - maximum) (GemmArgs::stats / stats_t: [rows][stats_ld] float2, row-major).  One workgroup per image:
//   A  the row's log-softmax pieces from its nblk pairs: M = max of the block maxima, S = sum_b s_b exp(m_b - M) in a fixed
//      order, ls = log S -- the same M the full pass finds, ls to rounding;
//   B  every block's maximum IS a candidate: u_b = run + ((m_b - M) - ls) is the score of the block's best word, computed
//      with the arithmetic of the per-word pass.  The k-th largest of one wave's lane maxima of u bounds the image's k-th
//      best from below (k distinct candidates reach it);
//   C  only blocks with u_b >= that bound can hold a winner ("hot" blocks: a handful); their 32 logits are read, scored and
//      the survivors ranked by counting (score descending, lower flat index first: the tie rule of the two-pass path);
//      a frozen beam contributes its k fixed candidates (word 0 at its running score, -999 for words 1..k-1) directly;
//   D  the bookkeeping of beam_update_kernel.
// 160 pairs per row instead of 10 201 logits; one launch instead of two.  Massive ties (a uniform row: every block hot)
// overflow the lists and take k rounds of an exhaustive arg-max over the image's width * V candidates.
constexpr int kHotCap = 512, kFusedSurvCap = 1024, kFusedThreads = 512, kFusedPairs = 4;   // 64 lanes x 4 loads x 2 blocks = 512 blocks

// 512 threads per image: wave w < width owns beam row w through phases A and B (no workgroup-level reduction), then all
// eight waves gather the hot blocks and wave 0 ranks the survivors in registers.
__global__ __launch_bounds__(kFusedThreads) void beam_fused_update_kernel(BeamUpdateArgs p, const float* __restrict__ stats, int nblk,
                                                                          int stats_ld, const float* __restrict__ running_in,
                                                                          long ld_row, long ld_word) {
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int W = p.width, k = p.k, V = p.V, T = p.T, t = p.t;
    __shared__ float rowM[kMaxK], rowLs[kMaxK], rowRun[kMaxK], thr[kMaxK];
    __shared__ int rowLive[kMaxK];
    __shared__ int nhot, nsurv;
    __shared__ unsigned short hot_blk[kHotCap];
    __shared__ uint8_t hot_row[kHotCap];
    __shared__ float surv_v[kFusedSurvCap], surv_x[kFusedSurvCap];
    __shared__ int surv_i[kFusedSurvCap];
    __shared__ float win_v[kMaxK], win_x[kMaxK];
    __shared__ int win_i[kMaxK];
    __shared__ int parent[kMaxK], word[kMaxK];

    if (tid == 0) { nhot = 0; nsurv = 0; }
    if (tid < kMaxK) { win_v[tid] = -INFINITY; win_i[tid] = 0x7fffffff; win_x[tid] = 0.f; }
    // ---- A + B: wave w = beam row w.  Lane l holds the block pairs l + 64 j (four 16-byte loads per lane, 1 KB per wave and
    //      load, all in flight together); log-softmax pieces, block bounds and the row's own k-th best bound by wave-level
    //      reductions only ---------------------------------------------------------------------------------------------------
    f32x4 st[kFusedPairs];
    float ub[kFusedPairs][2];
    float run = 0.f;
    bool live = false;
    if (wave < W) {
        const int row = b * W + wave;
        const float* srow = stats + 2 * (size_t)row * stats_ld;
#pragma unroll
        for (int j = 0; j < kFusedPairs; ++j)      // unconditional loads from clamped addresses, masked below
            st[j] = *reinterpret_cast<const f32x4*>(srow + 2 * min(2 * (lane + 64 * j), stats_ld - 2));
        run = running_in[row];
        live = p.alive_in[row] != 0.0f;
        float m = -INFINITY;
#pragma unroll
        for (int j = 0; j < kFusedPairs; ++j) {
            const int blk0 = 2 * (lane + 64 * j);
            if (blk0 >= nblk) { st[j][0] = -INFINITY; st[j][1] = 0.f; }
            if (blk0 + 1 >= nblk) { st[j][2] = -INFINITY; st[j][3] = 0.f; }
            m = fmaxf(m, fmaxf(st[j][0], st[j][2]));
        }
        const float M = wave_max_dpp(m);
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < kFusedPairs; ++j)      // a block past nblk holds (-inf, 0): 0 * exp(-inf) = 0
            sum += st[j][1] * __expf(st[j][0] - M) + st[j][3] * __expf(st[j][2] - M);
        const float ls = logf(wave_sum_dpp(sum));
        float lanemax = -INFINITY;
#pragma unroll
        for (int j = 0; j < kFusedPairs; ++j) {
            const int blk0 = 2 * (lane + 64 * j);
            // every block's maximum IS a candidate: its score, with the arithmetic of the per-word pass
            ub[j][0] = live && blk0 < nblk ? run + ((st[j][0] - M) - ls) : -INFINITY;
            ub[j][1] = live && blk0 + 1 < nblk ? run + ((st[j][2] - M) - ls) : -INFINITY;
            lanemax = fmaxf(lanemax, fmaxf(ub[j][0], ub[j][1]));
        }
        // k distinct candidates of this row reach the k-th largest lane maximum: a lower bound on the image's k-th best
        const float kth = wave_kth_largest(lanemax, k);
        if (lane == 0) {
            rowM[wave] = M; rowLs[wave] = ls; rowRun[wave] = run; rowLive[wave] = live ? 1 : 0; thr[wave] = kth;
            if (p.row_max_out) { p.row_max_out[row] = M; p.row_lsum_out[row] = ls; }
        }
    }
    __syncthreads();
    float Tthr = -INFINITY;
    for (int i = 0; i < W; ++i) Tthr = fmaxf(Tthr, thr[i]);

    // ---- C: hot blocks -> survivors -> ranks ------------------------------------------------------------------------------
    if (wave < W) {
#pragma unroll
        for (int j = 0; j < kFusedPairs; ++j)
#pragma unroll
            for (int u = 0; u < 2; ++u)
                if (ub[j][u] > -INFINITY && ub[j][u] >= Tthr) {
                    const int pos = atomicAdd(&nhot, 1);
                    if (pos < kHotCap) { hot_blk[pos] = (unsigned short)(2 * (lane + 64 * j) + u); hot_row[pos] = (uint8_t)wave; }
                }
        // a frozen beam (it has emitted <eos>) offers word 0 at its running score and -999 for every other word
        // (beam_search.py:52-55): its k best are words 0..k-1, whatever the logits are
        if (!live && lane < k && lane < V) {
            const int pos = atomicAdd(&nsurv, 1);                                          // pos < k * k <= the cap
            surv_v[pos] = lane == 0 ? run : -999.0f; surv_i[pos] = wave * V + lane; surv_x[pos] = 0.f;
        }
    }
    __syncthreads();
    const int H = nhot;
    bool exhaustive = H > kHotCap;
    if (!exhaustive) {
        // 16 hot blocks per pass; the loads of up to four passes are issued before the first survivor is appended
        constexpr int kPasses = 4, kPerPass = kFusedThreads / 32;
        for (int h0 = 0; h0 < H; h0 += kPasses * kPerPass) {
            float x[kPasses];
            int ri[kPasses], col[kPasses];
#pragma unroll
            for (int u = 0; u < kPasses; ++u) {
                const int h = min(h0 + u * kPerPass + (tid >> 5), H - 1);
                ri[u] = hot_row[h];
                col[u] = min(hot_blk[h] * 32 + (tid & 31), V - 1);
                x[u] = p.logits[(size_t)(b * W + ri[u]) * ld_row + (size_t)col[u] * ld_word];
            }
#pragma unroll
            for (int u = 0; u < kPasses; ++u) {
                const int h = h0 + u * kPerPass + (tid >> 5);
                if (h < H && hot_blk[h] * 32 + (tid & 31) < V) {
                    const float cand = rowRun[ri[u]] + ((x[u] - rowM[ri[u]]) - rowLs[ri[u]]);
                    if (cand >= Tthr) {
                        const int pos = atomicAdd(&nsurv, 1);
                        if (pos < kFusedSurvCap) { surv_v[pos] = cand; surv_i[pos] = ri[u] * V + col[u]; surv_x[pos] = x[u]; }
                    }
                }
            }
        }
        __syncthreads();
        exhaustive = nsurv > kFusedSurvCap;
    }
    if (!exhaustive) {
        const int ns = nsurv;
        if (ns <= 64) {
            // the usual case: wave 0 ranks in registers -- lane e holds survivor e and counts the survivors that beat it
            // (score descending, lower flat index first: a strict total order, so ranks are unique)
            if (wave == 0) {
                const float v = lane < ns ? surv_v[lane] : -INFINITY;
                const int idx = lane < ns ? surv_i[lane] : 0x7fffffff;
                int rank = 0;
                for (int o = 0; o < ns; ++o) {
                    const float ov = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), o));
                    const int oi = __builtin_amdgcn_readlane(idx, o);
                    rank += better(ov, oi, v, idx) ? 1 : 0;
                }
                if (lane < ns && rank < k) { win_v[rank] = v; win_i[rank] = idx; win_x[rank] = surv_x[lane]; }
            }
        } else {
            for (int e = tid; e < ns; e += kFusedThreads) {
                const float v = surv_v[e];
                const int idx = surv_i[e];
                int rank = 0;
                for (int o = 0; o < ns; ++o) rank += better(surv_v[o], surv_i[o], v, idx) ? 1 : 0;
                if (rank < k) { win_v[rank] = v; win_i[rank] = idx; win_x[rank] = surv_x[e]; }
            }
        }
    } else {
        // massive ties: k rounds of a block-wide arg-max over the candidates that come after the previous pick in the
        // (score descending, flat index ascending) order; rare, written for simplicity
        float pv = INFINITY;
        int pi = -1;
        for (int round = 0; round < k; ++round) {
            Cand c; c.v = -INFINITY; c.idx = 0x7fffffff;
            float cx = 0.f;
            for (int i = 0; i < W; ++i) {
                const float* x = p.logits + (size_t)(b * W + i) * ld_row;
                const float ri = rowRun[i], mi = rowM[i], li = rowLs[i];
                const bool alive_i = rowLive[i] != 0;
                for (int col = tid; col < V; col += kFusedThreads) {
                    const float xc = x[(size_t)col * ld_word];
                    const float cand = alive_i ? ri + ((xc - mi) - li) : (col == 0 ? ri : -999.0f);
                    const int idx = i * V + col;
                    const bool after = cand < pv || (cand == pv && idx > pi);
                    if (after && better(cand, idx, c.v, c.idx)) { c.v = cand; c.idx = idx; cx = xc; }
                }
            }
            const Cand wbest = wave_best(c);
            __syncthreads();
            if (c.idx == wbest.idx && wbest.idx != 0x7fffffff) { surv_v[wave] = c.v; surv_i[wave] = c.idx; surv_x[wave] = cx; }   // unique owner
            if (lane == 0 && wbest.idx == 0x7fffffff) { surv_v[wave] = -INFINITY; surv_i[wave] = 0x7fffffff; surv_x[wave] = 0.f; }
            __syncthreads();
            int best_w = 0;
#pragma unroll
            for (int w = 1; w < kFusedThreads / 64; ++w)
                if (better(surv_v[w], surv_i[w], surv_v[best_w], surv_i[best_w])) best_w = w;
            pv = surv_v[best_w]; pi = surv_i[best_w];
            if (tid == 0) { win_v[round] = pv; win_i[round] = pi; win_x[round] = surv_x[best_w]; }
        }
    }
    __syncthreads();

    // ---- D: bookkeeping (beam_update_kernel's, with the winner's logit carried along instead of re-read) -------------------
    if (tid < k) {
        const int f = (unsigned)win_i[tid] < (unsigned)(W * V) ? win_i[tid] : tid;      // no winner (NaN scores): as beam_record_winner
        const int par = f / V, wd = f - par * V;
        parent[tid] = par; word[tid] = wd;
        const float alive = rowLive[par] ? p.alive_in[b * W + par] : 0.0f;
        // the carried logit is the winner's own; without a winner, or for a frozen beam's fixed candidates (whose product with
        // alive = 0 only needs a finite operand), the logit is read as the two-pass path reads it
        const float x = ((unsigned)win_i[tid] < (unsigned)(W * V) && rowLive[par]) ? win_x[tid]
                                                                                   : p.logits[(size_t)(b * W + par) * ld_row + (size_t)wd * ld_word];
        const float lp = ((x - rowM[par]) - rowLs[par]) * alive;
        p.running_out[b * k + tid] = win_v[tid];
        p.alive_out[b * k + tid] = alive * (wd != p.eos ? 1.0f : 0.0f);
        p.hist_out[((size_t)b * k + tid) * T + t] = wd;
        p.lp_out[((size_t)b * k + tid) * T + t] = lp;
        p.next_tok[b * k + tid] = wd;
        p.anc_out[((size_t)b * k + tid) * T + t] = b * W + par;
    }
    if (p.alive_count && wave == 0) {      // early exit: beams of this image that go on (no valid winner = ended: NaN logits)
        const bool on = tid < k && (unsigned)win_i[tid] < (unsigned)(W * V) && rowLive[parent[tid]] &&
                        p.alive_in[b * W + parent[tid]] != 0.0f && word[tid] != p.eos;
        const int cnt = __popcll(__ballot(on));
        if (tid == 0 && cnt) atomicAdd(p.alive_count + t, cnt);
    }
    __syncthreads();
    beam_follow_winners<kFusedThreads>(p, b, tid, parent, word);
}

// masked_logp[row, c] = ((x - row_max) - row_lsum) * alive for every word (return_probs; beam_search.py:68-72) from the
// row pieces beam_fused_update_kernel published -- the values its decisions were taken on.
__global__ __launch_bounds__(256) void masked_logp_kernel(const float* __restrict__ logits, long ld_row, long ld_word,
                                                          const float* __restrict__ row_max, const float* __restrict__ row_lsum,
                                                          const float* __restrict__ alive, int V, float* __restrict__ out) {
    const int row = blockIdx.x;
    const float m = row_max[row], l = row_lsum[row], a = alive ? alive[row] : 1.0f;
    const float* x = logits + (size_t)row * ld_row;
    float* y = out + (size_t)row * V;
    for (int c = threadIdx.x; c < V; c += 256) y[c] = ((x[(size_t)c * ld_word] - m) - l) * a;
}

// Final ordering (beam_search.py:97-113): beams sorted by total score, descending, stable.
__global__ __launch_bounds__(64) void beam_finalize_kernel(BeamFinalArgs p) {
    const int b = blockIdx.x, tid = threadIdx.x, k = p.k, T = p.T;
    __shared__ int order[kMaxK];
    if (tid < k) {
        const float s = p.running[b * k + tid];
        int rank = 0;
        for (int i = 0; i < k; ++i) {
            const float o = p.running[b * k + i];
            if (o > s || (o == s && i < tid)) ++rank;
        }
        order[rank] = tid;
        if (p.order_out) p.order_out[b * k + rank] = tid;
    }
    __syncthreads();
    const int written = p.steps_run > 0 ? p.steps_run : T;     // early exit: later positions are word 0 / log-prob 0
    for (int idx = tid; idx < p.out_size * T; idx += 64) {
        const int o = idx / T, pos = idx - o * T;
        const size_t src = ((size_t)b * k + order[o]) * T + pos;
        p.ids_out[((size_t)b * p.out_size + o) * T + pos] = pos < written ? (int64_t)p.hist[src] : (int64_t)0;
        p.logp_out[((size_t)b * p.out_size + o) * T + pos] = pos < written ? p.lp[src] : 0.f;
    }
}

// all_out[b, o, t, :] = all_buf[t][b][order[b][o]][:] (t = 0: the single live beam)   (beam_search.py:68-72,103-107)
__global__ __launch_bounds__(256) void beam_gather_all_kernel(const float* __restrict__ all_buf, const int* __restrict__ order,
                                                              int B, int k, int T, int V, float* __restrict__ all_out) {
    const int b = blockIdx.x / (k * T);
    const int rem = blockIdx.x - b * k * T;
    const int o = rem / T, t = rem - o * T;
    const int beam = t == 0 ? 0 : order[b * k + o];
    // step 0 has one live beam per image: its rows are stored compactly as [B][V]
    const float* src = t == 0 ? all_buf + (size_t)b * V : all_buf + (((size_t)t * B + b) * k + beam) * V;
    float* dst = all_out + (((size_t)b * k + o) * T + t) * V;
    for (int c = threadIdx.x; c < V; c += 256) dst[c] = src[c];
}

}  // namespace

// Row pass: p.cand_v / p.cand_i [B*width][k] receive every row's k best candidates (flat index beam*V + word).
// With p.chosen set, a second tiny kernel merges them into the image's k winners (the engine's update kernel
// does that merge itself).
int ovc_beam_select_launch(const BeamSelectArgs& p, int B, hipStream_t stream) {
    if (B <= 0 || p.width <= 0 || p.width > kMaxK || p.k <= 0 || p.k > kMaxK || p.V <= 0 || !p.cand_v || !p.cand_i) return OVC_EINVAL;
    if ((long)p.width * p.V < p.k || (long)p.width * p.V > 0x7fffffffL) return OVC_EINVAL;
    const dim3 grid(B * p.width), block(kSelThreads);
    const bool vec = (p.ld & 3) == 0 && ovc_aligned16(p.logits);
#define OVC_SELECT(PT, VEC)                                                                                        \
    do {                                                                                                           \
        if (p.masked_logp) hipLaunchKernelGGL((beam_row_select_kernel<PT, VEC, true>), grid, block, 0, stream, p);  \
        else hipLaunchKernelGGL((beam_row_select_kernel<PT, VEC, false>), grid, block, 0, stream, p);               \
    } while (0)
#define OVC_SELECT_STREAMING() hipLaunchKernelGGL(beam_row_select_streaming_kernel, grid, block, 0, stream, p)
    if (vec) {
        const int per_thread = (p.V + 4 * kSelThreads - 1) / (4 * kSelThreads);      // 16-byte loads
        if (per_thread <= 1) OVC_SELECT(1, 4); else if (per_thread <= 4) OVC_SELECT(4, 4);
        else if (per_thread <= 10) OVC_SELECT(10, 4); else if (per_thread <= 16) OVC_SELECT(16, 4); else OVC_SELECT_STREAMING();
    } else {
        const int per_thread = (p.V + kSelThreads - 1) / kSelThreads;
        if (per_thread <= 4) OVC_SELECT(4, 1); else if (per_thread <= 16) OVC_SELECT(16, 1);
        else if (per_thread <= 40) OVC_SELECT(40, 1); else if (per_thread <= 64) OVC_SELECT(64, 1); else OVC_SELECT_STREAMING();
    }
#undef OVC_SELECT
#undef OVC_SELECT_STREAMING
    OVC_RETURN_IF_LAUNCH_FAILED();
    if (p.chosen) {
        hipLaunchKernelGGL(beam_merge_kernel, dim3(B), dim3(64), 0, stream, p);
        OVC_RETURN_IF_LAUNCH_FAILED();
    }
    return OVC_OK;
}

int ovc_beam_update_launch(const BeamUpdateArgs& p, int B, hipStream_t stream) {
    hipLaunchKernelGGL(beam_update_kernel, dim3(B), dim3(256), 0, stream, p);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

int ovc_beam_fused_update_launch(const BeamUpdateArgs& p, const float* stats, int nblk, int stats_ld, const float* running_in,
                                 long ld_row, long ld_word, int B, hipStream_t stream) {
    if (B <= 0 || p.width <= 0 || p.width > kMaxK || p.k <= 0 || p.k > kMaxK || p.V <= 0 || !stats || !running_in) return OVC_EINVAL;
    if (nblk != (p.V + 31) / 32 || nblk > 512 || stats_ld < nblk || (stats_ld & 1) || !ovc_aligned16(stats)) return OVC_EINVAL;
    if ((long)p.width * p.V < p.k) return OVC_EINVAL;
    if (ld_row <= 0 || ld_word <= 0) return OVC_EINVAL;
    hipLaunchKernelGGL(beam_fused_update_kernel, dim3(B), dim3(kFusedThreads), 0, stream, p, stats, nblk, stats_ld, running_in, ld_row, ld_word);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

namespace {
__global__ void collect_winners_kernel(const int32_t* anc, const int32_t* word, const float* running, int n, int k, int width, int V,
                                       int64_t* chosen, float* score) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int b = i / k;
    chosen[i] = (int64_t)(anc[i] - b * width) * V + word[i];
    score[i] = running[i];
}
}  // namespace

int ovc_debug_collect_winners_launch(const int32_t* anc, const int32_t* word, const float* running, int B, int width, int V, int k,
                                     int64_t* chosen, float* score, hipStream_t stream) {
    hipLaunchKernelGGL(collect_winners_kernel, dim3((B * k + 255) / 256), dim3(256), 0, stream, anc, word, running, B * k, k, width, V,
                       chosen, score);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

int ovc_masked_logp_launch(const float* logits, long ld_row, long ld_word, const float* row_max, const float* row_lsum,
                           const float* alive, int rows, int V, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(masked_logp_kernel, dim3(rows), dim3(256), 0, stream, logits, ld_row, ld_word, row_max, row_lsum, alive, V, out);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

int ovc_beam_finalize_launch(const BeamFinalArgs& p, int B, hipStream_t stream) {
    hipLaunchKernelGGL(beam_finalize_kernel, dim3(B), dim3(64), 0, stream, p);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

int ovc_beam_gather_all_launch(const float* all_buf, const int* order, int B, int k, int T, int V, float* all_out,
                               hipStream_t stream) {
    hipLaunchKernelGGL(beam_gather_all_kernel, dim3(B * k * T), dim3(256), 0, stream, all_buf, order, B, k, T, V, all_out);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

extern "C" int ovc_beam_select(const float* logp, const float* running, const float* alive, int B, int width, int V,
                               int k, int64_t* chosen, float* score, float* masked_logp, void* scratch,
                               size_t scratch_bytes, ovc_stream stream) {
    if (!logp || !running || !chosen || !score || B <= 0 || width <= 0 || k <= 0) return OVC_EINVAL;
    const size_t rows_k = (size_t)B * width * k;
    if (!scratch || !ovc_aligned16(scratch) || scratch_bytes < 8 * rows_k) return OVC_EWORKSPACE;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    BeamSelectArgs p{};
    p.cand_v = reinterpret_cast<float*>(scratch);
    p.cand_i = reinterpret_cast<int*>(scratch) + rows_k;
    p.logits = logp; p.ld = V; p.is_logp = 1;
    p.running = running; p.alive = alive;
    p.width = width; p.V = V; p.k = k;
    p.chosen = chosen; p.score = score; p.masked_logp = masked_logp;
    p.row_max_out = nullptr; p.row_lsum_out = nullptr;
    return ovc_beam_select_launch(p, B, ovc_hip_stream(stream));
}
