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
// log-sum-exp, the candidate scores and a k-way partial selection are fused in one pass per image:
// each thread keeps a sorted list of its k best candidates, then k rounds of a block-wide argmax
// over the list heads pick the winners.
#include "common.h"

namespace {

constexpr int kSelThreads = 1024;
constexpr int kMaxK = OVC_MAX_BEAM;

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

__global__ __launch_bounds__(kSelThreads) void beam_select_kernel(BeamSelectArgs p) {
    __shared__ float red[16];
    __shared__ float row_max[kMaxK], row_lsum[kMaxK];
    __shared__ float cand_v[16];
    __shared__ int cand_i[16];
    __shared__ int winner_idx;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const int W = p.width, V = p.V, k = p.k;
    const float* x = p.logits + (size_t)b * W * p.ld;

    // ---- log-sum-exp of every live row (skipped when the input already holds log-probabilities)
    for (int i = 0; i < W; ++i) {
        if (p.is_logp) { if (tid == 0) { row_max[i] = 0.f; row_lsum[i] = 0.f; } continue; }
        const float* xr = x + (size_t)i * p.ld;
        float mx = -INFINITY;
        for (int c = tid; c < V; c += kSelThreads) mx = fmaxf(mx, xr[c]);
        mx = wave_max(mx);
        if (lane == 0) red[wave] = mx;
        __syncthreads();
        mx = red[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) mx = fmaxf(mx, red[w]);
        __syncthreads();
        float s = 0.f;
        for (int c = tid; c < V; c += kSelThreads) s += expf(xr[c] - mx);
        s = wave_sum(s);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        if (tid == 0) {
            float tot = 0.f;
            for (int w = 0; w < 16; ++w) tot += red[w];
            row_max[i] = mx;
            row_lsum[i] = logf(tot);
        }
        __syncthreads();
    }
    __syncthreads();

    // ---- per-thread sorted top-k over the flattened (beam, word) axis -----------------------------------
    float lv[kMaxK];
    int li[kMaxK];
#pragma unroll
    for (int s = 0; s < kMaxK; ++s) { lv[s] = -INFINITY; li[s] = 0x7fffffff; }
    for (int i = 0; i < W; ++i) {
        const float run = p.running[b * W + i];
        const float alive = p.alive ? p.alive[b * W + i] : 1.0f;
        const float mx = row_max[i], ls = row_lsum[i];
        const float* xr = x + (size_t)i * p.ld;
        float* mrow = p.masked_logp ? p.masked_logp + ((size_t)b * W + i) * V : nullptr;
        for (int c = tid; c < V; c += kSelThreads) {
            const float lp = (xr[c] - mx) - ls;
            if (mrow) mrow[c] = lp * alive;
            // seq_mask * candidate + frozen * (1 - seq_mask), beam_search.py:52-55
            const float frozen = c == 0 ? run : -999.0f;
            const float cand = alive * (run + lp) + frozen * (1.0f - alive);
            if (cand > lv[k - 1] || (cand == lv[k - 1] && i * V + c < li[k - 1])) {
                float cv = cand;
                int ci = i * V + c;
#pragma unroll
                for (int s = 0; s < kMaxK; ++s) {
                    if (s < k && better(cv, ci, lv[s], li[s])) {
                        const float tv = lv[s]; const int ti = li[s];
                        lv[s] = cv; li[s] = ci; cv = tv; ci = ti;
                    }
                }
            }
        }
    }

    // ---- k rounds of block-wide argmax over the list heads ---------------------------------------------
    int head = 0;
    for (int round = 0; round < k; ++round) {
        Cand c;
        c.v = -INFINITY; c.idx = 0x7fffffff;
#pragma unroll
        for (int s = 0; s < kMaxK; ++s)
            if (s == head) { c.v = lv[s]; c.idx = li[s]; }
        c = wave_best(c);
        if (lane == 0) { cand_v[wave] = c.v; cand_i[wave] = c.idx; }
        __syncthreads();
        if (tid == 0) {
            float bv = cand_v[0]; int bi = cand_i[0];
            for (int w = 1; w < 16; ++w)
                if (better(cand_v[w], cand_i[w], bv, bi)) { bv = cand_v[w]; bi = cand_i[w]; }
            winner_idx = bi;
            p.chosen[b * k + round] = (int64_t)bi;
            p.score[b * k + round] = bv;
        }
        __syncthreads();
        bool mine = false;
#pragma unroll
        for (int s = 0; s < kMaxK; ++s)
            if (s == head && li[s] == winner_idx) mine = true;
        if (mine) ++head;
        __syncthreads();
    }
    if (tid < W && p.row_max_out) {
        p.row_max_out[b * W + tid] = row_max[tid];
        p.row_lsum_out[b * W + tid] = row_lsum[tid];
    }
}

// Per-image bookkeeping after a selection: histories, per-token log-probs, ancestor table, alive
// flags and next input tokens follow the selected beams (beam_search.py:58-81 and the state
// re-ordering of :19-34,61 expressed as an ancestor-slot table instead of cache gathers).
__global__ __launch_bounds__(64) void beam_update_kernel(BeamUpdateArgs p) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int k = p.k, W = p.width, V = p.V, T = p.T, t = p.t;
    __shared__ int parent[kMaxK], word[kMaxK];
    if (tid < k) {
        const int f = (int)p.chosen[b * k + tid];
        const int par = f / V, wd = f - par * V;
        parent[tid] = par; word[tid] = wd;
        const float alive = p.alive_in[b * W + par];
        const float x = p.logits[((size_t)b * W + par) * p.ld + wd];
        const float lp = ((x - p.row_max[b * W + par]) - p.row_lsum[b * W + par]) * alive;
        p.running_out[b * k + tid] = p.score[b * k + tid];
        p.alive_out[b * k + tid] = alive * (wd != p.eos ? 1.0f : 0.0f);
        p.hist_out[((size_t)b * k + tid) * T + t] = wd;
        p.lp_out[((size_t)b * k + tid) * T + t] = lp;
        p.next_tok[b * k + tid] = wd;
        p.anc_out[((size_t)b * k + tid) * T + t] = b * W + par;
    }
    __syncthreads();
    for (int idx = tid; idx < k * t; idx += 64) {
        const int j = idx / t, pos = idx - j * t;
        const size_t src = ((size_t)b * W + parent[j]) * T + pos, dst = ((size_t)b * k + j) * T + pos;
        p.hist_out[dst] = p.hist_in[src];
        p.lp_out[dst] = p.lp_in[src];
        p.anc_out[dst] = p.anc_in[src];
    }
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
    for (int idx = tid; idx < p.out_size * T; idx += 64) {
        const int o = idx / T, pos = idx - o * T;
        const size_t src = ((size_t)b * k + order[o]) * T + pos;
        p.ids_out[((size_t)b * p.out_size + o) * T + pos] = (int64_t)p.hist[src];
        p.logp_out[((size_t)b * p.out_size + o) * T + pos] = p.lp[src];
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

int ovc_beam_select_launch(const BeamSelectArgs& p, int B, hipStream_t stream) {
    if (B <= 0 || p.width <= 0 || p.width > kMaxK || p.k <= 0 || p.k > kMaxK || p.V <= 0) return OVC_EINVAL;
    if ((long)p.width * p.V < p.k || (long)p.width * p.V > 0x7fffffffL) return OVC_EINVAL;
    hipLaunchKernelGGL(beam_select_kernel, dim3(B), dim3(kSelThreads), 0, stream, p);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

int ovc_beam_update_launch(const BeamUpdateArgs& p, int B, hipStream_t stream) {
    hipLaunchKernelGGL(beam_update_kernel, dim3(B), dim3(64), 0, stream, p);
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
                               int k, int64_t* chosen, float* score, float* masked_logp, ovc_stream stream) {
    if (!logp || !running || !chosen || !score) return OVC_EINVAL;
    BeamSelectArgs p{};
    p.logits = logp; p.ld = V; p.is_logp = 1;
    p.running = running; p.alive = alive;
    p.width = width; p.V = V; p.k = k;
    p.chosen = chosen; p.score = score; p.masked_logp = masked_logp;
    p.row_max_out = nullptr; p.row_lsum_out = nullptr;
    return ovc_beam_select_launch(p, B, ovc_hip_stream(stream));
}
