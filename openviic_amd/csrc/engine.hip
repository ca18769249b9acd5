// The fused hot path: vision embedding + encoder + max_len beam-search steps + final ordering,
// issued as one stream of launches with no host round trip.
//
// Differences from the reference's algorithm (results identical, work not):
//   * cross-attention keys/values are projected ONCE per image and decoder layer and shared by
//     the k beams of the image; the reference replicates the encoder output k-fold, re-gathers it
//     every step and re-projects it every step (models/modules/beam_search.py:61,
//     attentions.py:47-49) -- 68 % of its FLOPs;
//   * self-attention caches PROJECTED keys/values, appended in place by the q|k|v GEMM epilogue;
//     the reference caches un-projected inputs and re-projects the whole history each step
//     (attentions.py:297-302);
//   * beam re-ordering is an ancestor-slot table (int32 [rows, T]) read by the self-attention
//     kernel; the reference physically gathers every cache (beam_search.py:19-34).
#include <algorithm>
#include <array>
#include <cstdlib>
#include <list>
#include <map>
#include <memory>
#include <atomic>
#include <mutex>
#include <tuple>
#include <vector>

#include "common.h"

namespace {

#define TRY(expr)                          \
    do {                                   \
        const int _rc = (expr);            \
        if (_rc != OVC_OK) return _rc;     \
    } while (0)
// a launch that is skipped while the engine only enumerates its GEMM shapes (Engine::dry)
#define RUN(expr)                          \
    do {                                   \
        if (!e.dry) TRY(expr);             \
    } while (0)

// ---------------------------------------------------------------------------------------------
// opt-in GEMM timing (bench.py roofline leg)
// ---------------------------------------------------------------------------------------------
struct ProfileRecord { hipEvent_t start, stop; double flops; int cls, tiling; };
struct ProfileBin { int64_t launches; double ms, flops; };
constexpr int kProfileTilings = 48;
std::atomic<bool> g_profile_on{false};
std::mutex g_profile_mutex;                      // guards everything below (several host threads may decode at once)
std::vector<ProfileRecord> g_profile;            // open records (events not yet resolved)
std::vector<ProfileRecord> g_profile_empty;      // back-to-back event pairs: the bracket's own overhead
ProfileBin g_by_class[OVC_PROFILE_CLASSES], g_by_tiling[kProfileTilings];
double g_profile_overhead_ms = 0.0;

void profile_resolve() {               // caller holds g_profile_mutex
    if (g_profile.empty() && g_profile_empty.empty()) return;
    std::vector<float> empties;
    for (ProfileRecord& r : g_profile_empty) {
        (void)hipEventSynchronize(r.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) empties.push_back(ms);
        (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop);
    }
    g_profile_empty.clear();
    if (!empties.empty()) {
        std::sort(empties.begin(), empties.end());
        g_profile_overhead_ms = empties[empties.size() / 2];
    }
    for (ProfileRecord& r : g_profile) {
        (void)hipEventSynchronize(r.stop);
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.start, r.stop) == hipSuccess) {
            const double net = ms;      // raw bracket: includes ~3 us of marker/dispatch latency per launch
            ProfileBin& c = g_by_class[r.cls];
            c.launches += 1; c.ms += net; c.flops += r.flops;
            if (r.tiling >= 0 && r.tiling < kProfileTilings) {
                ProfileBin& t = g_by_tiling[r.tiling];
                t.launches += 1; t.ms += net; t.flops += r.flops;
            }
        }
        (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop);
    }
    g_profile.clear();
}

// ---------------------------------------------------------------------------------------------
// workspace carving
// ---------------------------------------------------------------------------------------------
struct Bump {
    char* base; size_t off;
    template <typename T> T* take(size_t count) {
        off = (off + 255) & ~(size_t)255;
        T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct Workspace {
    // encoder
    uint8_t* enc_mask; float* pe; float* xe[2]; float* eq; float* ek; float* ev; float* eatt; float* ey;
    float* eff; float* einfo; float* egate; float* geometry; float* enc_levels;
    float* kx; float* vx;                         // [L][levels][B][N][h*dk|h*dv]
    // decoder
    float* x; float* x1; float* x2; float* y; float* q; float* att; float* ff; float* info; float* gate;
    float* enc_att; float* alpha; float* mixed; float* ymesh;
    float* part;                                  // [kMaxKSplit][R][d] partial outputs of K-split projections
    float* kc; float* vc;                         // [L][T][R][h*dk|h*dv]
    uint8_t* padflag;                             // [T][R]
    float* logits;                                // [R][V]
    float* stats;                                 // [R][blocks of 32 words, padded to even] float2: block maxima / sums of exponentials
    float* running[2]; float* alive[2]; int32_t* hist[2]; float* lp[2]; int32_t* anc[2];
    int32_t* tok; float* cand_v; int32_t* cand_i; float* row_max; float* row_lsum; int32_t* order;
    float* all_buf;
    int64_t* out_ids; float* out_logp;            // graph replay writes here, then copied to the caller
    int32_t* alive_count;                         // [T] beams still alive after each step (ovc_beam_search_early)
    size_t bytes;
};

bool model_ok(const ovc_model* m) {
    if (!m || m->abi != ovc_abi_version()) return false;
    if (m->n_enc < 1 || m->n_enc > OVC_MAX_LAYERS || m->n_dec < 1 || m->n_dec > OVC_MAX_LAYERS) return false;
    if (m->n_levels < 1 || m->n_levels > OVC_MAX_LEVELS) return false;
    if (m->d_model <= 0 || (m->d_model & 3) || m->d_model > 2048) return false;
    // head size: the decode attention kernels need d_k == d_v in {4, 8, 16, 32, 64} (attention.hip: the self-attention
    // reduces a head inside a power-of-two lane group), at most 32 heads and heads * d_k <= 1024
    if (m->d_k != m->d_v || m->d_k < 4 || m->d_k > 64 || (m->d_k & (m->d_k - 1))) return false;
    if ((m->d_feat & 3) || (m->d_ff & 3) || m->heads <= 0 || m->heads > 32 || m->heads * m->d_k > 1024 || m->vocab <= 1) return false;
    if (m->d_feat <= 0 || m->d_ff <= 0 || m->memory < 0) return false;
    if (m->max_len < 1 || m->max_len > 64) return false;
    if ((m->precision != 0 && m->precision != 3 && m->precision != 4) || m->tune_objective < 0 || m->tune_objective > 8) return false;
    if (m->bos_idx < 0 || m->bos_idx >= m->vocab || m->pad_idx < 0 || m->pad_idx >= m->vocab || m->eos_idx < 0 || m->eos_idx >= m->vocab) return false;
    // fused q|k|v and cross k|v GEMMs need segment widths that are multiples of the 64-wide tile
    if ((m->heads * m->d_k) % 64 || (m->heads * m->d_v) % 64 || (m->heads * m->d_k) != (m->heads * m->d_v)) return false;
    if (m->dec_kind == OVC_DEC_MESHED && m->enc_kind != OVC_ENC_MULTILEVEL) return false;
    if (m->dec_kind != OVC_DEC_MESHED && m->n_levels != 1) return false;
    // the multilevel encoder writes one level per layer (engine.hip run_encoder_layers): the meshed decoder must
    // consume exactly that many
    if (m->enc_kind == OVC_ENC_MULTILEVEL && m->n_levels != m->n_enc) return false;
    if (m->enc_kind != OVC_ENC_MULTILEVEL && m->n_levels != 1) return false;
    // products over a concatenated input [a ; b] (AoA gates, the meshed decoder's level gates) read the two blocks from their own
    // buffers: the seam has to fall on a K-tile boundary
    bool two_block = m->dec_kind == OVC_DEC_MESHED;
    for (int l = 0; l < m->n_enc; ++l) two_block = two_block || m->enc[l].att.aoa_i.w != nullptr;
    for (int l = 0; l < m->n_dec; ++l) two_block = two_block || m->dec[l].self_att.aoa_i.w != nullptr || m->dec[l].cross_att.aoa_i.w != nullptr;
    if (two_block && (m->d_model % 32)) return false;
    return true;
}

Workspace carve(const ovc_model* m, void* base, int B, int N, int k, int return_probs) {
    Workspace w{};
    Bump a{reinterpret_cast<char*>(base), 0};
    const size_t BN = (size_t)B * N, R = (size_t)B * k, d = m->d_model, T = m->max_len;
    const size_t hk = (size_t)m->heads * m->d_k, hv = (size_t)m->heads * m->d_v, lv = m->n_levels, L = m->n_dec;
    w.enc_mask = a.take<uint8_t>(BN);
    w.pe = a.take<float>((size_t)N * d);
    w.xe[0] = a.take<float>(BN * d);
    w.xe[1] = a.take<float>(BN * d);
    w.eq = a.take<float>(BN * hk); w.ek = a.take<float>(BN * hk); w.ev = a.take<float>(BN * hv);
    w.eatt = a.take<float>(BN * hv);
    w.ey = a.take<float>(BN * d);
    w.eff = a.take<float>(BN * m->d_ff);
    w.einfo = a.take<float>(BN * d); w.egate = a.take<float>(BN * d);
    w.geometry = a.take<float>(m->enc_kind == OVC_ENC_GEOMETRIC ? (size_t)B * m->heads * N * N : 0);
    w.enc_levels = a.take<float>(lv * BN * d);
    w.kx = a.take<float>(L * lv * BN * hk);
    w.vx = a.take<float>(L * lv * BN * hv);
    w.x = a.take<float>(R * d); w.x1 = a.take<float>(R * d); w.x2 = a.take<float>(R * d); w.y = a.take<float>(R * d);
    w.q = a.take<float>(R * hk);
    w.att = a.take<float>(lv * R * hv);
    w.ff = a.take<float>(R * m->d_ff);
    w.info = a.take<float>(R * d); w.gate = a.take<float>(R * d);
    w.enc_att = a.take<float>(lv * R * d); w.alpha = a.take<float>(lv * R * d); w.mixed = a.take<float>(R * d);
    w.ymesh = a.take<float>(lv > 1 ? lv * R * d : 0);
    w.part = a.take<float>(kMaxKSplit * R * d);
    w.kc = a.take<float>(L * T * R * hk);
    w.vc = a.take<float>(L * T * R * hv);
    w.padflag = a.take<uint8_t>(T * R);
    w.logits = a.take<float>(((R + 3) & ~(size_t)3) * (((size_t)m->vocab + 3) & ~(size_t)3));   // [R][V] or [V][R], rows padded to 16 bytes
    w.stats = a.take<float>(2 * ((((size_t)m->vocab + 31) / 32 + 1) & ~(size_t)1) * R);
    for (int i = 0; i < 2; ++i) {
        w.running[i] = a.take<float>(R); w.alive[i] = a.take<float>(R);
        w.hist[i] = a.take<int32_t>(R * T); w.lp[i] = a.take<float>(R * T); w.anc[i] = a.take<int32_t>(R * T);
    }
    w.tok = a.take<int32_t>(R); w.cand_v = a.take<float>(R * (size_t)k); w.cand_i = a.take<int32_t>(R * (size_t)k);
    w.row_max = a.take<float>(R); w.row_lsum = a.take<float>(R); w.order = a.take<int32_t>(R);
    w.all_buf = a.take<float>(return_probs ? T * R * (size_t)m->vocab : 0);
    w.out_ids = a.take<int64_t>(R * T); w.out_logp = a.take<float>(R * T);
    w.alive_count = a.take<int32_t>(T);
    w.bytes = (a.off + 255) & ~(size_t)255;
    return w;
}

// ---------------------------------------------------------------------------------------------
// small engine-only kernels
// ---------------------------------------------------------------------------------------------
// x[r,:] = word_emb[tok[r]] + pos_emb[t+1]; padflag[r] = (tok[r] == pad).  decoders.py:95-112 in
// stateful mode: the position is running_seq = t+1 for every row, also for <pad> rows.
__global__ __launch_bounds__(256) void decode_embed_kernel(const int32_t* __restrict__ tok, int bos, int pad, int t,
                                                           const float* __restrict__ table, const float* __restrict__ pos_table,
                                                           float* __restrict__ x, uint8_t* __restrict__ padflag, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int token = t == 0 ? bos : tok[row];
    if (lane == 0) padflag[row] = token == pad ? 1 : 0;
    const f32x4* e = reinterpret_cast<const f32x4*>(table + (size_t)token * d);
    const f32x4* p = reinterpret_cast<const f32x4*>(pos_table + (size_t)(t + 1) * d);
    f32x4* o = reinterpret_cast<f32x4*>(x + (size_t)row * d);
    for (int c = lane; c < (d >> 2); c += 64) o[c] = e[c] + p[c];
}

// Measurement hook (OVC_DEBUG_SKIP=mask): leave out classes of decode-step launches to see what each costs with several
// batches in flight (results are garbage; timing only).  1 = AddNorm LayerNorms, 2 = self-attention, 4 = cross-attention,
// 8 = the vocabulary projection's selection / update.  Never set in production.
int debug_skip() {
    static const int mask = [] { const char* e = OVC_HOOK_ENV("OVC_DEBUG_SKIP"); return e ? atoi(e) : 0; }();
    return mask;
}

// Measurement hook (OVC_DEBUG_EXTRA_LAUNCHES=n): n empty launches behind every AddNorm LayerNorm of the decode step -- how much
// does a kernel BOUNDARY cost the whole chip when several streams are in flight?  Never set in production.
__global__ void noop_kernel() {}
int extra_launches() {
    static const int n = [] { const char* e = OVC_HOOK_ENV("OVC_DEBUG_EXTRA_LAUNCHES"); return e ? atoi(e) : 0; }();
    return n;
}

__global__ void init_beam_state_kernel(float* running, float* alive, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { running[i] = 0.f; alive[i] = 1.f; }
}

// out[b, lvl, n, :] = levels[lvl][b][n][:]
__global__ void interleave_levels_kernel(const float* __restrict__ levels, float* __restrict__ out, int B, int lv, size_t nd4) {
    const size_t total = (size_t)B * lv * nd4;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i % nd4, bl = i / nd4;
        const int l = (int)(bl % lv), b = (int)(bl / lv);
        reinterpret_cast<f32x4*>(out)[i] = reinterpret_cast<const f32x4*>(levels)[((size_t)l * B + b) * nd4 + e];
    }
}

// ---------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------
// K slices of the M = B*k projections back to d_model whose consumer is the AddNorm LayerNorm (it sums the slices in
// order and applies bias and residual).  Like GemmArgs::kchains this is part of the product's definition -- the slices
// are summed in a fixed order -- so it is a pure function of K, never of a timing or of M: 2 slices below K = 1024,
// 4 from there on (measured on 1280 x 512 x {512, 2048}: 12.3 -> 11.1 us and 37.3 -> 27.2 us).  OVC_KSPLIT_SMALL /
// OVC_KSPLIT_LARGE override the two values for A/B measurements (they change the summation order, hence low-order bits).
int decode_ksplit(int K) {
    static const int small = [] { const char* e = OVC_HOOK_ENV("OVC_KSPLIT_SMALL"); return e ? atoi(e) : 2; }();
    static const int large = [] { const char* e = OVC_HOOK_ENV("OVC_KSPLIT_LARGE"); return e ? atoi(e) : 4; }();
    int s = K >= 1024 ? large : small;
    if (s != 1 && s != 2 && s != 4) s = 1;
    while (s > 1 && K % (s * 32)) s >>= 1;          // a slice is a whole number of 32-deep K tiles
    return s;
}

using GemmShape = std::array<int, 7>;               // M, seg_n, nseg, K, kchains, ksplit, epilogue (0 plain, 1 stats, 2 stats_t)

struct Engine {
    const ovc_model* m;
    hipStream_t stream;
    int gemm_class;                                  // profiling bin (OVC_PROFILE_CLASSES)
    int kchains = 1;                                 // K-order class of the GEMMs issued next (set per call site group)
    std::vector<GemmShape>* dry = nullptr;           // shape enumeration: record every GEMM, launch nothing

    // A weight segment; in the split-precision modes with the weight's pre-cut planes (ovc_lin::planes) when the host built them
    GemmSegment seg(const ovc_lin& l, float* C, const float* A2 = nullptr) const {
        return GemmSegment{l.w, l.b, C, A2, m->precision > 0 ? l.planes : nullptr};
    }

    int gemm(GemmArgs& a) {
        a.kchains = m->precision > 0 ? 100 + m->precision : kchains;   // opt-in split precision: its own K-order classes
        // fp16 planes cannot hold what lies outside fp16's range.  Weights are checked when the mode is selected, activations
        // behind a LayerNorm / softmax / ReLU of bounded operands are bounded -- the caller's features are not: their
        // projection takes the three-plane bf16 class, whose planes have fp32's exponent range.
        if (m->precision == 4 && gemm_class == 0) a.kchains = 103;
        a.objective = m->tune_objective;
        if (dry) {
            const GemmShape sh{a.M, a.seg_n, a.nseg, a.K1 + a.K2, a.kchains, a.ksplit > 1 ? a.ksplit : 1, a.stats ? 1 : (a.stats_t ? 2 : 0)};
            if (std::find(dry->begin(), dry->end(), sh) == dry->end()) dry->push_back(sh);
            return OVC_OK;
        }
        if (!g_profile_on.load()) return ovc_gemm_launch(a, stream);
        std::lock_guard<std::mutex> lock(g_profile_mutex);
        if (g_profile.empty() && g_profile_empty.empty()) {
            // calibrate the bracket: event pairs with nothing in between
            for (int i = 0; i < 16; ++i) {
                ProfileRecord e{};
                if (hipEventCreate(&e.start) != hipSuccess || hipEventCreate(&e.stop) != hipSuccess) return OVC_ELAUNCH;
                (void)hipEventRecord(e.start, stream);
                (void)hipEventRecord(e.stop, stream);
                g_profile_empty.push_back(e);
            }
        }
        ProfileRecord rec{};
        if (hipEventCreate(&rec.start) != hipSuccess || hipEventCreate(&rec.stop) != hipSuccess) return OVC_ELAUNCH;
        rec.flops = 2.0 * a.M * (double)a.seg_n * a.nseg * (a.K1 + a.K2);
        rec.cls = gemm_class;
        rec.tiling = ovc_gemm_pick_tiling(a);
        // kernel-scoped events: the dispatch's own begin / end timestamps (no marker latency in between)
        GemmLaunchOpts opts{};
        opts.start = rec.start; opts.stop = rec.stop;
        const int rc = ovc_gemm_launch(a, stream, opts);
        g_profile.push_back(rec);
        return rc;
    }

    // y = act(x W^T + b) + residual
    int linear(const float* x, int K, const ovc_lin& l, const float* residual, float* y, int M, int N, int act) {
        GemmArgs a{};
        a.A1 = x; a.lda1 = K; a.K1 = K; a.M = M; a.seg_n = N; a.nseg = 1; a.ldc = N;
        a.R = residual; a.ldr = N; a.act = act;
        a.seg[0] = seg(l, y);
        return gemm(a);
    }

    // out = LayerNorm(x W^T + b + residual), rows flagged in zero_rows cleared.  With a partial-product buffer (the
    // M = B*k decode-step projections back to d_model) the GEMM runs as decode_ksplit(K) slices writing raw partial
    // products; the LayerNorm kernel sums them in slice order and applies bias and residual.
    int linear_ln(const float* x, int K, const ovc_lin& l, const float* residual, const ovc_norm& ln,
                  const uint8_t* zero_rows, float* y_tmp, float* part, float* out, int M) {
        const int d = m->d_model;
        const int split = part && l.b && residual ? decode_ksplit(K) : 1;
        if (split != 2 && split != 4) {
            TRY(linear(x, K, l, residual, y_tmp, M, d, 0));
            if (dry) return OVC_OK;
            return ovc_layer_norm(y_tmp, nullptr, ln.g, ln.b, nullptr, 0, zero_rows, m->ln_eps, out, M, d, stream);
        }
        GemmArgs a{};
        a.A1 = x; a.lda1 = K; a.K1 = K; a.M = M; a.seg_n = d; a.nseg = 1; a.ldc = d;
        a.ksplit = split; a.part_stride = (long)M * d;
        a.seg[0] = seg(l, part); a.seg[0].bias = nullptr;     // raw partial products: bias applied by the consumer
        TRY(gemm(a));
        if (dry) return OVC_OK;
        if (!(debug_skip() & 1))
            TRY(ovc_layer_norm_parts(part, split, a.part_stride, l.b, residual, ln.g, ln.b, zero_rows, m->ln_eps, out, M, d, stream));
        for (int i = 0; i < extra_launches(); ++i) hipLaunchKernelGGL(noop_kernel, dim3(1), dim3(64), 0, stream);
        return OVC_OK;
    }

    // AoA gate (attentions.py:311-315): out = W_i [q; x] * sigmoid(W_g [q; x]), one two-segment GEMM.
    int aoa(const ovc_mha& w, const float* queries, float* x, float* info, float* gate, int M) {
        if (!w.aoa_i.w) return OVC_OK;
        const int d = m->d_model;
        GemmArgs a{};
        a.A1 = queries; a.lda1 = d; a.K1 = d; a.A2 = x; a.lda2 = d; a.K2 = d;
        a.M = M; a.seg_n = d; a.nseg = 2; a.ldc = d;
        a.seg[0] = seg(w.aoa_i, info);
        a.seg[1] = seg(w.aoa_g, gate);
        if (d % 64) {   // segments must align with tiles: fall back to two launches
            a.nseg = 1;
            TRY(gemm(a));
            a.seg[0] = seg(w.aoa_g, gate);
            TRY(gemm(a));
        } else {
            TRY(gemm(a));
        }
        if (dry) return OVC_OK;
        return ovc_sigmoid_gate(info, gate, x, (long)M * d, stream);
    }

    int ffn(const ovc_ffn& w, const float* x, float* ff, float* y, float* part, float* out, const uint8_t* zero_rows, int M) {
        TRY(linear(x, m->d_model, w.fc1, nullptr, ff, M, m->d_ff, 1));
        return linear_ln(ff, m->d_ff, w.fc2, x, w.ln, zero_rows, y, part, out, M);
    }
};

// The kernels that read caller-owned inputs (features, boxes): kept outside the captured graph, whose
// nodes may only reference the workspace and the weights.
int run_encoder_inputs(Engine& e, Workspace& w, const float* features, const float* boxes, int B, int N) {
    const ovc_model* m = e.m;
    const int BN = B * N, d = m->d_model;
    hipStream_t s = e.stream;
    if (!e.dry && m->enc_kind == OVC_ENC_GEOMETRIC && (!boxes || !m->fc_g_w || !m->fc_g_b)) return OVC_EINVAL;
    e.gemm_class = 0;
    e.kchains = 1;            // M = B*N products: one summation chain (gemm.hip, K-order classes)
    // K1 (models/utils.py:48-61): the padding mask is the row sum of the features.  In the fp32 mode the feature projection
    // finds it while it stages its A tiles (GemmArgs::zero_rows_out): one pass over the caller's 105 MB instead of two.
    {
        GemmArgs a{};
        a.A1 = features; a.lda1 = m->d_feat; a.K1 = m->d_feat; a.M = BN; a.seg_n = d; a.nseg = 1; a.ldc = d;
        a.seg[0] = e.seg(m->proj, w.ey);
        static const bool separate = OVC_HOOK_ENV("OVC_K1_SEPARATE") != nullptr;        // A/B switch: the round-1 mask kernel
        if (m->precision == 0 && !separate) a.zero_rows_out = w.enc_mask;
        else RUN(ovc_zero_row_mask(features, BN, m->d_feat, w.enc_mask, s));
        TRY(e.gemm(a));
    }
    if (m->enc_kind == OVC_ENC_GEOMETRIC)
        RUN(ovc_box_relation_weights(boxes, B, N, m->fc_g_w, m->fc_g_b, m->heads, m->d_g, m->trig, w.geometry, s));
    return OVC_OK;
}

int run_encoder_layers(Engine& e, Workspace& w, int B, int N) {
    const ovc_model* m = e.m;
    const int BN = B * N, d = m->d_model, hk = m->heads * m->d_k, hv = m->heads * m->d_v;
    hipStream_t s = e.stream;
    e.gemm_class = 1;
    e.kchains = 1;
    RUN(ovc_region_position_encoding(nullptr, 1, N, d, 10000.0f, 0, 0.f, w.pe, s));
    RUN(ovc_layer_norm(w.ey, nullptr, m->enc_ln.g, m->enc_ln.b, w.pe, N, nullptr, m->ln_eps, w.xe[0], BN, d, s));

    float* x = w.xe[0];
    float* x1 = w.xe[1];
    for (int l = 0; l < m->n_enc; ++l) {
        const ovc_mha& at = m->enc[l].att;
        GemmArgs a{};
        a.A1 = x; a.lda1 = d; a.K1 = d; a.M = BN; a.seg_n = hk; a.nseg = 3; a.ldc = hk;
        a.seg[0] = e.seg(at.q, w.eq);
        a.seg[1] = e.seg(at.k, w.ek);
        a.seg[2] = e.seg(at.v, w.ev);
        TRY(e.gemm(a));
        const int mem = at.m_k ? m->memory : 0;
        RUN(ovc_attention(w.eq, w.ek, w.ev, B, N, N, m->heads, m->d_k, m->d_v, w.enc_mask, N, 0,
                          m->enc_kind == OVC_ENC_GEOMETRIC ? w.geometry : nullptr, at.m_k, at.m_v, mem,
                          sqrtf((float)m->d_k), sqrtf((float)(mem > 0 ? mem : 1)), w.eatt, s));
        TRY(e.linear(w.eatt, hv, at.o, x, w.ey, BN, d, 0));
        RUN(ovc_layer_norm(w.ey, nullptr, at.ln.g, at.ln.b, nullptr, 0, nullptr, m->ln_eps, x1, BN, d, s));
        TRY(e.aoa(at, x, x1, w.einfo, w.egate, BN));
        // layer output: straight into the level slot (multilevel) or the ping-pong buffer
        float* out = m->enc_kind == OVC_ENC_MULTILEVEL ? w.enc_levels + (size_t)l * BN * d
                                                       : (l == m->n_enc - 1 ? w.enc_levels : x);
        TRY(e.ffn(m->enc[l].ffn, x1, w.eff, w.ey, nullptr, out, w.enc_mask, BN));
        x = out;
    }
    return OVC_OK;
}

int run_encoder(Engine& e, Workspace& w, const float* features, const float* boxes, int B, int N) {
    TRY(run_encoder_inputs(e, w, features, boxes, B, N));
    return run_encoder_layers(e, w, B, N);
}

// Projected cross-attention keys/values of every decoder layer: one GEMM per encoder level with
// 2*L segments (k_0, v_0, k_1, v_1, ...) sharing the encoder output as the A operand.
int project_cross_kv(Engine& e, Workspace& w, int B, int N) {
    const ovc_model* m = e.m;
    const int BN = B * N, d = m->d_model, hk = m->heads * m->d_k, lv = m->n_levels, L = m->n_dec;
    e.gemm_class = 1;
    e.kchains = 1;
    for (int lvl = 0; lvl < lv; ++lvl) {
        for (int l0 = 0; l0 < L; l0 += OVC_MAX_SEGMENTS / 2) {
            GemmArgs a{};
            a.A1 = w.enc_levels + (size_t)lvl * BN * d; a.lda1 = d; a.K1 = d; a.M = BN; a.seg_n = hk; a.ldc = hk;
            int ns = 0;
            for (int l = l0; l < L && ns + 2 <= OVC_MAX_SEGMENTS; ++l) {
                const ovc_mha& at = m->dec[l].cross_att;
                const size_t off = ((size_t)l * lv + lvl) * BN * hk;
                a.seg[ns++] = e.seg(at.k, w.kx + off);
                a.seg[ns++] = e.seg(at.v, w.vx + off);
            }
            a.nseg = ns;
            TRY(e.gemm(a));
        }
    }
    return OVC_OK;
}

int run_decode_step(Engine& e, Workspace& w, int B, int N, int k, int t, int return_probs, bool count_alive = false) {
    const ovc_model* m = e.m;
    hipStream_t s = e.stream;
    const int d = m->d_model, hk = m->heads * m->d_k, hv = m->heads * m->d_v, lv = m->n_levels, T = m->max_len;
    const int R = B * k, width = t == 0 ? 1 : k, rows = B * width;
    const int cur = t & 1, nxt = cur ^ 1;
    uint8_t* padflag_t = w.padflag + (size_t)t * R;

    if (t == 0 && !e.dry) {       // later steps: the previous step's update kernel has written the input rows and pad flags
        hipLaunchKernelGGL(decode_embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, w.tok, m->bos_idx, m->pad_idx, t,
                           m->word_emb, m->pos_emb, w.x, padflag_t, rows, d);
        OVC_RETURN_IF_LAUNCH_FAILED();
    }

    e.gemm_class = 2;
    e.kchains = 4;            // M = B*width products: four chains, so that 32x32 / 32x64 tiles can spread them over waves
    {   // measurement hook: another K-order class for the decode-step products (changes low-order bits; never set in production)
        static const int forced = [] { const char* v = OVC_HOOK_ENV("OVC_DEBUG_DECODE_KCHAINS"); return v ? atoi(v) : 0; }();
        if (forced == 1 || forced == 4) e.kchains = forced;
    }
    float* x = w.x;
    for (int l = 0; l < m->n_dec; ++l) {
        const ovc_dec_layer& dl = m->dec[l];
        // ---- masked self-attention over the beam's own history ------------------------------------
        float* kc = w.kc + (size_t)l * T * R * hk;
        float* vc = w.vc + (size_t)l * T * R * hv;
        GemmArgs a{};
        a.A1 = x; a.lda1 = d; a.K1 = d; a.M = rows; a.seg_n = hk; a.nseg = 3; a.ldc = hk;
        a.seg[0] = e.seg(dl.self_att.q, w.q);
        a.seg[1] = e.seg(dl.self_att.k, kc + (size_t)t * R * hk);
        a.seg[2] = e.seg(dl.self_att.v, vc + (size_t)t * R * hv);
        TRY(e.gemm(a));
        DecodeSelfArgs sa{};
        sa.q = w.q; sa.ldq = hk; sa.kcache = kc; sa.vcache = vc; sa.pos_stride = (size_t)R * hk; sa.ldkv = hk;
        sa.anc = w.anc[cur]; sa.anc_ld = T; sa.padflag = w.padflag; sa.pad_ld = R; sa.t = t; sa.width = width;
        sa.h = m->heads; sa.dk = m->d_k; sa.dv = m->d_v; sa.out = w.att; sa.ldo = hv;
        if (!(debug_skip() & 2)) RUN(ovc_decode_self_attention(sa, rows, s));
        TRY(e.linear_ln(w.att, hv, dl.self_att.o, x, dl.self_att.ln, nullptr, w.y, w.part, w.x1, rows));
        TRY(e.aoa(dl.self_att, x, w.x1, w.info, w.gate, rows));

        // ---- cross-attention: the image's beams share its projected encoder keys/values -----------
        TRY(e.linear(w.x1, d, dl.cross_att.q, nullptr, w.q, rows, hk, 0));
        DecodeCrossArgs ca{};
        ca.q = w.q; ca.ldq = hk;
        ca.kx = w.kx + (size_t)l * lv * B * N * hk; ca.vx = w.vx + (size_t)l * lv * B * N * hv;
        ca.level_stride = (size_t)B * N * hk; ca.ldkv = hk; ca.encmask = w.enc_mask; ca.n = N; ca.width = width;
        ca.heads = m->heads; ca.dk = m->d_k; ca.dv = m->d_v; ca.out = w.att; ca.out_level_stride = (size_t)rows * hv; ca.ldo = hv;
        if (!(debug_skip() & 4)) RUN(ovc_decode_cross_attention(ca, B, m->heads, lv, s));
        float* ffn_in;
        if (m->dec_kind == OVC_DEC_MESHED) {
            // decoders.py:51-73: one shared enc_attn per level, sigmoid-gated sum / sqrt(levels).  The levels'
            // attention outputs are stacked [levels][rows][h*dv], so the shared output projection and its
            // LayerNorm run once over levels*rows rows (the residual x1 is broadcast with res_mod).
            const size_t nrd = (size_t)rows * d;
            if (!dl.cross_att.aoa_i.w) {
                GemmArgs o{};
                o.A1 = w.att; o.lda1 = hv; o.K1 = hv; o.M = lv * rows; o.seg_n = d; o.nseg = 1; o.ldc = d;
                o.R = w.x1; o.ldr = d; o.res_mod = rows;
                o.seg[0] = e.seg(dl.cross_att.o, w.ymesh);
                TRY(e.gemm(o));
                RUN(ovc_layer_norm(w.ymesh, nullptr, dl.cross_att.ln.g, dl.cross_att.ln.b, nullptr, 0, nullptr, m->ln_eps,
                                   w.enc_att, lv * rows, d, s));
            } else {
                for (int lvl = 0; lvl < lv; ++lvl) {      // AoA gates need the per-level pair (x1, enc_att_l)
                    TRY(e.linear(w.att + (size_t)lvl * rows * hv, hv, dl.cross_att.o, w.x1, w.y, rows, d, 0));
                    RUN(ovc_layer_norm(w.y, nullptr, dl.cross_att.ln.g, dl.cross_att.ln.b, nullptr, 0, nullptr, m->ln_eps,
                                       w.enc_att + lvl * nrd, rows, d, s));
                    TRY(e.aoa(dl.cross_att, w.x1, w.enc_att + lvl * nrd, w.info, w.gate, rows));
                }
            }
            // level gates alpha_l = W_l [x1 ; enc_att_l] + b_l (decoders.py:59-66): one launch, a segment per level,
            // each with its own second input block (separate launches when the width does not align with the tiles)
            {
                GemmArgs g{};
                g.A1 = w.x1; g.lda1 = d; g.K1 = d; g.lda2 = d; g.K2 = d;
                g.M = rows; g.seg_n = d; g.ldc = d;
                const bool fused = d % 64 == 0 && lv <= OVC_MAX_SEGMENTS;
                for (int lvl = 0; lvl < lv; ++lvl) {
                    const GemmSegment seg = e.seg(dl.alpha[lvl], w.alpha + lvl * nrd, w.enc_att + lvl * nrd);
                    if (fused) { g.seg[lvl] = seg; continue; }
                    g.seg[0] = seg; g.nseg = 1;
                    TRY(e.gemm(g));
                }
                if (fused) { g.nseg = lv; TRY(e.gemm(g)); }
            }
            RUN(ovc_meshed_mix(w.alpha, w.enc_att, lv, (long)nrd, sqrtf((float)lv), w.mixed, s));
            ffn_in = w.mixed;
        } else {
            TRY(e.linear_ln(w.att, hv, dl.cross_att.o, w.x1, dl.cross_att.ln, nullptr, w.y, w.part, w.x2, rows));
            TRY(e.aoa(dl.cross_att, w.x1, w.x2, w.info, w.gate, rows));
            ffn_in = w.x2;
        }
        TRY(e.ffn(dl.ffn, ffn_in, w.ff, w.y, w.part, w.x, padflag_t, rows));
        x = w.x;
    }

    // ---- vocabulary projection, fused log-softmax + candidate scores + top-k, bookkeeping ---------
    e.gemm_class = 3;
    const int ldv = (m->vocab + 3) & ~3;            // 16-byte aligned logit rows: vector loads in the selection kernel
    // Fused selection (round 3): the GEMM epilogue leaves per (row, 32-column block) the maximum and the sum of exponentials,
    // and ONE kernel per image selects and updates from those pieces.  Vocabularies beyond 16 384 words (more than 512 blocks)
    // and OVC_SELECT_TWO_PASS (A/B switch) take the round-2 pair of kernels that read every logit back.
    const int nblk = (m->vocab + 31) / 32;
    static const bool two_pass = OVC_HOOK_ENV("OVC_SELECT_TWO_PASS") != nullptr;
    const bool fused_select = !two_pass && nblk <= 512;
    // fp32 mode + fused selection: the product runs TRANSPOSED -- logits^T [V][rows] = fc [V, d] . x^T, the same kernel with the
    // operands' roles swapped (both are K-contiguous) and the same bits (every dot product sums the same k order; a * b
    // commutes).  A lane of the accumulator then holds 16 WORDS of one beam row, which makes the block maximum / sum exp an
    // in-register reduction (~70 vector instructions per tile instead of ~370 across lanes), and a store instruction still
    // writes whole 128-byte lines (32 consecutive beam rows of one word).  The split-precision modes keep the row-major
    // product: their pre-cut weight planes are B-operand planes.
    static const bool row_major = OVC_HOOK_ENV("OVC_VOCAB_ROW_MAJOR") != nullptr;        // A/B switch
    const bool transposed = fused_select && m->precision == 0 && !row_major;
    const int ldt = (rows + 3) & ~3;               // row stride of logits^T
    {
        // K-order class of THIS call site: the transposed product has M = V rows -- thousands of output tiles whatever the batch
        // -- so it needs no chains spread over waves and takes the one-chain class of the other large-M products (a quarter of
        // the accumulator registers: +1.2 % captions/s with four batches in flight, +0.5 % on one stream, same-box A/B).  The
        // row-major form of the split-precision modes keeps its class.  OVC_DEBUG_VOCAB_KCHAINS=4: A/B switch (changes the
        // logits' low-order bits).
        static const int vocab_chains = [] { const char* v = OVC_HOOK_ENV("OVC_DEBUG_VOCAB_KCHAINS"); return v ? atoi(v) : 0; }();
        const int saved_chains = e.kchains;
        if (transposed) e.kchains = vocab_chains == 4 ? 4 : 1;
        GemmArgs g{};
        if (transposed) {
            g.A1 = m->fc; g.lda1 = d; g.K1 = d; g.M = m->vocab; g.seg_n = rows; g.nseg = 1; g.ldc = ldt;
            g.seg[0] = GemmSegment{x, nullptr, w.logits, nullptr, nullptr};
            g.stats_t = w.stats; g.stats_ld = (nblk + 1) & ~1;
        } else {
            g.A1 = x; g.lda1 = d; g.K1 = d; g.M = rows; g.seg_n = m->vocab; g.nseg = 1; g.ldc = ldv;
            g.seg[0] = GemmSegment{m->fc, nullptr, w.logits, nullptr, m->precision > 0 ? m->fc_planes : nullptr};
            g.stats = fused_select ? w.stats : nullptr; g.stats_ld = (nblk + 1) & ~1;
        }
        TRY(e.gemm(g));
        e.kchains = saved_chains;
    }
    const long ld_row = transposed ? 1 : ldv, ld_word = transposed ? ldt : 1;
    BeamUpdateArgs bu{};
    bu.cand_v = w.cand_v; bu.cand_i = w.cand_i; bu.logits = w.logits; bu.ld = ldv;
    bu.row_max = w.row_max; bu.row_lsum = w.row_lsum;
    bu.alive_in = w.alive[cur]; bu.alive_out = w.alive[nxt]; bu.running_out = w.running[nxt];
    bu.hist_in = w.hist[cur]; bu.hist_out = w.hist[nxt]; bu.lp_in = w.lp[cur]; bu.lp_out = w.lp[nxt];
    bu.anc_in = w.anc[cur]; bu.anc_out = w.anc[nxt]; bu.next_tok = w.tok;
    bu.width = width; bu.k = k; bu.V = m->vocab; bu.T = T; bu.t = t; bu.eos = m->eos_idx;
    bu.alive_count = count_alive ? w.alive_count : nullptr;
    if (t + 1 < T) {
        bu.word_emb = m->word_emb; bu.pos_emb = m->pos_emb; bu.next_x = w.x; bu.next_padflag = w.padflag + (size_t)(t + 1) * R;
        bu.d_model = d; bu.pad = m->pad_idx;
    }
    if (fused_select) {
        // selection + bookkeeping in one launch, from the block pieces the vocabulary GEMM's epilogue left: no pass over the logits
        bu.row_max_out = return_probs ? w.row_max : nullptr; bu.row_lsum_out = return_probs ? w.row_lsum : nullptr;
        if (!(debug_skip() & 8)) RUN(ovc_beam_fused_update_launch(bu, w.stats, nblk, (nblk + 1) & ~1, w.running[cur], ld_row, ld_word, B, s));
        if (return_probs)     // beam_search.py:68-72: every word's masked log-probability, from the pieces the decisions used
            RUN(ovc_masked_logp_launch(w.logits, ld_row, ld_word, w.row_max, w.row_lsum, w.alive[cur], rows, m->vocab,
                                       w.all_buf + (size_t)t * R * m->vocab, s));
        return OVC_OK;
    }
    BeamSelectArgs bs{};
    bs.logits = w.logits; bs.ld = ldv; bs.is_logp = 0;
    bs.running = w.running[cur]; bs.alive = w.alive[cur]; bs.width = width; bs.V = m->vocab; bs.k = k;
    bs.cand_v = w.cand_v; bs.cand_i = w.cand_i; bs.chosen = nullptr; bs.score = nullptr;   // merged by the update kernel
    bs.masked_logp = return_probs ? w.all_buf + (size_t)t * R * m->vocab : nullptr;
    bs.row_max_out = w.row_max; bs.row_lsum_out = w.row_lsum;
    RUN(ovc_beam_select_launch(bs, B, s));
    RUN(ovc_beam_update_launch(bu, B, s));
    return OVC_OK;
}

}  // namespace

extern "C" int ovc_abi_version(void) { return 7; }     // 7: ovc_beam_search_early, ovc_debug_vocab_select(kchains), OVC_MAX_REGIONS

extern "C" const char* ovc_build_info(void) {
#ifdef OVC_MEASUREMENT_HOOKS
    // a tools/ build: OVC_DEBUG_* / OVC_KSPLIT_* / the A/B switches are read from the environment -- never for results or a credited number
    return "libovc gfx950 (CDNA4) fp32: v_mfma_f32_32x32x2_f32 GEMM + attention, HIP " __DATE__ " +measurement-hooks";
#else
    return "libovc gfx950 (CDNA4) fp32: v_mfma_f32_32x32x2_f32 GEMM + attention, HIP " __DATE__;
#endif
}

extern "C" size_t ovc_workspace_bytes(const ovc_model* m, int B, int N, int k, int return_probs) {
    if (!model_ok(m) || B <= 0 || N <= 0 || N > OVC_MAX_REGIONS || k <= 0 || k > OVC_MAX_BEAM) return 0;
    return carve(m, nullptr, B, N, k, return_probs).bytes;
}

// Every distinct GEMM the engine issues for (B, N, k), found by running the launch sequence itself in dry mode (no
// launch, no device access: the workspace is carved at a fake base address that is never dereferenced).
extern "C" int ovc_engine_gemm_shapes(const ovc_model* m, int B, int N, int k, int32_t* shapes, int capacity) {
    if (!model_ok(m) || B <= 0 || N <= 0 || N > OVC_MAX_REGIONS || k <= 0 || k > OVC_MAX_BEAM || capacity < 0 || (capacity > 0 && !shapes)) return OVC_EINVAL;
    Workspace w = carve(m, reinterpret_cast<void*>(uintptr_t(1) << 20), B, N, k, 0);
    std::vector<GemmShape> found;
    Engine e{m, nullptr, 0};
    e.dry = &found;
    TRY(run_encoder(e, w, nullptr, nullptr, B, N));
    TRY(project_cross_kv(e, w, B, N));
    for (int t = 0; t < (m->max_len < 2 ? m->max_len : 2); ++t) TRY(run_decode_step(e, w, B, N, k, t, 0));   // step 0: B rows, later steps: B*k
    for (size_t i = 0; i < found.size() && (int)i < capacity; ++i)
        for (int j = 0; j < 7; ++j) shapes[i * 7 + j] = found[i][j];
    return (int)found.size();
}

extern "C" int ovc_encode(const ovc_model* m, const float* features, const float* boxes, int B, int N,
                          void* workspace, size_t workspace_bytes, float* enc_out, uint8_t* mask_out,
                          ovc_stream stream) {
    if (!model_ok(m) || !features || !workspace || !enc_out || !mask_out || B <= 0 || N <= 0 || N > OVC_MAX_REGIONS) return OVC_EINVAL;
    TRY(ovc_device_guard());
    if (!ovc_aligned16(features) || !ovc_aligned16(workspace) || !ovc_aligned16(enc_out)) return OVC_EINVAL;
    Workspace w = carve(m, workspace, B, N, 1, 0);
    if (w.bytes > workspace_bytes) return OVC_EWORKSPACE;
    Engine e{m, ovc_hip_stream(stream), 0};
    TRY(run_encoder(e, w, features, boxes, B, N));
    const size_t nd = (size_t)N * m->d_model;
    if (m->n_levels > 1) {
        hipLaunchKernelGGL(interleave_levels_kernel, dim3(1024), dim3(256), 0, e.stream, w.enc_levels, enc_out, B,
                           m->n_levels, nd / 4);
        OVC_RETURN_IF_LAUNCH_FAILED();
    } else if (hipMemcpyAsync(enc_out, w.enc_levels, sizeof(float) * B * nd, hipMemcpyDeviceToDevice, e.stream) != hipSuccess) {
        return OVC_ELAUNCH;
    }
    if (hipMemcpyAsync(mask_out, w.enc_mask, (size_t)B * N, hipMemcpyDeviceToDevice, e.stream) != hipSuccess) return OVC_ELAUNCH;
    return OVC_OK;
}

extern "C" int ovc_beam_search(const ovc_model* m, const float* features, const float* boxes, int B, int N, int k,
                               int out_size, void* workspace, size_t workspace_bytes, int64_t* ids_out,
                               float* logp_out, float* all_logp_out, ovc_stream stream) {
    if (!model_ok(m) || !features || !workspace || !ids_out || !logp_out) return OVC_EINVAL;
    TRY(ovc_device_guard());
    if (B <= 0 || N <= 0 || N > OVC_MAX_REGIONS || k <= 0 || k > OVC_MAX_BEAM || out_size <= 0 || out_size > k) return OVC_EINVAL;
    if ((long)m->vocab < k) return OVC_EINVAL;
    if (!ovc_aligned16(features) || !ovc_aligned16(workspace)) return OVC_EINVAL;
    const int return_probs = all_logp_out != nullptr;
    Workspace w = carve(m, workspace, B, N, k, return_probs);
    if (w.bytes > workspace_bytes) return OVC_EWORKSPACE;
    Engine e{m, ovc_hip_stream(stream), 0};
    const int R = B * k, T = m->max_len;

    TRY(run_encoder(e, w, features, boxes, B, N));
    TRY(project_cross_kv(e, w, B, N));
    hipLaunchKernelGGL(init_beam_state_kernel, dim3((R + 255) / 256), dim3(256), 0, e.stream, w.running[0], w.alive[0], R);
    OVC_RETURN_IF_LAUNCH_FAILED();
    for (int t = 0; t < T; ++t) TRY(run_decode_step(e, w, B, N, k, t, return_probs));

    const int fin = T & 1;
    BeamFinalArgs bf{};
    bf.running = w.running[fin]; bf.hist = w.hist[fin]; bf.lp = w.lp[fin];
    bf.k = k; bf.T = T; bf.out_size = out_size; bf.ids_out = ids_out; bf.logp_out = logp_out; bf.order_out = w.order;
    TRY(ovc_beam_finalize_launch(bf, B, e.stream));
    if (return_probs) TRY(ovc_beam_gather_all_launch(w.all_buf, w.order, B, k, T, m->vocab, all_logp_out, e.stream));
    return OVC_OK;
}

// ---------------------------------------------------------------------------------------------
// hipGraph replay of the beam search: everything after the input-dependent kernels is a fixed
// sequence of ~740 launches whose arguments (workspace, weights, shapes, step index) never change
// for a given (model, B, N, k, workspace), so it is captured once and replayed.
// ---------------------------------------------------------------------------------------------
namespace {
struct GraphKey {
    uint64_t model_hash; const void* ws; int B, N, k, out_size;
    bool operator<(const GraphKey& o) const {
        return std::tie(model_hash, ws, B, N, k, out_size) < std::tie(o.model_hash, o.ws, o.B, o.N, o.k, o.out_size);
    }
};
struct GraphEntry { int calls; bool unsupported; hipGraph_t graph; hipGraphExec_t exec; hipStream_t last_stream; uint64_t last_use; };
std::map<GraphKey, GraphEntry> g_graphs;
std::mutex g_graph_mutex;
uint64_t g_graph_tick = 0;

// Captured graphs hold ~740 kernel nodes each; real-data batches bring a new (N bucket, batch size) now and then, so
// the cache is bounded (OVC_GRAPH_CACHE_MAX entries, default 24) and evicts the least recently used entry.
size_t graph_cache_capacity() {
    static const size_t cap = [] { const char* e = getenv("OVC_GRAPH_CACHE_MAX"); const long v = e ? atol(e) : 24; return (size_t)(v < 1 ? 1 : v); }();
    return cap;
}

void destroy_entry(GraphEntry& g) {       // caller holds g_graph_mutex
    if (g.exec) {
        if (g.last_stream) (void)hipStreamSynchronize(g.last_stream);   // a replay may still be running
        (void)hipGraphExecDestroy(g.exec);
    }
    if (g.graph) (void)hipGraphDestroy(g.graph);
    g.exec = nullptr; g.graph = nullptr;
}

// The per-step graphs of ovc_beam_search_early (one entry = max_len + 1 graphs, pinned memory and events); declared here
// because the two caches share ONE bound.
struct EarlyEntry {
    int calls = 0;
    bool unsupported = false;
    hipGraph_t prologue_graph = nullptr; hipGraphExec_t prologue_exec = nullptr;
    std::vector<hipGraph_t> step_graph; std::vector<hipGraphExec_t> step_exec;
    std::vector<hipEvent_t> step_done;
    int32_t* host_alive = nullptr;                 // pinned [T]
    hipStream_t last_stream = nullptr;
    uint64_t last_use = 0;
    std::mutex in_use;                             // one search at a time per (model, shape, workspace)
    ~EarlyEntry() {
        if (last_stream) (void)hipStreamSynchronize(last_stream);
        for (hipGraphExec_t x : step_exec) if (x) (void)hipGraphExecDestroy(x);
        for (hipGraph_t g : step_graph) if (g) (void)hipGraphDestroy(g);
        if (prologue_exec) (void)hipGraphExecDestroy(prologue_exec);
        if (prologue_graph) (void)hipGraphDestroy(prologue_graph);
        for (hipEvent_t ev : step_done) if (ev) (void)hipEventDestroy(ev);
        if (host_alive) (void)hipHostFree(host_alive);
    }
};
std::map<GraphKey, std::shared_ptr<EarlyEntry>> g_early;        // guarded by g_graph_mutex

// Least-recently-used eviction over BOTH caches: together they hold at most OVC_GRAPH_CACHE_MAX entries.  `keep` / `keep_early`
// (the entry the caller is about to use) are never evicted; an early-exit entry that another thread is still using lives on in
// that thread's shared_ptr.  Caller holds g_graph_mutex.
void evict_lru(const GraphKey* keep, const EarlyEntry* keep_early) {
    while (g_graphs.size() + g_early.size() > graph_cache_capacity()) {
        auto victim = g_graphs.end();
        for (auto it = g_graphs.begin(); it != g_graphs.end(); ++it)
            if (!(keep && !(it->first < *keep) && !(*keep < it->first)) && (victim == g_graphs.end() || it->second.last_use < victim->second.last_use))
                victim = it;
        auto victim_early = g_early.end();
        for (auto it = g_early.begin(); it != g_early.end(); ++it)
            if (it->second.get() != keep_early && (victim_early == g_early.end() || it->second->last_use < victim_early->second->last_use))
                victim_early = it;
        const bool have = victim != g_graphs.end(), have_early = victim_early != g_early.end();
        if (!have && !have_early) return;
        if (have_early && (!have || victim_early->second->last_use < victim->second.last_use)) {
            g_early.erase(victim_early);
        } else {
            destroy_entry(victim->second);
            g_graphs.erase(victim);
        }
    }
}

uint64_t hash_bytes(const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    uint64_t h = 1469598103934665603ull;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

// The stream launch sequences are captured on (never the caller's: see ovc_beam_search_graph).  One device per process
// (ovc_device_guard): the stream belongs to the bound device.  Caller holds g_graph_mutex; nullptr = no capture support.
hipStream_t private_capture_stream() {
    static hipStream_t capture_stream = nullptr;
    if (!capture_stream && hipStreamCreateWithFlags(&capture_stream, hipStreamNonBlocking) != hipSuccess) {
        (void)hipGetLastError();
        capture_stream = nullptr;
    }
    return capture_stream;
}

int issue_decode_graph_body(Engine& e, Workspace& w, int B, int N, int k, int out_size) {
    const ovc_model* m = e.m;
    const int R = B * k, T = m->max_len;
    TRY(run_encoder_layers(e, w, B, N));
    TRY(project_cross_kv(e, w, B, N));
    hipLaunchKernelGGL(init_beam_state_kernel, dim3((R + 255) / 256), dim3(256), 0, e.stream, w.running[0], w.alive[0], R);
    OVC_RETURN_IF_LAUNCH_FAILED();
    for (int t = 0; t < T; ++t) TRY(run_decode_step(e, w, B, N, k, t, 0));
    const int fin = T & 1;
    BeamFinalArgs bf{};
    bf.running = w.running[fin]; bf.hist = w.hist[fin]; bf.lp = w.lp[fin];
    bf.k = k; bf.T = T; bf.out_size = out_size; bf.ids_out = w.out_ids; bf.logp_out = w.out_logp; bf.order_out = w.order;
    return ovc_beam_finalize_launch(bf, B, e.stream);
}
}  // namespace

extern "C" int ovc_beam_search_graph(const ovc_model* m, const float* features, const float* boxes, int B, int N, int k,
                                     int out_size, void* workspace, size_t workspace_bytes, int64_t* ids_out,
                                     float* logp_out, ovc_stream stream) {
    if (!model_ok(m) || !features || !workspace || !ids_out || !logp_out) return OVC_EINVAL;
    TRY(ovc_device_guard());
    if (B <= 0 || N <= 0 || N > OVC_MAX_REGIONS || k <= 0 || k > OVC_MAX_BEAM || out_size <= 0 || out_size > k) return OVC_EINVAL;
    if ((long)m->vocab < k) return OVC_EINVAL;
    if (!ovc_aligned16(features) || !ovc_aligned16(workspace)) return OVC_EINVAL;
    Workspace w = carve(m, workspace, B, N, k, 0);
    if (w.bytes > workspace_bytes) return OVC_EWORKSPACE;
    Engine e{m, ovc_hip_stream(stream), 0};
    const size_t out_n = (size_t)B * out_size * m->max_len;

    const GraphKey key{hash_bytes(m, sizeof(*m)), workspace, B, N, k, out_size};
    std::lock_guard<std::mutex> lock(g_graph_mutex);
    GraphEntry& entry = g_graphs[key];
    entry.calls += 1;
    entry.last_use = ++g_graph_tick;
    entry.last_stream = e.stream;
    evict_lru(&key, nullptr);

    TRY(run_encoder_inputs(e, w, features, boxes, B, N));
    // The launch sequence is captured on a PRIVATE stream, never on the caller's: while a stream is capturing, HIP
    // refuses queries of events that were recorded on it earlier (hipErrorCapturedEvent), and other components poll
    // such events from their own threads -- torch's NCCL watchdog does, for the all-gather that follows each batch.
    // Kernel nodes carry no stream, so the instantiated graph is launched on the caller's stream as usual.
    // (The legacy null stream can launch a graph but offers nothing else here; it takes the same path.)
    if (!entry.unsupported && !g_profile_on && entry.calls > 1 && !entry.exec) {
        hipStream_t capture_stream = private_capture_stream();
        if (!capture_stream) {
            entry.unsupported = true;
        } else if (hipStreamBeginCapture(capture_stream, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            entry.unsupported = true;
        } else {
            Engine ce{m, capture_stream, 0};
            const int rc = issue_decode_graph_body(ce, w, B, N, k, out_size);
            const hipError_t end = hipStreamEndCapture(capture_stream, &entry.graph);
            if (rc != OVC_OK || end != hipSuccess || !entry.graph ||
                hipGraphInstantiate(&entry.exec, entry.graph, nullptr, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                if (entry.graph) (void)hipGraphDestroy(entry.graph);
                entry.graph = nullptr; entry.exec = nullptr; entry.unsupported = true;
            }
        }
    }
    if (entry.exec && !g_profile_on) {
        if (hipGraphLaunch(entry.exec, e.stream) != hipSuccess) return OVC_ELAUNCH;
    } else {
        // first call of a shape (warms every kernel's one-off attribute set-up), profiling, or no capture support
        TRY(issue_decode_graph_body(e, w, B, N, k, out_size));
    }
    if (hipMemcpyAsync(ids_out, w.out_ids, sizeof(int64_t) * out_n, hipMemcpyDeviceToDevice, e.stream) != hipSuccess) return OVC_ELAUNCH;
    if (hipMemcpyAsync(logp_out, w.out_logp, sizeof(float) * out_n, hipMemcpyDeviceToDevice, e.stream) != hipSuccess) return OVC_ELAUNCH;
    return OVC_OK;
}

// ---------------------------------------------------------------------------------------------
// Early exit (round 4).  The reference always runs max_len steps (beam_search.py:94-95), although once every beam of every
// image has emitted <eos> a step only appends word 0 / log-prob 0 to every beam and (once) re-orders the beams by score --
// which the final ordering does anyway (beam_search.py:49-55, 97-113).  Here the update kernel of step t leaves the number of
// beams still alive in alive_count[t]; the host issues the search STEP BY STEP (one captured graph per step), copies that word
// to pinned memory behind each step and looks at it one step late -- the GPU always has the next step queued -- and stops
// issuing steps once it reads 0.  The final ordering then emits word 0 / log-prob 0 for the positions that were never
// written: results are identical to the full run (tests/test_engine_gpu.py::test_early_exit_*).  Assumes no total score
// below -999 (a frozen beam's other candidates, beam_search.py:54).
// ---------------------------------------------------------------------------------------------
namespace {

int issue_early_prologue(Engine& e, Workspace& w, int B, int N, int k) {
    const int R = B * k;
    TRY(run_encoder_layers(e, w, B, N));
    TRY(project_cross_kv(e, w, B, N));
    hipLaunchKernelGGL(init_beam_state_kernel, dim3((R + 255) / 256), dim3(256), 0, e.stream, w.running[0], w.alive[0], R);
    OVC_RETURN_IF_LAUNCH_FAILED();
    if (hipMemsetAsync(w.alive_count, 0, sizeof(int32_t) * e.m->max_len, e.stream) != hipSuccess) return OVC_ELAUNCH;
    return OVC_OK;
}

// Capture `issue` on the private stream into (graph, exec); false = capture not available (the caller launches plainly).
template <typename Issue>
bool capture_into(hipGraph_t* graph, hipGraphExec_t* exec, const ovc_model* m, Issue issue) {
    hipStream_t cs = private_capture_stream();
    if (!cs || hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal) != hipSuccess) { (void)hipGetLastError(); return false; }
    Engine ce{m, cs, 0};
    const int rc = issue(ce);
    const hipError_t end = hipStreamEndCapture(cs, graph);
    if (rc != OVC_OK || end != hipSuccess || !*graph || hipGraphInstantiate(exec, *graph, nullptr, nullptr, 0) != hipSuccess) {
        (void)hipGetLastError();
        if (*graph) (void)hipGraphDestroy(*graph);
        *graph = nullptr; *exec = nullptr;
        return false;
    }
    return true;
}
}  // namespace

extern "C" int ovc_beam_search_early(const ovc_model* m, const float* features, const float* boxes, int B, int N, int k,
                                     int out_size, void* workspace, size_t workspace_bytes, int64_t* ids_out,
                                     float* logp_out, int* steps_run_out, ovc_stream stream) {
    if (!model_ok(m) || !features || !workspace || !ids_out || !logp_out) return OVC_EINVAL;
    TRY(ovc_device_guard());
    if (B <= 0 || N <= 0 || N > OVC_MAX_REGIONS || k <= 0 || k > OVC_MAX_BEAM || out_size <= 0 || out_size > k) return OVC_EINVAL;
    if ((long)m->vocab < k) return OVC_EINVAL;
    if (!ovc_aligned16(features) || !ovc_aligned16(workspace)) return OVC_EINVAL;
    Workspace w = carve(m, workspace, B, N, k, 0);
    if (w.bytes > workspace_bytes) return OVC_EWORKSPACE;
    Engine e{m, ovc_hip_stream(stream), 0};
    const int T = m->max_len;

    std::shared_ptr<EarlyEntry> entry;
    {
        const GraphKey key{hash_bytes(m, sizeof(*m)), workspace, B, N, k, out_size};
        std::lock_guard<std::mutex> lock(g_graph_mutex);
        std::shared_ptr<EarlyEntry>& slot = g_early[key];
        if (!slot) slot = std::make_shared<EarlyEntry>();
        entry = slot;
        entry->last_use = ++g_graph_tick;
        evict_lru(nullptr, entry.get());                           // one bound for both caches, least recently used first
    }
    std::lock_guard<std::mutex> busy(entry->in_use);
    entry->calls += 1;
    entry->last_stream = e.stream;
    if (!entry->host_alive) {
        if (hipHostMalloc(reinterpret_cast<void**>(&entry->host_alive), sizeof(int32_t) * T, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError();
            entry->host_alive = nullptr;
            return OVC_ELAUNCH;
        }
        entry->step_graph.assign(T, nullptr); entry->step_exec.assign(T, nullptr); entry->step_done.assign(T, nullptr);
        for (int t = 0; t < T; ++t)
            if (hipEventCreateWithFlags(&entry->step_done[t], hipEventDisableTiming) != hipSuccess) return OVC_ELAUNCH;
    }
    const bool graphs = !entry->unsupported && !g_profile_on && entry->calls > 1;   // first call of a shape: plain (warms every kernel)

    TRY(run_encoder_inputs(e, w, features, boxes, B, N));
    if (graphs && !entry->prologue_exec) {
        std::lock_guard<std::mutex> lock(g_graph_mutex);           // the capture stream is shared process-wide
        if (!capture_into(&entry->prologue_graph, &entry->prologue_exec, m, [&](Engine& ce) { return issue_early_prologue(ce, w, B, N, k); }))
            entry->unsupported = true;
    }
    if (graphs && entry->prologue_exec) { if (hipGraphLaunch(entry->prologue_exec, e.stream) != hipSuccess) return OVC_ELAUNCH; }
    else TRY(issue_early_prologue(e, w, B, N, k));

    int steps_run = T;
    for (int t = 0; t < T; ++t) {
        if (graphs && !entry->unsupported && !entry->step_exec[t]) {
            std::lock_guard<std::mutex> lock(g_graph_mutex);
            if (!capture_into(&entry->step_graph[t], &entry->step_exec[t], m,
                              [&](Engine& ce) { return run_decode_step(ce, w, B, N, k, t, 0, true); }))
                entry->unsupported = true;
        }
        if (graphs && entry->step_exec[t]) { if (hipGraphLaunch(entry->step_exec[t], e.stream) != hipSuccess) return OVC_ELAUNCH; }
        else TRY(run_decode_step(e, w, B, N, k, t, 0, true));
        if (t + 1 == T) break;                                      // nothing left to skip
        if (hipMemcpyAsync(entry->host_alive + t, w.alive_count + t, sizeof(int32_t), hipMemcpyDeviceToHost, e.stream) != hipSuccess ||
            hipEventRecord(entry->step_done[t], e.stream) != hipSuccess) return OVC_ELAUNCH;
        // one step late: step t is queued, step t - 1's count is (about to be) on the host
        if (t >= 1) {
            if (hipEventSynchronize(entry->step_done[t - 1]) != hipSuccess) return OVC_ELAUNCH;
            if (entry->host_alive[t - 1] == 0) { steps_run = t + 1; break; }
        }
    }

    const int fin = steps_run & 1;
    BeamFinalArgs bf{};
    bf.running = w.running[fin]; bf.hist = w.hist[fin]; bf.lp = w.lp[fin];
    bf.k = k; bf.T = T; bf.out_size = out_size; bf.ids_out = ids_out; bf.logp_out = logp_out; bf.order_out = w.order;
    bf.steps_run = steps_run < T ? steps_run : 0;
    TRY(ovc_beam_finalize_launch(bf, B, e.stream));
    if (steps_run_out) *steps_run_out = steps_run;
    return OVC_OK;
}

// Test hook: ONE selection step of the fused path on caller-supplied decoder outputs -- the vocabulary product with its
// log-softmax epilogue (transposed != 0: the fp32 engine's form, logits^T = fc . x^T, which the engine runs in the one-chain
// class, kchains = 1; 0: the row-major form of the split-precision modes; kchains = 1 / 4 picks the fp32 K-order class, and
// with it the tiling instances whose epilogue runs -- ovc_debug_force_gemm_tiling narrows it to one) followed by beam_fused_update_kernel -- so that the selection can be checked against a stable sort
// at the operator level (tests/test_ops_gpu.py).  x [B*width, d], fc [V, d], running / alive [B*width]; chosen [B, k] receives
// flat indices beam * V + word in winning order, score [B, k] their scores.  scratch: ovc_debug_vocab_select_bytes.
extern "C" size_t ovc_debug_vocab_select_bytes(int B, int width, int V, int k) {
    if (B <= 0 || width <= 0 || V <= 0 || k <= 0) return 0;
    const size_t R = (size_t)B * width, nblk = ((size_t)V + 31) / 32, ld = (nblk + 1) & ~(size_t)1;
    return 4 * (((R + 3) & ~(size_t)3) * (((size_t)V + 3) & ~(size_t)3) + 2 * R * ld + 8 * (size_t)B * k + 64) + 4096;
}

extern "C" int ovc_debug_vocab_select(const float* x, const float* fc, const float* running, const float* alive, int B, int width,
                                      int V, int d, int k, int transposed, int kchains, void* scratch, size_t scratch_bytes,
                                      int64_t* chosen, float* score, ovc_stream stream) {
    if (!x || !fc || !running || !alive || !scratch || !chosen || !score || B <= 0 || width <= 0 || width > OVC_MAX_BEAM || k <= 0 ||
        k > OVC_MAX_BEAM || V < k || d <= 0 || (d & 3) || (V + 31) / 32 > 512 || (kchains != 1 && kchains != 4)) return OVC_EINVAL;
    if (scratch_bytes < ovc_debug_vocab_select_bytes(B, width, V, k) || !ovc_aligned16(scratch)) return OVC_EWORKSPACE;
    TRY(ovc_device_guard());
    hipStream_t s = ovc_hip_stream(stream);
    const int rows = B * width, nblk = (V + 31) / 32, ld = (nblk + 1) & ~1, ldv = (V + 3) & ~3, ldt = (rows + 3) & ~3;
    Bump a{reinterpret_cast<char*>(scratch), 0};
    float* logits = a.take<float>(((size_t)(rows + 3) & ~(size_t)3) * ldv);
    float* stats = a.take<float>(2 * (size_t)rows * ld);
    float* alive_out = a.take<float>((size_t)B * k); float* running_out = a.take<float>((size_t)B * k);
    float* lp_out = a.take<float>((size_t)B * k);
    int32_t* hist_out = a.take<int32_t>((size_t)B * k); int32_t* anc_out = a.take<int32_t>((size_t)B * k);
    int32_t* next_tok = a.take<int32_t>((size_t)B * k);
    GemmArgs g{};
    g.kchains = kchains; g.K1 = d; g.lda1 = d; g.nseg = 1; g.stats_ld = ld;     // the fp32 engine: transposed, one chain
    if (transposed) {
        g.A1 = fc; g.M = V; g.seg_n = rows; g.ldc = ldt; g.seg[0] = GemmSegment{x, nullptr, logits, nullptr, nullptr}; g.stats_t = stats;
    } else {
        g.A1 = x; g.M = rows; g.seg_n = V; g.ldc = ldv; g.seg[0] = GemmSegment{fc, nullptr, logits, nullptr, nullptr}; g.stats = stats;
    }
    TRY(ovc_gemm_launch(g, s));
    BeamUpdateArgs bu{};
    bu.logits = logits; bu.ld = ldv; bu.alive_in = alive; bu.alive_out = alive_out; bu.running_out = running_out;
    bu.hist_out = hist_out; bu.lp_out = lp_out; bu.anc_out = anc_out; bu.next_tok = next_tok;
    bu.hist_in = hist_out; bu.lp_in = lp_out; bu.anc_in = anc_out;                 // t = 0: nothing is copied from them
    bu.width = width; bu.k = k; bu.V = V; bu.T = 1; bu.t = 0; bu.eos = -1;
    TRY(ovc_beam_fused_update_launch(bu, stats, nblk, ld, running, transposed ? 1 : ldv, transposed ? ldt : 1, B, s));
    return ovc_debug_collect_winners_launch(anc_out, hist_out, running_out, B, width, V, k, chosen, score, s);
}

extern "C" int ovc_graph_cache_clear(void) {
    std::lock_guard<std::mutex> lock(g_graph_mutex);
    for (auto& kv : g_graphs) destroy_entry(kv.second);
    g_graphs.clear();
    g_early.clear();
    return OVC_OK;
}

extern "C" int ovc_graph_cache_drop_workspace(const void* workspace) {
    std::lock_guard<std::mutex> lock(g_graph_mutex);
    int dropped = 0;
    for (auto it = g_graphs.begin(); it != g_graphs.end();) {
        if (it->first.ws == workspace) { destroy_entry(it->second); it = g_graphs.erase(it); ++dropped; }
        else ++it;
    }
    for (auto it = g_early.begin(); it != g_early.end();) {
        if (it->first.ws == workspace) { it = g_early.erase(it); ++dropped; }
        else ++it;
    }
    return dropped;
}

extern "C" int ovc_graph_cache_size(void) {
    std::lock_guard<std::mutex> lock(g_graph_mutex);
    return (int)(g_graphs.size() + g_early.size());
}

extern "C" int ovc_profile_enable(int on) {
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    if (on && !g_profile_on) {
        profile_resolve();
        for (ProfileBin& b : g_by_class) b = ProfileBin{};
        for (ProfileBin& b : g_by_tiling) b = ProfileBin{};
    }
    g_profile_on = on != 0;
    return OVC_OK;
}

extern "C" int ovc_profile_read(int kind, int index, int64_t* launches, double* total_ms, double* total_flops) {
    if (!launches || !total_ms || !total_flops) return OVC_EINVAL;
    if (kind == 0 ? (index < 0 || index >= OVC_PROFILE_CLASSES) : (kind != 1 || index < 0 || index >= kProfileTilings)) return OVC_EINVAL;
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    profile_resolve();
    const ProfileBin& b = kind == 0 ? g_by_class[index] : g_by_tiling[index];
    *launches = b.launches; *total_ms = b.ms; *total_flops = b.flops;
    return OVC_OK;
}

extern "C" double ovc_profile_overhead_ms(void) {
    std::lock_guard<std::mutex> lock(g_profile_mutex);
    profile_resolve();
    return g_profile_overhead_ms;
}

extern "C" const char* ovc_profile_kernel_name(int tiling) { return ovc_gemm_tiling_name(tiling); }
