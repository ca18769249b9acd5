// Shared declarations of the gfx950 kernels behind include/ovc.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/ovc.h"

#define OVC_WAVE 64

#define OVC_RETURN_IF_LAUNCH_FAILED()                      \
    do {                                                   \
        if (hipGetLastError() != hipSuccess) return OVC_ELAUNCH; \
    } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

static inline hipStream_t ovc_hip_stream(ovc_stream s) { return reinterpret_cast<hipStream_t>(s); }

// OVC_OK when the calling thread's current device is the one the library is bound to (binding it on first use),
// OVC_EDEVICE otherwise (include/ovc.h: one device per process).  Defined in gemm.hip.
int ovc_device_guard();

static inline bool ovc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Run-time switches.  The shipped library reads no environment variable that can change a result bit, skip work or select
// another kernel: the measurement hooks (OVC_DEBUG_*, OVC_KSPLIT_*: some of them produce garbage by design -- timing only) and
// the A/B switches between alternative kernels exist only in a build with -DOVC_MEASUREMENT_HOOKS
// (`python -m openviic_amd.csrc.build --hooks` -> tools/libovc_hooks.so, loaded with OVC_LIBRARY=...; ovc_build_info() says
// so and bench.py refuses such a library for a credited line).  In the default build the names below do not even exist as
// strings.  The one variable the default build reads is OVC_GRAPH_CACHE_MAX (the LRU bound of the hipGraph cache).
#ifdef OVC_MEASUREMENT_HOOKS
#include <cstdlib>
#define OVC_HOOK_ENV(name) getenv(name)
#else
#define OVC_HOOK_ENV(name) (static_cast<const char*>(nullptr))
#endif

// ---- wave-level reductions (64 lanes) ------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
// ---- reductions over the 32 lanes of one half of a wave (all 32 lanes receive the result) --------------------------------
// four DPP steps (quad_perm xor 1, xor 2, row_half_mirror, row_mirror: full-rate VALU, no LDS) + one xor-16 exchange
template <int CTRL>
__device__ __forceinline__ float ovc_dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// the gfx950 row exchanges v_permlane16_swap / v_permlane32_swap: x, y = the two rows (16 lanes) / halves (32 lanes) a lane
// pairs with, no LDS (tools/wave_reduce_probe.hip: a wave all-reduce in 116 ns against 211 ns for the ds_bpermute butterfly;
// the empty asm statements keep hipcc from folding the two results of a swap whose inputs are the same value)
template <bool kRows32>
__device__ __forceinline__ void ovc_swap_rows(float v, float& x, float& y) {
    int a = __builtin_bit_cast(int, v), c = a;
    asm volatile("" : "+v"(c));
    int x0, x1;
    if (kRows32) { auto r = __builtin_amdgcn_permlane32_swap(a, c, false, false); x0 = r[0]; x1 = r[1]; }
    else { auto r = __builtin_amdgcn_permlane16_swap(a, c, false, false); x0 = r[0]; x1 = r[1]; }
    asm volatile("" : "+v"(x0), "+v"(x1));
    x = __builtin_bit_cast(float, x0); y = __builtin_bit_cast(float, x1);
}
__device__ __forceinline__ float half_wave_max(float v) {
    v = fmaxf(v, ovc_dpp<0xB1>(v));
    v = fmaxf(v, ovc_dpp<0x4E>(v));
    v = fmaxf(v, ovc_dpp<0x141>(v));
    v = fmaxf(v, ovc_dpp<0x140>(v));
    float x, y;
    ovc_swap_rows<false>(v, x, y);
    return fmaxf(x, y);
}
__device__ __forceinline__ float half_wave_sum(float v) {      // a fixed association order: the same bits on every tiling
    v += ovc_dpp<0xB1>(v);
    v += ovc_dpp<0x4E>(v);
    v += ovc_dpp<0x141>(v);
    v += ovc_dpp<0x140>(v);
    float x, y;
    ovc_swap_rows<false>(v, x, y);
    return x + y;
}
// all 64 lanes
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = half_wave_max(v);
    float x, y;
    ovc_swap_rows<true>(v, x, y);
    return fmaxf(x, y);
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v = half_wave_sum(v);
    float x, y;
    ovc_swap_rows<true>(v, x, y);
    return x + y;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}

// ---- internal GEMM interface (gemm.hip) ---------------------------------------------------
struct GemmSegment {
    const float* W;      // [seg_n, K] row-major
    const float* bias;   // [seg_n] or nullptr
    float* C;            // [M, seg_n] with row stride ldc
    const float* A2;     // this segment's own second input block [M, K2] (row stride lda2), or nullptr = GemmArgs::A2
    const void* Wp;      // split-precision classes only: W cut into 16-bit planes in MFMA-fragment order (ovc_split_weight),
                         // read by each wave straight from memory -- or nullptr: the kernel cuts W itself, through LDS
};

struct GemmArgs {
    const float* A1;     // [M, K1], row stride lda1
    const float* A2;     // [M, K2], row stride lda2 (nullptr when K2 == 0)
    int lda1, lda2, K1, K2;
    int M;
    int seg_n;           // columns per segment
    int nseg;            // number of segments (total N = nseg * seg_n)
    int ldc;
    const float* R;      // residual [M, seg_n] (only with nseg == 1) or nullptr
    int ldr;
    int res_mod;         // > 0: the residual has res_mod rows and row m reads row m % res_mod (broadcast over stacked blocks)
    int act;             // 0 none, 1 relu
    int ksplit;          // > 1: K is cut into ksplit equal slices, slice s writes its raw partial product (no bias /
                         // activation / residual) to seg[0].C + s * part_stride; the consumer sums the slices in order
    long part_stride;    // floats between two partial outputs
    int kchains;         // K-order class (gemm.hip): 1 (or 0) = one summation chain over k, 4 = four interleaved chains
                         // summed in chain order.  Part of the product's DEFINITION: every tiling of a class gives the
                         // same bits, so the caller fixes it per call site and no timing can change a result.
    uint8_t* zero_rows_out;  // optional (fp32 classes, K2 == 0, no K split): zero_rows_out[m] = (sum_k A1[m, k] == 0), the padding
                         // mask of models/utils.py:48-61, found by the workgroups of the first column tile while they stage A --
                         // the feature projection then is the only pass over the caller's features
    float* stats;        // optional (nseg == 1, no K split): per (row, 32-column block) the block's maximum and
    int stats_ld;        // sum exp(y - maximum) of the finished outputs, [M][stats_ld >= ceil(seg_n / 32)] float2 -- the
                         // vocabulary projection's log-softmax pieces, so that the beam update never reads all logits back
    float* stats_t;      // the same pieces for a TRANSPOSED product (rows of C are the words, columns the beam rows): per (column n,
                         // 32-row block) the maximum and sum exp over the block's rows, [seg_n][stats_ld] float2.  A lane of the
                         // accumulator then holds 16 words of ONE beam row, so the reductions are in-register (one half-wave swap each)
    int objective;       // which tuning table to consult: 0 / 1 = measured in isolation, c > 1 = measured with c co-running copies
                         // (speed only: every tiling of the class gives the same bits)
    GemmSegment seg[OVC_MAX_SEGMENTS];
};

// Per-launch options (never global state: several host threads may drive the library at once).
struct GemmLaunchOpts {
    int forced_tiling = -1;                 // >= 0: use exactly this tiling (must fit the problem and its class)
    int copies = 1;                         // tuner: gridDim.z identical copies co-running in one launch
    hipEvent_t start = nullptr, stop = nullptr;   // kernel-scoped events (dispatch begin / end timestamps) for profiling
};

// Launches C = act([A1|A2] W^T + bias) + R on `stream`; returns an OVC_* code.
int ovc_gemm_launch(const GemmArgs& args, hipStream_t stream, const GemmLaunchOpts& opts = GemmLaunchOpts{});
int ovc_gemm_pick_tiling(const GemmArgs& args, const GemmLaunchOpts& opts = GemmLaunchOpts{});   // tiling ovc_gemm_launch will use
int ovc_gemm_tiling_class(int tiling);               // chains of a tiling's K-order class (0 = no such tiling)
constexpr int kMaxKSplit = 4;
// LayerNorm(sum_s parts[s] + bias + residual), nparts in {2, 4}, bias and residual required: the consumer side of
// a K-split GEMM (rowops.hip).
int ovc_layer_norm_parts(const float* parts, int nparts, long part_stride, const float* bias, const float* residual,
                         const float* gamma, const float* beta, const uint8_t* zero_rows, float eps, float* y,
                         int rows, int d, hipStream_t stream);
const char* ovc_gemm_tiling_name(int tiling);        // kernel name as rocprofv3 prints it

// ---- decode-time attention (attention.hip) -------------------------------------------------
struct DecodeSelfArgs {
    const float* q;        // [rows, ldq]      projected queries of this step
    int ldq;
    const float* kcache;   // [T][slots][ldkv] projected keys, position-major
    const float* vcache;
    size_t pos_stride;     // floats between two positions
    int ldkv;
    const int32_t* anc;    // [rows][anc_ld]   slot of the ancestor that produced position j (< t)
    int anc_ld;
    const uint8_t* padflag;  // [T][pad_ld]    1 where the token fed at (position, slot) was <pad>
    int pad_ld;
    int t;                 // current position: keys 0..t (key t lives in the row's own slot)
    int width;             // rows per image at this step (1 at t = 0, the beam size later): rows b * width .. + width - 1 are
                           // image b's beams, and position j's cache block holds the image's slots b * width_j .. (width_0 = 1)
    int h, dk, dv;
    float* out;            // [rows, ldo]
    int ldo;
};
int ovc_decode_self_attention(const DecodeSelfArgs& p, int rows, hipStream_t stream);

struct DecodeCrossArgs {
    const float* q;        // [B*width, ldq]
    int ldq;
    const float* kx;       // [levels][B][N][ldkv] projected encoder keys (per decoder layer)
    const float* vx;
    size_t level_stride;
    int ldkv;
    const uint8_t* encmask;  // [B][N] or nullptr
    int n, width;
    int heads, dk, dv;
    float* out;            // [levels][B*width][ldo]
    size_t out_level_stride;
    int ldo;
};
int ovc_decode_cross_attention(const DecodeCrossArgs& p, int B, int h, int levels, hipStream_t stream);

// ---- beam search (beam.hip) ------------------------------------------------------------------
struct BeamSelectArgs {
    const float* logits;     // [B, width, ld] raw vocabulary logits (or log-probs when is_logp)
    int ld;
    int is_logp;
    const float* running;    // [B, width]
    const float* alive;      // [B, width] or nullptr (all alive)
    int width, V, k;
    float* cand_v;           // [B*width, k] every row's k best candidate scores ...
    int* cand_i;             // ... and their flat indices beam*V + word (0x7fffffff = none)
    int64_t* chosen;         // [B, k] the image's winners (flat indices), or nullptr when the caller merges the rows
    float* score;            // [B, k]
    float* masked_logp;      // [B, width, V] or nullptr
    float* row_max_out;      // [B, width] or nullptr: log-softmax pieces for the update kernel
    float* row_lsum_out;
};
int ovc_beam_select_launch(const BeamSelectArgs& p, int B, hipStream_t stream);

struct BeamUpdateArgs {
    const float* cand_v; const int* cand_i;         // [B*width, k] row candidates of ovc_beam_select_launch
    const float* logits; int ld;
    const float* row_max; const float* row_lsum;    // two-pass path: the selection kernel's log-softmax pieces per row
    float* row_max_out; float* row_lsum_out;        // fused path: where the pieces are published (return_probs), or nullptr
    const float* alive_in; float* alive_out; float* running_out;
    const int32_t* hist_in; int32_t* hist_out;      // [B*k, T] words
    const float* lp_in; float* lp_out;              // [B*k, T] per-token log-probs
    const int32_t* anc_in; int32_t* anc_out;        // [B*k, T] ancestor slots
    int32_t* next_tok;                              // [B*k]
    int width, k, V, T, t, eos;
    // input rows of step t + 1, written here instead of by a separate launch (nullptr: not wanted):
    //   next_x[b*k + j, :] = word_emb[word] + pos_emb[t + 2],  next_padflag[b*k + j] = (word == pad)
    const float* word_emb; const float* pos_emb; float* next_x; uint8_t* next_padflag; int d_model, pad;
    // early exit (ovc_beam_search_early; nullptr otherwise): alive_count[t] receives, summed over the batch's images, the beams
    // still alive after step t (an image without any valid candidate -- NaN logits: no region at all -- counts as ended);
    // zeroed by the caller before the first step.  0 = every later step only appends word 0 / log-prob 0 to every beam.
    int32_t* alive_count;
};
int ovc_beam_update_launch(const BeamUpdateArgs& p, int B, hipStream_t stream);
// Selection + update in one launch from the vocabulary GEMM's block pieces (GemmArgs::stats, [rows][stats_ld] float2,
// stats_ld even): no pass over the logits.  nblk = ceil(V / 32) <= 512; p.cand_* / p.row_max / p.row_lsum are not used.
// Logit (row, word) lives at logits[row * ld_row + word * ld_word]: (ld, 1) for the row-major product, (1, ld) for the
// transposed one.
int ovc_beam_fused_update_launch(const BeamUpdateArgs& p, const float* stats, int nblk, int stats_ld, const float* running_in,
                                 long ld_row, long ld_word, int B, hipStream_t stream);
int ovc_debug_collect_winners_launch(const int32_t* anc, const int32_t* word, const float* running, int B, int width, int V, int k,
                                     int64_t* chosen, float* score, hipStream_t stream);
int ovc_masked_logp_launch(const float* logits, long ld_row, long ld_word, const float* row_max, const float* row_lsum,
                           const float* alive, int rows, int V, float* out, hipStream_t stream);

struct BeamFinalArgs {
    const float* running; const int32_t* hist; const float* lp;
    int k, T, out_size;
    int64_t* ids_out; float* logp_out; int32_t* order_out;
    int steps_run;           // 0 = all T steps ran.  s in 1..T-1 (early exit: every beam had ended): positions s..T-1 were never
                             // written and are emitted as word 0 / log-prob 0 -- what the remaining steps would have appended
};
int ovc_beam_finalize_launch(const BeamFinalArgs& p, int B, hipStream_t stream);
int ovc_beam_gather_all_launch(const float* all_buf, const int* order, int B, int k, int T, int V, float* all_out,
                               hipStream_t stream);

// out = (sum_l sigmoid(alpha[l]) * enc[l]) / divisor over `levels` stacked [n] blocks (rowops.hip)
int ovc_meshed_mix(const float* alpha, const float* enc, int levels, long n, float divisor, float* out, hipStream_t stream);
