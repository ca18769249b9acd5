/*
 * ovc.h -- C ABI of the MI355X (gfx950) captioning engine ("libovc.so").
 *
 * The reference (hieunghia-pat/OpenViIC) has no native boundary: its hot path is eager PyTorch
 * (ATen) dispatched from Python classes.  This header is therefore the boundary a maintainer
 * would bind instead of those ATen call sites; every entry point names the reference lines it
 * replaces (paths relative to the reference checkout).  All pointers are DEVICE pointers to
 * dense row-major fp32 data unless stated otherwise; masks are one byte per element (0/1);
 * token ids are int64.  Nothing allocates, nothing synchronises, every launch goes to the
 * caller's hipStream_t; the return value is 0 on success or a negative OVC_E* code, and no
 * exception crosses the boundary.  Process-wide state is limited to caches and opt-in tools:
 * the GEMM tuning table, the hipGraph cache and the profiling counters, each behind its own
 * mutex: the library may be driven from several host threads (one stream per thread; the debug
 * hook ovc_debug_force_gemm_tiling is the one exception and says so).
 *
 * ONE DEVICE PER PROCESS.  That state belongs to a device (kernel attributes raised once, the
 * graph-capture stream, captured graphs, tiling timings), and the deployment model is one process
 * per GPU (torch.distributed / RCCL).  The first launching call binds the library to the device
 * that is current on the calling thread; every later call made while another device is current
 * returns OVC_EDEVICE instead of capturing or launching on the wrong device.  ovc_bound_device()
 * reports the binding (-1 = none yet).
 */
#ifndef OVC_H_
#define OVC_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ovc_stream;              /* a hipStream_t (NULL = the null stream) */

#define OVC_OK            0
#define OVC_EINVAL       -1            /* bad argument / unsupported shape   */
#define OVC_EWORKSPACE   -2            /* workspace too small                */
#define OVC_ELAUNCH      -3            /* hipGetLastError() != hipSuccess    */
#define OVC_EDEVICE      -4            /* current device != the device the library is bound to (one per process) */

#define OVC_MAX_LAYERS    8
#define OVC_MAX_LEVELS    4
#define OVC_MAX_BEAM      8
#define OVC_MAX_SEGMENTS  8
#define OVC_MAX_REGIONS   1024         /* regions (or grid cells) per image the engine accepts */

/* library / build identification ------------------------------------------------------ */
int         ovc_abi_version(void);             /* bumps when a struct layout changes     */
const char* ovc_build_info(void);              /* "gfx950 fp32-mfma ..."                 */
int         ovc_bound_device(void);            /* device the process is bound to, -1 = none yet */
int         ovc_debug_rebind_device(int device);   /* tests only: overwrite the binding (-1 = unbound) */

/* ======================================================================================
 * Operator level (parity-test surface; also what the host-side modules call)
 * ==================================================================================== */

/* y[M,N] = act([x | x2] W^T + bias) + residual
 *   x  [M,K1] (row stride ldx), x2 [M,K2] or NULL (row stride ldx2), W [N,K1+K2], bias [N] or
 *   NULL, residual [M,N] or NULL (row stride ldr), y row stride ldy.  act: 0 none, 1 relu.
 * Replaces every nn.Linear on the path: models/modules/vision_embeddings.py:17,
 * attentions.py:47-49,56, positionwise_feed_forward.py:24, decoders.py:121; with x2 the
 * concatenated-input gates attentions.py:311-314 and decoders.py:61.  fp32 MFMA
 * (v_mfma_f32_32x32x2_f32), fp32 accumulate. */
int ovc_linear(const float* x, int ldx, const float* x2, int ldx2, int K1, int K2,
               const float* W, const float* bias, const float* residual, int ldr,
               float* y, int ldy, int M, int N, int act, ovc_stream stream);

/* y[r,:] = LayerNorm(x[r,:] + residual[r,:]) * gamma + beta + add[r % add_rows,:]; rows with
 * zero_rows[r] != 0 are written as 0.  residual / add / zero_rows may be NULL.
 * Replaces nn.LayerNorm at attentions.py:309, positionwise_feed_forward.py:26, encoders.py:36
 * (with add = positional encoding) and the masked_fill at encoders.py:20 / decoders.py:26. */
int ovc_layer_norm(const float* x, const float* residual, const float* gamma, const float* beta,
                   const float* add, int add_rows, const uint8_t* zero_rows, float eps,
                   float* y, int rows, int d, ovc_stream stream);

/* Scaled dot-product attention on projected heads.
 *   q [b,nq,h*dk], k [b,nk,h*dk], v [b,nk,h*dv]  ->  out [b,nq,h*dv]
 *   score = q.k / sqrt(dk); masked (mask != 0) scores become -inf; with geometry [b,h,nq,nk]
 *   score = log(max(geometry, 1e-6)) + score; softmax over keys; out = P v.
 *   mask element (b,iq,ik) is mask[b*mask_sb + iq*mask_sq + ik] (mask_sq = 0 broadcasts over
 *   queries), NULL = no mask.  Memory slots (mem_k [m,h*dk], mem_v [m,h*dv], may be NULL):
 *   m extra keys mem_scale_k*mem_k and values mem_scale_v*mem_v appended after the nk real
 *   keys and never masked.  dk, dv multiples of 4 and <= 64; any nq, nk, m (the reference has no limit either:
 *   attentions.py:44-58): up to 192 keys (nk + m) and 128 queries the scores of a query stay in registers; beyond that the
 *   keys pass in tiles of 128, ascending, under an online softmax (a fixed order: results depend on the operands only).
 *   A query whose keys are all masked gets NaN, as torch.softmax over a row of -inf does.
 * Replaces attentions.py:51-55 (plain), :102-111 (geometry), :171-183 (memory). */
int ovc_attention(const float* q, const float* k, const float* v, int b, int nq, int nk, int h,
                  int dk, int dv, const uint8_t* mask, long mask_sb, long mask_sq,
                  const float* geometry, const float* mem_k, const float* mem_v, int m,
                  float mem_scale_k, float mem_scale_v, float* out, ovc_stream stream);

/* mask[r] = (sum_f x[r,f] == 0)  -- models/utils.py:48-61 on feature rows. */
int ovc_zero_row_mask(const float* x, int rows, int d, uint8_t* mask, ovc_stream stream);

/* pe[b,n,c] : DETR-style 1-D sinusoid, models/modules/pos_embeddings.py:58-72.
 * mask [b,n] or NULL; normalize != 0 divides the position by (last + 1e-6) and multiplies by
 * scale. */
int ovc_region_position_encoding(const uint8_t* mask, int b, int n, int d, float temperature,
                                 int normalize, float scale, float* pe, ovc_stream stream);

/* y[r,:] = table[tokens[r],:] + pos_table[positions[r],:] (pos_table/positions may be NULL).
 * table has table_rows rows, pos_table pos_rows; an index outside its table reads the nearest valid row
 * (nn.Embedding raises instead -- validate on the host where that matters; the device never reads out
 * of bounds).  Replaces text_embeddings.py:28 and decoders.py:111-112. */
int ovc_embed(const int64_t* tokens, const int64_t* positions, const float* table, int table_rows,
              const float* pos_table, int pos_rows, float* y, int rows, int d, ovc_stream stream);

/* y = a * sigmoid(g)  -- attentions.py:313-315. */
int ovc_sigmoid_gate(const float* a, const float* g, float* y, long n, ovc_stream stream);

/* acc_out = (acc_in + sigmoid(alpha) * x) / divisor; acc_in may be NULL -- decoders.py:60-68
 * (divisor = sqrt(levels) on the last level, 1 otherwise). */
int ovc_gated_accumulate(const float* acc_in, const float* alpha, const float* x, float divisor,
                         float* acc_out, long n, ovc_stream stream);

/* y[r,:] = x[r,:] - logsumexp(x[r,:])  -- decoders.py:123. */
int ovc_log_softmax(const float* x, float* y, int rows, int n, ovc_stream stream);

/* w[b,h,i,j] = relu(fc_w[h,:] . emb(box_i, box_j) + fc_b[h]) with emb the 4 log-ratio
 * features (trig = 0, d_g = 4) or their sin/cos embedding (trig != 0, d_g multiple of 8).
 * Replaces models/utils.py:156-216 + encoders.py:93-101. */
int ovc_box_relation_weights(const float* boxes, int b, int n, const float* fc_w,
                             const float* fc_b, int h, int d_g, int trig, float* w,
                             ovc_stream stream);

/* One beam-search selection step for B images -- beam_search.py:45-59.
 *   logp [B,width,V] log-probabilities; running [B,width]; alive [B,width] (1.0 / 0.0, already
 *   multiplied by "previous word != eos"); selects the k best of the width*V candidates per
 *   image (ties: lower flat index first, as torch's stable sort).  Frozen beams (alive = 0)
 *   offer word 0 at their running score and -999 elsewhere; their rows of logp are rewritten
 *   as logp*0 when masked_logp != NULL.  chosen [B,k] int64 flat indices, score [B,k].
 *   scratch: >= 8*B*width*k bytes of device memory (16-byte aligned) for the per-row candidates the
 *   two passes exchange (one workgroup per beam row, then a k-way merge per image). */
int ovc_beam_select(const float* logp, const float* running, const float* alive, int B,
                    int width, int V, int k, int64_t* chosen, float* score, float* masked_logp,
                    void* scratch, size_t scratch_bytes, ovc_stream stream);

/* ======================================================================================
 * Engine level: the fused hot path  (models/base_transformer.py:45-53 and everything below it)
 * ==================================================================================== */

typedef struct {
    const float* w; const float* b;   /* nn.Linear weight [out, in], bias [out] or NULL                       */
    const void* planes;               /* split-precision modes only (ovc_model::precision > 0), optional: `w` cut by
                                         ovc_split_weight in that mode (the feature projection of mode 4: mode 3) -- the
                                         engine's GEMMs then read the planes instead of cutting w in every workgroup.
                                         Same bits either way.  The host re-cuts them when it changes w.  NULL = none. */
} ovc_lin;
typedef struct { const float* g; const float* b; } ovc_norm;     /* nn.LayerNorm weight, bias */

typedef struct {
    ovc_lin  q, k, v, o;              /* attention.fc_q / fc_k / fc_v / fc_o              */
    ovc_norm ln;                      /* layer_norm                                        */
    ovc_lin  aoa_i, aoa_g;            /* informative_attention / gated_attention or NULLs  */
    const float* m_k;                 /* attention.m_k [m, h*dk] or NULL                   */
    const float* m_v;                 /* attention.m_v [m, h*dv] or NULL                   */
} ovc_mha;

typedef struct { ovc_lin fc1, fc2; ovc_norm ln; } ovc_ffn;
typedef struct { ovc_mha att; ovc_ffn ffn; } ovc_enc_layer;
typedef struct {
    ovc_mha self_att, cross_att;
    ovc_ffn ffn;
    ovc_lin alpha[OVC_MAX_LEVELS];    /* fc_alphas (meshed decoder) or NULLs               */
} ovc_dec_layer;

enum { OVC_ENC_PLAIN = 0, OVC_ENC_MULTILEVEL = 1, OVC_ENC_GEOMETRIC = 2 };
enum { OVC_DEC_PLAIN = 0, OVC_DEC_MESHED = 1 };

typedef struct {
    int32_t abi;                      /* = ovc_abi_version()                               */
    int32_t enc_kind, dec_kind;
    int32_t d_feat, d_model, heads, d_k, d_v, d_ff;
    int32_t n_enc, n_dec, n_levels;   /* n_levels = 1 unless dec_kind == OVC_DEC_MESHED     */
    int32_t memory;                   /* memory slots in encoder self-attention (0 = none)  */
    int32_t trig, d_g;                /* geometric encoder                                  */
    int32_t vocab, max_len, pad_idx, bos_idx, eos_idx;
    float   ln_eps;
    ovc_lin  proj;                    /* vision_embedding.proj                              */
    ovc_norm enc_ln;                  /* encoder.layer_norm                                 */
    const float* fc_g_w;              /* encoder.fc_gs stacked [h, d_g] or NULL             */
    const float* fc_g_b;              /* [h]                                                */
    ovc_enc_layer enc[OVC_MAX_LAYERS];
    ovc_dec_layer dec[OVC_MAX_LAYERS];
    const float* word_emb;            /* decoder.word_emb.components.weight [V, d]          */
    const float* pos_emb;             /* decoder.pos_emb.weight [max_len+1, d]              */
    const float* fc;                  /* decoder.fc.weight [V, d] (no bias)                 */
    const void*  fc_planes;           /* its planes (see ovc_lin::planes) or NULL           */
    int32_t tune_objective;           /* which GEMM tuning table the engine consults: 0 / 1 = tilings measured in
                                         isolation, c > 1 = measured with c co-running copies (ovc_gemm_tune_objective)
                                         -- for hosts that keep several batches in flight on different streams.
                                         Speed only: all tilings of a K-order class give the same bits.            */
    int32_t precision;                /* 0 = fp32 MFMA everywhere: the parity mode and the only one the headline numbers
                                         use.  OPT-IN, uncredited split precision (fp32 in, fp32 out; attention / LayerNorm /
                                         selection unchanged; low-order bits differ from mode 0):
                                         3 = "bf16x6": every GEMM cuts its fp32 operands into three bf16 planes and contracts
                                         them on the 16-bit matrix path with fp32 accumulation (6 plane products);
                                         4 = "f16x3": two fp16 planes with a scaled residual, 3 products.  Weights must lie in
                                         fp16's range (checked by the host); the feature projection takes mode 3, so features
                                         need not; any other activation outside +-65504 SATURATES at that value while it is
                                         cut (never inf / NaN).  K-order classes 103 / 104.  1 and 2 (the one- and two-plane
                                         bf16 modes of ABI 5) failed the parity bar and no longer exist: OVC_EINVAL.   */
} ovc_model;

/* Sizes the engine accepts (anything else: ovc_workspace_bytes returns 0, the calls OVC_EINVAL) -- the
 * reference itself has no such limits, these are the template instances built so far:
 *   regions N <= OVC_MAX_REGIONS (1024), memory slots on top of them without a limit of their own (N <= 128 with
 *   N + memory <= 192 runs on the register-resident attention instances -- the shipped meshed_memory_transformer.yaml,
 *   MEMORY: 40, for every N <= 128 -- anything larger on the key-tiled ones);  beam k <= OVC_MAX_BEAM (8);  max_len <= 64;  any vocabulary (above 16384 words the
 *   selection streams each row k + 2 times instead of holding it in registers);
 *   d_model <= 2048 (multiple of 4; of 32 for models with AoA gates or the meshed decoder, whose products over a
 *   concatenated input read the two halves from their own buffers);  d_k == d_v in {4, 8, 16, 32, 64}, heads <= 32,
 *   heads*d_k a multiple of 64 and <= 1024;  layers <= OVC_MAX_LAYERS (8);  meshed levels <= OVC_MAX_LEVELS (4) and equal to the
 *   number of encoder layers (the multilevel encoder emits one level per layer).
 * tests/test_engine_gpu.py::test_unusual_dimensions_against_oracle runs each limit against the CPU oracle,
 * tests/test_fuzz_gpu.py a seeded random sweep of the space in between.
 *
 * Bytes of scratch the engine needs for batch B, N regions, beam k (return_probs adds the
 * [B,k,T,V] buffer).  0 on invalid arguments. */
size_t ovc_workspace_bytes(const ovc_model* m, int B, int N, int k, int return_probs);

/* vision_embedding + encoder: features [B,N,d_feat] (zero rows = padding), boxes [B,N,4] or
 * NULL -> enc_out [B,N,d] (multilevel: [B,levels,N,d]), mask_out [B,N].
 * Replaces *.encoder_forward (models/standard_stransformer.py:33-42 etc.). */
int ovc_encode(const ovc_model* m, const float* features, const float* boxes, int B, int N,
               void* workspace, size_t workspace_bytes, float* enc_out, uint8_t* mask_out,
               ovc_stream stream);

/* Encoder + max_len beam-search steps + final ordering, all on the device.
 *   ids_out  [B, out_size, max_len] int64,  logp_out [B, out_size, max_len] fp32,
 *   all_logp_out [B, k, max_len, V] or NULL (return_probs).
 * Replaces BaseTransformer.beam_search (models/base_transformer.py:45-53) and
 * BeamSearch.apply/iter/select/_expand_state (models/modules/beam_search.py:19-118). */
int ovc_beam_search(const ovc_model* m, const float* features, const float* boxes, int B, int N,
                    int k, int out_size, void* workspace, size_t workspace_bytes,
                    int64_t* ids_out, float* logp_out, float* all_logp_out, ovc_stream stream);

/* Same result as ovc_beam_search (without all_logp_out), issued as a hipGraph: the kernels that read
 * the caller's features / boxes run as plain launches, everything else (encoder layers, every
 * decode step, final ordering: ~740 launches whose arguments depend only on the model, the shapes
 * and the workspace) is captured on the second call for a given (model contents, B, N, k, out_size,
 * workspace) and replayed by hipGraphLaunch afterwards.  Graphs are cached process-wide, for the ONE device the
 * process is bound to (see the top of this header; the workspace pointer in the key is a device address)
 * (ovc_graph_cache_clear releases them); the workspace must stay allocated while they exist. */
int ovc_beam_search_graph(const ovc_model* m, const float* features, const float* boxes, int B, int N,
                          int k, int out_size, void* workspace, size_t workspace_bytes,
                          int64_t* ids_out, float* logp_out, ovc_stream stream);
int ovc_graph_cache_clear(void);

/* Same result again, WITHOUT the steps nobody needs.  The reference always runs max_len steps (models/modules/beam_search.py:94-95),
 * although once every beam of every image has emitted <eos> a step only appends word 0 / log-prob 0 to every beam and re-orders the
 * beams by score once -- which the final ordering does anyway (beam_search.py:49-55, 97-113).  This entry point issues the search
 * step by step (one captured graph per step from the second call of a shape on), lets the update kernel of each step count the
 * beams still alive, reads that count on the host ONE STEP LATE (the next step is already queued: the GPU never idles) and stops
 * issuing steps once it reads 0; the final ordering emits word 0 / log-prob 0 for the positions never written.  ids_out / logp_out
 * equal ovc_beam_search_graph's (assuming no total score below -999, the score of a frozen beam's other candidates,
 * beam_search.py:54); an image without a single valid region (NaN logits, arbitrary words) counts as ended and its arbitrary
 * words may differ.  *steps_run_out (host memory, may be NULL) receives the steps issued, 2 .. max_len.
 * Unlike every other entry point this one BLOCKS the calling host thread (hipEventSynchronize) until the search is at most one
 * step from its end: hosts that keep several batches in flight on different streams drive each stream from its own thread (the
 * library is thread-safe; ctypes releases the GIL).  Pinned host memory (max_len ints) and one event per step are kept per
 * (model, shape, workspace) next to the graphs and released with them.  No return_probs form. */
int ovc_beam_search_early(const ovc_model* m, const float* features, const float* boxes, int B, int N,
                          int k, int out_size, void* workspace, size_t workspace_bytes,
                          int64_t* ids_out, float* logp_out, int* steps_run_out, ovc_stream stream);

/* Optional device timing of the engine's GEMM launches (bench.py's roofline leg).  While enabled,
 * every GEMM launch carries a pair of hipEvents on its launch stream (hipExtLaunchKernelGGL start /
 * stop events, i.e. the dispatch's own begin / end timestamps, the quantity rocprofv3 reports as
 * the kernel duration).  ovc_profile_overhead_ms reports the duration of an empty
 * hipEventRecord pair for reference.  ovc_profile_read synchronises the events and
 * returns launches, total milliseconds and total algorithmic FLOPs (2*M*N*K), either per GEMM
 * class (kind 0: 0 feature projection, 1 encoder, 2 decoder projections/FFN, 3 vocabulary) or per
 * kernel instance (kind 1: tiling index, name from ovc_profile_kernel_name).  Enabling resets. */
#define OVC_PROFILE_CLASSES 4
int ovc_profile_enable(int on);
int ovc_profile_read(int kind, int index, int64_t* launches, double* total_ms, double* total_flops);
double ovc_profile_overhead_ms(void);
const char* ovc_profile_kernel_name(int tiling);   /* "" past the last tiling */

/* GEMM K-order classes.  fp32 addition is not associative and beam search decides on fp32 comparisons, so the
 * order in which a product sums over K is part of its definition here, never a tuning outcome:
 *   kchains = 1   one fmaf chain over k (ovc_linear; the engine's M = B*N encoder-side products, and the fp32 engine's
 *                 vocabulary projection, which runs transposed -- M = V rows);
 *   kchains = 4   four interleaved chains summed in chain order (the engine's M = B*beam decode-step products; products
 *                 of up to 112 rows -- the reference's own loop decodes one image at a time -- have instances on 16-row
 *                 tiles, gemm_rows16_f32: v_mfma_f32_16x16x4_f32 carries the same fma chain per element, same bits);
 *   ksplit  = s   K cut into s contiguous slices whose raw partial products the consuming LayerNorm sums in
 *                 slice order (engine only; a fixed function of K).
 *   kchains = 103, 104     the opt-in split-precision classes (ovc_model::precision = 3, 4): one chain of
 *                 16-deep 16-bit MFMA steps, plane products in a fixed order.
 * All tilings of one class produce bit-identical results, so token ids do not depend on the batch size, on the
 * GPU box or on what a timing run picked (the reference is deterministic on CPU: torch.sort path,
 * models/modules/beam_search.py:36-39).
 *
 * ovc_gemm_tune measures every tiling OF THE GIVEN CLASS on the shape y[M, nseg*seg_n] = x[M,K] W^T (nseg weight
 * segments of seg_n rows; ksplit > 1 needs nseg == 1) and remembers the fastest for this process; later GEMMs of
 * that shape and class use it, and shapes whose M is within a factor of two of a measured one (or, single-segment products with
 * the same M, whose seg_n is) borrow its entry.
 * scratch: >= 4*(M*K + nseg*seg_n*K + ksplit*M*nseg*seg_n) + 64 bytes of device memory (contents are used as
 * operands); for the split-precision classes, nseg * ovc_split_weight_bytes(seg_n, K, kchains - 100) more bytes make the
 * measurement use pre-cut weight planes (what the engine runs when ovc_lin::planes are set).  `epilogue`: 0 = the plain
 * product; 1 / 2 = measured with the log-softmax epilogue the engine's vocabulary projection carries (1: row-major form,
 * 8 * M * (seg_n / 32 + 4) more scratch bytes; 2: transposed form, 8 * seg_n * (M / 32 + 4)) -- the seventh value
 * ovc_engine_gemm_shapes reports.
 * `objective` (1..8) = what is minimised: the time of that many identical products co-running in one launch.  1 ranks
 * tilings by isolated latency, which favours many small tiles; with several independent batches in flight on different
 * streams, rank with objective = that number: fewer, larger tiles then win because they spend fewer CU-seconds and less
 * L2 traffic per FLOP.  Each objective has its own table, named explicitly in every call (ABI 6: there is no process-wide
 * "current objective" any more, so host threads with different objectives cannot cross their entries); which table an
 * engine call consults is ovc_model::tune_objective.
 * SYNCHRONISES the stream -- set-up time only.  Thread-safe. */
int ovc_gemm_tune(int M, int seg_n, int nseg, int K, int kchains, int ksplit, int objective, int epilogue, void* scratch,
                  size_t scratch_bytes, ovc_stream stream);
long ovc_gemm_tune_calls(void);        /* measurements run so far in this process */

/* Read / preset the remembered tiling of (shape, class, objective): lets a host persist tuning results.  get returns
 * the tiling index or -1; near != 0 also accepts the entry of the same product with the closest M -- or, M equal and nseg == 1,
 * the closest seg_n -- within a factor of two (what a launch falls back to).  set refuses a tiling of another class. */
int ovc_gemm_tuned_get(int M, int seg_n, int nseg, int K, int kchains, int ksplit, int objective, int near);
int ovc_gemm_tuned_set(int M, int seg_n, int nseg, int K, int kchains, int ksplit, int objective, int tiling);

/* The distinct GEMMs the engine issues for batch B, N regions, beam k: up to `capacity` records of seven int32
 * (M, seg_n, nseg, K, kchains, ksplit, epilogue) are written to `shapes`; returns the number of distinct shapes (which may
 * exceed capacity) or a negative OVC_E* code.  Host only: no launch, no device access. */
int ovc_engine_gemm_shapes(const ovc_model* m, int B, int N, int k, int32_t* shapes, int capacity);

/* hipGraph cache housekeeping: entries are evicted least-recently-used beyond OVC_GRAPH_CACHE_MAX (default 24);
 * a host that frees or replaces a workspace must drop that workspace's graphs first. */
int ovc_graph_cache_drop_workspace(const void* workspace);   /* returns the number of entries dropped */
int ovc_graph_cache_size(void);

/* Debug / measurement hooks (tools/, tests/).  ovc_debug_force_gemm_tiling: every following GEMM of the forced
 * tiling's class uses it (-1 restores the automatic choice); process-wide, not for use while other threads decode.
 * ovc_debug_linear_tiling: y = x W^T + bias by ONE named tiling (its class follows from the tiling; the name is
 * ovc_profile_kernel_name(tiling)); with ksplit > 1, y receives the ksplit raw partial products [ksplit][M][N];
 * `iters` back-to-back launches. */
int ovc_debug_force_gemm_tiling(int tiling);
int ovc_debug_clear_tuning(void);                 /* forget every remembered tiling (tests) */
int ovc_debug_linear_tiling(const float* x, int K, const float* W, const float* bias, float* y, int M, int N,
                            int tiling, int ksplit, int iters, ovc_stream stream);

/* Test hook: ONE selection step of the engine's fused path on caller-supplied decoder outputs x [B*width, d] -- the
 * vocabulary product fc [V, d] with its log-softmax epilogue (transposed != 0: logits^T = fc . x^T as the fp32 engine runs
 * it; 0: the row-major form; kchains = 1 or 4: the fp32 K-order class of the product -- the engine runs the transposed form
 * with one chain -- and ovc_debug_force_gemm_tiling can pin one tiling of that class) and the fused select + update kernel,
 * which never reads the logits back -- against which a
 * stable sort can be checked at the operator level.  chosen [B, k] = flat indices beam * V + word in winning order,
 * score [B, k]; scratch >= ovc_debug_vocab_select_bytes(B, width, V, k), 16-byte aligned.  V <= 16384. */
size_t ovc_debug_vocab_select_bytes(int B, int width, int V, int k);
int ovc_debug_vocab_select(const float* x, const float* fc, const float* running, const float* alive, int B, int width,
                           int V, int d, int k, int transposed, int kchains, void* scratch, size_t scratch_bytes, int64_t* chosen,
                           float* score, ovc_stream stream);

/* Split-precision modes: a weight W [N, K] (K a multiple of 16) cut ONCE into the 16-bit planes of `mode` (3 or 4, as
 * ovc_model::precision), stored in MFMA-operand order so that the GEMM's waves read them straight from memory instead of
 * cutting W again in every workgroup.  ovc_split_weight_bytes = size of `planes` (0 = invalid arguments); the planes hold the
 * same bits the kernel would cut, so results do not change.  ovc_debug_linear_planes = ovc_debug_linear_tiling on them. */
size_t ovc_split_weight_bytes(int N, int K, int mode);
int ovc_split_weight(const float* W, int N, int K, int mode, void* planes, ovc_stream stream);
int ovc_debug_linear_planes(const float* x, int K, const float* W, const void* planes, const float* bias, float* y,
                            int M, int N, int tiling, int ksplit, int iters, ovc_stream stream);
/* `iters` back-to-back launches of y = x W^T + bias (x [M,K], W [N,K]) with no host work between. */
int ovc_debug_repeat_linear(const float* x, int K, const float* W, const float* bias, float* y,
                            int M, int N, int iters, ovc_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* OVC_H_ */
