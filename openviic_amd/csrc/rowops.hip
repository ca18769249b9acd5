// Row-wise and element-wise kernels of the captioning path (all HBM-bound; one 64-lane wave per
// row, 16-byte accesses, wave-shuffle reductions).  Reference call sites: include/ovc.h.
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------
// LayerNorm(x + residual) * gamma + beta + add, optional row zeroing.  One wave per row, the row
// lives in registers (kVecs float4 per lane, d <= 64 * 4 * kMaxVec), two-pass mean / variance like
// ATen's CPU kernel.
//
// With kParts > 1, x holds kParts partial products of a K-split GEMM (part_stride floats apart): the row is
// their sum in slice order, plus `bias` -- the epilogue the split GEMM could not apply.
//
// Everything a row needs (slices, bias, residual) is loaded before the first use: the operand set is a
// template parameter, because hipcc turns every run-time "load or skip" into a branch with a full
// s_waitcnt, which serialised the 3..6 loads of a row into as many memory round trips.
// ---------------------------------------------------------------------------------------------
constexpr int kMaxVec = 8;   // float4 per lane -> d <= 2048

template <int kVecs, int kParts, bool kBias, bool kRes>
__global__ __launch_bounds__(256) void layer_norm_rows(const float* __restrict__ x, long part_stride,
                                                       const float* __restrict__ bias, const float* __restrict__ residual,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ add, int add_rows,
                                                       const uint8_t* __restrict__ zero_rows, float eps,
                                                       float* __restrict__ y, int rows, int d) {
    // every argument in ONE scalar round trip (hipcc otherwise fetches `rows` for the guard below first and the pointers in a
    // second, dependent round: the kernel is nothing but a chain of round trips -- gemm.hip, round 4)
    asm volatile("" ::"s"(x), "s"(part_stride), "s"(bias), "s"(residual), "s"(gamma), "s"(beta), "s"(add), "s"(add_rows), "s"(zero_rows),
                 "s"(eps), "s"(y), "s"(rows), "s"(d));
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nvec = d >> 2;
    float* yrow = y + (size_t)row * d;
    // The row's "write zeros instead" flag travels WITH the row's loads and is applied as a select on the way out.  (Until round 4
    // this was an early `if (cleared) { store zeros; return; }` behind the loads in the source -- hipcc moved the test in front of
    // them and waited for the byte: one whole extra round trip in every launch that takes the flags.)
    // (an unconditional load from an always-valid address: a load inside `if (zero_rows)` gets a full s_waitcnt at the join)
    const uint8_t cleared_byte = *(zero_rows ? zero_rows + row : reinterpret_cast<const uint8_t*>(gamma));
    const bool cleared = zero_rows != nullptr && cleared_byte != 0;
    // column group of (lane, i), clamped in-range: out-of-range lanes load a valid address and are masked later
    int col[kVecs];
#pragma unroll
    for (int i = 0; i < kVecs; ++i) col[i] = min(lane + i * 64, nvec - 1);
    f32x4 part[kParts][kVecs], bv[kVecs], rv[kVecs];
#pragma unroll
    for (int s = 0; s < kParts; ++s)
#pragma unroll
        for (int i = 0; i < kVecs; ++i)
            part[s][i] = reinterpret_cast<const f32x4*>(x + s * part_stride + (size_t)row * d)[col[i]];
    if (kBias) {
#pragma unroll
        for (int i = 0; i < kVecs; ++i) bv[i] = reinterpret_cast<const f32x4*>(bias)[col[i]];
    }
    if (kRes) {
#pragma unroll
        for (int i = 0; i < kVecs; ++i) rv[i] = reinterpret_cast<const f32x4*>(residual + (size_t)row * d)[col[i]];
    }
    // gamma / beta do not depend on the row's moments: fetched now, not after them (one memory round trip less
    // on a kernel that is nothing but a chain of them)
    f32x4 gv[kVecs], bev[kVecs];
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
        gv[i] = reinterpret_cast<const f32x4*>(gamma)[col[i]];
        bev[i] = reinterpret_cast<const f32x4*>(beta)[col[i]];
    }
    f32x4 v[kVecs];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
        v[i] = part[0][i];
#pragma unroll
        for (int s = 1; s < kParts; ++s) v[i] += part[s][i];
        if (kBias) v[i] += bv[i];
        if (kRes) v[i] += rv[i];
        if (lane + i * 64 < nvec) sum += (v[i][0] + v[i][1]) + (v[i][2] + v[i][3]);
    }
    const float mean = wave_sum(sum) / (float)d;
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
        if (lane + i * 64 < nvec) {
            const f32x4 t = v[i] - mean;
            sq += (t[0] * t[0] + t[1] * t[1]) + (t[2] * t[2] + t[3] * t[3]);
        }
    }
    const float rstd = 1.0f / sqrtf(wave_sum(sq) / (float)d + eps);
    const f32x4* ar = add ? reinterpret_cast<const f32x4*>(add + (size_t)(row % add_rows) * d) : nullptr;
#pragma unroll
    for (int i = 0; i < kVecs; ++i) {
        const int c = lane + i * 64;
        if (c < nvec) {
            f32x4 o = (v[i] - mean) * rstd * gv[i] + bev[i];
            if (ar) o += ar[c];
            if (cleared) o = f32x4{0.f, 0.f, 0.f, 0.f};
            reinterpret_cast<f32x4*>(yrow)[c] = o;
        }
    }
}

template <int kParts, bool kBias, bool kRes>
int launch_layer_norm(const float* x, long part_stride, const float* bias, const float* residual, const float* gamma,
                      const float* beta, const float* add, int add_rows, const uint8_t* zero_rows, float eps, float* y,
                      int rows, int d, hipStream_t stream) {
    const int vecs = ((d >> 2) + 63) / 64;
    const dim3 grid((rows + 3) / 4), block(256);
#define OVC_LN(V) hipLaunchKernelGGL((layer_norm_rows<V, kParts, kBias, kRes>), grid, block, 0, stream, x, part_stride, bias, \
                                     residual, gamma, beta, add, add_rows, zero_rows, eps, y, rows, d)
    if (vecs <= 1) OVC_LN(1); else if (vecs <= 2) OVC_LN(2); else if (vecs <= 4) OVC_LN(4); else OVC_LN(8);
#undef OVC_LN
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

// mask[r] = (sum_f x[r,f] == 0): one wave per row, 16-byte coalesced loads.
__global__ __launch_bounds__(256) void zero_row_mask_kernel(const float* __restrict__ x, int rows, int d, uint8_t* __restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * d);
    float s = 0.f;
    for (int c = lane; c < (d >> 2); c += 64) {
        const f32x4 t = xr[c];
        s += (t[0] + t[1]) + (t[2] + t[3]);
    }
    s = wave_sum(s);
    if (lane == 0) mask[row] = (s == 0.f) ? 1 : 0;
}

// DETR-style sinusoid over the region index; one thread per output element.
__global__ void region_pe_kernel(const uint8_t* __restrict__ mask, int b, int n, int d, float temperature,
                                 int normalize, float scale, float* __restrict__ pe) {
    const long total = (long)b * n * d;
    const long idx = blockIdx.x * (long)blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % d);
    const int i = (int)((idx / d) % n);
    const int bi = (int)(idx / ((long)n * d));
    float pos, last;
    if (mask) {
        int cnt = 0, tot = 0;
        for (int j = 0; j < n; ++j) {
            const int keep = mask[bi * n + j] ? 0 : 1;
            tot += keep;
            if (j <= i) cnt += keep;
        }
        pos = (float)cnt; last = (float)tot;
    } else {
        pos = (float)(i + 1); last = (float)n;
    }
    if (normalize) pos = pos / (last + 1e-6f) * scale;
    const float div = powf(temperature, (float)(2 * (c / 2)) / (float)d);
    const float ang = pos / div;
    pe[idx] = (c & 1) ? cosf(ang) : sinf(ang);
}

__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ tokens, const int64_t* __restrict__ positions,
                                                    const float* __restrict__ table, int table_rows,
                                                    const float* __restrict__ pos_table, int pos_rows,
                                                    float* __restrict__ y, int rows, int d) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    // indices are clamped into their tables: a bad token id must not become a wild device read
    const int64_t tok = min(max(tokens[row], (int64_t)0), (int64_t)table_rows - 1);
    const f32x4* e = reinterpret_cast<const f32x4*>(table + (size_t)tok * d);
    const f32x4* p = nullptr;
    if (pos_table && positions) {
        const int64_t pos = min(max(positions[row], (int64_t)0), (int64_t)pos_rows - 1);
        p = reinterpret_cast<const f32x4*>(pos_table + (size_t)pos * d);
    }
    f32x4* o = reinterpret_cast<f32x4*>(y + (size_t)row * d);
    for (int c = lane; c < (d >> 2); c += 64) o[c] = p ? e[c] + p[c] : e[c];
}

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ void sigmoid_gate_kernel(const float* __restrict__ a, const float* __restrict__ g, float* __restrict__ y, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 av = reinterpret_cast<const f32x4*>(a)[i], gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 o;
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = av[j] * sigmoidf_(gv[j]);
        reinterpret_cast<f32x4*>(y)[i] = o;
    }
}

__global__ void gated_accumulate_kernel(const float* __restrict__ acc_in, const float* __restrict__ alpha,
                                        const float* __restrict__ x, float divisor, float* __restrict__ acc_out, long n4) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        const f32x4 al = reinterpret_cast<const f32x4*>(alpha)[i], xv = reinterpret_cast<const f32x4*>(x)[i];
        f32x4 o = acc_in ? reinterpret_cast<const f32x4*>(acc_in)[i] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = (o[j] + sigmoidf_(al[j]) * xv[j]) / divisor;
        reinterpret_cast<f32x4*>(acc_out)[i] = o;
    }
}

// Meshed-decoder mix (decoders.py:59-68): out = ((s(a0) e0 + s(a1) e1) + ...) / sqrt(levels), accumulated in the
// reference's order; alpha and enc are [levels][n] stacked.
__global__ void meshed_mix_kernel(const float* __restrict__ alpha, const float* __restrict__ enc, int levels, long n4,
                                  float divisor, float* __restrict__ out) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int l = 0; l < levels; ++l) {
            const f32x4 al = reinterpret_cast<const f32x4*>(alpha)[(size_t)l * n4 + i];
            const f32x4 xv = reinterpret_cast<const f32x4*>(enc)[(size_t)l * n4 + i];
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[j] = acc[j] + sigmoidf_(al[j]) * xv[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = acc[j] / divisor;
        reinterpret_cast<f32x4*>(out)[i] = acc;
    }
}

// y = x - logsumexp(x) per row; one 256-thread block per row (rows of ~10k vocabulary entries).
__global__ __launch_bounds__(256) void log_softmax_kernel(const float* __restrict__ x, float* __restrict__ y, int rows, int n) {
    __shared__ float red[4];
    const int row = blockIdx.x;
    const float* xr = x + (size_t)row * n;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < n; c += 256) mx = fmaxf(mx, xr[c]);
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < n; c += 256) s += expf(xr[c] - mx);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float lse = mx + logf((red[0] + red[1]) + (red[2] + red[3]));
    float* yr = y + (size_t)row * n;
    for (int c = threadIdx.x; c < n; c += 256) yr[c] = xr[c] - lse;
}

// Box-relation attention bias: w[b,h,i,j] = relu(fc_w[h,:] . emb(i,j) + fc_b[h]).
// One block per (b, i); thread per j.
__global__ void box_relation_kernel(const float* __restrict__ boxes, int n, const float* __restrict__ fc_w,
                                    const float* __restrict__ fc_b, int h, int d_g, int trig, float* __restrict__ w) {
    const int b = blockIdx.x / n, i = blockIdx.x - b * n;
    const float* bi = boxes + ((size_t)b * n + i) * 4;
    const float cxi = (bi[0] + bi[2]) * 0.5f, cyi = (bi[1] + bi[3]) * 0.5f;
    const float wi = (bi[2] - bi[0]) + 1.0f, hi = (bi[3] - bi[1]) + 1.0f;
    for (int j = threadIdx.x; j < n; j += blockDim.x) {
        const float* bj = boxes + ((size_t)b * n + j) * 4;
        const float cxj = (bj[0] + bj[2]) * 0.5f, cyj = (bj[1] + bj[3]) * 0.5f;
        const float wj = (bj[2] - bj[0]) + 1.0f, hj = (bj[3] - bj[1]) + 1.0f;
        float g[4];
        g[0] = logf(fmaxf(fabsf((cxi - cxj) / wi), 1e-3f));
        g[1] = logf(fmaxf(fabsf((cyi - cyj) / hi), 1e-3f));
        g[2] = logf(wi / wj);
        g[3] = logf(hi / hj);
        for (int hd = 0; hd < h; ++hd) {
            const float* fw = fc_w + (size_t)hd * d_g;
            float acc = fc_b[hd];
            if (!trig) {
                acc += fw[0] * g[0] + fw[1] * g[1] + fw[2] * g[2] + fw[3] * g[3];
            } else {
                // embedding = [sin(100 g_c f_q) for c, q] ++ [cos(...)], f_q = wave_len^(-q/(d_g/8))
                const int nf = d_g / 8;
                for (int c = 0; c < 4; ++c)
                    for (int q = 0; q < nf; ++q) {
                        const float freq = 1.0f / powf(1000.0f, (float)q / (float)nf);
                        const float ang = (100.0f * g[c]) * freq;
                        acc += fw[c * nf + q] * sinf(ang) + fw[d_g / 2 + c * nf + q] * cosf(ang);
                    }
            }
            w[(((size_t)b * h + hd) * n + i) * n + j] = fmaxf(acc, 0.f);
        }
    }
}

}  // namespace

extern "C" int ovc_layer_norm(const float* x, const float* residual, const float* gamma, const float* beta,
                              const float* add, int add_rows, const uint8_t* zero_rows, float eps,
                              float* y, int rows, int d, ovc_stream stream) {
    if (!x || !gamma || !beta || !y || rows <= 0 || d <= 0 || (d & 3) || d > 64 * 4 * kMaxVec) return OVC_EINVAL;
    if (add && add_rows <= 0) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    if (!ovc_aligned16(x) || !ovc_aligned16(y) || !ovc_aligned16(gamma) || !ovc_aligned16(beta) ||
        (residual && !ovc_aligned16(residual)) || (add && !ovc_aligned16(add))) return OVC_EINVAL;
    hipStream_t s = ovc_hip_stream(stream);
    return residual ? launch_layer_norm<1, false, true>(x, 0L, nullptr, residual, gamma, beta, add, add_rows, zero_rows, eps, y, rows, d, s)
                    : launch_layer_norm<1, false, false>(x, 0L, nullptr, nullptr, gamma, beta, add, add_rows, zero_rows, eps, y, rows, d, s);
}

int ovc_layer_norm_parts(const float* parts, int nparts, long part_stride, const float* bias, const float* residual,
                         const float* gamma, const float* beta, const uint8_t* zero_rows, float eps, float* y,
                         int rows, int d, hipStream_t stream) {
    if (!parts || !bias || !residual || !gamma || !beta || !y || rows <= 0 || d <= 0 || (d & 3) || d > 64 * 4 * kMaxVec) return OVC_EINVAL;
    if ((part_stride & 3) || !ovc_aligned16(parts) || !ovc_aligned16(y) || !ovc_aligned16(bias) || !ovc_aligned16(residual)) return OVC_EINVAL;
    if (nparts == 2) return launch_layer_norm<2, true, true>(parts, part_stride, bias, residual, gamma, beta, nullptr, 0, zero_rows, eps, y, rows, d, stream);
    if (nparts == 4) return launch_layer_norm<4, true, true>(parts, part_stride, bias, residual, gamma, beta, nullptr, 0, zero_rows, eps, y, rows, d, stream);
    return OVC_EINVAL;
}

extern "C" int ovc_zero_row_mask(const float* x, int rows, int d, uint8_t* mask, ovc_stream stream) {
    if (!x || !mask || rows <= 0 || d <= 0 || (d & 3) || !ovc_aligned16(x)) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    hipLaunchKernelGGL(zero_row_mask_kernel, dim3((rows + 3) / 4), dim3(256), 0, ovc_hip_stream(stream), x, rows, d, mask);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

extern "C" int ovc_region_position_encoding(const uint8_t* mask, int b, int n, int d, float temperature,
                                            int normalize, float scale, float* pe, ovc_stream stream) {
    if (!pe || b <= 0 || n <= 0 || d <= 0) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    const long total = (long)b * n * d;
    hipLaunchKernelGGL(region_pe_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ovc_hip_stream(stream), mask, b, n, d,
                       temperature, normalize, scale, pe);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

extern "C" int ovc_embed(const int64_t* tokens, const int64_t* positions, const float* table, int table_rows,
                         const float* pos_table, int pos_rows, float* y, int rows, int d, ovc_stream stream) {
    if (!tokens || !table || !y || rows <= 0 || d <= 0 || (d & 3) || table_rows <= 0) return OVC_EINVAL;
    if (pos_table && positions && pos_rows <= 0) return OVC_EINVAL;
    if (!ovc_aligned16(table) || !ovc_aligned16(y) || (pos_table && !ovc_aligned16(pos_table))) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    hipLaunchKernelGGL(embed_kernel, dim3((rows + 3) / 4), dim3(256), 0, ovc_hip_stream(stream), tokens, positions, table,
                       table_rows, pos_table, pos_rows, y, rows, d);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

static inline int elementwise_grid(long n4) {
    long blocks = (n4 + 255) / 256;
    return (int)(blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks));
}

extern "C" int ovc_sigmoid_gate(const float* a, const float* g, float* y, long n, ovc_stream stream) {
    if (!a || !g || !y || n <= 0 || (n & 3) || !ovc_aligned16(a) || !ovc_aligned16(g) || !ovc_aligned16(y)) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    hipLaunchKernelGGL(sigmoid_gate_kernel, dim3(elementwise_grid(n / 4)), dim3(256), 0, ovc_hip_stream(stream), a, g, y, n / 4);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

extern "C" int ovc_gated_accumulate(const float* acc_in, const float* alpha, const float* x, float divisor,
                                    float* acc_out, long n, ovc_stream stream) {
    if (!alpha || !x || !acc_out || n <= 0 || (n & 3)) return OVC_EINVAL;
    if (!ovc_aligned16(alpha) || !ovc_aligned16(x) || !ovc_aligned16(acc_out) || (acc_in && !ovc_aligned16(acc_in))) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    hipLaunchKernelGGL(gated_accumulate_kernel, dim3(elementwise_grid(n / 4)), dim3(256), 0, ovc_hip_stream(stream),
                       acc_in, alpha, x, divisor, acc_out, n / 4);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

int ovc_meshed_mix(const float* alpha, const float* enc, int levels, long n, float divisor, float* out, hipStream_t stream) {
    if (!alpha || !enc || !out || levels <= 0 || n <= 0 || (n & 3)) return OVC_EINVAL;
    hipLaunchKernelGGL(meshed_mix_kernel, dim3(elementwise_grid(n / 4)), dim3(256), 0, stream, alpha, enc, levels, n / 4, divisor, out);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

extern "C" int ovc_log_softmax(const float* x, float* y, int rows, int n, ovc_stream stream) {
    if (!x || !y || rows <= 0 || n <= 0) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    hipLaunchKernelGGL(log_softmax_kernel, dim3(rows), dim3(256), 0, ovc_hip_stream(stream), x, y, rows, n);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}

extern "C" int ovc_box_relation_weights(const float* boxes, int b, int n, const float* fc_w, const float* fc_b,
                                        int h, int d_g, int trig, float* w, ovc_stream stream) {
    if (!boxes || !fc_w || !fc_b || !w || b <= 0 || n <= 0 || h <= 0) return OVC_EINVAL;
    if ((!trig && d_g != 4) || (trig && (d_g <= 0 || (d_g & 7)))) return OVC_EINVAL;
    if (const int rc = ovc_device_guard()) return rc;          // one device per process (include/ovc.h)
    hipLaunchKernelGGL(box_relation_kernel, dim3(b * n), dim3(64), 0, ovc_hip_stream(stream), boxes, n, fc_w, fc_b, h, d_g,
                       trig, w);
    OVC_RETURN_IF_LAUNCH_FAILED();
    return OVC_OK;
}
