// Training-side streaming passes (round 4): what a training iteration at config [4]'s real size (K = 101 class graphs of
// 1024 vertices: a 404 MB edge_weights) spent in chains of library elementwise / reduce launches, one kernel each.
//   sn_pow2_scale            the power-of-two operand scale of split-fp16 planes from the largest magnitude of a tensor
//                            (ops.pow2_scale: abs, amax and nine scalar launches per operand, ten operands per iteration)
//   sn_sym_half_inplace      S <- (S + S^T) / 2 per graph: the chain rule through the GCN's (E + E^T)/2 + I
//                            (reference gnn.py:27-30) applied once to the SUM of the layers' dY . X^T products
//   sn_normalize_sum_rows    SchemaNet.normalize(): x <- clamp_min(x, m); x <- x / sum(x, -1); NaN -> 0; diagonal <- 0
//                            (reference schema_net.py:133-142, graph/utils.py:7-13; four passes over the parameter there)
// All HBM-bound, no MFMA: whole-line accesses, one read and one write of the tensor.
#include "sn_common.h"

namespace {

constexpr int kAmaxThreads = 256;

__device__ __forceinline__ unsigned wave_max_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned o = (unsigned)__shfl_xor((int)v, off, SN_WAVE);
        v = o > v ? o : v;
    }
    return v;
}

// the bit pattern of |x| orders like the magnitude, and every NaN sorts above inf: the maximum over the patterns is
// the amax of torch (NaN if any element is)
__global__ __launch_bounds__(kAmaxThreads) void amax_partial_kernel(const float *x, int64_t n, unsigned *partial)
{
    __shared__ unsigned red[kAmaxThreads / SN_WAVE];
    unsigned m = 0u;
    const int64_t n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;
    const uint4 *x4 = reinterpret_cast<const uint4 *>(x);
    for (int64_t i = (int64_t)blockIdx.x * kAmaxThreads + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kAmaxThreads) {
        const uint4 v = x4[i];
        const unsigned a = v.x & 0x7FFFFFFFu, b = v.y & 0x7FFFFFFFu, c = v.z & 0x7FFFFFFFu, d = v.w & 0x7FFFFFFFu;
        const unsigned ab = a > b ? a : b, cd = c > d ? c : d;
        const unsigned q = ab > cd ? ab : cd;
        m = q > m ? q : m;
    }
    for (int64_t i = n4 * 4 + (int64_t)blockIdx.x * kAmaxThreads + threadIdx.x; i < n; i += (int64_t)gridDim.x * kAmaxThreads) {
        const unsigned a = __float_as_uint(x[i]) & 0x7FFFFFFFu;
        m = a > m ? a : m;
    }
    m = wave_max_u32(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = red[0];
#pragma unroll
        for (int w = 1; w < kAmaxThreads / SN_WAVE; ++w) t = red[w] > t ? red[w] : t;
        partial[blockIdx.x] = t;
    }
}

// s = 2^clamp(floor(log2(top / b)), -60, 60), 1 for b == 0 or a non-finite b; floor(log2()) = the exponent field
// (top / b is a normal number or +inf for every finite b > 0: top / FLT_MAX > FLT_MIN)
__global__ __launch_bounds__(kAmaxThreads) void pow2_scale_finish_kernel(const unsigned *partial, int blocks, float top, float *scale, float *amax_out)
{
    __shared__ unsigned red[kAmaxThreads / SN_WAVE];
    unsigned m = 0u;
    for (int i = threadIdx.x; i < blocks; i += kAmaxThreads) m = partial[i] > m ? partial[i] : m;
    m = wave_max_u32(m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned t = red[0];
#pragma unroll
        for (int w = 1; w < kAmaxThreads / SN_WAVE; ++w) t = red[w] > t ? red[w] : t;
        const float b = __uint_as_float(t);
        float s = 1.0f;
        if (b > 0.0f && t < 0x7F800000u) {
            const float q = top / b;
            int e = q > 3.0e38f ? 60 : (int)((__float_as_uint(q) >> 23) & 0xFFu) - 127;
            e = e < -60 ? -60 : (e > 60 ? 60 : e);
            s = ldexpf(1.0f, e);
        }
        scale[0] = s;
        if (amax_out) amax_out[0] = b;
    }
}

// ------------------------------------------------------------------------------------------
// S <- (S + S^T) / 2, in place, per graph.  A workgroup holds the 64 x 64 tiles (I, J) and (J, I), J >= I, in LDS (row
// stride 65: the transposed reads are conflict-free) and writes both; only the T (T + 1) / 2 tile pairs are launched.
// n % 4 == 0 and a 16-byte aligned base: 16-byte accesses (a thread = 4 consecutive columns of 4 rows).
constexpr int kSymTile = 64, kSymLd = kSymTile + 1;

template <bool VEC>
__global__ __launch_bounds__(256) void sym_half_kernel(float *s, int n, int T)
{
    // pair index -> (I, J >= I): row I of the upper triangle starts at I T - I (I - 1) / 2
    int I = 0, rest = blockIdx.x;
    while (rest >= T - I) { rest -= T - I; ++I; }
    const int J = I + rest;
    __shared__ float ta[kSymTile * kSymLd], tb[kSymTile * kSymLd];
    float *g = s + (int64_t)blockIdx.y * n * n;
    const int ri = I * kSymTile, cj = J * kSymTile;
    if constexpr (VEC) {
        const int c = (threadIdx.x & 15) * 4, r0 = threadIdx.x >> 4;             // 16 threads per row, 16 rows per pass
#pragma unroll
        for (int r = r0; r < kSymTile; r += 16) {
            float4 va = {0.0f, 0.0f, 0.0f, 0.0f}, vb = va;
            if (ri + r < n && cj + c < n) va = *reinterpret_cast<const float4 *>(g + (int64_t)(ri + r) * n + cj + c);
            if (I != J && cj + r < n && ri + c < n) vb = *reinterpret_cast<const float4 *>(g + (int64_t)(cj + r) * n + ri + c);
            float *da = ta + r * kSymLd + c, *db = tb + r * kSymLd + c;
            da[0] = va.x; da[1] = va.y; da[2] = va.z; da[3] = va.w;
            if (I != J) { db[0] = vb.x; db[1] = vb.y; db[2] = vb.z; db[3] = vb.w; }
        }
        __syncthreads();
        const float *tt = I != J ? tb : ta;
#pragma unroll
        for (int r = r0; r < kSymTile; r += 16) {
            if (ri + r < n && cj + c < n) {
                const float *a = ta + r * kSymLd + c, *t = tt + c * kSymLd + r;
                float4 o;
                o.x = (a[0] + t[0]) * 0.5f; o.y = (a[1] + t[kSymLd]) * 0.5f; o.z = (a[2] + t[2 * kSymLd]) * 0.5f; o.w = (a[3] + t[3 * kSymLd]) * 0.5f;
                *reinterpret_cast<float4 *>(g + (int64_t)(ri + r) * n + cj + c) = o;
            }
            if (I != J && cj + r < n && ri + c < n) {
                const float *b = tb + r * kSymLd + c, *t = ta + c * kSymLd + r;
                float4 o;
                o.x = (b[0] + t[0]) * 0.5f; o.y = (b[1] + t[kSymLd]) * 0.5f; o.z = (b[2] + t[2 * kSymLd]) * 0.5f; o.w = (b[3] + t[3 * kSymLd]) * 0.5f;
                *reinterpret_cast<float4 *>(g + (int64_t)(cj + r) * n + ri + c) = o;
            }
        }
    } else {
        const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
#pragma unroll 4
        for (int r = r0; r < kSymTile; r += 4) {
            const bool in_a = ri + r < n && cj + c < n, in_b = cj + r < n && ri + c < n;
            ta[r * kSymLd + c] = in_a ? g[(int64_t)(ri + r) * n + cj + c] : 0.0f;
            if (I != J) tb[r * kSymLd + c] = in_b ? g[(int64_t)(cj + r) * n + ri + c] : 0.0f;
        }
        __syncthreads();
        const float *tt = I != J ? tb : ta;                      // the tile whose transpose pairs with (I, J)
#pragma unroll 4
        for (int r = r0; r < kSymTile; r += 4) {
            if (ri + r < n && cj + c < n) g[(int64_t)(ri + r) * n + cj + c] = (ta[r * kSymLd + c] + tt[c * kSymLd + r]) * 0.5f;
            if (I != J && cj + r < n && ri + c < n) g[(int64_t)(cj + r) * n + ri + c] = (tb[r * kSymLd + c] + ta[c * kSymLd + r]) * 0.5f;
        }
    }
}

// ------------------------------------------------------------------------------------------
// x[r, :] <- nan_to_num(clamp_min(x[r, :], m) / sum(clamp_min(x[r, :], m)), 0); x[r, r % diag_n] <- 0 if diag_n.
// One workgroup per row (grid-stride); a row of up to 4096 floats stays in registers between the sum and the division.
constexpr int kNormThreads = 256, kNormKeep = 4;

__device__ __forceinline__ float clamp_min_keep_nan(float v, float m) { return v < m ? m : v; }     // (NaN stays NaN: torch.clamp_min)
__device__ __forceinline__ float nan_to_num0(float v)
{
    if (v != v) return 0.0f;
    if (v > 3.4028234663852886e38f) return 3.4028234663852886e38f;
    if (v < -3.4028234663852886e38f) return -3.4028234663852886e38f;
    return v;
}

__global__ __launch_bounds__(kNormThreads) void normalize_sum_rows_kernel(float *x, int64_t rows, int n, float min_val, int diag_n)
{
    __shared__ float red[kNormThreads / SN_WAVE];
    const int tid = threadIdx.x;
    const bool vec = (n & 3) == 0 && n <= kNormThreads * 4 * kNormKeep && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    for (int64_t r = blockIdx.x; r < rows; r += gridDim.x) {
        float *row = x + r * n;
        float4 keep[kNormKeep];
        float acc = 0.0f;
        if (vec) {
#pragma unroll
            for (int u = 0; u < kNormKeep; ++u) {
                const int j = (u * kNormThreads + tid) * 4;
                if (j < n) {
                    float4 v = *reinterpret_cast<const float4 *>(row + j);
                    v.x = clamp_min_keep_nan(v.x, min_val); v.y = clamp_min_keep_nan(v.y, min_val);
                    v.z = clamp_min_keep_nan(v.z, min_val); v.w = clamp_min_keep_nan(v.w, min_val);
                    keep[u] = v;
                    acc += (v.x + v.y) + (v.z + v.w);
                }
            }
        } else {
            for (int j = tid; j < n; j += kNormThreads) acc += clamp_min_keep_nan(row[j], min_val);
        }
        acc = sn_wave_sum(acc);
        __syncthreads();                                         // (red[] of the previous row has been read)
        if ((tid & 63) == 0) red[tid >> 6] = acc;
        __syncthreads();
        const float total = (red[0] + red[1]) + (red[2] + red[3]);
        const int dcol = diag_n > 0 ? (int)(r % diag_n) : -1;
        if (vec) {
#pragma unroll
            for (int u = 0; u < kNormKeep; ++u) {
                const int j = (u * kNormThreads + tid) * 4;
                if (j < n) {
                    float4 v = keep[u];
                    v.x = nan_to_num0(v.x / total); v.y = nan_to_num0(v.y / total);
                    v.z = nan_to_num0(v.z / total); v.w = nan_to_num0(v.w / total);
                    if (dcol >= j && dcol < j + 4) {
                        if (dcol == j) v.x = 0.0f;
                        else if (dcol == j + 1) v.y = 0.0f;
                        else if (dcol == j + 2) v.z = 0.0f;
                        else v.w = 0.0f;
                    }
                    *reinterpret_cast<float4 *>(row + j) = v;
                }
            }
        } else {
            for (int j = tid; j < n; j += kNormThreads) {
                const float v = nan_to_num0(clamp_min_keep_nan(row[j], min_val) / total);
                row[j] = j == dcol ? 0.0f : v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// The GNN layer's tail in training: y = act(LayerNorm(mask(x))) (reference gnn.py:43-46) out of place - x is what the
// backward pass keeps - and its backward in ONE pass over (x, dy): the statistics are recomputed from x (the row is in
// registers anyway), dx written once, the column sums of d gamma / d beta kept in registers over the rows a wave walks
// and reduced block by block in a fixed order (a [blocks, 2, E] scratch + a finishing launch: no atomics, bit-reproducible).
// The library's chain was masked_fill, native_layer_norm, clamp forward and threshold_backward, two layer-norm backward
// kernels, a masked_fill backward: three passes forward, four back, over [G n, E] (106 MB at the Caltech configuration).
// One wave per row (sn_layernorm_row: the sums of the inference kernels), E <= 1024.
__global__ __launch_bounds__(256) void ln_act_forward_kernel(const float *x, float *y, int64_t rows, int n, int E, const int32_t *n_valid,
                                                             const float *gamma, const float *beta, float eps, int relu)
{
    const int lane = threadIdx.x & 63;
    float gm[SN_LN_MAX], bt[SN_LN_MAX];
    sn_layernorm_coeffs(gm, bt, lane, E, gamma, beta);
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int g = (int)(row / n), r = (int)(row % n);
        const bool masked = n_valid && r >= n_valid[g];
        const float *p = x + row * E;
        float v[SN_LN_MAX];
#pragma unroll
        for (int k = 0; k < SN_LN_MAX; ++k) {
            const int c = lane + SN_WAVE * k;
            v[k] = (c < E && !masked) ? p[c] : 0.0f;
        }
        sn_layernorm_row(v, lane, E, gm, bt, eps, relu);
        float *o = y + row * E;
#pragma unroll
        for (int k = 0; k < SN_LN_MAX; ++k) {
            const int c = lane + SN_WAVE * k;
            if (c < E) o[c] = v[k];
        }
    }
}

template <int KMAX>
__global__ __launch_bounds__(256) void ln_act_backward_kernel(const float *x, const float *dy, int64_t rows, int n, int E, const int32_t *n_valid,
                                                              const float *gamma, const float *beta, float eps, int relu, float *dx,
                                                              float *partial)
{
    __shared__ float red[4][2][SN_WAVE * KMAX];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    float gm[KMAX], bt[KMAX], dg[KMAX], db[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int c = lane + SN_WAVE * k;
        gm[k] = c < E ? gamma[c] : 0.0f;
        bt[k] = c < E ? beta[c] : 0.0f;
        dg[k] = 0.0f; db[k] = 0.0f;
    }
    for (int64_t row = (int64_t)blockIdx.x * 4 + wid; row < rows; row += (int64_t)gridDim.x * 4) {
        const int g = (int)(row / n), r = (int)(row % n);
        const bool masked = n_valid && r >= n_valid[g];
        const float *p = x + row * E, *q = dy + row * E;
        float v[KMAX], gy[KMAX];
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = lane + SN_WAVE * k;
            v[k] = (c < E && !masked) ? p[c] : 0.0f;
            gy[k] = c < E ? q[c] : 0.0f;
            s += v[k];
        }
        const float mean = sn_wave_sum(s) / (float)E;              // (the sums of sn_layernorm_row)
        float sq = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = lane + SN_WAVE * k;
            const float d = (c < E) ? v[k] - mean : 0.0f;
            sq += d * d;
        }
        const float rstd = 1.0f / sqrtf(sn_wave_sum(sq) / (float)E + eps);
        float a1 = 0.0f, a2 = 0.0f;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = lane + SN_WAVE * k;
            const float xh = (c < E) ? (v[k] - mean) * rstd : 0.0f;
            const float yv = xh * gm[k] + bt[k];
            const float gq = (relu && !(yv > 0.0f)) ? 0.0f : gy[k];       // threshold_backward: the gradient passes where y > 0
            db[k] += gq;
            dg[k] += gq * xh;
            const float dxh = gq * gm[k];
            a1 += dxh;
            a2 += dxh * xh;
            v[k] = xh; gy[k] = dxh;
        }
        const float m1 = sn_wave_sum(a1) / (float)E, m2 = sn_wave_sum(a2) / (float)E;
        float *o = dx + row * E;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = lane + SN_WAVE * k;
            if (c < E) o[c] = masked ? 0.0f : rstd * (gy[k] - m1 - v[k] * m2);     // (masked_fill's backward: no gradient into a padded row)
        }
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        red[wid][0][lane + SN_WAVE * k] = dg[k];
        red[wid][1][lane + SN_WAVE * k] = db[k];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 2 * SN_WAVE * KMAX; i += 256) {
        const int which = i / (SN_WAVE * KMAX), c = i % (SN_WAVE * KMAX);
        if (c < E) partial[((int64_t)blockIdx.x * 2 + which) * E + c] = (red[0][which][c] + red[1][which][c]) + (red[2][which][c] + red[3][which][c]);
    }
}

// out[which][c] = sum over blocks of partial[block][which][c]: sixteen slices of the blocks per column (a thread walks its
// slice in block order, eight loads in flight), the slices added in order.  (Four slices over 2048 blocks: 120 us of
// dependent-load latency per call, as much as the pass that produced the partial sums.)
constexpr int kFinishSlices = 16;
__global__ __launch_bounds__(kFinishSlices * SN_WAVE) void colsum_finish_kernel(const float *partial, int blocks, int E, float *out)
{
    __shared__ float red[kFinishSlices][SN_WAVE];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * SN_WAVE + lane, which = blockIdx.y;
    float acc = 0.0f;
    if (c < E) {
        int b = wid;
        for (; b + 7 * kFinishSlices < blocks; b += 8 * kFinishSlices) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = partial[((int64_t)(b + u * kFinishSlices) * 2 + which) * E + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += t[u];
        }
        for (; b < blocks; b += kFinishSlices) acc += partial[((int64_t)b * 2 + which) * E + c];
    }
    red[wid][lane] = acc;
    __syncthreads();
    if (wid == 0 && c < E) {
        float t = red[0][lane];
#pragma unroll
        for (int w = 1; w < kFinishSlices; ++w) t += red[w][lane];
        out[(int64_t)which * E + c] = t;
    }
}

// ------------------------------------------------------------------------------------------
// Gradient of an embedding lookup whose index tensor does not change between iterations (the class graphs' words,
// schema_net.py:121-126 -> gnn.py:83: the same [K, n] ids in every step): grad[w] = sum of dy[pos] over the positions pos
// that hold word w, taken from a sort of the ids done ONCE (order / seg), in position order - the library sorts the 103 k ids
// again in every backward pass (radix sort, segment offsets, two gather kernels: 0.3 ms of a 7 ms iteration).
// One workgroup per word, the four waves take every fourth occurrence, lane = 4 features of a 256-feature slab.
__global__ __launch_bounds__(256) void embedding_grad_sorted_kernel(const float *dy, const int64_t *order, const int64_t *seg, int E, int padding_idx,
                                                                   float *grad)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ f32x4 part[4][64];
    const int w = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int64_t b = seg[w], e = w == padding_idx ? seg[w] : seg[w + 1];
    for (int f0 = 0; f0 < E; f0 += 256) {
        const int f = f0 + 4 * lane;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (f < E) {
            for (int64_t o = b + wid; o < e; o += 16) {
                f32x4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int64_t oo = o + 4 * u;
                    const int64_t pos = order[oo < e ? oo : o];
                    v[u] = *reinterpret_cast<const f32x4 *>(dy + pos * E + f);
                    if (oo >= e) v[u] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
                acc += (v[0] + v[1]) + (v[2] + v[3]);
            }
        }
        __syncthreads();
        part[wid][lane] = acc;
        __syncthreads();
        if (wid == 0 && f < E) *reinterpret_cast<f32x4 *>(grad + (int64_t)w * E + f) = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
    }
}

// The same gradient for an index tensor that is new in every iteration (the instance graphs' words: 64 x 196 ids): no sort at all.
// One workgroup per word scans the ids 256 at a time (they stay in L2: 100 KB), keeps the positions that hold its word - in
// position order, through a ballot prefix - and adds their rows of dy, a feature (or four, E <= 1024) per thread: a fixed order,
// bit-reproducible.  (The library's embedding backward: radix sort, segment offsets, sum-and-scatter - 0.17 ms of an iteration,
// most of it launch latency of six small kernels.)
constexpr int kScanMaxE = 1024;
__global__ __launch_bounds__(256) void embedding_grad_scan_kernel(const float *dy, const int64_t *ids, int64_t n_ids, int E, int padding_idx, float *grad)
{
    __shared__ int hits[256];
    __shared__ int wave_cnt[4];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    float acc[kScanMaxE / 256];
#pragma unroll
    for (int k = 0; k < kScanMaxE / 256; ++k) acc[k] = 0.0f;
    if (w != padding_idx) {
        for (int64_t c0 = 0; c0 < n_ids; c0 += 256) {
            const int64_t i = c0 + tid;
            const bool hit = i < n_ids && ids[i] == (int64_t)w;
            const unsigned long long m = __ballot(hit);
            if (lane == 0) wave_cnt[wid] = __popcll(m);
            __syncthreads();
            int base = 0, total = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) { if (v < wid) base += wave_cnt[v]; total += wave_cnt[v]; }
            if (hit) hits[base + __popcll(m & ((1ull << lane) - 1ull))] = tid;
            __syncthreads();
            for (int h = 0; h < total; ++h) {                      // (block-uniform trip count; usually 0)
                const float *row = dy + (c0 + hits[h]) * E;
#pragma unroll
                for (int k = 0; k < kScanMaxE / 256; ++k) {
                    const int f = tid + 256 * k;
                    if (f < E) acc[k] += row[f];
                }
            }
            __syncthreads();
        }
    }
#pragma unroll
    for (int k = 0; k < kScanMaxE / 256; ++k) {
        const int f = tid + 256 * k;
        if (f < E) grad[(int64_t)w * E + f] = acc[k];
    }
}

// ------------------------------------------------------------------------------------------
// Backward of the node-weighted mean pooling pooled[g] = sum_i nodes[g][i] feat[g][i] / div (reference gnn.py:96), one pass over
// feat: d_feat[g][i] = nodes[g][i] g[g] / div, d_nodes[g][i] = feat[g][i] . g[g] / div.  One wave per row, 16-byte accesses.
__global__ __launch_bounds__(256) void weighted_pool_backward_kernel(const float *feat, const float *nodes, const float *gout, int64_t rows, int n,
                                                                    int E, const int32_t *div_dev, float *d_feat, float *d_nodes)
{
    const int lane = threadIdx.x & 63;
    const float inv = 1.0f / (div_dev ? (float)(*div_dev) : (float)n);
    for (int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); row < rows; row += (int64_t)gridDim.x * 4) {
        const int g = (int)(row / n);
        const float w = nodes[row] * inv;
        const float *f = feat + row * E, *go = gout + (int64_t)g * E;
        float *o = d_feat + row * E;
        float acc = 0.0f;
        for (int c = lane * 4; c < E; c += SN_WAVE * 4) {
            const float4 fv = *reinterpret_cast<const float4 *>(f + c), gv = *reinterpret_cast<const float4 *>(go + c);
            acc += (fv.x * gv.x + fv.y * gv.y) + (fv.z * gv.z + fv.w * gv.w);
            *reinterpret_cast<float4 *>(o + c) = float4{w * gv.x, w * gv.y, w * gv.z, w * gv.w};
        }
        acc = sn_wave_sum(acc);
        if (lane == 0) d_nodes[row] = acc * inv;
    }
}

// out[i] = attr2[i][0] w[0] + attr2[i][1] w[1]: the reference's `attr2 @ w` behind its instance graphs (large_scale_feat_to_v.cpp /
// large_scale_feat_to_e.cpp:141-147, squeezed) under autograd - round 6: one launch forward and two backward where the two products, the
// add, the selects of the weights and their backward nodes were ~30 library launches of an iteration.  The products are rounded one by
// one and then added (-ffp-contract=off): the values of `a0 * w0 + a1 * w1` in torch.
__global__ __launch_bounds__(256) void weigh_attributes_kernel(const float *attr2, int64_t n, const float *w, float *out)
{
    const float w0 = w[0], w1 = w[1];
    const int64_t n2 = ((reinterpret_cast<uintptr_t>(attr2) & 15) == 0 && (reinterpret_cast<uintptr_t>(out) & 7) == 0) ? n / 2 : 0;
    const float4 *a4 = reinterpret_cast<const float4 *>(attr2);
    float2 *o2 = reinterpret_cast<float2 *>(out);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        const float4 a = a4[i];
        o2[i] = float2{a.x * w0 + a.y * w1, a.z * w0 + a.w * w1};
    }
    for (int64_t i = n2 * 2 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = attr2[2 * i] * w0 + attr2[2 * i + 1] * w1;
}

// dw[k] = sum_i g[i] attr2[i][k]: per-block partial sums (fp32 per thread over its strided elements, fp64 from there on), summed in
// block order by the finishing launch - deterministic
__global__ __launch_bounds__(256) void weigh_attributes_backward_kernel(const float *attr2, const float *g, int64_t n, double *partial)
{
    __shared__ double red[2][4];
    float s0 = 0.0f, s1 = 0.0f;
    const int64_t n2 = ((reinterpret_cast<uintptr_t>(attr2) & 15) == 0 && (reinterpret_cast<uintptr_t>(g) & 7) == 0) ? n / 2 : 0;
    const float4 *a4 = reinterpret_cast<const float4 *>(attr2);
    const float2 *g2 = reinterpret_cast<const float2 *>(g);
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        const float4 a = a4[i];
        const float2 gg = g2[i];
        s0 += gg.x * a.x; s1 += gg.x * a.y;
        s0 += gg.y * a.z; s1 += gg.y * a.w;
    }
    for (int64_t i = n2 * 2 + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        s0 += g[i] * attr2[2 * i];
        s1 += g[i] * attr2[2 * i + 1];
    }
    double d0 = sn_wave_sum_f64((double)s0), d1 = sn_wave_sum_f64((double)s1);
    if ((threadIdx.x & 63) == 0) { red[0][threadIdx.x >> 6] = d0; red[1][threadIdx.x >> 6] = d1; }
    __syncthreads();
    if (threadIdx.x < 2) partial[(int64_t)blockIdx.x * 2 + threadIdx.x] = (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

__global__ __launch_bounds__(64) void weigh_attributes_finish_kernel(const double *partial, int blocks, float *dw)
{
    const int lane = threadIdx.x, k = blockIdx.x;
    double t = 0.0;
    for (int b = lane; b < blocks; b += 64) t += partial[(int64_t)b * 2 + k];
    t = sn_wave_sum_f64(t);
    if (lane == 0) dw[k] = (float)t;
}

// r(x) = x if x > a else a - 1 + 1 / (1 + a - x) and its derivative (1, or 1 / (1 + a - x)^2), elementwise: the "rectified" sparsity terms
// of the reference's loss (schema_inference_loss.py:61-67) - a handful of scalars, which as a select built from torch ops was seven launches
// forward and five back per term (round 6: one each way; the same operations in the same order: same values).
__global__ __launch_bounds__(64) void rectify_linear_kernel(const float *x, int n, float a, float *out, float *deriv)
{
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const bool above = v > a;
    const float den = above ? 1.0f : (1.0f + a) - v;         // (the unselected branch's denominator replaced by 1, as in the torch form)
    const float q = 1.0f / den;
    out[i] = above ? v : (a - 1.0f) + q;
    deriv[i] = above ? 1.0f : q * q;
}

}  // namespace

extern "C" int sn_pow2_scale_blocks(int64_t n)
{
    const int64_t b = (n + (int64_t)kAmaxThreads * 16 - 1) / ((int64_t)kAmaxThreads * 16);
    return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

extern "C" int sn_pow2_scale(const float *x, int64_t n, float top, void *partial, float *scale, float *amax_out, void *stream)
{
    SN_REQUIRE(n > 0 && top > 0.0f, SN_ERR_BAD_ARG, "sn_pow2_scale: bad n=%lld top=%g", (long long)n, (double)top);
    SN_REQUIRE(x && partial && scale, SN_ERR_BAD_ARG, "sn_pow2_scale: NULL pointer");
    const int blocks = sn_pow2_scale_blocks(n);
    hipLaunchKernelGGL(amax_partial_kernel, dim3(blocks), dim3(kAmaxThreads), 0, (hipStream_t)stream, x, n, (unsigned *)partial);
    SN_CHECK_LAUNCH("sn_pow2_scale");
    hipLaunchKernelGGL(pow2_scale_finish_kernel, dim3(1), dim3(kAmaxThreads), 0, (hipStream_t)stream, (const unsigned *)partial, blocks, top, scale, amax_out);
    SN_CHECK_LAUNCH("sn_pow2_scale");
    return SN_OK;
}

// out[g][i][j] = (corner[g][a][b] + corner[g][b][a]) / 2 for i = perm[a], j = perm[b], a, b < n_kept[g]; 0 elsewhere.  Two launches: the
// corner symmetrised in place, tile pairs through LDS as sym_half_kernel does (tiles past the graph's extent skipped); then one
// workgroup per (64-row block of the STORED order, graph): the inverse permutation of the graph staged in LDS, a thread per column,
// rows in turn - the writes are whole rows of `out`, the reads a gather inside ONE 4 KiB row of the corner per output row.
constexpr int kScatterMaxN = 1024;
__global__ __launch_bounds__(256) void sym_corner_kernel(float *s, int n, int T, const int32_t *n_kept)
{
    int I = 0, rest = blockIdx.x;
    while (rest >= T - I) { rest -= T - I; ++I; }
    const int J = I + rest;
    const int nk = min(max(n_kept[blockIdx.y], 0), n);
    const int ri = I * kSymTile, cj = J * kSymTile;
    if (ri >= nk || cj >= nk) return;                            // (whole workgroup, before any barrier; J >= I)
    __shared__ float ta[kSymTile * kSymLd], tb[kSymTile * kSymLd];
    float *g = s + (int64_t)blockIdx.y * n * n;
    const int c = threadIdx.x & 63, r0 = threadIdx.x >> 6;
#pragma unroll 4
    for (int r = r0; r < kSymTile; r += 4) {
        const bool in_a = ri + r < nk && cj + c < nk, in_b = cj + r < nk && ri + c < nk;
        ta[r * kSymLd + c] = in_a ? g[(int64_t)(ri + r) * n + cj + c] : 0.0f;
        if (I != J) tb[r * kSymLd + c] = in_b ? g[(int64_t)(cj + r) * n + ri + c] : 0.0f;
    }
    __syncthreads();
    const float *tt = I != J ? tb : ta;
#pragma unroll 4
    for (int r = r0; r < kSymTile; r += 4) {
        if (ri + r < nk && cj + c < nk) g[(int64_t)(ri + r) * n + cj + c] = (ta[r * kSymLd + c] + tt[c * kSymLd + r]) * 0.5f;
        if (I != J && cj + r < nk && ri + c < nk) g[(int64_t)(cj + r) * n + ri + c] = (tb[r * kSymLd + c] + ta[c * kSymLd + r]) * 0.5f;
    }
}

__global__ __launch_bounds__(256) void scatter_corner_kernel(const float *corner, const int32_t *perm, const int32_t *n_kept, int n, float *out)
{
    __shared__ short inv[kScatterMaxN];
    const int g = blockIdx.y, nk = min(max(n_kept[g], 0), n);
    for (int a = threadIdx.x; a < n; a += 256) inv[perm[(int64_t)g * n + a]] = (short)(a < nk ? a : -1);
    __syncthreads();
    const float *c = corner + (int64_t)g * n * n;
    float *o = out + (int64_t)g * n * n;
    const int i0 = blockIdx.x * 64;
    for (int i = i0; i < min(n, i0 + 64); ++i) {
        const int a = inv[i];                                     // (block-uniform)
        for (int j = threadIdx.x; j < n; j += 256) {
            const int b = inv[j];
            o[(int64_t)i * n + j] = (a >= 0 && b >= 0) ? c[(int64_t)a * n + b] : 0.0f;
        }
    }
}

extern "C" int sn_sym_scatter_corner(float *corner, const int32_t *perm, const int32_t *n_kept, int G, int n, float *out, void *stream)
{
    SN_REQUIRE(G >= 0 && n >= 0, SN_ERR_BAD_ARG, "sn_sym_scatter_corner: bad G=%d n=%d", G, n);
    if (G == 0 || n == 0) return SN_OK;
    SN_REQUIRE(corner && perm && n_kept && out, SN_ERR_BAD_ARG, "sn_sym_scatter_corner: NULL pointer");
    SN_REQUIRE(n <= kScatterMaxN && G <= 65535, SN_ERR_UNSUPPORTED, "sn_sym_scatter_corner: n=%d > %d or G=%d > 65535", n, kScatterMaxN, G);
    const int T = (n + kSymTile - 1) / kSymTile;
    hipLaunchKernelGGL(sym_corner_kernel, dim3((unsigned)(T * (T + 1) / 2), (unsigned)G), dim3(256), 0, (hipStream_t)stream, corner, n, T, n_kept);
    hipLaunchKernelGGL(scatter_corner_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)G), dim3(256), 0, (hipStream_t)stream, corner, perm, n_kept, n, out);
    SN_CHECK_LAUNCH("sn_sym_scatter_corner");
    return SN_OK;
}

extern "C" int sn_sym_half_inplace(float *s, int G, int n, void *stream)
{
    SN_REQUIRE(G >= 0 && n >= 0, SN_ERR_BAD_ARG, "sn_sym_half_inplace: bad G=%d n=%d", G, n);
    if (G == 0 || n == 0) return SN_OK;
    SN_REQUIRE(s, SN_ERR_BAD_ARG, "sn_sym_half_inplace: NULL pointer");
    const int T = (n + kSymTile - 1) / kSymTile;
    SN_REQUIRE(G <= 65535 && T <= 2048, SN_ERR_UNSUPPORTED, "sn_sym_half_inplace: G=%d or n=%d too large", G, n);
    const unsigned pairs = (unsigned)(T * (T + 1) / 2);
    if (n % 4 == 0 && ((uintptr_t)s & 15) == 0) hipLaunchKernelGGL(sym_half_kernel<true>, dim3(pairs, G), dim3(256), 0, (hipStream_t)stream, s, n, T);
    else hipLaunchKernelGGL(sym_half_kernel<false>, dim3(pairs, G), dim3(256), 0, (hipStream_t)stream, s, n, T);
    SN_CHECK_LAUNCH("sn_sym_half_inplace");
    return SN_OK;
}

extern "C" int sn_normalize_sum_rows(float *x, int64_t rows, int n, float min_val, int diag_n, void *stream)
{
    SN_REQUIRE(rows >= 0 && n > 0 && diag_n >= 0 && diag_n <= n, SN_ERR_BAD_ARG, "sn_normalize_sum_rows: bad rows=%lld n=%d diag_n=%d",
               (long long)rows, n, diag_n);
    if (rows == 0) return SN_OK;
    SN_REQUIRE(x, SN_ERR_BAD_ARG, "sn_normalize_sum_rows: NULL pointer");
    const int64_t cap = (int64_t)sn_device_cus() * 32;
    hipLaunchKernelGGL(normalize_sum_rows_kernel, dim3((unsigned)(rows < cap ? rows : cap)), dim3(kNormThreads), 0, (hipStream_t)stream, x, rows, n,
                       min_val, diag_n);
    SN_CHECK_LAUNCH("sn_normalize_sum_rows");
    return SN_OK;
}

extern "C" int sn_ln_act_blocks(int64_t rows)
{
    const int64_t want = (rows + 3) / 4, cap = (int64_t)sn_device_cus() * 4;
    return (int)(want < 1 ? 1 : (want < cap ? want : cap));
}

extern "C" int sn_mask_layernorm_act_forward(const float *x, float *y, int G, int n, int E, const int32_t *n_valid, const float *gamma,
                                             const float *beta, float eps, int relu, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0, SN_ERR_BAD_ARG, "sn_mask_layernorm_act_forward: bad G=%d n=%d E=%d", G, n, E);
    if (G == 0) return SN_OK;
    SN_REQUIRE(x && y && gamma && beta, SN_ERR_BAD_ARG, "sn_mask_layernorm_act_forward: NULL pointer");
    SN_REQUIRE(E <= SN_WAVE * SN_LN_MAX, SN_ERR_UNSUPPORTED, "sn_mask_layernorm_act_forward: E=%d > %d", E, SN_WAVE * SN_LN_MAX);
    const int64_t rows = (int64_t)G * n;
    const int64_t want = (rows + 3) / 4, cap = (int64_t)sn_device_cus() * 32;
    hipLaunchKernelGGL(ln_act_forward_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, (hipStream_t)stream, x, y, rows, n, E, n_valid,
                       gamma, beta, eps, relu);
    SN_CHECK_LAUNCH("sn_mask_layernorm_act_forward");
    return SN_OK;
}

extern "C" int sn_mask_layernorm_act_backward(const float *x, const float *dy, int G, int n, int E, const int32_t *n_valid, const float *gamma,
                                              const float *beta, float eps, int relu, float *dx, float *partial, float *dgamma_dbeta,
                                              void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0, SN_ERR_BAD_ARG, "sn_mask_layernorm_act_backward: bad G=%d n=%d E=%d", G, n, E);
    SN_REQUIRE(dgamma_dbeta, SN_ERR_BAD_ARG, "sn_mask_layernorm_act_backward: NULL pointer");
    if (G == 0) return sn_zero_async(dgamma_dbeta, (size_t)2 * E * sizeof(float), (hipStream_t)stream);
    SN_REQUIRE(x && dy && gamma && beta && dx && partial, SN_ERR_BAD_ARG, "sn_mask_layernorm_act_backward: NULL pointer");
    SN_REQUIRE(E <= SN_WAVE * SN_LN_MAX, SN_ERR_UNSUPPORTED, "sn_mask_layernorm_act_backward: E=%d > %d", E, SN_WAVE * SN_LN_MAX);
    const int64_t rows = (int64_t)G * n;
    const int blocks = sn_ln_act_blocks(rows);
#define SN_LN_BWD(KM)                                                                                                                      \
    hipLaunchKernelGGL(ln_act_backward_kernel<KM>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, dy, rows, n, E, n_valid, gamma, \
                       beta, eps, relu, dx, partial)
    if (E <= 256) SN_LN_BWD(4);
    else if (E <= 512) SN_LN_BWD(8);
    else SN_LN_BWD(16);
#undef SN_LN_BWD
    SN_CHECK_LAUNCH("sn_mask_layernorm_act_backward");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3((unsigned)((E + SN_WAVE - 1) / SN_WAVE), 2), dim3(kFinishSlices * SN_WAVE), 0, (hipStream_t)stream, partial, blocks, E,
                       dgamma_dbeta);
    SN_CHECK_LAUNCH("sn_mask_layernorm_act_backward");
    return SN_OK;
}

extern "C" int sn_embedding_grad_sorted(const float *dy, const int64_t *order, const int64_t *seg, int rows, int E, int padding_idx, float *grad,
                                        void *stream)
{
    SN_REQUIRE(rows >= 0 && E > 0 && E % 4 == 0, SN_ERR_BAD_ARG, "sn_embedding_grad_sorted: bad rows=%d E=%d (E a multiple of 4)", rows, E);
    if (rows == 0) return SN_OK;
    SN_REQUIRE(dy && order && seg && grad, SN_ERR_BAD_ARG, "sn_embedding_grad_sorted: NULL pointer");
    SN_REQUIRE((((uintptr_t)dy | (uintptr_t)grad) & 15) == 0, SN_ERR_BAD_ARG, "sn_embedding_grad_sorted: dy / grad must be 16-byte aligned");
    hipLaunchKernelGGL(embedding_grad_sorted_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dy, order, seg, E, padding_idx, grad);
    SN_CHECK_LAUNCH("sn_embedding_grad_sorted");
    return SN_OK;
}

extern "C" int sn_embedding_grad_scan(const float *dy, const int64_t *ids, int64_t n_ids, int rows, int E, int padding_idx, float *grad, void *stream)
{
    SN_REQUIRE(rows >= 0 && E > 0 && n_ids >= 0, SN_ERR_BAD_ARG, "sn_embedding_grad_scan: bad rows=%d E=%d n_ids=%lld", rows, E, (long long)n_ids);
    if (rows == 0) return SN_OK;
    SN_REQUIRE(grad && (n_ids == 0 || (dy && ids)), SN_ERR_BAD_ARG, "sn_embedding_grad_scan: NULL pointer");
    SN_REQUIRE(E <= kScanMaxE, SN_ERR_UNSUPPORTED, "sn_embedding_grad_scan: E=%d > %d", E, kScanMaxE);
    hipLaunchKernelGGL(embedding_grad_scan_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, dy, ids, n_ids, E, padding_idx, grad);
    SN_CHECK_LAUNCH("sn_embedding_grad_scan");
    return SN_OK;
}

extern "C" int sn_weighted_pool_backward(const float *feat, const float *nodes, const float *grad_pooled, int G, int n, int E,
                                         const int32_t *divisor_dev, float *grad_feat, float *grad_nodes, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0 && E % 4 == 0, SN_ERR_BAD_ARG, "sn_weighted_pool_backward: bad G=%d n=%d E=%d (E a multiple of 4)", G, n, E);
    if (G == 0) return SN_OK;
    SN_REQUIRE(feat && nodes && grad_pooled && grad_feat && grad_nodes, SN_ERR_BAD_ARG, "sn_weighted_pool_backward: NULL pointer");
    SN_REQUIRE((((uintptr_t)feat | (uintptr_t)grad_pooled | (uintptr_t)grad_feat) & 15) == 0, SN_ERR_BAD_ARG, "sn_weighted_pool_backward: 16-byte alignment");
    const int64_t rows = (int64_t)G * n, want = (rows + 3) / 4, cap = (int64_t)sn_device_cus() * 32;
    hipLaunchKernelGGL(weighted_pool_backward_kernel, dim3((unsigned)(want < cap ? want : cap)), dim3(256), 0, (hipStream_t)stream, feat, nodes,
                       grad_pooled, rows, n, E, divisor_dev, grad_feat, grad_nodes);
    SN_CHECK_LAUNCH("sn_weighted_pool_backward");
    return SN_OK;
}

extern "C" int sn_weigh_blocks(int64_t n)
{
    const int64_t b = (n + 256 * 16 - 1) / (256 * 16);
    return (int)(b < 1 ? 1 : (b > 1024 ? 1024 : b));
}

extern "C" int sn_weigh_attributes(const float *attr2, int64_t n, const float *w, float *out, void *stream)
{
    SN_REQUIRE(n >= 0, SN_ERR_BAD_ARG, "sn_weigh_attributes: bad n=%lld", (long long)n);
    if (n == 0) return SN_OK;
    SN_REQUIRE(attr2 && w && out, SN_ERR_BAD_ARG, "sn_weigh_attributes: NULL pointer");
    hipLaunchKernelGGL(weigh_attributes_kernel, dim3((unsigned)sn_weigh_blocks(n)), dim3(256), 0, (hipStream_t)stream, attr2, n, w, out);
    SN_CHECK_LAUNCH("sn_weigh_attributes");
    return SN_OK;
}

extern "C" int sn_weigh_attributes_backward(const float *attr2, const float *g, int64_t n, void *partial, float *dw, void *stream)
{
    SN_REQUIRE(n > 0, SN_ERR_BAD_ARG, "sn_weigh_attributes_backward: bad n=%lld", (long long)n);
    SN_REQUIRE(attr2 && g && partial && dw, SN_ERR_BAD_ARG, "sn_weigh_attributes_backward: NULL pointer");
    SN_REQUIRE(((uintptr_t)partial & 7) == 0, SN_ERR_BAD_ARG, "sn_weigh_attributes_backward: partial must be 8-byte aligned");
    const int blocks = sn_weigh_blocks(n);
    hipLaunchKernelGGL(weigh_attributes_backward_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, attr2, g, n, (double *)partial);
    SN_CHECK_LAUNCH("sn_weigh_attributes_backward");
    hipLaunchKernelGGL(weigh_attributes_finish_kernel, dim3(2), dim3(64), 0, (hipStream_t)stream, (const double *)partial, blocks, dw);
    SN_CHECK_LAUNCH("sn_weigh_attributes_backward");
    return SN_OK;
}

extern "C" int sn_rectify_linear(const float *x, int n, float a, float *out, float *deriv, void *stream)
{
    SN_REQUIRE(n >= 0, SN_ERR_BAD_ARG, "sn_rectify_linear: bad n=%d", n);
    if (n == 0) return SN_OK;
    SN_REQUIRE(x && out && deriv, SN_ERR_BAD_ARG, "sn_rectify_linear: NULL pointer");
    hipLaunchKernelGGL(rectify_linear_kernel, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, (hipStream_t)stream, x, n, a, out, deriv);
    SN_CHECK_LAUNCH("sn_rectify_linear");
    return SN_OK;
}
