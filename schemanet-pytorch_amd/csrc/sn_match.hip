// Graph-matching epilogues around the GCN GEMMs (which run on rocBLAS through torch):
// adjacency symmetrisation, masked LayerNorm + ReLU, node-weighted pooling and the
// instance-vs-atlas similarity scores.  All HBM-bound elementwise / row-reduction work with
// wavefront shuffle reductions; no MFMA.
//
// Reference being replaced:
//   schema_inference/graph/gnn.py:27-30    adj = (E + E^T) / 2 + I
//   schema_inference/graph/gnn.py:41-46    masked_fill_(feat_mask) -> LayerNorm -> activation
//   schema_inference/graph/gnn.py:93-96    feat * nodes[..., None]; mean(dim=1)  (padded length)
//   schema_inference/graph/match.py:21-31  cosine / euclidean / inner_product similarity
#include "sn_common.h"

namespace {

// ------------------------------------------------------------------ adjacency
__global__ __launch_bounds__(256) void gcn_adjacency_kernel(const float *edges, int n, float *adj)
{
    __shared__ float tile[32][33];
    const int g = blockIdx.z, bi = blockIdx.y * 32, bj = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    const float *e = edges + (int64_t)g * n * n;
    // tile of E^T: read block (bj.., bi..) row-major
    for (int r = ty; r < 32; r += 8) {
        const int i = bj + r, j = bi + tx;
        tile[r][tx] = (i < n && j < n) ? e[(int64_t)i * n + j] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int i = bi + r, j = bj + tx;
        if (i < n && j < n) {
            const float s = e[(int64_t)i * n + j] + tile[tx][r];
            float v = s / 2.0f;
            if (i == j) v = v + 1.0f;
            adj[(int64_t)g * n * n + (int64_t)i * n + j] = v;
        }
    }
}

// ------------------------------------------------------------------ mask + LayerNorm + ReLU
// one wave per row; E <= 64 * 16
constexpr int kLnMax = SN_LN_MAX;

__global__ __launch_bounds__(256) void mask_layernorm_act_kernel(float *x, int64_t rows, int n, int E,
                                                                 const int32_t *n_valid, const float *gamma,
                                                                 const float *beta, float eps, int relu)
{
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int g = (int)(row / n), r = (int)(row % n);
    const bool masked = n_valid && r >= n_valid[g];
    float *p = x + row * E;
    float v[kLnMax];
#pragma unroll
    for (int k = 0; k < kLnMax; ++k) {
        const int c = lane + SN_WAVE * k;
        v[k] = (c < E && !masked) ? p[c] : 0.0f;
    }
    float gm[kLnMax], bt[kLnMax];
    sn_layernorm_coeffs(gm, bt, lane, E, gamma, beta);
    sn_layernorm_row(v, lane, E, gm, bt, eps, relu);
#pragma unroll
    for (int k = 0; k < kLnMax; ++k) {
        const int c = lane + SN_WAVE * k;
        if (c < E) p[c] = v[k];
    }
}

// ------------------------------------------------------------------ weighted pooling
// block = 64 columns x 4 row slices (one wave each); partial sums are combined in wave order,
// so the result is deterministic
__global__ __launch_bounds__(256) void weighted_pool_kernel(const float *feat, const float *nodes, int n, int E,
                                                            const int32_t *divisor_dev, float *out)
{
    __shared__ float part[4][SN_WAVE];
    const int g = blockIdx.y, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int c = blockIdx.x * SN_WAVE + lane;
    const float *w = nodes + (int64_t)g * n;
    float acc = 0.0f;
    if (c < E) {
        const float *f = feat + (int64_t)g * n * E + c;
        // (eight rows requested at a time, added in the order of the plain loop: the same sums bit for bit, 8 x the bytes in flight)
        int r = wid;
        for (; r + 28 < n; r += 32) {
            float v[8], u[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) { v[t] = f[(int64_t)(r + 4 * t) * E]; u[t] = w[r + 4 * t]; }
#pragma unroll
            for (int t = 0; t < 8; ++t) acc = acc + v[t] * u[t];
        }
        for (; r < n; r += 4) acc = acc + f[(int64_t)r * E] * w[r];
    }
    part[wid][lane] = acc;
    __syncthreads();
    if (wid == 0 && c < E) {
        const float total = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
        const float div = divisor_dev ? (float)(*divisor_dev) : (float)n;
        out[(int64_t)g * E + c] = total / div;
    }
}

// mask + LayerNorm + activation of x [G, n, E] (as mask_layernorm_act_kernel) and the weighted pooling of the result (as
// weighted_pool_kernel) in one pass: the normalised rows are never stored.  One workgroup per graph; wave w normalises
// the rows r = w, w + 4, ... and adds them to its partial sums in that order, the four partial sums are combined as
// (0 + 1) + (2 + 3): the sums of weighted_pool_kernel, bit for bit.  The next row's loads are issued before the current
// row's reductions.
__global__ __launch_bounds__(256) void layernorm_weighted_pool_kernel(const float *x, const float *nodes, int n, int E,
                                                                      const int32_t *n_valid, const float *gamma, const float *beta,
                                                                      float eps, int relu, const int32_t *divisor_dev, float *out)
{
    extern __shared__ float part[];                      // [4][E]
    const int g = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int nv = n_valid ? n_valid[g] : n;
    const float *w = nodes + (int64_t)g * n;
    const float *xg = x + (int64_t)g * n * E;
    float acc[kLnMax], nxt[kLnMax], gm[kLnMax], bt[kLnMax];
    sn_layernorm_coeffs(gm, bt, lane, E, gamma, beta);
#pragma unroll
    for (int k = 0; k < kLnMax; ++k) acc[k] = 0.0f;
    auto load = [&](int r, float (&v)[kLnMax]) {
        const bool live = r < n && r < nv;
        const float *p = xg + (int64_t)r * E;
#pragma unroll
        for (int k = 0; k < kLnMax; ++k) {
            const int c = lane + SN_WAVE * k;
            v[k] = (c < E && live) ? p[c] : 0.0f;
        }
    };
    load(wid, nxt);
    for (int r = wid; r < n; r += 4) {
        float v[kLnMax];
#pragma unroll
        for (int k = 0; k < kLnMax; ++k) v[k] = nxt[k];
        load(r + 4, nxt);
        sn_layernorm_row(v, lane, E, gm, bt, eps, relu);
        const float wr = w[r];
#pragma unroll
        for (int k = 0; k < kLnMax; ++k) acc[k] = acc[k] + v[k] * wr;
    }
#pragma unroll
    for (int k = 0; k < kLnMax; ++k) {
        const int c = lane + SN_WAVE * k;
        if (c < E) part[wid * E + c] = acc[k];
    }
    __syncthreads();
    const float div = divisor_dev ? (float)(*divisor_dev) : (float)n;
    for (int c = threadIdx.x; c < E; c += 256) {
        const float total = (part[c] + part[E + c]) + (part[2 * E + c] + part[3 * E + c]);
        out[(int64_t)g * E + c] = total / div;
    }
}

// ------------------------------------------------------------------ similarity scores
// block = 4 waves = 16 classes of one image; a wave owns 4 classes whose three reductions run
// interleaved (one class at a time is a chain of dependent shuffles: latency-bound).
constexpr int kScoreCols = 16;
__global__ __launch_bounds__(256) void match_scores_kernel(const float *fi, const float *fk, int K, int E,
                                                           int similarity, float *pred)
{
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int k0 = blockIdx.y * kScoreCols + wid * 4;
    if (k0 >= K) return;
    const float *a = fi + (int64_t)b * E;
    float dot[4] = {0.0f, 0.0f, 0.0f, 0.0f}, nb[4] = {0.0f, 0.0f, 0.0f, 0.0f}, d2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float na = 0.0f;
    for (int c = lane; c < E; c += SN_WAVE) {
        const float x = a[c];
        na += x * x;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = k0 + j < K ? k0 + j : K - 1;
            const float y = fk[(int64_t)k * E + c];
            dot[j] += x * y;
            nb[j] += y * y;
            d2[j] += (x - y) * (x - y);
        }
    }
    if (similarity == 1) na = sqrtf(sn_wave_sum(na));
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r;
        if (similarity == 0) {
            r = sn_wave_sum(dot[j]);
        } else if (similarity == 1) {          // (cosine_similarity + 1) / 2, eps = 1e-8
            const float den = fmaxf(na, 1.0e-8f) * fmaxf(sqrtf(sn_wave_sum(nb[j])), 1.0e-8f);
            r = (sn_wave_sum(dot[j]) / den + 1.0f) / 2.0f;
        } else {                               // 1 / (1 + |a - b|_2)
            r = 1.0f / (1.0f + sqrtf(sn_wave_sum(d2[j])));
        }
        if (lane == 0 && k0 + j < K) pred[(int64_t)b * K + k0 + j] = r;
    }
}

// ------------------------------------------------------------------ pooled / divisor -> fc
// out[g][o] = bias[o] + sum_e (pooled[g][e] / div) * W[o][e]     (reference gnn.py:96-98)
__global__ __launch_bounds__(256) void pool_fc_kernel(const float *pooled, int parts, const int32_t *div_dev, float div_host, const float *W,
                                                      const float *bias, int E, int E_out, float *out)
{
    const int g = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int o0 = blockIdx.y * kScoreCols + wid * 4;
    if (o0 >= E_out) return;
    const float rdiv = 1.0f / (div_dev ? (float)div_dev[0] : div_host);     // one reciprocal (the kernel is instruction-bound: 1 ulp vs a division per element)
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int c = lane; c < E; c += SN_WAVE) {
        float ps = pooled[((int64_t)g * parts) * E + c];
        for (int t = 1; t < parts; ++t) ps += pooled[((int64_t)g * parts + t) * E + c];     // fixed order
        const float x = ps * rdiv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int o = o0 + j < E_out ? o0 + j : E_out - 1;
            acc[j] = fmaf(x, W[(int64_t)o * E + c], acc[j]);
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float r = sn_wave_sum(acc[j]);
        if (lane == 0 && o0 + j < E_out) out[(int64_t)g * E_out + o0 + j] = r + (bias ? bias[o0 + j] : 0.0f);
    }
}

// The same with the weight given TRANSPOSED (Wt[e][o], a weight-only operand the caller keeps): thread = output, so a wave
// reads whole 256-byte lines of Wt and needs no cross-lane reduction; a workgroup takes kFcGraphs graphs x 64 outputs, its four
// waves a quarter of E each (partials added in wave order: bit-reproducible).  pool_fc_kernel above reads W rows per
// wave and reduces every output over the lanes: 4 096 workgroups and 15 us at G = 256, E = 256, against ~3 us here.
constexpr int kFcGraphs = 4, kFcMaxE = 2048;
__global__ __launch_bounds__(256) void pool_fc_t_kernel(const float *pooled, int parts, int G, const int32_t *div_dev, float div_host,
                                                        const float *Wt, const float *bias, int E, int E_out, float *out)
{
    __shared__ __attribute__((aligned(16))) float xs[kFcGraphs][kFcMaxE + 16];
    __shared__ float part[4][kFcGraphs][64];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int g0 = blockIdx.x * kFcGraphs, o = blockIdx.y * 64 + lane;
    const int chunk = (((E + 3) / 4) + 3) & ~3;                  // a wave's share of E (multiple of 4: float4 reads of xs)
    const int e0 = wid * chunk, e1 = min(E, e0 + chunk);
    const float *w = Wt + (o < E_out ? o : E_out - 1);
    // the kernel lives on memory latency: the first kFcAhead weights of the wave's share are requested before anything
    // else (at E = 256 that is all of them: one round trip, overlapped with the staging of x)
    constexpr int kFcAhead = 64;
    float wreg[kFcAhead];
#pragma unroll
    for (int q = 0; q < kFcAhead; ++q) {                         // (clamped index + select: no branch between the loads)
        const float v = w[(int64_t)min(e0 + q, E - 1) * E_out];
        wreg[q] = e0 + q < e1 ? v : 0.0f;
    }
    const float rdiv = 1.0f / (div_dev ? (float)div_dev[0] : div_host);
    const int Ep = 4 * chunk;                                    // <= E + 15: the tail is zero
    for (int idx = tid; idx < kFcGraphs * Ep; idx += 256) {
        const int gg = idx / Ep, c = idx - gg * Ep, g = g0 + gg;
        float ps = 0.0f;
        if (g < G && c < E) {
            ps = pooled[((int64_t)g * parts) * E + c];
            for (int t = 1; t < parts; ++t) ps += pooled[((int64_t)g * parts + t) * E + c];     // fixed order
        }
        xs[gg][c] = ps * rdiv;
    }
    __syncthreads();
    float acc[kFcGraphs];
#pragma unroll
    for (int gg = 0; gg < kFcGraphs; ++gg) acc[gg] = 0.0f;
#pragma unroll
    for (int q = 0; q < kFcAhead; q += 4) {
        if (q < chunk)                                           // (wave-uniform)
#pragma unroll
        for (int gg = 0; gg < kFcGraphs; ++gg) {
            const float4 x = *reinterpret_cast<const float4 *>(&xs[gg][e0 + q]);     // same address for every lane: broadcast
            acc[gg] = fmaf(x.x, wreg[q], acc[gg]);
            acc[gg] = fmaf(x.y, wreg[q + 1], acc[gg]);
            acc[gg] = fmaf(x.z, wreg[q + 2], acc[gg]);
            acc[gg] = fmaf(x.w, wreg[q + 3], acc[gg]);
        }
    }
    for (int e = e0 + kFcAhead; e < e0 + chunk; e += 16) {       // wider layers: 16 weights in flight per pass
        float wv[16];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const float v = w[(int64_t)min(e + q, E - 1) * E_out];
            wv[q] = e + q < e1 ? v : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < 16; q += 4) {
            if (e + q < e0 + chunk)
#pragma unroll
            for (int gg = 0; gg < kFcGraphs; ++gg) {
                const float4 x = *reinterpret_cast<const float4 *>(&xs[gg][e + q]);
                acc[gg] = fmaf(x.x, wv[q], acc[gg]);
                acc[gg] = fmaf(x.y, wv[q + 1], acc[gg]);
                acc[gg] = fmaf(x.z, wv[q + 2], acc[gg]);
                acc[gg] = fmaf(x.w, wv[q + 3], acc[gg]);
            }
        }
    }
#pragma unroll
    for (int gg = 0; gg < kFcGraphs; ++gg) part[wid][gg][lane] = acc[gg];
    __syncthreads();
    {
        const int gg = wid, g = g0 + gg;                                              // wave gg finishes graph gg
        if (g < G && o < E_out)
            out[(int64_t)g * E_out + o] = ((part[0][gg][lane] + part[1][gg][lane]) + (part[2][gg][lane] + part[3][gg][lane])) + (bias ? bias[o] : 0.0f);
    }
}

// ------------------------------------------------------------------ per-class votes
// one wave per image: argmax over the K scores (first index on ties, like torch.argmax) with
// shuffle reductions, then one atomic add into the class's counter; votes[K] counts images.
__global__ __launch_bounds__(256) void class_votes_kernel(const float *pred, int B, int K, float *votes)
{
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += SN_WAVE) {
        const float v = pred[(int64_t)b * K + k];
        if (v > best || (v != v && best == best)) { best = v; bi = k; }     // NaN wins, like torch
    }
    if (bi == 0x7fffffff && lane < K) bi = lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, SN_WAVE);
        const int oi = __shfl_xor(bi, off, SN_WAVE);
        const bool take = (ov != ov && best == best) || (ov > best && best == best) || (ov == best && oi < bi) ||
                          (ov != ov && best != best && oi < bi);
        if (take) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        atomicAdd(&votes[bi < K ? bi : 0], 1.0f);
        atomicAdd(&votes[K], 1.0f);
    }
}


// Scores AND votes of an image in one launch (an evaluation loop: the scores' only reader besides the caller is the vote
// counter, and a kernel boundary costs more than both kernels' work).  One workgroup of sixteen waves per image: a wave
// owns four classes at a time exactly as in match_scores_kernel (same lane partials, same reductions: same scores bit for
// bit), the row of K scores stays in LDS, wave 0 takes its argmax exactly as class_votes_kernel.
constexpr int kVotesMaxK = 4096;
__global__ __launch_bounds__(1024) void match_scores_votes_kernel(const float *fi, const float *fk, int K, int E, int similarity,
                                                                  float *pred, float *votes)
{
    __shared__ float row[kVotesMaxK];
    const int b = blockIdx.x, lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const float *a = fi + (int64_t)b * E;
    for (int k0 = wid * 4; k0 < K; k0 += nw * 4) {
        float dot[4] = {0.0f, 0.0f, 0.0f, 0.0f}, nb[4] = {0.0f, 0.0f, 0.0f, 0.0f}, d2[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        float na = 0.0f;
        for (int c = lane; c < E; c += SN_WAVE) {
            const float x = a[c];
            na += x * x;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = k0 + j < K ? k0 + j : K - 1;
                const float y = fk[(int64_t)k * E + c];
                dot[j] += x * y;
                nb[j] += y * y;
                d2[j] += (x - y) * (x - y);
            }
        }
        if (similarity == 1) na = sqrtf(sn_wave_sum(na));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float r;
            if (similarity == 0) {
                r = sn_wave_sum(dot[j]);
            } else if (similarity == 1) {
                const float den = fmaxf(na, 1.0e-8f) * fmaxf(sqrtf(sn_wave_sum(nb[j])), 1.0e-8f);
                r = (sn_wave_sum(dot[j]) / den + 1.0f) / 2.0f;
            } else {
                r = 1.0f / (1.0f + sqrtf(sn_wave_sum(d2[j])));
            }
            if (lane == 0 && k0 + j < K) { pred[(int64_t)b * K + k0 + j] = r; row[k0 + j] = r; }
        }
    }
    __syncthreads();
    if (wid != 0) return;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int k = lane; k < K; k += SN_WAVE) {
        const float v = row[k];
        if (v > best || (v != v && best == best)) { best = v; bi = k; }     // NaN wins, like torch
    }
    if (bi == 0x7fffffff && lane < K) bi = lane;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off, SN_WAVE);
        const int oi = __shfl_xor(bi, off, SN_WAVE);
        const bool take = (ov != ov && best == best) || (ov > best && best == best) || (ov == best && oi < bi) ||
                          (ov != ov && best != best && oi < bi);
        if (take) { best = ov; bi = oi; }
    }
    if (lane == 0) {
        atomicAdd(&votes[bi < K ? bi : 0], 1.0f);
        atomicAdd(&votes[K], 1.0f);
    }
}

}  // namespace

extern "C" int sn_class_votes(const float *pred, int B, int K, float *votes, void *stream)
{
    SN_REQUIRE(B >= 0 && K > 0, SN_ERR_BAD_ARG, "sn_class_votes: bad B=%d K=%d", B, K);
    if (B == 0) return SN_OK;
    SN_REQUIRE(pred && votes, SN_ERR_BAD_ARG, "sn_class_votes: NULL pointer");
    hipLaunchKernelGGL(class_votes_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, (hipStream_t)stream, pred, B, K, votes);
    SN_CHECK_LAUNCH("sn_class_votes");
    return SN_OK;
}

extern "C" int sn_gcn_adjacency(const float *edges, int G, int n, float *adj, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_gcn_adjacency: bad G=%d n=%d", G, n);
    if (G == 0) return SN_OK;
    SN_REQUIRE(edges && adj && edges != adj, SN_ERR_BAD_ARG, "sn_gcn_adjacency: NULL or aliased pointers");
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_adjacency: G=%d > 65535", G);
    const unsigned t = (unsigned)((n + 31) / 32);
    hipLaunchKernelGGL(gcn_adjacency_kernel, dim3(t, t, (unsigned)G), dim3(256), 0, (hipStream_t)stream, edges, n, adj);
    SN_CHECK_LAUNCH("sn_gcn_adjacency");
    return SN_OK;
}

extern "C" int sn_mask_layernorm_act(float *x, int G, int n, int E, const int32_t *n_valid,
                                     const float *gamma, const float *beta, float eps, int relu, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0, SN_ERR_BAD_ARG, "sn_mask_layernorm_act: bad G=%d n=%d E=%d", G, n, E);
    if (G == 0) return SN_OK;
    SN_REQUIRE(x && gamma && beta, SN_ERR_BAD_ARG, "sn_mask_layernorm_act: NULL pointer");
    SN_REQUIRE(E <= SN_WAVE * kLnMax, SN_ERR_UNSUPPORTED, "sn_mask_layernorm_act: E=%d > %d", E, SN_WAVE * kLnMax);
    const int64_t rows = (int64_t)G * n;
    const int64_t blocks = (rows + 3) / 4;
    SN_REQUIRE(blocks <= 0x7fffffff, SN_ERR_UNSUPPORTED, "sn_mask_layernorm_act: too many rows");
    hipLaunchKernelGGL(mask_layernorm_act_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, rows,
                       n, E, n_valid, gamma, beta, eps, relu);
    SN_CHECK_LAUNCH("sn_mask_layernorm_act");
    return SN_OK;
}

extern "C" int sn_weighted_pool(const float *feat, const float *nodes, int G, int n, int E,
                                const int32_t *divisor_dev, float *out, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0, SN_ERR_BAD_ARG, "sn_weighted_pool: bad G=%d n=%d E=%d", G, n, E);
    if (G == 0) return SN_OK;
    SN_REQUIRE(feat && nodes && out, SN_ERR_BAD_ARG, "sn_weighted_pool: NULL pointer");
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_weighted_pool: G=%d > 65535", G);
    hipLaunchKernelGGL(weighted_pool_kernel, dim3((unsigned)((E + SN_WAVE - 1) / SN_WAVE), (unsigned)G), dim3(256), 0,
                       (hipStream_t)stream, feat, nodes, n, E, divisor_dev, out);
    SN_CHECK_LAUNCH("sn_weighted_pool");
    return SN_OK;
}

extern "C" int sn_layernorm_weighted_pool(const float *x, const float *nodes, int G, int n, int E, const int32_t *n_valid,
                                          const float *gamma, const float *beta, float eps, int relu,
                                          const int32_t *divisor_dev, float *out, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0, SN_ERR_BAD_ARG, "sn_layernorm_weighted_pool: bad G=%d n=%d E=%d", G, n, E);
    if (G == 0) return SN_OK;
    SN_REQUIRE(x && nodes && gamma && beta && out, SN_ERR_BAD_ARG, "sn_layernorm_weighted_pool: NULL pointer");
    SN_REQUIRE(E <= SN_WAVE * kLnMax, SN_ERR_UNSUPPORTED, "sn_layernorm_weighted_pool: E=%d > %d", E, SN_WAVE * kLnMax);
    hipLaunchKernelGGL(layernorm_weighted_pool_kernel, dim3((unsigned)G), dim3(256), (size_t)4 * E * sizeof(float), (hipStream_t)stream,
                       x, nodes, n, E, n_valid, gamma, beta, eps, relu, divisor_dev, out);
    SN_CHECK_LAUNCH("sn_layernorm_weighted_pool");
    return SN_OK;
}

extern "C" int sn_match_scores(const float *feat_inst, const float *feat_kg, int B, int K, int E,
                               int similarity, float *pred, void *stream)
{
    SN_REQUIRE(B >= 0 && K > 0 && E > 0, SN_ERR_BAD_ARG, "sn_match_scores: bad B=%d K=%d E=%d", B, K, E);
    if (B == 0) return SN_OK;
    SN_REQUIRE(feat_inst && feat_kg && pred, SN_ERR_BAD_ARG, "sn_match_scores: NULL pointer");
    SN_REQUIRE(similarity >= 0 && similarity <= 2, SN_ERR_BAD_ARG, "sn_match_scores: similarity=%d", similarity);
    SN_REQUIRE(B <= 0x7fffffff / 1 && (K + kScoreCols - 1) / kScoreCols <= 65535, SN_ERR_UNSUPPORTED, "sn_match_scores: K=%d too large", K);
    hipLaunchKernelGGL(match_scores_kernel, dim3((unsigned)B, (unsigned)((K + kScoreCols - 1) / kScoreCols)), dim3(256), 0,
                       (hipStream_t)stream, feat_inst, feat_kg, K, E, similarity, pred);
    SN_CHECK_LAUNCH("sn_match_scores");
    return SN_OK;
}

extern "C" int sn_match_scores_votes(const float *feat_inst, const float *feat_kg, int B, int K, int E, int similarity,
                                     float *pred, float *votes, void *stream)
{
    SN_REQUIRE(B >= 0 && K > 0 && E > 0, SN_ERR_BAD_ARG, "sn_match_scores_votes: bad B=%d K=%d E=%d", B, K, E);
    if (B == 0) return SN_OK;
    SN_REQUIRE(feat_inst && feat_kg && pred && votes, SN_ERR_BAD_ARG, "sn_match_scores_votes: NULL pointer");
    SN_REQUIRE(similarity >= 0 && similarity <= 2, SN_ERR_BAD_ARG, "sn_match_scores_votes: similarity=%d", similarity);
    if (K > kVotesMaxK) {                      // the row of scores does not fit the workgroup's LDS: the two kernels
        if (int rc = sn_match_scores(feat_inst, feat_kg, B, K, E, similarity, pred, stream)) return rc;
        return sn_class_votes(pred, B, K, votes, stream);
    }
    const int waves = K >= 64 ? 16 : (K + 3) / 4;
    hipLaunchKernelGGL(match_scores_votes_kernel, dim3((unsigned)B), dim3(64u * (unsigned)waves), 0, (hipStream_t)stream, feat_inst, feat_kg,
                       K, E, similarity, pred, votes);
    SN_CHECK_LAUNCH("sn_match_scores_votes");
    return SN_OK;
}

extern "C" int sn_pool_fc_t(const float *pooled_sum, int G, int parts, int E, const int32_t *divisor_dev, float divisor_host,
                            const float *weight_t, const float *bias, int E_out, float *out, void *stream)
{
    SN_REQUIRE(G >= 0 && parts > 0 && E > 0 && E_out > 0, SN_ERR_BAD_ARG, "sn_pool_fc_t: bad G=%d parts=%d E=%d E_out=%d", G, parts, E, E_out);
    if (G == 0) return SN_OK;
    SN_REQUIRE(pooled_sum && weight_t && out, SN_ERR_BAD_ARG, "sn_pool_fc_t: NULL pointer");
    SN_REQUIRE(divisor_dev || divisor_host != 0.0f, SN_ERR_BAD_ARG, "sn_pool_fc_t: no divisor");
    SN_REQUIRE(E <= kFcMaxE, SN_ERR_UNSUPPORTED, "sn_pool_fc_t: E=%d > %d (use sn_pool_fc)", E, kFcMaxE);
    SN_REQUIRE((E_out + 63) / 64 <= 65535, SN_ERR_UNSUPPORTED, "sn_pool_fc_t: E_out=%d too large", E_out);
    static_assert(kFcGraphs == 4, "one finishing wave per graph of the group");
    hipLaunchKernelGGL(pool_fc_t_kernel, dim3((unsigned)((G + kFcGraphs - 1) / kFcGraphs), (unsigned)((E_out + 63) / 64)), dim3(256), 0,
                       (hipStream_t)stream, pooled_sum, parts, G, divisor_dev, divisor_host, weight_t, bias, E, E_out, out);
    SN_CHECK_LAUNCH("sn_pool_fc_t");
    return SN_OK;
}

extern "C" int sn_pool_fc(const float *pooled_sum, int G, int parts, int E, const int32_t *divisor_dev, float divisor_host,
                          const float *weight, const float *bias, int E_out, float *out, void *stream)
{
    SN_REQUIRE(G >= 0 && parts > 0 && E > 0 && E_out > 0, SN_ERR_BAD_ARG, "sn_pool_fc: bad G=%d parts=%d E=%d E_out=%d", G, parts, E, E_out);
    if (G == 0) return SN_OK;
    SN_REQUIRE(pooled_sum && weight && out, SN_ERR_BAD_ARG, "sn_pool_fc: NULL pointer");
    SN_REQUIRE(divisor_dev || divisor_host != 0.0f, SN_ERR_BAD_ARG, "sn_pool_fc: no divisor");
    SN_REQUIRE((E_out + kScoreCols - 1) / kScoreCols <= 65535, SN_ERR_UNSUPPORTED, "sn_pool_fc: E_out=%d too large", E_out);
    hipLaunchKernelGGL(pool_fc_kernel, dim3((unsigned)G, (unsigned)((E_out + kScoreCols - 1) / kScoreCols)), dim3(256), 0,
                       (hipStream_t)stream, pooled_sum, parts, divisor_dev, divisor_host, weight, bias, E, E_out, out);
    SN_CHECK_LAUNCH("sn_pool_fc");
    return SN_OK;
}
