// Codebook extraction: the M-step and the distortion of Lloyd's k-means on the GPU
// (replaces scipy.cluster.vq.kmeans in the reference's scripts/extract_ingredients.py:28-56,117-124;
// the E-step is sn_assign_words).
//
// scipy's float32 path, restated (scipy/cluster/vq.py::_kmeans, _vq.update_cluster_means):
//   codes   = nearest centre of every observation                       -> sn_assign_words (exact, first index on ties)
//   avg     = mean_t |x_t - c_code(t)|_2                                  -> sn_kmeans_distances (fp64 per token)
//   centre' = (fp32 sum of the members IN OBSERVATION ORDER) / count      -> sn_kmeans_update (bit-identical sums)
//   clusters without members are dropped; stop when |avg_prev - avg| <= thresh.
// The member sums keep scipy's summation order (one thread per feature walks the members in token
// order), so on one GPU the centres are bit-identical to scipy's; with the tokens sharded over ranks the
// per-rank sums are added by one all-reduce (fp32 reassociation, ~1e-7 relative).
#include "sn_common.h"

namespace {

constexpr int kUpdThreads = 256;

// row of flat token t; `flat`: the grid is one contiguous run of rows (outer stride = n_inner x inner stride),
// so no 64-bit division per row (a vector 64-bit divide is ~150 instructions: with it the M-step was 30x slower)
__device__ __forceinline__ const float *km_row(const float *x, int64_t t, bool flat, int64_t n_inner, int64_t xso, int64_t xsi)
{
    return flat ? x + t * xsi : x + (t / n_inner) * xso + (t % n_inner) * xsi;
}

constexpr int kWaveSpan = 4096;                 // tokens one wave scans per round
constexpr int kChunk = 4 * kWaveSpan;           // tokens one workgroup scans per round

// One workgroup per centre.  Round: wave v scans tokens [t0 + 4096 v, t0 + 4096 (v + 1)) of the id stream and
// appends the centre's members to ITS list in LDS (ballot + prefix: token order is kept, no workgroup
// barrier inside the scan); then every thread owns features d = tid, tid + 256, ... and adds the members'
// values one after the other, lists in wave order = token order (8 rows in flight, adds in order).
__global__ __launch_bounds__(kUpdThreads) void kmeans_update_kernel(const float *x, int64_t n_tokens, int64_t n_inner, int64_t xso, int64_t xsi,
                                                                    const int64_t *ids, int64_t ids_so, int64_t ids_si, int D, float *sums,
                                                                    int64_t *counts)
{
    __shared__ int members[4][kWaveSpan];
    __shared__ int n_list[4];
    const int w = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const bool flat = ids_so == n_inner * ids_si;              // ids contiguous in token order: no division per token
    const bool flat_x = xso == n_inner * xsi;
    constexpr int kMaxPer = 4;                                  // D <= 1024
    float acc[kMaxPer] = {0.0f, 0.0f, 0.0f, 0.0f};
    int64_t total = 0;
    for (int64_t t0 = 0; t0 < n_tokens; t0 += kChunk) {
        __syncthreads();                                        // the lists of the previous round have been consumed
        int n_mine = 0;
        const int64_t w0 = t0 + (int64_t)wv * kWaveSpan;
        for (int i = 0; i < kWaveSpan; i += 8 * SN_WAVE) {         // 8 id loads in flight per lane
            if (w0 + i >= n_tokens) break;                       // (wave-uniform)
            int64_t idv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int64_t t = w0 + i + u * SN_WAVE + lane;
                idv[u] = t < n_tokens ? ids[flat ? t * ids_si : (t / n_inner) * ids_so + (t % n_inner) * ids_si] : (int64_t)-1;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool mine = idv[u] == (int64_t)w;
                const unsigned long long m = __ballot(mine);
                if (mine) members[wv][n_mine + __popcll(m & ((1ull << lane) - 1ull))] = (int)(w0 - t0) + i + u * SN_WAVE + lane;
                n_mine += __popcll(m);
            }
        }
        if (lane == 0) n_list[wv] = n_mine;
        __syncthreads();
        for (int v = 0; v < 4; ++v) {
            const int nm = n_list[v];
            total += nm;
            for (int i0 = 0; i0 < nm; i0 += 8) {
                float val[8][kMaxPer];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int64_t t = t0 + members[v][i0 + u < nm ? i0 + u : nm - 1];
                    const float *row = km_row(x, t, flat_x, n_inner, xso, xsi);
#pragma unroll
                    for (int q = 0; q < kMaxPer; ++q) {
                        const int d = tid + q * kUpdThreads;
                        val[u][q] = d < D ? row[d] : 0.0f;
                    }
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    if (i0 + u < nm) {
#pragma unroll
                        for (int q = 0; q < kMaxPer; ++q) acc[q] += val[u][q];
                    }
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kMaxPer; ++q) {
        const int d = tid + q * kUpdThreads;
        if (d < D) sums[(int64_t)w * D + d] = acc[q];
    }
    if (tid == 0) counts[w] = total;
}

// The same sums from a token order that is already grouped by centre (a stable sort of the ids keeps the token
// order inside a centre): workgroup k adds the rows order[offsets[k]] .. order[offsets[k+1] - 1] one after the
// other, 8 rows in flight - no walk over the id stream, HBM-bound on the token rows.
template <int kMaxPer, int kFlight>          // features per thread (D <= 256 kMaxPer), rows in flight
__global__ __launch_bounds__(kUpdThreads) void kmeans_update_sorted_kernel(const float *x, int64_t n_inner, int64_t xso, int64_t xsi,
                                                                           const int64_t *order, const int64_t *offsets, int D, float *sums,
                                                                           int64_t *counts)
{
    const int w = blockIdx.x, tid = threadIdx.x;
    const bool flat_x = xso == n_inner * xsi;
    float acc[kMaxPer];
#pragma unroll
    for (int q = 0; q < kMaxPer; ++q) acc[q] = 0.0f;
    const int64_t i0 = offsets[w], i1 = offsets[w + 1];
    // the adds of a centre are one sequential chain (SciPy's order), so its time is the HBM latency per batch:
    // as many rows in flight as the registers hold
    for (int64_t i = i0; i < i1; i += kFlight) {
        float val[kFlight][kMaxPer];
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
            const int64_t t = order[i + u < i1 ? i + u : i1 - 1];
            const float *row = km_row(x, t, flat_x, n_inner, xso, xsi);
#pragma unroll
            for (int q = 0; q < kMaxPer; ++q) {
                const int d = tid + q * kUpdThreads;
                val[u][q] = d < D ? row[d] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < kFlight; ++u) {
            if (i + u < i1) {
#pragma unroll
                for (int q = 0; q < kMaxPer; ++q) acc[q] += val[u][q];
            }
        }
    }
#pragma unroll
    for (int q = 0; q < kMaxPer; ++q) {
        const int d = tid + q * kUpdThreads;
        if (d < D) sums[(int64_t)w * D + d] = acc[q];
    }
    if (tid == 0) counts[w] = i1 - i0;
}

// one wave per token: fp64 |x - c|^2 in the oracle's order (lane-strided partial sums + xor butterfly)
__global__ __launch_bounds__(256) void kmeans_distance_kernel(const float *x, int64_t n_tokens, int64_t n_inner, int64_t xso, int64_t xsi,
                                                              const int64_t *ids, int64_t ids_so, int64_t ids_si, const float *centres,
                                                              int K, int D, double *dist)
{
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t t = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); t < n_tokens; t += n_waves) {
        const float *row = km_row(x, t, xso == n_inner * xsi, n_inner, xso, xsi);
        int64_t id = ids[ids_so == n_inner * ids_si ? t * ids_si : (t / n_inner) * ids_so + (t % n_inner) * ids_si];
        id = id < 0 ? 0 : (id >= K ? K - 1 : id);
        const float *c = centres + id * D;
        double p = 0.0;
        for (int k = lane; k < D; k += SN_WAVE) {
            const double d = (double)row[k] - (double)c[k];
            p = fma(d, d, p);
        }
        p = sn_wave_sum_f64(p);
        if (lane == 0) dist[t] = sqrt(p);
    }
}

}  // namespace

extern "C" int sn_kmeans_update(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer, int64_t x_stride_inner,
                                const int64_t *ids, int64_t ids_stride_outer, int64_t ids_stride_inner, int K, int D,
                                float *sums, int64_t *counts, void *stream)
{
    SN_REQUIRE(n_outer >= 0 && n_inner >= 0, SN_ERR_BAD_ARG, "sn_kmeans_update: negative token grid");
    SN_REQUIRE(K > 0 && K <= 65536, SN_ERR_BAD_ARG, "sn_kmeans_update: K=%d out of range", K);
    SN_REQUIRE(D > 0 && D <= 1024, SN_ERR_UNSUPPORTED, "sn_kmeans_update: D=%d must be <= 1024", D);
    SN_REQUIRE(sums && counts && (n_outer * n_inner == 0 || (x && ids)), SN_ERR_BAD_ARG, "sn_kmeans_update: NULL pointer");
    SN_REQUIRE(n_outer * n_inner < 0x7fffffff, SN_ERR_UNSUPPORTED, "sn_kmeans_update: too many tokens");
    hipLaunchKernelGGL(kmeans_update_kernel, dim3((unsigned)K), dim3(kUpdThreads), 0, (hipStream_t)stream, x, n_outer * n_inner,
                       n_inner > 0 ? n_inner : 1, x_stride_outer, x_stride_inner, ids, ids_stride_outer, ids_stride_inner, D, sums, counts);
    SN_CHECK_LAUNCH("sn_kmeans_update");
    return SN_OK;
}

extern "C" int sn_kmeans_distances(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer, int64_t x_stride_inner,
                                   const int64_t *ids, int64_t ids_stride_outer, int64_t ids_stride_inner, const float *centres,
                                   int K, int D, double *dist, void *stream)
{
    SN_REQUIRE(n_outer >= 0 && n_inner >= 0, SN_ERR_BAD_ARG, "sn_kmeans_distances: negative token grid");
    const int64_t n_tokens = n_outer * n_inner;
    if (n_tokens == 0) return SN_OK;
    SN_REQUIRE(x && ids && centres && dist, SN_ERR_BAD_ARG, "sn_kmeans_distances: NULL pointer");
    SN_REQUIRE(K > 0 && D > 0, SN_ERR_BAD_ARG, "sn_kmeans_distances: K=%d D=%d", K, D);
    const int64_t blocks = (n_tokens + 3) / 4;
    hipLaunchKernelGGL(kmeans_distance_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, (hipStream_t)stream, x, n_tokens,
                       n_inner, x_stride_outer, x_stride_inner, ids, ids_stride_outer, ids_stride_inner, centres, K, D, dist);
    SN_CHECK_LAUNCH("sn_kmeans_distances");
    return SN_OK;
}

extern "C" int sn_kmeans_update_sorted(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer, int64_t x_stride_inner,
                                       const int64_t *order, const int64_t *offsets, int K, int D, float *sums, int64_t *counts,
                                       void *stream)
{
    SN_REQUIRE(n_outer >= 0 && n_inner >= 0, SN_ERR_BAD_ARG, "sn_kmeans_update_sorted: negative token grid");
    SN_REQUIRE(K > 0 && K <= 65536, SN_ERR_BAD_ARG, "sn_kmeans_update_sorted: K=%d out of range", K);
    SN_REQUIRE(D > 0 && D <= 1024, SN_ERR_UNSUPPORTED, "sn_kmeans_update_sorted: D=%d must be <= 1024", D);
    SN_REQUIRE(sums && counts && offsets && (n_outer * n_inner == 0 || (x && order)), SN_ERR_BAD_ARG, "sn_kmeans_update_sorted: NULL pointer");
    const int64_t ni = n_inner > 0 ? n_inner : 1;
    if (D <= 256) hipLaunchKernelGGL((kmeans_update_sorted_kernel<1, 64>), dim3((unsigned)K), dim3(kUpdThreads), 0, (hipStream_t)stream, x, ni,
                                     x_stride_outer, x_stride_inner, order, offsets, D, sums, counts);
    else if (D <= 512) hipLaunchKernelGGL((kmeans_update_sorted_kernel<2, 32>), dim3((unsigned)K), dim3(kUpdThreads), 0, (hipStream_t)stream, x, ni,
                                          x_stride_outer, x_stride_inner, order, offsets, D, sums, counts);
    else hipLaunchKernelGGL((kmeans_update_sorted_kernel<4, 16>), dim3((unsigned)K), dim3(kUpdThreads), 0, (hipStream_t)stream, x, ni,
                            x_stride_outer, x_stride_inner, order, offsets, D, sums, counts);
    SN_CHECK_LAUNCH("sn_kmeans_update_sorted");
    return SN_OK;
}
