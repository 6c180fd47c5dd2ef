// S4: the GCN layers of the matcher on the matrix cores (reference schema_inference/graph/gnn.py:20-98,
// match.py:33-76).
//
// The two dense products of a layer, adj @ H and (.) @ W^T, are fp32 GEMMs in the reference and the
// scores must stay within 1e-5 of it, so fp16/bf16 inputs alone are not enough.  Every operand is
// therefore kept as TWO fp16 planes, x s = hi + lo with hi = fp16(x s), lo = fp16(x s - hi), s a power of two
// (v_mfma_f32_32x32x16_f16 neither flushes fp16 subnormals nor rounds the products, tools/mfma_denorm_probe.hip),
// and a product is three MFMAs accumulated in fp32:
//     a.b ~= hi_a.hi_b + hi_a.lo_b + lo_a.hi_b          (the dropped lo.lo term is <= 2^-22 relative)
// which is 3 x 2.5 PF-class instructions instead of one 157 TF-class fp32 MFMA.
//
// What hi + lo holds (the bound the scores rest on).  y = x s.  hi = fp16(y): |y - hi| <= 2^-11 |y|, exactly
// representable in fp32.  lo = fp16(y - hi): relative error 2^-11 while |y - hi| >= 2^-14 (fp16 normal), else the
// ABSOLUTE error 2^-25 of an fp16 subnormal.  So |y - hi - lo| <= max(2^-22 |y|, 2^-25): 22 significant bits for
// |y| >= 2^-3, and an absolute floor of 2^-25 / s in x below that.  Unscaled (s = 1) a typical adjacency entry 1 / 512
// would keep 16 bits and a 512-term row could be off by 512 x 2^-25 = 1.5e-5 of |B|.  Hence every operand carries a
// power-of-two scale s (exact) that puts the LARGEST magnitude it can hold at 2^13 .. 2^14: an element keeps 22 bits
// down to 2^-17 of that maximum and 2^-39 of it absolutely below - whatever the magnitudes of the weights - and
// nothing below 65504 / 4 can overflow.  The scales are device scalars next to the planes (`*_scale`; NULL = 1): static
// for the adjacency (entries <= 2 + |w_e|_1: s = 2^10), from the actual maximum for weight-only operands (table, W2),
// from the LayerNorm bound ceil(sqrt(E - 1)) max|gamma| + max|beta| for operands written by an epilogue.  The accumulators are
// multiplied by 1 / (s_a s_b) (exact) before the bias.  With da, db the representation errors above, a product row
// errs by at most  sum_k (|da_k| |b_k| + |a_k| |db_k| + |lo_a,k lo_b,k|) <= 3.1 x 2^-22 sum_k |a_k b_k| + 2^-38 n max|a| max|b|
// plus the fp32 accumulation of its 3 n MFMA terms - against n 2^-24 sum_k |a_k b_k| for the reference's fp32 GEMM.
//
// One kernel form serves every product of the layer: C[M,N] = A[M,K] . Bt[N,K]^T with BOTH operands
// K-contiguous, so no operand is ever transposed in memory:
//     H1      = LN(adj . Zt1^T + b1)      A = adj planes [n,n]     Bt = Zt1 [E,n]  (gathered table columns)
//     Zt2     = W2 . H1^T                 A = W2 planes  [E,E]     Bt = H1  [n,E]
//     pooled  = w^T LN(adj . Zt2^T + b2)  A = adj planes           Bt = Zt2 [E,n]
// (the second product is computed transposed so that its output is again K-contiguous).
// Bias, padding mask, LayerNorm, ReLU, the hi/lo split of the result and the node-weighted pooling are
// epilogues on the accumulator tile.
//
// Tiling: 256 threads = 4 waves own a 128 x 256 output tile, wave (wm, wn) a 64 x 128 sub-tile = 2 x 4
// MFMA 32x32 accumulators (128 VGPRs); two workgroups per CU (two waves per SIMD that are NOT in
// lockstep: while one workgroup waits at its stage barrier or issues DMA the other multiplies).  A
// k-stage is one MFMA k-step (16): 24 KiB of LDS holding both planes of both operands in MFMA fragment
// order, filled by LDS-DMA (global_load_lds_dwordx4, 6 contiguous 1 KiB blocks per wave per stage) into
// a 3-stage ring (96 KiB per CU in flight: the products stream their operands from HBM at an arithmetic
// intensity near the ridge, so latency hiding decides the speed); one barrier, 24 MFMAs and
// 12 ds_read_b128 per wave per stage.
#include "sn_common.h"

#ifndef SN_GEMM_SADDR
#define SN_GEMM_SADDR 1        // ring copies addressed by SGPR base + 32-bit lane offset (0: a 64-bit address per lane, rounds 1-5)
#endif
#ifndef SN_GEMM_FL_TRANSPOSED
#define SN_GEMM_FL_TRANSPOSED 1   // fused next-layer product: output tile as [node][feature] (plane pieces straight from registers); 0: rounds 3-5 (through LDS)
#endif
#ifndef SN_GEMM_ABLATE
#define SN_GEMM_ABLATE 0       // lab builds only (results are garbage): 1 no MFMAs, 2 no fragment reads, 4 no ring copies, 8 no copies of A, 16 no copies of B, 32 no stage barrier
#endif
#ifndef SN_GEMM_ISSUE_AT
#define SN_GEMM_ISSUE_AT 1     // where in a stage the ring copies of stage t + 2 are issued: 0 in front of the MFMAs, 1 behind the first 8, 2 behind 16 (fewer loop cycles, the same launch time)
#endif

#include <hip/hip_fp16.h>
#include <type_traits>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTileN = 256;             // output columns of a workgroup
constexpr int kStageK = 16;             // k per stage (one MFMA k-step)
constexpr int kChunksB = (kTileN / 32) * 2;          // 1 KiB blocks of B per stage: [column block][plane]
// Two tile heights (round 6).  TM = 128: 4 waves, 24 KiB stages, a 3-slot ring, two workgroups per CU (small graphs: the instance
// side, compacted class graphs).  TM = 256: 8 waves (wave (wm, wn) still owns 64 x 128), 32 KiB stages, a 4-slot ring, ONE workgroup
// per CU: a k-stage of a CU's 256 rows moves 32 KiB through the vector memory path where two 128-row workgroups move 48 -
// the ring of the 128-row form delivers ~27 B per cycle and CU where its MFMAs need 31, this one needs 21.
// The second wave of every SIMD (waves 4-7) runs half a stage behind the first, so that one of a SIMD's two waves always has
// MFMAs to issue while the other sits in the barrier, the copy issue or the fragment reads.
// TM = 64 (round 6, the FUSED product of small graphs only: instance graphs of ~110 vertices, compacted classes): 4 waves side by side,
// each 64 rows x 64 columns (2 x 2 accumulators), 20 KiB stages; a tile's 64 nodes are exactly one pass of the fused next-layer
// product, so a graph of <= 128 vertices is two workgroups that share nothing - where the 128-row form ran the main product and the
// LayerNorm TWICE (tile + twin) to split its two passes over two workgroups.
template <int TM> struct Geom {
    static constexpr int kTileM = TM;
    static constexpr int kWavesM = TM >= 128 ? TM / 64 : 1, kWavesN = TM >= 128 ? 2 : 4, kWaves = kWavesM * kWavesN, kThreads = 64 * kWaves;
    static constexpr int NJ = kTileN / 32 / kWavesN;                            // 32-column accumulators per wave and row block: 4 / 2
    static constexpr int kChunksA = (TM / 32) * 2;                              // 1 KiB blocks of A per stage: [row block][plane]
    static constexpr int kStageBytes = (kChunksA + kChunksB) * 1024;            // 24 / 32 KiB
    static constexpr int kRing = TM <= 128 ? 3 : 4;                             // LDS stages: kRing - 1 in flight
    static constexpr int kDmaPerWave = (kChunksA + kChunksB) / kWaves;          // 6 / 4
    // epilogue scratch: LayerNorm row statistics [5][TM] floats at 0; pooled partials [kWavesM][256] floats; the per-wave staging
    // of the plane stores (16 x kC8Stride dwords per wave) - TM = 128 keeps the offsets it always had
    static constexpr int kPoolOff = TM == 128 ? 2048 : 5 * TM * 4;
    static constexpr int kStgOff = TM == 128 ? 4096 : kPoolOff + kWavesM * 256 * 4;
    static constexpr int kStgBytes = kWaves * 16 * (32 * 8 + 8) * 4;
    static constexpr int kFragImage = 2 * 16 * 2 * 1056;                        // TM = 64: the fragment image of the fused product (64 nodes)
    static constexpr int kLdsBytes = TM < 128 ? (kRing * kStageBytes > kFragImage ? kRing * kStageBytes : kFragImage)
                                              : (kRing * kStageBytes > kStgOff + kStgBytes ? kRing * kStageBytes : kStgOff + kStgBytes);
};
constexpr int kBlockElems = 512;        // fp16 elements of one 32-row x 16-k block (1 KiB)
constexpr int kMaxPerm = 1024;          // vertices per class graph the compacted atlas producer stages a permutation for

// Operand planes are stored BLOCKED, in the order the MFMA consumes them: plane[g][row >> 5][k >> 4] is a
// 1 KiB block holding element (row, k) at ((k >> 3 & 1) * 32 + (row & 31)) * 8 + (k & 7) - i.e. lane
// (r, h) of a wave finds its 8 consecutive k of row r at byte lane * 16.  One LDS-DMA instruction moves
// one block (contiguous 1 KiB of global memory -> contiguous 1 KiB of LDS), every ds_read_b128 of a
// fragment is lane-linear, and consecutive k-stages of a row block are consecutive blocks.
__host__ __device__ inline int64_t blocked_index(int row, int k, int kb_count)
{
    return ((int64_t)(row >> 5) * kb_count + (k >> 4)) * kBlockElems + (((k >> 3) & 1) * 32 + (row & 31)) * 8 + (k & 7);
}

__device__ __forceinline__ void split2(float x, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// 8 consecutive k of one row -> one 16-byte piece per plane
__device__ __forceinline__ void store_piece(const float (&v)[8], _Float16 *out_h, _Float16 *out_l, int64_t o)
{
    half8 h, l;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        _Float16 a, b;
        split2(v[e], a, b);
        h[e] = a;
        l[e] = b;
    }
    *reinterpret_cast<half8 *>(out_h + o) = h;
    *reinterpret_cast<half8 *>(out_l + o) = l;
}

// ------------------------------------------------------------------ producers of operand planes
// adj = (E + E^T) / 2 + I (reference gnn.py:27-30) as blocked hi/lo planes, rows and columns >= n zero.
// One workgroup per 64 x 64 tile: E tile and E^T tile through LDS, then one 16-byte piece per thread pair.
// With `rowsum` the input is the pruned, un-normalised atlas: the edge value is then
// nan_to_num(max(x, 0) * rowsum[row]) (rowsum = 1 / row sum; zero on the diagonal when remove_self_loop),
// what atlas_normalize_kernel would have written (schema_net.py:152-175) - the normalised atlas is never stored.
// adj is symmetric: a workgroup loads the tile pair E[I][J], E[J][I] once and writes BOTH adj[I][J] and
// adj[J][I] (each atlas byte is read once, not twice).  Workgroup x of a graph owns row tiles x and T-1-x
// and walks J >= I for each (T + 1 tile pairs per workgroup, whatever x): few fat workgroups - one per
// 64 x 64 tile would be bound by the dispatch rate (~10 ns per workgroup chip-wide), not by HBM.
// Caller-supplied edges (no `rowsum`: the public GNN.forward / gcn_adjacency_planes route) carry the STATIC scale 2^10, chosen
// for adjacencies of normalised graphs (entries <= 3): an entry of magnitude >= 64 would leave the fp16 range and its inf
// would turn every score of the graph into NaN.  Such entries saturate at the largest fp16 instead (the product is then
// off by the clipped amount - the split cannot represent it - but finite); NaN entries stay NaN, as in the reference.
__device__ __forceinline__ float saturate_f16_range(float x)
{
    const float c = __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f);
    return x != x ? x : c;
}

// VEC (round 4; no permutation, n % 4 == 0, 16-byte aligned tensors): the tile pair is read - and the by-product written - with
// 16-byte accesses, thread = 4 consecutive columns of the rows r16, r16 + 16, ...; otherwise thread = one column of the rows ty, ty + 4, ...
template <bool VEC>
__global__ __launch_bounds__(256) void adjacency_planes_kernel(const float *edges, int n, int kb_count, int64_t batch_stride,
                                                               _Float16 *out_h, _Float16 *out_l, const float *rowsum, int remove_self_loop,
                                                               const int32_t *extent, const int32_t *n_valid, int pair_tiles, float scale,
                                                               float *edges_out, const int32_t *perm, int ext_stride, int graph_fast_flag)
{
    __shared__ float te[64][65], tt[64][65];
    __shared__ float rs_i[64], rs_j[64];
    // `perm` (compacted class graphs, round 4; n <= kMaxPerm): vertex a of the operand is vertex perm[g][a] of the stored graph
    // - the kept vertices of a pruned IR-Atlas class first, in their own order -, extent / n_valid are then per graph
    // (ext_stride 1): the operand is the n_kept x n_kept corner + identity, rows and k beyond the class's own extent are
    // neither produced nor read by sn_gcn_gemm.  Staged once per workgroup.
    __shared__ int perm_s[kMaxPerm];
    // Pair mode (small graphs, pair_tiles > 0): the grid is (graphs, pairs) - graph fastest.  With (pairs, graphs) the few pairs a small
    // graph really has (3 of 10 for <= 128 vertices of a graph padded to 4 tiles a side) sat at a fixed period of the dispatch order, and
    // the active workgroups fell on the even XCDs twice as often as on the odd ones (round 6; `SN_ADJ_GRAPH_MAJOR=0`: the old order).
    const bool graph_fast = pair_tiles > 0 && gridDim.y == (unsigned)(pair_tiles * (pair_tiles + 1) / 2) && graph_fast_flag;
    const int g = graph_fast ? blockIdx.x : blockIdx.y;
    const int bx = graph_fast ? blockIdx.y : blockIdx.x;
    if (perm) {
        for (int i = threadIdx.x; i < n; i += 256) perm_s[i] = perm[(int64_t)g * n + i];
        __syncthreads();
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const float *e = edges + (int64_t)g * n * n;
    // elements outside the graph's own nv x nv corner count as 0 and are not read (their storage may be unwritten);
    // the identity below still covers all n rows
    const int nv = n_valid ? min(max(n_valid[g], 0), n) : n;
    int rows_lim = (n + 31) & ~31, k_lim = kb_count * 16;
    if (extent) {                                              // nothing beyond the largest graph of the batch (ext_stride 1: of this graph) is ever read
        const int ext = extent[(int64_t)g * ext_stride];
        rows_lim = min(rows_lim, (ext + 31) & ~31);
        k_lim = min(k_lim, (ext + 15) & ~15);
    }
    const int T = (rows_lim + 63) / 64;                        // rows_lim >= k_lim: tiles that matter, both ways
    // Small graphs (pair_tiles > 0 = tiles per side of the padded graph, at most 4): one workgroup per tile pair I <= J
    // instead of a walk - with 1-3 pairs per graph that matter the walk was a chain of dependent rounds, 17 us alone and
    // 53 us beside other kernels for 34 MB (the instance side of the bench); pairs beyond the batch's extent exit at once.
    int pair_I = -1, pair_J = -1;
    if (pair_tiles > 0) {
        int x = bx;
        for (int i = 0; i < pair_tiles && pair_I < 0; ++i) {
            if (x < pair_tiles - i) { pair_I = i; pair_J = i + x; }
            x -= pair_tiles - i;
        }
        if (pair_I < 0 || pair_J >= T) return;                 // (whole workgroup, before any barrier)
    }
    // The tile pairs of this workgroup in walking order: (I0 = x; J = I0 .. T-1), then (I1 = T-1-x; J = I1 .. T-1) when
    // I1 > x (odd T: the middle tile once); pair mode: the one pair.  The NEXT pair's tiles are requested as soon as the
    // current pair's are in LDS, so its loads are in flight under the conversion and the plane stores (a walk used to be
    // T + 1 rounds of load latency + store latency, one after the other, with 1.5 workgroups per CU to hide them).
    const int x0 = bx;
    int cI, cJ, chalf = 0;
    bool have;
    // pair_tiles < 0 (large graphs, round 4): the T (T + 1) / 2 tile pairs of the graph are dealt round-robin over the graph's
    // gridDim.x workgroups (pair p -> workgroup p % gridDim.x): a walk of T + 1 pairs per workgroup was T + 1 rounds of HBM latency
    // with ~1.5 workgroups per CU (69 us for the 210 MB of the bench's class graphs, 3 TB/s); ~4 workgroups per CU walk a
    // quarter of that each.
    int cp = x0;
    auto pair_of = [&](int p, int &I, int &J) -> bool {          // linear pair index -> (I, J >= I); false beyond the last pair
        int i = 0;
        while (i < T && p >= T - i) { p -= T - i; ++i; }
        I = i; J = i + p;
        return i < T;
    };
    if (pair_tiles > 0) { cI = pair_I; cJ = pair_J; have = true; }
    else if (pair_tiles < 0) { have = pair_of(cp, cI, cJ); }
    else {
        cI = x0; cJ = x0; have = cI < T;
        if (!have) { chalf = 1; cI = T - 1 - x0; cJ = cI; have = cI >= 0 && cI < T && cI > x0; }
    }
    auto advance = [&](int &I, int &J, int &half) -> bool {       // -> the pair after (I, J), false when the walk is over
        if (pair_tiles > 0) return false;
        if (pair_tiles < 0) { cp += (int)gridDim.x; return pair_of(cp, I, J); }
        if (J + 1 < T) { ++J; return true; }
        if (half == 1) return false;
        half = 1; I = T - 1 - x0; J = I;
        return I >= 0 && I < T && I > x0;
    };
    float ve[16], vt[16], rsi = 0.0f, rsj = 0.0f;
    // branch-free loads (clamped index + select): a conditional load per element would serialise them
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int c4 = (threadIdx.x & 15) * 4, r16 = threadIdx.x >> 4;          // VEC: columns c4 .. c4 + 3 of the rows r16 + 16 it
    f32x4 we[4], wt[4];
    auto fetch = [&](int I, int J) {
        const int bi = I * 64, bj = J * 64;
        if constexpr (VEC) {
            if (rowsum && threadIdx.x < 64) {
                const int ri = bi + (int)threadIdx.x, rj = bj + (int)threadIdx.x;
                rsi = ri < n ? rowsum[(int64_t)g * n + ri] : 0.0f;
                rsj = rj < n ? rowsum[(int64_t)g * n + rj] : 0.0f;
            }
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = r16 + 16 * it;
                const int i = bi + rr, j = bj + c4, i2 = bj + rr, j2 = bi + c4;
                const bool in = i < nv && j < n, in2 = i2 < nv && j2 < n;         // (n % 4 == 0: a row's last piece is whole)
                f32x4 a = *reinterpret_cast<const f32x4 *>(e + (in ? (int64_t)i * n + j : 0));
                f32x4 b = *reinterpret_cast<const f32x4 *>(e + (in2 ? (int64_t)i2 * n + j2 : 0));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[q] = (in && j + q < nv) ? a[q] : 0.0f;
                    b[q] = (in2 && j2 + q < nv) ? b[q] : 0.0f;
                }
                we[it] = a; wt[it] = b;
            }
            return;
        }
        if (rowsum && threadIdx.x < 64) {
            const int ri = bi + (int)threadIdx.x, rj = bj + (int)threadIdx.x;
            rsi = ri < n ? rowsum[(int64_t)g * n + (perm ? perm_s[ri] : ri)] : 0.0f;
            rsj = rj < n ? rowsum[(int64_t)g * n + (perm ? perm_s[rj] : rj)] : 0.0f;
        }
        const int pj = perm ? perm_s[min(bj + tx, n - 1)] : bj + tx, pj2 = perm ? perm_s[min(bi + tx, n - 1)] : bi + tx;
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int rr = ty + 4 * it;
            const int i = bi + rr, j = bj + tx;
            const bool ok = i < nv && j < nv;
            const int pi = perm ? perm_s[min(i, n - 1)] : i;
            ve[it] = e[ok ? (int64_t)pi * n + pj : 0];                    // E[I][J] tile, [i - bi][j - bj]
            const int i2 = bj + rr, j2 = bi + tx;
            const bool ok2 = i2 < nv && j2 < nv;
            const int pi2 = perm ? perm_s[min(i2, n - 1)] : i2;
            vt[it] = e[ok2 ? (int64_t)pi2 * n + pj2 : 0];                 // E[J][I] tile, [j - bj][i - bi]
            ve[it] = ok ? ve[it] : 0.0f;
            vt[it] = ok2 ? vt[it] : 0.0f;
        }
    };
    if (have) fetch(cI, cJ);
    while (have) {
        const int I = cI, J = cJ, bi = I * 64, bj = J * 64;
        __syncthreads();                                   // previous tile pair fully consumed
        if (rowsum && threadIdx.x < 64) { rs_i[threadIdx.x] = rsi; rs_j[threadIdx.x] = rsj; }
        __syncthreads();                                   // row scales visible
        if constexpr (VEC) {
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int rr = r16 + 16 * it;
                f32x4 a = we[it], b = wt[it];
                if (rowsum) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        a[q] = fminf(fmaxf(a[q], 0.0f), 3.402823466e+38f) * rs_i[rr];      // row bi + rr
                        b[q] = fminf(fmaxf(b[q], 0.0f), 3.402823466e+38f) * rs_j[rr];      // row bj + rr
                        if (remove_self_loop && bi + rr == bj + c4 + q) a[q] = 0.0f;
                        if (remove_self_loop && bj + rr == bi + c4 + q) b[q] = 0.0f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) { te[rr][c4 + q] = a[q]; tt[rr][c4 + q] = b[q]; }
                if (edges_out) {
                    float *eo = edges_out + (int64_t)g * n * n;
                    if (bi + rr < n && bj + c4 < n) *reinterpret_cast<f32x4 *>(eo + (int64_t)(bi + rr) * n + bj + c4) = a;
                    if (J != I && bj + rr < n && bi + c4 < n) *reinterpret_cast<f32x4 *>(eo + (int64_t)(bj + rr) * n + bi + c4) = b;
                }
            }
        } else {
        if (rowsum) {
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int rr = ty + 4 * it;
                // == nan_to_num(max(x, 0) / row sum): the scale is finite (0 for empty / non-finite rows), NaN
                // weights clamp to 0 in fmaxf, +inf to FLT_MAX
                ve[it] = fminf(fmaxf(ve[it], 0.0f), 3.402823466e+38f) * rs_i[rr];   // row bi + rr
                vt[it] = fminf(fmaxf(vt[it], 0.0f), 3.402823466e+38f) * rs_j[rr];   // row bj + rr
                if (remove_self_loop && bi + rr == bj + tx) ve[it] = 0.0f;
                if (remove_self_loop && bj + rr == bi + tx) vt[it] = 0.0f;
            }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            te[ty + 4 * it][tx] = ve[it];
            tt[ty + 4 * it][tx] = vt[it];
        }
        if (edges_out) {
            // by-product (atlas form only): the normalised class edges themselves, [n, n] fp32 per graph - what
            // atlas_normalize_kernel writes (schema_net.py:152-175), for callers that return `class_edges` next to
            // the scores (SchemaNetPredictor's dictionary) without a second pass over the atlas
            float *eo = edges_out + (int64_t)g * n * n;
#pragma unroll
            for (int it = 0; it < 16; ++it) {
                const int rr = ty + 4 * it;
                if (bi + rr < n && bj + tx < n) eo[(int64_t)(bi + rr) * n + bj + tx] = ve[it];
                if (J != I && bj + rr < n && bi + tx < n) eo[(int64_t)(bj + rr) * n + bi + tx] = vt[it];
            }
        }
        }
        __syncthreads();
        have = advance(cI, cJ, chalf);
        if (have) fetch(cI, cJ);                           // (registers are free again; LDS holds the current pair)
        // pieces: 64 rows x 8 (k / 8) per output tile; thread -> row tx, pieces ty, ty + 4
        for (int pc = ty; pc < 8; pc += 4) {
            {   // adj[I][J]: row i = bi + tx, k = bj + 8 pc ..
                const int i = bi + tx, j0 = bj + pc * 8;
                if (i < rows_lim && j0 < k_lim) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int j = j0 + q;
                        float x = 0.0f;
                        if (i < n && j < n) {
                            x = (te[tx][pc * 8 + q] + tt[pc * 8 + q][tx]) * 0.5f;       // == / 2 exactly
                            if (i == j) x = x + 1.0f;
                        }
                        v[q] = rowsum ? x * scale : saturate_f16_range(x * scale);
                    }
                    store_piece(v, out_h, out_l, (int64_t)g * batch_stride + blocked_index(i, j0, kb_count));
                }
            }
            if (J != I) {   // adj[J][I] = adj[I][J]^T: row j = bj + tx, k = bi + 8 pc ..
                const int j = bj + tx, i0 = bi + pc * 8;
                if (j < rows_lim && i0 < k_lim) {
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int i = i0 + q;
                        float x = 0.0f;
                        if (i < n && j < n) {
                            x = (tt[tx][pc * 8 + q] + te[pc * 8 + q][tx]) * 0.5f;
                            if (i == j) x = x + 1.0f;
                        }
                        v[q] = rowsum ? x * scale : saturate_f16_range(x * scale);
                    }
                    store_piece(v, out_h, out_l, (int64_t)g * batch_stride + blocked_index(j, i0, kb_count));
                }
            }
        }
    }
}

// Zt[g][f][j] = table[ids[g][j]][f] as blocked planes (rows f < E, k = j; j >= n and ids outside the
// table give zero).  (layer 1 re-associated: adj @ Emb[ids] @ W^T == adj @ (Emb @ W^T)[ids], gnn.py:64-66 + 30)
__global__ __launch_bounds__(256) void gather_planes_kernel(const float *table, int rows_table, const int64_t *ids, int n, int kb_count,
                                                            int E, int64_t batch_stride, _Float16 *out_h, _Float16 *out_l, const int32_t *extent,
                                                            const float *scale_dev)
{
    __shared__ float tile[64][65];
    __shared__ int rid[64];
    const float scale = scale_dev ? *scale_dev : 1.0f;
    const int g = blockIdx.y, j0 = blockIdx.x * 64;
    if (extent && j0 >= ((*extent + 15) & ~15)) return;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    if (threadIdx.x < 64) {
        const int j = j0 + threadIdx.x;
        const int64_t id = j < n ? ids[(int64_t)g * n + j] : -1;
        rid[threadIdx.x] = (id >= 0 && id < rows_table) ? (int)id : -1;
    }
    __syncthreads();
    for (int fbase = 0; fbase < E; fbase += 256) {
        // all table reads of this 64-node x 256-feature slab in flight before the first use (16 rows x 4 per lane)
        float v[16][4];
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {               // wave wid owns the table rows of nodes wid*16 .. +15
            const int row = rid[wid * 16 + rr];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int f = fbase + c * 64 + lane;
                const bool ok = row >= 0 && f < E;
                const float x = table[ok ? (int64_t)row * E + f : 0];          // branch-free: clamped index + select
                v[rr][c] = ok ? x : 0.0f;
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int f0 = fbase + c * 64;
            if (f0 >= ((E + 31) & ~31)) break;
#pragma unroll
            for (int rr = 0; rr < 16; ++rr) tile[wid * 16 + rr][lane] = v[rr][c];
            __syncthreads();
            for (int pc = wid; pc < 8; pc += 4) {           // piece = (feature f0 + lane, nodes j0 + 8 pc .. + 7)
                const int f = f0 + lane, jj = j0 + pc * 8;
                if (f < ((E + 31) & ~31) && jj < kb_count * 16) {
                    float w[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) w[q] = f < E ? tile[pc * 8 + q][lane] * scale : 0.0f;
                    store_piece(w, out_h, out_l, (int64_t)g * batch_stride + blocked_index(f, jj, kb_count));
                }
            }
            __syncthreads();
        }
    }
}

// fp32 x [batches][rows_x][ld] (cols_x valid per row) -> blocked planes of the operand x (rows = rows_x, k = cols_x) or, with
// TR, of x^T (rows = cols_x, k = rows_x: the B operand of Y = adj . X without a transposed copy of X), zero padded to 32
// rows / 16 k.  A tile = 32 operand rows x up to kSplitKc k through LDS: coalesced row reads (256-byte runs; TR: 128-byte
// runs of 32 columns), then 16-byte pieces with consecutive threads on consecutive rows of a k block - 512 contiguous bytes
// per plane and half wave.  (Round 1's form read one 32-byte piece per thread at a row stride: 2 GB in 2.36 ms against
// 0.88 ms for a copy.)  Workgroups walk the tiles t, t + grid, ...; up to 66 KB of LDS each, so two or more share a CU.
constexpr int kSplitThreads = 512, kSplitKc = 512, kSplitPad = 4;
template <bool TR>
__global__ __launch_bounds__(kSplitThreads) void split_planes_kernel(const float *x, int rows_x, int cols_x, int64_t ld, int64_t x_batch_stride,
                                                                     int kb_count, int64_t batch_stride, _Float16 *out_h, _Float16 *out_l,
                                                                     const float *scale_dev, int kc, int row_blocks, int k_chunks, int64_t tiles,
                                                                     const int32_t *node_extent)
{
    extern __shared__ __attribute__((aligned(16))) float sp_tile[];      // [32][kc + kSplitPad]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ldt = kc + kSplitPad;
    const float scale = scale_dev ? *scale_dev : 1.0f;
    const int op_rows = TR ? cols_x : rows_x, op_k = TR ? rows_x : cols_x;
    for (int64_t t = blockIdx.x; t < tiles; t += gridDim.x) {
        const int kci = (int)(t % k_chunks), rb = (int)((t / k_chunks) % row_blocks), g = (int)(t / ((int64_t)k_chunks * row_blocks));
        const int k0 = kci * kc, r0 = rb * 32;
        // node_extent (training with compacted class graphs: x = [G, nodes, features]): the nodes of graph g from node_extent[g] on are
        // pad rows that every consumer skips (per-graph extents of sn_gcn_gemm) - tiles that hold nothing else are not produced
        if (node_extent && (TR ? k0 : r0) >= node_extent[g]) continue;     // (whole workgroup, in front of the tile's barriers)
        const int kn = min(kc, kb_count * 16 - k0);                         // k of this tile (a multiple of 16), zero beyond op_k
        const float *xg = x + (int64_t)g * x_batch_stride;
        if (!TR) {
            // wave w: tile rows w, w + 8, w + 16, w + 24; a row = kn floats, lane c = lane + 64 j
            float v[4][kSplitKc / 64];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = r0 + wid + 8 * i;
#pragma unroll
                for (int j = 0; j < kSplitKc / 64; ++j) {
                    const int c = lane + 64 * j;
                    const bool ok = r < op_rows && c < kn && k0 + c < op_k;
                    v[i][j] = ok ? xg[(int64_t)r * ld + k0 + c] : 0.0f;
                }
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < kSplitKc / 64; ++j) {
                    const int c = lane + 64 * j;
                    if (c < kn) sp_tile[(wid + 8 * i) * ldt + c] = v[i][j] * scale;
                }
        } else {
            // x rows (= operand k) kk = 2 wid + (lane >> 5) + 16 i; 32 consecutive x columns (= operand rows) per x row
            const int col = lane & 31, sub = lane >> 5;
            float v[kSplitKc / 16];
#pragma unroll
            for (int i = 0; i < kSplitKc / 16; ++i) {
                const int kk = 2 * wid + sub + 16 * i;
                const bool ok = kk < kn && k0 + kk < op_k && r0 + col < op_rows;
                v[i] = ok ? xg[(int64_t)(k0 + kk) * ld + r0 + col] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < kSplitKc / 16; ++i) {
                const int kk = 2 * wid + sub + 16 * i;
                if (kk < kn) sp_tile[col * ldt + kk] = v[i] * scale;
            }
        }
        __syncthreads();
        const int pieces = 32 * (kn / 8);
        for (int q = tid; q < pieces; q += kSplitThreads) {
            const int rr = q & 31, pk = q >> 5;
            const float4 a4 = *reinterpret_cast<const float4 *>(sp_tile + rr * ldt + pk * 8);
            const float4 b4 = *reinterpret_cast<const float4 *>(sp_tile + rr * ldt + pk * 8 + 4);
            const float w[8] = {a4.x, a4.y, a4.z, a4.w, b4.x, b4.y, b4.z, b4.w};
            store_piece(w, out_h, out_l, (int64_t)g * batch_stride + blocked_index(r0 + rr, k0 + pk * 8, kb_count));
        }
        __syncthreads();
    }
}

// mask + LayerNorm + activation of x [G][n][E] (exactly sn_mask_layernorm_act: sn_layernorm_row) and the hi/lo split of the
// result into the blocked planes of an [n, E] operand per graph (exactly split_planes_kernel), without storing the fp32
// rows in between: 8 bytes per element of HBM traffic instead of 24.  A tile = one block row (32 rows) of one graph: the
// eight waves of a workgroup normalise four rows each into an LDS tile, then every thread converts 16-byte pieces -
// consecutive threads, consecutive rows of one k block, i.e. 512 contiguous bytes per plane and half wave.  The workgroups
// are persistent (two per CU, each walking tiles t, t + grid, ...): the tile takes 132 KB of LDS at E = 1024, so with one
// workgroup per tile a CU ran its workgroups one after the other, each paying launch, first-load latency and drain alone
// (0.6 of 1.27 ms with loads and stores compiled out); the next tile's rows are requested before the current tile's pieces
// are stored, so the one resident workgroup keeps HBM busy in both phases.
constexpr int kLnSplitThreads = 512, kLnSplitPad = 4;      // row stride E + 4 floats: 16 lanes' b128 reads cover all banks
__global__ __launch_bounds__(kLnSplitThreads) void layernorm_split_planes_kernel(const float *x, int n, int E, const int32_t *n_valid,
                                                                                 const float *gamma, const float *beta, float eps, int relu,
                                                                                 int kb_count, int64_t batch_stride, _Float16 *out_h,
                                                                                 _Float16 *out_l, const float *scale_dev, int row_blocks,
                                                                                 int64_t tiles)
{
    extern __shared__ __attribute__((aligned(16))) float ln_tile[];      // [32][E + kLnSplitPad]
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int ldt = E + kLnSplitPad;
    const float scale = scale_dev ? *scale_dev : 1.0f;
    float gm[SN_LN_MAX], bt[SN_LN_MAX];
    sn_layernorm_coeffs(gm, bt, lane, E, gamma, beta);
    float v[4][SN_LN_MAX];
    auto request = [&](int64_t t) {                                       // rows wid, wid + 8, ... of tile t -> v
        const int g = (int)(t / row_blocks), rb = (int)(t % row_blocks);
        const int nv = (n_valid && t < tiles) ? n_valid[g] : n;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = rb * 32 + wid + 8 * i;
            const bool live = t < tiles && r < n && r < nv;
            const float *p = x + ((int64_t)g * n + r) * E;
#pragma unroll
            for (int k = 0; k < SN_LN_MAX; ++k) {
                const int c = lane + SN_WAVE * k;
                v[i][k] = (c < E && live) ? p[c] : 0.0f;
            }
        }
    };
    int64_t t = blockIdx.x;
    if (t < tiles) request(t);
    for (; t < tiles; t += gridDim.x) {
        const int g = (int)(t / row_blocks), rb = (int)(t % row_blocks);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int rr = wid + 8 * i;
            const bool row_exists = rb * 32 + rr < n;                      // rows that pad the operand to 32 stay zero
            if (row_exists) sn_layernorm_row(v[i], lane, E, gm, bt, eps, relu);
#pragma unroll
            for (int k = 0; k < SN_LN_MAX; ++k) {
                const int c = lane + SN_WAVE * k;
                if (c < E) ln_tile[rr * ldt + c] = row_exists ? v[i][k] * scale : 0.0f;
            }
        }
        __syncthreads();
        request(t + gridDim.x);                                            // (in flight under the stores below)
        const int pieces = 32 * kb_count * 2;
        for (int q = tid; q < pieces; q += kLnSplitThreads) {
            const int rr = q & 31, pk = q >> 5;
            const float4 a = *reinterpret_cast<const float4 *>(ln_tile + rr * ldt + pk * 8);
            const float4 b = *reinterpret_cast<const float4 *>(ln_tile + rr * ldt + pk * 8 + 4);
            const float w[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            store_piece(w, out_h, out_l, (int64_t)g * batch_stride + blocked_index(rb * 32 + rr, pk * 8, kb_count));
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ the GEMM
struct GemmArgs {
    const _Float16 *a_hi, *a_lo;
    int64_t a_batch_stride;
    const _Float16 *b_hi, *b_lo;
    int64_t b_batch_stride;
    int m, n, k;
    float *c;
    int64_t c_batch_stride;
    int ldc;
    _Float16 *c_hi, *c_lo;
    int64_t cp_batch_stride;
    int cp_cols;
    const float *bias, *gamma, *beta;
    float eps;
    int relu;
    const int32_t *rows_valid;
    const float *pool_w;
    int64_t pool_w_stride;
    float *pooled;
    int batches, tiles_x, tiles_y;   // logical grid (the launch is 1-D, see the XCD remap in the kernel)
    const int32_t *m_extent, *k_extent;   // device scalars (or NULL): rows / k beyond them are never consumed downstream
    int ext_stride;                       // 0: one value for the batch; 1: per graph (m_extent[batch], k_extent[batch])
    int accumulate;                       // c += result (plain products only)
    int zero_skipped;                     // row tiles past the extent are written as zeros
    int pooled_parts;                     // partial sums per graph in `pooled` (>= tiles_y: the caller may keep further slots)
    unsigned long long *stamps;      // diagnostics (sn_debug_set_gemm_stamps): 16 u64 per wave ([8], [9]: the 100 MHz clock at entry / end)
    int nt_a, nt_b;                  // stream that operand past the caches (read once by one workgroup)
    // gathered B (GB kernels): Bt[g][f][j] = table[ids[g][j]][f] from row-major fp16 hi/lo tables [tab_rows + 1][256]
    // (row tab_rows is zero: it stands for j >= ids_n and for ids outside the table)
    const _Float16 *tab_hi, *tab_lo;
    const int64_t *ids;
    int64_t ids_stride;
    int ids_n, tab_rows, tab_ld;            // tab_ld: features per table row (= n: 256, or a multiple of it - the workgroup gathers its own 256-column slice)
    // fused second product (FL kernels): planes of W2 [256, 256] with its columns in the order the epilogue holds them
    const _Float16 *w2_hi, *w2_lo;
    int fl_twin;                     // idle row tiles take the second half of the fused epilogue (SN_GEMM_FL_TWIN=0: off)
    int tile_major;                  // block order inside an XCD: all graphs' tile 0, then tile 1, ... (per-graph extents)
    int stagger;                     // 256-row tiles: waves 4-7 half a stage behind waves 0-3 (SN_GEMM_STAGGER=0: off)
    // power-of-two operand scales (device scalars, NULL = 1; see the header comment): the planes of A / B hold x * scale;
    // output planes are written as result * out_scale; FL: the W2 planes hold W * w2_scale, the H fragments H * h_scale
    const float *a_scale, *b_scale, *out_scale, *w2_scale, *h_scale;
};
constexpr int kMaxGatherK = 1024;      // nodes per graph the gathered-B form stages ids for
static unsigned long long *g_gemm_stamps = nullptr;

// GB: the B operand is not read from blocked planes but gathered from a row-major table (the layer-1 operand
// Zt1[f][j] = (Emb W1^T)[ids[j]][f], gnn.py:64-66 + 30, which gather_planes_kernel would otherwise write to HBM and this
// kernel read back).  A stage of B is then 16 table rows (one per node, 512 bytes per plane) copied by LDS-DMA with
// per-lane source addresses into a row-major [plane][node][256 features] image, and a B fragment - feature r, eight
// consecutive nodes - is two transposing LDS reads (ds_read_b64_tr_b16: a 4-row x 16-column block per 16 lanes,
// delivered column-major).  The 16-byte chunks of row `node` are stored at chunk ^ ((node & 3) << 2): the four rows of
// a block then lie in four different 64-byte bank groups (conflict-free).  Same products in the same order as with the
// gathered planes: bit-identical results.
// FL: the tile that leaves the LayerNorm epilogue - H1, 128 nodes x 256 features - is multiplied by the NEXT layer's Linear
// weight before it leaves the workgroup: Zt2[o][node] = sum_f W2[o][f] H1[node][f] (gnn.py:29 of layer 2, computed transposed
// like the stand-alone product so that its result is K-contiguous for the product that follows), written as blocked planes
// [256 rows = o][k = node].  H1 itself is never stored: one launch, one H1 write and one H1 read per layer pair less.
// The accumulators hold lane <-> feature, register <-> node, and an MFMA operand wants lane <-> node with eight
// consecutive k per lane, so half a tile at a time (the 32-row blocks i = 0, then i = 1 of every wave: 64 nodes) goes
// through LDS as fp16 hi / lo B fragments - 64 KB of the 72 KB ring, which is dead by then.  The contraction index may be
// visited in any order as long as both operands agree: slot kappa = (wn * 32 + r) * 4 + j stands for feature
// (4 wn + j) * 32 + r, i.e. the four features a lane holds (j = 0..3) are four consecutive slots, a lane stores them as ONE
// 8-byte piece per (node, plane), and the host permutes the columns of W2 the same way once (GNN.prepare).  Wave w then
// owns the output features [64 w, 64 w + 64) for the 64 nodes of the half: 2 x 2 accumulators, 16 k-steps of 12 MFMAs,
// A fragments (W2, 256 KB, L2-resident) straight from global memory three k-steps ahead, B fragments from the LDS image.
template <bool LN, bool GB = false, bool FL = false, int TM = 128>
__global__ __launch_bounds__(Geom<TM>::kThreads, 2) void gcn_gemm_kernel(const GemmArgs p)
{
    using GG = Geom<TM>;
    constexpr int kTileM = GG::kTileM, kGemmThreads = GG::kThreads, kChunksA = GG::kChunksA, kStageBytes = GG::kStageBytes, kRing = GG::kRing,
                  kDmaPerWave = GG::kDmaPerWave, kWaves = GG::kWaves, NJ = GG::NJ, kWavesN = GG::kWavesN;
    static_assert(TM >= 128 || FL, "64-row tiles are built for the fused next-layer product only");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const unsigned long long rt_entry = p.stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;      // (diagnostics: the 100 MHz clock all XCDs share)
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid / kWavesN, wn = wid % kWavesN;
    // XCD-aware block -> tile map.  Consecutive workgroup ids go round-robin to the 8 XCDs, each with
    // its own L2; the row tiles of one graph all stream the same Bt operand, so they must be neighbours
    // on ONE XCD (then Bt comes from HBM once, not once per row tile): id % 8 labels the XCD, id / 8 walks
    // that XCD's graphs (xcd, xcd + 8, ...) tile by tile.
    const int per_graph = p.tiles_x * p.tiles_y;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    // Per-graph extents (instance graphs, compacted classes): most graphs leave their LAST row tiles idle, and with the tiles of a
    // graph on consecutive ids the idle ones fell at a fixed period of the dispatch order - measured with 64-row tiles of ~110-vertex
    // graphs (tiles 2 and 3 of every four idle): half of the active workgroups started a whole workgroup time late, the idle
    // workgroups having taken their turn of the shader engines.  Tile-major order inside the XCD: every graph's tile 0, then every
    // graph's tile 1, ... - the idle workgroups come last and the active ones are dealt over all CUs at once; a graph's tiles are still
    // on one XCD and still run at the same time (they share Bt through its L2).
    const int graphs_per_xcd = (p.batches + 7) >> 3;
    const int batch = p.tile_major ? (slot % graphs_per_xcd) * 8 + xcd : (slot / per_graph) * 8 + xcd;
    const int tile = p.tile_major ? slot / graphs_per_xcd : slot % per_graph;
    if (batch >= p.batches) return;                            // (whole workgroup, before any barrier)
    int tile_m_ = (tile / p.tiles_x) * kTileM;
    // FL, small graphs: when the batch's extent leaves row tiles idle (instance graphs: padded to 196 rows = 2 tiles, the
    // largest graph of a batch has ~125 vertices = 1 tile), idle tile t_real + t repeats the product and the LayerNorm of
    // tile t and takes the SECOND half of its fused-Linear epilogue, tile t only the first: the two workgroups share a CU
    // (each graph otherwise leaves half of it idle) and the serial epilogue of a tile is halved.
    int fl_first = 0, fl_last = 2;
    if constexpr (FL) {
        if (TM == 128 && p.m_extent && p.tiles_x == 1 && p.fl_twin) {
            const int t_real = (p.m_extent[(int64_t)batch * p.ext_stride] + kTileM - 1) / kTileM, ty = tile;
            if (t_real > 0 && p.tiles_y >= 2 * t_real) {
                if (ty < t_real) fl_last = 1;
                else if (ty < 2 * t_real) { tile_m_ = (ty - t_real) * kTileM; fl_first = 1; }
            }
        }
    }
    const int tile_m = tile_m_, tile_n = (tile % p.tiles_x) * kTileN;
    if (p.m_extent && tile_m >= p.m_extent[(int64_t)batch * p.ext_stride]) {      // a row tile past the largest graph of the batch (ext_stride 1: past this graph)
        // (the pooled partial sums are kept per 128 rows whatever the tile height: a 256-row tile owns two of them)
        if (p.pooled && TM >= 128) {
            const int part = (tile / p.tiles_x) * (kTileM / 128) + (tid >> 8), col = tile_n + (tid & 255);
            if (part * 128 < p.m && col < p.n) p.pooled[((int64_t)batch * p.pooled_parts + part) * p.n + col] = 0.0f;
        }
        if (p.zero_skipped && p.c && !p.accumulate) {              // (its rows of the fp32 result: zeros, whole lines)
            const int cols = min(kTileN, p.n - tile_n), rows = min(kTileM, p.m - tile_m);
            float *c0 = p.c + (int64_t)batch * p.c_batch_stride + (int64_t)tile_m * p.ldc + tile_n;
            if ((cols & 3) == 0 && (p.ldc & 3) == 0 && (reinterpret_cast<uintptr_t>(c0) & 15) == 0) {
                const int c4 = cols >> 2;
                for (int i = tid; i < rows * c4; i += kGemmThreads) *reinterpret_cast<float4 *>(c0 + (int64_t)(i / c4) * p.ldc + 4 * (i % c4)) = float4{0.0f, 0.0f, 0.0f, 0.0f};
            } else {
                for (int i = tid; i < rows * cols; i += kGemmThreads) c0[(int64_t)(i / cols) * p.ldc + i % cols] = 0.0f;
            }
        }
        return;
    }
    const int kb_count = p.k / kStageK;
    __shared__ int ids_s[GB ? kMaxGatherK : 1];
    if constexpr (GB) {
        for (int j = tid; j < p.k; j += kGemmThreads) {
            const int64_t id = j < p.ids_n ? p.ids[(int64_t)batch * p.ids_stride + j] : -1;
            ids_s[j] = (id >= 0 && id < p.tab_rows) ? (int)id : p.tab_rows;
        }
        __syncthreads();
    }

    // ---- LDS-DMA sources: wave w copies chunks 6w .. 6w+5 of a stage.  chunk c < 8: A row block c>>1,
    // plane c&1; c >= 8: the same for B.  A chunk is one contiguous 1 KiB block of the blocked plane.
    const _Float16 *src[kDmaPerWave];
#if SN_GEMM_SADDR
    const unsigned char *sbase[kDmaPerWave];
    unsigned voff[kDmaPerWave];
    auto plane_of_gather = [](int pb) { return pb >> 3; };
#endif
    unsigned nt_mask = 0;
#pragma unroll
    for (int j = 0; j < kDmaPerWave; ++j) {
        const int c = wid * kDmaPerWave + j;
        const bool is_b = c >= kChunksA;
        if (is_b ? p.nt_b : p.nt_a) nt_mask |= 1u << j;
        const int t = (is_b ? c - kChunksA : c) >> 1, plane = c & 1;
        const int rb_max = ((is_b ? p.n : p.m) - 1) >> 5;
        int rb = ((is_b ? tile_n : tile_m) >> 5) + t;
        rb = rb < rb_max ? rb : rb_max;                                    // clamp: rows past the end are discarded in the epilogue
        const _Float16 *base = is_b ? (plane ? p.b_lo : p.b_hi) + (int64_t)batch * p.b_batch_stride
                                    : (plane ? p.a_lo : p.a_hi) + (int64_t)batch * p.a_batch_stride;
        src[j] = base + (int64_t)rb * kb_count * kBlockElems + lane * 8;
#if SN_GEMM_SADDR
        // (round 6) the copies take their address as a wave-uniform base in SGPRs + a 32-bit per-lane byte offset: the form with a
        // 64-bit address per lane kept the CU's address unit busy for ~37 cycles per 1 KiB copy - the 27 B per cycle and CU every
        // ring of this kernel ran at, whatever its depth -, this one about half of that (the S1 screens' rings use it)
        sbase[j] = reinterpret_cast<const unsigned char *>(GB && is_b ? (plane_of_gather(c - kChunksA) ? p.tab_lo : p.tab_hi) : base);
        voff[j] = (unsigned)((int64_t)rb * kb_count * kBlockElems * 2) + (unsigned)lane * 16u;
#endif
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // GB: the table row (id) of this lane's node for each of the wave's B pieces of the NEXT stage to be issued, read from
    // LDS one stage ahead: the LDS round trip is then off the copy-issue path
    int next_id[kDmaPerWave];
    auto load_ids = [&](int t) {
#pragma unroll
        for (int j = 0; j < kDmaPerWave; ++j) {
            next_id[j] = 0;
            if (GB && wid * kDmaPerWave + j >= kChunksA) {
                const int pb = wid * kDmaPerWave + j - kChunksA, row = 2 * (pb & 7) + h;
                const int node = t * kStageK + row;
                next_id[j] = ids_s[node < p.k ? node : 0];
            }
        }
    };
    if constexpr (GB) load_ids(0);
    auto issue_stage = [&](int t) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (t % kRing) * kStageBytes + wid * kDmaPerWave * 1024);
#pragma unroll
        for (int j = 0; j < kDmaPerWave; ++j) {
            unsigned keep;
            if (SN_GEMM_ABLATE & 4) continue;
            if ((SN_GEMM_ABLATE & 8) && wid * kDmaPerWave + j < kChunksA) continue;
            if ((SN_GEMM_ABLATE & 16) && wid * kDmaPerWave + j >= kChunksA) continue;
            if (GB && wid * kDmaPerWave + j >= kChunksA) {            // (wave-uniform) piece = plane, node pair; lane = (node, 16-byte chunk)
                const int pb = wid * kDmaPerWave + j - kChunksA, row = 2 * (pb & 7) + h;
                const int id = next_id[j];
#if SN_GEMM_SADDR
                const unsigned goff = (unsigned)(id * p.tab_ld + tile_n + ((r ^ ((row & 3) << 2)) << 3)) * 2u;       // (bytes into the table plane: < 4 GiB)
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, %2\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(goff), "s"(sbase[j]), "s"(dst + j * 1024) : "memory");
#else
                const _Float16 *gsrc = ((pb >> 3) ? p.tab_lo : p.tab_hi) + (int64_t)id * p.tab_ld + tile_n + ((r ^ ((row & 3) << 2)) << 3);
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst + j * 1024) : "memory");
#endif
                continue;
            }
#if SN_GEMM_SADDR
            if ((nt_mask >> j) & 1u)                                  // wave-uniform
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, %2 nt\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff[j] + (unsigned)t * 1024u), "s"(sbase[j]), "s"(dst + j * 1024) : "memory");
            else
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, %2\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(voff[j] + (unsigned)t * 1024u), "s"(sbase[j]), "s"(dst + j * 1024) : "memory");
            continue;
#endif
            if ((nt_mask >> j) & 1u)                                  // wave-uniform
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                             "global_load_lds_dwordx4 %1, off nt\n\t"
                             "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src[j] + (int64_t)t * kBlockElems), "s"(dst + j * 1024) : "memory");
            else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src[j] + (int64_t)t * kBlockElems), "s"(dst + j * 1024) : "memory");
        }
        if constexpr (GB) load_ids(t + 1);                          // (stages are issued in order)
    };

    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    int n_stages = kb_count;
    if (p.k_extent) { const int lim = (p.k_extent[(int64_t)batch * p.ext_stride] + kStageK - 1) / kStageK; n_stages = lim < n_stages ? lim : n_stages; }
    unsigned long long t_begin = 0, t_wait = 0, t_issue = 0, t_loop_end = 0;
    if (p.stamps) t_begin = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int t = 0; t < kRing - 1; ++t)
        if (t < n_stages) issue_stage(t);
    half8 ah[2], al[2], bh[NJ], bl[NJ];
    // this wave's part of stage t has landed: at most the kRing - 2 younger stages are outstanding; then everybody's, and the
    // slot of stage t - 1 is free (every wave consumed its fragments of it in front of this barrier)
    auto wait_stage = [&](int t) {
        const int younger = n_stages - 1 - t;
        if (kRing >= 4 && younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kDmaPerWave) : "memory");
        else if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kDmaPerWave) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (!(SN_GEMM_ABLATE & 32)) __builtin_amdgcn_s_barrier();
    };
    auto read_fragments = [&](int t) {
        if (SN_GEMM_ABLATE & 2) {
            if (t == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) { ah[i] = half8{1, 2, 3, 4, 5, 6, 7, 8}; al[i] = ah[i]; }
#pragma unroll
                for (int j = 0; j < NJ; ++j) { bh[j] = half8{1, 2, 3, 4, 5, 6, 7, 8}; bl[j] = bh[j]; }
            }
            return;
        }
        const unsigned char *sa = smem + (t % kRing) * kStageBytes, *sb = sa + kChunksA * 1024;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mt = 2 * wm + i;
            ah[i] = *reinterpret_cast<const half8 *>(sa + (mt * 2 + 0) * 1024 + lane * 16);
            al[i] = *reinterpret_cast<const half8 *>(sa + (mt * 2 + 1) * 1024 + lane * 16);
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int nt = NJ * wn + j;
            if constexpr (GB) {
                // lane 4q + p of its 16-lane group addresses row (node) 8h + 4 half + q, features 32 nt + 16 rh + 4p .. + 3
                typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
                const int q = (lane & 15) >> 2, pp = lane & 3, rh = (lane >> 4) & 1;
                const unsigned char *base = sb + (8 * h + q) * 512 + ((4 * (nt ^ q) + 2 * rh + (pp >> 1)) << 4) + 8 * (pp & 1);
                fp16x4 v[4];
#pragma unroll
                for (int x = 0; x < 4; ++x)                                       // x = 2 plane + half
                    v[x] = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4 *)(base + (x >> 1) * 8192 + (x & 1) * 2048));
                // (two 8-byte results side by side = the 16-byte fragment: no element moves)
                bh[j] = __builtin_bit_cast(half8, __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7));
                bl[j] = __builtin_bit_cast(half8, __builtin_shufflevector(v[2], v[3], 0, 1, 2, 3, 4, 5, 6, 7));
            } else {
                bh[j] = *reinterpret_cast<const half8 *>(sb + (nt * 2 + 0) * 1024 + lane * 16);
                bl[j] = *reinterpret_cast<const half8 *>(sb + (nt * 2 + 1) * 1024 + lane * 16);
            }
        }
    };
    // The 24 MFMAs of a stage in the order every form of this loop keeps per accumulator: lo.hi, hi.lo, hi.hi
    auto mfma_lo_hi = [&]() {
        if (SN_GEMM_ABLATE & 1) { asm volatile("" :: "v"(al[0]), "v"(al[1]), "v"(bh[0]), "v"(bh[1]), "v"(bh[NJ - 2]), "v"(bh[NJ - 1])); return; }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
    };
    auto mfma_hi_lo = [&](int i) {
        if (SN_GEMM_ABLATE & 1) { asm volatile("" :: "v"(ah[i]), "v"(bl[0]), "v"(bl[1]), "v"(bl[NJ - 2]), "v"(bl[NJ - 1])); return; }
#pragma unroll
        for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
    };
    auto mfma_hi_hi = [&]() {
        if (SN_GEMM_ABLATE & 1) return;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    };
    // TM = 256: waves 4-7 - the second wave of every SIMD - run HALF A STAGE BEHIND (MI355X_MICROARCH "Two waves per SIMD", item 9):
    // behind the barrier of stage t they still hold the fragments of stage t - 1 and issue its last 12 MFMAs while waves 0-3 wait
    // for their fragment reads of stage t; they read their own fragments of stage t under waves 0-3's MFMAs.  The order of the
    // three products per accumulator is the same in both halves of the workgroup: results bit for bit those of the 128-row form.
    const bool late = TM == 256 && wid >= kWaves / 2 && p.stagger;
    if (!late) {
        for (int t = 0; t < n_stages; ++t) {
            unsigned long long ta = 0, tb = 0;
            if (p.stamps) ta = __builtin_amdgcn_s_memtime();
            wait_stage(t);
            if (p.stamps) tb = __builtin_amdgcn_s_memtime();
#if SN_GEMM_ISSUE_AT == 0
            if (t + kRing - 1 < n_stages) issue_stage(t + kRing - 1);
#endif
            if (p.stamps) { t_wait += tb - ta; t_issue += __builtin_amdgcn_s_memtime() - tb; }
            read_fragments(t);
            mfma_lo_hi();
#if SN_GEMM_ISSUE_AT == 1
            __builtin_amdgcn_sched_barrier(0);
            if (t + kRing - 1 < n_stages) issue_stage(t + kRing - 1);
            __builtin_amdgcn_sched_barrier(0);
#endif
            mfma_hi_lo(0);
            mfma_hi_lo(1);
#if SN_GEMM_ISSUE_AT == 2
            __builtin_amdgcn_sched_barrier(0);
            if (t + kRing - 1 < n_stages) issue_stage(t + kRing - 1);
            __builtin_amdgcn_sched_barrier(0);
#endif
            mfma_hi_hi();
        }
    } else {
        for (int t = 0; t < n_stages; ++t) {
            unsigned long long ta = 0, tb = 0;
            if (p.stamps) ta = __builtin_amdgcn_s_memtime();
            wait_stage(t);
            if (p.stamps) { tb = __builtin_amdgcn_s_memtime(); t_wait += tb - ta; }
            if (t > 0) { mfma_hi_lo(1); mfma_hi_hi(); }                       // (stage t - 1, from registers)
            __builtin_amdgcn_sched_barrier(0);
            if (t + kRing - 1 < n_stages) issue_stage(t + kRing - 1);
            read_fragments(t);
            __builtin_amdgcn_sched_barrier(0);
            mfma_lo_hi();
            mfma_hi_lo(0);
        }
        if (n_stages > 0) { mfma_hi_lo(1); mfma_hi_hi(); }
    }
    __builtin_amdgcn_s_barrier();                              // LDS is reused by the epilogue
    if (p.stamps) t_loop_end = __builtin_amdgcn_s_memtime();

    // ---- epilogue.  Accumulator layout: lane (r, h) of tile (i, j) holds column n = tile_n + (4wn+j)*32 + r,
    // rows m = tile_m + (2wm+i)*32 + (q & 3) + 8 (q >> 2) + 4 h for q = 0..15.
    float *red = reinterpret_cast<float *>(smem);              // [2 (wn)][128 rows]
    const int nv = p.rows_valid ? p.rows_valid[batch] : p.m;
    // (powers of two: the reciprocal and the products are exact; 1.0f when no scale is given - results bit-identical)
    const float acc_mul = 1.0f / ((p.a_scale ? *p.a_scale : 1.0f) * (p.b_scale ? *p.b_scale : 1.0f));
    const float out_mul = p.out_scale ? *p.out_scale : 1.0f;
    float bias[NJ], gam[NJ], bet[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = tile_n + (NJ * wn + j) * 32 + r;
        bias[j] = (p.bias && n < p.n) ? p.bias[n] : 0.0f;
        gam[j] = (LN && n < p.n) ? p.gamma[n] : 1.0f;
        bet[j] = (LN && n < p.n) ? p.beta[n] : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = tile_m + (2 * wm + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            const bool live = m < nv;
#pragma unroll
            for (int j = 0; j < NJ; ++j) acc[i][j][q] = live ? acc[i][j][q] * acc_mul + bias[j] : 0.0f;     // pad rows -> 0 (gnn.py:43-45)
        }
    if (LN) {
        // LayerNorm over the 256 columns of a row: 4 lane-local values x 32 lanes x the two wn waves.
        // Two passes (mean, then centred sum of squares); the row statistics live in LDS, not registers.
        float *red2 = red + kWavesN * kTileM;
        // (the row totals of the wn waves, added in a fixed order: two waves, or four - 64-row tiles - as two pairs)
        auto across_waves = [&](const float *part, int row) {
            if constexpr (kWavesN == 2) return part[row] + part[kTileM + row];
            else return (part[row] + part[kTileM + row]) + (part[2 * kTileM + row] + part[3 * kTileM + row]);
        };
        auto row_of = [&](int i, int q) { return (2 * wm + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h; };
        // Row totals over the 32 lanes of a half by recursive halving: at every level a lane keeps half of
        // its values and hands the other half to its partner (xor 16 via v_permlane16_swap, then the DPP
        // mirrors inside the 16-lane row), so 32 values cost 31 exchanges instead of 32 full reductions,
        // and lane l ends with the total of value index l & 31 = (i, q) = (l >> 4 & 1, l & 15).
        auto reduce32 = [&](float (&v)[32]) -> float {
#pragma unroll
            for (int k = 0; k < 16; ++k) {                    // partner lane ^ 16
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[k]), __float_as_uint(v[k + 16]), false, false);
                v[k] = __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
            }
            const bool b3 = (lane & 8) != 0, b2 = (lane & 4) != 0, b1 = (lane & 2) != 0, b0 = (lane & 1) != 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {                     // partner 15 - i (row_mirror): opposite bit 3
                const float give = b3 ? v[k] : v[k + 8], keep = b3 ? v[k + 8] : v[k];
                v[k] = keep + SN_DPP_F32(give, 0x140);
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {                     // partner 7 - i inside the half row: opposite bit 2
                const float give = b2 ? v[k] : v[k + 4], keep = b2 ? v[k + 4] : v[k];
                v[k] = keep + SN_DPP_F32(give, 0x141);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {                     // partner 3 - i inside the quad: opposite bit 1
                const float give = b1 ? v[k] : v[k + 2], keep = b1 ? v[k + 2] : v[k];
                v[k] = keep + SN_DPP_F32(give, 0x1B);
            }
            const float give = b0 ? v[0] : v[1], keep = b0 ? v[1] : v[0];
            return keep + SN_DPP_F32(give, 0xB1);             // partner i ^ 1
        };
        const int my_row = row_of((lane >> 4) & 1, lane & 15);          // the row whose total this lane ends with
        {
            float v[32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if constexpr (NJ == 4) v[i * 16 + q] = (acc[i][0][q] + acc[i][1][q]) + (acc[i][2][q] + acc[i][3][q]);
                    else v[i * 16 + q] = acc[i][0][q] + acc[i][NJ - 1][q];
                }
            red[wn * kTileM + my_row] = reduce32(v);
        }
        __syncthreads();
        {
            float v[32];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = row_of(i, q);
                    const float mean = across_waves(red, row) * (1.0f / 256.0f);      // n == 256: exact
                    float s2 = 0.0f;
#pragma unroll
                    for (int j = 0; j < NJ; ++j) {
                        acc[i][j][q] -= mean;
                        s2 = fmaf(acc[i][j][q], acc[i][j][q], s2);
                    }
                    v[i * 16 + q] = s2;
                }
            red2[wn * kTileM + my_row] = reduce32(v);
        }
        __syncthreads();
        float *rstd_row = red2 + kWavesN * kTileM;              // one correctly rounded 1/sqrt per row, not per lane
        if (tid < kTileM) rstd_row[tid] = 1.0f / sqrtf(across_waves(red2, tid) * (1.0f / 256.0f) + p.eps);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float rstd = rstd_row[row_of(i, q)];
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j][q] = acc[i][j][q] * rstd * gam[j] + bet[j];
            }
        __syncthreads();
    }
    if (p.relu) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = fmaxf(acc[i][j][q], 0.0f);
    }
    if constexpr (FL) {
        constexpr int kFragHalf = 528, kFragBlock = 1056;       // bytes: a 32-node x 8-slot half / a (node block, k-step, plane) block;
                                                                // the 16- and 32-byte paddings spread a wave's 8-byte stores over all banks
#if !SN_GEMM_FL_TRANSPOSED
        constexpr int kC8Stride = 32 * 8 + 8;                   // dwords (output staging, as in the plane stores below)
#endif
        constexpr int kPF = 2;                                  // k-steps of W2 fragments in flight (16 % kPF == 0)
        unsigned char *frag = smem;
#if !SN_GEMM_FL_TRANSPOSED
        unsigned *stg = reinterpret_cast<unsigned *>(smem + GG::kStgOff) + wid * (16 * kC8Stride);
#endif
        const int kb_out = p.cp_cols / kStageK;
        // wave -> (fg: its 64 output features, np: its pair of 32-node blocks of the half).  128-row tiles: four waves x all 64 nodes
        // of a half; 256-row tiles: a half is 128 nodes, waves 4-7 take the second pair of node blocks for the same features.
        const int fg = TM <= 128 ? wid : (wid & 3), np = TM <= 128 ? 0 : (wid >> 2);
        auto load_w2 = [&](int s2, half8 (&dst)[4]) {            // [2 ob + plane]: rows 64 fg + 32 ob .., slots 16 s2 ..
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const int64_t idx = ((int64_t)(fg * 2 + ob) * (kTileN / kStageK) + s2) * kBlockElems + lane * 8;
                dst[2 * ob] = *reinterpret_cast<const half8 *>(p.w2_hi + idx);
                dst[2 * ob + 1] = *reinterpret_cast<const half8 *>(p.w2_lo + idx);
            }
        };
        const float h_mul = p.h_scale ? *p.h_scale : 1.0f;
        const float u_mul = out_mul / ((p.w2_scale ? *p.w2_scale : 1.0f) * h_mul);       // Zt * out_scale from (W s_w) . (H s_h)
        unsigned long long t_fl0 = 0, t_fl1 = 0, t_fl2 = 0, t_fl3 = 0;
        if (p.stamps) t_fl0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < (TM >= 128 ? 2 : 1); ++i) {           // passes of 64 (128: 256-row tiles) nodes; a 64-row tile is one pass
            if (i < fl_first || i >= fl_last) continue;          // (workgroup-uniform: the other half belongs to the twin workgroup)
            half8 wq[kPF][4];
#pragma unroll
            for (int s2 = 0; s2 < kPF; ++s2) load_w2(s2, wq[s2]);       // in flight under the fragment stores and the barriers
            __syncthreads();                                     // LDS free: LayerNorm scratch / the staging of the half before
            if constexpr (TM >= 128) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int nd = (q & 3) + 8 * (q >> 2) + 4 * h;
                const bool keep = tile_m + (2 * wm + i) * 32 + nd < p.m;
                unsigned hw[4], lw[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    _Float16 hi, lo;
                    split2(keep ? acc[i][j][q] * h_mul : 0.0f, hi, lo);
                    hw[j] = (unsigned)__builtin_bit_cast(unsigned short, hi);
                    lw[j] = (unsigned)__builtin_bit_cast(unsigned short, lo);
                }
                unsigned char *dst = frag + (size_t)((wm * 16 + wn * 8 + (r >> 2)) * 2) * kFragBlock + ((r >> 1) & 1) * kFragHalf + nd * 16 + (r & 1) * 8;
                *reinterpret_cast<uint2 *>(dst) = uint2{hw[0] | (hw[1] << 16), hw[2] | (hw[3] << 16)};
                *reinterpret_cast<uint2 *>(dst + kFragBlock) = uint2{lw[0] | (lw[1] << 16), lw[2] | (lw[3] << 16)};
            }
            } else {
                // 64-row tile: both row blocks (ib) of the wave in this one pass; the wave holds TWO features per lane and block (j = 0, 1:
                // features (2 wn + j) * 32 + r) = the slots 2 (wn & 1) + j of the four-slot group the 128-row form's wave wn >> 1 would hold,
                // so the image - and the column order of W2 (GNN.prepare) - are those of the other forms: one 4-byte piece per (node, plane)
#pragma unroll
                for (int ib = 0; ib < 2; ++ib)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int nd = (q & 3) + 8 * (q >> 2) + 4 * h;
                        const bool keep = tile_m + ib * 32 + nd < p.m;
                        unsigned hw[2], lw[2];
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            _Float16 hi, lo;
                            split2(keep ? acc[ib][j][q] * h_mul : 0.0f, hi, lo);
                            hw[j] = (unsigned)__builtin_bit_cast(unsigned short, hi);
                            lw[j] = (unsigned)__builtin_bit_cast(unsigned short, lo);
                        }
                        unsigned char *dst = frag + (size_t)((ib * 16 + (wn >> 1) * 8 + (r >> 2)) * 2) * kFragBlock + ((r >> 1) & 1) * kFragHalf + nd * 16 + (r & 1) * 8 + (wn & 1) * 4;
                        *reinterpret_cast<unsigned *>(dst) = hw[0] | (hw[1] << 16);
                        *reinterpret_cast<unsigned *>(dst + kFragBlock) = lw[0] | (lw[1] << 16);
                    }
            }
            __syncthreads();
            if (p.stamps && i == 0) t_fl1 = __builtin_amdgcn_s_memtime();
            f32x16 u[2][2];
#pragma unroll
            for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) u[ob][nb][q] = 0.0f;
#pragma unroll 1
            for (int s0 = 0; s0 < kTileN / kStageK; s0 += kPF) {  // rolled (unrolled, hipcc keeps 16 x 4 fragment addresses and spills): kPF k-steps per trip
#pragma unroll
                for (int t = 0; t < kPF; ++t) {
                    const int s2 = s0 + t;
                    half8 bh[2], bl[2];
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const unsigned char *src = frag + (size_t)(((2 * np + nb) * 16 + s2) * 2) * kFragBlock + h * kFragHalf + r * 16;
                        bh[nb] = *reinterpret_cast<const half8 *>(src);
                        bl[nb] = *reinterpret_cast<const half8 *>(src + kFragBlock);
                    }
                    half8 (&w)[4] = wq[t];
#if SN_GEMM_FL_TRANSPOSED
                    // (round 6) the SAME three products with the operands' roles exchanged: rows = nodes (the H fragments), columns = output
                    // features (the W2 fragments) - u[ob][nb] then holds a feature per lane and nodes along its registers, which is
                    // how the planes of Zt2 want them (below): no transposition through LDS
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) u[ob][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nb], w[2 * ob + 1], u[ob][nb], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) u[ob][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[nb], w[2 * ob], u[ob][nb], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) u[ob][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[nb], w[2 * ob], u[ob][nb], 0, 0, 0);
#else
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) u[ob][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * ob + 1], bh[nb], u[ob][nb], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) u[ob][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * ob], bl[nb], u[ob][nb], 0, 0, 0);
#pragma unroll
                    for (int ob = 0; ob < 2; ++ob)
#pragma unroll
                        for (int nb = 0; nb < 2; ++nb) u[ob][nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w[2 * ob], bh[nb], u[ob][nb], 0, 0, 0);
#endif
                    if (s2 + kPF < kTileN / kStageK) load_w2(s2 + kPF, wq[t]);
                }
            }
            if (p.stamps && i == 0) t_fl2 = __builtin_amdgcn_s_memtime();
#if SN_GEMM_FL_TRANSPOSED
            if (p.stamps && i == 0) t_fl3 = t_fl2;
            // u[ob][nb]: lane (r, h) = output feature 64 fg + 32 ob + r, register q = node (2 (2 np + nb) + i) * 32 + (q & 3) + 8 (q >> 2) + 4 h
            // of the tile.  A plane piece is eight consecutive nodes of one feature: a lane holds four (q & 3) of each group of eight and
            // its partner lane ^ 32 the other four - one v_permlane32_swap per register pair (q, q + 4) hands each lane a whole piece:
            // lane (r, 0) the nodes 0-7 and 16-23 of the 32-node block, lane (r, 1) the nodes 8-15 and 24-31, i.e. k-half h of the
            // 16-node k-blocks 0 and 1 - and the wave's 64 pieces of a k-block are its 1 KiB in order (byte lane * 16).
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const int rb = fg * 2 + ob;                      // 32-row block of the [256, cp_cols] result
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    unsigned pk[16];
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        _Float16 hi, lo;
                        split2(u[ob][nb][q] * u_mul, hi, lo);
                        pk[q] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                    }
#pragma unroll
                    for (int g8 = 0; g8 < 2; ++g8)
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq) {
                            const auto sw = __builtin_amdgcn_permlane32_swap(pk[8 * g8 + qq], pk[8 * g8 + 4 + qq], false, false);
                            pk[8 * g8 + qq] = sw[0];
                            pk[8 * g8 + 4 + qq] = sw[1];
                        }
#pragma unroll
                    for (int pi = 0; pi < 2; ++pi) {             // 16-node k-block pi of the node block; this lane's k-half = h
                        const int kb = TM >= 128 ? (tile_m >> 4) + (2 * (2 * np + nb) + i) * 2 + pi : (tile_m >> 4) + nb * 2 + pi;
                        if (kb < kb_out) {
                            const unsigned *q8 = pk + 8 * pi;
                            uint4 ph, pl;
                            ph.x = __builtin_amdgcn_perm(q8[1], q8[0], 0x05040100u); pl.x = __builtin_amdgcn_perm(q8[1], q8[0], 0x07060302u);
                            ph.y = __builtin_amdgcn_perm(q8[3], q8[2], 0x05040100u); pl.y = __builtin_amdgcn_perm(q8[3], q8[2], 0x07060302u);
                            ph.z = __builtin_amdgcn_perm(q8[5], q8[4], 0x05040100u); pl.z = __builtin_amdgcn_perm(q8[5], q8[4], 0x07060302u);
                            ph.w = __builtin_amdgcn_perm(q8[7], q8[6], 0x05040100u); pl.w = __builtin_amdgcn_perm(q8[7], q8[6], 0x07060302u);
                            const int64_t o = (int64_t)batch * p.cp_batch_stride + ((int64_t)rb * kb_out + kb) * kBlockElems + lane * 8;
                            *reinterpret_cast<uint4 *>(p.c_hi + o) = ph;
                            *reinterpret_cast<uint4 *>(p.c_lo + o) = pl;
                        }
                    }
                }
            }
#else
            __syncthreads();                                     // every wave is done with the fragment image: the staging overlaps it
            if (p.stamps && i == 0) t_fl3 = __builtin_amdgcn_s_memtime();
            // u[ob][nb]: lane (r, h) = node (2 (2 np + nb) + i) * 32 + r of the tile, registers = output features 64 fg + 32 ob + row(q, h).
            // Planes want eight consecutive nodes per 16-byte piece: transposed through the wave's own staging as packed
            // (hi | lo << 16) dwords, [8-node group][row][node % 8] (the scheme of the plane stores below).
#pragma unroll
            for (int ob = 0; ob < 2; ++ob) {
                const int rb = fg * 2 + ob;                      // 32-row block of the [256, cp_cols] result
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
                        _Float16 hi, lo;
                        split2(u[ob][nb][q] * u_mul, hi, lo);
                        const unsigned packed = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                        stg[(nb * 4 + (r >> 3)) * kC8Stride + row * 8 + (r & 7)] = packed;
                    }
                // (same wave wrote what it reads: LDS operations of a wave complete in order)
#pragma unroll
                for (int it = 0; it < 4; ++it) {                 // (node block nb, 16-node k-block pi inside it); lane half h = 8-node half of the k-block
                    const int nb = it >> 1, pi = it & 1;
                    const int c8 = nb * 4 + 2 * pi + h;
                    const int kb = (tile_m >> 4) + (2 * (2 * np + nb) + i) * 2 + pi;
                    const uint4 lo4 = *reinterpret_cast<const uint4 *>(stg + c8 * kC8Stride + r * 8);
                    const uint4 hi4 = *reinterpret_cast<const uint4 *>(stg + c8 * kC8Stride + r * 8 + 4);
                    if (kb < kb_out) {
                        uint4 ph, pl;
                        ph.x = __builtin_amdgcn_perm(lo4.y, lo4.x, 0x05040100u); pl.x = __builtin_amdgcn_perm(lo4.y, lo4.x, 0x07060302u);
                        ph.y = __builtin_amdgcn_perm(lo4.w, lo4.z, 0x05040100u); pl.y = __builtin_amdgcn_perm(lo4.w, lo4.z, 0x07060302u);
                        ph.z = __builtin_amdgcn_perm(hi4.y, hi4.x, 0x05040100u); pl.z = __builtin_amdgcn_perm(hi4.y, hi4.x, 0x07060302u);
                        ph.w = __builtin_amdgcn_perm(hi4.w, hi4.z, 0x05040100u); pl.w = __builtin_amdgcn_perm(hi4.w, hi4.z, 0x07060302u);
                        const int64_t o = (int64_t)batch * p.cp_batch_stride + ((int64_t)rb * kb_out + kb) * kBlockElems + lane * 8;
                        *reinterpret_cast<uint4 *>(p.c_hi + o) = ph;
                        *reinterpret_cast<uint4 *>(p.c_lo + o) = pl;
                    }
                }
            }
#endif
        }
        if (p.stamps && lane == 0) {
            unsigned long long *st = p.stamps + ((size_t)blockIdx.x * kWaves + wid) * 16;
            st[0] = t_begin; st[1] = t_loop_end; st[2] = __builtin_amdgcn_s_memtime(); st[3] = t_fl0; st[4] = t_fl1; st[5] = t_fl2; st[6] = t_fl3;
            st[8] = rt_entry; st[9] = __builtin_amdgcn_s_memrealtime();
        }
        return;
    }
    if constexpr (!FL) {      // (the fused-product kernels returned above)
    // ---- stores (64-row tiles: the pooled product only, the host launches nothing else on them)
    if constexpr (TM >= 128) {
    if (p.c) {
        bool added = false;
        if constexpr (!LN && !GB) {
            if (p.accumulate) {
                // c += result: the old values of four rows (16 per lane) are ALL requested before the first sum is stored (the stores go
                // through the same pointer, so the compiler keeps every later load behind them: a chain of 128 load - add - store round
                // trips per lane ran the 424 MB read-modify-write of config [4]'s edge gradient at 1 TB/s: 620 us against 206 us for the
                // plain store)
                added = true;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        float old[4][4];
                        float *row = p.c + (int64_t)batch * p.c_batch_stride + (int64_t)(tile_m + (2 * wm + i) * 32 + 8 * q4 + 4 * h) * p.ldc + tile_n + 4 * wn * 32 + r;
                        const int m0 = tile_m + (2 * wm + i) * 32 + 8 * q4 + 4 * h, n0 = tile_n + 4 * wn * 32 + r;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                old[qq][j] = (m0 + qq < p.m && n0 + 32 * j < p.n) ? __builtin_nontemporal_load(row + (int64_t)qq * p.ldc + 32 * j) : 0.0f;
#pragma unroll
                        for (int qq = 0; qq < 4; ++qq)
#pragma unroll
                            for (int j = 0; j < 4; ++j)
                                if (m0 + qq < p.m && n0 + 32 * j < p.n) row[(int64_t)qq * p.ldc + 32 * j] = old[qq][j] + acc[i][j][4 * q4 + qq];
                    }
            }
        }
        // (round 6, measured and not kept: 16-byte stores behind a 4 x 4 in-quad DPP transpose - 32 global_store_dwordx4 per wave instead
        // of 128 global_store_dword - left the store tail where it was: 28.3 k against 28.5 k cycles on a 256-row tile, 600 against
        // 570 us for 1000 graphs x 512 x 256; the tail is not bound by the number of store instructions.  DESIGN 8)
        if (!added) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int m = tile_m + (2 * wm + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                    if (m >= p.m) continue;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int n = tile_n + (4 * wn + j) * 32 + r;
                        if (n < p.n) p.c[(int64_t)batch * p.c_batch_stride + (int64_t)m * p.ldc + n] = acc[i][j][q];
                    }
                }
        }
    }
    // Blocked hi/lo planes of the result (row m, k = n).  The accumulator holds a column per lane; the
    // planes want 8 consecutive k per 16-byte piece, so each wave transposes its 32 x 128 sub-tile through
    // its own LDS region as packed (hi | lo << 16) dwords, [k/8][row][k%8] with a padded k/8 stride
    // (conflict-free writes), and writes whole 1 KiB blocks: lanes 0-31 = rows of the k-half 0, 32-63 of half 1.
    if (p.c_hi) {
        constexpr int kC8Stride = 32 * 8 + 8;                                  // dwords
        unsigned *stg = reinterpret_cast<unsigned *>(smem + GG::kStgOff) + wid * (16 * kC8Stride);
        const int kb_out = p.cp_cols / kStageK;
        const int rb_count = (p.m + 31) >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int rb = (tile_m >> 5) + 2 * wm + i;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = tile_n + (4 * wn + j) * 32 + r;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = (q & 3) + 8 * (q >> 2) + 4 * h;
                    const bool keep = rb * 32 + row < p.m && n < p.n;
                    _Float16 hi, lo;
                    split2(keep ? acc[i][j][q] * out_mul : 0.0f, hi, lo);
                    const unsigned packed = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
                    stg[(j * 4 + (r >> 3)) * kC8Stride + row * 8 + (r & 7)] = packed;
                }
            }
            // (same wave wrote what it reads: LDS operations of a wave complete in order)
            if (rb < rb_count) {
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int c8 = 2 * it + h;                                 // k / 8 inside the wave's 128 columns
                    const int kb = (tile_n >> 4) + wn * 8 + it;
                    const uint4 lo4 = *reinterpret_cast<const uint4 *>(stg + c8 * kC8Stride + r * 8);
                    const uint4 hi4 = *reinterpret_cast<const uint4 *>(stg + c8 * kC8Stride + r * 8 + 4);
                    if (kb < kb_out) {
                        uint4 ph, pl;
                        ph.x = __builtin_amdgcn_perm(lo4.y, lo4.x, 0x05040100u); pl.x = __builtin_amdgcn_perm(lo4.y, lo4.x, 0x07060302u);
                        ph.y = __builtin_amdgcn_perm(lo4.w, lo4.z, 0x05040100u); pl.y = __builtin_amdgcn_perm(lo4.w, lo4.z, 0x07060302u);
                        ph.z = __builtin_amdgcn_perm(hi4.y, hi4.x, 0x05040100u); pl.z = __builtin_amdgcn_perm(hi4.y, hi4.x, 0x07060302u);
                        ph.w = __builtin_amdgcn_perm(hi4.w, hi4.z, 0x05040100u); pl.w = __builtin_amdgcn_perm(hi4.w, hi4.z, 0x07060302u);
                        const int64_t o = (int64_t)batch * p.cp_batch_stride + ((int64_t)rb * kb_out + kb) * kBlockElems + lane * 8;
                        *reinterpret_cast<uint4 *>(p.c_hi + o) = ph;
                        *reinterpret_cast<uint4 *>(p.c_lo + o) = pl;
                    }
                }
            }
        }
    }
    // ---- node-weighted pooling of the tile's rows (gnn.py:96: sum_i w_i H[i, :], the caller divides)
    if (p.pooled) {
        float part[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int m = tile_m + (2 * wm + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
                const float w = m < p.m ? p.pool_w[(int64_t)batch * p.pool_w_stride + m] : 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) part[j] = fmaf(w, acc[i][j][q], part[j]);
            }
        float *pr = reinterpret_cast<float *>(smem + GG::kPoolOff);     // [kWavesM (wm)][256 cols]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            part[j] += __shfl_xor(part[j], 32, SN_WAVE);
            if (h == 0) pr[wm * 256 + (4 * wn + j) * 32 + r] = part[j];
        }
        __syncthreads();
        {   // one partial row per 128 rows: pooled[batch][part][n] (summed in a fixed order by sn_pool_fc: deterministic; a 256-row
            // tile writes the two partial rows its rows would have given as two 128-row tiles - the same sums in the same order).
            // (Round 6, measured and not kept: one partial row per 64 rows - no LDS round here, and a layout a 64-row pooled product
            // could share - made sn_pool_fc read twice the rows: 12.3 against 8.5 us on the class side, 8.6 against 6.7 on the instance side.)
            const int half = tid >> 8, n = tile_n + (tid & 255), part = (tile / p.tiles_x) * (kTileM / 128) + half;
            if (n < p.n && part * 128 < p.m) p.pooled[((int64_t)batch * p.pooled_parts + part) * p.n + n] = pr[(2 * half) * 256 + (tid & 255)] + pr[(2 * half + 1) * 256 + (tid & 255)];
        }
    }
    }
    }
    if (p.stamps && lane == 0) {
        unsigned long long *st = p.stamps + ((size_t)blockIdx.x * kWaves + wid) * 16;
        st[0] = t_begin; st[1] = t_loop_end; st[2] = __builtin_amdgcn_s_memtime(); st[3] = t_wait; st[4] = t_issue;
        st[8] = rt_entry; st[9] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" int64_t sn_gcn_plane_elems(int rows, int k)
{
    if (rows <= 0 || k <= 0) return 0;
    return (int64_t)((rows + 31) / 32) * ((k + 15) / 16) * kBlockElems;
}

// grid of the adjacency producer: (workgroups per graph, pair_tiles).  Up to 4 tiles per side: one workgroup per tile pair.
static void adjacency_grid(int n, int G, unsigned &wgs, int &pair_tiles)
{
    const int t_full = (((n + 31) & ~31) + 63) / 64;
    if (t_full <= 4) { pair_tiles = t_full; wgs = (unsigned)(t_full * (t_full + 1) / 2); return; }
    // larger graphs: the tile pairs dealt over W workgroups per graph, ~3.5 workgroups per CU in all (at least what the old walk
    // launched, (T + 1) / 2: many graphs - 1000 classes - keep that; SN_ADJ_WALK=1: the old walk)
    static const int old_walk = getenv("SN_ADJ_WALK") ? atoi(getenv("SN_ADJ_WALK")) : 0;
    const int pairs = t_full * (t_full + 1) / 2, lo = (t_full + 1) / 2;
    int w = (int)((7 * (int64_t)sn_device_cus() / 2 + G - 1) / (G > 0 ? G : 1));
    w = w < lo ? lo : (w > pairs ? pairs : w);
    if (old_walk) { pair_tiles = 0; wgs = (unsigned)lo; }
    else { pair_tiles = -1; wgs = (unsigned)w; }
}

static int adjacency_graph_fast()
{
    static const int v = getenv("SN_ADJ_GRAPH_MAJOR") ? atoi(getenv("SN_ADJ_GRAPH_MAJOR")) : 1;
    return v;
}
static dim3 adjacency_dim(unsigned wgs, int G, int pair_tiles)
{
    return (pair_tiles > 0 && adjacency_graph_fast()) ? dim3((unsigned)G, wgs) : dim3(wgs, (unsigned)G);
}

extern "C" int sn_gcn_adjacency_planes(const float *edges, int G, int n, const int32_t *extent_dev, float scale, void *adj_hi, void *adj_lo, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes: bad G=%d n=%d", G, n);
    if (G == 0) return SN_OK;
    SN_REQUIRE(edges && adj_hi && adj_lo, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes: NULL pointer");
    SN_REQUIRE(scale > 0.0f && scale <= 65536.0f, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes: scale %g (a power of two in (0, 65536])", (double)scale);
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_adjacency_planes: G=%d > 65535", G);
    const int kb = (n + 15) / 16;
    unsigned tiles; int pair_tiles;
    adjacency_grid(n, G, tiles, pair_tiles);
    if (n % 4 == 0 && ((uintptr_t)edges & 15) == 0)
        hipLaunchKernelGGL(adjacency_planes_kernel<true>, adjacency_dim(tiles, G, pair_tiles), dim3(256), 0, (hipStream_t)stream, edges, n, kb,
                           sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, (const float *)nullptr, 0, extent_dev,
                           (const int32_t *)nullptr, pair_tiles, scale, (float *)nullptr, (const int32_t *)nullptr, 0, adjacency_graph_fast());
    else
        hipLaunchKernelGGL(adjacency_planes_kernel<false>, adjacency_dim(tiles, G, pair_tiles), dim3(256), 0, (hipStream_t)stream, edges, n, kb,
                           sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, (const float *)nullptr, 0, extent_dev,
                           (const int32_t *)nullptr, pair_tiles, scale, (float *)nullptr, (const int32_t *)nullptr, 0, adjacency_graph_fast());
    SN_CHECK_LAUNCH("sn_gcn_adjacency_planes");
    return SN_OK;
}

extern "C" int sn_gcn_adjacency_planes_masked(const float *edges, int G, int n, const int32_t *n_valid, const int32_t *extent_dev,
                                              float scale, void *adj_hi, void *adj_lo, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_masked: bad G=%d n=%d", G, n);
    if (G == 0) return SN_OK;
    SN_REQUIRE(edges && adj_hi && adj_lo, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_masked: NULL pointer");
    SN_REQUIRE(scale > 0.0f && scale <= 65536.0f, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_masked: scale %g (a power of two in (0, 65536])", (double)scale);
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_adjacency_planes_masked: G=%d > 65535", G);
    const int kb = (n + 15) / 16;
    unsigned tiles; int pair_tiles;
    adjacency_grid(n, G, tiles, pair_tiles);
    if (n % 4 == 0 && ((uintptr_t)edges & 15) == 0)
        hipLaunchKernelGGL(adjacency_planes_kernel<true>, adjacency_dim(tiles, G, pair_tiles), dim3(256), 0, (hipStream_t)stream, edges, n, kb,
                           sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, (const float *)nullptr, 0, extent_dev, n_valid, pair_tiles, scale, (float *)nullptr, (const int32_t *)nullptr, 0, adjacency_graph_fast());
    else
        hipLaunchKernelGGL(adjacency_planes_kernel<false>, adjacency_dim(tiles, G, pair_tiles), dim3(256), 0, (hipStream_t)stream, edges, n, kb,
                           sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, (const float *)nullptr, 0, extent_dev, n_valid, pair_tiles, scale, (float *)nullptr, (const int32_t *)nullptr, 0, adjacency_graph_fast());
    SN_CHECK_LAUNCH("sn_gcn_adjacency_planes_masked");
    return SN_OK;
}

extern "C" int sn_gcn_adjacency_planes_per_graph(const float *edges, int G, int n, const int32_t *n_valid, float scale, void *adj_hi,
                                                 void *adj_lo, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_per_graph: bad G=%d n=%d", G, n);
    if (G == 0) return SN_OK;
    SN_REQUIRE(edges && adj_hi && adj_lo && n_valid, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_per_graph: NULL pointer");
    SN_REQUIRE(scale > 0.0f && scale <= 65536.0f, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_per_graph: scale %g (a power of two in (0, 65536])", (double)scale);
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_adjacency_planes_per_graph: G=%d > 65535", G);
    const int kb = (n + 15) / 16;
    unsigned tiles; int pair_tiles;
    adjacency_grid(n, G, tiles, pair_tiles);
    // (the vertex counts serve as the extents too: one per graph)
    if (n % 4 == 0 && ((uintptr_t)edges & 15) == 0)
        hipLaunchKernelGGL(adjacency_planes_kernel<true>, adjacency_dim(tiles, G, pair_tiles), dim3(256), 0, (hipStream_t)stream, edges, n, kb,
                           sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, (const float *)nullptr, 0, n_valid, n_valid, pair_tiles, scale, (float *)nullptr, (const int32_t *)nullptr, 1, adjacency_graph_fast());
    else
        hipLaunchKernelGGL(adjacency_planes_kernel<false>, adjacency_dim(tiles, G, pair_tiles), dim3(256), 0, (hipStream_t)stream, edges, n, kb,
                           sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, (const float *)nullptr, 0, n_valid, n_valid, pair_tiles, scale, (float *)nullptr, (const int32_t *)nullptr, 1, adjacency_graph_fast());
    SN_CHECK_LAUNCH("sn_gcn_adjacency_planes_per_graph");
    return SN_OK;
}

extern "C" int sn_gcn_atlas_adjacency_planes(const float *pruned_edge_weights, const float *row_sum, int K, int n, int remove_self_loop,
                                             float scale, void *adj_hi, void *adj_lo, float *class_edges_out, void *stream)
{
    SN_REQUIRE(K >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_gcn_atlas_adjacency_planes: bad K=%d n=%d", K, n);
    if (K == 0) return SN_OK;
    SN_REQUIRE(pruned_edge_weights && row_sum && adj_hi && adj_lo, SN_ERR_BAD_ARG, "sn_gcn_atlas_adjacency_planes: NULL pointer");
    SN_REQUIRE(scale > 0.0f && scale <= 65536.0f, SN_ERR_BAD_ARG, "sn_gcn_atlas_adjacency_planes: scale %g (a power of two in (0, 65536])", (double)scale);
    SN_REQUIRE(K <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_atlas_adjacency_planes: K=%d > 65535", K);
    const int kb = (n + 15) / 16;
    unsigned tiles; int pair_tiles;
    adjacency_grid(n, K, tiles, pair_tiles);
    if (n % 4 == 0 && (((uintptr_t)pruned_edge_weights | (uintptr_t)class_edges_out) & 15) == 0)
        hipLaunchKernelGGL(adjacency_planes_kernel<true>, adjacency_dim(tiles, K, pair_tiles), dim3(256), 0, (hipStream_t)stream, pruned_edge_weights, n,
                           kb, sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, row_sum, remove_self_loop, (const int32_t *)nullptr,
                           (const int32_t *)nullptr, pair_tiles, scale, class_edges_out, (const int32_t *)nullptr, 0, adjacency_graph_fast());
    else
        hipLaunchKernelGGL(adjacency_planes_kernel<false>, adjacency_dim(tiles, K, pair_tiles), dim3(256), 0, (hipStream_t)stream, pruned_edge_weights, n,
                           kb, sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, row_sum, remove_self_loop, (const int32_t *)nullptr,
                           (const int32_t *)nullptr, pair_tiles, scale, class_edges_out, (const int32_t *)nullptr, 0, adjacency_graph_fast());
    SN_CHECK_LAUNCH("sn_gcn_atlas_adjacency_planes");
    return SN_OK;
}

/* The same for a PRUNED atlas, compacted: vertex a of class k's operand is vertex perm[k][a] of the stored graph (the kept
 * vertices first), only its n_kept[k] x n_kept[k] corner (+ identity) is produced - rounded up to the 32 x 16 blocks, the
 * rest of a block zero - and sn_gcn_gemm consumes it with per-graph extents (extent_stride 1). */
extern "C" int sn_gcn_atlas_adjacency_planes_compact(const float *pruned_edge_weights, const float *row_sum, int K, int n, int remove_self_loop,
                                                     float scale, const int32_t *perm, const int32_t *n_kept, void *adj_hi, void *adj_lo,
                                                     void *stream)
{
    SN_REQUIRE(K >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_gcn_atlas_adjacency_planes_compact: bad K=%d n=%d", K, n);
    if (K == 0) return SN_OK;
    SN_REQUIRE(pruned_edge_weights && row_sum && adj_hi && adj_lo && perm && n_kept, SN_ERR_BAD_ARG, "sn_gcn_atlas_adjacency_planes_compact: NULL pointer");
    SN_REQUIRE(n <= kMaxPerm, SN_ERR_UNSUPPORTED, "sn_gcn_atlas_adjacency_planes_compact: n=%d > %d", n, kMaxPerm);
    SN_REQUIRE(scale > 0.0f && scale <= 65536.0f, SN_ERR_BAD_ARG, "sn_gcn_atlas_adjacency_planes_compact: scale %g (a power of two in (0, 65536])", (double)scale);
    SN_REQUIRE(K <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_atlas_adjacency_planes_compact: K=%d > 65535", K);
    const int kb = (n + 15) / 16;
    unsigned tiles; int pair_tiles;
    adjacency_grid(n, K, tiles, pair_tiles);
    hipLaunchKernelGGL(adjacency_planes_kernel<false>, adjacency_dim(tiles, K, pair_tiles), dim3(256), 0, (hipStream_t)stream, pruned_edge_weights, n,
                       kb, sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, row_sum, remove_self_loop, n_kept,
                       n_kept, pair_tiles, scale, (float *)nullptr, perm, 1, adjacency_graph_fast());
    SN_CHECK_LAUNCH("sn_gcn_atlas_adjacency_planes_compact");
    return SN_OK;
}

extern "C" int sn_gcn_adjacency_planes_compact(const float *edges, int G, int n, const int32_t *perm, const int32_t *n_kept, float scale,
                                               void *adj_hi, void *adj_lo, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_compact: bad G=%d n=%d", G, n);
    if (G == 0) return SN_OK;
    SN_REQUIRE(edges && adj_hi && adj_lo && perm && n_kept, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_compact: NULL pointer");
    SN_REQUIRE(n <= kMaxPerm, SN_ERR_UNSUPPORTED, "sn_gcn_adjacency_planes_compact: n=%d > %d", n, kMaxPerm);
    SN_REQUIRE(scale > 0.0f && scale <= 65536.0f, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes_compact: scale %g (a power of two in (0, 65536])", (double)scale);
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_adjacency_planes_compact: G=%d > 65535", G);
    const int kb = (n + 15) / 16;
    unsigned tiles; int pair_tiles;
    adjacency_grid(n, G, tiles, pair_tiles);
    hipLaunchKernelGGL(adjacency_planes_kernel<false>, adjacency_dim(tiles, G, pair_tiles), dim3(256), 0, (hipStream_t)stream, edges, n, kb,
                       sn_gcn_plane_elems(n, n), (_Float16 *)adj_hi, (_Float16 *)adj_lo, (const float *)nullptr, 0, n_kept, n_kept, pair_tiles,
                       scale, (float *)nullptr, perm, 1, adjacency_graph_fast());
    SN_CHECK_LAUNCH("sn_gcn_adjacency_planes_compact");
    return SN_OK;
}

extern "C" int sn_gcn_gather_planes(const float *table, int rows_table, const int64_t *ids, int G, int n, int E,
                                    const int32_t *extent_dev, const float *scale_dev, void *out_hi, void *out_lo, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0 && rows_table > 0, SN_ERR_BAD_ARG, "sn_gcn_gather_planes: bad G=%d n=%d E=%d", G, n, E);
    if (G == 0) return SN_OK;
    SN_REQUIRE(table && ids && out_hi && out_lo, SN_ERR_BAD_ARG, "sn_gcn_gather_planes: NULL pointer");
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_gather_planes: G=%d > 65535", G);
    const int kb = (n + 15) / 16;
    hipLaunchKernelGGL(gather_planes_kernel, dim3((unsigned)((kb * 16 + 63) / 64), (unsigned)G), dim3(256), 0, (hipStream_t)stream, table,
                       rows_table, ids, n, kb, E, sn_gcn_plane_elems(E, n), (_Float16 *)out_hi, (_Float16 *)out_lo, extent_dev, scale_dev);
    SN_CHECK_LAUNCH("sn_gcn_gather_planes");
    return SN_OK;
}

static int split_planes_launch(bool transposed, const float *x, int batches, int rows, int cols, int64_t ld, int64_t batch_stride,
                               const float *scale_dev, void *out_hi, void *out_lo, void *stream, const char *name, const int32_t *node_extent = nullptr)
{
    SN_REQUIRE(batches >= 0 && rows > 0 && cols > 0 && ld >= cols, SN_ERR_BAD_ARG, "%s: bad shape", name);
    if (batches == 0) return SN_OK;
    SN_REQUIRE(x && out_hi && out_lo, SN_ERR_BAD_ARG, "%s: NULL pointer", name);
    const int op_rows = transposed ? cols : rows, op_k = transposed ? rows : cols;
    const int kb = (op_k + 15) / 16, kpad = kb * 16;
    const int kc = kpad < kSplitKc ? kpad : kSplitKc;
    const int row_blocks = (op_rows + 31) / 32, k_chunks = (kpad + kc - 1) / kc;
    const int64_t tiles = (int64_t)batches * row_blocks * k_chunks, slots = 4 * (int64_t)sn_device_cus();
    const size_t lds = (size_t)32 * (kc + kSplitPad) * sizeof(float);
    const void *fn = transposed ? (const void *)split_planes_kernel<true> : (const void *)split_planes_kernel<false>;
    if (int rc = sn_ensure_dynamic_lds(fn, lds, name)) return rc;
    const dim3 grid((unsigned)(tiles < slots ? tiles : slots));
    if (transposed)
        hipLaunchKernelGGL(split_planes_kernel<true>, grid, dim3(kSplitThreads), lds, (hipStream_t)stream, x, rows, cols, ld, batch_stride, kb,
                           sn_gcn_plane_elems(op_rows, op_k), (_Float16 *)out_hi, (_Float16 *)out_lo, scale_dev, kc, row_blocks, k_chunks, tiles, node_extent);
    else
        hipLaunchKernelGGL(split_planes_kernel<false>, grid, dim3(kSplitThreads), lds, (hipStream_t)stream, x, rows, cols, ld, batch_stride, kb,
                           sn_gcn_plane_elems(op_rows, op_k), (_Float16 *)out_hi, (_Float16 *)out_lo, scale_dev, kc, row_blocks, k_chunks, tiles, node_extent);
    return SN_OK;
}

extern "C" int sn_split_planes_nodes(const float *x, int G, int n, int E, const float *scale_dev, const int32_t *node_extent, int transposed,
                                     void *out_hi, void *out_lo, void *stream)
{
    SN_REQUIRE(node_extent, SN_ERR_BAD_ARG, "sn_split_planes_nodes: NULL extents");
    if (int rc = split_planes_launch(transposed != 0, x, G, n, E, E, (int64_t)n * E, scale_dev, out_hi, out_lo, stream, "sn_split_planes_nodes", node_extent)) return rc;
    SN_CHECK_LAUNCH("sn_split_planes_nodes");
    return SN_OK;
}

extern "C" int sn_split_planes(const float *x, int batches, int rows, int cols, int64_t ld, int64_t batch_stride,
                               const float *scale_dev, void *out_hi, void *out_lo, void *stream)
{
    if (int rc = split_planes_launch(false, x, batches, rows, cols, ld, batch_stride, scale_dev, out_hi, out_lo, stream, "sn_split_planes")) return rc;
    SN_CHECK_LAUNCH("sn_split_planes");
    return SN_OK;
}

extern "C" int sn_split_planes_transposed(const float *x, int batches, int rows, int cols, int64_t ld, int64_t batch_stride,
                                          const float *scale_dev, void *out_hi, void *out_lo, void *stream)
{
    if (int rc = split_planes_launch(true, x, batches, rows, cols, ld, batch_stride, scale_dev, out_hi, out_lo, stream, "sn_split_planes_transposed")) return rc;
    SN_CHECK_LAUNCH("sn_split_planes_transposed");
    return SN_OK;
}

extern "C" int sn_layernorm_split_planes(const float *x, int G, int n, int E, const int32_t *n_valid, const float *gamma,
                                         const float *beta, float eps, int relu, const float *scale_dev, void *out_hi,
                                         void *out_lo, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0, SN_ERR_BAD_ARG, "sn_layernorm_split_planes: bad G=%d n=%d E=%d", G, n, E);
    if (G == 0) return SN_OK;
    SN_REQUIRE(x && gamma && beta && out_hi && out_lo, SN_ERR_BAD_ARG, "sn_layernorm_split_planes: NULL pointer");
    SN_REQUIRE(E % 16 == 0 && E <= SN_WAVE * SN_LN_MAX, SN_ERR_UNSUPPORTED, "sn_layernorm_split_planes: E=%d must be a multiple of 16, <= %d", E,
               SN_WAVE * SN_LN_MAX);
    const size_t lds = (size_t)32 * (E + kLnSplitPad) * sizeof(float);
    if (int rc = sn_ensure_dynamic_lds((const void *)layernorm_split_planes_kernel, lds, "sn_layernorm_split_planes")) return rc;
    const int row_blocks = (n + 31) / 32;
    const int64_t tiles = (int64_t)G * row_blocks, slots = 2 * (int64_t)sn_device_cus();
    hipLaunchKernelGGL(layernorm_split_planes_kernel, dim3((unsigned)(tiles < slots ? tiles : slots)), dim3(kLnSplitThreads), lds,
                       (hipStream_t)stream, x, n, E, n_valid, gamma, beta, eps, relu, E / 16, sn_gcn_plane_elems(n, E),
                       (_Float16 *)out_hi, (_Float16 *)out_lo, scale_dev, row_blocks, tiles);
    SN_CHECK_LAUNCH("sn_layernorm_split_planes");
    return SN_OK;
}

/* tile form of sn_gcn_gemm: see sn_debug_set_gemm_tile */
static int g_gemm_tile = -1, g_gemm_stagger = -1;
static int gemm_tile_setting()
{
    if (g_gemm_tile < 0) { const char *e = getenv("SN_GEMM_TM"); const int v = e ? atoi(e) : 0; g_gemm_tile = (v == 64 || v == 128 || v == 256) ? v : 0; }
    return g_gemm_tile;
}
static int gemm_stagger_setting()
{
    if (g_gemm_stagger < 0) { const char *e = getenv("SN_GEMM_STAGGER"); g_gemm_stagger = (e && atoi(e) == 0) ? 0 : 1; }
    return g_gemm_stagger;
}
extern "C" void sn_debug_set_gemm_tile(int tile_rows, int stagger)
{
    if (tile_rows >= 0) g_gemm_tile = (tile_rows == 64 || tile_rows == 128 || tile_rows == 256) ? tile_rows : 0;
    if (stagger >= 0) g_gemm_stagger = stagger ? 1 : 0;
}

/* diagnostics: device buffer of 16 x u64 per wave of the GEMM kernel (NULL = off) */
extern "C" void sn_debug_set_gemm_stamps(void *device_buffer) { g_gemm_stamps = (unsigned long long *)device_buffer; }

extern "C" int sn_gcn_gemm(const sn_gemm_args *u, void *stream)
{
    SN_REQUIRE(u, SN_ERR_BAD_ARG, "sn_gcn_gemm: NULL args");
    SN_REQUIRE(u->struct_size == sizeof(sn_gemm_args), SN_ERR_BAD_ARG, "sn_gcn_gemm: sn_gemm_args.struct_size=%u, this library (ABI %d) expects %zu",
               u->struct_size, sn_abi_version(), sizeof(sn_gemm_args));
    SN_REQUIRE(u->batches >= 0 && u->m > 0 && u->n > 0 && u->k > 0, SN_ERR_BAD_ARG, "sn_gcn_gemm: bad shape m=%d n=%d k=%d batches=%d",
               u->m, u->n, u->k, u->batches);
    if (u->batches == 0) return SN_OK;
    const bool gathered = u->b_table_hi != nullptr;
    SN_REQUIRE(u->a_hi && u->a_lo && (gathered || (u->b_hi && u->b_lo)), SN_ERR_BAD_ARG, "sn_gcn_gemm: NULL operand plane");
    if (gathered) {
        SN_REQUIRE(u->b_table_lo && u->b_ids && u->b_table_rows > 0 && u->b_ids_n >= 0, SN_ERR_BAD_ARG, "sn_gcn_gemm: incomplete gathered-B arguments");
        SN_REQUIRE(u->n % kTileN == 0 && (u->layernorm ? u->n == kTileN : !u->next_w_hi) && u->k <= kMaxGatherK, SN_ERR_UNSUPPORTED,
                   "sn_gcn_gemm: gathered B needs n a multiple of 256 (== 256 with the LayerNorm epilogue) and k <= %d (got n=%d k=%d)", kMaxGatherK, u->n, u->k);
        SN_REQUIRE(((uintptr_t)u->b_table_hi | (uintptr_t)u->b_table_lo) % 16 == 0, SN_ERR_BAD_ARG, "sn_gcn_gemm: table planes must be 16-byte aligned");
    }
    SN_REQUIRE(u->k % kStageK == 0, SN_ERR_BAD_ARG, "sn_gcn_gemm: k=%d must be a multiple of 16 (the planes' padded k)", u->k);
    SN_REQUIRE(((uintptr_t)u->a_hi | (uintptr_t)u->a_lo | (gathered ? 0 : ((uintptr_t)u->b_hi | (uintptr_t)u->b_lo))) % 16 == 0 &&
                   u->a_batch_stride % 8 == 0 && (gathered || u->b_batch_stride % 8 == 0), SN_ERR_BAD_ARG, "sn_gcn_gemm: planes must be 16-byte aligned");
    SN_REQUIRE(u->c || u->c_hi || u->pooled, SN_ERR_BAD_ARG, "sn_gcn_gemm: no output requested");
    const bool fused2 = u->next_w_hi != nullptr;
    if (fused2) {
        SN_REQUIRE(u->next_w_lo && u->layernorm && u->n == kTileN && u->c_hi && u->c_lo && !u->c && !u->pooled, SN_ERR_UNSUPPORTED,
                   "sn_gcn_gemm: the fused next-layer product needs n == 256, the LayerNorm epilogue and output planes only");
        SN_REQUIRE(u->cp_cols >= u->m && u->cp_cols % kStageK == 0, SN_ERR_BAD_ARG, "sn_gcn_gemm: bad output planes (fused next-layer product: [256, cp_cols >= m])");
        SN_REQUIRE(((uintptr_t)u->next_w_hi | (uintptr_t)u->next_w_lo) % 16 == 0, SN_ERR_BAD_ARG, "sn_gcn_gemm: next_w planes must be 16-byte aligned");
    }
    SN_REQUIRE(fused2 || !u->c_hi || (u->c_lo && u->cp_cols >= u->n && u->cp_cols % kStageK == 0), SN_ERR_BAD_ARG, "sn_gcn_gemm: bad output planes");
    SN_REQUIRE(!u->c || u->ldc >= u->n, SN_ERR_BAD_ARG, "sn_gcn_gemm: ldc=%d < n=%d", u->ldc, u->n);
    SN_REQUIRE(!u->layernorm || (u->n == kTileN && u->gamma && u->beta), SN_ERR_UNSUPPORTED,
               "sn_gcn_gemm: the LayerNorm epilogue needs n == 256 (got %d) and gamma/beta", u->n);
    SN_REQUIRE(!u->pooled || u->pool_w, SN_ERR_BAD_ARG, "sn_gcn_gemm: pooling without weights");
    SN_REQUIRE(!u->accumulate || (u->c && !u->c_hi && !u->pooled && !u->bias && !u->layernorm && !u->relu && !u->rows_valid && !gathered), SN_ERR_UNSUPPORTED,
               "sn_gcn_gemm: accumulate is for the plain fp32 product only");
    SN_REQUIRE(u->batches <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_gemm: batches=%d > 65535", u->batches);
    GemmArgs a;
    a.a_hi = (const _Float16 *)u->a_hi; a.a_lo = (const _Float16 *)u->a_lo; a.a_batch_stride = u->a_batch_stride;
    a.b_hi = (const _Float16 *)u->b_hi; a.b_lo = (const _Float16 *)u->b_lo; a.b_batch_stride = u->b_batch_stride;
    a.m = u->m; a.n = u->n; a.k = u->k;
    a.c = u->c; a.c_batch_stride = u->c_batch_stride; a.ldc = u->ldc;
    a.c_hi = (_Float16 *)u->c_hi; a.c_lo = (_Float16 *)u->c_lo; a.cp_batch_stride = u->cp_batch_stride; a.cp_cols = u->cp_cols;
    a.bias = u->bias; a.gamma = u->gamma; a.beta = u->beta; a.eps = u->eps; a.relu = u->relu;
    a.rows_valid = u->rows_valid; a.pool_w = u->pool_w; a.pool_w_stride = u->pool_w_stride; a.pooled = u->pooled;
    a.stamps = g_gemm_stamps;
    a.m_extent = u->m_extent; a.k_extent = u->k_extent; a.ext_stride = u->extent_stride != 0 ? 1 : 0;
    a.accumulate = u->accumulate != 0 ? 1 : 0;
    a.zero_skipped = u->zero_skipped != 0 ? 1 : 0;
    a.tab_hi = (const _Float16 *)u->b_table_hi; a.tab_lo = (const _Float16 *)u->b_table_lo;
    a.ids = u->b_ids; a.ids_stride = u->b_ids_stride; a.ids_n = u->b_ids_n; a.tab_rows = u->b_table_rows; a.tab_ld = u->n;
    a.w2_hi = (const _Float16 *)u->next_w_hi; a.w2_lo = (const _Float16 *)u->next_w_lo;
    a.a_scale = u->a_scale; a.b_scale = u->b_scale; a.out_scale = u->out_scale; a.w2_scale = u->next_w_scale; a.h_scale = u->next_h_scale;
    {
        static const int twin = getenv("SN_GEMM_FL_TWIN") ? atoi(getenv("SN_GEMM_FL_TWIN")) : 1;
        a.fl_twin = twin;
    }
    {   // A per-graph A operand (the adjacency) is read once, by the one workgroup that owns its row tile: its copies carry
        // the nt hint, so it does not displace what the other workgroups re-read (Bt of the graph, the atlas, the tokens of
        // the steps in flight): +1 % on the bench step.  The once-read B of the transposed Linear (A shared) gains nothing.
        // SN_GEMM_NT: bit 0 = A when per-graph (default), bit 1 = B when A is shared.
        static const int nt = getenv("SN_GEMM_NT") ? atoi(getenv("SN_GEMM_NT")) : 1;
        a.nt_a = (nt & 1) && u->a_batch_stride != 0;
        a.nt_b = (nt & 2) && u->a_batch_stride == 0;
    }
    const int cols = (u->c_hi && !fused2 && u->cp_cols > u->n) ? u->cp_cols : u->n;       // zero-filled plane columns need a tile too
    // Tile height (round 6): 256 rows - one 8-wave workgroup per CU - for graphs of at least 256 rows that share one extent (the class
    // graphs of the IR-Atlas, config [3]'s 500-vertex classes); 128 rows - two 4-wave workgroups per CU - for small graphs and for
    // batches with per-graph extents (instance graphs of ~110 vertices, compacted classes: a graph then costs whole tiles of its own
    // height).  SN_GEMM_TM=128 / 256 forces one form (256 needs m >= 129 to make sense; any m is correct).
    // 64-row tiles: the fused product (layer 1 + the next layer's Linear) of graphs with their own extents - the instance graphs of a
    // batch, the compacted classes of a pruned atlas; SN_GEMM_TM=64 forces it for every fused product, 128 / 256 switch it off.
    const int tm_set = gemm_tile_setting();
    // (the pooled LayerNorm product of such graphs was measured on 64-row tiles too: the same 14.6 us for 512 workgroups as for the 256
    // of the 128-row form - twice the CU time, and 8 us per step worse with four steps in flight; it stays on 128 rows)
    const bool small = fused2 && (tm_set == 64 || (tm_set == 0 && u->m_extent && u->extent_stride != 0));
    const int tile_m = small ? 64 : (tm_set == 256 || (tm_set != 128 && tm_set != 64 && u->m >= 256 && !(u->m_extent && u->extent_stride != 0))) ? 256 : 128;
    const bool tall = tile_m == 256;
    a.stagger = gemm_stagger_setting();
    {
        static const int tmaj = getenv("SN_GEMM_TILE_MAJOR") ? atoi(getenv("SN_GEMM_TILE_MAJOR")) : 1;
        a.tile_major = (tmaj && u->m_extent && u->extent_stride != 0) ? 1 : 0;
    }
    a.batches = u->batches; a.tiles_x = (cols + kTileN - 1) / kTileN; a.tiles_y = (u->m + tile_m - 1) / tile_m;
    const int parts = (u->m + 127) / 128;                                                  // pooled partial rows: one per 128 rows, whatever the tile
    SN_REQUIRE(u->pooled_parts == 0 || u->pooled_parts >= parts, SN_ERR_BAD_ARG, "sn_gcn_gemm: pooled_parts=%d < %d row tiles", u->pooled_parts, parts);
    a.pooled_parts = u->pooled_parts > 0 ? u->pooled_parts : parts;
    const int64_t n_blocks = (int64_t)8 * ((u->batches + 7) / 8) * a.tiles_x * a.tiles_y;
    SN_REQUIRE(n_blocks <= 0x7fffffff, SN_ERR_UNSUPPORTED, "sn_gcn_gemm: grid too large");
    const dim3 grid((unsigned)n_blocks);
    hipStream_t st = (hipStream_t)stream;
    int rc = SN_OK;
    auto launch = [&](auto kernel, int threads, size_t lds) {
        if ((rc = sn_ensure_dynamic_lds((const void *)kernel, lds, "sn_gcn_gemm"))) return;
        static const bool print_occupancy = getenv("SN_GEMM_OCC") != nullptr;          // (diagnostics: tools/time_gemm_fused.py)
        if (print_occupancy) {
            int nb = -1; hipFuncAttributes fa{};
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, threads, lds);
            (void)hipFuncGetAttributes(&fa, (const void *)kernel);
            fprintf(stderr, "sn_gcn_gemm: tile %d threads %d dynamic LDS %zu static LDS %zu regs %d -> %d workgroups per CU (runtime)\n", tile_m, threads, lds, fa.sharedSizeBytes, fa.numRegs, nb);
        }
        sn_prof_start(4, st);
        hipLaunchKernelGGL(kernel, grid, dim3((unsigned)threads), lds, st, a);
        sn_prof_stop(4, st);
    };
    auto pick = [&](auto tm) {
        constexpr int TMv = decltype(tm)::value;
        const int threads = Geom<TMv>::kThreads;
        const size_t lds = Geom<TMv>::kLdsBytes;
        if (fused2 && gathered) launch(gcn_gemm_kernel<true, true, true, TMv>, threads, lds);
        else if (fused2) launch(gcn_gemm_kernel<true, false, true, TMv>, threads, lds);
        else if (gathered && u->layernorm) launch(gcn_gemm_kernel<true, true, false, TMv>, threads, lds);
        else if (gathered) launch(gcn_gemm_kernel<false, true, false, TMv>, threads, lds);      // (wide GNNs: E = 512, 1024 - each column tile gathers its slice of the table rows)
        else if (u->layernorm) launch(gcn_gemm_kernel<true, false, false, TMv>, threads, lds);
        else launch(gcn_gemm_kernel<false, false, false, TMv>, threads, lds);
    };
    if (small) {
        if (gathered) launch(gcn_gemm_kernel<true, true, true, 64>, Geom<64>::kThreads, Geom<64>::kLdsBytes);
        else launch(gcn_gemm_kernel<true, false, true, 64>, Geom<64>::kThreads, Geom<64>::kLdsBytes);
    } else if (tall) pick(std::integral_constant<int, 256>{});
    else pick(std::integral_constant<int, 128>{});
    if (rc) return rc;
    SN_CHECK_LAUNCH("sn_gcn_gemm");
    return SN_OK;
}
