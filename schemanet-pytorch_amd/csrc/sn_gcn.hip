// S4: the GCN layers of the matcher on the matrix cores (reference schema_inference/graph/gnn.py:20-98,
// match.py:33-76).
//
// The two dense products of a layer, adj @ H and (.) @ W^T, are fp32 GEMMs in the reference and the
// scores must stay within 1e-5 of it, so fp16/bf16 inputs alone are not enough.  Every operand is
// therefore kept as TWO fp16 planes, x = hi + lo with hi = fp16(x), lo = fp16(x - hi) (22 significant
// bits; v_mfma_f32_32x32x16_f16 neither flushes fp16 subnormals nor rounds the products,
// tools/mfma_denorm_probe.hip), and a product is three MFMAs accumulated in fp32:
//     a.b ~= hi_a.hi_b + hi_a.lo_b + lo_a.hi_b          (the dropped lo.lo term is ~2^-22 relative)
// which is 3 x 2.5 PF-class instructions instead of one 157 TF-class fp32 MFMA.
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
// Tiling: 512 threads = 8 waves (2 per SIMD) own a 256 x 256 output tile, wave (wm, wn) a 64 x 128
// sub-tile = 2 x 4 MFMA 32x32 accumulators (128 VGPRs).  A k-stage is 32 wide: 64 KiB of LDS holding
// both planes of both operands in MFMA fragment order, filled by LDS-DMA (global_load_lds_dwordx4,
// 8 per wave per stage) into a 2-stage ring; one barrier per stage; 48 MFMAs and 24 ds_read_b128
// per wave per stage.
#include "sn_common.h"

#include <hip/hip_fp16.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2v __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTile = 256;              // output tile edge
constexpr int kStageK = 32;             // k per stage
constexpr int kStageBytes = 64 * 1024;  // [A: 8 m-tiles][2 planes][2 k16] + [B: 8 n-tiles][2 planes][2 k16] x 1 KiB
constexpr int kGemmThreads = 512;

__device__ __forceinline__ void split2(float x, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// ------------------------------------------------------------------ producers of operand planes
// adj = (E + E^T) / 2 + I as hi/lo planes [G][n][ld], zero for columns >= n (reference gnn.py:27-30)
__global__ __launch_bounds__(256) void adjacency_planes_kernel(const float *edges, int n, int ld, _Float16 *out_h, _Float16 *out_l)
{
    __shared__ float tr[64][65];
    const int g = blockIdx.z, bi = blockIdx.y * 64, bj = blockIdx.x * 64;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float *e = edges + (int64_t)g * n * n;
    // E^T tile: rows bj.., columns bi..
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int row = bj + ty * 8 + rr;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int col = bi + tx + 32 * c;
            tr[ty * 8 + rr][tx + 32 * c] = (row < n && col < n) ? e[(int64_t)row * n + col] : 0.0f;
        }
    }
    __syncthreads();
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
        const int il = ty * 8 + rr, i = bi + il;
        const int j = bj + 2 * tx;
        if (i >= n || j >= ld) continue;
        float v[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int jj = j + c;
            float x = 0.0f;
            if (jj < n) {
                x = (e[(int64_t)i * n + jj] + tr[2 * tx + c][il]) / 2.0f;
                if (i == jj) x = x + 1.0f;
            }
            v[c] = x;
        }
        _Float16 h0, l0, h1, l1;
        split2(v[0], h0, l0);
        split2(v[1], h1, l1);
        const half2v h = {h0, h1}, l = {l0, l1};
        const int64_t o = ((int64_t)g * n + i) * ld + j;
        *reinterpret_cast<half2v *>(out_h + o) = h;
        *reinterpret_cast<half2v *>(out_l + o) = l;
    }
}

// Zt[g][f][j] = table[ids[g][j]][f] as hi/lo planes [G][E][ld], zero for j >= n.
// (layer 1 re-associated: adj @ Emb[ids] @ W^T == adj @ (Emb @ W^T)[ids], gnn.py:64-66 + 30)
__global__ __launch_bounds__(256) void gather_planes_kernel(const float *table, int rows_table, const int64_t *ids, int n, int ld, int E,
                                                            _Float16 *out_h, _Float16 *out_l)
{
    __shared__ float tile[64][65];
    __shared__ int rid[64];
    const int g = blockIdx.y, j0 = blockIdx.x * 64;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    if (threadIdx.x < 64) {
        const int j = j0 + threadIdx.x;
        int64_t id = j < n ? ids[(int64_t)g * n + j] : -1;
        rid[threadIdx.x] = (id >= 0 && id < rows_table) ? (int)id : -1;
    }
    __syncthreads();
    for (int f0 = 0; f0 < E; f0 += 64) {
#pragma unroll
        for (int rr = 0; rr < 16; ++rr) {               // wave wid loads rows wid*16 .. +15, 64 features each
            const int jl = wid * 16 + rr, row = rid[jl], f = f0 + lane;
            tile[jl][lane] = (row >= 0 && f < E) ? table[(int64_t)row * E + f] : 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
            const int fl = ty * 8 + rr, f = f0 + fl, j = j0 + 2 * tx;
            if (f < E && j < ld) {
                _Float16 h0, l0, h1, l1;
                split2(tile[2 * tx][fl], h0, l0);
                split2(tile[2 * tx + 1][fl], h1, l1);
                const half2v h = {h0, h1}, l = {l0, l1};
                const int64_t o = ((int64_t)g * E + f) * ld + j;
                *reinterpret_cast<half2v *>(out_h + o) = h;
                *reinterpret_cast<half2v *>(out_l + o) = l;
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void split_planes_kernel(const float *x, int64_t count, _Float16 *out_h, _Float16 *out_l)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < count) split2(x[i], out_h[i], out_l[i]);
}

// ------------------------------------------------------------------ the GEMM
struct GemmArgs {
    const _Float16 *a_hi, *a_lo;
    int64_t a_batch_stride;
    int lda;
    const _Float16 *b_hi, *b_lo;
    int64_t b_batch_stride;
    int ldb;
    int m, n, k;
    float *c;
    int64_t c_batch_stride;
    int ldc;
    _Float16 *c_hi, *c_lo;
    int64_t cp_batch_stride;
    int ldcp, cp_cols;
    const float *bias, *gamma, *beta;
    float eps;
    int relu;
    const int32_t *rows_valid;
    const float *pool_w;
    int64_t pool_w_stride;
    float *pooled;
};

template <bool LN>
__global__ __launch_bounds__(kGemmThreads, 2) void gcn_gemm_kernel(const GemmArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int batch = blockIdx.z, tile_m = blockIdx.y * kTile, tile_n = blockIdx.x * kTile;

    // ---- LDS-DMA sources: wave w copies chunks 8w .. 8w+7 of a stage.  chunk c < 32: A m-tile c>>2,
    // plane (c>>1)&1, k16 step c&1; c >= 32: the same for B.  Lane (r, h) supplies row r, k = 8h..8h+7.
    const _Float16 *src[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = wid * 8 + j;
        const bool is_b = c >= 32;
        const int t = (c & 31) >> 2, plane = (c >> 1) & 1, ks = c & 1;
        int row = (is_b ? tile_n : tile_m) + t * 32 + r;
        const int lim = is_b ? p.n : p.m;
        row = row < lim ? row : lim - 1;                                   // clamp: padded rows are discarded in the epilogue
        const _Float16 *base = is_b ? (plane ? p.b_lo : p.b_hi) + (int64_t)batch * p.b_batch_stride + (int64_t)row * p.ldb
                                    : (plane ? p.a_lo : p.a_hi) + (int64_t)batch * p.a_batch_stride + (int64_t)row * p.lda;
        src[j] = base + ks * 16 + h * 8;
    }
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    auto issue_stage = [&](int t, int buf) {
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + buf * kStageBytes + wid * 8 * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off\n\t"
                         "s_mov_b32 m0, %0" : "=&s"(keep) : "v"(src[j] + (int64_t)t * kStageK), "s"(dst + j * 1024) : "memory");
        }
    };

    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.0f;

    const int n_stages = p.k / kStageK;
    issue_stage(0, 0);
    for (int t = 0; t < n_stages; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // this wave's part of stage t has landed
        __builtin_amdgcn_s_barrier();                          // ... everybody's; stage t-1's buffer is free
        if (t + 1 < n_stages) issue_stage(t + 1, (t + 1) & 1);
        const unsigned char *sa = smem + (t & 1) * kStageBytes, *sb = sa + 32 * 1024;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ah[2], al[2], bh[4], bl[4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int mt = 2 * wm + i;
                ah[i] = *reinterpret_cast<const half8 *>(sa + ((mt * 2 + 0) * 2 + ks) * 1024 + lane * 16);
                al[i] = *reinterpret_cast<const half8 *>(sa + ((mt * 2 + 1) * 2 + ks) * 1024 + lane * 16);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int nt = 4 * wn + j;
                bh[j] = *reinterpret_cast<const half8 *>(sb + ((nt * 2 + 0) * 2 + ks) * 1024 + lane * 16);
                bl[j] = *reinterpret_cast<const half8 *>(sb + ((nt * 2 + 1) * 2 + ks) * 1024 + lane * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __builtin_amdgcn_s_barrier();                              // LDS is reused by the epilogue

    // ---- epilogue.  Accumulator layout: lane (r, h) of tile (i, j) holds column n = tile_n + (4wn+j)*32 + r,
    // rows m = tile_m + (2wm+i)*32 + (q & 3) + 8 (q >> 2) + 4 h for q = 0..15.
    float *red = reinterpret_cast<float *>(smem);              // [2 (wn)][256 rows]
    const int nv = p.rows_valid ? p.rows_valid[batch] : p.m;
    float bias[4], gam[4], bet[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = tile_n + (4 * wn + j) * 32 + r;
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
            for (int j = 0; j < 4; ++j) acc[i][j][q] = live ? acc[i][j][q] + bias[j] : 0.0f;     // pad rows -> 0 (gnn.py:43-45)
        }
    if (LN) {
        // LayerNorm over the 256 columns of a row: 4 lane-local values x 32 lanes x the two wn waves.
        // Two passes (mean, then centred sum of squares); the row statistics live in LDS, not registers.
        float *red2 = red + 512;
        auto row_of = [&](int i, int q) { return (2 * wm + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h; };
        auto half_sum = [&](float s) {
            s += __shfl_xor(s, 16, SN_WAVE);
            s += __shfl_xor(s, 8, SN_WAVE);
            s += __shfl_xor(s, 4, SN_WAVE);
            s += __shfl_xor(s, 2, SN_WAVE);
            s += __shfl_xor(s, 1, SN_WAVE);
            return s;
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float s = half_sum((acc[i][0][q] + acc[i][1][q]) + (acc[i][2][q] + acc[i][3][q]));
                if (r == 0) red[wn * 256 + row_of(i, q)] = s;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = row_of(i, q);
                const float mean = (red[row] + red[256 + row]) / 256.0f;
                float s = 0.0f;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j][q] -= mean;
                    s = fmaf(acc[i][j][q], acc[i][j][q], s);
                }
                s = half_sum(s);
                if (r == 0) red2[wn * 256 + row] = s;
            }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = row_of(i, q);
                const float rstd = 1.0f / sqrtf((red2[row] + red2[256 + row]) / 256.0f + p.eps);
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j][q] = acc[i][j][q] * rstd * gam[j] + bet[j];
            }
        __syncthreads();
    }
    if (p.relu) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = fmaxf(acc[i][j][q], 0.0f);
    }
    // ---- stores
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            const int m = tile_m + (2 * wm + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
            if (m >= p.m) continue;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = tile_n + (4 * wn + j) * 32 + r;
                const float v = acc[i][j][q];
                if (p.c && n < p.n) p.c[(int64_t)batch * p.c_batch_stride + (int64_t)m * p.ldc + n] = v;
                if (p.c_hi && n < p.cp_cols) {
                    _Float16 hi, lo;
                    split2(n < p.n ? v : 0.0f, hi, lo);
                    const int64_t o = (int64_t)batch * p.cp_batch_stride + (int64_t)m * p.ldcp + n;
                    p.c_hi[o] = hi;
                    p.c_lo[o] = lo;
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
        float *pr = reinterpret_cast<float *>(smem) + 1024;    // [4 (wm)][256 cols]
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            part[j] += __shfl_xor(part[j], 32, SN_WAVE);
            if (h == 0) pr[wm * 256 + (4 * wn + j) * 32 + r] = part[j];
        }
        __syncthreads();
        if (tid < 256) {
            const int n = tile_n + tid;
            if (n < p.n) atomicAdd(&p.pooled[(int64_t)batch * p.n + n], (pr[tid] + pr[256 + tid]) + (pr[512 + tid] + pr[768 + tid]));
        }
    }
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" int sn_gcn_adjacency_planes(const float *edges, int G, int n, int ld, void *adj_hi, void *adj_lo, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes: bad G=%d n=%d", G, n);
    if (G == 0) return SN_OK;
    SN_REQUIRE(edges && adj_hi && adj_lo, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes: NULL pointer");
    SN_REQUIRE(ld >= n && ld % kStageK == 0, SN_ERR_BAD_ARG, "sn_gcn_adjacency_planes: ld=%d must be a multiple of 32 >= n=%d", ld, n);
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_adjacency_planes: G=%d > 65535", G);
    hipLaunchKernelGGL(adjacency_planes_kernel, dim3((unsigned)((ld + 63) / 64), (unsigned)((n + 63) / 64), (unsigned)G), dim3(256), 0,
                       (hipStream_t)stream, edges, n, ld, (_Float16 *)adj_hi, (_Float16 *)adj_lo);
    SN_CHECK_LAUNCH("sn_gcn_adjacency_planes");
    return SN_OK;
}

extern "C" int sn_gcn_gather_planes(const float *table, int rows_table, const int64_t *ids, int G, int n, int ld, int E,
                                    void *out_hi, void *out_lo, void *stream)
{
    SN_REQUIRE(G >= 0 && n > 0 && E > 0 && rows_table > 0, SN_ERR_BAD_ARG, "sn_gcn_gather_planes: bad G=%d n=%d E=%d", G, n, E);
    if (G == 0) return SN_OK;
    SN_REQUIRE(table && ids && out_hi && out_lo, SN_ERR_BAD_ARG, "sn_gcn_gather_planes: NULL pointer");
    SN_REQUIRE(ld >= n && ld % kStageK == 0, SN_ERR_BAD_ARG, "sn_gcn_gather_planes: ld=%d must be a multiple of 32 >= n=%d", ld, n);
    SN_REQUIRE(G <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_gather_planes: G=%d > 65535", G);
    hipLaunchKernelGGL(gather_planes_kernel, dim3((unsigned)((ld + 63) / 64), (unsigned)G), dim3(256), 0, (hipStream_t)stream, table,
                       rows_table, ids, n, ld, E, (_Float16 *)out_hi, (_Float16 *)out_lo);
    SN_CHECK_LAUNCH("sn_gcn_gather_planes");
    return SN_OK;
}

extern "C" int sn_split_planes(const float *x, int64_t count, void *out_hi, void *out_lo, void *stream)
{
    SN_REQUIRE(count >= 0, SN_ERR_BAD_ARG, "sn_split_planes: negative count");
    if (count == 0) return SN_OK;
    SN_REQUIRE(x && out_hi && out_lo, SN_ERR_BAD_ARG, "sn_split_planes: NULL pointer");
    const int64_t blocks = (count + 255) / 256;
    SN_REQUIRE(blocks <= 0x7fffffff, SN_ERR_UNSUPPORTED, "sn_split_planes: too many elements");
    hipLaunchKernelGGL(split_planes_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, count, (_Float16 *)out_hi,
                       (_Float16 *)out_lo);
    SN_CHECK_LAUNCH("sn_split_planes");
    return SN_OK;
}

extern "C" int sn_gcn_gemm(const sn_gemm_args *u, void *stream)
{
    SN_REQUIRE(u, SN_ERR_BAD_ARG, "sn_gcn_gemm: NULL args");
    SN_REQUIRE(u->batches >= 0 && u->m > 0 && u->n > 0 && u->k > 0, SN_ERR_BAD_ARG, "sn_gcn_gemm: bad shape m=%d n=%d k=%d batches=%d",
               u->m, u->n, u->k, u->batches);
    if (u->batches == 0) return SN_OK;
    SN_REQUIRE(u->a_hi && u->a_lo && u->b_hi && u->b_lo, SN_ERR_BAD_ARG, "sn_gcn_gemm: NULL operand plane");
    SN_REQUIRE(u->k % kStageK == 0 && u->lda >= u->k && u->ldb >= u->k, SN_ERR_BAD_ARG,
               "sn_gcn_gemm: k=%d must be a multiple of 32 (zero-padded planes) with lda=%d, ldb=%d >= k", u->k, u->lda, u->ldb);
    SN_REQUIRE(u->lda % 8 == 0 && u->ldb % 8 == 0 && u->a_batch_stride % 8 == 0 && u->b_batch_stride % 8 == 0, SN_ERR_BAD_ARG,
               "sn_gcn_gemm: plane rows must be 16-byte aligned");
    SN_REQUIRE(((uintptr_t)u->a_hi | (uintptr_t)u->a_lo | (uintptr_t)u->b_hi | (uintptr_t)u->b_lo) % 16 == 0, SN_ERR_BAD_ARG,
               "sn_gcn_gemm: planes must be 16-byte aligned");
    SN_REQUIRE(u->c || u->c_hi || u->pooled, SN_ERR_BAD_ARG, "sn_gcn_gemm: no output requested");
    SN_REQUIRE(!u->c_hi || (u->c_lo && u->cp_cols >= u->n && u->ldcp >= u->cp_cols), SN_ERR_BAD_ARG, "sn_gcn_gemm: bad output planes");
    SN_REQUIRE(!u->c || u->ldc >= u->n, SN_ERR_BAD_ARG, "sn_gcn_gemm: ldc=%d < n=%d", u->ldc, u->n);
    SN_REQUIRE(!u->layernorm || (u->n == kTile && u->gamma && u->beta), SN_ERR_UNSUPPORTED,
               "sn_gcn_gemm: the LayerNorm epilogue needs n == 256 (got %d) and gamma/beta", u->n);
    SN_REQUIRE(!u->pooled || u->pool_w, SN_ERR_BAD_ARG, "sn_gcn_gemm: pooling without weights");
    SN_REQUIRE(u->batches <= 65535, SN_ERR_UNSUPPORTED, "sn_gcn_gemm: batches=%d > 65535", u->batches);
    GemmArgs a;
    a.a_hi = (const _Float16 *)u->a_hi; a.a_lo = (const _Float16 *)u->a_lo; a.a_batch_stride = u->a_batch_stride; a.lda = u->lda;
    a.b_hi = (const _Float16 *)u->b_hi; a.b_lo = (const _Float16 *)u->b_lo; a.b_batch_stride = u->b_batch_stride; a.ldb = u->ldb;
    a.m = u->m; a.n = u->n; a.k = u->k;
    a.c = u->c; a.c_batch_stride = u->c_batch_stride; a.ldc = u->ldc;
    a.c_hi = (_Float16 *)u->c_hi; a.c_lo = (_Float16 *)u->c_lo; a.cp_batch_stride = u->cp_batch_stride; a.ldcp = u->ldcp; a.cp_cols = u->cp_cols;
    a.bias = u->bias; a.gamma = u->gamma; a.beta = u->beta; a.eps = u->eps; a.relu = u->relu;
    a.rows_valid = u->rows_valid; a.pool_w = u->pool_w; a.pool_w_stride = u->pool_w_stride; a.pooled = u->pooled;
    const dim3 grid((unsigned)((u->n + kTile - 1) / kTile), (unsigned)((u->m + kTile - 1) / kTile), (unsigned)u->batches);
    const size_t lds = 2 * (size_t)kStageBytes;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e1 = hipFuncSetAttribute((const void *)gcn_gemm_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipError_t e2 = hipFuncSetAttribute((const void *)gcn_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e1 != hipSuccess || e2 != hipSuccess) { sn_set_error("sn_gcn_gemm: LDS attribute failed"); return SN_ERR_LAUNCH; }
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    sn_prof_start(4, st);
    if (u->layernorm) hipLaunchKernelGGL(gcn_gemm_kernel<true>, grid, dim3(kGemmThreads), lds, st, a);
    else hipLaunchKernelGGL(gcn_gemm_kernel<false>, grid, dim3(kGemmThreads), lds, st, a);
    sn_prof_stop(4, st);
    SN_CHECK_LAUNCH("sn_gcn_gemm");
    return SN_OK;
}
