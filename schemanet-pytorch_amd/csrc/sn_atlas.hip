// IR-Atlas normalisation: one streaming pass over edge_weights [K, n, n] (105 MB at K=100,
// n=512; 419 MB at n=1024) instead of the reference's 3-4 passes + a bmm outer product for the
// prune mask.  HBM-bound: read n*n*4 B, write n*n*4 B per class (+ the in-place zeroing of
// pruned cells, which touches only those cells).
//
// Reference being replaced (schema_inference/graph/schema_net.py):
//   :144-150 get_class_vertices   clamp_min(1e-5) / row-sum, nan_to_num
//   :152-175 get_class_edges      prune mask from the normalised vertices (in-place masked_fill_
//                                 on the Parameter, :164, and a multiplicative mask, :166),
//                                 clamp_min(0) / row-sum, nan_to_num, optional zero diagonal
//   graph/utils.py:25-52          normalize_sum_clamp
#include "sn_common.h"

namespace {

constexpr int kRowsPerBlock = 16;

// what sn_gcn_atlas_adjacency_planes multiplies a clamped edge weight with: 1 / row sum, or 0 when the
// quotient x / s would be NaN or 0 anyway (s == 0: all x are 0; s = inf or NaN) - nan_to_num(x / s) == x * scale
// up to one rounding (one correctly rounded reciprocal per row instead of a division per element)
__device__ __forceinline__ float row_scale(float s) { return (s > 0.0f && s < INFINITY) ? 1.0f / s : 0.0f; }

// skip_pruned_rows (fused route only): the rows of pruned vertices are KNOWN to be zero - this kernel zeroed them in an
// earlier launch on the same parameter versions (the host keeps that fact, SchemaNet._pruned_in_place) - and are not read:
// their row sum is 0.  70 % of a trained atlas.
template <bool SKIP>
__global__ __launch_bounds__(256) void atlas_normalize_kernel(const float *vw, float *ew, int n,
                                                              int use_prune, float thr,
                                                              int remove_self_loop, float *cv, float *ce, float *rowsum, int skip_pruned_rows,
                                                              float *ent, float ent_eps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *keep = smem;                       // [n] vertex survives pruning
    float *red = (float *)(smem + ((n + 15) & ~15));  // [4]
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float *v = vw + (int64_t)k * n;

    // class vertices: clamp_min(1e-5) / sum   (every block of the class recomputes the n-vector)
    float part = 0.0f;
    for (int i = tid; i < n; i += 256) part += fmaxf(v[i], 1.0e-5f);
    part = sn_wave_sum(part);
    if (lane == 0) red[wid] = part;
    __syncthreads();
    const float vsum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = tid; i < n; i += 256) {
        const float c = sn_nan_to_num(fmaxf(v[i], 1.0e-5f) / vsum);
        keep[i] = (!use_prune || c > thr) ? 1 : 0;
        if (blockIdx.y == 0 && cv) cv[(int64_t)k * n + i] = c;
    }
    __syncthreads();

    // fused route (no class_edges output, n % 4 == 0): a wave takes its 4 rows together, 16 bytes per
    // lane per load, all loads of a 256-column chunk in flight before the first use
    if (!ce && (n & 3) == 0) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int i0 = blockIdx.y * kRowsPerBlock + wid * 4;
        // keep masks as all-ones / zero words: pruning is one v_and per element, "did anything change" one compare
        unsigned rowk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rowk[q] = (!use_prune || (i0 + q < n && keep[i0 + q])) ? 0xFFFFFFFFu : 0u;
        if (SKIP && skip_pruned_rows && (rowk[0] | rowk[1] | rowk[2] | rowk[3]) == 0u) {       // wave-uniform: four pruned rows, already zero
            if (rowsum && lane < 4 && i0 + lane < n) rowsum[(int64_t)k * n + i0 + lane] = 0.0f;
            return;
        }
        float s[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int j0 = 0; j0 < n; j0 += 256) {
            const int j = j0 + lane * 4;
            const bool in = j < n;
            f32x4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q < n ? i0 + q : n - 1;
                if (SKIP && skip_pruned_rows && rowk[q] == 0u) { x[q] = f32x4{0.0f, 0.0f, 0.0f, 0.0f}; continue; }      // (wave-uniform: a pruned row, known zero)
                x[q] = *reinterpret_cast<const f32x4 *>(ew + ((int64_t)k * n + i) * n + (in ? j : 0));
            }
            u32x4 colk = {0u, 0u, 0u, 0u};
            if (in) {
                const unsigned k4 = *reinterpret_cast<const unsigned *>(keep + j);          // four keep bytes (0 / 1)
                colk = u32x4{(k4 & 1u) ? ~0u : 0u, (k4 & 0x100u) ? ~0u : 0u, (k4 & 0x10000u) ? ~0u : 0u, (k4 & 0x1000000u) ? ~0u : 0u};
                if (!use_prune) colk = u32x4{~0u, ~0u, ~0u, ~0u};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 raw = __builtin_bit_cast(u32x4, x[q]);
                const u32x4 kept = raw & colk & rowk[q];                                    // masked_fill_(~mask, 0)  :164
                const f32x4 v = __builtin_bit_cast(f32x4, kept);
                s[q] += (fmaxf(v[0], 0.0f) + fmaxf(v[1], 0.0f)) + (fmaxf(v[2], 0.0f) + fmaxf(v[3], 0.0f));
                const u32x4 diff = raw ^ kept;
                if (in && i0 + q < n && ((diff[0] | diff[1]) | (diff[2] | diff[3])) != 0u)
                    *reinterpret_cast<f32x4 *>(ew + ((int64_t)k * n + i0 + q) * n + j) = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = sn_wave_sum(s[q]);
            if (rowsum && lane == 0 && i0 + q < n) rowsum[(int64_t)k * n + i0 + q] = row_scale(t);
        }
        return;
    }

    // class edges, rows of up to 1024 weights (n % 4 == 0; the training route at the Caltech configuration: 404 MB in, 404 MB out):
    // a wave takes its 4 rows together, 16 bytes per lane per access, every load in flight before the first use, and the rows stay
    // in registers between the sum and the quotients (the row-by-row form below: 4-byte loads, the second pass from L2 - 2.3 TB/s).
    // The sum of a row is taken in the order of the fused route above; atlas_normalize_backward_kernel mirrors it (it recomputes the
    // same sum: the quotients of the two passes agree bit for bit).
    if (ce && (n & 3) == 0 && n <= 1024 && ((reinterpret_cast<uintptr_t>(ce) | reinterpret_cast<uintptr_t>(ew)) & 15) == 0) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
        const int i0 = blockIdx.y * kRowsPerBlock + wid * 4;
        unsigned rowk[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rowk[q] = (!use_prune || (i0 + q < n && keep[i0 + q])) ? 0xFFFFFFFFu : 0u;
        f32x4 x[4][4];
        float sr[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = c * 256 + lane * 4;
            const bool in = j < n;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = i0 + q < n ? i0 + q : n - 1;
                x[q][c] = *reinterpret_cast<const f32x4 *>(ew + ((int64_t)k * n + i) * n + (in ? j : 0));
            }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int j = c * 256 + lane * 4;
            const bool in = j < n;
            u32x4 colk = {0u, 0u, 0u, 0u};
            if (in) {
                const unsigned k4 = *reinterpret_cast<const unsigned *>(keep + j);          // four keep bytes (0 / 1)
                colk = u32x4{(k4 & 1u) ? ~0u : 0u, (k4 & 0x100u) ? ~0u : 0u, (k4 & 0x10000u) ? ~0u : 0u, (k4 & 0x1000000u) ? ~0u : 0u};
                if (!use_prune) colk = u32x4{~0u, ~0u, ~0u, ~0u};
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x4 raw = __builtin_bit_cast(u32x4, x[q][c]);
                const u32x4 kept = raw & colk & rowk[q];                                    // masked_fill_(~mask, 0)  :164
                const f32x4 v = __builtin_bit_cast(f32x4, kept);
                sr[q] += (fmaxf(v[0], 0.0f) + fmaxf(v[1], 0.0f)) + (fmaxf(v[2], 0.0f) + fmaxf(v[3], 0.0f));
                const u32x4 diff = raw ^ kept;
                if (in && i0 + q < n && ((diff[0] | diff[1]) | (diff[2] | diff[3])) != 0u)
                    *reinterpret_cast<f32x4 *>(ew + ((int64_t)k * n + i0 + q) * n + j) = v;
                x[q][c] = v;
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = sn_wave_sum(sr[q]);
            const int i = i0 + q;
            if (i >= n) continue;                                                           // (wave-uniform)
            if (rowsum && lane == 0) rowsum[(int64_t)k * n + i] = row_scale(t);
            float ea = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = c * 256 + lane * 4;
                if (j >= n) continue;
                f32x4 y;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    y[e] = sn_nan_to_num(fmaxf(x[q][c][e], 0.0f) / t);
                    if (remove_self_loop && j + e == i) y[e] = 0.0f;
                }
                *reinterpret_cast<f32x4 *>(ce + ((int64_t)k * n + i) * n + j) = y;
                if (ent) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ea += y[e] * logf(y[e] + ent_eps);
                }
            }
            if (ent) {
                ea = sn_wave_sum(ea);
                if (lane == 0) ent[(int64_t)k * n + i] = -ea;
            }
        }
        return;
    }

    // class edges: 4 waves x 4 rows each
    for (int rr = wid; rr < kRowsPerBlock; rr += 4) {
        const int i = blockIdx.y * kRowsPerBlock + rr;
        if (i >= n) break;
        float *row = ew + ((int64_t)k * n + i) * n;
        const bool keep_i = keep[i] != 0;
        float s = 0.0f;
        for (int j = lane; j < n; j += SN_WAVE) {
            float x = row[j];
            if (use_prune && !(keep_i && keep[j])) {
                if (x != 0.0f || x != x) row[j] = 0.0f;   // edge_weights.masked_fill_(~mask, 0)  :164
                x = 0.0f;
            }
            s += fmaxf(x, 0.0f);
        }
        s = sn_wave_sum(s);
        if (rowsum && lane == 0) rowsum[(int64_t)k * n + i] = row_scale(s);
        if (!ce) continue;
        float *out = ce + ((int64_t)k * n + i) * n;
        // ent (training): the row entropy -sum y log(y + eps) of what is being written (schema_inference_loss.py:51-58), summed
        // in the order of row_entropy_kernel (sn_loss.hip) - the same bits as that kernel run on `ce` afterwards
        float ea = 0.0f;
        if ((n & 3) == 0 && ((reinterpret_cast<uintptr_t>(ce) | reinterpret_cast<uintptr_t>(ew)) & 15) == 0) {
            for (int j = lane * 4; j < n; j += SN_WAVE * 4) {
                const float4 raw = *reinterpret_cast<const float4 *>(row + j);
                float x[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (use_prune && !(keep_i && keep[j + q])) x[q] = 0.0f;
                    x[q] = sn_nan_to_num(fmaxf(x[q], 0.0f) / s);
                    if (remove_self_loop && j + q == i) x[q] = 0.0f;
                }
                *reinterpret_cast<float4 *>(out + j) = float4{x[0], x[1], x[2], x[3]};
                if (ent) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) ea += x[q] * logf(x[q] + ent_eps);
                }
            }
        } else {
            for (int j = lane; j < n; j += SN_WAVE) {
                float x = (use_prune && !(keep_i && keep[j])) ? 0.0f : row[j];
                x = sn_nan_to_num(fmaxf(x, 0.0f) / s);
                if (remove_self_loop && j == i) x = 0.0f;
                out[j] = x;
                if (ent) ea += x * logf(x + ent_eps);
            }
        }
        if (ent) {
            ea = sn_wave_sum(ea);
            if (lane == 0) ent[(int64_t)k * n + i] = -ea;
        }
    }
}


// Gradient of the normalised class edges with respect to edge_weights, one pass: what autograd computes through the
// reference's chain  ew * mask -> clamp_min(0) -> / sum(detached) -> nan_to_num [-> masked_fill(eye, 0)]
// (schema_net.py:152-175, utils.py:25-52), multiplication by multiplication so that the non-finite cases come out the same:
//   g_z = g_y where z = c / s is finite, else 0 (nan_to_num);  g_c = g_z / s  (a row whose sum is 0: 0 / 0 = NaN for every
//   cell);  g_m = g_c * [m >= 0] (clamp_min passes the gradient AT 0);  g_x = g_m * mask.
// edge_weights is read in its pruned state (the forward pass has zeroed the masked cells in place).  One wave per row,
// two passes over the row (its sum, then the gradients; the second from L2).
// g_ent (optional): the upstream gradient of the row entropies the forward pass returned next to y; its share
// -g_ent[i] (log(y + eps) + y / (y + eps)) (row_entropy_backward_kernel's expression) is added to g_y here - rows whose g_ent
// is 0 (all but K of them: the loss takes a maximum over rows) skip it.  gy may then be NULL (no other consumer of y).
__global__ __launch_bounds__(256) void atlas_normalize_backward_kernel(const float *vw, const float *ew, const float *gy, int n, int use_prune,
                                                                       float thr, int remove_self_loop, float *gx, const float *g_ent, float ent_eps)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char *keep = smem;                       // [n]
    float *red = (float *)(smem + ((n + 15) & ~15));  // [4]
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float *v = vw + (int64_t)k * n;
    float part = 0.0f;
    for (int i = tid; i < n; i += 256) part += fmaxf(v[i], 1.0e-5f);
    part = sn_wave_sum(part);
    if (lane == 0) red[wid] = part;
    __syncthreads();
    const float vsum = (red[0] + red[1]) + (red[2] + red[3]);
    for (int i = tid; i < n; i += 256) {
        const float c = sn_nan_to_num(fmaxf(v[i], 1.0e-5f) / vsum);
        keep[i] = (!use_prune || c > thr) ? 1 : 0;
    }
    __syncthreads();
    // rows of up to 1024 weights: the forward pass's fast form mirrored (same row sum, bit for bit), 16-byte accesses, the row and its
    // upstream gradient in registers
    if ((n & 3) == 0 && n <= 1024 && ((reinterpret_cast<uintptr_t>(gx) | reinterpret_cast<uintptr_t>(ew) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0) {
        typedef float f32x4 __attribute__((ext_vector_type(4)));
        for (int rr = wid * 4; rr < wid * 4 + 4; ++rr) {
            const int i = blockIdx.y * kRowsPerBlock + rr;
            if (i >= n) break;
            const float *row = ew + ((int64_t)k * n + i) * n;
            const float *grow = gy ? gy + ((int64_t)k * n + i) * n : nullptr;
            float *orow = gx + ((int64_t)k * n + i) * n;
            const float mi = keep[i] ? 1.0f : 0.0f;
            const float ge = g_ent ? g_ent[(int64_t)k * n + i] : 0.0f;       // (wave-uniform)
            f32x4 m4[4], g4[4], mk4[4];
            float sr = 0.0f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = c * 256 + lane * 4;
                if (j < n) {
                    m4[c] = *reinterpret_cast<const f32x4 *>(row + j);
                    g4[c] = grow ? *reinterpret_cast<const f32x4 *>(grow + j) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                }
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = c * 256 + lane * 4;
                if (j < n) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        mk4[c][e] = mi * (keep[j + e] ? 1.0f : 0.0f);
                        m4[c][e] = m4[c][e] * mk4[c][e];
                    }
                    sr += (fmaxf(m4[c][0], 0.0f) + fmaxf(m4[c][1], 0.0f)) + (fmaxf(m4[c][2], 0.0f) + fmaxf(m4[c][3], 0.0f));
                }
            }
            const float st = sn_wave_sum(sr);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int j = c * 256 + lane * 4;
                if (j < n) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m = m4[c][e];
                        const float z = fmaxf(m, 0.0f) / st;
                        float g = g4[c][e];
                        if (ge != 0.0f) {
                            const float y = (remove_self_loop && j + e == i) ? 0.0f : sn_nan_to_num(z), q = y + ent_eps;
                            g = g + (-ge * (logf(q) + y / q));
                        }
                        if (remove_self_loop && j + e == i) g = 0.0f;
                        const float gz = (z == z && fabsf(z) != INFINITY) ? g : 0.0f;
                        const float gc = gz / st;
                        const float gm = gc * (m >= 0.0f ? 1.0f : 0.0f);
                        o[e] = gm * mk4[c][e];
                    }
                    *reinterpret_cast<f32x4 *>(orow + j) = o;
                }
            }
        }
        return;
    }
    for (int rr = wid; rr < kRowsPerBlock; rr += 4) {
        const int i = blockIdx.y * kRowsPerBlock + rr;
        if (i >= n) break;
        const float *row = ew + ((int64_t)k * n + i) * n;
        const float *grow = gy ? gy + ((int64_t)k * n + i) * n : nullptr;
        float *orow = gx + ((int64_t)k * n + i) * n;
        const float mi = keep[i] ? 1.0f : 0.0f;
        const float ge = g_ent ? g_ent[(int64_t)k * n + i] : 0.0f;       // (wave-uniform)
        float s = 0.0f;
        for (int j = lane; j < n; j += SN_WAVE) s += fmaxf(row[j] * (mi * (keep[j] ? 1.0f : 0.0f)), 0.0f);
        s = sn_wave_sum(s);
        for (int j = lane; j < n; j += SN_WAVE) {
            const float mask = mi * (keep[j] ? 1.0f : 0.0f);
            const float m = row[j] * mask;
            const float z = fmaxf(m, 0.0f) / s;
            float g = grow ? grow[j] : 0.0f;
            if (ge != 0.0f) {
                const float y = (remove_self_loop && j == i) ? 0.0f : sn_nan_to_num(z), q = y + ent_eps;
                g = g + (-ge * (logf(q) + y / q));
            }
            if (remove_self_loop && j == i) g = 0.0f;                     // masked_fill(eye, 0) after the normalisation
            const float gz = (z == z && fabsf(z) != INFINITY) ? g : 0.0f;   // nan_to_num: no gradient where z is not finite
            const float gc = gz / s;
            const float gm = gc * (m >= 0.0f ? 1.0f : 0.0f);
            orow[j] = gm * mask;
        }
    }
}


// ------------------------------------------------------------------ a pruned atlas, compacted (round 4)
// perm[k] = the vertices of class k whose normalised weight is above the threshold (the rule of atlas_normalize_kernel:
// c > thr) first, in their own order, then the pruned ones in theirs (a stable partition); n_kept[k] = how many are kept.
// One workgroup per class, four consecutive vertices per thread (n <= 1024), a block-wide exclusive scan of the keep flags.
__global__ __launch_bounds__(256) void atlas_keep_perm_kernel(const float *cv, int n, float thr, int32_t *perm, int32_t *n_kept)
{
    __shared__ int wave_tot[4];
    const int k = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const float *c = cv + (int64_t)k * n;
    int keep[4], cnt = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = 4 * tid + e;
        keep[e] = (i < n && c[i] > thr) ? 1 : 0;
        cnt += keep[e];
    }
    int incl = cnt;                                     // inclusive scan over the wave
#pragma unroll
    for (int off = 1; off < SN_WAVE; off <<= 1) {
        const int up = __shfl_up(incl, off, SN_WAVE);
        if (lane >= off) incl += up;
    }
    if (lane == SN_WAVE - 1) wave_tot[wid] = incl;
    __syncthreads();
    int before = incl - cnt;
    for (int w = 0; w < wid; ++w) before += wave_tot[w];
    const int total = wave_tot[0] + wave_tot[1] + wave_tot[2] + wave_tot[3];
    if (tid == 0) n_kept[k] = total;
    int32_t *p = perm + (int64_t)k * n;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int i = 4 * tid + e;
        if (i >= n) break;
        if (keep[e]) p[before] = i;
        else p[total + (i - before)] = i;               // (i - before) = pruned vertices in front of i
        before += keep[e];
    }
}

// What the GNN needs of a compacted class graph: the words and node weights in the operand's vertex order (weights zero
// beyond the class's extent - those vertices are not in the products) and the isolated vertices' share of the pooled
// feature, pooled_iso[k][f] = sum over the pruned vertices a of class k, in perm order, of w_a * iso[word_a][f] (iso: what an
// isolated vertex of a word contributes per unit of weight, GNN.prepare()).  One workgroup per class; thread = feature.
constexpr int kCompactWaves = 16, kCompactThreads = kCompactWaves * 64, kCompactInFlight = 4;

__global__ __launch_bounds__(kCompactThreads) void class_compact_kernel(const int32_t *perm, const int32_t *n_kept, const float *nodes, const int64_t *ids,
                                                            const float *iso, int n, int E, int rows_iso, int64_t *ids_c, float *w_c,
                                                            float *pooled_iso, int64_t iso_stride)
{
    __shared__ int id_s[1024];
    __shared__ float w_s[1024];
    const int k = blockIdx.x, tid = threadIdx.x;
    const int nk = min(max(n_kept[k], 0), n);
    for (int a = tid; a < n; a += kCompactThreads) {
        const int p = perm[(int64_t)k * n + a];
        const int64_t id = ids[(int64_t)k * n + p];
        const float w = nodes[(int64_t)k * n + p];
        ids_c[(int64_t)k * n + a] = id;
        w_c[(int64_t)k * n + a] = a < nk ? w : 0.0f;
        id_s[a] = (id >= 0 && id < rows_iso) ? (int)id : -1;
        w_s[a] = w;
    }
    __syncthreads();
    // wave w takes the pruned vertices nk + w, nk + w + 16, ... (a fixed order: the same sums in every launch), lane l the
    // features 4 l .. 4 l + 3 of a 256-feature slab - one 16-byte load per lane = one whole row segment per wave-instruction,
    // four rows in flight per wave, sixteen waves: a class of 360 pruned vertices is six rounds of L2 latency (one workgroup
    // per class and K ~ 100: latency, not bandwidth, is what this kernel waits for; with four waves it took 14.5 us, with
    // thread = feature and one dependent 4-byte load per vertex 75 us); the waves' partial sums meet in LDS.
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ f32x4 part[kCompactWaves][64];
    const int lane = tid & 63, wid = tid >> 6;
    for (int f0 = 0; f0 < E; f0 += 256) {
        const int f = f0 + 4 * lane;
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
        if (f < E) {                                    // (E a multiple of 4)
            for (int a0 = nk + wid; a0 < n; a0 += kCompactWaves * kCompactInFlight) {
                f32x4 v[kCompactInFlight];
                float w[kCompactInFlight];
#pragma unroll
                for (int u = 0; u < kCompactInFlight; ++u) {
                    const int a = a0 + kCompactWaves * u;
                    const int id = a < n ? id_s[a] : -1;
                    w[u] = id >= 0 ? w_s[a] : 0.0f;
                    v[u] = *reinterpret_cast<const f32x4 *>(iso + (int64_t)(id >= 0 ? id : 0) * E + f);
                }
#pragma unroll
                for (int u = 0; u < kCompactInFlight; ++u)
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[q] = fmaf(w[u], v[u][q], acc[q]);
            }
        }
        __syncthreads();                                // (the previous slab's partial sums have been read)
        part[wid][lane] = acc;
        __syncthreads();
        if (wid == 0 && f < E) {
            f32x4 t = part[0][lane];
#pragma unroll
            for (int w2 = 1; w2 < kCompactWaves; ++w2) t += part[w2][lane];
            *reinterpret_cast<f32x4 *>(pooled_iso + (int64_t)k * iso_stride + f) = t;
        }
    }
}

}  // namespace

static int atlas_normalize_launch(const char *name, const float *vertex_weights, float *edge_weights, int K, int n, int use_prune,
                                  float prune_threshold, int remove_self_loop, float *class_vertices, float *class_edges, float *row_entropy,
                                  float entropy_eps, void *stream)
{
    SN_REQUIRE(K >= 0 && n > 0, SN_ERR_BAD_ARG, "%s: bad K=%d n=%d", name, K, n);
    if (K == 0) return SN_OK;
    SN_REQUIRE(vertex_weights && edge_weights, SN_ERR_BAD_ARG, "%s: NULL input", name);
    SN_REQUIRE(!row_entropy || class_edges, SN_ERR_BAD_ARG, "%s: row entropies are a by-product of writing class_edges", name);
    SN_REQUIRE(n <= 32768, SN_ERR_UNSUPPORTED, "%s: n=%d > 32768", name, n);
    const dim3 grid((unsigned)K, (unsigned)((n + kRowsPerBlock - 1) / kRowsPerBlock));
    SN_REQUIRE(grid.y <= 65535, SN_ERR_UNSUPPORTED, "%s: n too large", name);
    const size_t lds = ((size_t)(n + 15) & ~size_t(15)) + 16;
    sn_prof_start(3, (hipStream_t)stream);
    hipLaunchKernelGGL(atlas_normalize_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, vertex_weights,
                       edge_weights, n, use_prune, prune_threshold, remove_self_loop, class_vertices, class_edges, (float *)nullptr, 0, row_entropy, entropy_eps);
    sn_prof_stop(3, (hipStream_t)stream);
    SN_CHECK_LAUNCH(name);
    return SN_OK;
}

extern "C" int sn_atlas_normalize(const float *vertex_weights, float *edge_weights, int K, int n,
                                  int use_prune, float prune_threshold, int remove_self_loop,
                                  float *class_vertices, float *class_edges, void *stream)
{
    return atlas_normalize_launch("sn_atlas_normalize", vertex_weights, edge_weights, K, n, use_prune, prune_threshold, remove_self_loop,
                                  class_vertices, class_edges, nullptr, 0.0f, stream);
}

extern "C" int sn_atlas_normalize_entropy(const float *vertex_weights, float *edge_weights, int K, int n, int use_prune, float prune_threshold,
                                          int remove_self_loop, float *class_vertices, float *class_edges, float *row_entropy, float entropy_eps,
                                          void *stream)
{
    SN_REQUIRE(class_edges && row_entropy, SN_ERR_BAD_ARG, "sn_atlas_normalize_entropy: NULL output");
    return atlas_normalize_launch("sn_atlas_normalize_entropy", vertex_weights, edge_weights, K, n, use_prune, prune_threshold, remove_self_loop,
                                  class_vertices, class_edges, row_entropy, entropy_eps, stream);
}

static int g_skip_pruned_rows = 0;     // set by sn_atlas_skip_pruned_rows for the NEXT sn_atlas_prune_rowsum call of this thread's caller

/* The next sn_atlas_prune_rowsum may leave the rows of pruned vertices unread (they are zero: an earlier call on the same
 * versions of both parameters zeroed them).  One-shot: cleared by that call. */
extern "C" void sn_atlas_skip_pruned_rows(int on) { g_skip_pruned_rows = on; }

extern "C" int sn_atlas_prune_rowsum(const float *vertex_weights, float *edge_weights, int K, int n, int use_prune,
                                     float prune_threshold, float *class_vertices, float *row_sum, void *stream)
{
    const int skip = g_skip_pruned_rows && use_prune;
    g_skip_pruned_rows = 0;
    SN_REQUIRE(K >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_atlas_prune_rowsum: bad K=%d n=%d", K, n);
    if (K == 0) return SN_OK;
    SN_REQUIRE(vertex_weights && edge_weights && row_sum, SN_ERR_BAD_ARG, "sn_atlas_prune_rowsum: NULL pointer");
    SN_REQUIRE(n <= 32768, SN_ERR_UNSUPPORTED, "sn_atlas_prune_rowsum: n=%d > 32768", n);
    const dim3 grid((unsigned)K, (unsigned)((n + kRowsPerBlock - 1) / kRowsPerBlock));
    SN_REQUIRE(grid.y <= 65535, SN_ERR_UNSUPPORTED, "sn_atlas_prune_rowsum: n too large");
    const size_t lds = ((size_t)(n + 15) & ~size_t(15)) + 16;
    sn_prof_start(3, (hipStream_t)stream);
    if (skip)
        hipLaunchKernelGGL(atlas_normalize_kernel<true>, grid, dim3(256), lds, (hipStream_t)stream, vertex_weights, edge_weights, n, use_prune,
                           prune_threshold, 0, class_vertices, (float *)nullptr, row_sum, skip, (float *)nullptr, 0.0f);
    else
        hipLaunchKernelGGL(atlas_normalize_kernel<false>, grid, dim3(256), lds, (hipStream_t)stream, vertex_weights, edge_weights, n, use_prune,
                           prune_threshold, 0, class_vertices, (float *)nullptr, row_sum, skip, (float *)nullptr, 0.0f);
    sn_prof_stop(3, (hipStream_t)stream);
    SN_CHECK_LAUNCH("sn_atlas_prune_rowsum");
    return SN_OK;
}

/* Backward of the normalised class edges (sn_atlas_normalize's class_edges as a function of edge_weights): grad_edge_weights
 * [K, n, n] from grad_class_edges, with the values - NaN rows of empty classes' vertices included - autograd gives for the
 * reference's chain of torch ops (schema_net.py:152-175). */
static int atlas_backward_launch(const char *name, const float *vertex_weights, const float *edge_weights, const float *grad_class_edges,
                                 const float *grad_row_entropy, float entropy_eps, int K, int n, int use_prune, float prune_threshold,
                                 int remove_self_loop, float *grad_edge_weights, void *stream)
{
    SN_REQUIRE(K >= 0 && n > 0, SN_ERR_BAD_ARG, "%s: bad K=%d n=%d", name, K, n);
    if (K == 0) return SN_OK;
    SN_REQUIRE(vertex_weights && edge_weights && (grad_class_edges || grad_row_entropy) && grad_edge_weights, SN_ERR_BAD_ARG, "%s: NULL pointer", name);
    SN_REQUIRE(n <= 32768, SN_ERR_UNSUPPORTED, "%s: n=%d > 32768", name, n);
    const dim3 grid((unsigned)K, (unsigned)((n + kRowsPerBlock - 1) / kRowsPerBlock));
    SN_REQUIRE(grid.y <= 65535, SN_ERR_UNSUPPORTED, "%s: n too large", name);
    const size_t lds = ((size_t)(n + 15) & ~size_t(15)) + 16;
    hipLaunchKernelGGL(atlas_normalize_backward_kernel, grid, dim3(256), lds, (hipStream_t)stream, vertex_weights, edge_weights,
                       grad_class_edges, n, use_prune, prune_threshold, remove_self_loop, grad_edge_weights, grad_row_entropy, entropy_eps);
    SN_CHECK_LAUNCH(name);
    return SN_OK;
}

extern "C" int sn_atlas_normalize_backward(const float *vertex_weights, const float *edge_weights, const float *grad_class_edges, int K, int n,
                                           int use_prune, float prune_threshold, int remove_self_loop, float *grad_edge_weights,
                                           void *stream)
{
    SN_REQUIRE(K == 0 || grad_class_edges, SN_ERR_BAD_ARG, "sn_atlas_normalize_backward: NULL pointer");
    return atlas_backward_launch("sn_atlas_normalize_backward", vertex_weights, edge_weights, grad_class_edges, nullptr, 0.0f, K, n, use_prune,
                                 prune_threshold, remove_self_loop, grad_edge_weights, stream);
}

extern "C" int sn_atlas_normalize_entropy_backward(const float *vertex_weights, const float *edge_weights, const float *grad_class_edges,
                                                   const float *grad_row_entropy, float entropy_eps, int K, int n, int use_prune,
                                                   float prune_threshold, int remove_self_loop, float *grad_edge_weights, void *stream)
{
    return atlas_backward_launch("sn_atlas_normalize_entropy_backward", vertex_weights, edge_weights, grad_class_edges, grad_row_entropy, entropy_eps,
                                 K, n, use_prune, prune_threshold, remove_self_loop, grad_edge_weights, stream);
}

extern "C" int sn_atlas_keep_perm(const float *class_vertices, int K, int n, float prune_threshold, int32_t *perm, int32_t *n_kept, void *stream)
{
    SN_REQUIRE(K >= 0 && n > 0 && n <= 1024, SN_ERR_BAD_ARG, "sn_atlas_keep_perm: bad K=%d n=%d (n <= 1024)", K, n);
    if (K == 0) return SN_OK;
    SN_REQUIRE(class_vertices && perm && n_kept, SN_ERR_BAD_ARG, "sn_atlas_keep_perm: NULL pointer");
    hipLaunchKernelGGL(atlas_keep_perm_kernel, dim3((unsigned)K), dim3(256), 0, (hipStream_t)stream, class_vertices, n, prune_threshold, perm, n_kept);
    SN_CHECK_LAUNCH("sn_atlas_keep_perm");
    return SN_OK;
}

/* ids_c [K, n] int64 / w_c [K, n]: class_ingredients / node weights in the compacted operand's vertex order (w_c zero beyond
 * n_kept[k]); pooled_iso [K, E]: sum over class k's pruned vertices of weight * iso[word] (iso [rows_iso, E] fp32: the
 * per-word feature of an isolated vertex).  n <= 1024. */
extern "C" int sn_class_compact(const int32_t *perm, const int32_t *n_kept, const float *nodes, const int64_t *ids, const float *iso, int K, int n,
                                int E, int rows_iso, int64_t *ids_c, float *w_c, float *pooled_iso, int64_t pooled_iso_stride, void *stream)
{
    SN_REQUIRE(K >= 0 && n > 0 && n <= 1024 && E > 0 && rows_iso > 0, SN_ERR_BAD_ARG, "sn_class_compact: bad K=%d n=%d E=%d", K, n, E);
    if (K == 0) return SN_OK;
    SN_REQUIRE(perm && n_kept && nodes && ids && iso && ids_c && w_c && pooled_iso, SN_ERR_BAD_ARG, "sn_class_compact: NULL pointer");
    SN_REQUIRE(E % 4 == 0 && pooled_iso_stride >= E && pooled_iso_stride % 4 == 0 && ((uintptr_t)pooled_iso & 15) == 0, SN_ERR_BAD_ARG,
               "sn_class_compact: E=%d / pooled_iso_stride=%lld must be multiples of 4 (16-byte rows)", E, (long long)pooled_iso_stride);
    hipLaunchKernelGGL(class_compact_kernel, dim3((unsigned)K), dim3(kCompactThreads), 0, (hipStream_t)stream, perm, n_kept, nodes, ids, iso, n, E, rows_iso,
                       ids_c, w_c, pooled_iso, pooled_iso_stride);
    SN_CHECK_LAUNCH("sn_class_compact");
    return SN_OK;
}
