/*
 * schemanet_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's schema-inference hot path.  It exists so that the
 * HIP kernels can be checked against an independent, readable implementation.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product
 * package (schemanet-pytorch_amd/) never does and has no CPU fallback.
 *
 * Parity status: PINNED.  the npz fixtures under tests/golden were produced by running the reference itself
 * (/root/reference, C++ sources compiled unmodified into oracle/_ref) in the build container by
 * tests/golden/make_golden.py; tests/test_oracle_golden.py checks every function below against
 * them.  The reference ships no tests or golden vectors of its own.
 *
 * Every function cites the reference lines it follows (paths relative to /root/reference).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off: no FMA contraction, no fast-math, so the
 * float summation orders written here are the ones executed).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define SNO_LANES 64

/* ------------------------------------------------------------------------------------------
 * S1  nearest visual word            discretization/discretization.py:58-70
 *
 * reference:  ingredients = torch.cdist(seq, vocabulary.weight).argmin(dim=1)
 * torch.cdist (fp32, mm path) evaluates sqrt(max(|x|^2 + |c|^2 - 2 x.c, 0)) with a BLAS-defined
 * summation order, so its argmin is only defined up to fp32 rounding on near-ties.  The oracle
 * pins the one thing that IS well defined: the index of the codeword nearest in (near-)exact
 * arithmetic, lowest index on exact ties (argmin's first-occurrence rule).  Products of two
 * fp32 numbers are exact in fp64; the 64-way strided partial sums + xor-butterfly below are the
 * order the HIP re-rank kernel uses, so both sides produce identical fp64 scores bit for bit.
 * |x|^2 is constant per token and sqrt/clamp are monotone, so  score = |c|^2 - 2 x.c.
 * ------------------------------------------------------------------------------------------ */
static double sno_dot64(const float *a, const float *b, int D)
{
    double p[SNO_LANES], q[SNO_LANES];
    for (int l = 0; l < SNO_LANES; ++l) p[l] = 0.0;
    for (int k = 0; k < D; ++k) {
        /* exact product, then one fp64 add: identical to fma(a,b,p) because a*b is exact */
        p[k % SNO_LANES] = p[k % SNO_LANES] + (double)a[k] * (double)b[k];
    }
    for (int off = SNO_LANES / 2; off >= 1; off >>= 1) {
        for (int l = 0; l < SNO_LANES; ++l) q[l] = p[l] + p[l ^ off];
        memcpy(p, q, sizeof(p));
    }
    return p[0];
}

/* cnorm[m] = |c_m|^2 in the same summation order (what sn_codebook_prepare stores). */
void sno_codebook_norms(const float *cb, int M, int D, double *cnorm)
{
    for (int m = 0; m < M; ++m) cnorm[m] = sno_dot64(cb + (size_t)m * D, cb + (size_t)m * D, D);
}

/* x: [n_tok, D] contiguous, cb: [M, D].  out_idx[t] = argmin_m (cnorm[m] - 2 x_t.c_m), first
 * index on ties.  out_score (nullable) receives the winning fp64 score.  A NaN score never wins
 * (all comparisons false); an all-NaN row yields 0.  */
void sno_assign_words(const float *x, int64_t n_tok, const float *cb, int M, int D,
                      int64_t *out_idx, double *out_score)
{
    double *cnorm = (double *)malloc(sizeof(double) * (size_t)M);
    sno_codebook_norms(cb, M, D, cnorm);
    for (int64_t t = 0; t < n_tok; ++t) {
        const float *xt = x + (size_t)t * D;
        double best = INFINITY;
        int64_t bi = 0;
        for (int m = 0; m < M; ++m) {
            double s = cnorm[m] - 2.0 * sno_dot64(xt, cb + (size_t)m * D, D);
            if (s < best) { best = s; bi = m; }
        }
        out_idx[t] = bi;
        if (out_score) out_score[t] = best;
    }
    free(cnorm);
}

/* ------------------------------------------------------------------------------------------
 * codebook extraction: M-step and distortion of SciPy's k-means, float32 path
 * (reference scripts/extract_ingredients.py:33-36 -> scipy.cluster.vq.kmeans;
 *  scipy/cluster/vq.py::_kmeans, scipy/cluster/_vq.pyx::update_cluster_means: cb[label] += obs[i]
 *  in observation order in the observations' dtype, then cb[i] /= count[i]).
 * sums are NOT divided here (the driver divides after the cross-rank reduction).
 * ------------------------------------------------------------------------------------------ */
void sno_kmeans_update(const float *x, int64_t n_tok, int D, const int64_t *ids, int K, float *sums, int64_t *counts)
{
    memset(sums, 0, sizeof(float) * (size_t)K * (size_t)D);
    memset(counts, 0, sizeof(int64_t) * (size_t)K);
    for (int64_t t = 0; t < n_tok; ++t) {
        const int64_t k = ids[t];
        if (k < 0 || k >= K) continue;
        float *row = sums + (size_t)k * D;
        for (int d = 0; d < D; ++d) row[d] = row[d] + x[(size_t)t * D + d];
        counts[k] += 1;
    }
}

/* dist[t] = |x_t - c_ids[t]|_2, fp64, 64-way strided partial sums + xor butterfly (the HIP kernel's order) */
void sno_kmeans_distances(const float *x, int64_t n_tok, int D, const int64_t *ids, const float *centres, int K, double *dist)
{
    for (int64_t t = 0; t < n_tok; ++t) {
        int64_t k = ids[t];
        k = k < 0 ? 0 : (k >= K ? K - 1 : k);
        double p[SNO_LANES], q[SNO_LANES];
        for (int l = 0; l < SNO_LANES; ++l) p[l] = 0.0;
        for (int d = 0; d < D; ++d) {
            const double df = (double)x[(size_t)t * D + d] - (double)centres[(size_t)k * D + d];
            p[d % SNO_LANES] = fma(df, df, p[d % SNO_LANES]);
        }
        for (int off = SNO_LANES / 2; off >= 1; off >>= 1) {
            for (int l = 0; l < SNO_LANES; ++l) q[l] = p[l] + p[l ^ off];
            memcpy(p, q, sizeof(p));
        }
        dist[t] = sqrt(p[0]);
    }
}

/* ------------------------------------------------------------------------------------------
 * helpers shared by the graph builders
 * ------------------------------------------------------------------------------------------ */

/* cpp_extension/src/utils.cpp:6-15  ext::accumulate: sequential fp32 sum from 0.0f, then
 * `sum / container.size()` (size_t -> float conversion, fp32 divide). */
static float sno_accumulate(const float *v, int n, int mean)
{
    float sum = 0.0f;
    for (int i = 0; i < n; ++i) sum = sum + v[i];
    if (mean) sum = sum / (float)n;
    return sum;
}

/* at::nan_to_num_(x, 0): nan -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX */
static float sno_nan_to_num(float v)
{
    if (isnan(v)) return 0.0f;
    if (isinf(v)) return v > 0 ? FLT_MAX : -FLT_MAX;
    return v;
}

typedef struct { int64_t word; int pos; } sno_wp;

static int sno_wp_cmp(const void *a, const void *b)
{
    const sno_wp *x = (const sno_wp *)a, *y = (const sno_wp *)b;
    if (x->word != y->word) return x->word < y->word ? -1 : 1;
    return x->pos - y->pos;
}

/* Groups the L positions of one image by word, words ascending (std::map iteration order,
 * large_scale_feat_to_v.cpp:76-97 / large_scale_feat_to_e.cpp:73-88), positions ascending inside
 * a word (push_back order).  keep (nullable): per-position filter.  Returns the number of
 * distinct kept words; start[g]..start[g+1] index into sorted[]. */
static int sno_group(const int64_t *ing, int L, const unsigned char *keep, sno_wp *sorted, int *start)
{
    int n = 0;
    for (int i = 0; i < L; ++i)
        if (!keep || keep[i]) { sorted[n].word = ing[i]; sorted[n].pos = i; ++n; }
    qsort(sorted, (size_t)n, sizeof(sno_wp), sno_wp_cmp);
    int g = 0;
    for (int i = 0; i < n; ++i)
        if (i == 0 || sorted[i].word != sorted[i - 1].word) start[g++] = i;
    start[g] = n;
    return g;
}

/* ------------------------------------------------------------------------------------------
 * S2  instance vertices      cpp_extension/src/large_scale_feat_to_v.cpp:41-143
 *
 * ing [B,L] i64, attn_cls [B,L] f32 (already clamped+softmaxed by the caller,
 * schema_net.py:295-297), w = vertex_attribute_weights [2,1].
 * Outputs are the concatenations the reference returns (:138-142):
 *   ids     [sum n_i]      sorted distinct words per image
 *   attrs2  [sum n_i, 2]   (count, mean attn) / column max, nan_to_num      (:124)
 *   weights [sum n_i]      attrs2 @ w                                      (:125)
 *   num_v   [B]
 * Returns sum n_i.  Buffers must hold B*L entries.
 * ------------------------------------------------------------------------------------------ */
int64_t sno_instance_v(const int64_t *ing, const float *attn_cls, int B, int L, int mean,
                       const float *w, int64_t *ids, float *attrs2, float *weights, int64_t *num_v)
{
    sno_wp *sorted = (sno_wp *)malloc(sizeof(sno_wp) * (size_t)L);
    int *start = (int *)malloc(sizeof(int) * (size_t)(L + 1));
    float *tmp = (float *)malloc(sizeof(float) * (size_t)L);
    int64_t total = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t *bi = ing + (size_t)b * L;
        const float *ba = attn_cls + (size_t)b * L;
        int n = sno_group(bi, L, NULL, sorted, start);
        float *a2 = attrs2 + (size_t)total * 2;
        for (int g = 0; g < n; ++g) {
            int cnt = start[g + 1] - start[g];
            for (int t = 0; t < cnt; ++t) tmp[t] = ba[sorted[start[g] + t].pos];
            ids[total + g] = sorted[start[g]].word;
            a2[2 * g + 0] = (float)cnt;                    /* :113 */
            a2[2 * g + 1] = sno_accumulate(tmp, cnt, mean); /* :114 */
        }
        /* attrs.div_(attrs.max(0, keepdim)).nan_to_num_(0)   :124 */
        for (int c = 0; c < 2; ++c) {
            float mx = -INFINITY;
            int has_nan = 0;
            for (int g = 0; g < n; ++g) {
                float v = a2[2 * g + c];
                if (isnan(v)) has_nan = 1; else if (v > mx) mx = v;
            }
            if (has_nan) mx = NAN; /* at::max propagates NaN */
            for (int g = 0; g < n; ++g) a2[2 * g + c] = sno_nan_to_num(a2[2 * g + c] / mx);
        }
        /* attrs.matmul(w).squeeze(-1)   :125 */
        for (int g = 0; g < n; ++g) {
            float t0 = a2[2 * g + 0] * w[0];
            float t1 = a2[2 * g + 1] * w[1];
            weights[total + g] = t0 + t1;
        }
        num_v[b] = n;
        total += n;
    }
    free(sorted); free(start); free(tmp);
    return total;
}

/* slot lookup in a per-image dictionary given as parallel arrays sorted by key.
 * unordered_map::operator[] on a missing key inserts 0 (large_scale_feat_to_e.cpp:117-118). */
static int64_t sno_dict_get(const int64_t *keys, const int64_t *vals, int n, int64_t key)
{
    int lo = 0, hi = n - 1;
    while (lo <= hi) {
        int mid = (lo + hi) / 2;
        if (keys[mid] == key) return vals[mid];
        if (keys[mid] < key) lo = mid + 1; else hi = mid - 1;
    }
    return 0;
}

/* the (ci,cj) pair loop shared by both edge builders:
 * large_scale_feat_to_e.cpp:99-125 and feat_to_e.cpp:88-113.  Sum order inside one cell:
 * positions of ci ascending (outer) x positions of cj ascending (inner). */
static void sno_cell(const float *battn, const float *geo, int L, const sno_wp *sorted,
                     const int *start, int gi, int gj, int mean, float *scratch_a, float *scratch_g,
                     float *out_geo, float *out_attn)
{
    int n = 0;
    for (int a = start[gi]; a < start[gi + 1]; ++a)
        for (int c = start[gj]; c < start[gj + 1]; ++c) {
            int p = sorted[a].pos, q = sorted[c].pos;
            scratch_a[n] = battn[(size_t)p * L + q];
            scratch_g[n] = geo[(size_t)p * L + q];
            ++n;
        }
    *out_geo = sno_accumulate(scratch_g, n, mean);
    *out_attn = sno_accumulate(scratch_a, n, mean);
}

/* ------------------------------------------------------------------------------------------
 * S3  instance edges         cpp_extension/src/large_scale_feat_to_e.cpp:33-150
 *
 * attn [B,L,L] f32 (already clamped+softmaxed, schema_net.py:334-336), geo [L,L].
 * The per-image dictionary word->row is given as (dict_keys, dict_vals) rows of length
 * dict_len[b] (keys ascending) at offset dict_off[b].  For the dictionaries SchemaNet builds
 * (schema_net.py:345-348) keys = sorted distinct words and vals = 0..n-1.
 * Outputs, concatenated over images with image b occupying n_b*n_b cells (n_b = dict_len[b]):
 *   edges2 [sum n_b^2, 2]  means, divided by the row sum over cj (:135), nan_to_num, [diag=0]
 *   edges  [sum n_b^2]     edges2 @ w                                              (:140)
 * ------------------------------------------------------------------------------------------ */
void sno_instance_e(const int64_t *ing, const float *attn, const float *geo, int B, int L, int mean,
                    int remove_self_loop, const float *w, const int64_t *dict_keys,
                    const int64_t *dict_vals, const int64_t *dict_off, const int64_t *dict_len,
                    float *edges2, float *edges)
{
    sno_wp *sorted = (sno_wp *)malloc(sizeof(sno_wp) * (size_t)L);
    int *start = (int *)malloc(sizeof(int) * (size_t)(L + 1));
    float *sa = (float *)malloc(sizeof(float) * (size_t)L * L);
    float *sg = (float *)malloc(sizeof(float) * (size_t)L * L);
    size_t out = 0;
    for (int b = 0; b < B; ++b) {
        const int64_t *bi = ing + (size_t)b * L;
        const float *battn = attn + (size_t)b * L * L;
        const int64_t *keys = dict_keys + dict_off[b], *vals = dict_vals + dict_off[b];
        int nd = (int)dict_len[b];
        float *e2 = edges2 + out * 2;
        memset(e2, 0, sizeof(float) * (size_t)nd * nd * 2); /* at::zeros :61 */
        int n = sno_group(bi, L, NULL, sorted, start);
        for (int gi = 0; gi < n; ++gi)
            for (int gj = 0; gj < n; ++gj) {
                int64_t ri = sno_dict_get(keys, vals, nd, sorted[start[gi]].word);
                int64_t rj = sno_dict_get(keys, vals, nd, sorted[start[gj]].word);
                float *cell = e2 + ((size_t)ri * nd + rj) * 2;
                sno_cell(battn, geo, L, sorted, start, gi, gj, mean, sa, sg, &cell[0], &cell[1]);
            }
        /* instance_edges.div_(instance_edges.sum(1, true)).nan_to_num_(0)   :135 */
        for (int i = 0; i < nd; ++i)
            for (int c = 0; c < 2; ++c) {
                float s = 0.0f;
                for (int j = 0; j < nd; ++j) s = s + e2[((size_t)i * nd + j) * 2 + c];
                for (int j = 0; j < nd; ++j) {
                    float *p = &e2[((size_t)i * nd + j) * 2 + c];
                    *p = sno_nan_to_num(*p / s);
                }
            }
        if (remove_self_loop) /* :136-139 */
            for (int i = 0; i < nd; ++i) { e2[((size_t)i * nd + i) * 2] = 0.0f; e2[((size_t)i * nd + i) * 2 + 1] = 0.0f; }
        for (size_t c = 0; c < (size_t)nd * nd; ++c) {
            float t0 = e2[2 * c] * w[0];
            float t1 = e2[2 * c + 1] * w[1];
            edges[out + c] = t0 + t1;
        }
        out += (size_t)nd * nd;
    }
    free(sorted); free(start); free(sa); free(sg);
}

/* ------------------------------------------------------------------------------------------
 * init-time dense vertex attributes     cpp_extension/src/feat_to_v_attr.cpp:19-63, 74-148
 * attr [B, n_vertices, 2] (zero-filled here): attr[b, word] = (count, sum-or-mean attn).
 * ------------------------------------------------------------------------------------------ */
void sno_v_attr(const int64_t *ing, const float *attn_cls, int B, int L, int n_vertices, int mean,
                int ingredients_only, float *attr)
{
    sno_wp *sorted = (sno_wp *)malloc(sizeof(sno_wp) * (size_t)L);
    int *start = (int *)malloc(sizeof(int) * (size_t)(L + 1));
    float *tmp = (float *)malloc(sizeof(float) * (size_t)L);
    memset(attr, 0, sizeof(float) * (size_t)B * n_vertices * 2);
    for (int b = 0; b < B; ++b) {
        const int64_t *bi = ing + (size_t)b * L;
        float *ba = attr + (size_t)b * n_vertices * 2;
        int n = sno_group(bi, L, NULL, sorted, start);
        for (int g = 0; g < n; ++g) {
            int64_t word = sorted[start[g]].word;
            int cnt = start[g + 1] - start[g];
            ba[2 * word + 0] = (float)cnt;
            if (!ingredients_only) {
                for (int t = 0; t < cnt; ++t) tmp[t] = attn_cls[(size_t)b * L + sorted[start[g] + t].pos];
                ba[2 * word + 1] = sno_accumulate(tmp, cnt, mean);
            }
        }
    }
    free(sorted); free(start); free(tmp);
}

/* ------------------------------------------------------------------------------------------
 * init-time class-restricted dense edges      cpp_extension/src/feat_to_e.cpp:31-127
 * class_slot [K, Mtab] i32: slot of a word inside class k's graph, -1 if the word is not one of
 * the class's ingredients (dense form of class_ingredient_dict, schema_net.py:121-126).
 * attr [B, n_max, n_max, 2] (zero-filled here), raw means, NOT normalised (normalisation is
 * done by the caller, schema_net.py:249-254).
 * ------------------------------------------------------------------------------------------ */
void sno_feat_to_e(const int64_t *ing, const float *attn, const float *geo, int B, int L,
                   const int32_t *class_slot, int K, int Mtab, const int64_t *label, int n_max,
                   int mean, float *attr)
{
    (void)K;
    sno_wp *sorted = (sno_wp *)malloc(sizeof(sno_wp) * (size_t)L);
    int *start = (int *)malloc(sizeof(int) * (size_t)(L + 1));
    unsigned char *keep = (unsigned char *)malloc((size_t)L);
    float *sa = (float *)malloc(sizeof(float) * (size_t)L * L);
    float *sg = (float *)malloc(sizeof(float) * (size_t)L * L);
    memset(attr, 0, sizeof(float) * (size_t)B * n_max * n_max * 2);
    for (int b = 0; b < B; ++b) {
        const int64_t *bi = ing + (size_t)b * L;
        const float *battn = attn + (size_t)b * L * L;
        const int32_t *slot = class_slot + (size_t)label[b] * Mtab;
        float *ba = attr + (size_t)b * n_max * n_max * 2;
        for (int i = 0; i < L; ++i) /* :69 only words registered for this class */
            keep[i] = (bi[i] >= 0 && bi[i] < Mtab && slot[bi[i]] >= 0) ? 1 : 0;
        int n = sno_group(bi, L, keep, sorted, start);
        for (int gi = 0; gi < n; ++gi)
            for (int gj = 0; gj < n; ++gj) {
                int ri = slot[sorted[start[gi]].word], rj = slot[sorted[start[gj]].word];
                float *cell = ba + ((size_t)ri * n_max + rj) * 2;
                sno_cell(battn, geo, L, sorted, start, gi, gj, mean, sa, sg, &cell[0], &cell[1]);
            }
    }
    free(sorted); free(start); free(keep); free(sa); free(sg);
}
