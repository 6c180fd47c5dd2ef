/*
 * schemanet_hip.h -- C ABI of libschemanet_hip.so: the MI355X (gfx950) implementation of
 * SchemaNet's schema-inference hot path.
 *
 * This library REPLACES the reference's pybind11 module `cpp_extension.extension`
 * (reference cpp_extension/src/extension.cpp:6-11, four at::Tensor functions, CPU only) and the
 * torch.cdist/argmin inside Discretization.encode (discretization/discretization.py:58-70).
 *
 * Conventions
 *  - plain C: device pointers + sizes + a hipStream_t passed as void*; no torch types.
 *  - every pointer is DEVICE memory unless the name ends in _host.
 *  - nothing here allocates, synchronises or throws: outputs and workspaces are supplied by the
 *    caller, kernels are enqueued on `stream` and the call returns immediately.
 *  - return value: 0 = enqueued; <0 = rejected (nothing enqueued), message via sn_last_error().
 *  - int64 for word ids / labels (torch.long in the reference), float32 for attributes.
 *  - "words"/"ingredients" = visual-word ids, "vertices"/"edges" = IR-graph attributes,
 *    "atlas" = the per-class IR-Atlas, following the reference's vocabulary.
 */
#ifndef SCHEMANET_HIP_H
#define SCHEMANET_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SN_OK 0
#define SN_ERR_BAD_ARG (-1)      /* null pointer, negative size, unsupported shape */
#define SN_ERR_UNSUPPORTED (-2)  /* shape outside what the kernels are built for   */
#define SN_ERR_LAUNCH (-3)       /* hipLaunchKernel reported an error              */
#define SN_ERR_WORKSPACE (-4)    /* workspace too small                            */

#define SN_MAX_TOKENS 196        /* L: 14x14 patch tokens; the graph kernels keep one image in LDS */

/* ABI version, bumped on any change of a signature or of a by-pointer struct (10: round 5, struct_size members; 11: round 6 -
 * sn_gemm_args.zero_skipped and the entry points added after 10, sn_debug_set_gemm_tile; 12: round 6 - sn_weigh_attributes(_backward), sn_weigh_blocks, sn_rectify_linear; tests/test_host_cpu.py holds a hash of
 * this header's declarations next to the version, so that a change of either without the other fails the CPU suite).
 * The three by-pointer argument structs below start with `struct_size`: the caller stores sizeof(the struct it was compiled
 * against) there; a call whose struct_size differs from the library's own sizeof is rejected with SN_ERR_BAD_ARG before any
 * member is read, so a caller built against an older header can never have the library read past the end of its struct. */
int sn_abi_version(void);
/* Thread-local, NUL-terminated description of the last non-zero return on this thread. */
const char *sn_last_error(void);
/* 1 when a gfx950 device is visible to this process (hipGetDeviceProperties), else 0. */
int sn_device_ok(void);

/* ------------------------------------------------------------------------------------------
 * per-kernel timing with HIP events recorded on the launch stream (used by bench.py for the
 * roofline figure; off by default, zero cost when off).
 * kernel ids: 4 = GCN GEMM (split-fp16 MFMA); 0 = assignment screen (fp16 MFMA), 1 = assignment re-rank (fp64),
 *             2 = instance graph (S2+S3), 3 = atlas normalise
 * ------------------------------------------------------------------------------------------ */
#define SN_PROF_KERNELS 5
/* max_samples > 0: (re)start recording up to that many launches per kernel; 0: stop + free. */
int sn_profile_enable(int max_samples);
/* Number of launches recorded so far for kernel_id. */
int sn_profile_count(int kernel_id);
/* Waits for the recorded launches and writes their durations (ms) to out_ms_host[0..n). */
int sn_profile_elapsed_ms(int kernel_id, float *out_ms_host, int n);

/* ------------------------------------------------------------------------------------------
 * S1  visual-word assignment
 * replaces: torch.cdist(seq, vocabulary.weight).argmin(dim=1)  discretization/discretization.py:65
 *
 * Result contract: out[t] = index of the codeword nearest to token t in (near-)exact
 * arithmetic (fp64 re-rank of every candidate the fp16-MFMA screening cannot separate with a
 * rigorous error bound), lowest index on exact ties -- bit-identical to oracle sno_assign_words.
 * ------------------------------------------------------------------------------------------ */

/* Bytes of the packed codebook image for (M, D); 0 for shapes outside 1 <= M <= 65536, D a multiple of 32 with
 * 32 <= D <= 1024 (sn_codebook_prepare / sn_assign_words return SN_ERR_UNSUPPORTED for those).  Inside that range the
 * fp16-MFMA screen runs when D is 192, 384 or 768 and M <= 8192, the exact fp64 kernel otherwise. */
size_t sn_codebook_pack_bytes(int M, int D);

/* Packs codebook [M, D] f32 into `packed` (fp16 MFMA fragments, per-word fp64 norms, scale
 * factors).  Call once per codebook version; the packed image is read-only afterwards. */
int sn_codebook_prepare(const float *codebook, int M, int D, void *packed, void *stream);

/* Form of the fp16-MFMA screening kernel used by mode 0 (same ids, same records, different mapping to the chip):
 *   0  token-stationary: a wave keeps 32 tokens in registers, the codebook streams L2 -> LDS; two workgroups per CU (default)
 *   5  K-outer in one round: the accumulators of up to 208 tokens per CU against the whole codebook resident in eight 256-register
 *      waves, tokens and codebook stream under the matrix pipe (448 < M <= 512, D 192/384, fp32 tokens, n_tokens <= 208 x CUs; other
 *      shapes fall back to 0).  7.5 % less kernel time than 0 in isolation, but it owns every CU for the length of the launch:
 *      for a caller with nothing else in flight (codebook extraction, Discretization.encode on its own)
 * Initial value: environment variable SN_ASSIGN_VARIANT (default 0). */
int sn_assign_variant(void);
int sn_assign_set_variant(int variant);

/* Bytes of scratch sn_assign_words needs for n_tokens tokens. */
size_t sn_assign_workspace_bytes(int64_t n_tokens);

/* Tokens form an [n_outer, n_inner] grid of D-vectors: token (o, i) starts at
 * x + o*x_stride_outer + i*x_stride_inner (strides in floats, rows contiguous in D);
 * its word id is written to out[o*out_stride_outer + i*out_stride_inner].
 *   reference layout  (seq-first mid_feat[1:], [L, bs, D]): n_outer=L, n_inner=bs,
 *                      x strides (bs*D, D), out strides (bs, 1)  -> ingredients [L, bs]
 *   batch-first tokens ([B, 197, D], cls row skipped by passing x + D): n_outer=B, n_inner=196,
 *                      x strides (197*D, D), out strides (196, 1) -> ingredients [B, 196]
 * mode: 0 = fp16-MFMA screening + fp64 re-rank (fast path), 1 = fp64 full scan (slow, exact by
 * construction; used as fallback and cross-check).  Both give identical indices.
 * mode 2 = the screen of mode 0 with the re-rank left to sn_instance_graph (sn_rerank_args below; needs the workspace);
 * mode 3 = only the re-rank of an earlier mode-2 call with the same arguments (the stand-alone kernels: for a consumer
 * that cannot take the deferred finish after all). */
int sn_assign_words(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer,
                    int64_t x_stride_inner, const float *codebook, const void *packed, int M, int D,
                    int64_t *out, int64_t out_stride_outer, int64_t out_stride_inner,
                    void *workspace, size_t workspace_bytes, int mode, void *stream);
/* The same for bfloat16 tokens (what an autocast backbone hands over; strides in bf16 elements, 16-byte aligned rows for
 * the fast path).  The word ids are those of the tokens converted to fp32 (exact), i.e. what the reference computes after
 * `seq.float()`; half the token bytes of sn_assign_words. */
int sn_assign_words_bf16(const void *x_bf16, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer,
                         int64_t x_stride_inner, const float *codebook, const void *packed, int M, int D,
                         int64_t *out, int64_t out_stride_outer, int64_t out_stride_inner,
                         void *workspace, size_t workspace_bytes, int mode, void *stream);

/* ------------------------------------------------------------------------------------------
 * training loss, sparsity terms
 * replaces: entropy(p) = -sum(p * log(p + eps), dim=-1)   schema_inference/loss/schema_inference_loss.py:51-58
 * p [rows, n] contiguous (class_vertices [K, n] or class_edges viewed as [K n, n]).
 * ------------------------------------------------------------------------------------------ */
int sn_row_entropy(const float *p, int64_t rows, int n, float eps, float *entropy, void *stream);
/* grad_p[r][j] = -grad_entropy[r] * (log(p + eps) + p / (p + eps)); rows with grad_entropy[r] == 0 are zero-filled without reading p. */
int sn_row_entropy_backward(const float *p, const float *grad_entropy, int64_t rows, int n, float eps, float *grad_p, void *stream);

/* ------------------------------------------------------------------------------------------
 * codebook extraction: Lloyd's k-means with sn_assign_words as the E-step
 * replaces: scipy.cluster.vq.kmeans(x, num_clusters)   scripts/extract_ingredients.py:33-36
 * (SciPy float32 path: scipy/cluster/vq.py::_kmeans, _vq.update_cluster_means).
 * Token grid and strides as in sn_assign_words; ids = its output (any strides).
 * ------------------------------------------------------------------------------------------ */
/* sums[k, :] = fp32 sum of the tokens assigned to centre k, added IN TOKEN ORDER (SciPy's order, so the
 * result is bit-identical to it); counts[k] = number of members.  K workgroups. */
int sn_kmeans_update(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer, int64_t x_stride_inner,
                     const int64_t *ids, int64_t ids_stride_outer, int64_t ids_stride_inner, int K, int D,
                     float *sums, int64_t *counts, void *stream);
/* The same from a token order grouped by centre: order[offsets[k] .. offsets[k+1]) = flat token indices of centre k
 * in token order (a stable sort of the ids), offsets int64 [K + 1].  No walk over the id stream. */
int sn_kmeans_update_sorted(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer, int64_t x_stride_inner,
                            const int64_t *order, const int64_t *offsets, int K, int D, float *sums, int64_t *counts,
                            void *stream);
/* dist[t] = |x_t - centres[ids[t]]|_2 in fp64 (t = flat token index); its mean is SciPy's distortion. */
int sn_kmeans_distances(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer, int64_t x_stride_inner,
                        const int64_t *ids, int64_t ids_stride_outer, int64_t ids_stride_inner, const float *centres,
                        int K, int D, double *dist, void *stream);

/* ------------------------------------------------------------------------------------------
 * attention taps of the wrapper
 * replaces: IngredientModelWrapper.forward  schema_inference/utils/ingredient_model_wrapper.py:58-68
 * extracted [B*H, L+1, L+1] raw logits -> attn [B, L, L] (head mean, cls row/col dropped) and
 * attn_cls [B, L] (cls row, cls col dropped).
 * ------------------------------------------------------------------------------------------ */
int sn_head_mean_attention(const float *extracted, int B, int H, int L, float *attn,
                           float *attn_cls, void *stream);

/* ------------------------------------------------------------------------------------------
 * S2 + S3  instance IR-graph
 * replaces: ext::feat_to_instance_v  cpp_extension/src/large_scale_feat_to_v.cpp:41-143
 *           ext::feat_to_instance_e  cpp_extension/src/large_scale_feat_to_e.cpp:33-150
 *           and, when *_is_logits, the clamp / softmax / nan_to_num in front of them
 *           (schema_inference/graph/schema_net.py:295-297, 334-336).
 *
 * One workgroup per image.  Outputs are PADDED to n_pad vertices per image (pad id = pad_id,
 * pad weights / edges = 0): the layout Matcher.forward builds with F.pad (match.py:48-54).
 * ------------------------------------------------------------------------------------------ */
/* Deferred finish of S1 (round 4).  sn_assign_words(mode = 2) runs the fp16 screen only: `out` holds its words - final
 * wherever the screen could prove them - and the workspace keeps a flag word and the candidate codes of every token it could
 * not decide (~7 % on isotropic tokens, none on k-means-like ones).  sn_instance_graph with `rerank` set finishes them
 * inside its row phase - fp64, the oracle's summation order, the same ids bit for bit as mode 0 - and writes them back
 * to `out`: no stand-alone re-rank launch between S1 and the graph (reference op: discretization/discretization.py:65,
 * consumer schema_net.py:278-305).  Applies to the edges kernel (attn != NULL) in its prediction configuration and to
 * D in {192, 384, 768} (768: round 5), M <= 2048, L <= 210 (sn_assign_defers); sn_instance_graph returns SN_ERR_UNSUPPORTED otherwise. */
typedef struct sn_rerank_args {
    uint32_t struct_size;         /* sizeof(sn_rerank_args) of the caller's header (checked: see sn_abi_version)     */
    const void *x;                /* the tokens sn_assign_words screened: token (b, l) is the row at element offset
                                     b*x_stride_b + l*x_stride_l (fp32, or bf16 when x_bf16)                      */
    int64_t x_stride_b, x_stride_l;
    int x_bf16;
    int64_t tok_stride_b, tok_stride_l; /* its flat index in the screen's [n_outer, n_inner] grid: b*tok_stride_b + l*tok_stride_l
                                     (batch-first: (n_inner, 1); sequence-first: (1, n_inner))                     */
    int64_t n_tokens;             /* n_outer * n_inner of that call                                              */
    const float *codebook;        /* [M, D]                                                                      */
    const void *packed;           /* sn_codebook_prepare image                                                   */
    int M, D;
    const void *workspace;        /* of the mode-2 call                                                          */
    int64_t *ids;                 /* its `out`: word of (b, l) at [b*ids_stride_b + l*ids_stride_l] (rewritten)   */
    int64_t ids_stride_b, ids_stride_l;
} sn_rerank_args;

/* 1 when sn_assign_words(mode = 2) on this shape leaves flagged tokens to a consumer (0: it finishes them itself and
 * clears the flag words: nothing to pass on) - what the host needs to know before it hands `rerank` to the graph kernel */
int sn_assign_defers(int M, int D);

typedef struct sn_graph_args {
    uint32_t struct_size;         /* sizeof(sn_graph_args) of the caller's header (checked: see sn_abi_version) */
    /* inputs */
    const int64_t *ingredients;   /* word of token (b, l) at [b*ing_stride_b + l*ing_stride_l] */
    int64_t ing_stride_b, ing_stride_l;
    const float *attn_cls;        /* element (b, l) of head h at
                                     [b*acls_stride_b + h*acls_stride_h + l]; NULL = skip vertices */
    const float *attn;            /* element (b, p, q) of head h at [b*attn_stride_b +
                                     h*attn_stride_h + p*attn_stride_r + q]; NULL = skip edges   */
    int64_t acls_stride_b, acls_stride_h;              /* 0, 0 = contiguous [B, L]              */
    int64_t attn_stride_b, attn_stride_r, attn_stride_h; /* 0, 0, 0 = contiguous [B, L, L]      */
    int acls_heads, attn_heads;   /* >1: the kernel averages that many heads first (the
                                     wrapper's torch.mean over heads, ingredient_model_wrapper.py
                                     :58-62), so `extracted` [B*H, L+1, L+1] can be consumed in
                                     place: pass a pointer to element (1,1) resp. (0,1), row
                                     stride L+1, head stride (L+1)^2, batch stride H*(L+1)^2.
                                     0 or 1 = single (already averaged) map                      */
    int B, L;
    int attn_cls_is_logits;       /* 1: apply clamp_v / softmax / nan_to_num(0) first           */
    int attn_is_logits;           /* 1: apply clamp_e / row softmax (no nan_to_num) first       */
    int use_clamp_v, use_clamp_e; /* reference: clamp_*_attn is not None                        */
    float clamp_v, clamp_e;
    const float *geo;             /* [L, L] table, or NULL = grid similarity computed in-kernel */
    int feat_h, feat_w;           /* grid (feat_h*feat_w == L) when geo == NULL                 */
    float dist_alpha, dist_pow;   /* graph/utils.py:72-81                                       */
    const float *w_v, *w_e;       /* [2] each, device (vertex_/edge_attribute_weights)          */
    int mean;                     /* reference `mean` flag (all callers pass 1)                 */
    int remove_self_loop;
    /* optional non-canonical word->row dictionaries (batch_ingredient_dict of the reference):
     * image b owns dict_keys/vals[dict_off[b] .. +dict_len[b]), keys ascending.  NULL = the
     * canonical mapping (rank among the image's sorted distinct words). */
    const int64_t *dict_keys, *dict_vals, *dict_off, *dict_len;
    /* outputs (any may be NULL) */
    int n_pad;                    /* padded vertex count, >= every image's vertex count         */
    int64_t pad_id;               /* id written to padded slots (Matcher: num_codes)            */
    int64_t *out_ids;             /* [B, n_pad]                                                 */
    float *out_v2;                /* [B, n_pad, 2] (count, mean attn) / col max, nan_to_num     */
    float *out_v;                 /* [B, n_pad]     out_v2 @ w_v                                */
    float *out_e2;                /* [B, n_pad, n_pad, 2] means / row sum, nan_to_num, [diag 0] */
    float *out_e;                 /* [B, n_pad, n_pad]     out_e2 @ w_e                         */
    int32_t *out_n;               /* [B] vertices per image                                     */
    int32_t *out_n_max;           /* [1] max over images (atomicMax; caller zeroes it)          */
    float *attn_cls_masked;       /* [B, L] logits after the clamp (the reference's in-place
                                     masked_fill_ side effect, schema_net.py:296); may alias
                                     attn_cls                                                   */
    int skip_edge_padding;        /* 1: rows and columns >= the image's vertex count of out_e /
                                     out_e2 are left UNWRITTEN (two thirds of the padded batch
                                     are such zeros); the consumer must mask by out_n, as
                                     sn_gcn_adjacency_planes_masked does                         */
    const sn_rerank_args *rerank; /* NULL, or the deferred finish of the S1 call that produced `ingredients`
                                     (host pointer, read during the call)                        */
} sn_graph_args;

int sn_instance_graph(const sn_graph_args *args, void *stream);

/* ------------------------------------------------------------------------------------------
 * atlas-initialisation statistics ("per-class schema statistics")
 * replaces: ext::feat_to_v_attr  cpp_extension/src/feat_to_v_attr.cpp:19-63, 74-148
 *           ext::feat_to_e       cpp_extension/src/feat_to_e.cpp:31-127
 *           + their epilogues in SchemaNet.feat_to_full_vertices / feat_to_limited_edges
 *             (schema_net.py:188-207, 222-254)
 *           + the per-class sums of scripts/init_schema_net.py:33-35, 59-61
 * ------------------------------------------------------------------------------------------ */

/* out_attr2 [B, M, 2]: (count, sum-or-mean attn) scattered at the word id, zeros elsewhere
 * (== cpp_feat_to_v_attr).  out_v [B, M]: normalize_max_(dim=1) then @ w_v.  Either may be NULL.
 * is_logits: clamp / softmax WITHOUT nan_to_num first (schema_net.py:200-202). */
int sn_full_vertices(const int64_t *ingredients, int64_t ing_stride_b, int64_t ing_stride_l,
                     const float *attn_cls, int B, int L, int M, int is_logits, int use_clamp,
                     float clamp, int mean, int ingredients_only, const float *w_v,
                     float *out_attr2, float *out_v, void *stream);

/* class_slot [K, Mtab] i32: slot of a word in class k's graph or -1 (dense form of
 * class_ingredient_dict, schema_net.py:121-126).  label [B] i64.
 * out_attr2 [B, n_max, n_max, 2]: raw means at (slot_i, slot_j), zeros elsewhere
 * (== cpp_feat_to_e).  out_e [B, n_max, n_max]: normalize_sum_(dim=2), [diag 0], @ w_e. */
int sn_limited_edges(const int64_t *ingredients, int64_t ing_stride_b, int64_t ing_stride_l,
                     const float *attn, int B, int L, int is_logits, int use_clamp, float clamp,
                     const float *geo, int feat_h, int feat_w, float dist_alpha, float dist_pow,
                     const int32_t *class_slot, int K, int Mtab, const int64_t *label, int n_max,
                     int mean, int remove_self_loop, const float *w_e, float *out_attr2,
                     float *out_e, void *stream);

/* class_sum[label[b], :] += feat[b, :] for b = 0..B-1 IN IMAGE ORDER (deterministic, same order
 * as the reference's python loop); n_tracked[label[b]] += 1.  feat [B, F]. */
int sn_stats_accumulate(const float *feat, const int64_t *label, int B, int64_t F, int K,
                        float *class_sum, float *n_tracked, void *stream);

/* ------------------------------------------------------------------------------------------
 * IR-Atlas normalisation
 * replaces: SchemaNet.get_class_vertices / get_class_edges  schema_net.py:144-175
 * vertex_weights [K, n], edge_weights [K, n, n].  With use_prune the edges touching a vertex
 * whose normalised weight is <= prune_threshold are zeroed IN PLACE in edge_weights (the
 * reference's masked_fill_ on the Parameter, :164) before normalisation.
 * ------------------------------------------------------------------------------------------ */
int sn_atlas_normalize(const float *vertex_weights, float *edge_weights, int K, int n,
                       int use_prune, float prune_threshold, int remove_self_loop,
                       float *class_vertices, float *class_edges, void *stream);

/* Gradient of class_edges (above) with respect to edge_weights, for training: one pass instead of autograd's dozen through
 * the reference's torch ops (schema_net.py:152-175: ew * mask, clamp_min(0), / detached row sum, nan_to_num, zero diagonal),
 * with the same values including the NaN rows of vertices whose row sum is 0.  edge_weights in its pruned state. */
int sn_atlas_normalize_backward(const float *vertex_weights, const float *edge_weights, const float *grad_class_edges, int K, int n,
                                int use_prune, float prune_threshold, int remove_self_loop, float *grad_edge_weights, void *stream);
/* The training form (round 4): class_edges AND the entropy of each of its rows, row_entropy[k][i] = -sum_j y log(y + eps)
 * (schema_inference/loss/schema_inference_loss.py:51-58: the sparsity term takes a maximum over them), from the pass that
 * writes class_edges - the same bits as sn_row_entropy on the result - and, backwards, the gradient of BOTH outputs in one
 * pass: grad_edge_weights from grad_class_edges (may be NULL: zero) + grad_row_entropy [K, n] (may be NULL); what autograd
 * would add up from sn_row_entropy_backward and the other consumers of class_edges first. */
int sn_atlas_normalize_entropy(const float *vertex_weights, float *edge_weights, int K, int n, int use_prune, float prune_threshold,
                               int remove_self_loop, float *class_vertices, float *class_edges, float *row_entropy, float entropy_eps,
                               void *stream);
int sn_atlas_normalize_entropy_backward(const float *vertex_weights, const float *edge_weights, const float *grad_class_edges,
                                        const float *grad_row_entropy, float entropy_eps, int K, int n, int use_prune,
                                        float prune_threshold, int remove_self_loop, float *grad_edge_weights, void *stream);

/* ------------------------------------------------------------------------------------------
 * S4  graph matching
 * replaces pieces of Matcher.forward / GNN.forward  schema_inference/graph/match.py:33-76,
 * gnn.py:20-31, 78-98.  (The dense Linear / bmm GEMMs run on rocBLAS through torch.)
 * ------------------------------------------------------------------------------------------ */

/* adj[g] = (edges[g] + edges[g]^T) / 2 + I      gnn.py:27-30.   edges, adj: [G, n, n] */
int sn_gcn_adjacency(const float *edges, int G, int n, float *adj, void *stream);

/* x [G, n, E] in place: rows r >= n_valid[g] are zeroed (feat_mask, gnn.py:44-45), then
 * LayerNorm(E) with gamma/beta/eps, then ReLU (relu != 0).  n_valid NULL = no mask. */
int sn_mask_layernorm_act(float *x, int G, int n, int E, const int32_t *n_valid,
                          const float *gamma, const float *beta, float eps, int relu, void *stream);

/* out[g, :] = (sum_r feat[g, r, :] * nodes[g, r]) / divisor      gnn.py:94-96
 * divisor = *divisor_dev if non-NULL (device int32: the batch's max vertex count) else n. */
int sn_weighted_pool(const float *feat, const float *nodes, int G, int n, int E,
                     const int32_t *divisor_dev, float *out, void *stream);

/* The two above without the normalised rows in between (GNN widths other than 256, where LayerNorm is not a GEMM epilogue):
 * mask + LayerNorm + activation of x [G, n, E] exactly as sn_mask_layernorm_act (x is NOT modified), then
 *   sn_layernorm_weighted_pool: out [G, E] as sn_weighted_pool of the normalised rows (last layer, gnn.py:94-96);
 *   sn_layernorm_split_planes (below, with the operand planes): the next product's operand.
 * Both are bit-identical to the two-call sequences they replace.  E <= 1024. */
int sn_layernorm_weighted_pool(const float *x, const float *nodes, int G, int n, int E, const int32_t *n_valid,
                               const float *gamma, const float *beta, float eps, int relu,
                               const int32_t *divisor_dev, float *out, void *stream);

/* pred [B, K] from feat_inst [B, E], feat_kg [K, E]        match.py:21-31
 * similarity: 0 inner_product, 1 cosine ((cos+1)/2), 2 euclidean (1/(1+dist)). */
int sn_match_scores(const float *feat_inst, const float *feat_kg, int B, int K, int E,
                    int similarity, float *pred, void *stream);

/* per-class vote aggregation for evaluation: votes[argmax_k pred[b, k]] += 1 for every image
 * (first index on ties, torch.argmax), votes[K] += 1 per image.  votes: [K + 1] f32, accumulated
 * in place; merged across ranks with one all-reduce (replaces the meter.sync() of the reference,
 * schema_inference/eval/evaluation.py:95-97). */
int sn_class_votes(const float *pred, int B, int K, float *votes, void *stream);

/* sn_match_scores followed by sn_class_votes as ONE launch (an evaluation loop that only counts votes per class,
 * eval/evaluation.py:81-97): pred [B, K] is written as by sn_match_scores, votes [K + 1] accumulated as by sn_class_votes,
 * both bit-identical to the two calls. */
int sn_match_scores_votes(const float *feat_inst, const float *feat_kg, int B, int K, int E, int similarity,
                          float *pred, float *votes, void *stream);

/* out[g][o] = bias[o] + sum_e (pooled[g][e] / divisor) * weight[o][e] with pooled[g][e] = sum_t
 * pooled_parts[g][t][e], t < parts: the mean over the padded length followed by the GNN's final
 * Linear (gnn.py:96-98).  divisor = *divisor_dev (int32 on the device, e.g. the batch-maximum
 * vertex count) when non-NULL, else divisor_host. */
int sn_pool_fc(const float *pooled_parts, int G, int parts, int E, const int32_t *divisor_dev, float divisor_host,
               const float *weight, const float *bias, int E_out, float *out, void *stream);
/* The same with the weight given transposed, weight_t[E][E_out] (a weight-only operand the caller keeps across steps):
 * thread = output, no cross-lane reduction; E <= 2048. */
int sn_pool_fc_t(const float *pooled_parts, int G, int parts, int E, const int32_t *divisor_dev, float divisor_host,
                 const float *weight_t, const float *bias, int E_out, float *out, void *stream);

/* ---- S4 on the matrix cores: GCN layers with split-fp16 operands ----------------------------
 * Replaces torch.bmm(adj, feat) + nn.Linear + masked_fill + LayerNorm + ReLU + pooling of the
 * reference's GNN (schema_inference/graph/gnn.py:20-98).  Every operand is a pair of fp16 planes
 * (x s = hi + lo with s a power-of-two scale kept next to the planes), a product is three fp16 MFMAs accumulated in
 * fp32.  Precision: |x s - hi - lo| <= max(2^-22 |x s|, 2^-25) (the second term: lo is an fp16 subnormal when
 * |x s| < 2^-3), so with s chosen to put the operand's largest possible magnitude at 2^13 .. 2^14 an element keeps 22
 * significant bits down to 2^-17 of that maximum and errs by 2^-39 of it below; scale_dev arguments (device scalars,
 * NULL = 1) and the *_scale fields of sn_gemm_args carry s.
 * Planes are BLOCKED in MFMA fragment order: a [rows, k] operand is ceil(rows/32) x ceil(k/16)
 * blocks of 1 KiB; element (row, kk) lives in block (row >> 5, kk >> 4) at fp16 index
 * ((kk >> 3 & 1) * 32 + (row & 31)) * 8 + (kk & 7); rows / k beyond the operand are zero. */

/* Fused atlas -> GCN operand route (no materialised class_edges):
 *  sn_atlas_prune_rowsum: class_vertices, the in-place pruning of edge_weights (schema_net.py:164)
 *    and row_sum[k][i] = 1 / sum_j max(pruned edge_weights[k][i][j], 0) (0 when that sum is 0, inf or NaN);
 *  sn_gcn_atlas_adjacency_planes: adj = (E + E^T)/2 + I with E[i][j] = nan_to_num(max(w_ij, 0) *
 *    row_sum[i]) (diagonal zero when remove_self_loop) - the same values sn_atlas_normalize followed
 *    by sn_gcn_adjacency_planes produce, without writing and re-reading the [K, n, n] atlas.  class_edges_out
 *    (optional, [K, n, n] fp32): E itself as a by-product of the same pass, for callers that hand `class_edges` on
 *    (schema_inference/graph/__init__.py:52-54 of the reference returns it next to the scores); E = w * (1 / row sum),
 *    i.e. within one rounding of sn_atlas_normalize's w / row sum. */
int sn_atlas_prune_rowsum(const float *vertex_weights, float *edge_weights, int K, int n, int use_prune,
                          float prune_threshold, float *class_vertices, float *row_sum, void *stream);
/* One-shot hint for the NEXT sn_atlas_prune_rowsum call: the rows of pruned vertices are already zero (an earlier call on
 * the same versions of vertex_weights and edge_weights zeroed them in place) and need not be read - 70 % of a trained atlas. */
void sn_atlas_skip_pruned_rows(int on);
int sn_gcn_atlas_adjacency_planes(const float *pruned_edge_weights, const float *row_sum, int K, int n,
                                  int remove_self_loop, float scale, void *adj_hi, void *adj_lo, float *class_edges_out,
                                  void *stream);
/* A PRUNED atlas, compacted (round 4).  The sparsity terms of the training loss push most vertex weights of a class under
 * prune_node_threshold (schema_net.py:152-166): their rows and columns of the class graph are zero, the vertex is an
 * isolated node.  perm [K, n] int32: vertex a of class k's operand is vertex perm[k][a] of the stored graph - the kept
 * vertices first, in their own order; n_kept [K] int32.  Only the n_kept[k] x n_kept[k] corner (+ identity) of the operand
 * is produced (rounded up to the 32 x 16 blocks, zero inside them), from the kept rows of the atlas only; sn_gcn_gemm
 * consumes it with per-graph extents (sn_gemm_args.extent_stride = 1, m_extent = k_extent = rows_valid = n_kept).  The
 * isolated vertices' share of the class feature does not need a product (host: GNN.prepare()["iso"]).  n <= 1024. */
int sn_atlas_keep_perm(const float *class_vertices, int K, int n, float prune_threshold, int32_t *perm, int32_t *n_kept,
                       void *stream);
/* ids_c / w_c [K, n]: class_ingredients and node weights in the compacted operand's vertex order (w_c zero beyond
 * n_kept[k]); pooled_iso[k * pooled_iso_stride + f], f < E = sum over class k's pruned vertices of weight * iso[word]
 * (iso [rows_iso, E] fp32; pooled_iso_stride >= E floats: the rows may be a slot of sn_gcn_gemm's `pooled`). */
int sn_class_compact(const int32_t *perm, const int32_t *n_kept, const float *nodes, const int64_t *ids, const float *iso,
                     int K, int n, int E, int rows_iso, int64_t *ids_c, float *w_c, float *pooled_iso, int64_t pooled_iso_stride, void *stream);
int sn_gcn_atlas_adjacency_planes_compact(const float *pruned_edge_weights, const float *row_sum, int K, int n,
                                          int remove_self_loop, float scale, const int32_t *perm, const int32_t *n_kept,
                                          void *adj_hi, void *adj_lo, void *stream);

/* fp16 elements of one plane of a [rows, k] operand (per batch entry). */
int64_t sn_gcn_plane_elems(int rows, int k);

/* adj = (E + E^T)/2 + I  (gnn.py:27-30) as blocked planes of a [n, n] operand per graph.
 * extent_dev (optional, here and below): int32 on the device, e.g. the largest vertex count of the
 * batch; blocks whose rows / k lie entirely beyond it (rounded up to 32 / 16) are not produced and
 * must not be consumed - sn_gcn_gemm's m_extent / k_extent skip exactly those. */
/* The same with a vertex count per graph: element (i, j) of graph g is taken as 0 unless i, j < n_valid[g] and is not
 * read (edges written with sn_graph_args.skip_edge_padding); the identity still covers all n rows (gnn.py:27-30 on the
 * zero-padded batch, match.py:48-54).  n_valid NULL = sn_gcn_adjacency_planes. */
/* scale (here and in sn_gcn_atlas_adjacency_planes): the planes hold adj * scale, a power of two in (0, 65536] given by
 * value; |adj| * scale must stay below 65504 (normalised graphs: adj <= 2 + |w_e|_1, scale 2^10 leaves room up to 63). */
int sn_gcn_adjacency_planes_masked(const float *edges, int G, int n, const int32_t *n_valid, const int32_t *extent_dev,
                                   float scale, void *adj_hi, void *adj_lo, void *stream);
int sn_gcn_adjacency_planes(const float *edges, int G, int n, const int32_t *extent_dev, float scale, void *adj_hi,
                            void *adj_lo, void *stream);
/* sn_gcn_adjacency_planes_masked with every graph's OWN extent (round 5): only the blocks of graph g that hold a row or a k
 * below n_valid[g] (rounded up to 32 / 16) are produced - for sn_gcn_gemm with one extent per graph (m_extent = k_extent =
 * n_valid, extent_stride 1), which reads no other block's rows below the count and masks the rows above it (rows_valid). */
int sn_gcn_adjacency_planes_per_graph(const float *edges, int G, int n, const int32_t *n_valid, float scale, void *adj_hi,
                                      void *adj_lo, void *stream);

/* sn_gcn_adjacency_planes_per_graph through a vertex permutation (round 5, training with compacted class graphs): vertex a of graph
 * g's operand is vertex perm[g][a] of `edges` (sn_atlas_keep_perm: the kept vertices first), only the n_kept[g] x n_kept[g] corner
 * (+ identity) is produced.  n <= 1024. */
int sn_gcn_adjacency_planes_compact(const float *edges, int G, int n, const int32_t *perm, const int32_t *n_kept, float scale,
                                    void *adj_hi, void *adj_lo, void *stream);

/* Zt[g][f][j] = table[ids[g][j]][f] as blocked planes of an [E, n] operand per graph (ids outside
 * [0, rows_table) give zero): the transposed, gathered B operand of layer 1 (gnn.py:64-66 with the
 * Linear folded into the embedding table). */
int sn_gcn_gather_planes(const float *table, int rows_table, const int64_t *ids, int G, int n, int E,
                         const int32_t *extent_dev, const float *scale_dev, void *out_hi, void *out_lo, void *stream);

/* hi/lo split of fp32 x [batches][rows][ld] (cols valid per row, batch stride in floats) into
 * blocked planes of a [rows, cols] operand per batch entry. */
int sn_split_planes(const float *x, int batches, int rows, int cols, int64_t ld, int64_t batch_stride,
                    const float *scale_dev, void *out_hi, void *out_lo, void *stream);

/* The same for x^T: planes of the [cols, rows] operand per batch entry (k = rows), read from the untransposed x - the B
 * operand of Y = adj . X for X [n, E] row-major (training: ops.sym_adj_matmul) without a transposed copy of X. */
int sn_split_planes_transposed(const float *x, int batches, int rows, int cols, int64_t ld, int64_t batch_stride,
                               const float *scale_dev, void *out_hi, void *out_lo, void *stream);

/* sn_split_planes / sn_split_planes_transposed of x [G, n, E] (contiguous: nodes x features per graph) with one node extent per graph
 * (round 5, training with compacted class graphs): the blocks that hold no node below node_extent[g] - row blocks of the plain
 * form, k chunks of the transposed one - are not produced; for consumers that skip them (sn_gcn_gemm with per-graph extents). */
int sn_split_planes_nodes(const float *x, int G, int n, int E, const float *scale_dev, const int32_t *node_extent, int transposed,
                          void *out_hi, void *out_lo, void *stream);
/* sn_mask_layernorm_act(x [G, n, E]) followed by sn_split_planes of the result (times *scale_dev) as one pass that leaves x
 * untouched and never stores the normalised rows: blocked planes of an [n, E] operand per graph (gnn.py:43-46 feeding the
 * next layer's Linear).  E % 16 == 0, E <= 1024.  Bit-identical to the two calls. */
int sn_layernorm_split_planes(const float *x, int G, int n, int E, const int32_t *n_valid, const float *gamma,
                              const float *beta, float eps, int relu, const float *scale_dev, void *out_hi,
                              void *out_lo, void *stream);

/* C[b] = A[b] . Bt[b]^T for b < batches; A = planes of an [m, k] operand, Bt = planes of an [n, k]
 * operand, k % 16 == 0 (the padded k of the planes); batch strides in fp16 elements, 0 = shared
 * operand.  Epilogue, in this order: + bias[n]; rows >= rows_valid[b] set to 0 (gnn.py:43-45);
 * LayerNorm over the n == 256 columns with gamma/beta/eps (gnn.py:46); ReLU; then any of: fp32 C
 * [m][ldc]; blocked hi/lo planes of C as an [m, cp_cols] operand (columns [n, cp_cols) zero);
 * pooled[b][t][n] = sum over the rows m of row group t (128 rows, whatever the tile height the
 * library picks) of pool_w[b][m] * C[m][n], t < ceil(m / 128): partial sums of the node-weighted
 * pooling (gnn.py:96), added up in a fixed order by sn_pool_fc (no atomics: results are
 * bit-reproducible). */
typedef struct sn_gemm_args {
    uint32_t struct_size;                 /* sizeof(sn_gemm_args) of the caller's header (checked: see sn_abi_version) */
    const void *a_hi, *a_lo; int64_t a_batch_stride;
    const void *b_hi, *b_lo; int64_t b_batch_stride;
    int m, n, k, batches;
    float *c; int64_t c_batch_stride; int ldc;
    void *c_hi, *c_lo; int64_t cp_batch_stride; int cp_cols;
    const float *bias;
    const float *gamma, *beta; float eps; int layernorm, relu;
    const int32_t *rows_valid;
    const float *pool_w; int64_t pool_w_stride; float *pooled;
    const int32_t *m_extent, *k_extent;   /* device scalars or NULL: row tiles >= *m_extent are skipped (their
                                             pooled partial is zero), the k loop stops at *k_extent (per graph with
                                             extent_stride = 1) */
    /* gathered B (optional; then b_hi / b_lo are ignored): Bt[g][f][j] = table[b_ids[g * b_ids_stride + j]][f] for
     * j < b_ids_n, zero for other j and for ids outside [0, b_table_rows).  b_table_hi / _lo: row-major fp16 planes
     * [b_table_rows + 1][256] of the table split as hi + lo, the last row zero.  Needs n == 256, the LayerNorm
     * epilogue and k <= 1024.  (Layer 1 of the GNN: (Emb W1^T)[ids], gnn.py:64-66 + 30, without the gathered copy.) */
    const void *b_table_hi, *b_table_lo;
    const int64_t *b_ids; int64_t b_ids_stride; int b_ids_n, b_table_rows;
    /* fused next-layer product (optional; needs n == 256, the LayerNorm epilogue, c_hi / c_lo only): the epilogue result H
     * [m, 256] is not stored; the output planes hold Zt = W . H^T as a [256, cp_cols >= m] operand (rows = output features,
     * k = rows of H; what the stand-alone product A = W planes, Bt = H planes would write: the Linear of the next GraphConv,
     * gnn.py:29).  next_w_hi / _lo: planes of the [256, 256] weight with its COLUMNS permuted: column kappa of the
     * planes = column (4 (kappa >> 7) + (kappa & 3)) * 32 + ((kappa >> 2) & 31) of W (the order the kernel's accumulators
     * hold the features in; sn_gcn.hip). */
    const void *next_w_hi, *next_w_lo;
    /* power-of-two operand scales (device scalars; NULL = 1).  The planes of A / B hold x * (*a_scale) / x * (*b_scale)
     * (b_scale also covers the gathered table); the accumulators are multiplied by 1 / (a_scale * b_scale) before the
     * bias; output planes are written as result * (*out_scale) (the consumer passes that scalar as its a_scale / b_scale).
     * Fused next-layer product: the W planes hold W * (*next_w_scale), the epilogue's H fragments are formed as
     * H * (*next_h_scale) (a bound on |H|: 16 max|gamma| + max|beta| for a LayerNorm output), the result planes hold
     * Zt * (*out_scale).  Every scale must be a power of two (then all of this is exact) chosen so that the largest
     * magnitude the operand can hold lands at 2^13 .. 2^14: see "precision" above. */
    const float *a_scale, *b_scale, *out_scale, *next_w_scale, *next_h_scale;
    int extent_stride;            /* 0: m_extent / k_extent are one value for the batch; 1: one per graph ([batches]) */
    int accumulate;               /* 1: the fp32 result is ADDED to what c holds (c += A . Bt^T; plain product only: no bias / LayerNorm /
                                     ReLU): the layers of a training pass sum their dY . X^T into one adjacency gradient */
    int pooled_parts;             /* 0, or the number of [n] partial sums per graph `pooled` is laid out with (>= ceil(m / 128): the
                                     caller keeps the further slots, e.g. sn_class_compact's pooled_iso) */
    int zero_skipped;             /* 1 (with m_extent and c): the row tiles past a graph's extent, which are not multiplied, are
                                     written as zeros by the workgroups that skip them - with rows_valid the whole fp32 result is
                                     then defined without a clearing pass over it (round 5: training with compacted class graphs) */
} sn_gemm_args;
int sn_gcn_gemm(const sn_gemm_args *args, void *stream);

/* ------------------------------------------------------------------------------------------
 * training-side streaming passes (csrc/sn_train.hip): one kernel each for chains of library launches
 * ------------------------------------------------------------------------------------------ */
/* *scale = the power of two s with s * max|x| in (top / 2, top] (clamped to 2^-60 .. 2^60; 1 for an all-zero or non-finite
 * x): the operand scale of split-fp16 planes (top = 2^13), from ONE read of x and without a host synchronisation.
 * partial: sn_pow2_scale_blocks(n) 32-bit words of scratch; amax_out (optional) receives max|x| (NaN if x holds one). */
int sn_pow2_scale_blocks(int64_t n);
int sn_pow2_scale(const float *x, int64_t n, float top, void *partial, float *scale, float *amax_out, void *stream);
/* s [G, n, n] <- (s + s^T) / 2 per graph, in place: the chain rule through the GCN operand (E + E^T)/2 + I
 * (reference schema_inference/graph/gnn.py:27-30), applied once to the sum of the layers' dY . X^T. */
int sn_sym_half_inplace(float *s, int G, int n, void *stream);
/* Training with compacted class graphs (round 5).  corner [G, n, n]: the sum of the layers' dY . X^T taken in the COMPACTED vertex
 * order (vertex a of graph g = vertex perm[g][a] of the stored graph, the kept vertices first: sn_atlas_keep_perm), written for
 * rows < n_kept[g] only.  out [G, n, n] <- the edge gradient in the stored order: out[g][i][j] = (corner[g][a][b] +
 * corner[g][b][a]) / 2 with i = perm[g][a], j = perm[g][b] when both a, b < n_kept[g], 0 elsewhere (the chain rule through
 * (E + E^T)/2 + I, reference gnn.py:27-30; a pruned vertex's row and column of E are constants).  The corner is symmetrised IN
 * PLACE on the way.  n <= 1024. */
int sn_sym_scatter_corner(float *corner, const int32_t *perm, const int32_t *n_kept, int G, int n, float *out, void *stream);
/* x [rows, n] <- nan_to_num(clamp_min(x, min_val) / sum(clamp_min(x, min_val), -1), 0), then x[r, r % diag_n] <- 0 when
 * diag_n > 0 (x = a [K, n, n] tensor viewed as [K n, n]): `SchemaNet.normalize()` on one parameter in one pass
 * (reference schema_net.py:133-142, graph/utils.py:7-13, :59-61).  The row sum is taken in fp32 in a fixed order
 * (not torch's: the quotient can differ from the library chain in the last bit). */
int sn_normalize_sum_rows(float *x, int64_t rows, int n, float min_val, int diag_n, void *stream);
/* The GNN layer's tail with autograd (reference gnn.py:43-46: pad rows -> 0, LayerNorm, ReLU): y = act(LN(mask(x))) out of place
 * (the values of sn_mask_layernorm_act), and its backward as one pass over (x, dy): dx [G, n, E]; dgamma_dbeta [2, E] = the
 * column sums, reduced in a fixed order through `partial` (sn_ln_act_blocks(G n) * 2 * E floats of scratch).  E <= 1024. */
int sn_ln_act_blocks(int64_t rows);
int sn_mask_layernorm_act_forward(const float *x, float *y, int G, int n, int E, const int32_t *n_valid, const float *gamma,
                                  const float *beta, float eps, int relu, void *stream);
int sn_mask_layernorm_act_backward(const float *x, const float *dy, int G, int n, int E, const int32_t *n_valid, const float *gamma,
                                   const float *beta, float eps, int relu, float *dx, float *partial, float *dgamma_dbeta,
                                   void *stream);
/* Gradient of table[ids] with respect to the table for an index tensor sorted beforehand: order [N] = the positions of the
 * flattened ids in ascending id order (stable), seg [rows + 1] = where each id's run starts in it; grad [rows, E] (every row
 * written; row padding_idx, if in range, zero) from dy [N, E].  E % 4 == 0.  (reference gnn.py:83: nn.Embedding over the class
 * graphs' words - constant between training iterations.) */
int sn_embedding_grad_sorted(const float *dy, const int64_t *order, const int64_t *seg, int rows, int E, int padding_idx, float *grad,
                             void *stream);
/* The same gradient without a sort, for ids that are new in every iteration (round 5): grad[w][:] = sum over the positions p (in
 * position order) with ids[p] == w of dy[p][:]; row padding_idx = 0; ids outside [0, rows) contribute nothing.  Meant for short index
 * tensors (every word's workgroup scans all n_ids: the instance graphs' 12 544 ids); E <= 1024. */
int sn_embedding_grad_scan(const float *dy, const int64_t *ids, int64_t n_ids, int rows, int E, int padding_idx, float *grad, void *stream);
/* Backward of sn_weighted_pool (pooled[g] = sum_i nodes[g][i] feat[g][i] / divisor, reference gnn.py:96) in one pass over feat:
 * grad_feat [G, n, E], grad_nodes [G, n] from grad_pooled [G, E].  E % 4 == 0. */
int sn_weighted_pool_backward(const float *feat, const float *nodes, const float *grad_pooled, int G, int n, int E,
                              const int32_t *divisor_dev, float *grad_feat, float *grad_nodes, void *stream);
/* out[i] = attr2[i][0] * w[0] + attr2[i][1] * w[1], i < n: the reference's `attr2 @ w` behind its instance vertices / edges
 * (cpp_extension/src/large_scale_feat_to_v.cpp, large_scale_feat_to_e.cpp:141-147; squeezed) as one pass under autograd (round 6),
 * and its gradient with respect to the two weights: dw[k] = sum_i g[i] * attr2[i][k], reduced in a fixed order through `partial`
 * (sn_weigh_blocks(n) * 2 doubles of scratch, 8-byte aligned). */
int sn_weigh_blocks(int64_t n);
int sn_weigh_attributes(const float *attr2, int64_t n, const float *w, float *out, void *stream);
int sn_weigh_attributes_backward(const float *attr2, const float *g, int64_t n, void *partial, float *dw, void *stream);
/* out[i] = x[i] if x[i] > a else a - 1 + 1 / (1 + a - x[i]); deriv[i] = d out / d x (1, or 1 / (1 + a - x[i])^2): the rectified sparsity
 * terms of the reference's loss (schema_inference/loss/schema_inference_loss.py:61-67) without its python branch on a device scalar. */
int sn_rectify_linear(const float *x, int n, float a, float *out, float *deriv, void *stream);
/* `graph`: a captured, not yet instantiated hipGraph_t (torch.cuda.CUDAGraph(keep_graph=True).raw_cuda_graph()).  Every
 * one-dimensional memset node is replaced by a kernel node with the same predecessors and successors (a captured memset
 * node was seen not to clear on replay on ROCm 7.2; a graph PyTorch captured holds the library's own: semaphores of
 * multi-block reductions, the zero fill of embedding_dense_backward).  *n_replaced / *n_left (optional): nodes replaced /
 * memset nodes left as they are (two-dimensional ones). */
int sn_graph_replace_memsets(void *graph, int *n_replaced, int *n_left);

/* ------------------------------------------------------------------------------------------
 * Diagnostics (tools/ and the A/B parity tests; not part of the drop-in contract, no reference counterpart).
 * None of them changes a result.
 * ------------------------------------------------------------------------------------------ */
/* launch options of the token-stationary S1 screen: token-phase gate, balanced token map; 1 on, 0 off, -1 = from the
 * environment (SN_ASSIGN_GATE / SN_ASSIGN_BALANCE, default off) */
void sn_debug_set_assign_options(int gate, int balance);
/* resident workgroups per CU the runtime reports for the S1 screen kernel (D = 384) with `lds` bytes of dynamic LDS */
int sn_debug_screen_occupancy(int lds);
/* device buffers the S1 screen / instance-graph / GCN GEMM kernels write s_memtime stamps to (NULL = off): 16 x u64 per wave */
void sn_debug_set_stamps(void *device_buffer);
void sn_debug_set_graph_stamps(void *device_buffer);
void sn_debug_set_gemm_stamps(void *device_buffer);
/* sn_gcn_gemm's tile form (round 6): tile_rows 0 = chosen per call (256-row tiles, one 8-wave workgroup per CU, for m >= 256 with
 * one extent per batch; 128-row tiles, two 4-wave workgroups per CU, otherwise), 128 / 256 = forced; stagger 1 / 0 = waves 4-7 of a
 * 256-row workgroup half a stage behind waves 0-3 or in step with them; -1 leaves a setting as it is.  tile_rows 64 (chosen per call
 * for the fused next-layer product of graphs with per-graph extents) forces 64-row tiles for every fused next-layer product.  Initial values: environment
 * SN_GEMM_TM (0) and SN_GEMM_STAGGER (1).  Both forms give the same bits. */
void sn_debug_set_gemm_tile(int tile_rows, int stagger);

#ifdef __cplusplus
}
#endif
#endif /* SCHEMANET_HIP_H */
