// Instance IR-graph construction and the atlas-initialisation statistics, one workgroup per
// image.  HBM-bound gather/scatter work: the image's attention map is streamed once with
// coalesced row loads into LDS (153.7 KB of the CU's 160 KB at L = 196), soft-maxed per row by
// one wave with shuffle reductions, and every (word_i, word_j) cell is then summed out of LDS
// in exactly the reference's order (positions of word_i ascending x positions of word_j
// ascending, fp32 sequential from 0).  No MFMA: there is no contraction here.
//
// Reference being replaced (paths relative to /root/reference):
//   cpp_extension/src/large_scale_feat_to_v.cpp:41-143   ext::feat_to_instance_v
//   cpp_extension/src/large_scale_feat_to_e.cpp:33-150   ext::feat_to_instance_e
//   cpp_extension/src/feat_to_v_attr.cpp:19-148          ext::feat_to_v_attr
//   cpp_extension/src/feat_to_e.cpp:31-127               ext::feat_to_e
//   schema_inference/graph/schema_net.py:188-254, 278-356 (clamp / softmax / normalise / @ w)
//   schema_inference/utils/ingredient_model_wrapper.py:58-68 (head mean + slicing)
//   scripts/init_schema_net.py:33-35, 59-61 (per-class sums)
#include "sn_common.h"
#include "sn_assign_shared.h"
#ifndef SN_S3_ROWS_NT
#define SN_S3_ROWS_NT 1
#endif

#include <stdlib.h>
#include <type_traits>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kCellsPerLane = 4;               // 4 x 64 lanes = 256 output columns per row
constexpr int kMaxCols = kCellsPerLane * SN_WAVE;
constexpr int kTFloats = 512;                   // feat_h << ceil(log2(feat_w)) <= 2 L <= 392

__host__ __device__ inline size_t up16(size_t x) { return (x + 15) & ~size_t(15); }

// ------------------------------------------------------------------------------------------
// LDS carve shared by the three graph kernels
// ------------------------------------------------------------------------------------------
struct Lds {
    float *A;                  // [L, L] soft-maxed attention of this image (edges only)
    int64_t *words;            // [L]
    float *acls;               // [L] soft-maxed attention to the cls token
    float *T;                  // [512] grid similarity by (|drow| << tshift) + |dcol|
    int *rev;                  // [256] output row/col -> group (or -1)
    unsigned short *gstart;    // [L+1] group g owns pos_sorted[gstart[g] .. gstart[g+1])
    unsigned short *prc;       // [L] (row << 8) | col of each position on the feature grid
    unsigned char *pos_sorted; // [L] positions grouped by word (ascending), ascending inside
    unsigned char *flag;       // [L] first-occurrence flag / keep flag
    float *red;                // [64] reduction scratch
    int *misc;                 // [16]
    unsigned short *pless;     // [L] per position: number of kept positions holding a smaller word
    unsigned short *pcnt;      // [L] per position: occurrences of its word
    float *psum;               // [L] per first occurrence: attention sum of its word (position order)
    unsigned char *pgroup;     // [L] per first occurrence: index of its word among the sorted distinct words
};

__host__ __device__ inline size_t lds_bytes(int L, bool with_attn)
{
    size_t n = 0;
    if (with_attn) n += up16((size_t)L * L * 4);
    n += up16((size_t)L * 8) + up16((size_t)L * 4) + up16(kTFloats * 4) + up16(kMaxCols * 4);
    n += up16((size_t)(L + 1) * 2) + up16((size_t)L * 2) + up16(L) * 2 + up16(64 * 4) + up16(16 * 4);
    n += up16((size_t)L * 2) * 2 + up16((size_t)L * 4) + up16(L);
    return n;
}

__device__ inline Lds carve(unsigned char *p, int L, bool with_attn)
{
    Lds s;
    s.A = (float *)p;                 if (with_attn) p += up16((size_t)L * L * 4);
    s.words = (int64_t *)p;           p += up16((size_t)L * 8);
    s.acls = (float *)p;              p += up16((size_t)L * 4);
    s.T = (float *)p;                 p += up16(kTFloats * 4);
    s.rev = (int *)p;                 p += up16(kMaxCols * 4);
    s.gstart = (unsigned short *)p;   p += up16((size_t)(L + 1) * 2);
    s.prc = (unsigned short *)p;      p += up16((size_t)L * 2);
    s.pos_sorted = p;                 p += up16(L);
    s.flag = p;                       p += up16(L);
    s.red = (float *)p;               p += up16(64 * 4);
    s.misc = (int *)p;                p += up16(16 * 4);
    s.pless = (unsigned short *)p;    p += up16((size_t)L * 2);
    s.pcnt = (unsigned short *)p;     p += up16((size_t)L * 2);
    s.psum = (float *)p;              p += up16((size_t)L * 4);
    s.pgroup = p;
    return s;
}

// ------------------------------------------------------------------------------------------
// attention rows -> LDS.  One wave per row, 4 elements per lane (L <= 256), optional head mean
// (ingredient_model_wrapper.py:58-62), optional clamp + softmax (schema_net.py:334-336; an
// all-clamped row becomes NaN exactly like torch: (-inf) - (-inf)).
// src points at element (row 0, col 0) of head 0; rows are `stride_r` floats apart.
// ------------------------------------------------------------------------------------------
// Branch-free row load: out-of-range lanes read a clamped (valid) address and are masked later.
template <bool kVec>
__device__ __forceinline__ void load_row4(const float *row, int L, int lane, float x[4])
{
    if (kVec) {                     // L % 4 == 0 and 16-byte aligned rows: one dwordx4 per lane
        const int c = min(lane * 4, L - 4);
        typedef float f32x4_nt __attribute__((ext_vector_type(4)));
#if SN_S3_ROWS_NT
        const f32x4_nt v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt *>(row + c));     // read once: keep it out of the other kernels' L2
#else
        const f32x4_nt v = *reinterpret_cast<const f32x4_nt *>(row + c);
#endif
        x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = row[min(lane + SN_WAVE * k, L - 1)];
    }
}

template <bool kVec>
__device__ __forceinline__ int col_of(int lane, int k) { return kVec ? lane * 4 + k : lane + SN_WAVE * k; }

// 64-lane reductions for the prediction path: DPP-fused max / add (one instruction per step; hipcc emits a v_mov_dpp, a
// canonicalising max and the operation for each step of sn_wave_max), four steps inside the 16-lane rows, row_bcast:15 /
// row_bcast:31 across them, the total read from lane 63.  (s_nop 1: a DPP operand written by the previous VALU instruction
// needs two wait states.)  The sum's tree order differs from sn_wave_sum's: fp32 rounding, prediction path only.
__device__ __forceinline__ float wave_max_fast(float v)
{
    asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return SN_READLANE_F32(v, 63);
}

__device__ __forceinline__ float wave_sum_fast(float v)
{
    asm volatile("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[3,2,1,0] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1" : "+v"(v));
    return SN_READLANE_F32(v, 63);
}

// kFast (prediction path only): v_exp_f32 on (x - m) * log2(e) and one reciprocal per row instead of
// libm-grade expf and a correctly rounded division per element: ~1e-6 relative on the probabilities,
// inside the 1e-5 budget of the instance graph; the init statistics (bit-exact against the reference)
// keep the exact form.
template <bool kVec, bool kFast = false>
__device__ __forceinline__ void softmax_row4(float x[4], int L, int lane, bool use_clamp, float clamp)
{
    if constexpr (kVec && kFast) {
        // The row phase issues ~65 VALU instructions per row in the general form below and is bound by them.  Here a lane's
        // four columns are valid together (4 lane < L, L a multiple of 4): masked and clamped elements become -inf once, and
        // everything behind that needs no selects - exp2(-inf) is 0; (x - m) log2(e) is one packed fma per column pair
        // (x log2(e) - m log2(e): 3e-7 relative on the probabilities); ~40 instructions.
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const bool lane_ok = 4 * lane < L;
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = (!lane_ok || (use_clamp && x[k] < clamp)) ? -INFINITY : x[k];
        const float m = wave_max_fast(fmaxf(fmaxf(x[0], x[1]), fmaxf(x[2], x[3])));
        constexpr float kLog2e = 1.44269504088896340736f;
        const float nm = -m * kLog2e;                 // all-clamped row: m = -inf, x log2(e) + nm = NaN, like torch ((-inf) - (-inf))
        const f32x2 a01 = __builtin_elementwise_fma(f32x2{x[0], x[1]}, f32x2{kLog2e, kLog2e}, f32x2{nm, nm});
        const f32x2 a23 = __builtin_elementwise_fma(f32x2{x[2], x[3]}, f32x2{kLog2e, kLog2e}, f32x2{nm, nm});
        f32x2 e01 = {__builtin_amdgcn_exp2f(a01.x), __builtin_amdgcn_exp2f(a01.y)};
        f32x2 e23 = {__builtin_amdgcn_exp2f(a23.x), __builtin_amdgcn_exp2f(a23.y)};
        const f32x2 s2 = e01 + e23;
        const float r = __builtin_amdgcn_rcpf(wave_sum_fast(s2.x + s2.y));
        e01 = e01 * f32x2{r, r};
        e23 = e23 * f32x2{r, r};
        x[0] = e01.x; x[1] = e01.y; x[2] = e23.x; x[3] = e23.y;
        return;
    }
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool ok = col_of<kVec>(lane, k) < L;
        x[k] = (ok && use_clamp && x[k] < clamp) ? -INFINITY : x[k];
        m = ok ? fmaxf(m, x[k]) : m;
    }
    m = kFast ? wave_max_fast(m) : sn_wave_max(m);
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const bool ok = col_of<kVec>(lane, k) < L;
        x[k] = ok ? (kFast ? __builtin_amdgcn_exp2f((x[k] - m) * 1.44269504088896340736f) : expf(x[k] - m)) : 0.0f;
        s += x[k];
    }
    s = kFast ? wave_sum_fast(s) : sn_wave_sum(s);
    if (kFast) {
        const float r = __builtin_amdgcn_rcpf(s);    // (1 ulp; a correctly rounded 1 / s is ten instructions)  all-clamped row: m = -inf -> x = NaN, s = NaN, like torch
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = x[k] * r;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) x[k] = x[k] / s;
    }
}

struct NoPreWork { __device__ __forceinline__ void operator()() const {} };

// ------------------------------------------------------------------------------------------
// Deferred finish of S1 inside the row phase (sn_assign_words mode 2, sn_graph_args.rerank).
// The screen has left a flag word per token and, for the tokens it could not decide, the codes of the candidate words
// (csrc/sn_assign_shared.h).  The stand-alone re-rank is a launch of its own on the critical path of a prediction - 13-19 us
// of dependent fetches for ~13 tokens of an image - while this kernel's row waves sit out the HBM burst of the attention
// maps.  Here row wave w OWNS the positions l = w, w + 15, ...: it requests their flag words and candidate records in
// front of its first rows (one round trip, together with them), then - its first two batches of rows landed - the token row
// and the first candidate pair of up to three flagged tokens together (a second round trip, with its next rows in flight
// and its first batch soft-maxed meanwhile), scores them in fp64, further pairs / tokens one by one (rare), and writes the
// final words (to the ingredients tensor and to s.words).  The arithmetic is the stand-alone kernel's
// (assign_rerank_kernel): fp64, the oracle's summation order, lowest index on ties - the same ids bit for bit.  The sorting
// wave waits for the fifteen `done` signals (a counter in LDS) before it sorts, and finishes the overflow tokens (a handful
// per 50 000) itself meanwhile: rerank_overflow_token.  NT = D / 64.
// (Measured and dropped, tools/time_defer.py: (i) one token moved one stage per iteration of the row loop - hipcc waits for a
// loop-carried load with vmcnt(0), i.e. for the rows just requested: one batch in flight per wave instead of three, rows in
// at 28-33 k cycles instead of 21 k, the finish at 31 k, the launch 43-45 us against 31.7; (ii) five waves finishing the
// whole image before they join the row dealing: three tokens in flight, two rounds - every round trip is ~6 k cycles under
// the burst - the sorting wave starts at 24 k instead of 11 k: 39.7 us.)
// ------------------------------------------------------------------------------------------
struct NoTick {
    __device__ __forceinline__ void begin() {}
    __device__ __forceinline__ void request(bool) {}
    __device__ __forceinline__ void complete() {}
};

constexpr int kRowWaves = 15;           // the sixteen-wave kernel: waves 0 .. 14 bring rows in, wave 15 sorts
constexpr int kOwnMax = 14;             // positions a row wave owns in the deferred S1 finish: ceil(L / 15) (L <= 210)
constexpr int kFinishItems = 4;         // candidate pairs a row wave has in flight at once (0.9 flagged tokens, ~1.1 pairs expected per wave)
// ... at D = 384 (NT = 6: 18 registers per pair) two: with four the kernel's 128 registers did not hold them - 79 spilled registers,
// 88 B of scratch per lane, and the parked rows came back from scratch right in front of their fp64 products.  Two in flight: 114
// registers, no scratch, the kernel 38.0 us instead of 40.9 (tools/time_defer.py); a third / fourth pair of a wave (rare) is scored
// one at a time behind them.  D = 768 (NT = 12, round 5: config [3]'s DeiT-Base tokens): one pair in flight - 120 registers, no scratch.
template <int NT> constexpr int finish_items() { return NT >= 12 ? 1 : (NT >= 6 ? 2 : kFinishItems); }

template <int NT>
struct RerankWave {
    sn_s1::RerankView rv;
    int b, L, lane, wid;
    int64_t *words;             // s.words (LDS): final word of every flagged position, -1 when none could be ranked
    int *done_counter;          // s.misc[6]
    unsigned long long *stamp;  // diagnostics (sn_debug_set_graph_stamps): slots 14 / 15 of the image = wave 3's requests out / finish complete

    // One work item = one candidate pair of one owned token: the token's row and the two words' rows (a token with more than
    // two candidates is several items).  A token's running best lives in LANE jj (its index among the owned positions) of
    // three registers, its remaining candidates in lane jj of `cmv`: items are scored in any grouping, v_readlane and a
    // select on the token's lane pick its state.
    struct Item {
        int jj, ma, mb;
        bool two;
        float xf[NT], ra[NT], rb[NT];
    };
    unsigned f, cp0, cp1;       // lane j: flag word of position wid + 15 j;  lane t: dwords 2 (t % 3), + 1 of the record of owned position t / 3
    unsigned cmv;               // lane j: candidate slots of token j not yet requested
    unsigned best_lo, best_hi;  // lane j: best fp64 score of token j so far (+inf)
    int best_i;                 // lane j: its word (0x7fffffff: none)
    Item it[finish_items<NT>()];
    int cnt;

    __device__ __forceinline__ int64_t token_index(int pos) const { return (int64_t)b * rv.tsb + (int64_t)pos * rv.tsl; }

    __device__ __forceinline__ void load_row(float (&dst)[NT], const void *row, int x_bf16) const
    {
        // (D == 64 NT: the host launches no other shape.  One branch per row: a select per element makes hipcc branch and wait per element.)
        if (x_bf16) {
            const unsigned short *r = reinterpret_cast<const unsigned short *>(row);
            unsigned short u[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) u[t] = r[lane + SN_WAVE * t];
#pragma unroll
            for (int t = 0; t < NT; ++t) dst[t] = __uint_as_float((unsigned)u[t] << 16);
        } else {
            const float *r = reinterpret_cast<const float *>(row);
#pragma unroll
            for (int t = 0; t < NT; ++t) dst[t] = r[lane + SN_WAVE * t];
        }
    }

    // stage 1, in front of everything: flag words and candidate records of the owned positions (one round trip)
    __device__ __forceinline__ void begin()
    {
        f = 0u; cp0 = 0u; cp1 = 0u; cnt = 0;
        const int pj = wid + kRowWaves * lane;
        if (lane < kOwnMax && pj < L) f = rv.flags[token_index(pj)];
        const int pt = wid + kRowWaves * (lane / 3);
        if (lane < 3 * kOwnMax && pt < L) {
            const uint2 v = *reinterpret_cast<const uint2 *>(rv.codes + token_index(pt) * sn_s1::kCodeBytes + 8 * (lane % 3));
            cp0 = v.x; cp1 = v.y;
        }
    }

    // the next pair of the lowest pending token -> item `q` (false: nothing pending)
    __device__ __forceinline__ bool issue_item(Item &q)
    {
        const unsigned long long pend = __ballot(cmv != 0u);
        if (!pend) return false;                                     // wave-uniform
        const int jj = __ffsll((long long)pend) - 1;
        unsigned cm = (unsigned)__builtin_amdgcn_readlane((int)cmv, jj);
        const int ca = __ffs((int)cm) - 1;
        cm &= cm - 1;
        q.two = cm != 0u;
        const int cb2 = q.two ? __ffs((int)cm) - 1 : ca;
        if (q.two) cm &= cm - 1;
        cmv = lane == jj ? cm : cmv;
        // word of candidate slot `lane` (< 24): byte `lane` of the record, held as dword pairs by lanes 3 jj .. 3 jj + 2
        const int src = 3 * jj + ((lane & 31) >> 3);
        const unsigned d0 = (unsigned)__shfl((int)cp0, src, SN_WAVE), d1 = (unsigned)__shfl((int)cp1, src, SN_WAVE);
        const unsigned code = (((lane >> 2) & 1) ? d1 : d0) >> (8 * (lane & 3)) & 0xFFu;
        const int my_word = lane < sn_s1::kMaxCand ? sn_s1::slot_word(lane, code) : 0;
        q.jj = jj;
        q.ma = __builtin_amdgcn_readlane(my_word, ca);
        q.mb = __builtin_amdgcn_readlane(my_word, cb2);
        const int l = wid + kRowWaves * jj;
        load_row(q.xf, sn_s1::token_row_ptr(rv.x, rv.x_bf16, (int64_t)b * rv.xsb + (int64_t)l * rv.xsl), rv.x_bf16);
        load_row(q.ra, rv.cb + (int64_t)q.ma * rv.D, 0);
        load_row(q.rb, rv.cb + (int64_t)q.mb * rv.D, 0);
        return true;
    }

    __device__ __forceinline__ void score_item(const Item &q)
    {
        double pa = 0.0, pb = 0.0;
#pragma unroll
        for (int k = 0; k < NT; ++k) { pa = fma((double)q.xf[k], (double)q.ra[k], pa); pb = fma((double)q.xf[k], (double)q.rb[k], pb); }
        const double sa = rv.cn64[q.ma] - 2.0 * sn_wave_sum_f64(pa);
        const double sb = rv.cn64[q.mb] - 2.0 * sn_wave_sum_f64(pb);
        double best = __hiloint2double(__builtin_amdgcn_readlane((int)best_hi, q.jj), __builtin_amdgcn_readlane((int)best_lo, q.jj));
        int bi = __builtin_amdgcn_readlane(best_i, q.jj);
        if (sa < best || (sa == best && q.ma < bi)) { best = sa; bi = q.ma; }
        if (q.two && (sb < best || (sb == best && q.mb < bi))) { best = sb; bi = q.mb; }
        if (lane == q.jj) { best_lo = (unsigned)__double2loint(best); best_hi = (unsigned)__double2hiint(best); best_i = bi; }
    }

    // stage 2: token and candidate rows of up to kFinishItems pairs requested
    __device__ __forceinline__ void request(bool)
    {
        cmv = (f >> 31) ? 0u : (f & 0xFFFFFFu);                      // (overflow tokens: the sorting wave's)
        best_lo = 0u; best_hi = 0x7FF00000u; best_i = 0x7fffffff;    // +inf
        cnt = 0;
#pragma unroll
        for (int i = 0; i < finish_items<NT>(); ++i) {
            if (!issue_item(it[i])) break;
            cnt = i + 1;
        }
        if (stamp && wid == 3 && lane == 0) stamp[14] = __builtin_amdgcn_s_memtime();
    }

    // stage 3: the requested pairs scored (straight-line); whatever is left (more than kFinishItems pairs: rare) one pair at a
    // time, blocking; every owned token's word written by its lane; then the `done` signal
    __device__ __forceinline__ void complete()
    {
#pragma unroll
        for (int i = 0; i < finish_items<NT>(); ++i) {
            if (i >= cnt) break;                                     // wave-uniform
            score_item(it[i]);
        }
        {
            Item q;
            while (issue_item(q)) score_item(q);
        }
        const int pj = wid + kRowWaves * lane;
        if (lane < kOwnMax && pj < L && f != 0u && (f >> 31) == 0u) {
            const int w = best_i != 0x7fffffff ? best_i : -1;        // (-1: keep the screen's word)
            if (w >= 0) rv.ids[(int64_t)b * rv.isb + (int64_t)pj * rv.isl] = w;
            words[pj] = w;
        }
        // tell the sorting wave (LDS operations of a wave are performed in order: the words are written)
        if (lane == 0) atomicAdd(done_counter, 1);
        if (stamp && wid == 3 && lane == 0) stamp[15] = __builtin_amdgcn_s_memtime();
    }
};

// `pre_work()`: register-only work of the caller that has nothing to do with the rows; the dynamic form calls it once its first
// two batches of loads are issued (the wave would otherwise sit out their HBM latency), the other forms up front.
// `tick`: the deferred S1 finish (RerankWave; NoTick: none): begin() in front of the first row loads, request() behind them
// (and the caller's register-only work), complete() behind the first batch's soft-max.
template <bool kVec, bool kFast, class Pre, class Tick>
__device__ __forceinline__ void attn_rows_to_lds_impl(float *A, const float *src, int64_t stride_r, int heads,
                                                      int64_t stride_h, int L, bool is_logits, bool use_clamp,
                                                      float clamp, int rb, int re, int rs, int lane, int *next_row,
                                                      Pre pre_work, Tick &tick)
{
    constexpr int kRowsInFlight = 7;        // HBM latency: 7 rows of loads in flight per wave (196 rows on 15 waves: two batches; four in flight = four batches of exposed latency: 25.6 k -> see DESIGN 3.2)
    tick.begin();
    if (rs != 0 || heads > 1) {
        pre_work();
        tick.request(true);         // (the static forms: the finish runs ahead of the rows; six heads of rows follow)
        tick.complete();
    }
    if (heads > 1) {
        // Head mean fused (the backbone's [bs, H, L+1, L+1] tap): the loads of ALL heads of a pair of rows are issued before
        // the first add - up to 6 heads x 2 rows = 12 row loads in flight per wave (one head after the other was a chain
        // of H dependent HBM round trips per row batch: 6 heads, 931 KB per image).  Heads are added in order 0, 1, 2, ...
        // as before (same sums bit for bit).
        constexpr int kHB = 6, kRB = 2;
        for (int r0 = rb; r0 < re; r0 += rs * kRB) {
            float x[kRB][4];
            for (int h0 = 0; h0 < heads; h0 += kHB) {
                float y[kHB][kRB][4];
#pragma unroll
                for (int hh = 0; hh < kHB; ++hh) {
                    const int hc = min(h0 + hh, heads - 1);         // (past the last head: a repeated load nobody adds)
#pragma unroll
                    for (int i = 0; i < kRB; ++i)
                        load_row4<kVec>(src + hc * stride_h + (int64_t)min(r0 + rs * i, L - 1) * stride_r, L, lane, y[hh][i]);
                }
#pragma unroll
                for (int hh = 0; hh < kHB; ++hh) {
                    if (h0 + hh >= heads) break;                     // wave-uniform
#pragma unroll
                    for (int i = 0; i < kRB; ++i)
#pragma unroll
                        for (int k = 0; k < 4; ++k) x[i][k] = (h0 + hh == 0) ? y[hh][i][k] : x[i][k] + y[hh][i][k];
                }
            }
#pragma unroll
            for (int i = 0; i < kRB; ++i) {
                const int r = r0 + rs * i;
#pragma unroll
                for (int k = 0; k < 4; ++k) x[i][k] = x[i][k] / (float)heads;
                if (is_logits) softmax_row4<kVec, kFast>(x[i], L, lane, use_clamp, clamp);
                if (r < re) {                                  // wave-uniform
                    if (kVec) {
                        if (lane * 4 < L) *reinterpret_cast<float4 *>(A + r * L + lane * 4) = make_float4(x[i][0], x[i][1], x[i][2], x[i][3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int c = lane + SN_WAVE * k;
                            if (c < L) A[r * L + c] = x[i][k];
                        }
                    }
                }
            }
        }
        return;
    }
    if (rs == 0) {
        // Dynamic batches (the sixteen-wave prediction kernel): a wave takes kDyn rows at a time from a counter in LDS and
        // has the next two batches' loads in flight while it soft-maxes this one (small batches: the phase ends one batch
        // after the last grab).  The row phase ends with the slowest wave,
        // and a static split loses both to the wave that shares its SIMD with the sorting wave and to whichever wave's
        // loads come back late (first barrier at 35-37 k cycles with every wave's own work done by 27 k).
        constexpr int kDyn = 2;                 // rows per batch; two batches of loads in flight behind the one being soft-maxed
        int r0 = rb, r1 = rb + kDyn;            // (the first two batches of a wave are fixed: rows 4 wid .. 4 wid + 3)
        float x[kDyn][4], y[kDyn][4];
        if constexpr (!std::is_same<Tick, NoTick>::value) {
            // The deferred S1 finish goes out AHEAD of the rows: a request issued behind the burst of 256 images' maps
            // queues for 6-10 k cycles (two dependent round trips behind it: the sorting wave started at 25 k cycles instead
            // of 11 k, tools/time_defer.py); in front of it the memory system is idle - the flag words come back, the token
            // and candidate rows are requested, and the rows right behind them; the caller's register-only work (4 k cycles
            // of table building) runs under all of it.
            tick.request(true);
        }
#pragma unroll
        for (int i = 0; i < kDyn; ++i) load_row4<kVec>(src + (int64_t)min(r0 + i, L - 1) * stride_r, L, lane, x[i]);
#pragma unroll
        for (int i = 0; i < kDyn; ++i) load_row4<kVec>(src + (int64_t)min(r1 + i, L - 1) * stride_r, L, lane, y[i]);
        pre_work();
        // one iteration: the next batch requested, the oldest soft-maxed and stored, [the deferred S1 finish completed], rotate
        auto iteration = [&](auto with_finish) {
            int rn = 0;
            if (lane == 0) rn = atomicAdd(next_row, kDyn);
            rn = __builtin_amdgcn_readfirstlane(rn);
            float z[kDyn][4];
            if (rn < L) {
#pragma unroll
                for (int i = 0; i < kDyn; ++i) load_row4<kVec>(src + (int64_t)min(rn + i, L - 1) * stride_r, L, lane, z[i]);
            }
#pragma unroll
            for (int i = 0; i < kDyn; ++i) {
                const int r = r0 + i;
                if (is_logits) softmax_row4<kVec, kFast>(x[i], L, lane, use_clamp, clamp);
                if (r < L) {                                   // wave-uniform
                    if (kVec) {
                        if (lane * 4 < L) *reinterpret_cast<float4 *>(A + r * L + lane * 4) = make_float4(x[i][0], x[i][1], x[i][2], x[i][3]);
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const int c = lane + SN_WAVE * k;
                            if (c < L) A[r * L + c] = x[i][k];
                        }
                    }
                }
            }
            // (behind the first soft-max: the finish's loads are older than this iteration's rows - a counted wait, straight-line code)
            if constexpr (decltype(with_finish)::value) tick.complete();
#pragma unroll
            for (int i = 0; i < kDyn; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) { x[i][k] = y[i][k]; y[i][k] = z[i][k]; }
            r0 = r1;
            r1 = rn;
        };
        // (the first iteration apart: inside the loop the finish's registers would be live around the whole row phase)
        if constexpr (!std::is_same<Tick, NoTick>::value) iteration(std::true_type{});       // (r0 < L: every wave has fixed first rows)
        while (r0 < L) iteration(std::false_type{});
        return;
    }
    for (int r0 = rb; r0 < re; r0 += rs * kRowsInFlight) {
        float x[kRowsInFlight][4];
#pragma unroll
        for (int i = 0; i < kRowsInFlight; ++i)
            load_row4<kVec>(src + (int64_t)min(r0 + rs * i, L - 1) * stride_r, L, lane, x[i]);
#pragma unroll
        for (int i = 0; i < kRowsInFlight; ++i) {
            const int r = r0 + rs * i;
            if (is_logits) softmax_row4<kVec, kFast>(x[i], L, lane, use_clamp, clamp);
            if (r < re) {                                  // wave-uniform
                if (kVec) {
                    if (lane * 4 < L) *reinterpret_cast<float4 *>(A + r * L + lane * 4) = make_float4(x[i][0], x[i][1], x[i][2], x[i][3]);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const int c = lane + SN_WAVE * k;
                        if (c < L) A[r * L + c] = x[i][k];
                    }
                }
            }
        }
    }
}

template <bool kFast = false, class Pre = NoPreWork, class Tick = NoTick>
__device__ __forceinline__ void attn_rows_to_lds(float *A, const float *src, int64_t stride_r, int heads,
                                        int64_t stride_h, int L, bool is_logits, bool use_clamp,
                                        float clamp, int rb, int re, int rs, int lane, int *next_row = nullptr, Pre pre_work = Pre(),
                                        Tick *tick = nullptr)
{
    // (rows need not be 16-byte aligned in global memory: gfx950 serves a dword-aligned global_load_dwordx4 correctly,
    // tools/unaligned_probe.hip; the slices of the backbone's [.., 197, 197] tap never are.  The LDS rows are: L % 4 == 0.)
    const bool vec = (L % 4 == 0) && (kFast || ((stride_r % 4 == 0) && (stride_h % 4 == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)));
    NoTick none;
    if constexpr (std::is_same<Tick, NoTick>::value) {
        if (vec) attn_rows_to_lds_impl<true, kFast, Pre, NoTick>(A, src, stride_r, heads, stride_h, L, is_logits, use_clamp, clamp, rb, re, rs, lane, next_row, pre_work, none);
        else attn_rows_to_lds_impl<false, kFast, Pre, NoTick>(A, src, stride_r, heads, stride_h, L, is_logits, use_clamp, clamp, rb, re, rs, lane, next_row, pre_work, none);
    } else {
        if (vec) attn_rows_to_lds_impl<true, kFast, Pre, Tick>(A, src, stride_r, heads, stride_h, L, is_logits, use_clamp, clamp, rb, re, rs, lane, next_row, pre_work, *tick);
        else attn_rows_to_lds_impl<false, kFast, Pre, Tick>(A, src, stride_r, heads, stride_h, L, is_logits, use_clamp, clamp, rb, re, rs, lane, next_row, pre_work, *tick);
    }
}

// ------------------------------------------------------------------------------------------
// grid similarity table (graph/utils.py:55-81): geo[p,q] = 1 / (1 + |grid_p - grid_q|_pow / alpha)
// depends only on (|drow|, |dcol|).  pow == 2 uses sqrt: integer inputs, correctly rounded
// sqrt / div => bit-identical to the reference's torch.cdist based table.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ int grid_shift(int feat_w)
{
    int sh = 0;
    while ((1 << sh) < feat_w) ++sh;
    return sh;
}

// entry (dr, dc) of the grid similarity table: the ONE expression every writer of s.T uses (same bits from every thread)
__device__ __forceinline__ float grid_table_entry(int dr, int dc, int feat_h, int feat_w, float alpha, float pw)
{
    float v = 0.0f;
    if (dr < feat_h && dc < feat_w) {
        float d;
        if (pw == 2.0f) d = sqrtf((float)(dr * dr + dc * dc));
        else d = powf(powf((float)dr, pw) + powf((float)dc, pw), 1.0f / pw);
        d = d / alpha;
        v = 1.0f / (1.0f + d);
    }
    return v;
}

__device__ inline void build_grid_T(const Lds &s, int L, int feat_w, float alpha, float pw, int tid)
{
    const int sh = grid_shift(feat_w), feat_h = L / feat_w;
    for (int i = tid; i < kTFloats; i += blockDim.x)
        s.T[i] = grid_table_entry(i >> sh, i & ((1 << sh) - 1), feat_h, feat_w, alpha, pw);
}

__device__ inline void build_grid_prc(const Lds &s, int L, int feat_w, int tid)
{
    if (tid < L) s.prc[tid] = (unsigned short)(((tid / feat_w) << 8) | (tid % feat_w));
}

__device__ inline void build_grid_table(const Lds &s, int L, int feat_w, float alpha, float pw, int tid)
{
    build_grid_T(s, L, feat_w, alpha, pw, tid);
    build_grid_prc(s, L, feat_w, tid);
}

__device__ inline float geo_at(const Lds &s, const float *geo, int L, int feat_w, int p, int q)
{
    if (geo) return geo[p * L + q];
    const int a = s.prc[p], b = s.prc[q];
    const int dr = abs((a >> 8) - (b >> 8)), dc = abs((a & 255) - (b & 255));
    return s.T[(dr << grid_shift(feat_w)) + dc];
}

// ------------------------------------------------------------------------------------------
// grouping: positions -> sorted distinct words ("groups", std::map order of the reference)
// thread p < L owns position p; kept == false positions are ignored (class restriction).
// Fills pos_sorted, gstart, misc[0] = number of groups.  Returns this position's group, its
// count, and whether it is the word's first occurrence.  attn-sum (position order) optional.
// ------------------------------------------------------------------------------------------
struct PosInfo { int group, cnt, first; float attn_sum; };

// Wave-parallel: every wave takes positions p = wid, wid+nw, ...; the 64 lanes hold the words of
// q = lane + 64k (k < 4, L <= 256) and count with ballots + popcounts:
//   cnt  = #{q : w_q == w_p}      rank = #{q < p : w_q == w_p}      less = #{q : w_q < w_p}
// so position p lands at pos_sorted[less + rank]; a second ballot pass over the first-occurrence
// flags gives the index of the word among the sorted distinct words.
__device__ inline PosInfo group_positions(const Lds &s, int L, int tid, bool kept, bool want_sum)
{
    const int lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    if (tid == 0) { s.misc[0] = 0; s.misc[1] = 0; }
    if (tid < L) s.flag[tid] = kept ? 1 : 0;
    __syncthreads();
    int64_t wq[4];
    bool vq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = lane + SN_WAVE * k;
        vq[k] = q < L && s.flag[q] != 0;
        wq[k] = vq[k] ? s.words[q] : 0;
    }
    if (wid == 0) {
        int total = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) total += __popcll(__ballot(vq[k]));
        if (lane == 0) s.misc[1] = total;
    }
    // The first occurrence of a word (no equal word at a smaller position) does the work for ALL positions of that
    // word: the lanes that hold them write their own records in parallel (index inside the word = mbcnt of the
    // equality masks); every other position stops after its four equality ballots.
    // It also marks them (flag bit 2): a wave that comes to a marked position skips it without any ballot.  (A mark
    // that is not visible yet only costs the ballots: the race is benign.  The phase is bound by the CU's one scalar
    // unit - ~70 scalar instructions per visited position from 16 waves - so skipped positions are what pays.)
    for (int p = wid; p < L; p += nw) {
        const int fp = __builtin_amdgcn_readfirstlane((int)s.flag[p]);
        if (fp == 0 || (fp & 4)) continue;
        const int64_t wp = s.words[p];
        const int ps = __builtin_amdgcn_readfirstlane(p);         // (scalar copy: the masks below stay on the scalar unit)
        unsigned long long eq[4];
        bool first = true;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            eq[k] = __ballot(vq[k] && wq[k] == wp);
            const int lo = ps - SN_WAVE * k;                      // bits below position p in chunk k
            const unsigned long long below = lo >= SN_WAVE ? ~0ull : (lo > 0 ? ((1ull << lo) - 1ull) : 0ull);
            first = first && (eq[k] & below) == 0ull;
        }
        if (!first) continue;                                     // (wave-uniform)
        int cnt = 0, less = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            less += __popcll(__ballot(vq[k] && wq[k] < wp));
            cnt += __popcll(eq[k]);
        }
        float sum = 0.0f;
        if (want_sum) {                                           // position order (utils.cpp:9)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                unsigned long long m = eq[k];
                while (m) {
                    const int bit = __ffsll((long long)m) - 1;
                    sum = sum + s.acls[SN_WAVE * k + bit];
                    m &= m - 1;
                }
            }
        }
        int base = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if ((eq[k] >> lane) & 1ull) {                         // this lane's position lane + 64 k holds the word
                const int q = lane + SN_WAVE * k;
                const int idx = base + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(eq[k] >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)eq[k], 0u));
                s.pless[q] = (unsigned short)less;
                s.pcnt[q] = (unsigned short)cnt;
                s.pos_sorted[less + idx] = (unsigned char)q;
                s.flag[q] = (unsigned char)(1 | (idx == 0 ? 2 : 0) | 4);
                if (idx == 0) s.psum[q] = sum;
            }
            base += __popcll(eq[k]);
        }
    }
    __syncthreads();
    bool fq[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = lane + SN_WAVE * k;
        fq[k] = vq[k] && (s.flag[q] & 2) != 0;
    }
    for (int p = wid; p < L; p += nw) {
        if (!(__builtin_amdgcn_readfirstlane((int)s.flag[p]) & 2)) continue;
        const int64_t wp = s.words[p];
        int g = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) g += __popcll(__ballot(fq[k] && wq[k] < wp));
        if (lane == 0) {
            s.pgroup[p] = (unsigned char)g;
            s.gstart[g] = s.pless[p];
            atomicMax(&s.misc[0], g + 1);
        }
    }
    __syncthreads();
    if (tid == 0) s.gstart[s.misc[0]] = (unsigned short)s.misc[1];
    __syncthreads();
    PosInfo r = {0, 0, 0, 0.0f};
    if (tid < L && kept) {
        r.first = (s.flag[tid] & 2) != 0;
        r.cnt = s.pcnt[tid];
        r.group = r.first ? s.pgroup[tid] : 0;
        r.attn_sum = r.first ? s.psum[tid] : 0.0f;
    }
    return r;
}

// The same records by ONE wave and a sort (prediction path; every position kept): the 64 lanes hold four keys each,
// key = word << 8 | position (words below 2^24, L <= 256), a bitonic network over the 256 slots (36 compare-exchange
// stages: the strides below 4 between a lane's own registers, the others one ds_bpermute per key) leaves them
// ascending, i.e. grouped by word with ascending positions inside a word - the order of the reference's std::map and
// of its position lists - and everything else is local: a word starts where the word part changes (head), its index
// is the number of heads before it (ballots), its run ends at the next head.  ~600 wave-instructions instead of the
// ~36 000 of the ballot form above, and the other fifteen waves spend the time on the attention rows.
// `w4[e]` / key slot e of lane l: position l + 64 e on entry.  Returns false (nothing written) when a word does not
// fit the key; the caller then runs group_positions.
struct SortedGroups {                 // what a lane of the sorting wave knows about its four sorted slots (4 lane + e)
    bool head[4];                     // first position of its word
    int g[4], cnt[4];                 // group index, positions of the word
    float sum[4];                     // heads: cls-attention sum of the word in position order
    unsigned key[4];                  // word << 8 | position (0xFFFFFFFF: empty slot)
    int n_groups;
};

// kLean (the prediction kernel's hand-over, see the kernel): nobody reads the per-position records - the vertices are written
// from `sg_out`, the row map is the identity - so they are not written (their storage holds the signed grid table by then),
// the cls attention in sorted order is staged in `tmp` instead of s.psum, and s.flag receives what the edges phase wants
// there: position -> sorted index.
template <bool kLean = false>
__device__ __forceinline__ bool group_positions_sorted(const Lds &s, int L, int lane, const int64_t (&w4)[4], bool want_sum, SortedGroups &sg_out,
                                                       float *tmp = nullptr)
{
    float *const stage = kLean ? tmp : s.psum;
    unsigned key[4];
    bool fits = true;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int q = lane + SN_WAVE * e;
        const bool ok = q < L;
        fits = fits && (!ok || (w4[e] >= 0 && w4[e] < (1 << 24) - 1));     // (word 2^24 - 1 at position 255 would be the empty-slot key)
        key[e] = ok ? (((unsigned)w4[e] << 8) | (unsigned)q) : 0xFFFFFFFFu;
    }
    if (__any(!fits)) return false;
    // ---- bitonic sort, element index i = 4 lane + e
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 4) {
                const int lj = j >> 2;
                const bool lower = (lane & lj) == 0;
                const bool up = k == 256 || (lane & (k >> 2)) == 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const unsigned other = (unsigned)__shfl_xor((int)key[e], lj, SN_WAVE);
                    key[e] = (lower == up) ? min(key[e], other) : max(key[e], other);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e & j) continue;
                    const bool up = k == 2 ? (e & 2) == 0 : (k == 4 ? (lane & 1) == 0 : (k == 256 || (lane & (k >> 2)) == 0));
                    const unsigned lo = min(key[e], key[e | j]), hi = max(key[e], key[e | j]);
                    key[e] = up ? lo : hi;
                    key[e | j] = up ? hi : lo;
                }
            }
        }
    }
    // ---- heads, group indices
    const unsigned prevk = (unsigned)__shfl_up((int)key[3], 1, SN_WAVE);
    bool valid[4], head[4];
    unsigned long long hm[4];
    int n_kept = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        valid[e] = key[e] != 0xFFFFFFFFu;
        const unsigned pk = e == 0 ? prevk : key[e - 1];
        head[e] = valid[e] && ((e == 0 && lane == 0) || (key[e] >> 8) != (pk >> 8));
        hm[e] = __ballot(head[e]);
        n_kept += __popcll(__ballot(valid[e]));
    }
    const unsigned long long below = (1ull << lane) - 1ull;
    int g[4];
    int run = __popcll(hm[0] & below) + __popcll(hm[1] & below) + __popcll(hm[2] & below) + __popcll(hm[3] & below) - 1;
#pragma unroll
    for (int e = 0; e < 4; ++e) { run += head[e] ? 1 : 0; g[e] = run; }
    const int n_groups = __popcll(hm[0]) + __popcll(hm[1]) + __popcll(hm[2]) + __popcll(hm[3]);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (!valid[e]) continue;
        const int i = 4 * lane + e, pos = (int)(key[e] & 255u);
        s.pos_sorted[i] = (unsigned char)pos;
        if (head[e]) s.gstart[g[e]] = (unsigned short)i;
        if (want_sum) stage[i] = s.acls[pos];                     // (temporarily: cls attention in sorted order)
    }
    if (lane == 0) { s.gstart[n_groups] = (unsigned short)n_kept; s.misc[0] = n_groups; s.misc[1] = n_kept; }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float sum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    int gs[4], cnt[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        gs[e] = 0; cnt[e] = 0;
        if (!valid[e]) continue;
        gs[e] = s.gstart[g[e]];
        cnt[e] = (int)s.gstart[g[e] + 1] - gs[e];
        if (want_sum && head[e])
            for (int t = 0; t < cnt[e]; ++t) sum[e] = sum[e] + stage[gs[e] + t];       // position order (utils.cpp:9)
    }
    __builtin_amdgcn_wave_barrier();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");             // every sum is in registers: psum can take its final contents
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        if (!valid[e]) continue;
        const int pos = (int)(key[e] & 255u);
        if (kLean) { s.flag[pos] = (unsigned char)(4 * lane + e); stage[pos] = 1.0f / (float)cnt[e]; continue; }      // (stage: now 1 / positions of the position's word - the edges phase's per-position weight)
        s.pless[pos] = (unsigned short)gs[e];
        s.pcnt[pos] = (unsigned short)cnt[e];
        s.flag[pos] = (unsigned char)(1 | (head[e] ? 2 : 0) | 4);
        if (head[e]) { s.pgroup[pos] = (unsigned char)g[e]; s.psum[pos] = sum[e]; }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        sg_out.head[e] = head[e]; sg_out.g[e] = g[e]; sg_out.cnt[e] = cnt[e]; sg_out.sum[e] = sum[e]; sg_out.key[e] = key[e];
    }
    sg_out.n_groups = n_groups;
    return true;
}

// block-wide max over first-occurrence values; NaN propagates like at::max (large_scale_feat_to_v.cpp:124)
__device__ inline float block_max_nan(const Lds &s, float v, bool valid, int tid, int wid, int nw, int lane)
{
    float m = valid && !(v != v) ? v : -INFINITY;
    const int has_nan = __any(valid && (v != v));
    m = sn_wave_max(m);
    __syncthreads();
    if (lane == 0) { s.red[wid] = m; s.red[32 + wid] = has_nan ? 1.0f : 0.0f; }
    __syncthreads();
    float out = -INFINITY, nanf_ = 0.0f;
    for (int i = 0; i < nw; ++i) { out = fmaxf(out, s.red[i]); nanf_ += s.red[32 + i]; }
    return nanf_ > 0.0f ? NAN : out;
}

// Edge cells.  A lane owns up to 4 output columns for every row its wave processes, so the
// positions of its columns' words are cached in registers once (<= 4 positions per word; longer
// lists fall back to LDS).  One cell = sequential fp32 sums over (p in group gi) x (q in group
// gj), p-major (large_scale_feat_to_e.cpp:99-125).
constexpr int kQCache = 4;

struct ColCache {
    int cnt[kCellsPerLane];                 // positions of the column's word (0 = no column)
    int ja[kCellsPerLane];                  // start in pos_sorted
    int q4[kCellsPerLane][kQCache];         // byte offset of the position inside an attention row
    int qr[kCellsPerLane][kQCache];         // grid row of the position
    int qc4[kCellsPerLane][kQCache];        // 4 * grid column
    int tmax[kCellsPerLane];                // wave-uniform: largest cnt of chunk k over the 64 lanes
};

__device__ inline void cache_columns(const Lds &s, const int (&gj)[kCellsPerLane], ColCache &cc)
{
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) {
        cc.cnt[k] = 0; cc.ja[k] = 0;
        if (gj[k] >= 0) { cc.ja[k] = s.gstart[gj[k]]; cc.cnt[k] = s.gstart[gj[k] + 1] - cc.ja[k]; }
        int tm = 0;
#pragma unroll
        for (int t = 0; t < kQCache; ++t) {
            const int q = t < cc.cnt[k] ? s.pos_sorted[cc.ja[k] + t] : 0;
            const int rc = s.prc[q];
            cc.q4[k][t] = q * 4;
            cc.qr[k][t] = rc >> 8;
            cc.qc4[k][t] = (rc & 255) * 4;
        }
        tm = (int)sn_wave_max((float)cc.cnt[k]);
        cc.tmax[k] = tm;
    }
}

template <bool kGrid>
__device__ __forceinline__ void row_cells_impl(const Lds &s, const float *geo, int L, int feat_w, int gi, const ColCache &cc,
                                               bool any_long, int mean, float (&c0)[kCellsPerLane], float (&c1)[kCellsPerLane])
{
    const int ia = __builtin_amdgcn_readfirstlane((int)s.gstart[gi]);
    const int ib = __builtin_amdgcn_readfirstlane((int)s.gstart[gi + 1]);
    const int tsh = grid_shift(feat_w) + 2;                 // byte shift of a table row
    float sa[kCellsPerLane], sg[kCellsPerLane];
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) { sa[k] = 0.0f; sg[k] = 0.0f; }
    for (int x = ia; x < ib; ++x) {
        const int p = __builtin_amdgcn_readfirstlane((int)s.pos_sorted[x]);
        const int prc = __builtin_amdgcn_readfirstlane((int)s.prc[p]);
        const unsigned pr = prc >> 8, pc4 = (prc & 255) * 4;
        const char *arow = reinterpret_cast<const char *>(s.A + p * L);
        const char *tbase = reinterpret_cast<const char *>(s.T);
        const float *grow = kGrid ? nullptr : geo + (int64_t)p * L;
        (void)any_long;
#pragma unroll
        for (int k = 0; k < kCellsPerLane; ++k) {
#pragma unroll
            for (int t = 0; t < kQCache; ++t) {             // positions cached in registers
                if (t >= cc.tmax[k]) break;                 // wave-uniform: nobody has a t-th position
                const bool on = t < cc.cnt[k];
                float av = *reinterpret_cast<const float *>(arow + cc.q4[k][t]);
                float gv;
                if (kGrid) {
                    const unsigned dr = __usad(pr, (unsigned)cc.qr[k][t], 0u);
                    const unsigned off = (dr << tsh) + __usad(pc4, (unsigned)cc.qc4[k][t], 0u);
                    gv = *reinterpret_cast<const float *>(tbase + off);
                } else {
                    gv = grow[cc.q4[k][t] >> 2];
                }
                av = on ? av : 0.0f;                        // x + 0 == x: unused slots do not disturb the sum order
                gv = on ? gv : 0.0f;
                sa[k] = sa[k] + av;
                sg[k] = sg[k] + gv;
            }
            for (int t = kQCache; t < cc.tmax[k]; ++t) {    // words covering > 4 positions: list read from LDS
                const bool on = t < cc.cnt[k];
                const int q = on ? (int)s.pos_sorted[cc.ja[k] + t] : 0;
                float av = *reinterpret_cast<const float *>(arow + q * 4);
                float gv;
                if (kGrid) {
                    const int rc = s.prc[q];
                    const unsigned dr = __usad(pr, (unsigned)(rc >> 8), 0u);
                    const unsigned off = (dr << tsh) + __usad(pc4, (unsigned)((rc & 255) * 4), 0u);
                    gv = *reinterpret_cast<const float *>(tbase + off);
                } else {
                    gv = grow[q];
                }
                av = on ? av : 0.0f;
                gv = on ? gv : 0.0f;
                sa[k] = sa[k] + av;
                sg[k] = sg[k] + gv;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) {
        const bool has = cc.cnt[k] > 0;
        const float n = (float)((ib - ia) * (has ? cc.cnt[k] : 1));
        c0[k] = has ? (mean ? sg[k] / n : sg[k]) : 0.0f;
        c1[k] = has ? (mean ? sa[k] / n : sa[k]) : 0.0f;
    }
}

__device__ inline void row_cells(const Lds &s, const float *geo, int L, int feat_w, int gi, const ColCache &cc,
                                 bool any_long, int mean, float (&c0)[kCellsPerLane], float (&c1)[kCellsPerLane])
{
    if (geo) row_cells_impl<false>(s, geo, L, feat_w, gi, cc, any_long, mean, c0, c1);
    else row_cells_impl<true>(s, geo, L, feat_w, gi, cc, any_long, mean, c0, c1);
}

// ------------------------------------------------------------------------------------------
// S2 + S3 kernel
// ------------------------------------------------------------------------------------------
static unsigned long long *g_graph_stamps = nullptr;      // diagnostics only (sn_debug_set_graph_stamps)

#define SN_GSTAMP(slot)                                                                          \
    do {                                                                                          \
        if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + (slot)] = __builtin_amdgcn_s_memtime(); \
    } while (0)

// kFast: the configuration of the prediction path, fixed at compile time - edges weighted by w_e only (no [.., 2]
// attribute tensor), means, self loops kept, canonical rows (no dictionary), the built-in grid table in its signed form, the
// zero padding of the edges left to the consumer, L a multiple of 4 and at most 256.  The edges phase is bound by the
// VALU instructions it issues (4 SIMD-cycles each): every run-time option is a branch, a select or a scalar that lives
// in a spilled SGPR (a v_readlane per use) in the row loop.
// RR > 0: the deferred finish of S1 rides in the row phase (RerankWave<RR>, RR = ceil(D / 64); sixteen waves only).
template <bool kEdges, bool kFast = false, int RR = 0>
__global__ __launch_bounds__(1024) void instance_graph_kernel(const sn_graph_args a, unsigned long long *stamps, int signed_table,
                                                              const sn_s1::RerankView rv)
{
    const bool c_mean = kFast || a.mean != 0;
    const bool c_rsl = !kFast && a.remove_self_loop != 0;
    const bool c_skip = kFast || a.skip_edge_padding != 0;
    float *const c_out_e2 = kFast ? nullptr : a.out_e2;
    const float *const c_geo = kFast ? nullptr : a.geo;
    const int64_t *const c_dict = kFast ? nullptr : a.dict_keys;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int L = a.L, b = blockIdx.x, tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    const Lds s = carve(smem, L, kEdges);
    const bool do_v = a.attn_cls != nullptr;

    SN_GSTAMP(0);
    // ---- stream this image's attention map into LDS (the only large HBM read)
    // The last wave groups the positions by word (one-wave sort, group_positions_sorted) while the others bring the rows in.
    const bool sorter = nw > 1 && wid == nw - 1;                // wave-uniform
    const bool dyn_rows = kEdges && nw == 16 && a.attn_heads <= 1;          // block-uniform
    // lean hand-over (prediction kernel, sixteen waves, one head): the sorting wave also prepares what the edges phase needs -
    // the identity row map, the inverse of pos_sorted, the signed grid table, the edge-row counter - while the others are
    // still bringing rows in; behind the first barrier nobody rebuilds them (two barriers, a table build with two integer
    // divisions per thread and the record reads of 1 024 threads used to sit between the phases: 2.7 k cycles per image)
    const int ts_S = 2 * a.feat_w - 1;
    const bool lean_ok = kFast && dyn_rows && ts_S <= SN_WAVE;
    if (dyn_rows || RR > 0) {
        // ([4]: next attention row to deal - behind the fixed first rows; [6]: row waves whose share of the deferred S1 finish is done)
        if (tid == 0) { s.misc[4] = (nw - 1) * 4; s.misc[6] = 0; }
        for (int c = tid; c < kMaxCols; c += blockDim.x) s.rev[c] = -1;       // (in front of the barrier: ordered before the sorting wave's row map)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                            // (everybody is at the start of the kernel: a cheap barrier)
    }
    // (one wave against fifteen on an issue-bound CU: without priority its ~2 500 instructions - cls soft-max, sort, records,
    // vertices - last longer than the row phase of the others, and everybody waits for it at the first barrier)
    if (sorter) __builtin_amdgcn_s_setprio(3);
    if (kEdges) {
        if (!sorter) {
            // rows of this wave: strided over the row waves, or - sixteen waves, one head - in batches of four from a
            // counter (s.misc[4]; the first batch of every wave is fixed, the counter starts behind them)
            int rb = wid, re = L, rs = nw > 1 ? nw - 1 : 1;
            if (dyn_rows) { rb = wid * 4; rs = 0; }
            // lean form: this wave's rows of the signed grid table (straight from the expression of the unsigned one,
            // grid_table_entry: the same floats; over the grouping scratch nobody uses in the lean form; rows fh - 1 +- w of the table,
            // lane = column), computed while its first loads are on their way
            auto table_rows = [&]() {
                if (!(lean_ok && !a.geo)) return;
                build_grid_prc(s, L, a.feat_w, tid);             // (positions -> grid coordinates: an integer division per thread)
                if (lane >= ts_S) return;
                float *ts = reinterpret_cast<float *>(s.pless);
                const int fh = L / a.feat_w;
                const int dc = lane - (a.feat_w - 1), adc = dc < 0 ? -dc : dc;
                for (int dr = wid; dr < fh; dr += nw - 1) {       // table rows fh - 1 +- dr hold the same values
                    const float v = grid_table_entry(dr, adc, fh, a.feat_w, a.dist_alpha, a.dist_pow);
                    ts[(fh - 1 + dr) * ts_S + lane] = v;
                    ts[(fh - 1 - dr) * ts_S + lane] = v;
                }
            };
            if constexpr (RR > 0) {
                RerankWave<RR> tick;
                tick.rv = rv; tick.b = b; tick.L = L; tick.lane = lane; tick.wid = wid; tick.words = s.words; tick.done_counter = &s.misc[6];
                tick.stamp = stamps ? stamps + (size_t)blockIdx.x * 16 : nullptr;
                attn_rows_to_lds<true, decltype(table_rows), RerankWave<RR>>(s.A, a.attn + (int64_t)b * a.attn_stride_b, a.attn_stride_r, a.attn_heads,
                                       a.attn_stride_h, L, a.attn_is_logits != 0, a.use_clamp_e != 0, a.clamp_e, rb, re, rs, lane, &s.misc[4],
                                       table_rows, &tick);
            } else {
                attn_rows_to_lds<true>(s.A, a.attn + (int64_t)b * a.attn_stride_b, a.attn_stride_r, a.attn_heads,
                                       a.attn_stride_h, L, a.attn_is_logits != 0, a.use_clamp_e != 0, a.clamp_e, rb, re, rs, lane, &s.misc[4],
                                       table_rows);
            }
        }
        if (!a.geo) {
            // (lean form: grid coordinates and signed table are written under the row waves' first loads, the unsigned table only
            // when the hand-over fails: behind the barrier)
            if (!lean_ok) build_grid_table(s, L, a.feat_w, a.dist_alpha, a.dist_pow, tid);
        }
    }
    // (RR: s.words is the sorting wave's - it holds every word, final ones included, once the row waves have finished theirs)
    if (RR == 0 && tid < L) s.words[tid] = a.ingredients[(int64_t)b * a.ing_stride_b + (int64_t)tid * a.ing_stride_l];
    if (!dyn_rows) for (int c = tid; c < kMaxCols; c += blockDim.x) s.rev[c] = -1;
    if (!sorter && wid == nw - 1 && lane == 0) s.misc[2] = 0;   // misc[2] = 1: the sorter wave has written the grouping records (only ever written by the last wave)

    // The sorting wave's inputs - the words, the vertex attribute weights - are requested before its cls-attention row is
    // soft-maxed: one global round trip instead of three dependent ones on the wave everybody waits for at the barrier.
    int64_t w4[4] = {0, 0, 0, 0};
    unsigned f4[4] = {0u, 0u, 0u, 0u};                          // (RR: flag words of the screen - non-zero: the word is still to come)
    float wv0 = 0.0f, wv1 = 0.0f;
    if (sorter && L <= 4 * SN_WAVE) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int q = lane + SN_WAVE * e;
            w4[e] = q < L ? a.ingredients[(int64_t)b * a.ing_stride_b + (int64_t)q * a.ing_stride_l] : 0;
            if (RR > 0 && q < L) f4[e] = rv.flags[(int64_t)b * rv.tsb + (int64_t)q * rv.tsl];
        }
        if (do_v && a.out_v) { wv0 = a.w_v[0]; wv1 = a.w_v[1]; }
    }
    // ---- attention to the cls token: clamp / softmax / nan_to_num(0)  (schema_net.py:295-297)
    if (do_v && wid == nw - 1) {
        const float *row = a.attn_cls + (int64_t)b * a.acls_stride_b;
        float x[4];
        load_row4<false>(row, L, lane, x);
        for (int h = 1; h < a.acls_heads; ++h) {
            float y[4];
            load_row4<false>(row + h * a.acls_stride_h, L, lane, y);
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] += y[k];
        }
        if (a.acls_heads > 1) {
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = x[k] / (float)a.acls_heads;
        }
        if (a.attn_cls_is_logits) {
            if (a.attn_cls_masked) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int c = lane + SN_WAVE * k;
                    if (c < L) a.attn_cls_masked[(int64_t)b * L + c] =
                        (a.use_clamp_v && x[k] < a.clamp_v) ? -INFINITY : x[k];
                }
            }
            softmax_row4<false>(x, L, lane, a.use_clamp_v != 0, a.clamp_v);      // (exact form in every configuration: the vertex weights do not depend on which kernel variant ran)
#pragma unroll
            for (int k = 0; k < 4; ++k) x[k] = sn_nan_to_num(x[k]);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + SN_WAVE * k;
            if (c < L) s.acls[c] = x[k];
        }
    }
    SortedGroups sg;
    bool done = false;                                          // (the sorting wave's: it grouped the positions)
    sg.n_groups = 0;
    if (sorter) {
        if constexpr (RR > 0) {
            // overflow tokens of this image (rare): finished here, while the row waves finish the flagged ones.  (ONE call
            // site: the four position slots of a lane are walked through uniform selects, not an unrolled loop)
            unsigned long long om0 = __ballot(lane < L && (f4[0] >> 31) != 0u), om1 = __ballot(lane + SN_WAVE < L && (f4[1] >> 31) != 0u);
            unsigned long long om2 = __ballot(lane + 2 * SN_WAVE < L && (f4[2] >> 31) != 0u), om3 = __ballot(lane + 3 * SN_WAVE < L && (f4[3] >> 31) != 0u);
            while (om0 | om1 | om2 | om3) {
                const int e = om0 ? 0 : (om1 ? 1 : (om2 ? 2 : 3));
                const unsigned long long me = e == 0 ? om0 : (e == 1 ? om1 : (e == 2 ? om2 : om3));
                const int src = __ffsll((long long)me) - 1;
                const unsigned long long rest = me & (me - 1);
                om0 = e == 0 ? rest : om0; om1 = e == 1 ? rest : om1; om2 = e == 2 ? rest : om2; om3 = e == 3 ? rest : om3;
                const unsigned fsel = e == 0 ? f4[0] : (e == 1 ? f4[1] : (e == 2 ? f4[2] : f4[3]));
                const unsigned fj = (unsigned)__builtin_amdgcn_readlane((int)fsel, src);
                const int q = src + SN_WAVE * e;
                const int64_t n = (int64_t)b * rv.tsb + (int64_t)q * rv.tsl;
                const unsigned code = lane < sn_s1::kMaxCand ? (unsigned)rv.codes[n * sn_s1::kCodeBytes + lane] : 0u;
                const int wfin = sn_s1::rerank_overflow_token<RR>(rv, b, q, lane, fj, lane < sn_s1::kMaxCand ? sn_s1::slot_word(lane, code) : 0);
                if (wfin >= 0) {
                    if (lane == src) {
                        w4[0] = e == 0 ? (int64_t)wfin : w4[0]; w4[1] = e == 1 ? (int64_t)wfin : w4[1];
                        w4[2] = e == 2 ? (int64_t)wfin : w4[2]; w4[3] = e == 3 ? (int64_t)wfin : w4[3];
                    }
                    if (lane == 0) rv.ids[(int64_t)b * rv.isb + (int64_t)q * rv.isl] = wfin;
                }
            }
            // the flagged tokens' final words: every row wave has done its share (fifteen signals), the words lie in s.words
            // (-1: none of the candidates could be ranked - keep the screen's)
            while (__hip_atomic_load(&s.misc[6], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < nw - 1) __builtin_amdgcn_s_sleep(2);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int q = lane + SN_WAVE * e;
                if (q < L) {
                    if (f4[e] != 0u && (f4[e] >> 31) == 0u) { const int64_t w = s.words[q]; if (w >= 0) w4[e] = w; }
                    s.words[q] = w4[e];
                }
            }
        }
        if (stamps && lane == 0) stamps[(size_t)blockIdx.x * 16 + 12] = __builtin_amdgcn_s_memtime();
        if (L <= 4 * SN_WAVE) {
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // (this wave's own s.acls stores)
        if (lean_ok) done = group_positions_sorted<true>(s, L, lane, w4, do_v, sg, s.T);      // (s.T: unused before the barrier in the lean form)
        else done = group_positions_sorted(s, L, lane, w4, do_v, sg);
        if (stamps && lane == 0) stamps[(size_t)blockIdx.x * 16 + 13] = __builtin_amdgcn_s_memtime();
        }
        if (lean_ok && done) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {                         // canonical rows: output row / column g = group g
                const int c = lane + SN_WAVE * k;
                if (c < sg.n_groups && c < kMaxCols) s.rev[c] = c;
            }
            if (lane == 0) s.misc[5] = nw;                        // next edge row to deal (see the row loop)
        }
        if (lane == 0) s.misc[2] = done ? 1 : 0;
        __builtin_amdgcn_s_setprio(0);
        if (stamps && lane == 0) stamps[(size_t)blockIdx.x * 16 + 10] = __builtin_amdgcn_s_memtime();      // diagnostics: the sorting wave is done
    }
    // ---- vertices (large_scale_feat_to_v.cpp:100-125) by the sorting wave, which holds every word's count and cls-attention
    // sum in registers (the block-wide form further down - two reductions over sixteen waves and the padding loops behind
    // four barriers - took 8 k cycles of every image).  In the edges kernel it runs BEHIND the first barrier: the edge rows
    // are dealt dynamically, so the wave joins them a little later and nobody waits for it.
    auto write_vertices = [&]() {
            const int n_groups = sg.n_groups;
            if (a.out_n && lane == 0) a.out_n[b] = n_groups;
            if (a.out_n_max && lane == 0) atomicMax(a.out_n_max, n_groups);
            if (do_v) {
                float a0[4], a1[4], m0 = -INFINITY, m1 = -INFINITY;
                bool nan0 = false, nan1 = false;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    a0[e] = (float)sg.cnt[e];
                    a1[e] = a.mean ? sg.sum[e] / (float)sg.cnt[e] : sg.sum[e];
                    if (sg.head[e]) {                             // max over the words, NaN propagating like at::max (:124)
                        nan0 = nan0 || (a0[e] != a0[e]); nan1 = nan1 || (a1[e] != a1[e]);
                        m0 = (a0[e] != a0[e]) ? m0 : fmaxf(m0, a0[e]);
                        m1 = (a1[e] != a1[e]) ? m1 : fmaxf(m1, a1[e]);
                    }
                }
                m0 = __any(nan0) ? NAN : sn_wave_max(m0);
                m1 = __any(nan1) ? NAN : sn_wave_max(m1);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!sg.head[e] || sg.g[e] >= a.n_pad) continue;
                    const float v0 = sn_nan_to_num(a0[e] / m0), v1 = sn_nan_to_num(a1[e] / m1);
                    const int64_t o = (int64_t)b * a.n_pad + sg.g[e];
                    if (a.out_v2) { a.out_v2[2 * o] = v0; a.out_v2[2 * o + 1] = v1; }
                    if (a.out_v) {
                        const float t0 = v0 * wv0, t1 = v1 * wv1;
                        a.out_v[o] = t0 + t1;
                    }
                }
                for (int c = n_groups + lane; c < a.n_pad; c += SN_WAVE) {
                    const int64_t o = (int64_t)b * a.n_pad + c;
                    if (a.out_v2) { a.out_v2[2 * o] = 0.0f; a.out_v2[2 * o + 1] = 0.0f; }
                    if (a.out_v) a.out_v[o] = 0.0f;
                }
            }
            if (a.out_ids) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (sg.head[e] && sg.g[e] < a.n_pad) a.out_ids[(int64_t)b * a.n_pad + sg.g[e]] = (int64_t)(sg.key[e] >> 8);
                for (int c = n_groups + lane; c < a.n_pad; c += SN_WAVE) a.out_ids[(int64_t)b * a.n_pad + c] = a.pad_id;
            }
    };
    if (sorter && done && !kEdges) write_vertices();
    if (stamps && threadIdx.x == 0) stamps[(size_t)blockIdx.x * 16 + 11] = __builtin_amdgcn_s_memtime();   // ... wave 0 has its rows in
    if (RR == 0 && stamps && threadIdx.x == 3 * 64) stamps[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memtime(); // ... wave 3 (on the sorting wave's SIMD)
    if (RR == 0 && stamps && threadIdx.x == 5 * 64) stamps[(size_t)blockIdx.x * 16 + 15] = __builtin_amdgcn_s_memtime(); // ... wave 5
    // (LDS contents only: __syncthreads() would also wait for the sorting wave's global stores and its atomic on the batch
    // maximum - one address for every image of the launch - to complete, 7 k cycles with everybody standing at the barrier;
    // nobody reads those in this kernel)
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    SN_GSTAMP(1);
    const bool lean = lean_ok && s.misc[2] != 0;                 // block-uniform: the sorting wave has prepared the edges phase
    // ---- group positions by word (the ballot form only when the sort does not apply: words >= 2^24)
    PosInfo me = {0, 0, 0, 0.0f};
    if (lean_ok && !lean && !a.geo) build_grid_T(s, L, a.feat_w, a.dist_alpha, a.dist_pow, tid);    // (complete at the barrier in front of the signed table's build)
    if (lean) {
        // (nothing to read: the vertices are the sorting wave's, the row map is the identity)
    } else if (s.misc[2] == 0) {
        me = group_positions(s, L, tid, tid < L, do_v);
    } else if (tid < L) {
        me.first = (s.flag[tid] & 2) != 0;
        me.cnt = s.pcnt[tid];
        me.group = me.first ? s.pgroup[tid] : 0;
        me.attn_sum = me.first ? s.psum[tid] : 0.0f;
    }
    SN_GSTAMP(2);
    const int n_groups = s.misc[0];
    const bool owner = tid < L && me.first;

    // ---- vertices  (large_scale_feat_to_v.cpp:100-125); written by the sorting wave when it grouped the positions
    const bool v_done = s.misc[2] != 0;                          // block-uniform
    if (!v_done && a.out_n && tid == 0) a.out_n[b] = n_groups;
    if (!v_done && a.out_n_max && tid == 0) atomicMax(a.out_n_max, n_groups);
    if (do_v && !v_done) {
        const float a0 = (float)me.cnt;
        const float a1 = a.mean ? me.attn_sum / (float)me.cnt : me.attn_sum;
        const float m0 = block_max_nan(s, a0, owner, tid, wid, nw, lane);
        const float m1 = block_max_nan(s, a1, owner, tid, wid, nw, lane);
        if (owner && me.group < a.n_pad) {
            const float v0 = sn_nan_to_num(a0 / m0), v1 = sn_nan_to_num(a1 / m1);
            const int64_t o = (int64_t)b * a.n_pad + me.group;
            if (a.out_v2) { a.out_v2[2 * o] = v0; a.out_v2[2 * o + 1] = v1; }
            if (a.out_v) {
                const float t0 = v0 * a.w_v[0], t1 = v1 * a.w_v[1];
                a.out_v[o] = t0 + t1;
            }
        }
        for (int c = n_groups + tid; c < a.n_pad; c += blockDim.x) {
            const int64_t o = (int64_t)b * a.n_pad + c;
            if (a.out_v2) { a.out_v2[2 * o] = 0.0f; a.out_v2[2 * o + 1] = 0.0f; }
            if (a.out_v) a.out_v[o] = 0.0f;
        }
    }
    if (a.out_ids && !v_done) {
        if (owner && me.group < a.n_pad) a.out_ids[(int64_t)b * a.n_pad + me.group] = s.words[tid];
        for (int c = n_groups + tid; c < a.n_pad; c += blockDim.x) a.out_ids[(int64_t)b * a.n_pad + c] = a.pad_id;
    }
    if (!kEdges) return;
    SN_GSTAMP(3);

    // ---- output row of every group: canonical rank, or the caller's dictionary
    // (large_scale_feat_to_e.cpp:117-118; missing key -> 0; on collisions the last (gi, gj) in
    // iteration order wins, i.e. the largest group index)
    int n_out = n_groups;
    if (lean) {
        // (row map, signed table, inverse of pos_sorted, row counter: written in front of the barrier)
    } else if (c_dict) {
        n_out = (int)a.dict_len[b];
        if (owner) {
            const int64_t *keys = a.dict_keys + a.dict_off[b], *vals = a.dict_vals + a.dict_off[b];
            const int64_t w = s.words[tid];
            int lo = 0, hi = n_out - 1;
            int64_t row = 0;
            while (lo <= hi) {
                const int mid = (lo + hi) >> 1;
                const int64_t k = keys[mid];
                if (k == w) { row = vals[mid]; break; }
                if (k < w) lo = mid + 1; else hi = mid - 1;
            }
            if (row >= 0 && row < n_out && row < kMaxCols) atomicMax(&s.rev[(int)row], me.group);
        }
    } else if (owner && me.group < kMaxCols) {
        s.rev[me.group] = me.group;
    }
    // Signed grid table over the (now dead) grouping scratch: TS[(dr + feat_h - 1) * S + (dc + feat_w - 1)] =
    // T[|dr|][|dc|], S = 2 feat_w - 1, so that the table offset of a (row position p, column position q) pair is
    // P(p) + Q(q) - one scalar and one per-lane constant - instead of two absolute differences, a shift and an add
    // per cell and row visit (the column-sum loop below is instruction-bound).  Same floats, same sums.
    float *TS = nullptr;
    if (lean) {
        TS = reinterpret_cast<float *>(s.pless);
    } else if (kFast || signed_table) {                          // kernel argument: uniform
        __syncthreads();                                         // every thread has read its grouping results
        TS = reinterpret_cast<float *>(s.pless);
        const int sh = grid_shift(a.feat_w), fh = L / a.feat_w, n_ts = (2 * fh - 1) * ts_S;
        for (int i = tid; i < n_ts; i += blockDim.x) {
            const int dr = i / ts_S - (fh - 1), dc = i % ts_S - (a.feat_w - 1);
            TS[i] = s.T[((dr < 0 ? -dr : dr) << sh) + (dc < 0 ? -dc : dc)];
        }
    }
    // s.flag is free after the grouping: it now holds the inverse of pos_sorted (position -> sorted index)
    const int n_kept = s.misc[1];
    if (!lean) {
        if (tid < n_kept) s.flag[s.pos_sorted[tid]] = (unsigned char)tid;
        if (tid == 0) s.misc[5] = nw;                            // next edge row to deal (see the row loop)
        __syncthreads();
    }

    SN_GSTAMP(4);
    // ---- edges: one wave per output row, lanes over output columns
    // (large_scale_feat_to_e.cpp:90-140)
    // (read into scalar registers HERE: left as pending vector loads, hipcc waits with vmcnt(0) at every use in
    // the row loop, i.e. for the previous row's stores, before each store)
    const float w0 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.w_e[0])));
    const float w1 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(a.w_e[1])));
    int gcol[kCellsPerLane];
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) {
        const int c = lane + SN_WAVE * k;
        gcol[k] = c < n_out ? s.rev[c] : -1;
    }
    ColCache cc;
    cache_columns(s, gcol, cc);
    // Cells of output row r (group gi), two regular passes instead of a (p in gi) x (q in gj) loop per cell
    // whose trip count is set by the longest word of the wave:
    //   a) column sums over the group's rows: cs[q] = sum_{p in gi} A[p][q] (and of the grid similarity) for
    //      ALL L positions q, lanes over q in position order - contiguous LDS reads, no divergence,
    //      cnt(gi) iterations; over an image that is L row visits, whatever the word histogram;
    //   b) cs is staged in LDS - in row p0 of A, the group's own first row, which nobody reads any more:
    //      row p of A is only ever read by the wave that handles group(p) - and lane c gathers the
    //      positions of ITS column's word: cell = sum_{q in gj} cs[q].
    // Sum order: q-major instead of the reference's p-major (large_scale_feat_to_e.cpp:99-125): equal to
    // fp32 rounding (1e-7); the init statistics keep the reference order (limited_edges_kernel).
    int qv4[kCellsPerLane], qrow[kCellsPerLane], qcol4[kCellsPerLane], qsidx4[kCellsPerLane], qts[kCellsPerLane];
    bool qok[kCellsPerLane];
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) {
        const int q = lane + SN_WAVE * k;
        qok[k] = q < L;
        const int qq = qok[k] ? q : L - 1;                  // (lanes past L accumulate a valid column nobody reads)
        qv4[k] = qq * 4;
        const int rc = c_geo ? 0 : (int)s.prc[qq];
        qrow[k] = rc >> 8;
        qcol4[k] = (rc & 255) * 4;
        qts[k] = TS ? ((L / a.feat_w - 1 - (rc >> 8)) * ts_S + (a.feat_w - 1 - (rc & 255))) * 4 : 0;
        qsidx4[k] = (int)s.flag[qq] * 4;                    // where this position's column sum is staged (sorted order)
    }
    const int tsh = grid_shift(a.feat_w) + 2;               // byte shift of a table row
    // segment heads of the sorted position order, for the lane's four sorted indices 4 lane + e (bit e):
    // a word's first position; indices past the kept positions are their own (empty) segments
    const bool scan_ok = kFast || ((L & 3) == 0 && L <= 4 * SN_WAVE);
    unsigned hd = 0;
    if (scan_ok) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int i = 4 * lane + e;
            bool head = true;
            if (i > 0 && i < n_kept) head = s.words[s.pos_sorted[i]] != s.words[s.pos_sorted[i - 1]];
            hd |= head ? (1u << e) : 0u;
        }
    }
    // Cross-lane half of the segmented scan on the DPP network (a ds_bpermute per step costs an exposed LDS round
    // trip each, 14 per row): Hillis-Steele inside the 16-lane rows (row_shr 1, 2, 4, 8), then the last lane of row
    // 0 / 2 into rows 1 / 3 (row_bcast:15) and lane 31 into rows 2, 3 (row_bcast:31).  A step adds the incoming
    // value iff the lane has not seen a head since the sender: that flag half depends on the heads only, so it is
    // run once here (bit i of `take`).
    unsigned take = 0;
    if (scan_ok) {
        int fl = hd != 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int fup = __shfl_up(fl, 1 << i, SN_WAVE);
            if ((lane & 15) >= (1 << i)) { take |= fl ? 0u : (1u << i); fl |= fup; }
        }
        {
            const int fup = __shfl(fl, (lane & 48) - 1, SN_WAVE);       // last lane of the previous row (rows 1 and 3 listen)
            if ((lane >> 4) & 1) { take |= fl ? 0u : (1u << 4); fl |= fup; }
        }
        {
            const int fup = __shfl(fl, 31, SN_WAVE);
            if (lane >= 32) { take |= fl ? 0u : (1u << 5); fl |= fup; }
        }
    }
    // the same flags as multipliers (see the scan below)
    const float nh1 = (hd & 2) ? 0.0f : 1.0f, nh2 = (hd & 4) ? 0.0f : 1.0f, nh3 = (hd & 8) ? 0.0f : 1.0f;
    const float cf0 = (hd & 1) ? 0.0f : 1.0f, cf1 = (hd & 3) ? 0.0f : 1.0f, cf2 = (hd & 7) ? 0.0f : 1.0f, cf3 = (hd & 15) ? 0.0f : 1.0f;
    float tk[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) tk[i] = ((take >> i) & 1u) ? 1.0f : 0.0f;
    // lean form: 1 / (positions of the word of the lane's k-th POSITION) (group_positions_sorted<true> leaves it in s.T): the weight
    // that turns column sums into the summands of the cell means before the scan (see the fused row below)
    float rq[kCellsPerLane];
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) rq[k] = (lean && qok[k]) ? s.T[lane + SN_WAVE * k] : 0.0f;
    float rcnt[kCellsPerLane];                             // 1 / (positions of the column's word)
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) rcnt[k] = 1.0f / (float)(cc.cnt[k] > 0 ? cc.cnt[k] : 1);
    unsigned long long dt_a = 0, dt_b = 0, dt_c = 0, dt_pre = 0;      // diagnostics (stamps on): wave 0's time in passes a / b, in normalise + store, before the row loop
    // KC = 64-column cells of an output row a lane holds: an image with at most 64 / 128 distinct words (the usual case:
    // ~113 of 196 positions) needs one / two, not four - every per-cell instruction of the scan gathers, the means, the
    // normalisation and the stores is issued KC times per row
    if (stamps && threadIdx.x == 0) dt_pre = __builtin_amdgcn_s_memtime() - stamps[(size_t)blockIdx.x * 16 + 4];
    // Rows are dealt dynamically: a wave takes its first row by its index and every further one from a counter in LDS
    // (s.misc[5], set to nw before the barrier above), asked for before the current row is worked on.  Rows cost what
    // their word's position count costs and the sorting wave starts late (it writes the vertices first): a static deal
    // ended with the slowest wave, 18 % behind the median.  Only rows that can hold a word are dealt (r < n_out); the
    // zero rows of the padding come afterwards.
    const int n_rows = n_out < a.n_pad ? n_out : a.n_pad;
    if (kEdges && sorter && done) write_vertices();              // (behind the last barrier before the row loop: see write_vertices)
    auto edge_rows = [&](auto kc_c) {
    constexpr int KC = decltype(kc_c)::value;
    for (int r = wid < n_rows ? wid : n_rows + wid; r < a.n_pad;) {      // (padding rows: n_rows + wid, + nw, ... for every wave)
        const int r_this = r;
        {
            int rn = 0;
            if (r_this < n_rows) {                               // wave-uniform
                if (lane == 0) rn = atomicAdd(&s.misc[5], 1);
                rn = __builtin_amdgcn_readfirstlane(rn);
                r = rn < n_rows ? rn : n_rows + wid;             // (out of real rows: on to this wave's share of the padding rows)
            } else {
                r = r_this + nw;
            }
        }
        const int gi = (r_this < n_out && r_this < kMaxCols) ? s.rev[r_this] : -1;
        float c0[kCellsPerLane], c1[kCellsPerLane];
        float t0 = 0.0f, t1 = 0.0f;
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0;
        if (gi >= 0) {
            const int ia = __builtin_amdgcn_readfirstlane((int)s.gstart[gi]);
            const int ib = __builtin_amdgcn_readfirstlane((int)s.gstart[gi + 1]);
            float csa[kCellsPerLane], csg[kCellsPerLane];
#pragma unroll
            for (int k = 0; k < kCellsPerLane; ++k) { csa[k] = 0.0f; csg[k] = 0.0f; }
            if (stamps) ts0 = __builtin_amdgcn_s_memtime();
            for (int x = ia; x < ib; ++x) {                 // a) column sums over the rows of group gi
                const int p = __builtin_amdgcn_readfirstlane((int)s.pos_sorted[x]);
                const char *arow = reinterpret_cast<const char *>(s.A + p * L);
                if (c_geo) {
                    const float *grow = c_geo + (int64_t)p * L;
#pragma unroll
                    for (int k = 0; k < kCellsPerLane; ++k) {
                        const float av = *reinterpret_cast<const float *>(arow + qv4[k]);
                        const float gv = grow[qv4[k] >> 2];
                        csa[k] += qok[k] ? av : 0.0f;
                        csg[k] += qok[k] ? gv : 0.0f;
                    }
                } else if (kFast || TS) {
                    const int prc = __builtin_amdgcn_readfirstlane((int)s.prc[p]);
                    const char *tp = reinterpret_cast<const char *>(TS) + ((prc >> 8) * ts_S + (prc & 255)) * 4;
#pragma unroll
                    for (int k = 0; k < kCellsPerLane; ++k) {
                        csa[k] += *reinterpret_cast<const float *>(arow + qv4[k]);
                        csg[k] += *reinterpret_cast<const float *>(tp + qts[k]);
                    }
                } else {
                    const int prc = __builtin_amdgcn_readfirstlane((int)s.prc[p]);
                    const unsigned pr = prc >> 8, pc4 = (prc & 255) * 4;
                    const char *tbase = reinterpret_cast<const char *>(s.T);
#pragma unroll
                    for (int k = 0; k < kCellsPerLane; ++k) {
                        const float av = *reinterpret_cast<const float *>(arow + qv4[k]);
                        const unsigned off = (__usad(pr, (unsigned)qrow[k], 0u) << tsh) + __usad(pc4, (unsigned)qcol4[k], 0u);
                        const float gv = *reinterpret_cast<const float *>(tbase + off);
                        csa[k] += av;
                        csg[k] += gv;
                    }
                }
            }
            if (stamps) ts1 = __builtin_amdgcn_s_memtime();
            // b) stage the column sums, in SORTED position order, in the group's first row; the positions of
            // an output column's word are then a contiguous run: independent, pipelined LDS reads
            char *stage = reinterpret_cast<char *>(s.A + __builtin_amdgcn_readfirstlane((int)s.pos_sorted[ia]) * L);
            if constexpr (kFast) {
                // Fused row (round 5).  Means, row normalisation and the weighting by w_e are linear in the column sums, so they are
                // applied BEFORE the scan: u = column sum / positions of its word; the row totals T0, T1 of the cell means are then
                // plain wave sums of u (every position belongs to one word; the row's own 1 / n_i cancels), and the output row is ONE
                // segmented scan of z = (w_e0 / T0) u0 + (w_e1 / T1) u1 - instead of two scans and a pass of multiplies over the
                // cells behind them.  Same cells to fp32 rounding (another association: 1e-7).  Rows whose totals are not positive and
                // finite (an all-clamped attention row: NaN) keep the two-scan form with nan_to_num below.
                if (lean) {
                    float u0[kCellsPerLane], u1[kCellsPerLane];
#pragma unroll
                    for (int k = 0; k < kCellsPerLane; ++k) { u0[k] = csg[k] * rq[k]; u1[k] = csa[k] * rq[k]; }
                    const float T0 = wave_sum_fast((u0[0] + u0[1]) + (u0[2] + u0[3]));
                    const float T1 = wave_sum_fast((u1[0] + u1[1]) + (u1[2] + u1[3]));
                    if (T0 > 0.0f && T0 < INFINITY && T1 > 0.0f && T1 < INFINITY) {             // wave-uniform
                        const float a0 = w0 * __builtin_amdgcn_rcpf(T0), a1 = w1 * __builtin_amdgcn_rcpf(T1);
                        __builtin_amdgcn_wave_barrier();
#pragma unroll
                        for (int k = 0; k < kCellsPerLane; ++k)
                            if (qok[k]) *reinterpret_cast<float *>(stage + qsidx4[k]) = fmaf(a0, u0[k], a1 * u1[k]);
                        __builtin_amdgcn_wave_barrier();
                        f32x4 v4 = {0.0f, 0.0f, 0.0f, 0.0f};
                        if (4 * lane < L) v4 = *reinterpret_cast<const f32x4 *>(stage + 16 * lane);
                        const float s0 = v4.x;
                        const float s1 = fmaf(s0, nh1, v4.y);
                        const float s2 = fmaf(s1, nh2, v4.z);
                        const float s3 = fmaf(s2, nh3, v4.w);
                        float run = s3;
#define SN_SCAN_STEP(i, ctrl, rows)                                                                                            \
                        asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " ctrl " row_mask:" rows " bank_mask:0xf bound_ctrl:0" : "+v"(run) : "v"(tk[i]));
                        SN_SCAN_STEP(0, "row_shr:1", "0xf")
                        SN_SCAN_STEP(1, "row_shr:2", "0xf")
                        SN_SCAN_STEP(2, "row_shr:4", "0xf")
                        SN_SCAN_STEP(3, "row_shr:8", "0xf")
                        SN_SCAN_STEP(4, "row_bcast:15", "0xa")
                        SN_SCAN_STEP(5, "row_bcast:31", "0xc")
#undef SN_SCAN_STEP
                        float carry;
                        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(carry) : "v"(run));
                        v4.x = fmaf(carry, cf0, s0);
                        v4.y = fmaf(carry, cf1, s1);
                        v4.z = fmaf(carry, cf2, s2);
                        v4.w = fmaf(carry, cf3, s3);
                        __builtin_amdgcn_wave_barrier();
                        if (4 * lane < L) *reinterpret_cast<f32x4 *>(stage + 16 * lane) = v4;
                        __builtin_amdgcn_wave_barrier();
                        const int64_t rowbase_f = ((int64_t)b * a.n_pad + r_this) * a.n_pad;
#pragma unroll
                        for (int k = 0; k < KC; ++k) {
                            const int c = lane + SN_WAVE * k;
                            const int last = cc.cnt[k] > 0 ? cc.ja[k] + cc.cnt[k] - 1 : 0;
                            const float v = *reinterpret_cast<const float *>(stage + last * 4);
                            if (c < n_out && c < a.n_pad) a.out_e[rowbase_f + c] = v;
                        }
                        if (stamps) { ts2 = __builtin_amdgcn_s_memtime(); dt_a += ts1 - ts0; dt_b += ts2 - ts1; }
                        continue;
                    }
                }
            }
            float sa[kCellsPerLane], sg[kCellsPerLane];
            if (scan_ok) {
                // cell(gj) = sum of a contiguous run of the staged vector -> ONE segmented inclusive scan of the
                // wave (segments = words, heads precomputed) gives every cell at the last element of its run:
                // lane l owns the sorted indices 4l..4l+3, 6 shuffle steps across lanes.  Only additions inside
                // a segment (no prefix differences): fp32 like the sequential sum, tree order.
#pragma unroll
                for (int pass = 0; pass < 2; ++pass) {
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < kCellsPerLane; ++k)
                        if (qok[k]) *reinterpret_cast<float *>(stage + qsidx4[k]) = pass ? csg[k] : csa[k];
                    __builtin_amdgcn_wave_barrier();
                    f32x4 v4 = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (4 * lane < L) v4 = *reinterpret_cast<const f32x4 *>(stage + 16 * lane);
                    // (run += incoming * tk: tk = 1.0 where the step adds, 0.0 where a head lies in between - one v_fmac with
                    // a DPP source per step instead of a move, an add and a select.  x * 1.0 is exact; the values of one row's
                    // scan are either all finite or all NaN (a fully clamped attention row), so 0 * NaN changes nothing.)
                    float s0 = v4.x;
                    float s1 = fmaf(s0, nh1, v4.y);
                    float s2 = fmaf(s1, nh2, v4.z);
                    float s3 = fmaf(s2, nh3, v4.w);
                    float run = s3;                                 // sum since the last head of this lane (or of all four)
#define SN_SCAN_STEP(i, ctrl, rows)                                                                                            \
                    asm volatile("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 " ctrl " row_mask:" rows " bank_mask:0xf bound_ctrl:0" : "+v"(run) : "v"(tk[i]));
                    SN_SCAN_STEP(0, "row_shr:1", "0xf")
                    SN_SCAN_STEP(1, "row_shr:2", "0xf")
                    SN_SCAN_STEP(2, "row_shr:4", "0xf")
                    SN_SCAN_STEP(3, "row_shr:8", "0xf")
                    SN_SCAN_STEP(4, "row_bcast:15", "0xa")            // -> rows 1, 3
                    SN_SCAN_STEP(5, "row_bcast:31", "0xc")            // -> rows 2, 3
#undef SN_SCAN_STEP
                    // running sum that reaches into this lane = the previous lane's total (wave_shr:1; lane 0: 0)
                    float carry;
                    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(carry) : "v"(run));
                    v4.x = fmaf(carry, cf0, s0);
                    v4.y = fmaf(carry, cf1, s1);
                    v4.z = fmaf(carry, cf2, s2);
                    v4.w = fmaf(carry, cf3, s3);
                    __builtin_amdgcn_wave_barrier();
                    if (4 * lane < L) *reinterpret_cast<f32x4 *>(stage + 16 * lane) = v4;
                    __builtin_amdgcn_wave_barrier();
#pragma unroll
                    for (int k = 0; k < KC; ++k) {
                        const int last = cc.cnt[k] > 0 ? cc.ja[k] + cc.cnt[k] - 1 : 0;
                        const float v = *reinterpret_cast<const float *>(stage + last * 4);
                        if (pass) sg[k] = v; else sa[k] = v;
                    }
                }
            } else {
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < kCellsPerLane; ++k)
                    if (qok[k]) *reinterpret_cast<float *>(stage + qsidx4[k]) = pass ? csg[k] : csa[k];
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    float acc = 0.0f;
                    const char *run = stage + cc.ja[k] * 4;
                    const int last = cc.cnt[k] > 0 ? cc.cnt[k] - 1 : 0;
                    for (int t = 0; t < cc.tmax[k]; ++t) {  // wave-uniform trip count; lanes past their run re-read its end
                        const float v = *reinterpret_cast<const float *>(run + (t < last ? t : last) * 4);
                        acc += t < cc.cnt[k] ? v : 0.0f;
                    }
                    if (pass) sg[k] = acc; else sa[k] = acc;
                }
            }
            }
            if (stamps) { ts2 = __builtin_amdgcn_s_memtime(); dt_a += ts1 - ts0; dt_b += ts2 - ts1; }
#pragma unroll
            for (int k = 0; k < KC; ++k) {
                const bool has = cc.cnt[k] > 0;
                c0[k] = has ? sg[k] : 0.0f;
                c1[k] = has ? sa[k] : 0.0f;
            }
            if (c_mean) {                                    // wave-uniform; 1 / (rows x columns) as two reciprocals (2 ulp)
                const float rr = __builtin_amdgcn_rcpf((float)(ib - ia));
#pragma unroll
                for (int k = 0; k < KC; ++k) {
                    const float inv = rr * rcnt[k];
                    c0[k] = c0[k] * inv;
                    c1[k] = c1[k] * inv;
                }
            }
        } else {                                             // padding row: zeros, nothing to normalise
            if (c_skip) continue;                            // (the consumer masks by the vertex count)
            const int64_t rowbase0 = ((int64_t)b * a.n_pad + r_this) * a.n_pad;
            for (int c = lane; c < a.n_pad; c += SN_WAVE) {
                if (c_out_e2) { c_out_e2[2 * (rowbase0 + c)] = 0.0f; c_out_e2[2 * (rowbase0 + c) + 1] = 0.0f; }
                if (a.out_e) a.out_e[rowbase0 + c] = 0.0f;
            }
            continue;
        }
#pragma unroll
        for (int k = 0; k < KC; ++k) { t0 += c0[k]; t1 += c1[k]; }
        t0 = wave_sum_fast(t0);   // instance_edges.sum(1, keepdim)  :135
        t1 = wave_sum_fast(t1);
        const float i0 = __builtin_amdgcn_rcpf(t0), i1 = __builtin_amdgcn_rcpf(t1);     // one v_rcp_f32 per row (x * rcp(t) vs x / t: 2 ulp; a correctly rounded 1 / t is ten instructions)
        // nan_to_num only matters when a row sum is 0 / inf / NaN (then some quotient is not finite)
        const bool plain = t0 > 0.0f && t0 < INFINITY && t1 > 0.0f && t1 < INFINITY;      // wave-uniform
        const int64_t rowbase = ((int64_t)b * a.n_pad + r_this) * a.n_pad;
#pragma unroll
        for (int k = 0; k < KC; ++k) {
            const int c = lane + SN_WAVE * k;
            if (c >= a.n_pad || (c_skip && c >= n_out)) continue;
            float e0 = 0.0f, e1 = 0.0f;
            if (c < n_out) {
                e0 = c0[k] * i0;
                e1 = c1[k] * i1;
                if (!plain) { e0 = sn_nan_to_num(e0); e1 = sn_nan_to_num(e1); }
                if (c_rsl && c == r_this) { e0 = 0.0f; e1 = 0.0f; }
            }
            if (c_out_e2) { c_out_e2[2 * (rowbase + c)] = e0; c_out_e2[2 * (rowbase + c) + 1] = e1; }
            if (a.out_e) {
                const float p0 = e0 * w0, p1 = e1 * w1;
                a.out_e[rowbase + c] = p0 + p1;
            }
        }
        for (int c = SN_WAVE * KC + lane; c < a.n_pad && !c_skip; c += SN_WAVE) {          // (columns past the cells this variant holds: padding)
            if (c_out_e2) { c_out_e2[2 * (rowbase + c)] = 0.0f; c_out_e2[2 * (rowbase + c) + 1] = 0.0f; }
            if (a.out_e) a.out_e[rowbase + c] = 0.0f;
        }
        if (stamps) dt_c += __builtin_amdgcn_s_memtime() - ts2;
    }
    };
    {
        const int n_cols = a.n_pad < n_out ? a.n_pad : n_out;
        if (n_cols <= SN_WAVE) edge_rows(std::integral_constant<int, 1>{});
        else if (n_cols <= 2 * SN_WAVE) edge_rows(std::integral_constant<int, 2>{});
        else edge_rows(std::integral_constant<int, kCellsPerLane>{});
    }
    __syncthreads();
    SN_GSTAMP(5);
    if (stamps && threadIdx.x == 0) { stamps[(size_t)blockIdx.x * 16 + 6] = dt_a; stamps[(size_t)blockIdx.x * 16 + 7] = dt_b; stamps[(size_t)blockIdx.x * 16 + 8] = dt_c; stamps[(size_t)blockIdx.x * 16 + 9] = dt_pre; }
}

// ------------------------------------------------------------------------------------------
// init: dense vertex attributes  (feat_to_v_attr.cpp:74-148 + schema_net.py:200-207)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void full_vertices_kernel(
    const int64_t *ing, int64_t sb, int64_t sl, const float *attn_cls, int L, int M, int is_logits,
    int use_clamp, float clamp, int mean, int ingredients_only, const float *w_v, float *out_attr2,
    float *out_v)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    const Lds s = carve(smem, L, false);
    if (tid < L) s.words[tid] = ing[(int64_t)b * sb + (int64_t)tid * sl];
    const bool want_attn = !ingredients_only && attn_cls != nullptr;
    if (want_attn && wid == nw - 1) {
        float x[4];
        load_row4<false>(attn_cls + (int64_t)b * L, L, lane, x);
        if (is_logits) softmax_row4<false>(x, L, lane, use_clamp != 0, clamp);   // no nan_to_num (:202)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = lane + SN_WAVE * k;
            if (c < L) s.acls[c] = x[k];
        }
    }
    // zero-fill this image's dense outputs while the grouping runs
    if (out_attr2) for (int i = tid; i < 2 * M; i += blockDim.x) out_attr2[(int64_t)b * 2 * M + i] = 0.0f;
    if (out_v) for (int i = tid; i < M; i += blockDim.x) out_v[(int64_t)b * M + i] = 0.0f;
    __syncthreads();
    const PosInfo me = group_positions(s, L, tid, tid < L, want_attn);
    const bool owner = tid < L && me.first;
    const float a0 = (float)me.cnt;
    const float a1 = want_attn ? (mean ? me.attn_sum / (float)me.cnt : me.attn_sum) : 0.0f;
    // normalize_max_(dim=1) runs over all M vertices; absent words hold 0 (graph/utils.py:16-22)
    const bool has_absent = s.misc[0] < M;
    float m0 = block_max_nan(s, a0, owner, tid, wid, nw, lane);
    float m1 = block_max_nan(s, a1, owner, tid, wid, nw, lane);
    if (has_absent) { m0 = (m0 != m0) ? m0 : fmaxf(m0, 0.0f); m1 = (m1 != m1) ? m1 : fmaxf(m1, 0.0f); }
    if (owner) {
        const int64_t w = s.words[tid];
        if (w >= 0 && w < M) {
            if (out_attr2) { out_attr2[((int64_t)b * M + w) * 2] = a0; out_attr2[((int64_t)b * M + w) * 2 + 1] = a1; }
            if (out_v) {
                const float t0 = sn_nan_to_num(a0 / m0) * w_v[0], t1 = sn_nan_to_num(a1 / m1) * w_v[1];
                out_v[(int64_t)b * M + w] = t0 + t1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// init: class-restricted dense edges  (feat_to_e.cpp:31-127 + schema_net.py:237-254)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void limited_edges_kernel(
    const int64_t *ing, int64_t sb, int64_t sl, const float *attn, int L, int is_logits, int use_clamp,
    float clamp, const float *geo, int feat_w, float alpha, float pw, const int32_t *class_slot,
    int Mtab, const int64_t *label, int n_max, int mean, int remove_self_loop, const float *w_e,
    float *out_attr2, float *out_e)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wid = tid >> 6, nw = blockDim.x >> 6;
    const Lds s = carve(smem, L, true);
    attn_rows_to_lds(s.A, attn + (int64_t)b * L * L, L, 1, 0, L, is_logits != 0, use_clamp != 0, clamp, wid, L, nw, lane);
    if (!geo) build_grid_table(s, L, feat_w, alpha, pw, tid);
    const int32_t *slot_tab = class_slot + (int64_t)label[b] * Mtab;
    int my_slot = -1;
    if (tid < L) {
        const int64_t w = ing[(int64_t)b * sb + (int64_t)tid * sl];
        s.words[tid] = w;
        if (w >= 0 && w < Mtab) my_slot = slot_tab[w];
        if (my_slot >= n_max) my_slot = -1;
    }
    // zero-fill the dense [n_max, n_max] outputs (at::zeros, feat_to_e.cpp:46)
    const int64_t cells = (int64_t)n_max * n_max;
    if (out_attr2) for (int64_t i = tid; i < 2 * cells; i += blockDim.x) out_attr2[(int64_t)b * 2 * cells + i] = 0.0f;
    if (out_e) for (int64_t i = tid; i < cells; i += blockDim.x) out_e[(int64_t)b * cells + i] = 0.0f;
    __syncthreads();
    const PosInfo me = group_positions(s, L, tid, my_slot >= 0, false);
    const int n_groups = s.misc[0];
    // group -> class slot, kept in rev[] (n_groups <= L <= kMaxCols)
    if (tid < L && me.first && my_slot >= 0) s.rev[me.group] = my_slot;
    __syncthreads();
    const float w0 = w_e ? w_e[0] : 0.0f, w1 = w_e ? w_e[1] : 0.0f;
    int gcol[kCellsPerLane];
#pragma unroll
    for (int k = 0; k < kCellsPerLane; ++k) gcol[k] = (lane + SN_WAVE * k) < n_groups ? lane + SN_WAVE * k : -1;
    ColCache cc;
    cache_columns(s, gcol, cc);
    const bool any_long = __any(cc.cnt[0] > kQCache || cc.cnt[1] > kQCache || cc.cnt[2] > kQCache || cc.cnt[3] > kQCache) != 0;
    for (int gi = wid; gi < n_groups; gi += nw) {
        const int si = s.rev[gi];
        float c0[kCellsPerLane], c1[kCellsPerLane];
        float t0 = 0.0f, t1 = 0.0f;
        row_cells(s, geo, L, feat_w, gi, cc, any_long, mean, c0, c1);
#pragma unroll
        for (int k = 0; k < kCellsPerLane; ++k) { t0 += c0[k]; t1 += c1[k]; }
        t0 = sn_wave_sum(t0);     // normalize_sum_(edges_attr, dim=2)  schema_net.py:249
        t1 = sn_wave_sum(t1);
#pragma unroll
        for (int k = 0; k < kCellsPerLane; ++k) {
            const int gj = lane + SN_WAVE * k;
            if (gj >= n_groups) continue;
            const int sj = s.rev[gj];
            const int64_t o = (int64_t)b * cells + (int64_t)si * n_max + sj;
            if (out_attr2) { out_attr2[2 * o] = c0[k]; out_attr2[2 * o + 1] = c1[k]; }
            if (out_e) {
                float e0 = sn_nan_to_num(c0[k] / t0), e1 = sn_nan_to_num(c1[k] / t1);
                if (remove_self_loop && si == sj) { e0 = 0.0f; e1 = 0.0f; }
                const float p0 = e0 * w0, p1 = e1 * w1;
                out_e[o] = p0 + p1;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// per-class sums in image order (scripts/init_schema_net.py:33-35, 59-61)
// grid: (ceil(F / 1024), K); each thread owns up to 4 feature columns of one class.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void stats_accumulate_kernel(const float *feat, const int64_t *label,
                                                               int B, int64_t F, float *class_sum,
                                                               float *n_tracked)
{
    const int k = blockIdx.y;
    const int64_t c0 = (int64_t)blockIdx.x * 1024 + threadIdx.x;
    float acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t c = c0 + 256 * j;
        acc[j] = c < F ? class_sum[(int64_t)k * F + c] : 0.0f;
    }
    int n = 0;
    for (int b = 0; b < B; ++b) {
        if (label[b] != k) continue;     // wave-uniform
        ++n;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t c = c0 + 256 * j;
            if (c < F) acc[j] = acc[j] + feat[(int64_t)b * F + c];
        }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int64_t c = c0 + 256 * j;
        if (c < F) class_sum[(int64_t)k * F + c] = acc[j];
    }
    if (n_tracked && blockIdx.x == 0 && threadIdx.x == 0) n_tracked[k] = n_tracked[k] + (float)n;
}

// ------------------------------------------------------------------------------------------
// wrapper taps: head mean + slicing  (ingredient_model_wrapper.py:58-68)
// grid (L+1 rows, B); row 0 -> attn_cls, row p+1 -> attn[:, p, :]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void head_mean_kernel(const float *ext, int H, int L, float *attn, float *attn_cls)
{
    const int row = blockIdx.x, b = blockIdx.y, q = threadIdx.x;
    if (q >= L) return;
    const int64_t S = (int64_t)(L + 1) * (L + 1);
    const float *src = ext + (int64_t)b * H * S + (int64_t)row * (L + 1) + 1 + q;
    float acc = src[0];
    for (int h = 1; h < H; ++h) acc = acc + src[h * S];
    acc = acc / (float)H;
    if (row == 0) attn_cls[(int64_t)b * L + q] = acc;
    else attn[((int64_t)b * L + (row - 1)) * L + q] = acc;
}

int ensure_lds(const void *fn, size_t bytes, const char *name)
{
    if (bytes > 160 * 1024) {
        sn_set_error("%s: needs %zu bytes of LDS (> 160 KiB)", name, bytes);
        return SN_ERR_UNSUPPORTED;
    }
    return sn_ensure_dynamic_lds(fn, bytes > 64 * 1024 ? (size_t)160 * 1024 : bytes, name);      // (raised once per kernel and device: to the maximum)
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" int sn_instance_graph(const sn_graph_args *args, void *stream)
{
    SN_REQUIRE(args, SN_ERR_BAD_ARG, "sn_instance_graph: args is NULL");
    SN_REQUIRE(args->struct_size == sizeof(sn_graph_args), SN_ERR_BAD_ARG, "sn_instance_graph: sn_graph_args.struct_size=%u, this library (ABI %d) expects %zu",
               args->struct_size, sn_abi_version(), sizeof(sn_graph_args));
    sn_graph_args a = *args;
    SN_REQUIRE(a.B >= 0 && a.L > 0, SN_ERR_BAD_ARG, "sn_instance_graph: bad B=%d L=%d", a.B, a.L);
    if (a.B == 0) return SN_OK;
    SN_REQUIRE(a.ingredients, SN_ERR_BAD_ARG, "sn_instance_graph: ingredients is NULL");
    SN_REQUIRE(a.attn || a.attn_cls, SN_ERR_BAD_ARG, "sn_instance_graph: nothing to do (attn and attn_cls NULL)");
    SN_REQUIRE(a.L <= SN_MAX_TOKENS, SN_ERR_UNSUPPORTED, "sn_instance_graph: L=%d > %d tokens", a.L, SN_MAX_TOKENS);
    SN_REQUIRE(a.n_pad > 0, SN_ERR_BAD_ARG, "sn_instance_graph: n_pad=%d", a.n_pad);
    SN_REQUIRE(a.B <= 0x7fffffff / 2, SN_ERR_BAD_ARG, "sn_instance_graph: B too large");
    if (a.attn_cls) SN_REQUIRE(a.w_v || !a.out_v, SN_ERR_BAD_ARG, "sn_instance_graph: out_v needs w_v");
    if (a.attn) {
        SN_REQUIRE(a.w_e, SN_ERR_BAD_ARG, "sn_instance_graph: w_e is NULL");
        SN_REQUIRE(a.geo || (a.feat_h > 0 && a.feat_w > 0 && a.feat_h * a.feat_w == a.L && a.feat_h < 256 && a.feat_w < 256),
                   SN_ERR_BAD_ARG, "sn_instance_graph: feat_h*feat_w (%d*%d) != L (%d)", a.feat_h, a.feat_w, a.L);
        SN_REQUIRE(a.geo || a.dist_alpha > 0.0f, SN_ERR_BAD_ARG, "sn_instance_graph: dist_alpha must be > 0");
    }
    if (a.dict_keys) SN_REQUIRE(a.dict_vals && a.dict_off && a.dict_len, SN_ERR_BAD_ARG, "sn_instance_graph: incomplete dictionary");
    // defaults: contiguous single-head maps
    if (a.attn_heads < 1) a.attn_heads = 1;
    if (a.acls_heads < 1) a.acls_heads = 1;
    if (a.attn_stride_r == 0) a.attn_stride_r = a.L;
    if (a.attn_stride_b == 0) a.attn_stride_b = (int64_t)a.L * a.L * a.attn_heads;
    if (a.attn_stride_h == 0) a.attn_stride_h = (int64_t)a.L * a.L;
    if (a.acls_stride_b == 0) a.acls_stride_b = (int64_t)a.L * a.acls_heads;
    if (a.acls_stride_h == 0) a.acls_stride_h = a.L;
    hipStream_t st = (hipStream_t)stream;
    if (a.attn) {
        size_t lds = lds_bytes(a.L, true);
        // signed grid table (see the kernel): it lies over the grouping scratch at the end of the carve and may need a
        // few hundred bytes more; used when that still fits the CU's LDS
        int signed_table = 0;
        if (!a.geo) {
            const size_t dead = up16((size_t)a.L * 2) * 2 + up16((size_t)a.L * 4) + up16(a.L);
            const size_t need = up16((size_t)(2 * a.feat_h - 1) * (2 * a.feat_w - 1) * 4);
            const size_t extra = need > dead ? need - dead : 0;
            if (lds + extra <= 160 * 1024) { lds += extra; signed_table = 1; }
        }
        const bool fast = signed_table && !a.geo && !a.out_e2 && a.out_e && a.mean && !a.remove_self_loop && !a.dict_keys &&
                          a.skip_edge_padding && (a.L & 3) == 0 && a.L <= 4 * SN_WAVE;
        sn_s1::RerankView rv = {};
        int rr = 0;
        if (a.rerank) {
            SN_REQUIRE(a.rerank->struct_size == sizeof(sn_rerank_args), SN_ERR_BAD_ARG, "sn_instance_graph: sn_rerank_args.struct_size=%u, this library (ABI %d) expects %zu",
                       a.rerank->struct_size, sn_abi_version(), sizeof(sn_rerank_args));
            const sn_rerank_args &r = *a.rerank;
            SN_REQUIRE(fast, SN_ERR_UNSUPPORTED, "sn_instance_graph: the deferred S1 finish needs the prediction configuration of the edges kernel");
            SN_REQUIRE(r.x && r.codebook && r.packed && r.workspace && r.ids && r.n_tokens > 0, SN_ERR_BAD_ARG, "sn_instance_graph: incomplete rerank arguments");
            SN_REQUIRE(sn_assign_defers(r.M, r.D) == 1, SN_ERR_UNSUPPORTED, "sn_instance_graph: no deferred S1 finish for M=%d D=%d", r.M, r.D);
            SN_REQUIRE(a.L <= kRowWaves * kOwnMax, SN_ERR_UNSUPPORTED, "sn_instance_graph: deferred S1 finish for L=%d > %d", a.L, kRowWaves * kOwnMax);
            const sn_s1::PackLayout lay = sn_s1::pack_layout(r.M, r.D);
            const unsigned char *ws = (const unsigned char *)r.workspace, *pk = (const unsigned char *)r.packed;
            rv.flags = (const unsigned *)(ws + 32);
            rv.codes = ws + 32 + (size_t)r.n_tokens * 4;
            rv.x = r.x; rv.xsb = r.x_stride_b; rv.xsl = r.x_stride_l; rv.x_bf16 = r.x_bf16;
            rv.tsb = r.tok_stride_b; rv.tsl = r.tok_stride_l;
            rv.cb = r.codebook;
            rv.cn64 = (const double *)(pk + lay.cn64_off);
            rv.tiles = pk + lay.tiles_off;
            rv.scal = (const unsigned *)(pk + lay.scal_off);
            rv.ids = r.ids; rv.isb = r.ids_stride_b; rv.isl = r.ids_stride_l;
            rv.M = r.M; rv.D = r.D; rv.n_tiles = lay.n_tiles;
            rr = r.D / 64;                                        // (sn_assign_defers: D in {192, 384, 768})
        }
        const void *fn = rr == 3 ? (const void *)instance_graph_kernel<true, true, 3>
                       : rr == 6 ? (const void *)instance_graph_kernel<true, true, 6>
                       : rr == 12 ? (const void *)instance_graph_kernel<true, true, 12>
                       : fast ? (const void *)instance_graph_kernel<true, true> : (const void *)instance_graph_kernel<true>;
        int rc = ensure_lds(fn, lds, "sn_instance_graph");
        if (rc) return rc;
        sn_prof_start(2, st);
        if (rr == 3) hipLaunchKernelGGL((instance_graph_kernel<true, true, 3>), dim3(a.B), dim3(1024), lds, st, a, g_graph_stamps, signed_table, rv);
        else if (rr == 6) hipLaunchKernelGGL((instance_graph_kernel<true, true, 6>), dim3(a.B), dim3(1024), lds, st, a, g_graph_stamps, signed_table, rv);
        else if (rr == 12) hipLaunchKernelGGL((instance_graph_kernel<true, true, 12>), dim3(a.B), dim3(1024), lds, st, a, g_graph_stamps, signed_table, rv);
        else if (fast) hipLaunchKernelGGL((instance_graph_kernel<true, true>), dim3(a.B), dim3(1024), lds, st, a, g_graph_stamps, signed_table, rv);
        else hipLaunchKernelGGL(instance_graph_kernel<true>, dim3(a.B), dim3(1024), lds, st, a, g_graph_stamps, signed_table, rv);
        sn_prof_stop(2, st);
    } else {
        const size_t lds = lds_bytes(a.L, false);
        SN_REQUIRE(!a.rerank, SN_ERR_UNSUPPORTED, "sn_instance_graph: the deferred S1 finish needs the edges kernel (attn != NULL)");
        hipLaunchKernelGGL(instance_graph_kernel<false>, dim3(a.B), dim3(256), lds, st, a, (unsigned long long *)nullptr, 0, sn_s1::RerankView{});
    }
    SN_CHECK_LAUNCH("sn_instance_graph");
    return SN_OK;
}

/* diagnostics: device buffer of 16 x u64 per image (slots 0 .. 15) for instance_graph_kernel<true> (NULL = off) */
extern "C" void sn_debug_set_graph_stamps(void *device_buffer) { g_graph_stamps = (unsigned long long *)device_buffer; }

extern "C" int sn_full_vertices(const int64_t *ingredients, int64_t ing_stride_b, int64_t ing_stride_l,
                                const float *attn_cls, int B, int L, int M, int is_logits, int use_clamp,
                                float clamp, int mean, int ingredients_only, const float *w_v,
                                float *out_attr2, float *out_v, void *stream)
{
    SN_REQUIRE(B >= 0 && L > 0 && M > 0, SN_ERR_BAD_ARG, "sn_full_vertices: bad B=%d L=%d M=%d", B, L, M);
    if (B == 0) return SN_OK;
    SN_REQUIRE(ingredients, SN_ERR_BAD_ARG, "sn_full_vertices: ingredients is NULL");
    SN_REQUIRE(ingredients_only || attn_cls, SN_ERR_BAD_ARG, "sn_full_vertices: attn_cls is NULL");
    SN_REQUIRE(L <= SN_MAX_TOKENS, SN_ERR_UNSUPPORTED, "sn_full_vertices: L=%d > %d", L, SN_MAX_TOKENS);
    SN_REQUIRE(out_attr2 || out_v, SN_ERR_BAD_ARG, "sn_full_vertices: no output");
    SN_REQUIRE(!out_v || w_v, SN_ERR_BAD_ARG, "sn_full_vertices: out_v needs w_v");
    hipLaunchKernelGGL(full_vertices_kernel, dim3(B), dim3(256), lds_bytes(L, false), (hipStream_t)stream,
                       ingredients, ing_stride_b, ing_stride_l, attn_cls, L, M, is_logits, use_clamp, clamp,
                       mean, ingredients_only, w_v, out_attr2, out_v);
    SN_CHECK_LAUNCH("sn_full_vertices");
    return SN_OK;
}

extern "C" int sn_limited_edges(const int64_t *ingredients, int64_t ing_stride_b, int64_t ing_stride_l,
                                const float *attn, int B, int L, int is_logits, int use_clamp, float clamp,
                                const float *geo, int feat_h, int feat_w, float dist_alpha, float dist_pow,
                                const int32_t *class_slot, int K, int Mtab, const int64_t *label, int n_max,
                                int mean, int remove_self_loop, const float *w_e, float *out_attr2,
                                float *out_e, void *stream)
{
    SN_REQUIRE(B >= 0 && L > 0 && K > 0 && Mtab > 0 && n_max > 0, SN_ERR_BAD_ARG,
               "sn_limited_edges: bad B=%d L=%d K=%d Mtab=%d n_max=%d", B, L, K, Mtab, n_max);
    if (B == 0) return SN_OK;
    SN_REQUIRE(ingredients && attn && class_slot && label, SN_ERR_BAD_ARG, "sn_limited_edges: NULL input");
    SN_REQUIRE(L <= SN_MAX_TOKENS, SN_ERR_UNSUPPORTED, "sn_limited_edges: L=%d > %d", L, SN_MAX_TOKENS);
    SN_REQUIRE(geo || (feat_h > 0 && feat_w > 0 && feat_h * feat_w == L && feat_h < 256 && feat_w < 256 && dist_alpha > 0.0f),
               SN_ERR_BAD_ARG, "sn_limited_edges: bad grid %dx%d for L=%d", feat_h, feat_w, L);
    SN_REQUIRE(out_attr2 || out_e, SN_ERR_BAD_ARG, "sn_limited_edges: no output");
    SN_REQUIRE(!out_e || w_e, SN_ERR_BAD_ARG, "sn_limited_edges: out_e needs w_e");
    const size_t lds = lds_bytes(L, true);
    int rc = ensure_lds((const void *)limited_edges_kernel, lds, "sn_limited_edges");
    if (rc) return rc;
    hipLaunchKernelGGL(limited_edges_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, ingredients,
                       ing_stride_b, ing_stride_l, attn, L, is_logits, use_clamp, clamp, geo, feat_w,
                       dist_alpha, dist_pow, class_slot, Mtab, label, n_max, mean, remove_self_loop, w_e,
                       out_attr2, out_e);
    SN_CHECK_LAUNCH("sn_limited_edges");
    return SN_OK;
}

extern "C" int sn_stats_accumulate(const float *feat, const int64_t *label, int B, int64_t F, int K,
                                   float *class_sum, float *n_tracked, void *stream)
{
    SN_REQUIRE(B >= 0 && F > 0 && K > 0, SN_ERR_BAD_ARG, "sn_stats_accumulate: bad B=%d F=%lld K=%d", B, (long long)F, K);
    if (B == 0) return SN_OK;
    SN_REQUIRE(feat && label && class_sum, SN_ERR_BAD_ARG, "sn_stats_accumulate: NULL pointer");
    SN_REQUIRE(K <= 65535, SN_ERR_UNSUPPORTED, "sn_stats_accumulate: K=%d > 65535", K);
    const dim3 grid((unsigned)((F + 1023) / 1024), (unsigned)K);
    hipLaunchKernelGGL(stats_accumulate_kernel, grid, dim3(256), 0, (hipStream_t)stream, feat, label, B, F,
                       class_sum, n_tracked);
    SN_CHECK_LAUNCH("sn_stats_accumulate");
    return SN_OK;
}

extern "C" int sn_head_mean_attention(const float *extracted, int B, int H, int L, float *attn,
                                      float *attn_cls, void *stream)
{
    SN_REQUIRE(B >= 0 && H > 0 && L > 0, SN_ERR_BAD_ARG, "sn_head_mean_attention: bad B=%d H=%d L=%d", B, H, L);
    if (B == 0) return SN_OK;
    SN_REQUIRE(extracted && attn && attn_cls, SN_ERR_BAD_ARG, "sn_head_mean_attention: NULL pointer");
    SN_REQUIRE(L <= 256, SN_ERR_UNSUPPORTED, "sn_head_mean_attention: L=%d > 256", L);
    SN_REQUIRE(B <= 65535, SN_ERR_UNSUPPORTED, "sn_head_mean_attention: B=%d > 65535", B);
    hipLaunchKernelGGL(head_mean_kernel, dim3(L + 1, B), dim3(256), 0, (hipStream_t)stream, extracted, H, L,
                       attn, attn_cls);
    SN_CHECK_LAUNCH("sn_head_mean_attention");
    return SN_OK;
}
