// What the S1 screen (csrc/sn_assign.hip) and a consumer that finishes its flagged tokens (the instance-graph kernel,
// csrc/sn_graph.hip) share: the layout of the packed codebook image, the per-token record format of the
// token-stationary screen, and the fp64 re-rank arithmetic in the oracle's order (oracle/schemanet_oracle.c::
// sno_assign_words: lane l accumulates k = l, l + 64, ... of x . c in fp64, xor butterfly, score = |c|^2 - 2 x . c,
// lowest index on ties).  Reference op: discretization/discretization.py:65 (torch.cdist(...).argmin).
#pragma once

#include "sn_common.h"

namespace sn_s1 {

constexpr int kMaxCand = 24;            // 2 half-lanes x 4 accumulator groups x top-3
constexpr int kCodeBytes = 24;          // per-token candidate record: one key code per candidate slot
// flag word of a token (workspace + 32 + 4 n):
//   0                      final: the screen's word is provably the nearest
//   24-bit mask            candidate slots inside the error window (bit c = slot c = 12 h + 3 g + j: half-lane h, accumulator
//                          group g, j-th smallest key of that group); the record holds code (tile << 2 | e) of every slot:
//                          word = 32 tile + 8 g + 4 h + e
//   bit 31 + 24-bit mask   as above, and a sorted triple lies inside the window whole (bits 3 (4 h + g) .. + 2 all set): a 4th
//                          word of that group - the 64 words 32 t + 8 g + 4 h + e - may hide behind it ("overflow")
//   bits 31 and 30         the screen could not bound the token at all (non-finite / huge components): every word is a candidate
constexpr unsigned kFlagOverflow = 0x80000000u;
constexpr unsigned kFlagFullScan = 0xC0000000u;

struct PackLayout {
    size_t tiles_off, cn64_off, scal_off, tiles5_off, total;      // tiles5_off == 0: no permuted image for this shape
    int n_tiles, n_steps, tile_bytes, m_pad;
};

__host__ __device__ inline PackLayout pack_layout(int M, int D)
{
    PackLayout p;
    p.n_tiles = 2 * ((M + 63) / 64);      // even: the screen kernel walks tiles in pairs; padding words carry |c|^2 = inf
    p.n_steps = D / 16;
    p.m_pad = p.n_tiles * 32;
    p.tile_bytes = (p.n_steps + 1) * 1024;
    p.tiles_off = 0;
    p.cn64_off = (size_t)p.n_tiles * p.tile_bytes;
    p.scal_off = p.cn64_off + (((size_t)p.m_pad * 8 + 255) & ~size_t(255));
    // the word-permuted fragment image of the one-round K-outer screen (csrc/sn_assign.hip, tiles5): codebooks of 16 tiles
    const bool has5 = p.n_tiles == 16 && (D == 192 || D == 384);
    p.tiles5_off = has5 ? p.scal_off + 256 : 0;
    p.total = p.scal_off + 256 + (has5 ? (size_t)16 * p.n_steps * 1024 : 0);
    return p;
}

// Device view of a deferred finish: sn_assign_words(mode = 2) has run the screen, the flag words and candidate records of
// its workspace are still to be resolved.  flags == nullptr: nothing deferred.
struct RerankView {
    const unsigned *flags;        // [n_tokens]
    const unsigned char *codes;   // [n_tokens][kCodeBytes]
    const void *x;                // token (b, l): row at element offset b * xsb + l * xsl (fp32, or bf16 when x_bf16)
    int64_t xsb, xsl;
    int64_t tsb, tsl;             // flat token index of (b, l) in the screen's grid: n = b * tsb + l * tsl
    const float *cb;              // [M, D] fp32 codebook
    const double *cn64;           // [M] |c|^2 in fp64 (packed image)
    const unsigned char *tiles;   // packed fp16 tile image (overflow scan)
    const unsigned *scal;         // its scalars: max |c|_2, max |c|_1, max |c|^2, max |c_mk| (float bits)
    int64_t *ids;                 // where the final word of (b, l) goes: ids[b * isb + l * isl] (the screen's own output)
    int64_t isb, isl;
    int M, D, x_bf16, n_tiles;
};

// element k of a token row as fp32 (bf16 -> fp32 is exact)
__device__ __forceinline__ float token_elem(const void *row, int x_bf16, int k)
{
    return x_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(row)[k] << 16) : reinterpret_cast<const float *>(row)[k];
}

__device__ __forceinline__ const void *token_row_ptr(const void *x, int x_bf16, int64_t elem_off)
{
    return x_bf16 ? (const void *)(reinterpret_cast<const unsigned short *>(x) + elem_off) : (const void *)(reinterpret_cast<const float *>(x) + elem_off);
}

// fp64 dot in the oracle's order: lane l accumulates k = l, l+64, ... then xor-butterfly.
template <int NT>
__device__ __forceinline__ double dot64(const double (&x)[NT], const float *c, int D, int lane)
{
    double p = 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int k = lane + SN_WAVE * t;
        if (k < D) p = fma(x[t], (double)c[k], p);
    }
    return sn_wave_sum_f64(p);
}

// word of candidate slot c (0 .. 23) from its code byte (records of the token-stationary screen, 8-bit codes)
__device__ __forceinline__ int slot_word(int c, unsigned code)
{
    const int hh = c / 12, g = (c % 12) / 3;
    return (int)(code >> 2) * 32 + 8 * g + 4 * hh + (int)(code & 3u);
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

// An overflow token (a handful per 50 000), by ONE wave.  Callers: the re-rank kernel (csrc/sn_assign.hip: its first blocks
// take the overflow list, one wave per token) and, in the deferred finish, the sorting wave of the instance-graph kernel
// (csrc/sn_graph.hip), in front of its wait for the row waves.  Candidates = the slots of the token's mask, plus - for every group
// whose three slots are all inside the window - whatever a scan of the group's words 32 t + 8 g + 4 h + e through the
// fp16 tile image (v_dot2_f32_f16, one word per lane, the token's fp16 pairs broadcast from registers) leaves inside
// the rigorous fp16 window of the group's best (the window of the round-1..3 overflow scan; the group's best is no better
// than the token's, so nothing that could win is cut); every word when the screen could not bound the token.
// Returns the word (0 for an all-NaN row, like the oracle), or -1: keep the screen's.  my_word: word of candidate slot `lane`.
template <int NT>
__device__ __forceinline__ int rerank_overflow_token(const RerankView &rv, int b, int l, int lane, unsigned fj, int my_word)
{
    float xf[NT];
    const void *row = token_row_ptr(rv.x, rv.x_bf16, (int64_t)b * rv.xsb + (int64_t)l * rv.xsl);
#pragma unroll
    for (int t = 0; t < NT; ++t) xf[t] = token_elem(row, rv.x_bf16, lane + SN_WAVE * t);       // (D == 64 NT)
    double best = (double)INFINITY;
    int bi = 0x7fffffff;
    auto eval2 = [&](int ma, int m1, bool two) {                   // two words per round: their loads overlap
        const int mb = two ? m1 : ma;
        const float *ra = rv.cb + (int64_t)ma * rv.D, *rb = rv.cb + (int64_t)mb * rv.D;
        double pa = 0.0, pb = 0.0;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int k = lane + SN_WAVE * t;
            pa = fma((double)xf[t], (double)ra[k], pa); pb = fma((double)xf[t], (double)rb[k], pb);
        }
        const double sa = rv.cn64[ma] - 2.0 * sn_wave_sum_f64(pa);
        const double sb = rv.cn64[mb] - 2.0 * sn_wave_sum_f64(pb);
        if (sa < best || (sa == best && ma < bi)) { best = sa; bi = ma; }
        if (two && (sb < best || (sb == best && mb < bi))) { best = sb; bi = mb; }
    };
    if ((fj & kFlagFullScan) == kFlagFullScan) {
        for (int m = 0; m < rv.M; m += 2) eval2(m, m + 1, m + 1 < rv.M);
        return bi != 0x7fffffff ? bi : 0;                          // all-NaN row -> 0 (oracle)
    }
    const unsigned mask = fj & 0xFFFFFFu;
    // the token's statistics and the fp16 window (fp16 rounding of both operands, subnormal flush, fp32 accumulation of the dot2 chain, rounding of |c|^2/2)
    float sq = 0.0f, sabs = 0.0f, mabs = 0.0f;
    unsigned xh2[NT];                                               // even lanes: fp16 pair (x[k], x[k + 1]), k = lane + 64 t
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const float v = xf[t];
        sq = fmaf(v, v, sq); sabs += fabsf(v); mabs = fmaxf(mabs, fabsf(v));
        const float nx = __shfl_down(v, 1, SN_WAVE);
        half2_t hp; hp.x = (_Float16)v; hp.y = (_Float16)nx;
        xh2[t] = __builtin_bit_cast(unsigned, hp);
    }
    const float X2 = sqrtf(sn_wave_sum(sq)) * 1.001f, X1 = sn_wave_sum(sabs) * 1.001f, XM = sn_wave_max(mabs);
    const float C2 = __uint_as_float(rv.scal[0]), C1 = __uint_as_float(rv.scal[1]);
    const float CN = __uint_as_float(rv.scal[2]), CMAX = __uint_as_float(rv.scal[3]);
    const float vmax = 0.5f * CN + X2 * C2;
    const float e16 = 1.01f * (2.01f * 4.8828125e-4f * X2 * C2 + 5.96e-8f * (X1 + C1) +
                               2.0f * (float)(rv.D + 2) * 5.9604645e-8f * vmax + 1.2e-7f * vmax);
    const bool finite = (XM <= 3.0e4f) && (CMAX <= 3.0e4f) && (e16 < 1.0e30f);
    constexpr int kSteps = 4 * NT;                                   // D / 16
    const int tile_bytes = (kSteps + 1) * 1024;
    for (int G = 0; G < 8; ++G) {                                   // G = 4 h + g
        const unsigned bits = (mask >> (3 * G)) & 7u;
        if (bits != 7u) {
            for (int j = 0; j < 3; ++j)
                if ((bits >> j) & 1u) eval2(__builtin_amdgcn_readlane(my_word, 3 * G + j), 0, false);
            continue;
        }
        const int g = G & 3, hh = G >> 2;
        for (int t0 = 0; t0 < rv.n_tiles; t0 += 16) {              // 16 tiles x 4 rows = one word per lane
            const int tile = t0 + (lane >> 2), i = 8 * g + 4 * hh + (lane & 3), m = 32 * tile + i;
            const bool in = tile < rv.n_tiles && m < rv.M;
            const unsigned char *ta = rv.tiles + (size_t)(in ? tile : 0) * tile_bytes;
            float a0 = 0.0f, a1 = 0.0f;
#pragma unroll
            for (int s0 = 0; s0 < kSteps; s0 += 2) {                // two k-steps = 32 consecutive k (half of xh2[s0 / 4]) per round of loads
                // (opaque copy: the v_readlane of a round are otherwise hoisted out of the group / tile loops and live in
                // scalar registers for the whole kernel; the scheduling barrier keeps the rounds one after the other: all
                // 8 NT 16-byte loads in flight at once do not fit the register file)
                unsigned xt = xh2[s0 / 4];
                asm volatile("" : "+v"(xt));
                __builtin_amdgcn_sched_barrier(0);
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));      // (an array of float4 STRUCTS lives in scratch memory)
                u32x4 fr[4];
#pragma unroll
                for (int q = 0; q < 4; ++q)                          // (step s0 + (q >> 1), lane half q & 1)
                    fr[q] = *reinterpret_cast<const u32x4 *>(ta + (size_t)(s0 + (q >> 1)) * 1024 + (i + 32 * (q & 1)) * 16);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int sstep = s0 + (q >> 1);
                    const int k0 = 32 * (sstep >> 1) + 16 * (q & 1) + 8 * (sstep & 1);     // pack_frag_kernel: s = 2 u + e, k = 32 u + 16 h + 8 e + j
#pragma unroll
                    for (int jp = 0; jp < 4; ++jp) {
                        const int k = k0 + 2 * jp;                  // (k >> 6 == s0 / 4)
                        const unsigned xs = (unsigned)__builtin_amdgcn_readlane((int)xt, k & 63);
                        const half2_t xv = __builtin_bit_cast(half2_t, xs), cv = __builtin_bit_cast(half2_t, (unsigned)fr[q][jp]);
                        if (jp & 1) a1 = __builtin_amdgcn_fdot2(xv, cv, a1, false);
                        else a0 = __builtin_amdgcn_fdot2(xv, cv, a0, false);
                    }
                }
                asm volatile("" : "+v"(a0), "+v"(a1));              // (the sums are formed HERE, in every lane: left alone the products sink into the branch of the select below, their 32 NT broadcast operands spilled on the way)
                __builtin_amdgcn_sched_barrier(0);
            }
            const float hn = *reinterpret_cast<const float *>(ta + (size_t)kSteps * 1024 + ((g * 2 + hh) * 4 + (lane & 3)) * 4);
            const float sc = (in && finite) ? hn + (a0 + a1) : INFINITY;       // dist^2/2 - |x|^2/2 (tiles hold -c)
            const float smin = sn_wave_min(sc);
            // (nothing finite: every word of the group goes to fp64)
            unsigned long long sv = __ballot(in && (!(smin < INFINITY) || sc <= smin + 2.0f * e16));
            while (sv) {
                const int la = __ffsll((long long)sv) - 1;
                sv &= sv - 1;
                const bool has1 = sv != 0ull;
                const int lb = has1 ? __ffsll((long long)sv) - 1 : la;
                if (has1) sv &= sv - 1;
                eval2(__builtin_amdgcn_readlane(m, la), __builtin_amdgcn_readlane(m, lb), has1);
            }
        }
    }
    return bi != 0x7fffffff ? bi : -1;
}

}  // namespace sn_s1
