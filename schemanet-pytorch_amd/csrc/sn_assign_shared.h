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
    size_t tiles_off, cn64_off, scal_off, frag2_off, hn2_off, total;
    int n_tiles, n_steps, tile_bytes, m_pad;
    int nt2, ks2;                       // nt2 == 0: no register-stationary image for this shape
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
    p.frag2_off = p.scal_off + 256;
    // the whole fp16 codebook must fit the register file of one CU: 4 waves x nt2 x ks2 fragments of
    // 4 registers, at most 96 fragments per wave
    p.ks2 = D / 16;
    p.nt2 = M <= 256 ? 2 : (M <= 512 ? 4 : 0);
    if (D % 32 != 0 || (p.ks2 != 12 && p.ks2 != 24) || p.nt2 * p.ks2 > 96) p.nt2 = 0;
    p.hn2_off = p.frag2_off + (size_t)4 * p.nt2 * p.ks2 * 1024;
    p.total = p.hn2_off + (((size_t)4 * p.nt2 * 128 + 255) & ~size_t(255));
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

}  // namespace sn_s1
