// S1: nearest-visual-word assignment (replaces torch.cdist(...).argmin(1),
// reference discretization/discretization.py:58-70).
//
// The token x codebook distance matrix is the one genuinely dense contraction on the path, so
// it runs on the matrix cores: fp16 MFMA (v_mfma_f32_32x32x16_f16, fp32 accumulate) computes
//   acc'[word, token] = OFF_token - |c_word|^2 / 2 + x_token . c_word      (= OFF - dist^2/2 + |x|^2/2)
// for a 32-token tile per wave held stationary in registers, against 32-word codebook tiles
// streamed L2 -> LDS with global_load_lds (the packed image is laid out in fragment order, so
// the copy is linear and every ds_read_b128 is conflict-free).  fp16 rounding cannot decide
// near-ties, so the MFMA pass is only a SCREEN: each lane keeps its three largest acc' as
// packed (value | 8-bit word code) keys, and every token whose runner-up lies within a
// rigorous error window of the best is appended to a work list.  A second small kernel
// re-ranks exactly those candidates in fp64 with the summation order of the oracle
// (oracle/schemanet_oracle.c: 64-way strided partial sums + xor butterfly), so the final
// index is bit-identical to the oracle for every token.  mode 1 skips the screen and scans
// every word in fp64 (slow; fallback for shapes the screen is not built for, and cross-check).
//
// HBM traffic (algorithmic): tokens read once (D*4 B each) + 8 B index out; the packed
// codebook (M*D*2 B) stays L2 resident.
#include "sn_common.h"
#include "sn_assign_shared.h"

#include <hip/hip_fp16.h>
#include <stdlib.h>
#include <utility>

#ifndef SN_S1_EXACT_LOSS
#define SN_S1_EXACT_LOSS 1  // window from the token's measured fp16 rounding loss |x - fp16(x)|_2 (0: from its bound u |x|_2: two VALU per pair less in the token phase, 1.4 x as many tokens to re-rank)
#endif
#ifndef SN_S1_TOKENS_NT
#define SN_S1_TOKENS_NT 1   // token rows with the non-temporal hint: read once by the screen (the re-rank re-reads the 6.6 % it flags: +0.4 us there, -1.8 us here, +1.1 % on the replayed bench; round 1, with 13 % flagged, it lost)
#endif
#ifndef SN_S1_STAGE
#define SN_S1_STAGE 1       // token rows through LDS in whole cache lines (0: fragment loads straight from global memory)
#endif
namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kTokPerWave = 32;
constexpr int kWavesPerBlock = 4;
using sn_s1::kMaxCand;                  // 24 = 2 half-lanes x 4 accumulator groups x top-3
using sn_s1::kCodeBytes;                // per-token candidate record: one key code per candidate slot (sn_assign_shared.h)
using sn_s1::PackLayout;
using sn_s1::pack_layout;
using sn_s1::dot64;
// workspace per token: token-stationary records = flag word + 24 code bytes + overflow-list slot (32 B)
constexpr int kWsPerToken2 = 4 + 48 + 4;          // the largest record format: flag word + 24 16-bit codes + overflow-list slot
constexpr int kMaxTilesScreen = 256;    // tile code in the keys: 6 bits (M <= 2048, byte codes) or 8 bits (M <= 8192, 16-bit codes)
constexpr int kCodeBytesWide = 48;      // candidate record with 16-bit codes
constexpr float kU16 = 4.8828125e-4f;  // 2^-11, fp16 unit round-off
constexpr float kHugeIn = 3.0e4f;      // |value| above this does not go through fp16
// fp32 accumulate of v_mfma_f32_32x32x16_f16: measured (tools/mfma_probe.hip, MI355X) total error
// after 24 chained MFMAs <= 12.3 x 2^-24 x max|partial sum| (about 0.5 per instruction).  The
// window below budgets 8 per instruction (15x the observed total).
constexpr float kAccUlpPerMfma = 8.0f * 5.9604645e-8f;

// packed codebook image -------------------------------------------------------------------
//   tiles   [n_tiles][n_steps + 1][1 KiB]   per 32-word tile: n_steps chunks of fp16 MFMA A-fragments
//                                           holding -c (negated), then one chunk whose first 128 B
//                                           are |c|^2 / 2 (fp32) in accumulator-row order
//   cn64    [M_pad] f64  |c|^2 (oracle summation order)
//   scal    [0] max |c|_2  [1] max |c|_1  [2] max |c|^2  [3] max |c_mk|  [4] max |c - fp16(c)|_2   (uint bits of floats)
//   frag2   [4 waves][nt2 tiles][ks2 k-steps][1 KiB]   (register-stationary screen, see assign_screen2_kernel)
//                                           v_mfma_f32_32x32x16_f16 A-fragments of -c: wave q owns words
//                                           [32 nt2 q, 32 nt2 (q+1)), tile a = 32 of them, lane (r, h) of k-step js
//                                           holds word row r at k = 32 (js / 2) + {4g..4g+3, 16+4g..16+4g+3},
//                                           g = 2 (js & 1) + h (the order the token converter produces)
//   hn2     [4][nt2][2 halves][16] f32      |c|^2 / 2 in accumulator-register order (padding words: 1e30)
constexpr float kPadHalfNorm = 1.0e30f;     // |c|^2/2 of the padding words of the frag2 image (finite: keys stay ordered floats)

// ------------------------------------------------------------------------------------------
// codebook_prepare
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_frag_kernel(const float *cb, int M, int D, unsigned char *tiles,
                                                        int n_steps, int tile_bytes)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over [M_pad, D]
    const int m = (int)(idx / D), k = (int)(idx % D);
    const int w = m >> 5, i = m & 31;
    // k -> (step s, lane half h, element j): a lane's 16 consecutive floats feed two k-steps
    const int u = k >> 5, rem = k & 31, h = rem >> 4, e = (rem >> 3) & 1, j = rem & 7;
    const int s = 2 * u + e;
    const float v = m < M ? cb[(int64_t)m * D + k] : 0.0f;
    _Float16 *frag = (_Float16 *)(tiles + (size_t)w * tile_bytes + (size_t)s * 1024);
    frag[(i + 32 * h) * 8 + j] = (_Float16)(-v);
}

// frag2 image: one thread per (word of the padded codebook, k)
__device__ __forceinline__ int hn2_index(int m, int nt2)     // word m -> slot of its |c|^2/2 in hn2
{
    const int q = m / (32 * nt2), a = (m / 32) % nt2, r = m & 31;
    return ((q * nt2 + a) * 2 + ((r >> 2) & 1)) * 16 + (r & 3) + 4 * (r >> 3);       // accumulator row = (reg & 3) + 8 (reg >> 2) + 4 half
}

__global__ __launch_bounds__(256) void pack_frag2_kernel(const float *cb, int M, int D, unsigned char *frag2, float *hn2, int nt2, int ks2)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over [128 nt2, D]
    const int m = (int)(idx / D), k = (int)(idx % D);
    const int q = m / (32 * nt2), a = (m / 32) % nt2, r = m & 31;
    const int c = k >> 5, rem = k & 31;
    const int g = (rem & 15) >> 2, e = (rem & 3) + 4 * (rem >> 4);
    const int js = 2 * c + (g >> 1), h = g & 1;
    const float v = m < M ? cb[(int64_t)m * D + k] : 0.0f;
    _Float16 *frag = (_Float16 *)(frag2 + ((size_t)(q * nt2 + a) * ks2 + js) * 1024);
    frag[(r + 32 * h) * 8 + e] = (_Float16)(-v);
    if (k == 0 && m >= M) hn2[hn2_index(m, nt2)] = kPadHalfNorm;           // real words: pack_norm_kernel
}

__global__ __launch_bounds__(256) void pack_norm_kernel(const float *cb, int M, int D, int m_pad, unsigned char *tiles,
                                                        int n_steps, int tile_bytes, double *cn64, unsigned *scal, float *hn2, int nt2)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= m_pad) return;
    // half-norm slot of word m inside its tile: accumulator row order [g][h][e], row = e + 8g + 4h
    const int i = m & 31, g = i >> 3, h = (i >> 2) & 1, e = i & 3;
    float *hc = (float *)(tiles + (size_t)(m >> 5) * tile_bytes + (size_t)n_steps * 1024) + (g * 2 + h) * 4 + e;
    if (m >= M) {                       // padding words can never win
        if (lane == 0) { *hc = INFINITY; cn64[m] = (double)INFINITY; }
        return;
    }
    const float *c = cb + (int64_t)m * D;
    double p = 0.0, dq = 0.0;
    float l1 = 0.0f, mx = 0.0f;
    for (int k = lane; k < D; k += SN_WAVE) {
        const float v = c[k];
        p = fma((double)v, (double)v, p);
        l1 += fabsf(v);
        mx = fmaxf(mx, fabsf(v));
        const double dv = (double)v - (double)(float)(_Float16)v;        // what the fp16 image of the word loses (exact)
        dq = fma(dv, dv, dq);
    }
    p = sn_wave_sum_f64(p);
    dq = sn_wave_sum_f64(dq);
    l1 = sn_wave_sum(l1);
    mx = sn_wave_max(mx);
    if (lane == 0) {
        cn64[m] = p;
        *hc = (float)(0.5 * p);
        if (nt2) hn2[hn2_index(m, nt2)] = (float)(0.5 * p);
        const float up = 1.0f + 1.0e-6f;
        atomicMax(&scal[0], __float_as_uint(sqrtf((float)p) * up));
        atomicMax(&scal[1], __float_as_uint(l1 * (1.0f + 1.0e-4f)));
        atomicMax(&scal[2], __float_as_uint((float)p * up));
        atomicMax(&scal[3], __float_as_uint(mx));
        atomicMax(&scal[4], __float_as_uint((float)sqrt(dq) * up + 1.0e-30f));      // max_m |c_m - fp16(c_m)|_2
    }
}

// ------------------------------------------------------------------------------------------
// shared argument block
// ------------------------------------------------------------------------------------------
struct AssignArgs {
    const float *x;
    int64_t n_tokens, n_inner, xso, xsi;
    const float *cb;
    const unsigned char *packed;
    int M, D;
    int64_t *out;
    int64_t oso, osi;
    int *work;          // header: [1] #overflow tokens
    unsigned *flags;    // per token: 0 = final, bit 31 = overflow (full scan), else 24-bit candidate mask
    unsigned char *codes;   // per token 24 key codes (tile << 2 | e), written only for flagged tokens
    int *overflow;      // token ids that need a full scan
    unsigned long long *stamps;   // diagnostics only (sn_debug_set_stamps): 16 u64 slots per wave
    // register-stationary screen (assign_screen2_kernel): the flag word has the same meaning (24-bit
    // candidate mask, bit 3c+j = key j of lane slot c = 2 wave + accumulator half), the codes are 8
    // dwords per token (slot c: code_j << 8j, code = tile << 4 | accumulator register)
    unsigned *codes32;
    int64_t n_sets;     // ceil(n_tokens / 32)
    // token-phase gate of the token-stationary screen (NULL = off): per CU {arrivals, waves whose tokens have
    // landed}; zeroed before every launch.  See assign_screen_kernel.
    unsigned *gate;
    // token -> wave map of the token-stationary screen: waves 0 .. full_waves-1 of workgroup b own the tokens
    // [32 full_waves b + 32 w, +32); wave `full_waves` (if the workgroup has one) owns [extra_base + 32 b, +32);
    // waves without tokens only keep the codebook ring going.  Default: full_waves = waves per workgroup.
    int full_waves;
    int64_t extra_base;
    int tps;            // K-outer screen (assign_screen3_kernel): tokens per set (<= 32); set i holds tokens [i tps, (i + 1) tps)
    int64_t n_sets3;    // ... number of sets, ceil(n_tokens / tps)
    int tps4;           // one-round K-outer screen (assign_screen4_kernel): tokens per workgroup (<= kS4Rows)
    int x_bf16;         // tokens are bfloat16 (x points at 2-byte elements, strides in elements); results are defined on their fp32 values
};

constexpr int kGateSlots = 4096;        // (XCC_ID[3:0] << 8) | HW_ID[15:8] (CU_ID, SH_ID, SE_ID)
constexpr size_t kGateBytes = (size_t)kGateSlots * 8;

__device__ __forceinline__ void stamp(const AssignArgs &p, int slot, int lane, int wave_id)
{
    if (p.stamps && lane == 0) p.stamps[(size_t)wave_id * 16 + slot] = __builtin_amdgcn_s_memtime();
}

static unsigned long long *g_stamps = nullptr;
// (n_tokens < 2^31 is enforced by sn_assign_words: 32-bit quotient / remainder; a vector 64-bit divide is ~150
// instructions and these run once per lane)
__device__ __forceinline__ const float *token_row(const AssignArgs &p, int64_t n)
{
    const unsigned ni = (unsigned)p.n_inner, o = (unsigned)n / ni, i = (unsigned)n - o * ni;
    const int64_t e = (int64_t)o * p.xso + (int64_t)i * p.xsi;                  // element offset of the row
    return p.x_bf16 ? reinterpret_cast<const float *>(reinterpret_cast<const unsigned short *>(p.x) + e) : p.x + e;
}

// element k of a token row as fp32 (bf16 -> fp32 is exact)
__device__ __forceinline__ float token_elem(const AssignArgs &p, const float *row, int k)
{
    return p.x_bf16 ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short *>(row)[k] << 16) : row[k];
}

__device__ __forceinline__ int64_t out_index(const AssignArgs &p, int64_t n)
{
    const unsigned ni = (unsigned)p.n_inner, o = (unsigned)n / ni, i = (unsigned)n - o * ni;
    return (int64_t)o * p.oso + (int64_t)i * p.osi;
}

// exact fp64 scan of words [m0, m1) for one token held in x[]: returns (score, index) with
// first-index tie-break, NaN scores never win (oracle sno_assign_words).
template <int NT>
__device__ __forceinline__ void exact_scan(const double (&x)[NT], const AssignArgs &p, const double *cn64,
                                           int m0, int m1, int lane, double &best, int &bi)
{
    for (int m = m0; m < m1; ++m) {
        const double s = cn64[m] - 2.0 * dot64<NT>(x, p.cb + (int64_t)m * p.D, p.D, lane);
        if (s < best) { best = s; bi = m; }
    }
}

template <int NT>
__device__ __forceinline__ void load_token64(double (&x)[NT], const AssignArgs &p, const float *row, int D, int lane)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int k = lane + SN_WAVE * t;
        x[t] = k < D ? (double)token_elem(p, row, k) : 0.0;
    }
}

// ------------------------------------------------------------------------------------------
// mode 1: exact kernel, one wave per token
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void assign_exact_kernel(const AssignArgs p)
{
    const int lane = threadIdx.x & 63;
    const PackLayout lay = pack_layout(p.M, p.D);
    const double *cn64 = (const double *)(p.packed + lay.cn64_off);
    const int64_t n_waves = (int64_t)gridDim.x * kWavesPerBlock;
    for (int64_t n = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); n < p.n_tokens; n += n_waves) {
        double x[NT];
        load_token64<NT>(x, p, token_row(p, n), p.D, lane);
        double best = (double)INFINITY;
        int bi = 0;
        exact_scan<NT>(x, p, cn64, 0, p.M, lane, best, bi);
        if (lane == 0) p.out[out_index(p, n)] = bi;
    }
}

// ------------------------------------------------------------------------------------------
// mode 0, pass 2: fp64 re-rank (sn_assign_words(mode = 2) does not launch it: the instance-graph kernel finishes the
// flagged tokens of its image itself, csrc/sn_graph.hip)
//   blocks kOverflowBlocks ..    one wave per flagged token: the <= 24 candidates the screen could not separate
//   blocks 0 .. kOverflowBlocks  one WAVE per overflow token (sn_s1::rerank_overflow_token): its candidates plus, for every
//                                group whose triple lies inside the window whole, a v_dot2_f32_f16 scan of the group's 64
//                                words through the fp16 tile image, survivors inside a rigorous window re-ranked in fp64
// All fp64 scores use the oracle's summation order.
// (Round 4.  Rounds 1-3 scanned EVERY word of an overflow token with a whole block - 139 registers under the 80 of
// amdgpu_waves_per_eu(6, 8): every instantiation spilled, 46 VGPRs / 188 bytes of scratch in <6, 0>, a latency-chain kernel
// that needed a scratch segment at dispatch.  As two kernels - flagged / overflow, no scratch in either - the pair took
// 19.3 us instead of 13.0: the two latency chains of ~10 us ran one after the other.  The screen now writes the candidate
// mask of an overflow token too, the scan is limited to the groups that can hide a word - one wave, one kernel.)
// ------------------------------------------------------------------------------------------
constexpr int kOverflowBlocks = 64;     // blocks reserved for the overflow list (4 tokens each per round)

// FMT 0: records of assign_screen_kernel (24 code bytes per token); FMT 1: records of
// assign_screen2_kernel (8 code dwords, slot c = 2 wave + accumulator half, key j: bit 3c + j);
// FMT 2: 16-bit codes (M > 2048); FMT 3: records of assign_screen3_kernel (24 code bytes, its own slot order).
template <int NT, int FMT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void assign_rerank_kernel(const AssignArgs p)
{
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const PackLayout lay = pack_layout(p.M, p.D);
    const double *cn64 = (const double *)(p.packed + lay.cn64_off);
    // a block owns 32 consecutive tokens per round, reads their flag words, and its four waves share the flagged ones
    // round-robin.  Candidate slot c = 12 h + 3 g + j holds code (tile << 2 | e): word = 32 tile + 8 g + 4 h + e.
    const int64_t n_chunks = (p.n_tokens + 31) / 32;
    for (int64_t chunk = (int64_t)blockIdx.x - kOverflowBlocks; chunk < n_chunks && (int)blockIdx.x >= kOverflowBlocks;
         chunk += (int64_t)gridDim.x - kOverflowBlocks) {
        const int64_t t = chunk * 32 + (lane & 31);
        unsigned long long flag = 0ull;
        if (lane < 32 && t < p.n_tokens) {
            const unsigned f = p.flags[t];
            flag = (f >> 31) ? (1ull << 63) : (unsigned long long)f;
        }
        unsigned long long todo = __ballot(flag != 0ull && !(flag >> 63));
        for (int i = 0; todo; ++i) {
            const int tl = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            if ((i & 3) != wid) continue;
            const int64_t n = chunk * 32 + tl;
            const unsigned long long cmask = ((unsigned long long)(unsigned)__shfl((int)(flag >> 32), tl, SN_WAVE) << 32) |
                                             (unsigned long long)(unsigned)__shfl((int)flag, tl, SN_WAVE);
            int my_word = 0;
            if constexpr (FMT == 0 || FMT == 2) {
                if (lane < kMaxCand) {
                    const unsigned code = FMT == 0 ? (unsigned)p.codes[n * kCodeBytes + lane]
                                                   : (unsigned)reinterpret_cast<const unsigned short *>(p.codes)[n * kMaxCand + lane];
                    const int hh = lane / 12, g = (lane % 12) / 3;
                    my_word = (int)(code >> 2) * 32 + 8 * g + 4 * hh + (int)(code & 3u);
                }
            } else if constexpr (FMT == 3) {
                // records of assign_screen3_kernel: slot c = 3 (2 q + h) + j, code = tile << 4 | accumulator register
                if (lane < kMaxCand) {
                    const unsigned code = (unsigned)p.codes[n * kCodeBytes + lane];
                    const int qh = lane / 3, reg = (int)(code & 15u);
                    my_word = ((qh >> 1) * (lay.n_tiles / 4) + (int)(code >> 4)) * 32 + 8 * (reg >> 2) + 4 * (qh & 1) + (reg & 3);
                }
            } else {
                if (lane < 24 && ((cmask >> lane) & 1ull)) {          // only slots with a candidate were written
                    const int c = lane / 3, jj = lane % 3;
                    const unsigned code = (p.codes32[n * 8 + c] >> (8 * jj)) & 0xFFu;
                    const int reg = (int)(code & 15u);
                    my_word = (c >> 1) * (32 * lay.nt2) + (int)(code >> 4) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * (c & 1);
                }
            }
            double x[NT];
            load_token64<NT>(x, p, token_row(p, n), p.D, lane);
            double best = (double)INFINITY;
            int bi = 0x7fffffff;
            for (unsigned long long cm = cmask; cm;) {           // two candidates per round: their loads overlap
                const int ca = __ffsll((long long)cm) - 1;
                cm &= cm - 1;
                const bool two = cm != 0ull;
                const int cb2 = two ? __ffsll((long long)cm) - 1 : ca;
                if (two) cm &= cm - 1;
                const int ma = __shfl(my_word, ca, SN_WAVE), mb = __shfl(my_word, cb2, SN_WAVE);
                const float *ra = p.cb + (int64_t)ma * p.D, *rb = p.cb + (int64_t)mb * p.D;
                double pa = 0.0, pb = 0.0;
#pragma unroll
                for (int t2 = 0; t2 < NT; ++t2) {
                    const int k = lane + SN_WAVE * t2;
                    if (k < p.D) { pa = fma(x[t2], (double)ra[k], pa); pb = fma(x[t2], (double)rb[k], pb); }
                }
                const double sa = cn64[ma] - 2.0 * sn_wave_sum_f64(pa);
                const double sb = cn64[mb] - 2.0 * sn_wave_sum_f64(pb);
                if (sa < best || (sa == best && ma < bi)) { best = sa; bi = ma; }
                if (two && (sb < best || (sb == best && mb < bi))) { best = sb; bi = mb; }
            }
            if (bi != 0x7fffffff && lane == 0) p.out[out_index(p, n)] = bi;
            // (all candidates NaN cannot happen: such tokens are routed to the overflow list)
        }
    }

    // ---- the overflow list: blocks 0 .. kOverflowBlocks - 1, one wave per token
    if ((int)blockIdx.x >= kOverflowBlocks) return;
    const int n_over = p.work[1];
    for (int e = blockIdx.x * 4 + wid; e < n_over; e += kOverflowBlocks * 4) {
        const int64_t n = p.overflow[e];
        sn_s1::RerankView rv;
        rv.x = token_row(p, n); rv.xsb = 0; rv.xsl = 0; rv.x_bf16 = p.x_bf16;
        rv.cb = p.cb; rv.cn64 = cn64; rv.tiles = p.packed + lay.tiles_off; rv.scal = (const unsigned *)(p.packed + lay.scal_off);
        rv.M = p.M; rv.D = p.D; rv.n_tiles = lay.n_tiles;
        unsigned fj = p.flags[n];
        int my_word = 0;
        if constexpr (FMT == 0) {
            if (lane < kMaxCand) my_word = sn_s1::slot_word(lane, (unsigned)p.codes[n * kCodeBytes + lane]);
        } else if constexpr (FMT == 2) {            // 16-bit codes (tile < 256): the same word formula
            if (lane < kMaxCand) my_word = sn_s1::slot_word(lane, (unsigned)reinterpret_cast<const unsigned short *>(p.codes)[n * kMaxCand + lane]);
        } else {
            fj = sn_s1::kFlagFullScan;              // (the other record formats carry no candidate mask for overflow tokens: every word)
        }
        const int w = sn_s1::rerank_overflow_token<NT>(rv, 0, 0, lane, fj, my_word);
        if (lane == 0 && w >= 0) p.out[out_index(p, n)] = w;
    }
}

// ------------------------------------------------------------------------------------------
// mode 0, pass 1: fp16-MFMA screen
// Each lane tracks, for its token and its half of the words, the three SMALLEST values of
//   v[word] = |c|^2/2 + (|x|^2/2 + 2E) - x.c      (= dist^2/2 + 2E  >= 0)
// as packed keys (float bits with the low 8 mantissa bits replaced by a word code).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned med3u(unsigned a, unsigned b, unsigned c)
{
    return max(min(a, b), min(max(a, b), c));    // v_med3_u32
}

// NW waves per workgroup (32 tokens each) share one codebook-tile ring of R LDS slots.
// CB = width of the word code in a key: 8 (tile < 64: M <= 2048) or 10 (tile < 256: M <= 8192; the keys lose two
// more mantissa bits, which the error window accounts for, and the candidate records hold 16-bit codes).
// DUAL: the k-steps of a tile alternate between TWO accumulator chains (even steps: `cur`, initialised with |c|^2/2 +
// shift; odd steps: `accQ`, started from C = 0), summed once per tile before the keys are formed.  A wave alone on its
// SIMD issues a dependent chain at one v_mfma_f32_32x32x16_f16 per 46 cycles, two independent chains at one per 32
// (tools/mfma_loop_probe.hip): with the token-phase gate a workgroup is alone on its CU for most of its main loop.
// Program order of the steps: 0, 2, [sum of the previous tile], 1, 4, 3, 6, 5, ... - the odd chain lags by one step so
// that the sum (which reads accQ) sits two MFMA issues behind the previous tile's last odd step and ahead of the new
// tile's first one.
template <int NSTEPS, int NW, int R, int CB = 8, bool DUAL = false>
__global__ __launch_bounds__(64 * NW, (NSTEPS <= 24 ? 2 : 1)) void assign_screen_kernel(const AssignArgs p)
{
    constexpr unsigned kCodeMask = (1u << CB) - 1u, kTileMask = (1u << (CB - 2)) - 1u;
    constexpr float kKeyTrunc = CB == 8 ? 3.1e-5f : 1.23e-4f;                 // 2^-15 / 2^-13
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kChunks = NSTEPS + 1;
    constexpr int kTileBytes = kChunks * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const PackLayout lay = pack_layout(p.M, p.D);
    const unsigned char *tiles = p.packed + lay.tiles_off;
    const unsigned *scal = (const unsigned *)(p.packed + lay.scal_off);
    const int n_tiles = lay.n_tiles;

    // codebook tiles: L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no
    // VGPRs).  Issued through inline asm and waited for by hand: when hipcc sees an LDS-DMA it
    // drains it with vmcnt(0) before every later ds_read, which serialises the prefetch.  Wave w
    // copies chunks w, w + NW, w + 2 NW, ...: kDmaMin of them, one more on the first kDmaExtra waves.
    constexpr int kDmaMin = kChunks / NW, kDmaExtra = kChunks % NW;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // wave w copies the contiguous chunks [c0, c0 + kDmaMin (+1 on the first kDmaExtra waves)).  The instruction
    // offset of global_load_lds moves the LDS address together with the global one, so up to four pieces
    // (offsets 0, 1, 2, 3 KiB) share one address register pair and one M0 value: one asm statement per group.
    const int dma_c0 = wid * kDmaMin + (wid < kDmaExtra ? wid : kDmaExtra);
    auto issue_tile = [&](int w, int slot) {
        const unsigned char *src = tiles + (size_t)w * kTileBytes + dma_c0 * 1024 + lane * 16;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * kTileBytes + dma_c0 * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
#pragma unroll
        for (int j = 0; j < kDmaMin; j += 4) {
            const unsigned char *sj = src + (size_t)j * 1024;
            const unsigned dj = dst + j * 1024;
            if (kDmaMin - j >= 4)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, off offset:2048\n\tglobal_load_lds_dwordx4 %0, off offset:3072" :: "v"(sj), "s"(dj) : "memory");
            else if (kDmaMin - j == 3)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, off offset:2048" :: "v"(sj), "s"(dj) : "memory");
            else if (kDmaMin - j == 2)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\tglobal_load_lds_dwordx4 %0, off offset:1024" :: "v"(sj), "s"(dj) : "memory");
            else
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(sj), "s"(dj) : "memory");
        }
        if (kDmaExtra != 0 && wid < kDmaExtra)                                    // wave-uniform
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src + (size_t)kDmaMin * 1024), "s"(dst + kDmaMin * 1024) : "memory");
        asm volatile("s_mov_b32 m0, %0" :: "s"(keep));
    };
    // wait until this wave's copies of all but the newest `ahead` tiles have landed (the first
    // kDmaExtra waves over-wait by up to `ahead` chunks: the immediate must be a constant)
    auto wait_tiles = [&](int ahead) {
        if (R >= 5 && ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kDmaMin) : "memory");
        else if (R >= 4 && ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kDmaMin) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    static_assert(R >= 3 && R <= 5, "ring depth");
    const int wave_id = blockIdx.x * NW + wid;
    const int64_t wave_tok0 = wid < p.full_waves ? (int64_t)blockIdx.x * (kTokPerWave * p.full_waves) + wid * kTokPerWave
                            : (wid == p.full_waves ? p.extra_base + (int64_t)blockIdx.x * kTokPerWave : p.n_tokens);
    const bool wave_active = wave_tok0 < p.n_tokens;                        // wave-uniform
    stamp(p, 0, lane, wave_id);
    if (p.stamps && lane == 0) p.stamps[(size_t)wave_id * 16 + 9] = __builtin_amdgcn_s_memrealtime();      // 100 MHz
    // ---- token-phase gate.  The token loads are a chip-wide HBM phase during which the matrix pipe idles, and
    // the MFMA loop afterwards leaves HBM idle.  With two workgroups per CU both phases would run in lockstep on
    // every CU; instead the a-th workgroup to arrive on a CU loads its tokens only after the a earlier ones
    // have theirs, so the second workgroup's HBM phase runs under the first one's MFMA loop (and the first
    // generation gets the whole HBM bandwidth: it starts its MFMA loop earlier).  A workgroup only ever waits
    // for workgroups that arrived before it on the same CU and those never wait for it: no deadlock.
    unsigned *gate = nullptr;
    unsigned arrival = 0;
    if (p.gate) {
        const unsigned cu = __builtin_amdgcn_s_getreg((8 - 1) << 11 | 8 << 6 | 4);      // HW_REG_HW_ID[15:8]
        const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);    // HW_REG_XCC_ID[3:0]
        gate = p.gate + 2 * ((xcc << 8) | cu);
        if (tid == 0) arrival = atomicAdd(gate, 1u);
    }
    if (!SN_S1_STAGE) {
#pragma unroll
        for (int t = 0; t < R - 1; ++t)
            if (t < n_tiles) issue_tile(t, t);
    }
    if (p.gate) {
        if (tid == 0) {
            const unsigned target = arrival * NW;
            while (__hip_atomic_load(gate + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(8);
        }
        __builtin_amdgcn_s_barrier();
        if (p.stamps && tid == 0) { p.stamps[(size_t)wave_id * 16 + 6] = arrival; p.stamps[(size_t)wave_id * 16 + 7] = (unsigned long long)(gate - p.gate) / 2; }
        stamp(p, 8, lane, wave_id);
    }

    // ---- this wave's 32 tokens: fp32 -> fp16 B fragments, kept in registers for the whole kernel
    const int64_t n = wave_tok0 + r;
    const bool valid = n < p.n_tokens;
    half8 b[NSTEPS];
    float sumsq = 0.0f, sumd = 0.0f;
    // sumd = |x - fp16(x)|^2, what the fp16 fragments lose (exact differences, fp32 sum): with the codebook's own loss
    // (scal[4]) it replaces the worst-case rounding term 2.01 u |x| |c| of the window by |x| |dc| + |dx| |c~| - 2.5 x
    // tighter on ordinary data, so 2-3 x fewer tokens go to the fp64 re-rank
    // (packed fp32 arithmetic - v_pk_fma_f32 / v_pk_add_f32, two elements per instruction - keeps the VALU count of the
    // token phase where it was before the loss was measured: sumsq and sumd are then two partial sums each)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef _Float16 half2s __attribute__((ext_vector_type(2)));
    f32x2 sq2 = {0.0f, 0.0f}, sd2 = {0.0f, 0.0f};
    auto convert = [&](int u, const float (&f)[16]) {          // 16 consecutive floats of the lane's token -> k-steps 2u, 2u+1
#pragma unroll
        for (int j = 0; j < 16; j += 2) {                      // (neighbours: the pair is a register pair as loaded, no copies)
            const f32x2 v = {f[j], f[j + 1]};
            sq2 = __builtin_elementwise_fma(v, v, sq2);
            const half2s hp = __builtin_convertvector(v, half2s);                       // v_cvt_pk_f16_f32 (round to nearest)
            b[2 * u + (j >> 3)][j & 7] = hp.x; b[2 * u + (j >> 3)][(j & 7) + 1] = hp.y;
#if SN_S1_EXACT_LOSS
            const f32x2 d = v - __builtin_convertvector(hp, f32x2);                     // from the SAME converted pair
            sd2 = __builtin_elementwise_fma(d, d, sd2);
#endif
        }
    };
#if SN_S1_STAGE
    // Loading a B fragment straight from global memory makes every wave-instruction touch 64 different cache
    // lines for 16 bytes each (lane = token row): the L1 tag rate, one line per cycle, then bounds the token
    // phase at ~11 B/cycle/CU.  Instead the rows come in whole 128-byte lines by LDS-DMA - one chunk = 32 floats
    // of each of the wave's 32 rows = four 1 KiB instructions, 8 lanes per row - into a per-wave ring of 4 KiB
    // buffers inside the (still unused) codebook ring, and are read back in fragment order.  The DMA lays lanes
    // out linearly, so lane (row, slot) fetches piece slot ^ ((row >> 1) & 7) of its row: with that swizzle the
    // four ds_read_b128 of a lane (row r, pieces 4h..4h+3) are bank-conflict-free.  Only this wave touches its
    // buffers: counted vmcnt waits, no barrier.
    constexpr int kU = NSTEPS / 2;                                            // chunks (u-steps) per token
    constexpr int kBufsFit = (R * kTileBytes / NW) / 4096;
    constexpr int kStageBufs = kBufsFit < 4 ? kBufsFit : 4;
    static_assert(kStageBufs >= 2, "token staging needs two 4 KiB buffers per wave inside the codebook ring");
    const unsigned stage_base = __builtin_amdgcn_readfirstlane(lds_base + wid * (kStageBufs * 4096));
    const unsigned char *rowq[4];                                             // (bytes: fp32 and bf16 rows alike, 128-byte chunks)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int rq = 8 * q + (lane >> 3);
        const int64_t nq = wave_tok0 + rq;
        rowq[q] = reinterpret_cast<const unsigned char *>(token_row(p, nq < p.n_tokens ? nq : 0)) + 16 * ((lane & 7) ^ ((rq >> 1) & 7));
    }
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
    if (wave_active) {
    auto issue_chunk = [&](int u) {
        const unsigned dst = stage_base + (u % kStageBufs) * 4096;
#pragma unroll
        for (int q = 0; q < 4; ++q)
#if SN_S1_TOKENS_NT
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" :: "v"(rowq[q] + 128 * u), "s"(dst + q * 1024) : "memory");
#else
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(rowq[q] + 128 * u), "s"(dst + q * 1024) : "memory");
#endif
    };
    const unsigned char *frag_src = smem + wid * (kStageBufs * 4096) + (r >> 3) * 1024 + (r & 7) * 128;
    const int sw = (r >> 1) & 7;
    // chunk c = bytes [128 c, 128 c + 128) of every row: 32 fp32 elements (k-steps 2c, 2c+1) or 64 bf16 elements
    // (k-steps 4c .. 4c+3).  A lane reads the four 16-byte pieces that hold its k: fp32 4h..4h+3; bf16 2h, 2h+1 of
    // each 64-byte half.
    auto wait_chunk = [&](int ahead) {
        if (ahead >= 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else if (ahead == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    constexpr int kAheadMax = kStageBufs - 1;
    if (!p.x_bf16) {
#pragma unroll
        for (int u = 0; u < kStageBufs && u < kU; ++u) issue_chunk(u);
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            wait_chunk((kU - 1 - u) < kAheadMax ? (kU - 1 - u) : kAheadMax);      // chunks issued after chunk u
            f32x4 raw[4];
#pragma unroll
            for (int v = 0; v < 4; ++v)
                raw[v] = *reinterpret_cast<const f32x4 *>(frag_src + (u % kStageBufs) * 4096 + (((4 * h + v) ^ sw) * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                        // in registers: the buffer may be refilled
            if (u + kStageBufs < kU) issue_chunk(u + kStageBufs);
            const float f[16] = {raw[0].x, raw[0].y, raw[0].z, raw[0].w, raw[1].x, raw[1].y, raw[1].z, raw[1].w,
                                 raw[2].x, raw[2].y, raw[2].z, raw[2].w, raw[3].x, raw[3].y, raw[3].z, raw[3].w};
            convert(u, f);
        }
    } else {
        constexpr int kUB = NSTEPS / 4;                                               // 128-byte chunks of a bf16 row
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int c = 0; c < kStageBufs && c < kUB; ++c) issue_chunk(c);
#pragma unroll
        for (int c = 0; c < kUB; ++c) {
            wait_chunk((kUB - 1 - c) < kAheadMax ? (kUB - 1 - c) : kAheadMax);
            u32x4 raw[4];                                                             // [half uu][piece e]
#pragma unroll
            for (int v = 0; v < 4; ++v)
                raw[v] = *reinterpret_cast<const u32x4 *>(frag_src + (c % kStageBufs) * 4096 + (((4 * (v >> 1) + 2 * h + (v & 1)) ^ sw) * 16));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (c + kStageBufs < kUB) issue_chunk(c + kStageBufs);
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                float f[16];
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const unsigned w2 = raw[2 * uu + e][j];                       // two bf16: element 2j (low half), 2j + 1
                        f[8 * e + 2 * j] = __uint_as_float(w2 << 16);
                        f[8 * e + 2 * j + 1] = __uint_as_float(w2 & 0xFFFF0000u);
                    }
                convert(2 * c + uu, f);
            }
        }
    }
    }
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep_m0));
    // the staging buffers are dead: the codebook ring can be primed once every wave is here
    __builtin_amdgcn_s_barrier();
#pragma unroll
    for (int t = 0; t < R - 1; ++t)
        if (t < n_tiles) issue_tile(t, t);
#else
    const float *row = token_row(p, valid ? n : 0);
    // all loads of a half-row are issued before the first conversion (the MFMA loop's registers are
    // not live yet, so up to 24 x 16 B per lane can be in flight), in two batches
    constexpr int kHalf = NSTEPS / 4;                 // u-steps per batch
#pragma unroll
    for (int batch = 0; batch < 2; ++batch) {
        f32x4 raw[kHalf][4];
#pragma unroll
        for (int uu = 0; uu < kHalf; ++uu) {
            const f32x4 *q = reinterpret_cast<const f32x4 *>(row + 32 * (batch * kHalf + uu) + 16 * h);
#pragma unroll
            for (int v = 0; v < 4; ++v) raw[uu][v] = q[v];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int uu = 0; uu < kHalf; ++uu) {
            const float f[16] = {raw[uu][0].x, raw[uu][0].y, raw[uu][0].z, raw[uu][0].w, raw[uu][1].x, raw[uu][1].y,
                                 raw[uu][1].z, raw[uu][1].w, raw[uu][2].x, raw[uu][2].y, raw[uu][2].z, raw[uu][2].w,
                                 raw[uu][3].x, raw[uu][3].y, raw[uu][3].z, raw[uu][3].w};
            convert(batch * kHalf + uu, f);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
    if (p.gate && lane == 0) atomicAdd(gate + 1, 1u);       // this wave's tokens have landed (result unused: no wait)
    sumsq = sq2.x + sq2.y;
    sumd = sd2.x + sd2.y;
    sumsq += __shfl_xor(sumsq, 32, SN_WAVE);
    sumd += __shfl_xor(sumd, 32, SN_WAVE);

    // ---- per-token error window (DESIGN.md "S1 error window"): |v_key - v_exact| <= E
    const float C2 = __uint_as_float(scal[0]), C1 = __uint_as_float(scal[1]);
    const float CN = __uint_as_float(scal[2]), CMAX = __uint_as_float(scal[3]);
    const float DC = __uint_as_float(scal[4]);                                // max |c - fp16(c)|_2
    const float X2 = sqrtf(sumsq) * 1.001f, X1 = X2 * sqrtf((float)p.D);      // (|x|_1 <= sqrt(D) |x|_2)
#if SN_S1_EXACT_LOSS
    const float DX = sqrtf(sumd) * 1.001f + 1.0e-30f;                         // |x - fp16(x)|_2 (inf / NaN tokens: `bad` below)
#else
    const float DX = kU16 * X2 + 1.0e-30f;                                    // |x_k - fp16(x_k)| <= u |x_k| (+ the subnormal term below)
#endif
    const float hx = 0.5f * sumsq;
    const float vmax = 0.5f * CN + 0.5f * X2 * X2 + X2 * C2;                 // >= any v (before the shift)
    // x.c - x~.c~ = x.(c - c~) + (x - x~).c~  (exactly), each term by Cauchy-Schwarz; |c~| <= (1 + u) |c|
    const float E = 1.01f * (1.001f * (X2 * DC + DX * C2 * 1.0005f)            // fp16 rounding of x and c
                             + 5.96e-8f * (X1 + C1)                           // fp16 subnormal flush
                             + (float)NSTEPS * kAccUlpPerMfma * vmax          // MFMA fp32 accumulate (starts at |c|^2/2 + shift)
                             + vmax * ((DUAL ? 4.0f : 3.0f) * 5.96e-8f + kKeyTrunc));         // hx/hc/adds rounding (DUAL: + the chain sum) + key truncation
    const float shift = hx + 2.0f * E;                                        // keeps every key non-negative
    const float window = 2.0f * E;
    // (|x|_2 <= kHugeIn bounds every component; inf / NaN components make the sum of squares inf / NaN)
    const bool bad = !(sumsq <= kHugeIn * kHugeIn) || !(CMAX <= kHugeIn) || !(vmax < 1.0e30f);   // NaN-safe

    stamp(p, 1, lane, wave_id);
    unsigned m1[4], m2[4], m3[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) m1[g] = m2[g] = m3[g] = 0xFFFFFFFFu;

    // ---- main loop, software pipelined inside each wave.  The A-fragment stream is continuous
    // across tiles (register ring of kRingA ds_read_b128 in flight); two accumulator sets alternate,
    // so while the MFMAs of tile w run, the wave (a) turns the accumulators of tile w-1 into keys
    // (4 VALU per value, one value per MFMA gap from step 2 on - hand-placed, see key_insert),
    // (b) at step kInitStep-1 waits for the DMA of tile w+1, passes the block barrier and issues the
    // DMA of tile w+R-1 into the slot tile w-1 occupied, (c) from step kInitStep on reads tile w+1's
    // first fragments and initialises its accumulators (|c|^2/2 + shift).  One barrier per tile.
    constexpr int kRingA = (NSTEPS % 8 == 0) ? 8 : 4;
    static_assert(NSTEPS % kRingA == 0 && NSTEPS >= 2 * kRingA, "ring phase must repeat every tile");
    constexpr int kInitStep = NSTEPS - kRingA;            // first position that touches tile w+1
    constexpr int kKeyStep0 = DUAL ? 3 : 2;               // first gap with a key (see key_insert; DUAL: behind the chain sum)
    constexpr int kKeysPerStep = (16 + (NSTEPS - kKeyStep0) - 1) / (NSTEPS - kKeyStep0);
    static_assert(kKeyStep0 + (3 + kKeysPerStep) / kKeysPerStep <= kInitStep, "group 0 must be keyed before it is re-initialised");
    static_assert(!DUAL || (NSTEPS % 2 == 0 && kInitStep >= 4), "two chains need an even number of k-steps");
    half8 ar[kRingA];
    f32x16 accA, accB, accQ;
    unsigned keymask = ~kCodeMask;
    asm volatile("" : "+v"(keymask));                     // keep the mask in a VGPR (VOP3 has no literals on gfx9)
    auto frag_at = [&](int tile, int step) {
        return *reinterpret_cast<const half8 *>(smem + (tile % R) * kTileBytes + step * 1024 + lane * 16);
    };
    auto init_group = [&](f32x16 &acc, int tile, int g) {
        const float *hc = reinterpret_cast<const float *>(smem + (tile % R) * kTileBytes + NSTEPS * 1024);
        const float4 c4 = *reinterpret_cast<const float4 *>(hc + (g * 2 + h) * 4);
        acc[4 * g + 0] = c4.x + shift; acc[4 * g + 1] = c4.y + shift;
        acc[4 * g + 2] = c4.z + shift; acc[4 * g + 3] = c4.w + shift;
    };
    // key = (value bits & ~0xFF) | code, inserted into the sorted triple (m1 <= m2 <= m3) of its
    // accumulator group: m3 = med3(k, m2, m3); m2 = med3(k, m1, m2); m1 = min(k, m1).
    // Written as volatile asm because hipcc otherwise gathers all key arithmetic of a tile pair in
    // the loop latch (the matrix pipe idles meanwhile and the accumulators get copied).  The asm
    // reads MFMA results the compiler's hazard recogniser cannot see: callers place it at least two
    // MFMA issues (> 64 cycles) after the last MFMA that wrote `v` (DUAL: `v` comes out of the chain sum, a VALU result).
    auto key_insert = [&](float v, unsigned code, int g) {
        unsigned k;
        asm volatile("v_and_or_b32 %0, %4, %5, %6\n\t"
                     "v_med3_u32 %3, %0, %2, %3\n\t"
                     "v_med3_u32 %2, %0, %1, %2\n\t"
                     "v_min_u32 %1, %0, %1"
                     : "=&v"(k), "+v"(m1[g]), "+v"(m2[g]), "+v"(m3[g]) : "v"(v), "v"(keymask), "s"(code));
    };
    auto key_value = [&](float v, int tile, int idx) {     // same thing in C++ (compiler-scheduled, hazard-checked)
        const unsigned code = (((unsigned)tile & kTileMask) << 2) | (unsigned)(idx & 3);
        const unsigned k = (__float_as_uint(v) & ~kCodeMask) | code;
        const int g = idx >> 2;
        m3[g] = med3u(k, m2[g], m3[g]);
        m2[g] = med3u(k, m1[g], m2[g]);
        m1[g] = min(k, m1[g]);
    };
    unsigned long long t_sync = 0, t_dma = 0;             // diagnostics (only when stamps are on)
    auto tile_step = [&](int w, f32x16 &cur, f32x16 &oth) {
        const unsigned code0 = (((unsigned)(w - 1)) & kTileMask) << 2;
#pragma unroll
        for (int pos = 0; pos < NSTEPS; ++pos) {
            // k-step issued at this position (DUAL: 0, 2, 1, 4, 3, ..., NSTEPS-2, NSTEPS-3, NSTEPS-1)
            const int s = !DUAL ? pos : (pos == 0 ? 0 : (pos == NSTEPS - 1 ? NSTEPS - 1 : ((pos & 1) ? pos + 1 : pos - 1)));
            if (DUAL && pos == 2) {
                // the previous tile's two chains become one value per word; accQ is free for this tile's odd chain
#pragma unroll
                for (int q = 0; q < 16; ++q) oth[q] += accQ[q];
            }
            if (!DUAL || (s & 1) == 0) cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[s % kRingA], b[s], cur, 0, 0, 0);
            else if (s == 1) {
                f32x16 zero;
#pragma unroll
                for (int q = 0; q < 16; ++q) zero[q] = 0.0f;                  // (an inline constant as the C operand: no registers)
                accQ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[s % kRingA], b[s], zero, 0, 0, 0);
            } else accQ = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[s % kRingA], b[s], accQ, 0, 0, 0);
            if (DUAL && pos == kInitStep - 2) {
                // (the position after this one prefetches tile w+1's first fragments: the ring barrier comes first)
                int ahead = n_tiles - 2 - w;
                ahead = ahead < 0 ? 0 : (ahead > R - 3 ? R - 3 : ahead);
                unsigned long long ta = 0, tb = 0;
                if (p.stamps) ta = __builtin_amdgcn_s_memtime();
                wait_tiles(ahead);
                __builtin_amdgcn_s_barrier();
                if (p.stamps) tb = __builtin_amdgcn_s_memtime();
                if (w + R - 1 < n_tiles) issue_tile(w + R - 1, (w + R - 1) % R);
                if (p.stamps) { t_sync += tb - ta; t_dma += __builtin_amdgcn_s_memtime() - tb; }
            }
            if (s + kRingA < NSTEPS) ar[s % kRingA] = frag_at(w, s + kRingA);
            else ar[s % kRingA] = frag_at(w + 1, s + kRingA - NSTEPS);      // (stale slot after the last tile: unused)
            // the 16 accumulators of tile w-1 become keys (w == 0: oth holds +inf, those keys never win)
            if (pos >= kKeyStep0) {
#pragma unroll
                for (int q = (pos - kKeyStep0) * kKeysPerStep; q < (pos - kKeyStep0 + 1) * kKeysPerStep && q < 16; ++q)
                    key_insert(oth[q], code0 | (unsigned)(q & 3), q >> 2);
            }
            if (pos >= kInitStep && pos < kInitStep + 4) init_group(oth, w + 1, pos - kInitStep);
            __builtin_amdgcn_sched_barrier(0);                              // pin: MFMA, its DS read, this gap's VALU
            if (!DUAL && pos == kInitStep - 1) {
                int ahead = n_tiles - 2 - w;                                 // tiles in flight beyond w+1
                ahead = ahead < 0 ? 0 : (ahead > R - 3 ? R - 3 : ahead);
                unsigned long long ta = 0, tb = 0;
                if (p.stamps) ta = __builtin_amdgcn_s_memtime();
                wait_tiles(ahead);                                          // this wave's part of tile w+1 has landed
                __builtin_amdgcn_s_barrier();        // ... everybody's; tile w-1 is no longer read
                if (p.stamps) tb = __builtin_amdgcn_s_memtime();
                if (w + R - 1 < n_tiles) issue_tile(w + R - 1, (w + R - 1) % R);
                if (p.stamps) { t_sync += tb - ta; t_dma += __builtin_amdgcn_s_memtime() - tb; }
            }
        }
    };
    // prologue: tile 0 in LDS, ring primed, accumulators of tile 0 initialised
    {
        int ahead = n_tiles - 1;
        ahead = ahead > R - 2 ? R - 2 : ahead;
        if (R >= 5 && ahead >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * kDmaMin) : "memory");
        else if (R >= 4 && ahead == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kDmaMin) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kDmaMin) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    if (!wave_active) {                   // a wave without tokens: its share of the ring copies and the barriers, nothing else
        for (int w = 0; w < n_tiles; ++w) {
            int ahead = n_tiles - 2 - w;
            ahead = ahead < 0 ? 0 : (ahead > R - 3 ? R - 3 : ahead);
            wait_tiles(ahead);
            __builtin_amdgcn_s_barrier();
            if (w + R - 1 < n_tiles) issue_tile(w + R - 1, (w + R - 1) % R);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
#pragma unroll
    for (int q = 0; q < kRingA; ++q) ar[q] = frag_at(0, q);
#pragma unroll
    for (int g = 0; g < 4; ++g) init_group(accA, 0, g);
#pragma unroll
    for (int q = 0; q < 16; ++q) { accB[q] = INFINITY; accQ[q] = 0.0f; }
    for (int w = 0; w < n_tiles; w += 2) {                    // n_tiles is even (pack_layout)
        tile_step(w, accA, accB);
        tile_step(w + 1, accB, accA);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // no DMA may be in flight when the LDS is released
#pragma unroll
    for (int q = 0; q < 16; ++q) key_value(DUAL ? accB[q] + accQ[q] : accB[q], n_tiles - 1, q);      // keys of the last tile

    stamp(p, 2, lane, wave_id);
    if (p.stamps && lane == 0) { p.stamps[(size_t)wave_id * 16 + 4] = t_sync; p.stamps[(size_t)wave_id * 16 + 5] = t_dma; }
    // ---- candidates: every key within the window of the token's best, over both half-lanes
    unsigned kmin = min(min(m1[0], m1[1]), min(m1[2], m1[3]));
    const unsigned kmin_o = __shfl_xor(kmin, 32, SN_WAVE);
    // word of a key: tile = code >> 2, e = code & 3, row = 8g + 4h + e
    auto word_of = [&](unsigned k, int g, int hh) { return (int)(((k & kCodeMask) >> 2) * 32 + 8 * g + 4 * hh + (k & 3u)); };
    int gmin = 0;
#pragma unroll
    for (int g = 1; g < 4; ++g) if (m1[g] == kmin) gmin = g;
#pragma unroll
    for (int g = 3; g >= 0; --g) if (m1[g] == kmin) gmin = g;      // lowest group on ties
    const int my_best = word_of(kmin, gmin, h);
    const int ot_best = __shfl_xor(my_best, 32, SN_WAVE);
    const unsigned vmy = kmin & ~kCodeMask, vot = kmin_o & ~kCodeMask;
    const bool mine = vmy < vot || (vmy == vot && my_best < ot_best);
    const int best_w = mine ? my_best : ot_best;
    const float vbest = __uint_as_float(mine ? vmy : vot);
    const bool any_finite = (mine ? vmy : vot) < 0x7F800000u;
    const unsigned cutkey = __float_as_uint(vbest + window) | kCodeMask;   // key <= cutkey  <=>  value <= cut
    unsigned hmask = 0;                                                // 12 bits: group g -> bits 3g..3g+2
    bool hover = false;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        if (m1[g] <= cutkey) hmask |= 1u << (3 * g);
        if (m2[g] <= cutkey) hmask |= 2u << (3 * g);
        if (m3[g] <= cutkey) { hmask |= 4u << (3 * g); hover = true; }   // a 4th may hide behind it
    }
    const unsigned omask = __shfl_xor(hmask, 32, SN_WAVE);
    const bool oover = __shfl_xor((int)hover, 32, SN_WAVE) != 0;
    const int nc = __popc(hmask) + __popc(omask);
    const bool overflow = bad || !any_finite || hover || oover;
    const bool writer = valid && h == 0;
    if (writer) p.out[out_index(p, n)] = any_finite ? best_w : 0;
    // No compaction here: a chip-wide atomic per wave on one counter costs more than the whole
    // MFMA loop when every wave has a flagged token.  Each token gets a flag word (dense, coalesced)
    // and, if flagged, its 24 candidate codes at a fixed slot; the re-rank kernel walks the flags.
    const bool flagged = !overflow && nc > 1;
    // (overflow with a bounded window: the candidate mask and the codes are written as well - a consumer may then restrict
    // itself to the candidates and the 64 words of every group whose triple is inside the window whole, sn_assign_shared.h;
    // the stand-alone overflow kernel scans every word)
    const bool fullscan = bad || !any_finite;
    if (writer) p.flags[n] = fullscan ? sn_s1::kFlagFullScan : ((overflow ? sn_s1::kFlagOverflow : 0u) | ((flagged || overflow) ? (hmask | (omask << 12)) : 0u));
    if (valid && (flagged || (overflow && !fullscan))) {   // both half-lanes of the token write their 12 codes
        const unsigned c0 = m1[0] & kCodeMask, c1 = m2[0] & kCodeMask, c2 = m3[0] & kCodeMask, c3 = m1[1] & kCodeMask;
        const unsigned c4 = m2[1] & kCodeMask, c5 = m3[1] & kCodeMask, c6 = m1[2] & kCodeMask, c7 = m2[2] & kCodeMask;
        const unsigned c8 = m3[2] & kCodeMask, c9 = m1[3] & kCodeMask, c10 = m2[3] & kCodeMask, c11 = m3[3] & kCodeMask;
        if (CB == 8) {
            unsigned *cd = reinterpret_cast<unsigned *>(p.codes + (int64_t)n * kCodeBytes + 12 * h);
            cd[0] = c0 | (c1 << 8) | (c2 << 16) | (c3 << 24);
            cd[1] = c4 | (c5 << 8) | (c6 << 16) | (c7 << 24);
            cd[2] = c8 | (c9 << 8) | (c10 << 16) | (c11 << 24);
        } else {
            unsigned *cd = reinterpret_cast<unsigned *>(p.codes + (int64_t)n * kCodeBytesWide + 24 * h);
            cd[0] = c0 | (c1 << 16); cd[1] = c2 | (c3 << 16); cd[2] = c4 | (c5 << 16);
            cd[3] = c6 | (c7 << 16); cd[4] = c8 | (c9 << 16); cd[5] = c10 | (c11 << 16);
        }
    }
    const bool need_b = writer && overflow;
    const unsigned long long mask_b = __ballot(need_b);
    stamp(p, 3, lane, wave_id);
    if (p.stamps && lane == 0) p.stamps[(size_t)wave_id * 16 + 10] = __builtin_amdgcn_s_memrealtime();
    if (mask_b) {                                          // rare: tokens the screen cannot bound
        int base = 0;
        const int leader = __ffsll((long long)mask_b) - 1;
        if (lane == leader) base = atomicAdd(&p.work[1], __popcll(mask_b));
        base = __shfl(base, leader, SN_WAVE);
        if (need_b) p.overflow[base + __popcll(mask_b & ((1ull << lane) - 1ull))] = (int)n;
    }
}

// ------------------------------------------------------------------------------------------
// mode 0, pass 1, register-stationary form (M <= 512 at D = 384 / 192: the whole fp16 codebook fits the
// register file of one CU).
//
// One persistent workgroup (4 waves, one per SIMD, 512 registers each) per CU.  Wave q keeps the A
// fragments of its quarter of the codebook (NT tiles of 32 words x D: NT KS x 4 registers, 256 AGPRs +
// the rest in VGPRs) for the whole kernel; token sets of 32 stream through:
//   HBM --LDS-DMA (coalesced 128-B lines, piece-swizzled, two half sets of 16)--> raw fp32 slots
//       --each wave converts a quarter of the 32-float chunks--> fp16 B fragments in LDS (+ |x|^2 by a Gram MFMA)
//       --every wave: v_mfma_f32_32x32x16_f16 against its own words, one tile at a time--> keys (sorted triple per lane)
//       --LDS min over the 8 lanes that hold a token--> window test, flag word, candidate codes.
// The codebook is read from L2 once per CU and the tokens from HBM exactly once, in flight while
// earlier sets are on the matrix pipe.  Values: u[word] = |c|^2/2 - x~.c~ (tokens and words rounded to
// fp16, fp32 accumulate), keys = float bits with the low 8 mantissa bits replaced by (tile << 4 | reg),
// compared as floats.  A lone wave per SIMD issues one instruction per ~4 cycles, so everything that
// is not an MFMA is counted: the key arithmetic rides in the same asm statement as the MFMA it hides
// behind, the other stages are dealt over the remaining gaps.
//
// Software pipeline, one barrier per iteration `it` (set indices local to the workgroup):
//   DMA(it+2)  CVT(it+1)  MMA(it) [+ keys of the previous tile]  WIN(it)  CMP(it-2)  FLG(it-3)
// ------------------------------------------------------------------------------------------
constexpr int kS2RawSlots = 4;          // raw fp32 half sets (16 tokens) in LDS: set it+1 being converted, set it+2 in flight
constexpr int kS2SmallSlots = 8;        // ring of per-set scalars
constexpr int kS2StashSlots = 2;
constexpr float kBigKey = 3.0e38f;      // "no key yet"
constexpr float kKeyLimit = 1.0e29f;    // above this a best value is a padding word / nothing finite

struct S2Tok {                          // per token of a set in flight (LDS)
    float best;                         // smallest key of the token (ds_min over the 8 lanes that hold it)
    float win;                          // 2E (NaN: the token cannot be screened)
    unsigned mask;                      // candidate mask being assembled: bit 3c + j = key j of slot c = 2 wave + half (bit 31: overflow)
    float nrm;                          // |x~|^2, summed over the four waves' chunks
};
struct S2Small { S2Tok tok[32]; };

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) as straight-line code
// (the MFMA stream below is far beyond the size a `#pragma unroll` is allowed to expand)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// One MFMA step and up to two key halves in ONE asm statement (nothing can be scheduled, and no s_nop
// inserted, between them).  Key insert of value v with code c into the sorted triple m1 <= m2 <= m3:
//   half A: k = (v & ~0xFF) | c;  m3 = med3(k, m2, m3)      half B: m2 = med3(k, m1, m2);  m1 = min(k, m1)
// PAT 0: MFMA   1: MFMA, A   2: MFMA, B   3: MFMA, A, B   4: MFMA, B, A (B finishes the previous value)
#define S2_MF "v_mfma_f32_32x32x16_f16 %[acc], %[a], %[b], %[acc]\n\t"
#define S2_KA "v_and_or_b32 %[kk], %[v], %[km], %[code]\n\tv_med3_f32 %[m3], %[kk], %[m2], %[m3]\n\t"
#define S2_KB "v_med3_f32 %[m2], %[kk], %[m1], %[m2]\n\tv_min_f32 %[m1], %[kk], %[m1]\n\t"
#define S2_OUT [acc] "+v"(acc), [kk] "+v"(kk), [m1] "+v"(m1), [m2] "+v"(m2), [m3] "+v"(m3)
#define S2_IN(ACLS) [a] ACLS(a), [b] "v"(b), [v] "v"(v), [km] "v"(keymask), [code] "i"(CODE)
template <bool AG, int PAT, int CODE>
__device__ __forceinline__ void s2_step(f32x16 &acc, const half8 &a, const half8 &b, float &kk, float &m1, float &m2, float &m3,
                                        float v, unsigned keymask)
{
    if constexpr (AG) {
        if constexpr (PAT == 0) asm volatile(S2_MF : [acc] "+v"(acc) : [a] "a"(a), [b] "v"(b));
        if constexpr (PAT == 1) asm volatile(S2_MF S2_KA : S2_OUT : S2_IN("a"));
        if constexpr (PAT == 2) asm volatile(S2_MF S2_KB : S2_OUT : S2_IN("a"));
        if constexpr (PAT == 3) asm volatile(S2_MF S2_KA S2_KB : S2_OUT : S2_IN("a"));
        if constexpr (PAT == 4) asm volatile(S2_MF S2_KB S2_KA : S2_OUT : S2_IN("a"));
    } else {
        if constexpr (PAT == 0) asm volatile(S2_MF : [acc] "+v"(acc) : [a] "v"(a), [b] "v"(b));
        if constexpr (PAT == 1) asm volatile(S2_MF S2_KA : S2_OUT : S2_IN("v"));
        if constexpr (PAT == 2) asm volatile(S2_MF S2_KB : S2_OUT : S2_IN("v"));
        if constexpr (PAT == 3) asm volatile(S2_MF S2_KA S2_KB : S2_OUT : S2_IN("v"));
        if constexpr (PAT == 4) asm volatile(S2_MF S2_KB S2_KA : S2_OUT : S2_IN("v"));
    }
}
template <int CODE>
__device__ __forceinline__ void s2_key_a(float &kk, float &m2, float &m3, float v, unsigned keymask)
{
    asm volatile(S2_KA : [kk] "+v"(kk), [m2] "+v"(m2), [m3] "+v"(m3) : [v] "v"(v), [km] "v"(keymask), [code] "i"(CODE));
}
__device__ __forceinline__ void s2_key_b(float &kk, float &m1, float &m2)
{
    asm volatile(S2_KB : [kk] "+v"(kk), [m1] "+v"(m1), [m2] "+v"(m2));
}
#undef S2_MF
#undef S2_KA
#undef S2_KB
#undef S2_OUT
#undef S2_IN

template <int NT, int KS>
__global__ __launch_bounds__(256, 1) void assign_screen2_kernel(const AssignArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int KC = KS / 2;                                  // 32-float chunks per token
    constexpr int NA = NT * KS < 64 ? NT * KS : 64;             // fragments pinned in AGPRs
    constexpr int kHalfRaw = KC * 2048, kSetFrag = KS * 1024;   // bytes: raw half set (16 tokens), fp16 fragments of a set
    constexpr int kDmaPerWave = KC / 2;                         // 1 KiB LDS-DMA instructions per wave and half set
    constexpr int kCvtPerHalf = (KC + 3) / 4;                   // chunks a wave converts per half set
    constexpr int kSteps = NT * KS;                             // MFMA steps per set: (tile, k-step)
    constexpr int kKeyStart = 2, kKeyEnd = KS - 5;              // k-steps of a tile phase that carry key halves of the previous tile
    constexpr int kKeySteps = kKeyEnd - kKeyStart + 1;
    static_assert(NT == 2 || NT == 4, "tiles per wave");
    static_assert(KS == 12 || KS == 24, "k-steps");
    unsigned char *raw = smem;
    unsigned char *frag = smem + kS2RawSlots * kHalfRaw;
    S2Small *small = reinterpret_cast<S2Small *>(frag + 2 * kSetFrag);
    f32x4 *stash = reinterpret_cast<f32x4 *>(reinterpret_cast<unsigned char *>(small) + kS2SmallSlots * sizeof(S2Small));
    f32x4 *hnl = stash + kS2StashSlots * 256;                     // [4 waves][NT tiles][2 halves][4] x 4 half norms

    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tok = lane & 31, hh = lane >> 5;                   // MFMA side: token column, accumulator row half
    const int tau = lane & 15, g = lane >> 4;                    // conversion side: token of the half set, piece pair
    const PackLayout lay = pack_layout(p.M, p.D);
    const unsigned char *frag2 = p.packed + lay.frag2_off;
    const float *hn2 = reinterpret_cast<const float *>(p.packed + lay.hn2_off);
    const unsigned *scal = reinterpret_cast<const unsigned *>(p.packed + lay.scal_off);
    const int wave_id = blockIdx.x * 4 + wid;
    stamp(p, 0, lane, wave_id);

    // this workgroup's sets: [sb, sb + ns)
    const int64_t sb = (int64_t)blockIdx.x * p.n_sets / gridDim.x;
    const int ns = (int)((int64_t)(blockIdx.x + 1) * p.n_sets / gridDim.x - sb);
    const int n_inner32 = (int)p.n_inner;

    // ---- per-set scalars
    for (int i = tid; i < kS2SmallSlots * 32; i += 256) {
        S2Tok &t0 = small[i >> 5].tok[i & 31];
        t0.best = kBigKey; t0.win = 0.0f; t0.mask = 0u; t0.nrm = 0.0f;
    }

    // ---- token half sets: HBM -> LDS by LDS-DMA.  Instruction x = 2 c + h (32-float chunk c, token half h)
    // copies the 128-B lines of 8 tokens; wave w issues x = w, w + 4, ...: always the same token half, so a
    // lane needs one row pointer per half set.  Lane l: token 8 h + l / 8, LDS piece slot l % 8 holds
    // piece slot ^ ((token >> 1) & 7) of the line (conflict-free ds_read_b128 of the fragments below).
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    int d_o, d_i;                                               // (outer, inner) of the first token of the next half set to copy
    {
        const int64_t n0 = sb * 32;
        d_o = (int)(n0 / p.n_inner); d_i = (int)(n0 % p.n_inner);
    }
    int c_o = d_o, c_i = d_i;                                   // ... of the next set to compare (CMP)
    const int last_o = (int)((p.n_tokens - 1) / p.n_inner), last_i = (int)((p.n_tokens - 1) % p.n_inner);
    auto advance = [&](int &o, int &i, int by) {                 // (n_inner >= 32: at most one wrap)
        i += by;
        if (i >= n_inner32) { i -= n_inner32; ++o; }
    };
    auto issue_half = [&](int hs) {                              // hs = 2 set + half, local to the workgroup
        const int d = 8 * (wid & 1) + (lane >> 3);
        int o = d_o, i = d_i + d;
        if (i >= n_inner32) { i -= n_inner32; ++o; }
        const int64_t n = sb * 32 + (int64_t)hs * 16 + d;
        if (n >= p.n_tokens) { o = last_o; i = last_i; }         // (tail of the last set: any valid row)
        const int piece = (lane & 7) ^ ((d >> 1) & 7);
        const float *src = p.x + (int64_t)o * p.xso + (int64_t)i * p.xsi + (wid >> 1) * 32 + piece * 4;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (hs & (kS2RawSlots - 1)) * kHalfRaw + (wid >> 1) * 2048 + (wid & 1) * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
#pragma unroll
        for (int x = 0; x < kDmaPerWave; ++x)                    // (the instruction offset moves the LDS address too)
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off offset:%2"
                         :: "v"(src), "s"(dst + x * 4096 - x * 256), "i"(x * 256) : "memory");
        asm volatile("s_mov_b32 m0, %0" :: "s"(keep));
        advance(d_o, d_i, 16);
    };
    if (tid < 4 * NT * 2 * 4) hnl[tid] = *reinterpret_cast<const f32x4 *>(hn2 + tid * 4);    // |c|^2/2: LDS, re-read per accumulator chain
    float C2 = __uint_as_float(scal[0]), C1 = __uint_as_float(scal[1]);
    float CN = __uint_as_float(scal[2]), CMAX = __uint_as_float(scal[3]);
    asm volatile("" : "+v"(C2), "+v"(C1), "+v"(CN), "+v"(CMAX));       // every ordinary load is consumed here, before ...
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // ... anything below is in flight (hipcc would drain it with vmcnt(0) at the first use)
    for (int hs = 0; hs < 4 && hs < 2 * ns; ++hs) issue_half(hs);

    // ---- this wave's quarter of the codebook -> registers (stays there).  The NT KS loads are issued here,
    // tile by tile, and NOT waited for: 400 KB per CU out of L2 takes ~15 k cycles; the first set starts
    // on tile 0 as soon as its fragments are in and waits per tile (counted vmcnt, first iteration below).
    half8 A[NT][KS];
#pragma unroll
    for (int a = 0; a < NT; ++a) {
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const unsigned char *src = frag2 + ((size_t)(wid * NT + a) * KS + j) * 1024 + lane * 16;
            if (a * KS + j < NA) asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(A[a][j]) : "v"(src) : "memory");
            else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(A[a][j]) : "v"(src) : "memory");
        }
    }
    constexpr int kLoads = NT * KS;
    // loads of tile a' <= a have landed when at most kLoads - KS (a + 1) operations are outstanding (younger
    // operations only make this stricter); the counter saturates at 63
    constexpr auto a_wait = [](int a) constexpr { return kLoads - KS * (a + 1) < 63 ? kLoads - KS * (a + 1) : 63; };
    const f32x4 *hn_w = hnl + (wid * NT * 2 + hh) * 4;           // tile a: hn_w[8 a + i], i = 0..3: accumulator registers 4 i .. 4 i + 3

    // ---- CVT: raw half set -> fp16 B fragments of chunks wid, wid + 4, ... (+ Gram diagonal = |x~|^2).
    // Lane (tau, g) converts the floats {4g..4g+3, 16+4g..16+4g+3} of the chunk: k-step 2 c + (g >> 1), row half g & 1.
    f32x4 c_lo, c_hi;
    f32x4 nacc = {0.0f, 0.0f, 0.0f, 0.0f};
    const int swz = (tau >> 1) & 7;
    const int rd_lo = tau * 128 + ((g ^ swz) << 4), rd_hi = tau * 128 + (((g + 4) ^ swz) << 4);
    const int wr_off = (g >> 1) * 1024 + (tau + 32 * (g & 1)) * 16;
    auto cvt_read = [&](int hs, int c) {
        const int ch = wid + 4 * c;
        if (KC % 4 == 0 || ch < KC) {
            const unsigned char *base = raw + (hs & (kS2RawSlots - 1)) * kHalfRaw + ch * 2048;
            c_lo = *reinterpret_cast<const f32x4 *>(base + rd_lo);
            c_hi = *reinterpret_cast<const f32x4 *>(base + rd_hi);
        }
    };
    auto cvt_write = [&](int hs, int c) {
        const int ch = wid + 4 * c;
        if (KC % 4 == 0 || ch < KC) {
            half8 hb;
            hb[0] = (_Float16)c_lo.x; hb[1] = (_Float16)c_lo.y; hb[2] = (_Float16)c_lo.z; hb[3] = (_Float16)c_lo.w;
            hb[4] = (_Float16)c_hi.x; hb[5] = (_Float16)c_hi.y; hb[6] = (_Float16)c_hi.z; hb[7] = (_Float16)c_hi.w;
            *reinterpret_cast<half8 *>(frag + ((hs >> 1) & 1) * kSetFrag + ch * 2048 + (hs & 1) * 256 + wr_off) = hb;
            asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %1, %0" : "+v"(nacc) : "v"(hb));       // Gram matrix of the 16 tokens
        }
    };
    auto cvt_norm = [&](int hs) {                            // diagonal (tau, tau): lane group tau >> 2, register tau & 3
        asm volatile("s_nop 7\n\ts_nop 7" : "+v"(nacc));                // MFMA -> VALU read distance (the asm MFMA is invisible to the hazard recogniser)
        const float dg = (tau & 2) ? ((tau & 1) ? nacc[3] : nacc[2]) : ((tau & 1) ? nacc[1] : nacc[0]);
        if (g == (tau >> 2))
            __builtin_amdgcn_ds_faddf((__attribute__((address_space(3))) float *)&small[(hs >> 1) & (kS2SmallSlots - 1)].tok[16 * (hs & 1) + tau].nrm, dg, 0, 0, false);
        nacc = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    };

    // ---- WIN(s): error window of each token of the set (DESIGN.md "S1 error window"); lanes 0..31 of one wave
    float w_nr = 0.0f;
    auto win_read = [&](int s) { w_nr = small[s & (kS2SmallSlots - 1)].tok[tok].nrm; };
    auto win_do = [&](int s) {
        const float X2 = sqrtf(w_nr) * 1.002f + 1.0e-6f;              // |x|_2 from the fp16-rounded token
        const float X1 = X2 * sqrtf((float)p.D);
        const float vmax = 0.5f * CN + X2 * C2;                        // >= |any partial sum|
        const float E = 1.01f * (2.01f * kU16 * X2 * C2 + 5.96e-8f * (X1 + C1)
                                 + (float)KS * kAccUlpPerMfma * vmax + vmax * (3.0f * 5.96e-8f + 3.1e-5f));
        const bool ok = (w_nr < 1.0e30f) && (CMAX <= kHugeIn) && (vmax < 1.0e28f);       // false for NaN / inf
        if (lane < 32) small[s & (kS2SmallSlots - 1)].tok[tok].win = ok ? 2.0f * E : __builtin_nanf("");
    };

    // ---- CMP(s): which of this lane's three keys are inside the window of the token's best
    f32x4 c_t3;
    float c_best = 0.0f, c_win = 0.0f;
    auto cmp_read = [&](int s) {
        c_t3 = stash[(s & (kS2StashSlots - 1)) * 256 + tid];
        const S2Tok &tk = small[s & (kS2SmallSlots - 1)].tok[tok];
        c_best = tk.best; c_win = tk.win;
    };
    auto cmp_do = [&](int s, bool en) {
        const float cut = c_best + c_win;                              // NaN window -> no hit -> overflow
        const int64_t n = (sb + s) * 32 + tok;
        const bool live = en && n < p.n_tokens && cut < kKeyLimit;
        const bool h1 = live && c_t3.x <= cut, h2 = live && c_t3.y <= cut, h3 = live && c_t3.z <= cut;
        const unsigned k1 = __float_as_uint(c_t3.x), k2 = __float_as_uint(c_t3.y), k3 = __float_as_uint(c_t3.z);
        int o = c_o, i = c_i + tok;
        if (i >= n_inner32) { i -= n_inner32; ++o; }                   // (n_inner >= 32: at most one wrap)
        if (h1) {
            const int reg = (int)(k1 & 15u);                           // word row inside its 32 x 32 tile: (reg & 3) + 8 (reg >> 2) + 4 hh
            p.out[(int64_t)o * p.oso + (int64_t)i * p.osi] = wid * (32 * NT) + (int)((k1 & 0xFFu) >> 4) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * hh;
            const int c = wid * 2 + hh;
            const unsigned bits = ((1u | (h2 ? 2u : 0u) | (h3 ? 4u : 0u)) << (3 * c)) | (h3 ? 0x80000000u : 0u);   // h3: a fourth key may hide behind the third
            __hip_atomic_fetch_or(&small[s & (kS2SmallSlots - 1)].tok[tok].mask, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            p.codes32[n * 8 + c] = (k1 & 0xFFu) | ((k2 & 0xFFu) << 8) | ((k3 & 0xFFu) << 16);
        }
        if (en) advance(c_o, c_i, 32);
    };

    // ---- FLG(s): flag word of each token, overflow list, recycle the scalar slot; lanes 0..31 of one wave
    unsigned f_mk = 0u;
    auto flag_read = [&](int s) { f_mk = small[s & (kS2SmallSlots - 1)].tok[tok].mask; };
    auto flag_do = [&](int s, bool en) {
        if (lane < 32) {
            S2Tok &tk = small[s & (kS2SmallSlots - 1)].tok[lane];
            const int64_t n = (sb + s) * 32 + lane;
            const unsigned cand = f_mk & 0xFFFFFFu;
            const bool over = (f_mk >> 31) != 0u || cand == 0u;
            const bool valid = en && n < p.n_tokens;
            if (valid) p.flags[n] = over ? 0x80000000u : (__popc(cand) > 1 ? cand : 0u);
            const bool need_b = valid && over;
            const unsigned long long mask_b = __ballot(need_b);
            if (mask_b) {
                int base = 0;
                const int leader = __ffsll((long long)mask_b) - 1;
                if (lane == leader) base = atomicAdd(&p.work[1], __popcll(mask_b));
                base = __shfl(base, leader, SN_WAVE);
                if (need_b) p.overflow[base + __popcll(mask_b & ((1ull << lane) - 1ull))] = (int)n;   // phase B writes out[]
            }
            tk.best = kBigKey; tk.mask = 0u; tk.nrm = 0.0f;
        }
    };

    // ---- prologue: sets 0, 1 landed (their DMA is older than the fragment loads), set 0 converted
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kLoads < 63 ? kLoads : 63) : "memory");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if (ns > 0) {
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
            for (int c = 0; c < kCvtPerHalf; ++c) { cvt_read(h2, c); cvt_write(h2, c); }
            cvt_norm(h2);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(p, 1, lane, wave_id);

    // accumulators: tile a uses buffer a & 1; a buffer is keyed while the next tile runs, then re-initialised
    // to |c|^2/2 of the tile after that (4 ds_read_b128 straight into the accumulator registers)
    f32x16 acc0, acc1;
    auto load_hn = [&](f32x16 &acc, int a) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 v4 = hn_w[8 * a + i];
            acc[4 * i + 0] = v4.x; acc[4 * i + 1] = v4.y; acc[4 * i + 2] = v4.z; acc[4 * i + 3] = v4.w;
        }
    };
    load_hn(acc0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = kBigKey;
    float m1 = kBigKey, m2 = kBigKey, m3 = kBigKey, kk = kBigKey;
    unsigned keymask = 0xFFFFFF00u;
    asm volatile("" : "+v"(keymask));
    auto publish = [&](int s) {                                 // triple of set s is complete (s == -1: all keys are kBigKey, a no-op)
        __builtin_amdgcn_ds_fminf((__attribute__((address_space(3))) float *)&small[s & (kS2SmallSlots - 1)].tok[tok].best, m1, 0, 0, false);
        stash[(s & (kS2StashSlots - 1)) * 256 + tid] = f32x4{m1, m2, m3, 0.0f};
        m1 = m2 = m3 = kBigKey;
    };
    half8 bq[4];
    unsigned long long t_vm = 0, t_bar = 0, t_top = 0;           // diagnostics (only when stamps are on)
    auto end_of_iteration = [&]() {
        unsigned long long e0 = 0, e1 = 0;
        if (p.stamps) e0 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // set it+2 has landed (its DMA was issued at the start of this iteration)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (p.stamps) e1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (p.stamps) { t_vm += e1 - e0; t_bar += __builtin_amdgcn_s_memtime() - e1; }
    };
    // step of the stream at which a stage runs (positions are given for 96 steps and scaled)
    constexpr auto at = [](int x) constexpr { return x * kSteps / 96; };

    // ---- main loop.  Everything except the DMA issue runs unconditionally (no per-step branches):
    // in the first iterations the "previous" keys are kBigKey, CMP / FLG are predicated off by `en`,
    // in the last one CVT converts stale slots nobody reads.
    auto body = [&](auto first_c, const int it) {
        constexpr bool FIRST = decltype(first_c)::value;        // iteration 0: the fragment loads are still in flight
        const bool do_dma = it + 2 < ns;
        const unsigned char *fb = frag + (it & 1) * kSetFrag + lane * 16;
        bq[0] = *reinterpret_cast<const half8 *>(fb);
        bq[1] = *reinterpret_cast<const half8 *>(fb + 1024);
        bq[2] = *reinterpret_cast<const half8 *>(fb + 2048);
        cmp_read(it - 2);
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long tt0 = 0;
        if (p.stamps) tt0 = __builtin_amdgcn_s_memtime();
        static_for<kSteps>([&](auto st_c) {
            constexpr int st = decltype(st_c)::value;
            constexpr int a = st / KS, j = st % KS;
            constexpr int pa = (a + NT - 1) % NT;                    // the tile whose accumulators are being keyed
            if (FIRST && j == 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(a_wait(a)) : "memory");     // this tile's fragments are in
            if (st == KS) publish(it - 1);                           // (keys of set it-1 ended with the previous tile phase)
            // key halves of this step: half index hi = 2 * value + (0: A, 1: B), hi -> k-step kKeyStart + hi * kKeySteps / 32
            constexpr int lo = j < kKeyStart || j > kKeyEnd ? 0 : ((j - kKeyStart) * 32 + kKeySteps - 1) / kKeySteps;
            constexpr int hi_end = j < kKeyStart || j > kKeyEnd ? 0 : ((j + 1 - kKeyStart) * 32 + kKeySteps - 1) / kKeySteps;
            constexpr int n = hi_end - lo;
            constexpr int PAT = n == 0 ? 0 : (n == 1 ? ((lo & 1) ? 2 : 1) : ((lo & 1) ? 4 : 3));
            constexpr int va = (lo + 1) / 2;                         // the value whose half A rides with this MFMA (if any)
            constexpr int CODE = (pa << 4) | (va & 15);
            f32x16 &acc = (a & 1) ? acc1 : acc0;
            f32x16 &accp = (a & 1) ? acc0 : acc1;
            const float v = accp[va & 15];
            if (a * KS + j < NA) s2_step<true, PAT, CODE>(acc, A[a][j], bq[st & 3], kk, m1, m2, m3, v, keymask);
            else s2_step<false, PAT, CODE>(acc, A[a][j], bq[st & 3], kk, m1, m2, m3, v, keymask);
            if (st + 3 < kSteps) bq[(st + 3) & 3] = *reinterpret_cast<const half8 *>(fb + ((st + 3) % KS) * 1024);
            // further halves of this step (only the D = 192 shapes have more than two per step)
            static_for<(n > 2 ? n - 2 : 0)>([&](auto x_c) {
                constexpr int hx = lo + 2 + decltype(x_c)::value;
                if constexpr (hx & 1) s2_key_b(kk, m1, m2);
                else s2_key_a<(pa << 4) | ((hx / 2) & 15)>(kk, m2, m3, accp[(hx / 2) & 15], keymask);
            });
            // the buffer just keyed starts its next chain (tile a + 1) at |c|^2/2
            if (j == kKeyEnd + 1) load_hn(accp, (a + 1) % NT);
            // the other stages, dealt over the steps
            if (st == at(3)) cmp_do(it - 2, it >= 2);
            if (st == at(6) && wid == (it & 3)) win_read(it);
            if (st == at(9) && wid == (it & 3)) win_do(it);
            if (st == at(6) && wid == ((it + 2) & 3)) flag_read(it - 3);
            if (st == at(9) && wid == ((it + 2) & 3)) flag_do(it - 3, it >= 3);
            // (first iteration: behind the last fragment wait, so that the counted waits see only fragment loads)
            if (st == (FIRST ? (NT - 1) * KS + 1 : at(12)) && do_dma) issue_half(2 * it + 4);
            if (st == (FIRST ? (NT - 1) * KS + 3 : at(15)) && do_dma) issue_half(2 * it + 5);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
#pragma unroll
                for (int c = 0; c < kCvtPerHalf; ++c) {
                    if (st == at(18) + (h2 * kCvtPerHalf + c) * (at(72) / (2 * kCvtPerHalf))) cvt_read(2 * it + 2 + h2, c);
                    if (st == at(18) + (h2 * kCvtPerHalf + c) * (at(72) / (2 * kCvtPerHalf)) + 3) cvt_write(2 * it + 2 + h2, c);
                }
                if (st == at(18) + ((h2 + 1) * kCvtPerHalf) * (at(72) / (2 * kCvtPerHalf))) cvt_norm(2 * it + 2 + h2);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (p.stamps) t_top += __builtin_amdgcn_s_memtime() - tt0;
        end_of_iteration();
    };
    if (ns > 0) body(std::true_type{}, 0);
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int it = 1; it < ns; ++it) body(std::false_type{}, it);
    // ---- drain: keys of the very last tile, then the compare / flag stages of the last sets
    for (int it = ns; it < ns + 3 && ns > 0; ++it) {
        if (it == ns) {
            f32x16 &accl = ((NT - 1) & 1) ? acc1 : acc0;
            asm volatile("s_nop 7\n\ts_nop 7" : "+v"(accl));
            static_for<16>([&](auto r_c) {
                constexpr int r = decltype(r_c)::value;
                s2_key_a<((NT - 1) << 4) | r>(kk, m2, m3, accl[r], keymask);
                s2_key_b(kk, m1, m2);
            });
            publish(ns - 1);
        }
        if (it - 2 >= 0 && it - 2 < ns) { cmp_read(it - 2); cmp_do(it - 2, true); }
        if (it - 3 >= 0 && it - 3 < ns && wid == ((it + 2) & 3)) { flag_read(it - 3); flag_do(it - 3, true); }
        end_of_iteration();
    }
    stamp(p, 2, lane, wave_id);
    if (p.stamps && lane == 0) {
        p.stamps[(size_t)wave_id * 16 + 4] = t_top; p.stamps[(size_t)wave_id * 16 + 5] = t_vm; p.stamps[(size_t)wave_id * 16 + 6] = t_bar;
        p.stamps[(size_t)wave_id * 16 + 7] = (unsigned long long)ns;
    }
}

// ------------------------------------------------------------------------------------------
// mode 0, pass 1, K-outer form (opt-in, sn_assign_set_variant(3); codebooks of 8 or 16 tiles of 32 words: 192 < M <= 256,
// 448 < M <= 512): the token stream and the matrix pipe overlap by construction.
//
// A workgroup = 16 waves = 4 token sets x 4 word quarters, one workgroup per CU, persistent over rounds (sets dealt
// round-robin over the workgroups, so a last, partial round is spread over all CUs).  Wave
// (set ps, quarter q) keeps the accumulators of its set (<= 32 tokens) against its quarter of the codebook - NTW
// tiles of 32 words, 16 registers each - for a whole round, and K is the OUTER loop: a 32-float chunk of every
// token row arrives from HBM by LDS-DMA (whole 128-byte lines, piece-swizzled: the token staging of
// assign_screen_kernel) into a 3-deep ring per set, is converted to two fp16 B fragments by the four waves of the
// set, and is multiplied against the two k-steps of the codebook image that belong to it: 2 x NTW MFMAs per wave
// and chunk.  Those k-steps stream L2 -> LDS through a 3-slot ring (one slot = the two k-steps of a chunk of every
// tile = the 1 KiB blocks the token-stationary form reads tile by tile) shared by the sixteen waves: one barrier
// per chunk.  Nothing waits for a whole token: the first MFMA starts when the first 128 bytes of each row are in,
// HBM stays busy until the last chunk of a round, and the first chunks of the NEXT round are requested before the
// keys of this one are formed.
// At the end of a round every wave turns its accumulators into keys (one sorted triple per lane) and the eight
// lanes that hold a token merge through LDS (float min of the best key, or of the candidate masks) into the
// same flag word / candidate-code record the re-rank kernel reads (format 3).
//
// Values: u[word] = |c|^2/2 - x~.c~ (accumulators start at |c|^2/2; no per-token shift: keys are compared as
// floats, as in assign_screen2_kernel), window 2E from the fp32 sum of squares of the token.
// vmcnt bookkeeping (LDS-DMA and loads retire in issue order): per wave, barrier(u) - between the two k-steps of
// chunk u - is followed by the copies A(u+2) (two pieces) and tok(u+3) (one piece), and publishes A(u+1) and
// tok(u+1): the only younger operation that may stay in flight across it is tok(u+2).
//
// Measured (MI355X, 50 176 tokens): 34.8 us against 36.2 us for the token-stationary form in isolation, but
// 5 % fewer images/s in the replayed bench (it holds every CU's whole LDS and sixteen waves for the length of the launch):
// opt-in.  History: with a rolled chunk loop (ring positions, wait counts and conditions computed at run time: 170
// instructions per wave and chunk) the stream was bound by the INSTRUCTIONS the CU can issue - 45.8 us, and with the token
// copies, the codebook copies, the barriers, the MFMAs and the fragment reads all compiled out the loop still took 85 % of
// its time.  Fully unrolled (this version: ~70 instructions per wave and chunk, every ring position an immediate) the
// loop is 15.9 k cycles per round against 12.3 k of matrix pipe; what bounds it now is the codebook stream (393 KB per CU
// and round out of L2, ~27 B per cycle and CU), which a round with half of its sets missing does not shorten.
// ------------------------------------------------------------------------------------------
constexpr int kS3RingA = 3, kS3RingT = 3, kS3Sets = 4, kS3Quarters = 4;      // ring slots: codebook chunks (2 k-steps), token chunks

// LDS-DMA pieces of assign_screen3_kernel with every ring position an immediate: the LDS address is M0 + instruction
// offset + 16 x lane, the global address SGPR base + 32-bit lane offset + the same instruction offset (so M0 carries
// the slot minus that offset).
template <int LDS_OFF, int GOFF>
__device__ __forceinline__ void s3_dma1(unsigned voff, const void *sbase, unsigned lds_dst)
{
    static_assert(GOFF >= 0 && GOFF < 4096, "instruction offset");
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%4"
                 :: "v"(voff), "s"(sbase), "s"(lds_dst), "n"(LDS_OFF - GOFF), "n"(GOFF) : "memory", "scc");
}
template <int LDS_OFF>
__device__ __forceinline__ void s3_dma2(unsigned voff, const void *sbase, unsigned lds_dst)     // two consecutive KiB
{
    asm volatile("s_add_u32 m0, %2, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024"
                 :: "v"(voff), "s"(sbase), "s"(lds_dst), "n"(LDS_OFF) : "memory", "scc");
}

template <int NTW, int NCH>
__global__ __launch_bounds__(1024, 4) void assign_screen3_kernel(const AssignArgs p)
{
    constexpr int NT = kS3Quarters * NTW;                       // tiles of the (padded) codebook: 8 or 16
    constexpr int RG = 2;                                       // A fragments read ahead (registers)
    constexpr int kSlotA = NT * 2048;                           // one chunk = two k-steps of every tile: [tile][k-step][1 KiB]
    constexpr int kOffT = kS3RingA * kSlotA;                    // token rings  [4 sets][3][4 KiB]
    constexpr int kOffHc = kOffT + kS3Sets * kS3RingT * 4096;   // |c|^2/2 of every word, accumulator-row order [NT][32]
    constexpr int kOffBest = kOffHc + NT * 128;                 // [4][32] best key of a token
    constexpr int kOffMask = kOffBest + kS3Sets * 32 * 4;       // [4][32] candidate mask being assembled
    static_assert(NTW == 2 || NTW == 4, "tiles per wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *hcs = reinterpret_cast<float *>(smem + kOffHc);
    float *tbest = reinterpret_cast<float *>(smem + kOffBest);
    unsigned *tmask = reinterpret_cast<unsigned *>(smem + kOffMask);

    const int wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int ps = wid >> 2, q = wid & 3;
    const PackLayout lay = pack_layout(p.M, p.D);
    const unsigned char *tiles = p.packed + lay.tiles_off;
    const int wave_id = blockIdx.x * 16 + wid;
    stamp(p, 0, (int)threadIdx.x & 63, wave_id);

    if (threadIdx.x < kS3Sets * 32) { tbest[threadIdx.x] = kBigKey; tmask[threadIdx.x] = 0u; }

    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const unsigned a_dst = __builtin_amdgcn_readfirstlane(lds_base + (wid % NT) * 2048);            // + slot
    const unsigned t_dst = __builtin_amdgcn_readfirstlane(lds_base + kOffT + ps * (kS3RingT * 4096) + q * 1024);    // + buffer
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
    // Sets are dealt round-robin: in round `rnd` workgroup b's slot ps holds set (4 rnd + ps) G + b (G = workgroups), so the
    // last round's sets spread over ALL workgroups (50 176 tokens = 1 568 sets on 256 CUs: round 1 has four sets everywhere,
    // round 2 two or three): a slot without a set only serves the codebook ring and the barriers, and a SIMD that hosts
    // fewer computing waves finishes its chunk sooner (the loop is bound by the matrix pipe).
    const int64_t G = gridDim.x;
    const int64_t n_rounds = (p.n_sets3 + kS3Sets * G - 1) / (kS3Sets * G);

    for (int64_t rnd = 0; rnd < n_rounds; ++rnd) {
        // (everything that depends on the lane is formed again in every round, behind an opaque copy of the thread id:
        // hoisted out of the round loop these values do not fit beside the accumulators and are spilled, and a spill reload
        // waits for vmcnt(0), i.e. for the token chunks in flight)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        const int lane = tid & 63, r = lane & 31, h = lane >> 5;
        // copies: codebook piece = tile (wid % NT), lane offset inside the image; token piece = rows 8 q + lane / 8 of the set
        // (a row past the set or past the last token: the set's first row, or token 0: no wave skips a copy)
        const unsigned a_voff = (unsigned)((wid % NT) * lay.tile_bytes + lane * 16);
        auto set_of = [&](int64_t rd) -> int64_t { return (kS3Sets * rd + ps) * G + blockIdx.x; };
        auto tok_voff = [&](int64_t rd) -> unsigned {
            const int64_t tok0 = set_of(rd) * p.tps;
            const int rq = 8 * q + (lane >> 3);
            const int64_t nq = tok0 + rq;
            const bool ok = rq < p.tps && nq < p.n_tokens;
            const int64_t nn = ok ? nq : (tok0 < p.n_tokens ? tok0 : 0);
            const unsigned ni = (unsigned)p.n_inner, o = (unsigned)nn / ni, i = (unsigned)nn - o * ni;
            return (unsigned)(((int64_t)o * p.xso + (int64_t)i * p.xsi) * 4 + 16 * ((lane & 7) ^ ((rq >> 1) & 7)));
        };
        // issue order of a round's first copies: T0, A0, T1, A1, T2 (one barrier later: A2, T3, ...)
        auto begin_round = [&](unsigned tv, bool act) {
            if (act) s3_dma1<0 * 4096, 0>(tv, p.x, t_dst);
            s3_dma2<0 * kSlotA>(a_voff, tiles, a_dst);
            if (NCH > 1) { if (act) s3_dma1<1 * 4096, 128>(tv, p.x, t_dst); s3_dma2<1 * kSlotA>(a_voff, tiles + 2048, a_dst); }
            if (NCH > 2 && act) s3_dma1<2 * 4096, 256>(tv, p.x, t_dst);
        };
        const bool active = set_of(rnd) < p.n_sets3;            // wave-uniform
        unsigned t_voff = tok_voff(rnd);
        if (rnd == 0) {
            begin_round(t_voff, active);
            // |c|^2/2 of every word -> LDS, once (behind the first copies: its wait is theirs).  Padding words: +inf in the
            // image, kept finite here so that keys never become NaNs.
            if (tid < NT * 32)
                hcs[tid] = fminf(reinterpret_cast<const float *>(tiles + (size_t)(tid >> 5) * lay.tile_bytes + (size_t)lay.n_steps * 1024)[tid & 31], kPadHalfNorm);
        }

        // ---- accumulators start at |c|^2/2
        f32x16 acc[NTW];
        if (rnd == 0) { stamp(p, 6, lane, wave_id); __syncthreads(); stamp(p, 7, lane, wave_id); }                          // (hcs written above)
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c4 = *reinterpret_cast<const float4 *>(hcs + (q * NTW + i) * 32 + (g * 2 + h) * 4);
                acc[i][4 * g + 0] = c4.x; acc[i][4 * g + 1] = c4.y; acc[i][4 * g + 2] = c4.z; acc[i][4 * g + 3] = c4.w;
            }
        }
        if (active) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NCH > 2 ? 4 : (NCH > 1 ? 3 : 0)) : "memory");     // T0 and A0 are in (younger: T1, A1 x 2, T2)
        else asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NCH > 1 ? 2 : 0) : "memory");                           // A0 is in (younger: A1 x 2)
        if (rnd == 0) stamp(p, 8, lane, wave_id);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (rnd == 0) stamp(p, 11, lane, wave_id);

        // token chunk of this wave's set -> the lane's 16 floats (row r, k = 32 u + 16 h + 0..15)
        const unsigned char *t_frag = smem + kOffT + ps * (kS3RingT * 4096) + (r >> 3) * 1024 + (r & 7) * 128;
        const int sw = (r >> 1) & 7;
        const unsigned char *tq0 = t_frag + ((4 * h + 0) ^ sw) * 16, *tq1 = t_frag + ((4 * h + 1) ^ sw) * 16;
        const unsigned char *tq2 = t_frag + ((4 * h + 2) ^ sw) * 16, *tq3 = t_frag + ((4 * h + 3) ^ sw) * 16;
        const unsigned char *a_frag = smem + (q * NTW) * 2048 + lane * 16;
        f32x4 raw[4];
        float sumsq = 0.0f;
        auto convert = [&](half8 &b, const f32x4 &lo, const f32x4 &hi) {
            sumsq = fmaf(lo.x, lo.x, sumsq); sumsq = fmaf(lo.y, lo.y, sumsq); sumsq = fmaf(lo.z, lo.z, sumsq); sumsq = fmaf(lo.w, lo.w, sumsq);
            sumsq = fmaf(hi.x, hi.x, sumsq); sumsq = fmaf(hi.y, hi.y, sumsq); sumsq = fmaf(hi.z, hi.z, sumsq); sumsq = fmaf(hi.w, hi.w, sumsq);
            b[0] = (_Float16)lo.x; b[1] = (_Float16)lo.y; b[2] = (_Float16)lo.z; b[3] = (_Float16)lo.w;
            b[4] = (_Float16)hi.x; b[5] = (_Float16)hi.y; b[6] = (_Float16)hi.z; b[7] = (_Float16)hi.w;
            asm volatile("" : "+v"(sumsq));                     // (otherwise the whole chain of squares sinks to the epilogue and every chunk's floats are spilled until then)
        };
        half8 bc0, bc1;
        if (active) {
            raw[0] = *reinterpret_cast<const f32x4 *>(tq0); raw[1] = *reinterpret_cast<const f32x4 *>(tq1);
            raw[2] = *reinterpret_cast<const f32x4 *>(tq2); raw[3] = *reinterpret_cast<const f32x4 *>(tq3);
            convert(bc0, raw[0], raw[1]);
            convert(bc1, raw[2], raw[3]);
        }
        stamp(p, rnd == 0 ? 12 : 1, lane, wave_id);

        // ---- main loop, fully unrolled (NCH chunks): one barrier per chunk, between its two k-steps; every ring position,
        // wait count and condition is a constant.  Fragment c = e NTW + i of a chunk sits at (2 i + e) KiB of this wave's
        // part of the slot; the register ring runs RG fragments ahead (the last RG of a chunk fetch the next chunk's first).
        if (active) {
        half8 ar[RG];
#pragma unroll
        for (int c = 0; c < RG; ++c) ar[c] = *reinterpret_cast<const half8 *>(a_frag + ((c % NTW) * 2 + c / NTW) * 1024);
        static_for<NCH>([&](auto u_c) {
            constexpr int u = decltype(u_c)::value;
            constexpr int slot = (u % kS3RingA) * kSlotA, nslot = ((u + 1) % kS3RingA) * kSlotA;
            auto k_step = [&](const half8 &b, auto e_c) {
                constexpr int e = decltype(e_c)::value;
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const int c = e * NTW + i, cn = c + RG;
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[c % RG], b, acc[i], 0, 0, 0);
                    if (cn < 2 * NTW) ar[c % RG] = *reinterpret_cast<const half8 *>(a_frag + slot + ((cn % NTW) * 2 + cn / NTW) * 1024);
                    else if (u + 1 < NCH) ar[c % RG] = *reinterpret_cast<const half8 *>(a_frag + nslot + (((cn - 2 * NTW) % NTW) * 2 + (cn - 2 * NTW) / NTW) * 1024);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            k_step(bc0, std::integral_constant<int, 0>{});
            if constexpr (u + 2 < NCH) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");      // (in flight across the barrier: T(u+2))
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                       // chunk u+1 (codebook and tokens) is in LDS for everybody; slot of chunk u-1 and buffer of chunk u are free
            asm volatile("" ::: "memory");
            if constexpr (u + 2 < NCH) s3_dma2<((u + 2) % kS3RingA) * kSlotA>(a_voff, tiles + (size_t)(u + 2) * 2048, a_dst);
            if constexpr (u + 3 < NCH) s3_dma1<((u + 3) % kS3RingT) * 4096, 128 * (u + 3)>(t_voff, p.x, t_dst);
            if constexpr (u + 1 < NCH) {
                constexpr int rb = ((u + 1) % kS3RingT) * 4096;
                raw[0] = *reinterpret_cast<const f32x4 *>(tq0 + rb); raw[1] = *reinterpret_cast<const f32x4 *>(tq1 + rb);
                raw[2] = *reinterpret_cast<const f32x4 *>(tq2 + rb); raw[3] = *reinterpret_cast<const f32x4 *>(tq3 + rb);
                convert(bc0, raw[0], raw[1]);                   // (bc0 of chunk u has been issued to the matrix pipe)
            }
            k_step(bc1, std::integral_constant<int, 1>{});
            if constexpr (u + 1 < NCH) convert(bc1, raw[2], raw[3]);
        });
        } else {                                                // a slot without a set: its share of the codebook copies, and the barriers
            static_for<NCH>([&](auto u_c) {
                constexpr int u = decltype(u_c)::value;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (u + 2 < NCH) s3_dma2<((u + 2) % kS3RingA) * kSlotA>(a_voff, tiles + (size_t)(u + 2) * 2048, a_dst);
            });
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // nobody reads the rings any more
        asm volatile("" ::: "memory");
        stamp(p, rnd == 0 ? 13 : 2, lane, wave_id);
        // ---- the next round's first chunks travel while this round's keys are formed
        const int64_t tok0 = set_of(rnd) * p.tps;
        if (rnd + 1 < n_rounds) { t_voff = tok_voff(rnd + 1); begin_round(t_voff, set_of(rnd + 1) < p.n_sets3); }

        // ---- keys: one sorted triple per lane; code = tile << 4 | accumulator register
        // (volatile asm, four instructions per value: left to itself hipcc forms the three chains one after the other and
        // keeps every intermediate minimum alive - spills.  The asm reads MFMA results behind the hazard recogniser's
        // back: the barrier and the copies above are far more than the last MFMA's write-back.)
        float m1 = kBigKey, m2 = kBigKey, m3 = kBigKey;
        unsigned keymask = 0xFFFFFF00u;
        asm volatile("s_nop 15\n\ts_nop 15" : "+s"(keymask));
        if (active)
#pragma unroll
        for (int i = 0; i < NTW; ++i) {
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                float k;
                asm volatile("v_and_or_b32 %0, %4, %5, %6\n\t"
                             "v_med3_f32 %3, %0, %2, %3\n\t"
                             "v_med3_f32 %2, %0, %1, %2\n\t"
                             "v_min_f32 %1, %0, %1"
                             : "=&v"(k), "+v"(m1), "+v"(m2), "+v"(m3) : "v"(acc[i][x]), "s"(keymask), "n"((i << 4) | x));
            }
        }
        sumsq += __shfl_xor(sumsq, 32, SN_WAVE);
        // ---- per-token error window (DESIGN.md "S1 error window"): |key - exact value| <= E
        const unsigned *scal = reinterpret_cast<const unsigned *>(p.packed + lay.scal_off);
        const float C2 = __uint_as_float(scal[0]), C1 = __uint_as_float(scal[1]), CN = __uint_as_float(scal[2]), CMAX = __uint_as_float(scal[3]);
        const float X2 = sqrtf(sumsq) * 1.001f + 1.0e-6f, X1 = X2 * sqrtf((float)p.D);
        const float vmax = 0.5f * CN + X2 * C2;                               // >= |any partial sum|
        const float E = 1.01f * (2.01f * kU16 * X2 * C2 + 5.96e-8f * (X1 + C1)
                                 + (float)(2 * NCH) * kAccUlpPerMfma * vmax + vmax * (3.0f * 5.96e-8f + 3.1e-5f));
        const bool bad = !(sumsq <= kHugeIn * kHugeIn) || !(CMAX <= kHugeIn) || !(vmax < 1.0e28f);   // NaN-safe; |x|_2 <= 3e4 bounds every component
        const float window = 2.0f * E;
        const int64_t n = tok0 + r;
        const bool valid = active && r < p.tps && n < p.n_tokens;
        const float kmin = fminf(m1, __shfl_xor(m1, 32, SN_WAVE));
        if (valid && h == 0) __builtin_amdgcn_ds_fminf((__attribute__((address_space(3))) float *)&tbest[ps * 32 + r], kmin, 0, 0, false);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const float best = tbest[ps * 32 + r];
        const bool any_finite = best < kKeyLimit;
        const float cut = best + window;
        const unsigned hmask = (m1 <= cut ? 1u : 0u) | (m2 <= cut ? 2u : 0u) | (m3 <= cut ? 4u : 0u);
        const bool hover = m3 <= cut;                                         // a 4th may hide behind it
        const unsigned contrib = (bad || !any_finite) ? 0x80000000u : ((hmask << (3 * (2 * q + h))) | (hover ? 0x80000000u : 0u));
        if (valid && contrib) __hip_atomic_fetch_or(&tmask[ps * 32 + r], contrib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        const unsigned mk = tmask[ps * 32 + r];
        const unsigned cand = mk & 0xFFFFFFu;
        const bool overflow = (mk >> 31) != 0u || cand == 0u;
        const int nc = __popc(cand);
        if (valid && !overflow) {
            if (nc == 1 && hmask != 0u) {                                     // the only candidate: final
                const unsigned code = __float_as_uint(m1) & 0xFFu;
                p.out[out_index(p, n)] = (q * NTW + (int)(code >> 4)) * 32 + 8 * (int)((code >> 2) & 3u) + 4 * h + (int)(code & 3u);
            }
            if (nc > 1) {                                       // the eight lanes of the token write their three codes
                unsigned char *cd = p.codes + (int64_t)n * kCodeBytes + 3 * (2 * q + h);
                cd[0] = (unsigned char)(__float_as_uint(m1) & 0xFFu);
                cd[1] = (unsigned char)(__float_as_uint(m2) & 0xFFu);
                cd[2] = (unsigned char)(__float_as_uint(m3) & 0xFFu);
            }
        }
        const bool writer = valid && q == 0 && h == 0;
        if (writer) p.flags[n] = overflow ? 0x80000000u : (nc > 1 ? cand : 0u);
        const bool need_b = writer && overflow;
        const unsigned long long mask_b = __ballot(need_b);
        if (mask_b) {                                          // rare: tokens the screen cannot bound (phase B of the re-rank writes out[])
            int base = 0;
            const int leader = __ffsll((long long)mask_b) - 1;
            if (lane == leader) base = atomicAdd(&p.work[1], __popcll(mask_b));
            base = __shfl(base, leader, SN_WAVE);
            if (need_b) p.overflow[base + __popcll(mask_b & ((1ull << lane) - 1ull))] = (int)n;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // everybody has read the merged words: reset them for the next round
        asm volatile("" ::: "memory");
        if (q == 0 && h == 0) { tbest[ps * 32 + r] = kBigKey; tmask[ps * 32 + r] = 0u; }
        stamp(p, rnd == 0 ? 14 : 3, lane, wave_id);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // no DMA may be in flight when the LDS is released
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep_m0));
}

// ------------------------------------------------------------------------------------------
// mode 0, pass 1, one-round K-outer form (round 4; sn_assign_set_variant(4); codebooks of 16 tiles of 32 words - 448 < M <= 512 -,
// D in {192, 384}, fp32 tokens, at most kS4Rows tokens per CU): the accumulators of EVERY token of the workgroup against the
// whole codebook stay in registers for the length of the launch, so the codebook crosses a CU's memory path once and the
// token stream runs under the matrix pipe.
//
// One workgroup of four 512-register waves per CU.  Wave q holds, for its quarter of the words (4 tiles), the accumulators of
// six 32-token sets (24 tiles x 16 registers: sets 0-3 in the 256 AGPRs, sets 4-5 in VGPRs - the compiler keeps the
// MFMAs of a function in one register file, so the instruction is written out) and of up to sixteen further tokens
// ("leftover": v_mfma_f32_16x16x32_f16, 8 blocks of 16 words x 4 registers): 196 tokens per CU = 50 176 / 256 is 6 x 32 + 4.
// K is the outer loop: per 32-float chunk the token rows (LDS-DMA, whole 128-byte lines, piece-swizzled; three slots, two
// chunks ahead - HBM has the long latency) and the two k-steps of every tile of the codebook image (two slots, one ahead,
// L2) arrive while the previous chunk is multiplied; one barrier per chunk.  A wave forms the fp16 B fragments it multiplies
// from the raw fp32 rows itself, right before their MFMAs (no fragment buffer, no second barrier).
// Values, keys, error window, merge through LDS and the flag word / candidate record (format 3: slot 3 (2 q + h) + j) are
// those of assign_screen3_kernel; the leftover tokens' keys of the four accumulator-row groups are merged pairwise
// (lanes l, l ^ 32) so that what is left has the slot structure of a 32 x 32 tile (code bits: see kS4 comments below).
// Probe (tools/proto_screen4.hip): intake alone 12.9 us (6 TB/s), with the MFMAs 21 us of kernel time.
// Measured (MI355X, 50 176 tokens): 34.5 us of kernel time (40.4 by the library's event pair; the default form: 30.4 / 36.3).
// The main loop is ~19 us, but the keys of 448 accumulator values per lane are 2 300 dependent VALU instructions on a wave
// that is ALONE on its SIMD - 9 us with nothing to overlap them with in a one-round form - and the merge another 4.5 us.
// (Two independent key chains per set in one asm statement: no faster - a lone wave issues ~one VALU instruction per 8 cycles
// whatever their dependences; only a second wave on the SIMD would hide the keys.)
// Opt-in; DESIGN 8.  What this kernel ran into, all because an asm MFMA is invisible to the compiler (hazard recogniser,
// register allocator): (1) a VALU conversion scheduled right in front of the first MFMA of a group fed it a stale B register
// (one tile in four of a set wrong): wait states are written into the asm; (2) VALU work dealt BETWEEN the four MFMAs of a
// step (38 us, 2.4 us faster) gave whole sets of garbage although the instruction stream read correctly (bisected: the eight squares
// - reads only - between the MFMAs are harmless, the four v_cvt_pk in front of the fourth MFMA are what breaks it, although they
// write nobody's operands; keeping every fragment alive three more steps changes nothing; `s_nop 3` in front of that MFMA cures it:
// an asm MFMA right behind VALU writes needs wait states whatever the registers.  With them the interleaved loop is correct - and
// no faster than this one: 40.2 us) - the next step's conversion runs behind the fourth MFMA; (3) any instrumentation between the phases (stamps)
// made the compiler park whole accumulator sets in scratch - stamps 0 and 4 only; (4) a lambda nested in the kernel's generic
// lambdas does not capture a variable that only appears as an asm operand (clang): the MFMA is a function.
// ------------------------------------------------------------------------------------------
// one MFMA of assign_screen4_kernel, accumulator in the AGPRs (AG) or the VGPRs, NOPS wait states in front
template <bool AG, int NOPS>
__device__ __forceinline__ void s4_mfma(f32x16 &ac, const half8 &a, const half8 &b)
{
    if constexpr (AG) {
        if constexpr (NOPS == 3) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ac) : "v"(a), "v"(b));
        else if constexpr (NOPS == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ac) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(ac) : "v"(a), "v"(b));
    } else {
        if constexpr (NOPS == 3) asm volatile("s_nop 3\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ac) : "v"(a), "v"(b));
        else if constexpr (NOPS == 1) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ac) : "v"(a), "v"(b));
        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(ac) : "v"(a), "v"(b));
    }
}

constexpr int kS4Sets = 6, kS4Lo = 16;                          // full sets, leftover tokens
constexpr int kS4Rows = kS4Sets * 32 + kS4Lo;                   // tokens a workgroup can hold (208)
constexpr int kS4RowsPad = 224;                                 // token rows copied per chunk (28 x 8: seven 1 KiB copies per wave)
constexpr int kS4Slab = 16 * 2048, kS4Tok = kS4RowsPad * 128;   // bytes per chunk: codebook (16 tiles x 2 k-steps), token rows
constexpr int kS4OffT = 2 * kS4Slab;
constexpr int kS4OffHc = kS4OffT + 3 * kS4Tok;                  // |c|^2/2, accumulator-row order [16][32]
constexpr int kS4OffBest = kS4OffHc + 16 * 128;                 // [7][32] best key of a token
constexpr int kS4OffMask = kS4OffBest + 7 * 128;                // [7][32] candidate mask being assembled
constexpr int kS4OffSum = kS4OffMask + 7 * 128;                 // [7][32] |x|^2 of a token
constexpr int kS4Lds = kS4OffSum + 7 * 128;

template <int NCH>
__global__ __launch_bounds__(256, 1) void assign_screen4_kernel(const AssignArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *hcs = reinterpret_cast<float *>(smem + kS4OffHc);
    float *tbest = reinterpret_cast<float *>(smem + kS4OffBest);
    unsigned *tmask = reinterpret_cast<unsigned *>(smem + kS4OffMask);
    float *tsum = reinterpret_cast<float *>(smem + kS4OffSum);
    const int tid = threadIdx.x, lane = tid & 63, q = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;                      // 32 x 32 side: token column, accumulator row half = k half of a fragment
    const int j16 = lane & 15, kg = lane >> 4;                   // 16 x 16 side: token column, k group of a fragment = accumulator row group
    const PackLayout lay = pack_layout(p.M, p.D);
    const unsigned char *tiles = p.packed + lay.tiles_off;
    const int64_t tok0 = (int64_t)blockIdx.x * p.tps4;           // this workgroup's tokens: [tok0, tok0 + n_mine)
    const int n_mine = (int)(p.n_tokens - tok0 < p.tps4 ? p.n_tokens - tok0 : p.tps4);
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int wave_id = blockIdx.x * 4 + q;
    stamp(p, 0, lane, wave_id);

    if (tid < 7 * 32) { tbest[tid] = kBigKey; tmask[tid] = 0u; tsum[tid] = 0.0f; }
    // |c|^2/2 of every word (padding words: +inf in the image, kept finite so that keys never become NaNs)
    for (int i = tid; i < 16 * 32; i += 256)
        hcs[i] = fminf(reinterpret_cast<const float *>(tiles + (size_t)(i >> 5) * lay.tile_bytes + (size_t)lay.n_steps * 1024)[i & 31], kPadHalfNorm);

    // ---- copies.  Token rows: copy i of wave q covers the rows 8 (q + 4 i) .. + 7 of the workgroup (a row past its tokens: its
    // first token - nobody reads it); lane -> (row, piece slot), the slot holds piece slot ^ ((row >> 1) & 7) of the line.
    unsigned tv[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
        const int row = 8 * (q + 4 * i) + (lane >> 3);
        const int64_t n = tok0 + (row < n_mine ? row : 0);
        const unsigned ni = (unsigned)p.n_inner, o = (unsigned)n / ni, ii = (unsigned)n - o * ni;
        tv[i] = (unsigned)(((int64_t)o * p.xso + (int64_t)ii * p.xsi) * 4 + 16 * ((lane & 7) ^ ((row >> 1) & 7)));
    }
    auto issue_tok = [&](int c) {
        const unsigned slot = (unsigned)(c % 3);
#pragma unroll
        for (int i = 0; i < 7; ++i) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + kS4OffT + slot * kS4Tok + (q + 4 * i) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(tv[i] + (unsigned)c * 128u), "s"(p.x), "s"(dst) : "memory");
        }
    };
    // codebook: wave q copies the two k-steps of its four tiles (8 KiB of the chunk's 32)
    const unsigned av = (unsigned)(lane * 16);
    auto issue_slab = [&](int c) {
        const unsigned slot = (unsigned)(c & 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int tile = 4 * q + (i >> 1);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * kS4Slab + tile * 2048 + (i & 1) * 1024);
            const unsigned src = __builtin_amdgcn_readfirstlane((unsigned)(tile * lay.tile_bytes + c * 2048 + (i & 1) * 1024));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(av + src), "s"(tiles), "s"(dst) : "memory");
        }
    };
    // issue order: slab(0), tok(0), tok(1); per chunk c: slab(c + 1), tok(c + 2)
    issue_slab(0); issue_tok(0);
    if (NCH > 1) issue_tok(1);
    __syncthreads();                                            // hcs / the merge words are written

    // ---- accumulators start at |c|^2/2
    f32x16 acc[kS4Sets][4];
    f32x4 accl[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        f32x16 c0;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 c4 = *reinterpret_cast<const float4 *>(hcs + (q * 4 + i) * 32 + (g * 2 + h) * 4);
            c0[4 * g + 0] = c4.x; c0[4 * g + 1] = c4.y; c0[4 * g + 2] = c4.z; c0[4 * g + 3] = c4.w;
        }
#pragma unroll
        for (int s = 0; s < kS4Sets; ++s) acc[s][i] = c0;
    }
#pragma unroll
    for (int b = 0; b < 8; ++b) {                                // block b = words 16 b .. + 15 of the quarter; lane: rows 4 kg .. + 3 of it
        const int rho = 16 * (b & 1) + 4 * kg;                   // row inside the tile b >> 1 (a multiple of 4)
        const float4 c4 = *reinterpret_cast<const float4 *>(hcs + (q * 4 + (b >> 1)) * 32 + (2 * (rho >> 3) + ((rho >> 2) & 1)) * 4);
        accl[b] = f32x4{c4.x, c4.y, c4.z, c4.w};
    }
    float sumsq0 = 0.0f, sumsq1 = 0.0f;                          // |x|^2 of the lane's share of the sets q, q + 4 (set 6 = the leftover tokens)
    auto squares = [&](float &acc_, const f32x4 &lo, const f32x4 &hi) {
        acc_ = fmaf(lo.x, lo.x, acc_); acc_ = fmaf(lo.y, lo.y, acc_); acc_ = fmaf(lo.z, lo.z, acc_); acc_ = fmaf(lo.w, lo.w, acc_);
        acc_ = fmaf(hi.x, hi.x, acc_); acc_ = fmaf(hi.y, hi.y, acc_); acc_ = fmaf(hi.z, hi.z, acc_); acc_ = fmaf(hi.w, hi.w, acc_);
        asm volatile("" : "+v"(acc_));
    };
    auto to_half8 = [](const f32x4 &lo, const f32x4 &hi) {
        half8 b;
        b[0] = (_Float16)lo.x; b[1] = (_Float16)lo.y; b[2] = (_Float16)lo.z; b[3] = (_Float16)lo.w;
        b[4] = (_Float16)hi.x; b[5] = (_Float16)hi.y; b[6] = (_Float16)hi.z; b[7] = (_Float16)hi.w;
        return b;
    };
    const int swz = (r >> 1) & 7, swz16 = (j16 >> 1) & 7;        // (rows 32 s + r and 192 + j16: the set offset does not touch bits 1..3)
    // lane parts of the LDS addresses.  B fragment of k-step ks: k half h of the image = floats 16 h + 8 ks .. + 7 of the chunk
    // (pack_frag_kernel) = pieces 4 h + 2 ks, + 1 of the token's line
    const unsigned a_lane = (unsigned)(q * 4 * 2048 + lane * 16);
    unsigned t_lo[2], t_hi[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        t_lo[ks] = (unsigned)(r * 128 + (((4 * h + 2 * ks) ^ swz) << 4));
        t_hi[ks] = (unsigned)(r * 128 + (((4 * h + 2 * ks + 1) ^ swz) << 4));
    }
    const int pl = 4 * (kg & 1) + 2 * (kg >> 1);
    const unsigned l_lo = (unsigned)((32 * kS4Sets + j16) * 128 + ((pl ^ swz16) << 4));
    const unsigned l_hi = (unsigned)((32 * kS4Sets + j16) * 128 + (((pl + 1) ^ swz16) << 4));
    const unsigned a16_lane = (unsigned)(q * 4 * 2048 + (kg >> 1) * 1024 + (j16 + 32 * (kg & 1)) * 16);

    // (which of the sets this wave sums the squares of, as factors: a branch inside the loop makes the compiler move every
    // accumulator through scratch at the loop's back edge - 200 registers per chunk)
    float sel0[kS4Sets], sel1[kS4Sets];
#pragma unroll
    for (int s = 0; s < kS4Sets; ++s) { sel0[s] = s == q ? 1.0f : 0.0f; sel1[s] = s == q + 4 ? 1.0f : 0.0f; }
    const float sel_lo = q == 2 ? 1.0f : 0.0f;

    // ---- main loop, fully unrolled (no back edge: the accumulators never move)
    static_for<NCH>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        // outstanding, oldest first: tok(c), slab(c), tok(c + 1): everything but the seven copies of tok(c + 1)
        if constexpr (c + 1 < NCH) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // chunk c is in LDS for everybody; the slots of chunk c - 1 are free
        asm volatile("" ::: "memory");
        if constexpr (c + 1 < NCH) issue_slab(c + 1);
        if constexpr (c + 2 < NCH) issue_tok(c + 2);
        // (LDS addresses: lane part + an immediate; the token slot's base goes through an opaque scalar per chunk - with the loop
        // unrolled the compiler otherwise keeps the addresses of every (set, k-step, slot) alive across chunks, in scratch)
        unsigned rb = (unsigned)(kS4OffT + (c % 3) * kS4Tok);
        asm volatile("" : "+s"(rb));
        constexpr int sl = (c & 1) * kS4Slab;
        // Software pipeline over the chunk's twelve steps (k-step, set): a wave alone on its SIMD issues in order, so the NEXT
        // step's raw rows are requested in front of THIS step's four MFMAs and converted right behind them, in the shadow of the
        // queued MFMAs (32 cycles of matrix pipe each) - with the request, the wait and the conversion of a step in front of its
        // own MFMAs the loop ran at 1.75 us per chunk against 0.9 us of matrix pipe.  sched_barrier pins the order the source gives.
        half8 a0[4], a1[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) a0[t] = *reinterpret_cast<const half8 *>(smem + a_lane + (sl + t * 2048));
        const unsigned char *plo0 = smem + rb + t_lo[0], *phi0 = smem + rb + t_hi[0];
        const unsigned char *plo1 = smem + rb + t_lo[1], *phi1 = smem + rb + t_hi[1];
        f32x4 lo = *reinterpret_cast<const f32x4 *>(plo0), hi = *reinterpret_cast<const f32x4 *>(phi0);
        // (an MFMA reads its A / B registers when it STARTS, and up to four of them queue in front of the matrix pipe: a register
        // the compiler believes free right behind the asm statement - for the next LDS read, the next conversion - may be
        // overwritten before the queued MFMA has read it.  Seen as whole sets of garbage.  So every operand is kept alive one
        // step longer by an empty asm that "reads" it: bprev below, the A fragments behind their k-step.)
        half8 bcur, bprev;
        {
            float sq = 0.0f;
            squares(sq, lo, hi);
            sumsq0 = fmaf(sel0[0], sq, sumsq0);
            sumsq1 = fmaf(sel1[0], sq, sumsq1);
            asm volatile("" : "+v"(sumsq0), "+v"(sumsq1));
            bcur = to_half8(lo, hi);
            bprev = bcur;
        }
        static_for<2 * kS4Sets>([&](auto j_c) {
            constexpr int j = decltype(j_c)::value, ks = j / kS4Sets, st = j % kS4Sets;
            constexpr int jn = j + 1, ksn = jn / kS4Sets, sn = jn % kS4Sets;
            constexpr bool more = jn < 2 * kS4Sets;
            // the next step's raw rows (and, in the middle of k-step 0, the A fragments of k-step 1)
            if constexpr (more) {
                lo = *reinterpret_cast<const f32x4 *>((ksn ? plo1 : plo0) + sn * 4096);
                hi = *reinterpret_cast<const f32x4 *>((ksn ? phi1 : phi0) + sn * 4096);
            }
            if constexpr (j == 2) {
#pragma unroll
                for (int t = 0; t < 4; ++t) a1[t] = *reinterpret_cast<const half8 *>(smem + a_lane + (sl + t * 2048 + 1024));
            }
            __builtin_amdgcn_sched_barrier(0);
            constexpr bool AG = st < 4;
            constexpr int N0 = j == 0 ? 3 : 1;                    // (first MFMA of the chunk: its B fragment has just been packed)
            float sq = 0.0f;
            half8 bnext;
            s4_mfma<AG, N0>(acc[st][0], ks ? a1[0] : a0[0], bcur);
            s4_mfma<AG, 0>(acc[st][1], ks ? a1[1] : a0[1], bcur);
            s4_mfma<AG, 0>(acc[st][2], ks ? a1[2] : a0[2], bcur);
            s4_mfma<AG, 0>(acc[st][3], ks ? a1[3] : a0[3], bcur);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (more) {
                squares(sq, lo, hi);
                sumsq0 = fmaf(sel0[sn], sq, sumsq0);
                sumsq1 = fmaf(sel1[sn], sq, sumsq1);
                asm volatile("" : "+v"(sumsq0), "+v"(sumsq1));
                bnext = to_half8(lo, hi);
                asm volatile("" : "+v"(bnext));
            }
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("" :: "v"(bprev));                       // (the previous step's B fragment may go now)
            bprev = bcur;
            if constexpr (more) bcur = bnext;
            if constexpr (j == kS4Sets + 1) asm volatile("" :: "v"(a0[0]), "v"(a0[1]), "v"(a0[2]), "v"(a0[3]));
        });
        asm volatile("s_nop 7" ::: "memory");                   // (the last MFMAs' operand registers are free game for the compiler from here on)
        {   // leftover tokens: one 16 x 16 x 32 step per block and chunk.  Lane (j16, kg) multiplies the image's (k-step kg >> 1, k half
            // kg & 1) = floats 16 (kg & 1) + 8 (kg >> 1) .. + 7 of the chunk = pieces pl, pl + 1
            const f32x4 lo = *reinterpret_cast<const f32x4 *>(smem + rb + l_lo);
            const f32x4 hi = *reinterpret_cast<const f32x4 *>(smem + rb + l_hi);
            float sq = 0.0f;
            squares(sq, lo, hi);
            sumsq1 = fmaf(sel_lo, sq, sumsq1);
            asm volatile("" : "+v"(sumsq1));
            const half8 b = to_half8(lo, hi);
            // A fragment of block bl: words 16 (bl & 1) + j16 of tile bl >> 1, k = 8 kg .. + 7: k-step kg >> 1, k half kg & 1 of the image
            // (all eight read up front into registers of their own, and everything kept alive behind the last MFMA: see above)
            half8 a16[8];
#pragma unroll
            for (int bl = 0; bl < 8; ++bl) a16[bl] = *reinterpret_cast<const half8 *>(smem + a16_lane + (sl + (bl >> 1) * 2048 + (bl & 1) * 256));
#pragma unroll
            for (int bl = 0; bl < 8; ++bl) {
                f32x4 &al = accl[bl];
                if (bl == 0) asm volatile("s_nop 3\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(al) : "v"(a16[bl]), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(al) : "v"(a16[bl]), "v"(b));
            }
            asm volatile("s_nop 15\n\ts_nop 15" :: "v"(bprev), "v"(bcur), "v"(a1[0]), "v"(a1[1]), "v"(a1[2]), "v"(a1[3]));
            asm volatile("s_nop 15\n\ts_nop 15" :: "v"(b), "v"(a16[0]), "v"(a16[1]), "v"(a16[2]), "v"(a16[3]), "v"(a16[4]), "v"(a16[5]), "v"(a16[6]), "v"(a16[7]));
        }
    });
    // (the asm MFMAs are invisible to the hazard recogniser and their results are read below - first those of the LAST ones
    // issued: up to twelve of them are still queued in front of the matrix pipe when the loop ends, ~250 cycles of work)
#pragma unroll
    for (int i = 0; i < 5; ++i) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

    // ---- keys: one sorted triple per lane and set; code = tile << 4 | accumulator register
    float m1[7], m2[7], m3[7];
    unsigned keymask = 0xFFFFFF00u;
    asm volatile("" : "+s"(keymask));
#pragma unroll
    for (int s = 0; s < kS4Sets; ++s) {
        m1[s] = kBigKey; m2[s] = kBigKey; m3[s] = kBigKey;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                float k;
                const float v = acc[s][i][x];
                asm volatile("v_and_or_b32 %0, %4, %5, %6\n\t"
                             "v_med3_f32 %3, %0, %2, %3\n\t"
                             "v_med3_f32 %2, %0, %1, %2\n\t"
                             "v_min_f32 %1, %0, %1"
                             : "=&v"(k), "+v"(m1[s]), "+v"(m2[s]), "+v"(m3[s]) : "v"(v), "s"(keymask), "n"((i << 4) | x));
            }
        }
    }
    {   // leftover tokens.  Lane (j16, kg) holds, of block bl, the words 16 bl + 4 kg + e: in the frame of a 32 x 32 tile that is
        // tile bl >> 1, accumulator row 16 (bl & 1) + 4 kg + e = 8 A + 4 H + e with A = 2 (bl & 1) + (kg >> 1), H = kg & 1: the code
        // (bl >> 1) << 4 | A << 2 | e in the slot of row half H decodes like every other code.  The lanes kg and kg ^ 2 share H:
        // their triples are merged (lane ^ 32) and the lanes kg < 2 carry on as (r = j16, h = kg).
        float a1 = kBigKey, a2 = kBigKey, a3 = kBigKey;
        const unsigned lane_code = (unsigned)((kg >> 1) << 2);
#pragma unroll
        for (int bl = 0; bl < 8; ++bl) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const unsigned code = lane_code | (unsigned)(((bl >> 1) << 4) | ((2 * (bl & 1)) << 2) | e);
                const float k = __uint_as_float((__float_as_uint(accl[bl][e]) & 0xFFFFFF00u) | code);
                a3 = __builtin_amdgcn_fmed3f(k, a2, a3);
                a2 = __builtin_amdgcn_fmed3f(k, a1, a2);
                a1 = fminf(k, a1);
            }
        }
        const float o1 = __shfl_xor(a1, 32, SN_WAVE), o2 = __shfl_xor(a2, 32, SN_WAVE), o3 = __shfl_xor(a3, 32, SN_WAVE);
        const float ks3[3] = {o1, o2, o3};
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const float k = ks3[u];
            a3 = __builtin_amdgcn_fmed3f(k, a2, a3);
            a2 = __builtin_amdgcn_fmed3f(k, a1, a2);
            a1 = fminf(k, a1);
        }
        m1[6] = a1; m2[6] = a2; m3[6] = a3;
    }
    // ---- |x|^2 of every token -> LDS (wave q summed the sets q and q + 4; wave 2 the leftover tokens as its "set 6")
    sumsq0 += __shfl_xor(sumsq0, 32, SN_WAVE);
    if (q == 2) { sumsq1 += __shfl_xor(sumsq1, 16, SN_WAVE); sumsq1 += __shfl_xor(sumsq1, 32, SN_WAVE); }
    else sumsq1 += __shfl_xor(sumsq1, 32, SN_WAVE);
    if (h == 0) tsum[q * 32 + r] = sumsq0;
    if (q < 2 && h == 0) tsum[(q + 4) * 32 + r] = sumsq1;
    if (q == 2 && lane < 16) tsum[6 * 32 + lane] = sumsq1;
    // ---- per set: the best key of a token over the eight lanes that hold it
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const bool lane_on = s < kS4Sets || lane < 32;           // (leftover: the lanes kg < 2, as r = j16 = lane & 15, h = kg = lane >> 4)
        const int rr = s < kS4Sets ? r : j16, hh = s < kS4Sets ? h : (kg & 1);
        const int row = 32 * s + rr;
        const bool valid = lane_on && (s < kS4Sets || rr < kS4Lo) && row < n_mine;
        const float other = __shfl_xor(m1[s], s < kS4Sets ? 32 : 16, SN_WAVE);
        const float kmin = fminf(m1[s], other);
        if (valid && hh == 0) __builtin_amdgcn_ds_fminf((__attribute__((address_space(3))) float *)&tbest[s * 32 + rr], kmin, 0, 0, false);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const unsigned *scal = reinterpret_cast<const unsigned *>(p.packed + lay.scal_off);
    const float C2 = __uint_as_float(scal[0]), C1 = __uint_as_float(scal[1]), CN = __uint_as_float(scal[2]), CMAX = __uint_as_float(scal[3]);
    unsigned hm[7];
    bool badv[7];
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const bool lane_on = s < kS4Sets || lane < 32;
        const int rr = s < kS4Sets ? r : j16, hh = s < kS4Sets ? h : (kg & 1);
        const int row = 32 * s + rr;
        const bool valid = lane_on && (s < kS4Sets || rr < kS4Lo) && row < n_mine;
        // per-token error window (DESIGN.md "S1 error window"): |key - exact value| <= E
        const float sumsq = tsum[s * 32 + rr];
        const float X2 = sqrtf(sumsq) * 1.001f + 1.0e-6f, X1 = X2 * sqrtf((float)p.D);
        const float vmax = 0.5f * CN + X2 * C2;                               // >= |any partial sum|
        const float E = 1.01f * (2.01f * kU16 * X2 * C2 + 5.96e-8f * (X1 + C1)
                                 + (float)(2 * NCH) * kAccUlpPerMfma * vmax + vmax * (3.0f * 5.96e-8f + 3.1e-5f));
        const bool bad = !(sumsq <= kHugeIn * kHugeIn) || !(CMAX <= kHugeIn) || !(vmax < 1.0e28f);   // NaN-safe; |x|_2 <= 3e4 bounds every component
        const float best = tbest[s * 32 + rr];
        const bool any_finite = best < kKeyLimit;
        const float cut = best + 2.0f * E;
        const unsigned hmask = (m1[s] <= cut ? 1u : 0u) | (m2[s] <= cut ? 2u : 0u) | (m3[s] <= cut ? 4u : 0u);
        const bool hover = m3[s] <= cut;                                      // a 4th may hide behind it
        const unsigned contrib = (bad || !any_finite) ? 0x80000000u : ((hmask << (3 * (2 * q + hh))) | (hover ? 0x80000000u : 0u));
        if (valid && contrib) __hip_atomic_fetch_or(&tmask[s * 32 + rr], contrib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        hm[s] = hmask; badv[s] = bad;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < 7; ++s) {
        const bool lane_on = s < kS4Sets || lane < 32;
        const int rr = s < kS4Sets ? r : j16, hh = s < kS4Sets ? h : (kg & 1);
        const int row = 32 * s + rr;
        const bool valid = lane_on && (s < kS4Sets || rr < kS4Lo) && row < n_mine;
        const int64_t n = tok0 + row;
        const unsigned mk = tmask[s * 32 + rr];
        const unsigned cand = mk & 0xFFFFFFu;
        const bool overflow = (mk >> 31) != 0u || cand == 0u;
        const int nc = __popc(cand);
        if (valid && !overflow) {
            if (nc == 1 && hm[s] != 0u) {                                     // the only candidate: final
                const unsigned code = __float_as_uint(m1[s]) & 0xFFu;
                p.out[out_index(p, n)] = (q * 4 + (int)(code >> 4)) * 32 + 8 * (int)((code >> 2) & 3u) + 4 * hh + (int)(code & 3u);
            }
            if (nc > 1) {                                       // the eight lanes of the token write their three codes
                unsigned char *cd = p.codes + (int64_t)n * kCodeBytes + 3 * (2 * q + hh);
                cd[0] = (unsigned char)(__float_as_uint(m1[s]) & 0xFFu);
                cd[1] = (unsigned char)(__float_as_uint(m2[s]) & 0xFFu);
                cd[2] = (unsigned char)(__float_as_uint(m3[s]) & 0xFFu);
            }
        }
        const bool writer = valid && q == 0 && hh == 0;
        if (writer) p.flags[n] = overflow ? 0x80000000u : (nc > 1 ? cand : 0u);
        const bool need_b = writer && overflow;
        const unsigned long long mask_b = __ballot(need_b);
        if (mask_b) {                                          // rare: tokens the screen cannot bound (phase B of the re-rank writes out[])
            int base = 0;
            const int leader = __ffsll((long long)mask_b) - 1;
            if (lane == leader) base = atomicAdd(&p.work[1], __popcll(mask_b));
            base = __shfl(base, leader, SN_WAVE);
            if (need_b) p.overflow[base + __popcll(mask_b & ((1ull << lane) - 1ull))] = (int)n;
        }
    }
    stamp(p, 4, lane, wave_id);
    asm volatile("s_mov_b32 m0, %0" :: "s"(keep_m0));
}

template <int NT>
int launch_exact(const AssignArgs &a, hipStream_t st)
{
    const int64_t blocks = (a.n_tokens + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(assign_exact_kernel<NT>, dim3(grid), dim3(256), 0, st, a);
    return 0;
}

int device_cus() { return sn_device_cus(); }

// launch options of the token-stationary screen: [0] token-phase gate, [1] balanced token map.  -1 = take the
// environment variable (default off: both measured slower at the bench shape); sn_debug_set_assign_options
// overrides (A/B timing inside one process).
int g_assign_opt[2] = {-1, -1};
bool assign_dual()          // two accumulator chains per tile in the token-stationary screen (D = 384): SN_ASSIGN_DUAL
{
    static const int v = getenv("SN_ASSIGN_DUAL") ? atoi(getenv("SN_ASSIGN_DUAL")) : 0;
    return v != 0;
}
bool assign_option(int i, const char *env)
{
    if (g_assign_opt[i] < 0) { const char *e = getenv(env); g_assign_opt[i] = (e && atoi(e) != 0) ? 1 : 0; }
    return g_assign_opt[i] != 0;
}

// the kernel that finishes a screen's flagged / overflow tokens
template <int NT, int FMT>
void launch_rerank(const AssignArgs &a, hipStream_t st)
{
    const int64_t chunks = (a.n_tokens + 31) / 32;
    sn_prof_start(1, st);
    hipLaunchKernelGGL((assign_rerank_kernel<NT, FMT>), dim3(kOverflowBlocks + (unsigned)(chunks < 4096 ? chunks : 4096)), dim3(256), 0, st, a);
    sn_prof_stop(1, st);
}

template <int NSTEPS, int NW, int R, int CB = 8, bool DUAL = false>
int launch_screen(const AssignArgs &a, hipStream_t st, bool defer = false)
{
    size_t lds = (size_t)R * (NSTEPS + 1) * 1024;
    if (const char *pad = getenv("SN_ASSIGN_LDS_PAD")) lds += (size_t)atoi(pad);     // diagnostics: force 1 workgroup per CU
    if (int rc = sn_ensure_dynamic_lds((const void *)assign_screen_kernel<NSTEPS, NW, R, CB, DUAL>, lds, "sn_assign_words")) return rc;
    const int tok_per_block = kTokPerWave * NW;
    unsigned grid = (unsigned)((a.n_tokens + tok_per_block - 1) / tok_per_block);
    AssignArgs ag = a;
    ag.full_waves = NW;
    ag.extra_base = a.n_tokens;
    // Between one and two workgroups per CU (two fit): with 128 tokens each, some CUs get twice the bytes and
    // twice the matrix work of the others and the launch lasts as long as they do.  Deal the tokens evenly
    // instead: 2 x CUs workgroups of f full waves each, the remaining 32-token sets one per workgroup as wave f.
    const bool balance_on = assign_option(1, "SN_ASSIGN_BALANCE");
    const int cus = device_cus();
    if (balance_on && NSTEPS <= 24 && NW == 4 && (int)grid > cus && (int)grid < 2 * cus) {
        const int64_t G = 2 * (int64_t)cus;
        const int f = (int)(a.n_tokens / (kTokPerWave * G));                  // 2 or 3 here
        ag.full_waves = f;
        ag.extra_base = G * kTokPerWave * f;
        grid = (unsigned)G;
    }
    const bool gate_on = assign_option(0, "SN_ASSIGN_GATE");
    // the gate pays when two workgroups share a CU and there is more than one workgroup per CU to stagger
    if (!(gate_on && NSTEPS <= 24 && NW == 4 && (int)grid > cus)) ag.gate = nullptr;
    else if (int rc = sn_zero_async(ag.gate, kGateBytes, st)) return rc;
    sn_prof_start(0, st);
    hipLaunchKernelGGL((assign_screen_kernel<NSTEPS, NW, R, CB, DUAL>), dim3(grid), dim3(64 * NW), lds, st, ag);
    sn_prof_stop(0, st);
    constexpr int NT = NSTEPS / 4;
    if (!defer) launch_rerank<NT, (CB == 8 ? 0 : 2)>(ag, st);      // (deferred: the consumer of the ids finishes them, sn_assign_words mode 2)
    return 0;
}

template <int NT, int KS>
int launch_screen2(const AssignArgs &a, hipStream_t st)
{
    const size_t lds = (size_t)kS2RawSlots * (KS / 2) * 2048 + (size_t)2 * KS * 1024 + kS2SmallSlots * sizeof(S2Small) + (size_t)kS2StashSlots * 256 * 16 + (size_t)4 * NT * 128;
    if (int rc = sn_ensure_dynamic_lds((const void *)assign_screen2_kernel<NT, KS>, lds, "sn_assign_words")) return rc;
    const int cus = device_cus();
    const unsigned grid = (unsigned)(a.n_sets < cus ? a.n_sets : cus);     // one persistent workgroup per CU
    sn_prof_start(0, st);
    hipLaunchKernelGGL((assign_screen2_kernel<NT, KS>), dim3(grid), dim3(256), lds, st, a);
    sn_prof_stop(0, st);
    constexpr int NTR = KS / 4;                                             // fp64 re-rank: 64 k per lane-step
    launch_rerank<NTR, 1>(a, st);
    return 0;
}

template <int NTW, int NCH>
int launch_screen3(const AssignArgs &a, hipStream_t st)
{
    constexpr int NTR = (NCH + 1) / 2;                          // fp64 re-rank: 64 k per lane-step
    constexpr size_t lds = (size_t)kS3RingA * kS3Quarters * NTW * 2048 + (size_t)kS3Sets * kS3RingT * 4096 + (size_t)kS3Quarters * NTW * 128 + 1024;
    if (int rc = sn_ensure_dynamic_lds((const void *)assign_screen3_kernel<NTW, NCH>, lds, "sn_assign_words")) return rc;
    AssignArgs ag = a;
    const int64_t cus = device_cus();
    // full 32-token sets whenever there is more than one round of them (the loop is bound by the matrix pipe, which a
    // smaller set does not relieve); a single partial round is spread over all CUs with smaller sets
    int64_t tps = 32;
    if (a.n_tokens < cus * kS3Sets * 32) tps = (a.n_tokens + cus * kS3Sets - 1) / (cus * kS3Sets);
    if (const char *e = getenv("SN_ASSIGN_TPS")) tps = atoi(e);
    tps = tps < 16 ? 16 : (tps > 32 ? 32 : tps);
    ag.tps = (int)tps;
    ag.n_sets3 = (a.n_tokens + tps - 1) / tps;
    const unsigned grid = (unsigned)(ag.n_sets3 < cus ? ag.n_sets3 : cus);      // one persistent workgroup per CU; sets are dealt round-robin
    sn_prof_start(0, st);
    hipLaunchKernelGGL((assign_screen3_kernel<NTW, NCH>), dim3(grid), dim3(1024), lds, st, ag);
    sn_prof_stop(0, st);
    launch_rerank<NTR, 3>(ag, st);
    return 0;
}

template <int NCH>
int launch_screen4(const AssignArgs &a, hipStream_t st)
{
    constexpr int NTR = (NCH + 1) / 2;                          // fp64 re-rank: 64 k per lane-step
    if (int rc = sn_ensure_dynamic_lds((const void *)assign_screen4_kernel<NCH>, kS4Lds, "sn_assign_words")) return rc;
    AssignArgs ag = a;
    const int64_t cus = device_cus();
    int64_t tpw = (a.n_tokens + cus - 1) / cus;                 // one workgroup per CU, one round (the caller checked n_tokens <= cus kS4Rows)
    if (const char *e = getenv("SN_ASSIGN_TPW")) tpw = atoi(e);
    tpw = tpw < 1 ? 1 : (tpw > kS4Rows ? kS4Rows : tpw);
    ag.tps4 = (int)tpw;
    const unsigned grid = (unsigned)((a.n_tokens + tpw - 1) / tpw);
    sn_prof_start(0, st);
    hipLaunchKernelGGL((assign_screen4_kernel<NCH>), dim3(grid), dim3(256), kS4Lds, st, ag);
    sn_prof_stop(0, st);
    launch_rerank<NTR, 3>(ag, st);
    return 0;
}

// form of the screen kernel: 0 (default) = token-stationary, 4 waves x 3-slot codebook ring, two
// workgroups per CU; 1 = token-stationary, 8 waves x 5-slot ring; 2 = register-stationary codebook
// (assign_screen2_kernel; shapes with M <= 512, D in {192, 384}, n_inner >= 32; else falls back to 0);
// 3 = K-outer token stream (assign_screen3_kernel; 8 or 16 tiles of 32 words, i.e. 192 < M <= 256 or 448 < M <= 512,
// fp32 tokens; else falls back to 0).
// Initialised from SN_ASSIGN_VARIANT, changed with sn_assign_set_variant().
int g_variant = -1;
int screen_variant()
{
    if (g_variant < 0) { const char *e = getenv("SN_ASSIGN_VARIANT"); g_variant = e ? atoi(e) : 0; }
    return g_variant;
}

}  // namespace

// ============================================================================================
// C ABI
// ============================================================================================
extern "C" size_t sn_codebook_pack_bytes(int M, int D)
{
    if (M <= 0 || D <= 0 || D % 32 != 0 || D > 1024 || M > 65536) return 0;
    return pack_layout(M, D).total;
}

extern "C" int sn_codebook_prepare(const float *codebook, int M, int D, void *packed, void *stream)
{
    SN_REQUIRE(codebook && packed, SN_ERR_BAD_ARG, "sn_codebook_prepare: NULL pointer");
    SN_REQUIRE(M > 0 && M <= 65536, SN_ERR_BAD_ARG, "sn_codebook_prepare: M=%d out of range", M);
    SN_REQUIRE(D > 0 && D % 32 == 0 && D <= 1024, SN_ERR_UNSUPPORTED, "sn_codebook_prepare: D=%d must be a multiple of 32, <= 1024", D);
    SN_REQUIRE((reinterpret_cast<uintptr_t>(packed) & 255) == 0, SN_ERR_BAD_ARG, "sn_codebook_prepare: packed must be 256-byte aligned");
    const PackLayout lay = pack_layout(M, D);
    hipStream_t st = (hipStream_t)stream;
    unsigned char *base = (unsigned char *)packed;
    if (int rc = sn_zero_async(base + lay.scal_off, 256, st)) return rc;
    const int64_t elems = (int64_t)lay.m_pad * D;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)(elems / 256)), dim3(256), 0, st, codebook, M, D,
                       base + lay.tiles_off, lay.n_steps, lay.tile_bytes);
    if (lay.nt2)
        hipLaunchKernelGGL(pack_frag2_kernel, dim3((unsigned)((int64_t)128 * lay.nt2 * D / 256)), dim3(256), 0, st, codebook, M, D,
                           base + lay.frag2_off, (float *)(base + lay.hn2_off), lay.nt2, lay.ks2);
    hipLaunchKernelGGL(pack_norm_kernel, dim3((unsigned)((lay.m_pad + 3) / 4)), dim3(256), 0, st, codebook, M, D, lay.m_pad,
                       base + lay.tiles_off, lay.n_steps, lay.tile_bytes, (double *)(base + lay.cn64_off),
                       (unsigned *)(base + lay.scal_off), (float *)(base + lay.hn2_off), lay.nt2);
    SN_CHECK_LAUNCH("sn_codebook_prepare");
    return SN_OK;
}

extern "C" int sn_assign_variant(void) { return screen_variant(); }

// mode 2 leaves its flagged tokens to the consumer only on the default (token-stationary) screen with byte codes
extern "C" int sn_assign_defers(int M, int D)
{
    const int v = screen_variant();
    const bool dflt = v == 0 || (v == 2 && pack_layout(M, D).nt2 == 0) || (v == 3 && !(pack_layout(M, D).n_tiles == 8 || pack_layout(M, D).n_tiles == 16)) ||
                      (v == 4 && pack_layout(M, D).n_tiles != 16);      // (variant 4 with 16 tiles may still fall back for a large batch: it then takes the stand-alone finish)
    return (dflt && M > 0 && M <= 2048 && (D == 192 || D == 384)) ? 1 : 0;      // (D = 768: the finish's state - three 12-register rows - does not fit the graph kernel's 128 registers)
}

extern "C" int sn_assign_set_variant(int variant)
{
    SN_REQUIRE(variant >= 0 && variant <= 4, SN_ERR_BAD_ARG, "sn_assign_set_variant: variant=%d", variant);
    g_variant = variant;
    return SN_OK;
}

/* diagnostics: resident workgroups per CU the runtime reports for the screen kernel (D = 384) with
 * `lds` bytes of dynamic LDS */
extern "C" int sn_debug_screen_occupancy(int lds)
{
    int n = -1;
    if (lds > 64 * 1024) (void)hipFuncSetAttribute((const void *)assign_screen_kernel<24, 4, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)assign_screen_kernel<24, 4, 3>, 256, (size_t)lds) != hipSuccess) return -1;
    return n;
}

/* diagnostics: token-phase gate / balanced token map of the token-stationary screen on (1), off (0), from the environment (-1) */
extern "C" void sn_debug_set_assign_options(int gate, int balance) { g_assign_opt[0] = gate; g_assign_opt[1] = balance; }

/* diagnostics: device buffer of 16 x u64 per wave of the screen kernel (NULL = off) */
extern "C" void sn_debug_set_stamps(void *device_buffer) { g_stamps = (unsigned long long *)device_buffer; }

extern "C" size_t sn_assign_workspace_bytes(int64_t n_tokens)
{
    if (n_tokens < 0) return 0;
    // header + flag words + candidate codes + overflow token ids (the larger record format) + per-CU gate table
    return ((32 + (size_t)n_tokens * kWsPerToken2 + 15) & ~size_t(15)) + kGateBytes;
}

static int assign_words_impl(const void *x_any, int x_bf16, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer,
                             int64_t x_stride_inner, const float *codebook, const void *packed, int M, int D,
                             int64_t *out, int64_t out_stride_outer, int64_t out_stride_inner,
                             void *workspace, size_t workspace_bytes, int mode, void *stream)
{
    const float *x = (const float *)x_any;
    SN_REQUIRE(n_outer >= 0 && n_inner >= 0, SN_ERR_BAD_ARG, "sn_assign_words: negative token grid");
    const int64_t n_tokens = n_outer * n_inner;
    if (n_tokens == 0) return SN_OK;
    SN_REQUIRE(x && codebook && packed && out, SN_ERR_BAD_ARG, "sn_assign_words: NULL pointer");
    SN_REQUIRE(M > 0 && M <= 65536, SN_ERR_BAD_ARG, "sn_assign_words: M=%d out of range", M);
    SN_REQUIRE(D > 0 && D % 32 == 0 && D <= 1024, SN_ERR_UNSUPPORTED, "sn_assign_words: D=%d must be a multiple of 32, <= 1024", D);
    SN_REQUIRE(n_tokens < 0x7fffffff, SN_ERR_UNSUPPORTED, "sn_assign_words: too many tokens");
    SN_REQUIRE(mode >= 0 && mode <= 3, SN_ERR_BAD_ARG, "sn_assign_words: mode=%d", mode);
    // mode 3 = only the re-rank of an earlier mode-2 call with the same arguments (a consumer that could not take the
    // deferred finish after all): the stand-alone kernels on the records of the workspace.
    const bool finish_only = mode == 3;
    // mode 2 = mode 0 with the re-rank left to the consumer of the ids (sn_instance_graph with `rerank` set): the screen
    // runs, `out` holds its words (final wherever the flag word is 0), the flag words and candidate records stay in the
    // workspace.  Where the deferred form does not apply (screen forms other than the default, codebooks of more than
    // 2048 words, shapes the screen is not built for) the call does everything itself and clears the flag words.
    const bool want_defer = mode == 2;
    if (want_defer || finish_only) {
        mode = 0;
        SN_REQUIRE(workspace && workspace_bytes >= sn_assign_workspace_bytes(n_tokens), SN_ERR_WORKSPACE,
                   "sn_assign_words: mode 2 needs the workspace (%zu < %zu bytes)", workspace_bytes, sn_assign_workspace_bytes(n_tokens));
    }
    bool deferred = false;
    AssignArgs a;
    a.x = x; a.n_tokens = n_tokens; a.n_inner = n_inner; a.xso = x_stride_outer; a.xsi = x_stride_inner;
    a.cb = codebook; a.packed = (const unsigned char *)packed; a.M = M; a.D = D;
    a.out = out; a.oso = out_stride_outer; a.osi = out_stride_inner; a.work = (int *)workspace;
    unsigned char *ws = (unsigned char *)workspace;
    a.flags = ws ? (unsigned *)(ws + 32) : nullptr;
    a.codes = ws ? ws + 32 + (size_t)n_tokens * 4 : nullptr;
    a.overflow = ws ? (int *)(ws + 32 + (size_t)n_tokens * (4 + kCodeBytes)) : nullptr;
    a.stamps = g_stamps;
    a.codes32 = ws ? (unsigned *)(ws + 32 + (size_t)n_tokens * 4) : nullptr;
    a.n_sets = (n_tokens + 31) / 32;
    a.full_waves = kWavesPerBlock; a.extra_base = n_tokens;
    a.x_bf16 = x_bf16;
    a.tps = 32; a.n_sets3 = 0; a.tps4 = 0;
    a.gate = ws ? (unsigned *)(ws + ((32 + (size_t)n_tokens * kWsPerToken2 + 15) & ~size_t(15))) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int per16 = x_bf16 ? 8 : 4;                            // elements per 16 bytes
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && x_stride_outer % per16 == 0 && x_stride_inner % per16 == 0;
    const bool screen_ok = mode == 0 && aligned && (SN_S1_STAGE || !x_bf16) && M <= 32 * kMaxTilesScreen && (D == 192 || D == 384 || D == 768);
    if (finish_only) {
        // (where mode 2 did not defer it has cleared the flag words and the overflow count: the kernels find nothing)
        if (D == 192) launch_rerank<3, 0>(a, st);
        else if (D == 384) launch_rerank<6, 0>(a, st);
        else SN_REQUIRE(false, SN_ERR_UNSUPPORTED, "sn_assign_words: mode 3 for D=%d", D);
        SN_CHECK_LAUNCH("sn_assign_words");
        return SN_OK;
    }
    if (screen_ok) {
        SN_REQUIRE(workspace && workspace_bytes >= sn_assign_workspace_bytes(n_tokens), SN_ERR_WORKSPACE,
                   "sn_assign_words: workspace %zu < %zu bytes", workspace_bytes, sn_assign_workspace_bytes(n_tokens));
        if (int rc0 = sn_zero_async(workspace, 32, st)) return rc0;
        int rc = 0;
        const bool wide = screen_variant() == 1;
        const PackLayout lay = pack_layout(M, D);
        // (K-outer form: fp32 tokens whose byte offsets fit 32 bits: the copies address them as base + 32-bit lane offset)
        if (screen_variant() == 3 && !x_bf16 && (lay.n_tiles == 8 || lay.n_tiles == 16) &&
            ((n_outer - 1) * x_stride_outer + (n_inner - 1) * x_stride_inner + D) * 4 < (int64_t)0xFFFFF000ll && x_stride_outer >= 0 && x_stride_inner >= 0) {
            if (lay.n_tiles == 16) rc = D == 192 ? launch_screen3<4, 6>(a, st) : (D == 384 ? launch_screen3<4, 12>(a, st) : launch_screen3<4, 24>(a, st));
            else rc = D == 192 ? launch_screen3<2, 6>(a, st) : (D == 384 ? launch_screen3<2, 12>(a, st) : launch_screen3<2, 24>(a, st));
        } else if (screen_variant() == 4 && !x_bf16 && lay.n_tiles == 16 && (D == 192 || D == 384) && n_tokens <= (int64_t)device_cus() * kS4Rows &&
                   ((n_outer - 1) * x_stride_outer + (n_inner - 1) * x_stride_inner + D) * 4 < (int64_t)0xFFFFF000ll && x_stride_outer >= 0 && x_stride_inner >= 0) {
            rc = D == 192 ? launch_screen4<6>(a, st) : launch_screen4<12>(a, st);
        } else if (screen_variant() == 2 && lay.nt2 != 0 && n_inner >= 32 && !x_bf16) {
            a.overflow = (int *)(ws + 32 + (size_t)n_tokens * (4 + 32));
            if (lay.ks2 == 24) rc = lay.nt2 == 4 ? launch_screen2<4, 24>(a, st) : launch_screen2<2, 24>(a, st);
            else rc = lay.nt2 == 4 ? launch_screen2<4, 12>(a, st) : launch_screen2<2, 12>(a, st);
        } else if (M > 2048) {               // more than 64 tiles: 10-bit word codes in the keys, 16-bit codes in the records
            a.overflow = (int *)(ws + 32 + (size_t)n_tokens * (4 + kCodeBytesWide));
            if (D == 192) rc = launch_screen<12, 4, 3, 10>(a, st);
            else if (D == 384) rc = launch_screen<24, 4, 3, 10>(a, st);
            else rc = launch_screen<48, 4, 3, 10>(a, st);
        } else if (D == 192) rc = wide ? launch_screen<12, 8, 5>(a, st) : launch_screen<12, 4, 3>(a, st, deferred = want_defer);
        else if (D == 384) rc = wide ? launch_screen<24, 8, 5>(a, st) : (assign_dual() ? launch_screen<24, 4, 3, 8, true>(a, st) : launch_screen<24, 4, 3>(a, st, deferred = want_defer));
        else rc = launch_screen<48, 4, 3>(a, st);
        if (rc) return rc;
    } else {
        const int nt = (D + 63) / 64;
        if (nt <= 3) launch_exact<3>(a, st);
        else if (nt <= 6) launch_exact<6>(a, st);
        else if (nt <= 12) launch_exact<12>(a, st);
        else launch_exact<16>(a, st);
    }
    if (want_defer && !deferred) {                      // everything is final: a consumer must find no flag
        if (int rc0 = sn_zero_async(a.flags, (size_t)n_tokens * 4, st)) return rc0;
    }
    SN_CHECK_LAUNCH("sn_assign_words");
    return SN_OK;
}

extern "C" int sn_assign_words(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer,
                               int64_t x_stride_inner, const float *codebook, const void *packed, int M, int D,
                               int64_t *out, int64_t out_stride_outer, int64_t out_stride_inner,
                               void *workspace, size_t workspace_bytes, int mode, void *stream)
{
    return assign_words_impl(x, 0, n_outer, n_inner, x_stride_outer, x_stride_inner, codebook, packed, M, D, out, out_stride_outer,
                             out_stride_inner, workspace, workspace_bytes, mode, stream);
}

extern "C" int sn_assign_words_bf16(const void *x_bf16, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer,
                                    int64_t x_stride_inner, const float *codebook, const void *packed, int M, int D,
                                    int64_t *out, int64_t out_stride_outer, int64_t out_stride_inner,
                                    void *workspace, size_t workspace_bytes, int mode, void *stream)
{
    return assign_words_impl(x_bf16, 1, n_outer, n_inner, x_stride_outer, x_stride_inner, codebook, packed, M, D, out, out_stride_outer,
                             out_stride_inner, workspace, workspace_bytes, mode, stream);
}
