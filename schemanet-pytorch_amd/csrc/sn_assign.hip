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
#ifndef SN_S1_SADDR
#define SN_S1_SADDR 1       // codebook ring copies addressed by an SGPR base + 32-bit lane offset (0: a 64-bit address per lane, rounds 1-5)
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
[[maybe_unused]] constexpr float kU16 = 4.8828125e-4f;  // 2^-11, fp16 unit round-off (SN_S1_EXACT_LOSS 0)
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
//   tiles5  [4 q][4 v][n_steps][1 KiB]    (one-round K-outer screen, assign_screen5_kernel; codebooks of 16 tiles only)
//                                           the same fragments with the WORDS permuted: row i of virtual tile (q, v) is word
//                                           32 (4 v + (i >> 3)) + 8 q + (i & 7), so that the lane (r, h) of the wave that owns
//                                           quarter q holds exactly the words 32 t + 8 q + 4 h + e, t < 16: candidate slot
//                                           group (h, g = q) of the record format below
constexpr float kPadHalfNorm = 1.0e30f;     // |c|^2/2 of padding words where keys are compared as floats (finite: keys stay ordered)

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

// tiles5 image (see above): one thread per (word of the padded codebook, k); only for 16-tile codebooks
__global__ __launch_bounds__(256) void pack_frag5_kernel(const float *cb, int M, int D, unsigned char *tiles5, int n_steps)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over [512, D]
    const int m = (int)(idx / D), k = (int)(idx % D);
    const int t = m >> 5, row = m & 31;                           // word m = 32 t + 8 g + 4 hh + e
    const int q = row >> 3, v = t >> 2, i = 8 * (t & 3) + (row & 7);
    const int u = k >> 5, rem = k & 31, h = rem >> 4, e = (rem >> 3) & 1, j = rem & 7;     // (as pack_frag_kernel)
    const int s = 2 * u + e;
    const float val = m < M ? cb[(int64_t)m * D + k] : 0.0f;
    _Float16 *frag = (_Float16 *)(tiles5 + ((size_t)(q * 4 + v) * n_steps + s) * 1024);
    frag[(i + 32 * h) * 8 + j] = (_Float16)(-val);
}

__global__ __launch_bounds__(256) void pack_norm_kernel(const float *cb, int M, int D, int m_pad, unsigned char *tiles,
                                                        int n_steps, int tile_bytes, double *cn64, unsigned *scal)
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
    // token-phase gate of the token-stationary screen (NULL = off): per CU {arrivals, waves whose tokens have
    // landed}; zeroed before every launch.  See assign_screen_kernel.
    unsigned *gate;
    // token -> wave map of the token-stationary screen: waves 0 .. full_waves-1 of workgroup b own the tokens
    // [32 full_waves b + 32 w, +32); wave `full_waves` (if the workgroup has one) owns [extra_base + 32 b, +32);
    // waves without tokens only keep the codebook ring going.  Default: full_waves = waves per workgroup.
    int full_waves;
    int64_t extra_base;
    int tps4;           // one-round K-outer screen (assign_screen5_kernel): tokens per workgroup (<= kS5Rows)
    int dbg;            // diagnostics (SN_ASSIGN_DBG; results are WRONG with any bit set): bit 0 = every workgroup of the K-outer screen reads the first workgroup's tokens (no HBM stream)
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

// FMT 0: records of assign_screen_kernel / assign_screen5_kernel (24 code bytes per token); FMT 2: 16-bit codes (M > 2048);

template <int NT, int FMT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(NT > 6 ? 4 : 6, 8))) void assign_rerank_kernel(const AssignArgs p)
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
#if SN_S1_SADDR
    // (round 6) wave-uniform source base in SGPRs + a 32-bit lane offset: with a 64-bit address per lane the CU's address unit took
    // ~37 cycles per 1 KiB copy (the "28 B per cycle and CU" of this ring; measured on the GCN product's rings, 8d)
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_tile = [&](int w, int slot) {
        const unsigned char *sbase = tiles + (size_t)w * kTileBytes + dma_c0 * 1024;
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * kTileBytes + dma_c0 * 1024);
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0" : "=s"(keep));
#pragma unroll
        for (int j = 0; j < kDmaMin; j += 4) {
            const unsigned char *sj = sbase + (size_t)j * 1024;
            const unsigned dj = dst + j * 1024;
            if (kDmaMin - j >= 4)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, %2 offset:2048\n\tglobal_load_lds_dwordx4 %0, %2 offset:3072" :: "v"(lane16), "s"(dj), "s"(sj) : "memory");
            else if (kDmaMin - j == 3)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024\n\t"
                             "global_load_lds_dwordx4 %0, %2 offset:2048" :: "v"(lane16), "s"(dj), "s"(sj) : "memory");
            else if (kDmaMin - j == 2)
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\tglobal_load_lds_dwordx4 %0, %2 offset:1024" :: "v"(lane16), "s"(dj), "s"(sj) : "memory");
            else
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(lane16), "s"(dj), "s"(sj) : "memory");
        }
        if (kDmaExtra != 0 && wid < kDmaExtra)                                    // wave-uniform
            asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2" :: "v"(lane16), "s"(dst + kDmaMin * 1024), "s"(sbase + (size_t)kDmaMin * 1024) : "memory");
        asm volatile("s_mov_b32 m0, %0" :: "s"(keep));
    };
#else
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
#endif
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

constexpr float kBigKey = 3.0e38f;      // "no key yet"
constexpr float kKeyLimit = 1.0e29f;    // above this a best value is a padding word / nothing finite

// compile-time loop: f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>) as straight-line code
// (the chunk loops of the K-outer screens are far beyond the size a `#pragma unroll` is allowed to expand)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F &f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// ------------------------------------------------------------------------------------------
// mode 0, pass 1, one-round K-outer form (round 5; opt-in, sn_assign_set_variant(5); codebooks of 16 tiles of 32 words - 448 < M <= 512 -,
// D in {192, 384}, fp32 tokens, at most kS5Rows tokens per CU).
//
// The accumulators of EVERY token of the workgroup against the whole codebook stay in registers for the length of the launch and
// K is the outer loop, so the token stream (HBM) runs under the matrix pipe and the codebook crosses the CU's memory path once.
// One workgroup of eight 256-register waves per CU: wave (sg, q) = (tid >> 8, (tid >> 6) & 3) holds the accumulators of the three
// 32-token sets 3 sg .. 3 sg + 2 against the quarter q of the words (4 tiles x 16 registers x 3 sets = 192) and of up to sixteen
// further tokens ("leftover", v_mfma_f32_16x16x32_f16; 196 tokens per CU = 50 176 / 256 is 6 x 32 + 4) against half of that quarter
// (4 blocks x 4 registers): 208 of its 256 registers, every one an architectural VGPR, so the MFMAs are plain builtins (the compiler
// sees their hazards and their operands' lifetimes) and two waves share a SIMD.  (Round 4's form of the same idea ran four
// 512-register waves with asm MFMAs on AGPRs: 34.5 us - a lone wave issues its 2 300 key instructions at one per 8 cycles.)
// The quarter of a wave is not a run of four tiles but an accumulator-row group: the tiles5 image (pack_frag5_kernel) permutes the
// words so that lane (r, h) of wave (sg, q) holds, of its token, exactly the words 32 t + 8 q + 4 h + e (t < 16, e < 4) - the
// candidate-slot group (h, g = q) of the DEFAULT screen's record (sn_assign_shared.h): a sorted triple per lane and set IS that
// group's triple, so this kernel writes the same flag words and 24-byte records as assign_screen_kernel and both finishers
// (assign_rerank_kernel<., 0>, the deferred finish inside the instance-graph kernel) take them as they are.
// The leftover tokens' words are split by accumulator-row half: wave (sg, q) multiplies them by the words 32 t + 8 q + 4 sg + e
// - slot group (h = sg, g = q) again, spread over the four lanes (token, kg) that are merged with two shuffles.
// Values u[word] = |c|^2/2 - x~.c~ (accumulators start at |c|^2/2, no per-token shift: |x|^2 is only known at the end), keys = float
// bits with the low byte replaced by the word code, compared as floats; window from the token's measured fp16 rounding loss.
//
// Measured (MI355X, 50 176 tokens, rocprofv3 kernel trace): 31.2 us against 33.7 us for assign_screen_kernel in the same process
// (32.6 / 35.2 by the library's event pair), same ids, 6.4 % of the tokens flagged (6.6 %).  In the replayed bench (four steps in
// flight) the step is 1.5 % SLOWER with it: it owns every CU's registers and LDS for the length of the launch, and the kernels of
// the other steps cannot fill in - hence opt-in.  Where its 59 k cycles go (in-kernel stamps, tools/diag_s5.py): prologue to first
// MFMA 6 k, the eleven steady chunks 32.4 k = 2 950 per chunk against 1 790 of matrix pipe and 2 070 of HBM (ablation builds: the
// LDS-read skeleton alone 1 080 per chunk, + the fp32 -> fp16 conversion 2 010, + the MFMAs 2 750: an in-order wave with ONE partner
// on its SIMD overlaps the three only partly; two steps of lead for the raw rows would need eight registers the waves do not have),
// last chunk 2 k, keys 4.6 - 7.7 k (832 VALU instructions per wave), merge + records 5 k.
// ------------------------------------------------------------------------------------------
constexpr int kS5Sets = 6, kS5Lo = 16;
constexpr int kS5Rows = kS5Sets * 32 + kS5Lo;                   // tokens a workgroup can hold (208)
constexpr int kS5RowsPad = 208;                                 // token rows copied per chunk: 26 copies of 8 rows
constexpr int kS5Slab = 16 * 2048, kS5Tok = kS5RowsPad * 128;   // bytes per chunk: codebook (16 virtual tiles x 2 k-steps), token rows
constexpr int kS5OffT = 2 * kS5Slab;
constexpr int kS5OffHc = kS5OffT + 3 * kS5Tok;                  // |c|^2/2, [tile][accumulator-row order]
constexpr int kS5OffBest = kS5OffHc + 16 * 128;                 // [7][32] best key of a token
constexpr int kS5OffMask = kS5OffBest + 7 * 128;                // [7][32] candidate mask being assembled
constexpr int kS5OffSum = kS5OffMask + 7 * 128;                 // [7][32] window (2 E) of a token, NaN: not screened
constexpr int kS5OffTv = kS5OffSum + 7 * 128;                  // [4 copies][8 waves][64 lanes] lane addresses of a wave's token copies (no registers across the loop)
constexpr int kS5Lds = kS5OffTv + 4 * 8 * 256;

// one 1 KiB LDS-DMA piece: global address = scalar base + 32-bit lane offset + OFF, LDS address = lds_dst + 16 x lane (the
// instruction's offset field moves both, so M0 carries the destination minus OFF).  A function, not a statement inside the
// kernel's generic lambdas: clang does not count an asm operand there as a use of a captured variable.
template <int OFF, bool NT = false>
__device__ __forceinline__ void s5_dma(unsigned voff, const void *sbase, unsigned lds_dst)
{
    static_assert(OFF >= 0 && OFF < 4096, "instruction offset");
    if constexpr (NT) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3 nt" :: "v"(voff), "s"(sbase), "s"(lds_dst - (unsigned)OFF), "n"(OFF) : "memory");
    else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:%3" :: "v"(voff), "s"(sbase), "s"(lds_dst - (unsigned)OFF), "n"(OFF) : "memory");
}

// ABL (diagnostic builds, SN_S5_ABL; results are wrong): bit 0 = no squares / conversion, bit 1 = no 32 x 32 MFMAs, bit 2 = no copies behind chunk 0
template <int NCH, int ABL = 0>
__global__ __launch_bounds__(512, 2) void assign_screen5_kernel(const AssignArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float *hcs = reinterpret_cast<float *>(smem + kS5OffHc);
    float *tbest = reinterpret_cast<float *>(smem + kS5OffBest);
    unsigned *tmask = reinterpret_cast<unsigned *>(smem + kS5OffMask);
    float *tsum = reinterpret_cast<float *>(smem + kS5OffSum);
    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int sg = w >> 2, q = w & 3;
    const int r = lane & 31, h = lane >> 5;                      // 32 x 32 side: token column, accumulator row half = k half of a fragment
    const int kg = lane >> 4;                                    // 16 x 16 side: lane = (token column j16 = lane & 15, k group of a fragment = accumulator row group kg)
    const PackLayout lay = pack_layout(p.M, p.D);
    const unsigned char *tiles = p.packed + lay.tiles_off;
    const unsigned char *tiles5 = p.packed + lay.tiles5_off;
    const int64_t tok0 = (int64_t)blockIdx.x * p.tps4;           // this workgroup's tokens: [tok0, tok0 + n_mine)
    const int n_mine = (int)(p.n_tokens - tok0 < p.tps4 ? p.n_tokens - tok0 : p.tps4);
    unsigned keep_m0;
    asm volatile("s_mov_b32 %0, m0" : "=s"(keep_m0));
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const int wave_id = blockIdx.x * 8 + w;
    stamp(p, 0, lane, wave_id);

    if (tid < 7 * 32) { tbest[tid] = kBigKey; tmask[tid] = 0u; tsum[tid] = 0.0f; }
    // |c|^2/2 of every word, [tile][accumulator-row order] as in the image: two copies of eight tiles' 128 bytes (waves 0, 1),
    // the first thing in flight; whoever reads them does so behind the first chunk's barrier
    if (w < 2)
        s5_dma<0>((unsigned)((8 * w + (lane >> 3)) * lay.tile_bytes + lay.n_steps * 1024 + (lane & 7) * 16), tiles, __builtin_amdgcn_readfirstlane(lds_base + kS5OffHc + w * 1024));

    // ---- copies.  Token rows: copy i of wave w covers the rows 8 (w + 8 i) .. + 7 of the workgroup, i < 4 (w < 2) or 3 (a row
    // past its tokens: its first token - nobody reads it); lane -> (row, piece slot), the slot holds piece slot ^ ((row >> 1) & 7).
    unsigned *tvl = reinterpret_cast<unsigned *>(smem + kS5OffTv) + w * 64 + lane;      // copy i: tvl[512 i] (written and read by this lane only)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = 8 * (w + 8 * i) + (lane >> 3);
        const int64_t n = ((p.dbg & 1) ? 0 : tok0) + (row < n_mine ? row : 0);
        const unsigned ni = (unsigned)p.n_inner, o = (unsigned)n / ni, ii = (unsigned)n - o * ni;
        const unsigned tvi = (unsigned)(((int64_t)o * p.xso + (int64_t)ii * p.xsi) * 4 + 16 * ((lane & 7) ^ ((row >> 1) & 7)));
        tvl[512 * i] = tvi;
    }
    // (the chunk's 128 c bytes ride in the instruction's offset field, which moves the LDS address along with the global one:
    // M0 = slot address - 128 c; no address arithmetic on the vector side)
    auto issue_tok = [&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        const unsigned slot = (unsigned)(c % 3);
        unsigned tv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) tv[i] = (i < 3 || w < 2) ? tvl[512 * i] : 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i == 3 && w >= 2) break;                         // (wave-uniform)
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + kS5OffT + slot * kS5Tok + (w + 8 * i) * 1024);
            s5_dma<c * 128, SN_S1_TOKENS_NT != 0>(tv[i], p.x, dst);     // (read once: the non-temporal hint keeps the codebook and the finish's operands in L2)
        }
    };
    const unsigned a_lane = (unsigned)(q * 4 * 2048 + lane * 16);   // A fragment of virtual tile (q, v), k-step ks: + v * 2048 + ks * 1024
    // codebook: wave w copies the two k-steps of the virtual tiles 2 w, 2 w + 1 (4 KiB of the chunk's 32): scalar base, lane x 16
    auto issue_slab = [&](int c) {
        const unsigned slot = (unsigned)(c & 1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int vt = 2 * w + (i >> 1);
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * kS5Slab + vt * 2048 + (i & 1) * 1024);
            // (lane part = a_lane = 8192 q + 16 lane, the register the fragment reads use: the scalar base takes the - 8192 q)
            const unsigned char *src = tiles5 + (ptrdiff_t)__builtin_amdgcn_readfirstlane((vt * lay.n_steps + 2 * c + (i & 1)) * 1024 - q * 8192);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(a_lane), "s"(src), "s"(dst) : "memory");
        }
    };
    // issue order: slab(0), tok(0), slab(1), tok(1), tok(2); behind barrier c (which publishes chunk c + 1): slab(c + 2), tok(c + 3)
    static_assert(NCH >= 3, "ring depths");
    issue_slab(0); issue_tok(std::integral_constant<int, 0>{});
    issue_slab(1); issue_tok(std::integral_constant<int, 1>{});
    issue_tok(std::integral_constant<int, 2>{});
    stamp(p, 1, lane, wave_id);

    f32x16 acc[3][4];
    f32x4 accl[4];
    // |x|^2 and |x - fp16(x)|^2 of a token (the measured rounding loss replaces the worst-case 2^-11 |x| in the window: half as many
    // tokens to re-rank) are summed by ONE of the four quarter waves that convert its rows: wave (sg, q < 3) for set 3 sg + q, wave 3
    // for the leftover tokens.  (As LDS float atomics - no registers across the loop - the loop took 2.8 x as long.)
    auto to_half8 = [](const f32x4 &lo, const f32x4 &hi) {       // (pairs: v_cvt_pk_f16_f32, round to nearest)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        typedef _Float16 half2s __attribute__((ext_vector_type(2)));
        const half2s p0 = __builtin_convertvector(f32x2{lo.x, lo.y}, half2s), p1 = __builtin_convertvector(f32x2{lo.z, lo.w}, half2s);
        const half2s p2 = __builtin_convertvector(f32x2{hi.x, hi.y}, half2s), p3 = __builtin_convertvector(f32x2{hi.z, hi.w}, half2s);
        half8 b;
        b[0] = p0.x; b[1] = p0.y; b[2] = p1.x; b[3] = p1.y; b[4] = p2.x; b[5] = p2.y; b[6] = p3.x; b[7] = p3.y;
        return b;
    };
    float sumsq = 0.0f, sumd = 0.0f;
    // the owner's form of to_half8: the same fragment (v_cvt_pk_f16_f32), and the two sums; the loss from the converted halves with a
    // mixed-precision fma each (v_fma_mix_f32: x - fp16(x), exact) - 28 VALU instructions against 4 for the other three waves
    auto tally_half8 = [&](const f32x4 &lo, const f32x4 &hi) {
        const half8 b = to_half8(lo, hi);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            sumsq = fmaf(v[e], v[e], sumsq);
            const float dd = fmaf((float)b[e], -1.0f, v[e]);
            sumd = fmaf(dd, dd, sumd);
        }
        asm volatile("" : "+v"(sumsq), "+v"(sumd));              // (pinned: left alone the chains sink to the epilogue, every raw row spilled until then)
        return b;
    };
    const int swz = (r >> 1) & 7;                                // (row 32 s + r: the set offset does not touch bits 1..3)
    // lane parts of the LDS addresses.  B fragment of k-step ks: k half h of the image = floats 16 h + 8 ks .. + 7 of the chunk
    // (pack_frag_kernel) = pieces 4 h + 2 ks, + 1 of the token's line
    // piece slot of (k-step ks, 16-byte half b) = (4 h + 2 ks + b) ^ swz = (4 h ^ swz) ^ (2 ks + b): ONE lane register, the rest an
    // xor with a constant folded into the address add (v_xad_u32)
    const unsigned t_base = (unsigned)((96 * sg + r) * 128 + (((4 * h) ^ swz) << 4));
    // (through an opaque copy at every use: left alone the compiler keeps all four t_base ^ const alive across the whole loop)
    auto t_addr = [&](unsigned rb_, int ks, int b) {
        unsigned tb = t_base;
        asm volatile("" : "+v"(tb));
        return (tb ^ (unsigned)((2 * ks + b) << 4)) + rb_;
    };
    // (the leftover step's two lane addresses are formed again in every chunk, from a lane register the loop keeps anyway: a dozen
    // VALU instructions per chunk against two registers the loop does not have)
    auto leftover_addr = [&](unsigned &l_lo, unsigned &a16_lane) {
        unsigned al = a_lane;                                     // (= 8192 q + 16 lane, alive anyway; the empty asm keeps the arithmetic below inside its chunk)
        asm volatile("" : "+v"(al));
        const int ln = (int)((al >> 4) & 63u);
        const int j16_ = ln & 15, kg_ = ln >> 4;
        const int pl = 4 * (kg_ & 1) + 2 * (kg_ >> 1);
        l_lo = (unsigned)((32 * kS5Sets + j16_) * 128 + ((pl ^ ((j16_ >> 1) & 7)) << 4));      // (pl is even: the second piece is this ^ 16)
        // A fragment of leftover block b: row j16 of the block = virtual tile (q, b), row 8 (j16 >> 2) + 4 sg + (j16 & 3); k = 8 kg .. + 7:
        // k-step kg >> 1, k half kg & 1 of the image
        a16_lane = (unsigned)(q * 4 * 2048 + (kg_ >> 1) * 1024 + ((8 * (j16_ >> 2) + 4 * sg + (j16_ & 3)) + 32 * (kg_ & 1)) * 16);
    };
    unsigned long long tq[6] = {0, 0, 0, 0, 0, 0};               // diagnostics: s_memtime inside chunk 5 (scalar registers; written at the end)
    // ---- main loop, fully unrolled.  The work of a chunk is seven steps: S0 .. S5 = (k-step ks, set st) - four 32 x 32 x 16 MFMAs, the
    // wave's four virtual tiles against one B fragment - and L = the leftover tokens' four 16 x 16 x 32, in the order
    //     S0 S1 S2 L S3 S4 | barrier c | S5
    // as ONE software-pipelined stream over all chunks: the NEXT step's raw rows are requested in front of a step's MFMAs and squared /
    // converted behind them (while the partner wave's MFMAs have the pipe), the A fragments of the next k-step / of the leftover
    // blocks / of the next chunk take the registers of the current ones as each is used for the last time.
    // The barrier does not open a chunk, it sits INSIDE one (stamps of the form with "wait, barrier, load, convert, multiply" per
    // chunk: ~1 000 of its 2 700 cycles were the pipe waiting for the first LDS rows behind the barrier, on both waves of a SIMD at
    // once).  Barrier c publishes chunk c + 1 - every wave has waited for its copies of it - and certifies that every LDS read of
    // chunk c has been made (S5's operands are in registers), so the copies of slab(c + 2) / tok(c + 3) go into chunk c's slots
    // right behind it, and S5 requests the first rows and fragments of chunk c + 1 with MFMAs on both sides of the barrier.
    // The squares of a row are summed by ONE of the four quarter waves of its set (wave-uniform branches).
    half8 a[4], bcur;
    unsigned a16_lane = 0;
    // one step: kind 0 .. 2 = set, 3 = leftover; reload: 0 none, else a[v] <- the fragment at reload_base + 2048 v behind MFMA v
    auto run_step = [&](auto kind_c, auto reload_c, auto next_c, unsigned nlo, unsigned nhi, bool nsq, int nset, unsigned reload_base) {
        constexpr int kind = decltype(kind_c)::value;
        constexpr bool reload = decltype(reload_c)::value != 0, has_next = decltype(next_c)::value != 0;
        f32x4 lo, hi;
        if constexpr (has_next) {
            lo = *reinterpret_cast<const f32x4 *>(smem + nlo);
            hi = *reinterpret_cast<const f32x4 *>(smem + nhi);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            if constexpr (kind == 3) accl[v] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[v], bcur, accl[v], 0, 0, 0);
            else if constexpr (ABL & 2) asm volatile("" :: "v"(a[v]), "v"(bcur));
            else acc[kind][v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[v], bcur, acc[kind][v], 0, 0, 0);
            if constexpr (reload) a[v] = *reinterpret_cast<const half8 *>(smem + (reload_base + v * 2048));
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (has_next) {
            if constexpr (ABL & 1) bcur = __builtin_bit_cast(half8, lo);
            else {
                if (nsq) bcur = tally_half8(lo, hi);
                else bcur = to_half8(lo, hi);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    {   // static priority (diagnostics: SN_ASSIGN_DBG bit 1 = the younger half, bit 2 = the older half at priority 1)
        if ((p.dbg & 2) && sg == 1) __builtin_amdgcn_s_setprio(1);
        if ((p.dbg & 4) && sg == 0) __builtin_amdgcn_s_setprio(1);
    }
    using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
    using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
    {   // chunk 0 is in LDS: everything but slab(1), tok(1), tok(2) of this wave's copies
        if (w < 2) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        stamp(p, 5, lane, wave_id);
        // ---- accumulators start at |c|^2/2.  Leftover block b, accumulator row 4 kg + e: word 32 (4 b + kg) + 8 q + 4 sg + e
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const float4 c4 = *reinterpret_cast<const float4 *>(hcs + (4 * b + kg) * 32 + (q * 2 + sg) * 4);
            accl[b] = f32x4{fminf(c4.x, kPadHalfNorm), fminf(c4.y, kPadHalfNorm), fminf(c4.z, kPadHalfNorm), fminf(c4.w, kPadHalfNorm)};
            asm volatile("" : "+v"(accl[b]));
        }
        // Register x of virtual tile v, lane half h: word 32 (4 v + (x >> 2)) + 8 q + 4 h + (x & 3)
#pragma unroll
        for (int v = 0; v < 4; ++v) {
            f32x16 c0;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 c4 = *reinterpret_cast<const float4 *>(hcs + (4 * v + g) * 32 + (q * 2 + h) * 4);
                c0[4 * g + 0] = fminf(c4.x, kPadHalfNorm); c0[4 * g + 1] = fminf(c4.y, kPadHalfNorm);
                c0[4 * g + 2] = fminf(c4.z, kPadHalfNorm); c0[4 * g + 3] = fminf(c4.w, kPadHalfNorm);      // (padding words: +inf in the image, kept finite so that keys never become NaNs)
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[s][v] = c0;
        }
        // the first step's operands
#pragma unroll
        for (int v = 0; v < 4; ++v) a[v] = *reinterpret_cast<const half8 *>(smem + a_lane + v * 2048);
        const f32x4 lo = *reinterpret_cast<const f32x4 *>(smem + t_addr((unsigned)kS5OffT, 0, 0));
        const f32x4 hi = *reinterpret_cast<const f32x4 *>(smem + t_addr((unsigned)kS5OffT, 0, 1));
        if (q == 0) bcur = tally_half8(lo, hi);
        else bcur = to_half8(lo, hi);
        __builtin_amdgcn_sched_barrier(0);
    }
    static_for<NCH>([&](auto c_c) {
        constexpr int c = decltype(c_c)::value;
        constexpr bool last = c + 1 == NCH;
        // (the token slot's base goes through an opaque scalar per chunk - with the loop unrolled the compiler otherwise keeps the
        // addresses of every (set, k-step, slot) alive across chunks)
        unsigned rb = (unsigned)(kS5OffT + (c % 3) * kS5Tok), rbn = (unsigned)(kS5OffT + ((c + 1) % 3) * kS5Tok);
        asm volatile("" : "+s"(rb), "+s"(rbn));
        constexpr unsigned sl = (c & 1) * kS5Slab, sln = ((c + 1) & 1) * kS5Slab;
        if constexpr (c == 5) tq[0] = __builtin_amdgcn_s_memtime();
        run_step(I0{}, I0{}, I1{}, t_addr(rb, 0, 0) + 4096, t_addr(rb, 0, 1) + 4096, q == 1, 3 * sg + 1, 0u);           // S0; next: S1's rows
        // the other wave of the SIMD issues the copies its partner issued right behind the barrier one step later
        if constexpr (c >= 1 && !(ABL & 4)) {
            if (sg == 1) {
                if constexpr (c + 1 < NCH) issue_slab(c + 1);
                if constexpr (c + 2 < NCH) issue_tok(std::integral_constant<int, c + 2 < NCH ? c + 2 : 0>{});
            }
        }
        run_step(I1{}, I0{}, I1{}, t_addr(rb, 0, 0) + 8192, t_addr(rb, 0, 1) + 8192, q == 2, 3 * sg + 2, 0u);           // S1; next: S2's
        if constexpr (c == 5) tq[1] = __builtin_amdgcn_s_memtime();
        unsigned l_lo;
        leftover_addr(l_lo, a16_lane);
        run_step(I2{}, I1{}, I1{}, l_lo + rb, (l_lo ^ 16u) + rb, w == 3, 6, a16_lane + sl);                                   // S2; next: the leftover rows; A <- leftover blocks
        if constexpr (c == 5) tq[2] = __builtin_amdgcn_s_memtime();
        run_step(I3{}, I1{}, I1{}, t_addr(rb, 1, 0), t_addr(rb, 1, 1), q == 0, 3 * sg, a_lane + sl + 1024);                       // L; next: S3's rows; A <- k-step 1
        run_step(I0{}, I0{}, I1{}, t_addr(rb, 1, 0) + 4096, t_addr(rb, 1, 1) + 4096, q == 1, 3 * sg + 1, 0u);           // S3
        run_step(I1{}, I0{}, I1{}, t_addr(rb, 1, 0) + 8192, t_addr(rb, 1, 1) + 8192, q == 2, 3 * sg + 2, 0u);           // S4; next: S5's rows - the last LDS read of chunk c
        if constexpr (c == 5) tq[3] = __builtin_amdgcn_s_memtime();
        if constexpr (!last) {
            // barrier c.  Outstanding, oldest first: ..., slab(c + 1), tok(c + 1) | tok(c + 2): everything but the copies of tok(c + 2)
            if constexpr (c + 2 < NCH) {
                if (w < 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // (this wave's reads of chunk c have returned)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if constexpr (c + 2 == NCH) stamp(p, 6, lane, wave_id);
            if constexpr (c == 5) tq[4] = __builtin_amdgcn_s_memtime();
            if constexpr (!(ABL & 4)) {
                if (sg == 0) {
                    if constexpr (c + 2 < NCH) issue_slab(c + 2);
                    if constexpr (c + 3 < NCH) issue_tok(std::integral_constant<int, c + 3 < NCH ? c + 3 : 0>{});
                }
            }
            run_step(I2{}, I1{}, I1{}, t_addr(rbn, 0, 0), t_addr(rbn, 0, 1), q == 0, 3 * sg, a_lane + sln);                       // S5; next: S0 of chunk c + 1; A <- its k-step 0
        } else {
            run_step(I2{}, I0{}, I0{}, 0u, 0u, false, 0, 0u);                                                                // the very last step
        }
        if constexpr (c == 5) tq[5] = __builtin_amdgcn_s_memtime();
    });
    stamp(p, 2, lane, wave_id);
    // (the epilogue's lane-derived values, formed again from the lane number the hardware counts (v_mbcnt): carried across the
    // loop they cost eight registers the loop does not have - spilled in the prologue, a scratch segment at every dispatch)
    const int lane_e = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    unsigned ni_e = (unsigned)p.n_inner;                         // (and the divisor of out_index: its reciprocal is formed again, not kept)
    asm volatile("" : "+s"(ni_e));
    auto out_index_e = [&](int64_t n) {
        const unsigned o = (unsigned)n / ni_e, i = (unsigned)n - o * ni_e;
        return (int64_t)o * p.oso + (int64_t)i * p.osi;
    };
    const int r_e = lane_e & 31, h_e = lane_e >> 5, j16_e = lane_e & 15, kg_e = lane_e >> 4;
    // ---- keys: one sorted triple per lane and set; code = tile << 2 | e (the default screen's), tile = 4 v + (x >> 2), e = x & 3
    float m1[4], m2[4], m3[4];
    unsigned keymask = 0xFFFFFF00u;
    asm volatile("" : "+s"(keymask));
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        m1[s] = kBigKey; m2[s] = kBigKey; m3[s] = kBigKey;
#pragma unroll
        for (int v = 0; v < 4; ++v) {
#pragma unroll
            for (int x = 0; x < 16; ++x) {
                float k;
                const float val = acc[s][v][x];
                asm volatile("v_and_or_b32 %0, %4, %5, %6\n\t"
                             "v_med3_f32 %3, %0, %2, %3\n\t"
                             "v_med3_f32 %2, %0, %1, %2\n\t"
                             "v_min_f32 %1, %0, %1"
                             : "=&v"(k), "+v"(m1[s]), "+v"(m2[s]), "+v"(m3[s]) : "v"(val), "s"(keymask), "n"(((4 * v + (x >> 2)) << 2) | (x & 3)));
            }
        }
    }
    {   // leftover tokens.  Lane (j16, kg) holds, of block bl, the words 32 (4 bl + kg) + 8 q + 4 sg + e: code (4 bl + kg) << 2 | e in slot
        // group (h = sg, g = q); the four lanes kg of a token share the group: their triples are merged (lane ^ 16, lane ^ 32), after
        // which every one of them holds the group's triple
        float a1 = kBigKey, a2 = kBigKey, a3 = kBigKey;
        const unsigned lane_code = (unsigned)(kg_e << 2);
        auto ins = [&](float k) {
            a3 = __builtin_amdgcn_fmed3f(k, a2, a3);
            a2 = __builtin_amdgcn_fmed3f(k, a1, a2);
            a1 = fminf(k, a1);
        };
#pragma unroll
        for (int bl = 0; bl < 4; ++bl) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                ins(__uint_as_float((__float_as_uint(accl[bl][e]) & 0xFFFFFF00u) | lane_code | (unsigned)((bl << 4) | e)));
        }
#pragma unroll
        for (int off = 16; off <= 32; off *= 2) {
            const float o1 = __shfl_xor(a1, off, SN_WAVE), o2 = __shfl_xor(a2, off, SN_WAVE), o3 = __shfl_xor(a3, off, SN_WAVE);
            ins(o1); ins(o2); ins(o3);
        }
        m1[3] = a1; m2[3] = a2; m3[3] = a3;
    }
    stamp(p, 3, lane_e, wave_id);
    // ---- the window of every token -> LDS, by the wave that summed its squares (wave (sg, q < 3): set 3 sg + q; wave 3: the leftover
    // tokens as "set 6").  Per-token error window (DESIGN.md "S1 error window"): |key - exact value| <= E; stored: 2 E, NaN for a
    // token the screen cannot bound
    const unsigned *scal = reinterpret_cast<const unsigned *>(p.packed + lay.scal_off);
    const float C2 = __uint_as_float(scal[0]), C1 = __uint_as_float(scal[1]), CN = __uint_as_float(scal[2]), CMAX = __uint_as_float(scal[3]);
    const float DC = __uint_as_float(scal[4]);                                // max |c - fp16(c)|_2
    auto window_of = [&](float ssq, float sloss) {
        const float X2 = sqrtf(ssq) * 1.001f + 1.0e-6f, X1 = X2 * sqrtf((float)p.D);
        const float DX = sqrtf(sloss) * 1.001f + 1.0e-30f;                    // |x - fp16(x)|_2
        const float vmax = 0.5f * CN + X2 * C2;                               // >= |any partial sum|
        // x.c - x~.c~ = x.(c - c~) + (x - x~).c~ (exactly), each term by Cauchy-Schwarz; |c~| <= (1 + u) |c|
        const float E = 1.01f * (1.001f * (X2 * DC + DX * C2 * 1.0005f) + 5.96e-8f * (X1 + C1)
                                 + (float)(2 * NCH) * kAccUlpPerMfma * vmax + vmax * (3.0f * 5.96e-8f + 3.1e-5f));
        const bool bad = !(ssq <= kHugeIn * kHugeIn) || !(CMAX <= kHugeIn) || !(vmax < 1.0e28f);   // NaN-safe; |x|_2 <= 3e4 bounds every component
        return bad ? __uint_as_float(0x7FC00000u) : 2.0f * E;
    };
    if (q < 3) {
        sumsq += __shfl_xor(sumsq, 32, SN_WAVE); sumd += __shfl_xor(sumd, 32, SN_WAVE);
        if (h_e == 0) tsum[(3 * sg + q) * 32 + r_e] = window_of(sumsq, sumd);
    } else if (w == 3) {
        sumsq += __shfl_xor(sumsq, 16, SN_WAVE); sumd += __shfl_xor(sumd, 16, SN_WAVE);
        sumsq += __shfl_xor(sumsq, 32, SN_WAVE); sumd += __shfl_xor(sumd, 32, SN_WAVE);
        if (lane_e < 16) tsum[6 * 32 + lane_e] = window_of(sumsq, sumd);
    }
    // per slot of this wave: s < 3 = set 3 sg + s (lanes (r, h)), s == 3 = the leftover tokens (lanes < 16, row half sg)
    auto slot_of = [&](int s, int &set, int &rr, int &hh, bool &valid) {
        set = s < 3 ? 3 * sg + s : 6;
        rr = s < 3 ? r_e : j16_e;
        hh = s < 3 ? h_e : sg;
        const int row = 32 * set + rr;
        valid = (s < 3 || lane_e < 16) && row < n_mine;
    };
    // ---- per set: the best key of a token over the eight lanes that hold it
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int set, rr, hh; bool valid;
        slot_of(s, set, rr, hh, valid);
        if (valid) __builtin_amdgcn_ds_fminf((__attribute__((address_space(3))) float *)&tbest[set * 32 + rr], m1[s], 0, 0, false);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(p, 7, lane_e, wave_id);
    unsigned hm[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int set, rr, hh; bool valid;
        slot_of(s, set, rr, hh, valid);
        const float win = tsum[set * 32 + rr];
        const bool bad = !(win == win);
        const float best = tbest[set * 32 + rr];
        const bool any_finite = best < kKeyLimit;
        const float cut = best + win;
        const unsigned hmask = (m1[s] <= cut ? 1u : 0u) | (m2[s] <= cut ? 2u : 0u) | (m3[s] <= cut ? 4u : 0u);
        const bool hover = m3[s] <= cut;                                      // a 4th may hide behind it
        const unsigned contrib = (bad || !any_finite) ? sn_s1::kFlagFullScan : ((hmask << (12 * hh + 3 * q)) | (hover ? sn_s1::kFlagOverflow : 0u));
        if (valid && contrib) __hip_atomic_fetch_or(&tmask[set * 32 + rr], contrib, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        hm[s] = hmask;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    stamp(p, 8, lane_e, wave_id);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        int set, rr, hh; bool valid;
        slot_of(s, set, rr, hh, valid);
        const int64_t n = tok0 + 32 * set + rr;
        const unsigned mk = tmask[set * 32 + rr];
        const unsigned cand = mk & 0xFFFFFFu;
        const bool fullscan = ((mk >> 30) & 1u) != 0u || cand == 0u;
        const bool overflow = fullscan || (mk >> 31) != 0u;
        const int nc = __popc(cand);
        const unsigned code = __float_as_uint(m1[s]) & 0xFFu;
        // the only candidate: final (a flagged or overflow token's word is written by whoever finishes it)
        if (valid && !overflow && nc == 1 && (hm[s] & 1u) != 0u)
            p.out[out_index_e(n)] = (int)(code >> 2) * 32 + 8 * q + 4 * hh + (int)(code & 3u);
        if (valid && !fullscan && (nc > 1 || overflow)) {        // the eight lanes of the token write their three codes
            unsigned char *cd = p.codes + (int64_t)n * kCodeBytes + 12 * hh + 3 * q;
            cd[0] = (unsigned char)code;
            cd[1] = (unsigned char)(__float_as_uint(m2[s]) & 0xFFu);
            cd[2] = (unsigned char)(__float_as_uint(m3[s]) & 0xFFu);
        }
        const bool writer = valid && q == 0 && hh == 0;
        if (writer) {
            p.flags[n] = fullscan ? sn_s1::kFlagFullScan : ((overflow ? sn_s1::kFlagOverflow : 0u) | ((nc > 1 || overflow) ? cand : 0u));
            if (fullscan) p.out[out_index_e(n)] = 0;
        }
        const bool need_b = writer && overflow;
        const unsigned long long mask_b = __ballot(need_b);
        if (mask_b) {                                          // rare: tokens the screen cannot bound
            int base = 0;
            const int leader = __ffsll((long long)mask_b) - 1;
            if (lane_e == leader) base = atomicAdd(&p.work[1], __popcll(mask_b));
            base = __shfl(base, leader, SN_WAVE);
            if (need_b) p.overflow[base + __popcll(mask_b & ((1ull << lane_e) - 1ull))] = (int)n;
        }
    }
    stamp(p, 4, lane_e, wave_id);
    if (p.stamps && lane_e == 0) {
#pragma unroll
        for (int t = 0; t < 6; ++t) p.stamps[(size_t)wave_id * 16 + 9 + t] = tq[t];
    }
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

template <int NCH>
int launch_screen5(const AssignArgs &a, hipStream_t st, bool defer)
{
    constexpr int NTR = (NCH + 1) / 2;                          // fp64 re-rank: 64 k per lane-step
    if (int rc = sn_ensure_dynamic_lds((const void *)assign_screen5_kernel<NCH>, kS5Lds, "sn_assign_words")) return rc;
    AssignArgs ag = a;
    const int64_t cus = device_cus();
    int64_t tpw = (a.n_tokens + cus - 1) / cus;                 // one workgroup per CU, one round (the caller checked n_tokens <= cus kS5Rows)
    if (const char *e = getenv("SN_ASSIGN_TPW")) tpw = atoi(e);
    tpw = tpw < 1 ? 1 : (tpw > kS5Rows ? kS5Rows : tpw);
    ag.tps4 = (int)tpw;
    const unsigned grid = (unsigned)((a.n_tokens + tpw - 1) / tpw);
    sn_prof_start(0, st);
#ifdef SN_S5_ABLATION      // diagnostic builds (make FLAGS+=-DSN_S5_ABLATION): SN_S5_ABL selects an ablated form of the D = 384 kernel; results are wrong
    static const int abl = getenv("SN_S5_ABL") ? atoi(getenv("SN_S5_ABL")) : 0;
    if (abl != 0 && NCH == 12) {
        const void *fn = abl == 1 ? (const void *)assign_screen5_kernel<12, 1> : abl == 2 ? (const void *)assign_screen5_kernel<12, 2> : abl == 3 ? (const void *)assign_screen5_kernel<12, 3>
                       : abl == 4 ? (const void *)assign_screen5_kernel<12, 4> : abl == 6 ? (const void *)assign_screen5_kernel<12, 6> : (const void *)assign_screen5_kernel<12, 7>;
        if (int rc = sn_ensure_dynamic_lds(fn, kS5Lds, "sn_assign_words")) return rc;
        if (abl == 1) hipLaunchKernelGGL((assign_screen5_kernel<12, 1>), dim3(grid), dim3(512), kS5Lds, st, ag);
        else if (abl == 2) hipLaunchKernelGGL((assign_screen5_kernel<12, 2>), dim3(grid), dim3(512), kS5Lds, st, ag);
        else if (abl == 3) hipLaunchKernelGGL((assign_screen5_kernel<12, 3>), dim3(grid), dim3(512), kS5Lds, st, ag);
        else if (abl == 4) hipLaunchKernelGGL((assign_screen5_kernel<12, 4>), dim3(grid), dim3(512), kS5Lds, st, ag);
        else if (abl == 6) hipLaunchKernelGGL((assign_screen5_kernel<12, 6>), dim3(grid), dim3(512), kS5Lds, st, ag);
        else hipLaunchKernelGGL((assign_screen5_kernel<12, 7>), dim3(grid), dim3(512), kS5Lds, st, ag);
    } else
#endif
    hipLaunchKernelGGL((assign_screen5_kernel<NCH>), dim3(grid), dim3(512), kS5Lds, st, ag);
    sn_prof_stop(0, st);
    if (!defer) launch_rerank<NTR, 0>(ag, st);                  // (its records are the default screen's)
    return 0;
}

// form of the screen kernel: 0 = token-stationary (assign_screen_kernel: 4 waves x 3-slot codebook ring, two workgroups per CU: the
// default - it leaves half of every CU to the kernels of the other steps in flight); 5 = one-round K-outer on eight 256-register
// waves (assign_screen5_kernel: 7.5 % less kernel time alone, but it owns the CU; codebooks of 16 tiles - 448 < M <= 512 -, D in
// {192, 384}, fp32 tokens, at most 208 tokens per CU; other shapes take form 0).  Both write the same records.
// Initialised from SN_ASSIGN_VARIANT, changed with sn_assign_set_variant().
// (The lab forms of rounds 1-4 - 8-wave workgroups, codebook-stationary, K-outer in rounds, K-outer on four 512-register waves - are
// in the history and in profiles/NOTES_r01-r05.md / DESIGN.md section 8.)
int g_variant = -1;
int screen_variant()
{
    if (g_variant < 0) {
        // the environment and sn_assign_set_variant() agree on what exists: 0 and 5.  The lab forms 1-4 of rounds 1-4 left the library
        // (an old script that still exports one of them gets the default, with one note - not a form that sn_assign_variant() would
        // report and no kernel implements, which silently dropped the deferred S1 finish: ADVICE r05)
        const char *e = getenv("SN_ASSIGN_VARIANT");
        const int v = e ? atoi(e) : 0;
        if (v != 0 && v != 5) fprintf(stderr, "libschemanet_hip: SN_ASSIGN_VARIANT=%d is not a screen form of this library (0 or 5): using 0\n", v);
        g_variant = v == 5 ? 5 : 0;
    }
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
    if (lay.tiles5_off)
        hipLaunchKernelGGL(pack_frag5_kernel, dim3((unsigned)((int64_t)512 * D / 256)), dim3(256), 0, st, codebook, M, D,
                           base + lay.tiles5_off, lay.n_steps);
    hipLaunchKernelGGL(pack_norm_kernel, dim3((unsigned)((lay.m_pad + 3) / 4)), dim3(256), 0, st, codebook, M, D, lay.m_pad,
                       base + lay.tiles_off, lay.n_steps, lay.tile_bytes, (double *)(base + lay.cn64_off),
                       (unsigned *)(base + lay.scal_off));
    SN_CHECK_LAUNCH("sn_codebook_prepare");
    return SN_OK;
}

extern "C" int sn_assign_variant(void) { return screen_variant(); }

// mode 2 leaves its flagged tokens to the consumer only on the default (token-stationary) screen with byte codes
extern "C" int sn_assign_defers(int M, int D)
{
    const int v = screen_variant();
    const bool dflt = v == 0 || v == 5;                          // (form 5 writes the default form's records)
    return (dflt && M > 0 && M <= 2048 && (D == 192 || D == 384 || D == 768)) ? 1 : 0;      // (D = 768, round 5: one candidate pair in flight per row wave - three 12-register rows - fits the graph kernel's 128 registers; four did not)
}

extern "C" int sn_assign_set_variant(int variant)
{
    SN_REQUIRE(variant == 0 || variant == 5, SN_ERR_BAD_ARG, "sn_assign_set_variant: variant=%d (0 or 5)", variant);
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
    a.full_waves = kWavesPerBlock; a.extra_base = n_tokens;
    a.x_bf16 = x_bf16;
    a.tps4 = 0;
    { static const int dbg = getenv("SN_ASSIGN_DBG") ? atoi(getenv("SN_ASSIGN_DBG")) : 0; a.dbg = dbg; }
    a.gate = ws ? (unsigned *)(ws + ((32 + (size_t)n_tokens * kWsPerToken2 + 15) & ~size_t(15))) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    const int per16 = x_bf16 ? 8 : 4;                            // elements per 16 bytes
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && x_stride_outer % per16 == 0 && x_stride_inner % per16 == 0;
    const bool screen_ok = mode == 0 && aligned && (SN_S1_STAGE || !x_bf16) && M <= 32 * kMaxTilesScreen && (D == 192 || D == 384 || D == 768);
    if (finish_only) {
        // (where mode 2 did not defer it has cleared the flag words and the overflow count: the kernels find nothing)
        if (D == 192) launch_rerank<3, 0>(a, st);
        else if (D == 384) launch_rerank<6, 0>(a, st);
        else if (D == 768) launch_rerank<12, 0>(a, st);
        else SN_REQUIRE(false, SN_ERR_UNSUPPORTED, "sn_assign_words: mode 3 for D=%d", D);
        SN_CHECK_LAUNCH("sn_assign_words");
        return SN_OK;
    }
    if (screen_ok) {
        SN_REQUIRE(workspace && workspace_bytes >= sn_assign_workspace_bytes(n_tokens), SN_ERR_WORKSPACE,
                   "sn_assign_words: workspace %zu < %zu bytes", workspace_bytes, sn_assign_workspace_bytes(n_tokens));
        if (int rc0 = sn_zero_async(workspace, 32, st)) return rc0;
        int rc = 0;
        const PackLayout lay = pack_layout(M, D);
        // (K-outer form: fp32 tokens whose byte offsets fit 32 bits: the copies address them as base + 32-bit lane offset)
        if (screen_variant() == 5 && !x_bf16 && lay.tiles5_off != 0 && n_tokens <= (int64_t)device_cus() * kS5Rows &&
            ((n_outer - 1) * x_stride_outer + (n_inner - 1) * x_stride_inner + D) * 4 < (int64_t)0xFFFFF000ll && x_stride_outer >= 0 && x_stride_inner >= 0) {
            rc = D == 192 ? launch_screen5<6>(a, st, deferred = want_defer) : launch_screen5<12>(a, st, deferred = want_defer);
        } else if (M > 2048) {               // more than 64 tiles: 10-bit word codes in the keys, 16-bit codes in the records
            a.overflow = (int *)(ws + 32 + (size_t)n_tokens * (4 + kCodeBytesWide));
            if (D == 192) rc = launch_screen<12, 4, 3, 10>(a, st);
            else if (D == 384) rc = launch_screen<24, 4, 3, 10>(a, st);
            else rc = launch_screen<48, 4, 3, 10>(a, st);
        } else if (D == 192) rc = launch_screen<12, 4, 3>(a, st, deferred = want_defer);
        else if (D == 384) rc = (assign_dual() ? launch_screen<24, 4, 3, 8, true>(a, st) : launch_screen<24, 4, 3>(a, st, deferred = want_defer));
        else rc = launch_screen<48, 4, 3>(a, st, deferred = want_defer);
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
