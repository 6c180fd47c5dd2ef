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

#include <hip/hip_fp16.h>

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kTokPerWave = 32;
constexpr int kWavesPerBlock = 4;
constexpr int kTokPerBlock = kTokPerWave * kWavesPerBlock;
constexpr int kRing = 3;               // LDS slots for codebook tiles
constexpr int kMaxCand = 6;
constexpr int kOverflow = 255;         // work-list entry: scan every word (a 6-bit mask is < 64)
constexpr float kU16 = 4.8828125e-4f;  // 2^-11, fp16 unit round-off
constexpr float kHugeIn = 3.0e4f;      // |value| above this does not go through fp16

// packed codebook image -------------------------------------------------------------------
struct PackLayout {
    size_t frag_off, cn32_off, cn64_off, scal_off, total;
    int n_tiles, n_steps;
};

__host__ __device__ inline PackLayout pack_layout(int M, int D)
{
    PackLayout p;
    p.n_tiles = (M + 31) / 32;
    p.n_steps = D / 16;
    const size_t mp = (size_t)p.n_tiles * 32;
    p.frag_off = 0;
    p.cn32_off = mp * D * 2;
    p.cn64_off = p.cn32_off + ((mp * 4 + 255) & ~size_t(255));
    p.scal_off = p.cn64_off + ((mp * 8 + 255) & ~size_t(255));
    p.total = p.scal_off + 256;
    return p;
}
// scalars (uint bit patterns of non-negative floats, so atomicMax orders them):
//   [0] max |c|_2   [1] max |c|_1   [2] max |c|^2   [3] max |c_mk|

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

// ------------------------------------------------------------------------------------------
// codebook_prepare
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_frag_kernel(const float *cb, int M, int D, _Float16 *frag, int n_steps)
{
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;   // over [M_pad, D]
    const int m = (int)(idx / D), k = (int)(idx % D);
    const int w = m >> 5, i = m & 31;
    const int u = k >> 5, rem = k & 31, h = rem >> 4, e = (rem >> 3) & 1, j = rem & 7;
    const int s = 2 * u + e;
    const float v = m < M ? cb[(int64_t)m * D + k] : 0.0f;
    frag[(((int64_t)w * n_steps + s) * 64 + (i + 32 * h)) * 8 + j] = (_Float16)v;
}

__global__ __launch_bounds__(256) void pack_norm_kernel(const float *cb, int M, int D, int m_pad, float *cn32,
                                                        double *cn64, unsigned *scal)
{
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (m >= m_pad) return;
    if (m >= M) {                       // padding words can never win
        if (lane == 0) { cn32[m] = INFINITY; cn64[m] = (double)INFINITY; }
        return;
    }
    const float *c = cb + (int64_t)m * D;
    double p = 0.0;
    float l1 = 0.0f, mx = 0.0f;
    for (int k = lane; k < D; k += SN_WAVE) {
        const float v = c[k];
        p = fma((double)v, (double)v, p);
        l1 += fabsf(v);
        mx = fmaxf(mx, fabsf(v));
    }
    p = sn_wave_sum_f64(p);
    l1 = sn_wave_sum(l1);
    mx = sn_wave_max(mx);
    if (lane == 0) {
        cn64[m] = p;
        cn32[m] = (float)p;
        const float up = 1.0f + 1.0e-6f;
        atomicMax(&scal[0], __float_as_uint(sqrtf((float)p) * up));
        atomicMax(&scal[1], __float_as_uint(l1 * (1.0f + 1.0e-4f)));
        atomicMax(&scal[2], __float_as_uint((float)p * up));
        atomicMax(&scal[3], __float_as_uint(mx));
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
    int *work;          // [0] = entry count, entries start at int 8, 8 ints each
};

__device__ __forceinline__ const float *token_row(const AssignArgs &p, int64_t n)
{
    return p.x + (n / p.n_inner) * p.xso + (n % p.n_inner) * p.xsi;
}

__device__ __forceinline__ int64_t out_index(const AssignArgs &p, int64_t n)
{
    return (n / p.n_inner) * p.oso + (n % p.n_inner) * p.osi;
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
__device__ __forceinline__ void load_token64(double (&x)[NT], const float *row, int D, int lane)
{
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int k = lane + SN_WAVE * t;
        x[t] = k < D ? (double)row[k] : 0.0;
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
        load_token64<NT>(x, token_row(p, n), p.D, lane);
        double best = (double)INFINITY;
        int bi = 0;
        exact_scan<NT>(x, p, cn64, 0, p.M, lane, best, bi);
        if (lane == 0) p.out[out_index(p, n)] = bi;
    }
}

// ------------------------------------------------------------------------------------------
// mode 0, pass 2: fp64 re-rank of the work list
// ------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void assign_rerank_kernel(const AssignArgs p)
{
    const int lane = threadIdx.x & 63;
    const PackLayout lay = pack_layout(p.M, p.D);
    const double *cn64 = (const double *)(p.packed + lay.cn64_off);
    const int count = p.work[0];
    const int n_waves = gridDim.x * kWavesPerBlock;
    for (int e = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6); e < count; e += n_waves) {
        const int *ent = p.work + 8 + (int64_t)e * 8;
        const int64_t n = ent[0];
        const int cmask = ent[1];
        double x[NT];
        load_token64<NT>(x, token_row(p, n), p.D, lane);
        double best = (double)INFINITY;
        int bi = 0x7fffffff;
        if (cmask == kOverflow) {
            bi = 0;
            exact_scan<NT>(x, p, cn64, 0, p.M, lane, best, bi);
        } else {
            for (int c = 0; c < kMaxCand; ++c) {
                if (!((cmask >> c) & 1)) continue;
                const int m = ent[2 + c];
                const double s = cn64[m] - 2.0 * dot64<NT>(x, p.cb + (int64_t)m * p.D, p.D, lane);
                if (s < best || (s == best && m < bi)) { best = s; bi = m; }
            }
            if (bi == 0x7fffffff) {    // every candidate NaN: fall back to the full scan
                bi = 0;
                exact_scan<NT>(x, p, cn64, 0, p.M, lane, best, bi);
            }
        }
        if (lane == 0) p.out[out_index(p, n)] = bi;
    }
}

// ------------------------------------------------------------------------------------------
// mode 0, pass 1: fp16-MFMA screen
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void top3_insert(unsigned k, unsigned &m1, unsigned &m2, unsigned &m3)
{
    const unsigned lo = min(k, m2);
    m2 = max(min(k, m1), min(max(k, m1), m2));   // v_med3_u32
    m1 = max(k, m1);
    m3 = max(m3, lo);
}

struct Cand { float v; int w; };

__device__ __forceinline__ void cand_insert(Cand (&c)[3], float v, int w)
{
    // descending by value; on equal value the lower word index first
    if (v > c[0].v || (v == c[0].v && w < c[0].w)) { c[2] = c[1]; c[1] = c[0]; c[0].v = v; c[0].w = w; }
    else if (v > c[1].v || (v == c[1].v && w < c[1].w)) { c[2] = c[1]; c[1].v = v; c[1].w = w; }
    else if (v > c[2].v || (v == c[2].v && w < c[2].w)) { c[2].v = v; c[2].w = w; }
}

template <int NSTEPS>
__global__ __launch_bounds__(256, (NSTEPS <= 24 ? 2 : 1)) void assign_screen_kernel(const AssignArgs p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int kTileBytes = NSTEPS * 1024;
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const PackLayout lay = pack_layout(p.M, p.D);
    const unsigned char *frag = p.packed + lay.frag_off;
    const float *cn32 = (const float *)(p.packed + lay.cn32_off);
    const unsigned *scal = (const unsigned *)(p.packed + lay.scal_off);
    const int n_tiles = lay.n_tiles;

    auto issue_tile = [&](int w, int slot) {
#pragma unroll
        for (int j = 0; j < NSTEPS / kWavesPerBlock; ++j) {
            const int c = wid + kWavesPerBlock * j;
            const unsigned char *src = frag + (size_t)w * kTileBytes + c * 1024 + lane * 16;
            unsigned char *dst = smem + slot * kTileBytes + c * 1024;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
        }
    };
    issue_tile(0, 0);
    if (n_tiles > 1) issue_tile(1, 1);

    // ---- this wave's 32 tokens: fp32 -> fp16 B fragments, kept in registers for the whole kernel
    const int64_t n = (int64_t)blockIdx.x * kTokPerBlock + wid * kTokPerWave + r;
    const bool valid = n < p.n_tokens;
    const float *row = token_row(p, valid ? n : 0);
    half8 b[NSTEPS];
    float sumsq = 0.0f, sumabs = 0.0f, maxabs = 0.0f;
#pragma unroll
    for (int u = 0; u < NSTEPS / 2; ++u) {
        const float4 *q = reinterpret_cast<const float4 *>(row + 32 * u + 16 * h);
        const float4 v0 = q[0], v1 = q[1], v2 = q[2], v3 = q[3];
        const float f[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w,
                             v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            sumsq = fmaf(f[j], f[j], sumsq);
            sumabs += fabsf(f[j]);
            maxabs = fmaxf(maxabs, fabsf(f[j]));
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) { b[2 * u][j] = (_Float16)f[j]; b[2 * u + 1][j] = (_Float16)f[8 + j]; }
    }
    sumsq += __shfl_xor(sumsq, 32, SN_WAVE);
    sumabs += __shfl_xor(sumabs, 32, SN_WAVE);
    maxabs = fmaxf(maxabs, __shfl_xor(maxabs, 32, SN_WAVE));

    // ---- per-token constants of the error analysis (see DESIGN.md "S1 error window")
    const float C2 = __uint_as_float(scal[0]), C1 = __uint_as_float(scal[1]);
    const float CN = __uint_as_float(scal[2]), CMAX = __uint_as_float(scal[3]);
    const float X2 = sqrtf(sumsq) * 1.001f, X1 = sumabs * 1.001f;
    const float OFF = 0.5f * CN + 1.02f * X2 * C2;
    // |acc'_computed - acc'_exact| <= E for every word
    const float E = 1.01f * (2.01f * kU16 * X2 * C2 + 5.96e-8f * (X1 + C1) +
                             OFF * (1.2e-7f + (float)(16 * NSTEPS) * 4.8e-7f + 6.2e-5f));
    const float window = 2.0f * E;
    const bool bad = !(maxabs <= kHugeIn) || !(CMAX <= kHugeIn) || !(OFF < 1.0e30f);   // NaN-safe

    unsigned m1 = 0, m2 = 0, m3 = 0;
    Cand top[3] = {{-1.0f, -1}, {-1.0f, -1}, {-1.0f, -1}};

    for (int w = 0; w < n_tiles; ++w) {
        __syncthreads();                 // tile w landed (vmcnt(0) + barrier); slot (w+2)%3 is free
        if (w + 2 < n_tiles) issue_tile(w + 2, (w + 2) % kRing);
        const unsigned char *slot = smem + (w % kRing) * kTileBytes;
        f32x16 acc;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 c4 = *reinterpret_cast<const float4 *>(cn32 + 32 * w + 8 * g + 4 * h);
            acc[4 * g + 0] = OFF - 0.5f * c4.x;
            acc[4 * g + 1] = OFF - 0.5f * c4.y;
            acc[4 * g + 2] = OFF - 0.5f * c4.z;
            acc[4 * g + 3] = OFF - 0.5f * c4.w;
        }
#pragma unroll
        for (int s = 0; s < NSTEPS; ++s) {
            const half8 a = *reinterpret_cast<const half8 *>(slot + s * 1024 + lane * 16);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b[s], acc, 0, 0, 0);
        }
        const unsigned tcode = (unsigned)(w & 15) << 4;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const float v = fmaxf(acc[reg], 0.0f);      // also maps NaN / -inf (padding words) to 0
            const unsigned k = (__float_as_uint(v) & 0xFFFFFF00u) | (tcode | (unsigned)reg);
            top3_insert(k, m1, m2, m3);
        }
        if ((w & 15) == 15 || w == n_tiles - 1) {        // unpack this 16-tile chunk
            const int base = (w & ~15) * 32;
            const unsigned ks[3] = {m1, m2, m3};
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const unsigned k = ks[q];
                const float v = __uint_as_float(k & 0xFFFFFF00u);
                const int code = (int)(k & 0xFFu), reg = code & 15;
                const int word = base + (code >> 4) * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * h;
                if (v > 0.0f && word < p.M) cand_insert(top, v, word);
            }
            m1 = m2 = m3 = 0;
        }
    }

    // ---- merge the two half-lanes that share a token
    Cand oth[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        oth[q].v = __shfl_xor(top[q].v, 32, SN_WAVE);
        oth[q].w = __shfl_xor(top[q].w, 32, SN_WAVE);
    }
    float vbest = fmaxf(top[0].v, oth[0].v);
    int best_w = (top[0].v > oth[0].v || (top[0].v == oth[0].v && (unsigned)top[0].w < (unsigned)oth[0].w)) ? top[0].w : oth[0].w;
    const float cut = vbest - window;
    // candidate set as a 6-bit mask over {top[0..2], oth[0..2]} (no runtime-indexed arrays)
    unsigned cmask = 0;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        if (top[q].w >= 0 && top[q].v >= cut) cmask |= 1u << q;
        if (oth[q].w >= 0 && oth[q].v >= cut) cmask |= 8u << q;
    }
    const int nc = __popc(cmask);
    // a lane whose third-best is still inside the window may hide a fourth
    const bool overflow = bad || best_w < 0 || (top[2].w >= 0 && top[2].v >= cut) || (oth[2].w >= 0 && oth[2].v >= cut);
    const bool writer = valid && h == 0;
    if (writer) p.out[out_index(p, n)] = best_w < 0 ? 0 : best_w;
    const bool need = writer && (overflow || nc > 1);
    const unsigned long long mask = __ballot(need);
    if (mask) {
        int base = 0;
        const int leader = __ffsll((long long)mask) - 1;
        if (lane == leader) base = atomicAdd(&p.work[0], __popcll(mask));
        base = __shfl(base, leader, SN_WAVE);
        if (need) {
            const int slot_i = base + __popcll(mask & ((1ull << lane) - 1ull));
            int *ent = p.work + 8 + (int64_t)slot_i * 8;
            ent[0] = (int)n;
            ent[1] = overflow ? kOverflow : (int)cmask;
#pragma unroll
            for (int q = 0; q < 3; ++q) { ent[2 + q] = top[q].w; ent[5 + q] = oth[q].w; }
        }
    }
}

template <int NT>
int launch_exact(const AssignArgs &a, hipStream_t st)
{
    const int64_t blocks = (a.n_tokens + kWavesPerBlock - 1) / kWavesPerBlock;
    const unsigned grid = (unsigned)(blocks < 8192 ? blocks : 8192);
    hipLaunchKernelGGL(assign_exact_kernel<NT>, dim3(grid), dim3(256), 0, st, a);
    return 0;
}

template <int NSTEPS>
int launch_screen(const AssignArgs &a, hipStream_t st)
{
    const size_t lds = (size_t)kRing * NSTEPS * 1024;
    static bool attr_set = false;
    if (!attr_set && lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)assign_screen_kernel<NSTEPS>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { sn_set_error("sn_assign_words: LDS attribute: %s", hipGetErrorString(e)); return SN_ERR_LAUNCH; }
        attr_set = true;
    }
    const unsigned grid = (unsigned)((a.n_tokens + kTokPerBlock - 1) / kTokPerBlock);
    sn_prof_start(0, st);
    hipLaunchKernelGGL(assign_screen_kernel<NSTEPS>, dim3(grid), dim3(256), lds, st, a);
    sn_prof_stop(0, st);
    constexpr int NT = NSTEPS / 4;
    sn_prof_start(1, st);
    hipLaunchKernelGGL(assign_rerank_kernel<NT>, dim3(1024), dim3(256), 0, st, a);
    sn_prof_stop(1, st);
    return 0;
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
    if (hipMemsetAsync(base + lay.scal_off, 0, 256, st) != hipSuccess) {
        sn_set_error("sn_codebook_prepare: memset failed");
        return SN_ERR_LAUNCH;
    }
    const int m_pad = lay.n_tiles * 32;
    const int64_t elems = (int64_t)m_pad * D;
    hipLaunchKernelGGL(pack_frag_kernel, dim3((unsigned)(elems / 256)), dim3(256), 0, st, codebook, M, D,
                       (_Float16 *)(base + lay.frag_off), lay.n_steps);
    hipLaunchKernelGGL(pack_norm_kernel, dim3((unsigned)((m_pad + 3) / 4)), dim3(256), 0, st, codebook, M, D, m_pad,
                       (float *)(base + lay.cn32_off), (double *)(base + lay.cn64_off), (unsigned *)(base + lay.scal_off));
    SN_CHECK_LAUNCH("sn_codebook_prepare");
    return SN_OK;
}

extern "C" size_t sn_assign_workspace_bytes(int64_t n_tokens)
{
    if (n_tokens < 0) return 0;
    return 32 + (size_t)n_tokens * 32;
}

extern "C" int sn_assign_words(const float *x, int64_t n_outer, int64_t n_inner, int64_t x_stride_outer,
                               int64_t x_stride_inner, const float *codebook, const void *packed, int M, int D,
                               int64_t *out, int64_t out_stride_outer, int64_t out_stride_inner,
                               void *workspace, size_t workspace_bytes, int mode, void *stream)
{
    SN_REQUIRE(n_outer >= 0 && n_inner >= 0, SN_ERR_BAD_ARG, "sn_assign_words: negative token grid");
    const int64_t n_tokens = n_outer * n_inner;
    if (n_tokens == 0) return SN_OK;
    SN_REQUIRE(x && codebook && packed && out, SN_ERR_BAD_ARG, "sn_assign_words: NULL pointer");
    SN_REQUIRE(M > 0 && M <= 65536, SN_ERR_BAD_ARG, "sn_assign_words: M=%d out of range", M);
    SN_REQUIRE(D > 0 && D % 32 == 0 && D <= 1024, SN_ERR_UNSUPPORTED, "sn_assign_words: D=%d must be a multiple of 32, <= 1024", D);
    SN_REQUIRE(n_tokens < 0x7fffffff, SN_ERR_UNSUPPORTED, "sn_assign_words: too many tokens");
    SN_REQUIRE(mode == 0 || mode == 1, SN_ERR_BAD_ARG, "sn_assign_words: mode=%d", mode);
    AssignArgs a;
    a.x = x; a.n_tokens = n_tokens; a.n_inner = n_inner; a.xso = x_stride_outer; a.xsi = x_stride_inner;
    a.cb = codebook; a.packed = (const unsigned char *)packed; a.M = M; a.D = D;
    a.out = out; a.oso = out_stride_outer; a.osi = out_stride_inner; a.work = (int *)workspace;
    hipStream_t st = (hipStream_t)stream;
    const bool aligned = (reinterpret_cast<uintptr_t>(x) & 15) == 0 && x_stride_outer % 4 == 0 && x_stride_inner % 4 == 0;
    const bool screen_ok = mode == 0 && aligned && M % 32 == 0 && (D == 192 || D == 384 || D == 768);
    if (screen_ok) {
        SN_REQUIRE(workspace && workspace_bytes >= sn_assign_workspace_bytes(n_tokens), SN_ERR_WORKSPACE,
                   "sn_assign_words: workspace %zu < %zu bytes", workspace_bytes, sn_assign_workspace_bytes(n_tokens));
        if (hipMemsetAsync(workspace, 0, 32, st) != hipSuccess) {
            sn_set_error("sn_assign_words: memset failed");
            return SN_ERR_LAUNCH;
        }
        int rc = 0;
        if (D == 192) rc = launch_screen<12>(a, st);
        else if (D == 384) rc = launch_screen<24>(a, st);
        else rc = launch_screen<48>(a, st);
        if (rc) return rc;
    } else {
        const int nt = (D + 63) / 64;
        if (nt <= 3) launch_exact<3>(a, st);
        else if (nt <= 6) launch_exact<6>(a, st);
        else if (nt <= 12) launch_exact<12>(a, st);
        else launch_exact<16>(a, st);
    }
    SN_CHECK_LAUNCH("sn_assign_words");
    return SN_OK;
}
