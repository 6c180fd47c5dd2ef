// Feasibility probe for a one-round, accumulator-resident K-outer S1 screen (NOTES: "form 4"): one workgroup of four
// 512-register waves per CU keeps the accumulators of 192 tokens x 512 words (6 sets x 4 tiles x 16 registers per wave) for
// the whole launch; per 32-float chunk of K the tokens (24 KB, HBM) and the codebook slab (32 KB, L2) arrive by LDS-DMA into
// two-slot rings while the previous chunk's 48 MFMAs per wave run.  No conversion, no keys: it measures what the CU's
// intake + matrix pipe allow.   hipcc --offload-arch=gfx950 -O3 -o proto4 tools/proto_screen4.hip && ./proto4
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#ifndef SETS
#define SETS 6
#endif
constexpr int NTW = 4, CH = 12, D = 384;
constexpr int kSlab = 32768, kTok = SETS * 32 * 128;          // bytes per chunk: codebook slab, raw tokens
constexpr int kOffT = 2 * kSlab;
constexpr int kLds = 2 * kSlab + 2 * kTok;

template <bool DMA_T, bool DMA_A, bool MFMA>
__global__ __launch_bounds__(256, 1) void probe(const float *x, const unsigned char *cb, float *out, int tokens_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const float *xb = x + (size_t)blockIdx.x * tokens_per_wg * D;
    // token copies: instruction i of wave w covers tokens 8 (6 w + i) .. + 7, 128 B each
    unsigned tv[SETS];
#pragma unroll
    for (int i = 0; i < SETS; ++i) tv[i] = (unsigned)(((8 * (SETS * wid + i) + (lane >> 3)) * D) * 4 + (lane & 7) * 16);
    const unsigned av = (unsigned)(wid * 8 * 1024 + lane * 16);
    auto issue = [&](int c) {
        const unsigned slot = (unsigned)(c & 1);
        if (DMA_T) {
#pragma unroll
            for (int i = 0; i < SETS; ++i) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + kOffT + slot * kTok + (SETS * wid + i) * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(tv[i] + (unsigned)c * 128u), "s"(xb), "s"(dst) : "memory");
            }
        }
        if (DMA_A) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * kSlab + (wid * 8 + i) * 1024);
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(av + (unsigned)(c * kSlab + i * 1024)), "s"(cb), "s"(dst) : "memory");
            }
        }
    };
    f32x16 acc[SETS][NTW];
#pragma unroll
    for (int s = 0; s < SETS; ++s)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.0f;
    issue(0);
    for (int c = 0; c < CH; ++c) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 1 < CH) issue(c + 1);
        if (MFMA) {
            const unsigned char *slab = smem + (c & 1) * kSlab + wid * NTW * 2048 + lane * 16;
            const unsigned char *tok = smem + kOffT + (c & 1) * kTok + lane * 16;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 a[NTW], b[SETS];
#pragma unroll
                for (int t = 0; t < NTW; ++t) a[t] = *reinterpret_cast<const half8 *>(slab + t * 2048 + ks * 1024);
#pragma unroll
                for (int s = 0; s < SETS; ++s) b[s] = *reinterpret_cast<const half8 *>(tok + s * 4096 + ks * 1024);
                // (the accumulators of sets 0..3 live in AGPRs, those of sets 4, 5 in VGPRs: 256 + 128 registers; the compiler
                // keeps every MFMA of a function in ONE of the two files, so the instruction is written out)
#pragma unroll
                for (int s = 0; s < SETS; ++s)
#pragma unroll
                    for (int t = 0; t < NTW; ++t) {
                        if (s < 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[s][t]) : "v"(a[t]), "v"(b[s]));
                        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[s][t]) : "v"(a[t]), "v"(b[s]));
                    }
            }
        }
    }
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < SETS; ++s)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[s][t][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <bool T, bool A, bool M>
static int run(const char *name, float **xs, unsigned char *cb, float *out)
{
    CK(hipFuncSetAttribute((const void *)probe<T, A, M>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((probe<T, A, M>), dim3(256), dim3(256), kLds, 0, xs[i & 3], cb, out, 196);
    CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0.0f;
    for (int i = 0; i < 20; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<T, A, M>), dim3(256), dim3(256), kLds, 0, xs[i & 3], cb, out, 196);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; tot += ms;
    }
    printf("%-28s best %.1f us  mean %.1f us\n", name, best * 1e3f, tot / 20 * 1e3f);
    return 0;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---- variant 3: LDS-DMA for both streams, tokens TWO chunks ahead (three raw slots), codebook one ahead (two slots); the fp16
// B fragments are formed in registers from the raw fp32 rows right before their MFMAs (each wave converts what it multiplies:
// no fragment buffer, one barrier per chunk).  LDS: 2 x 32 KB + 3 x 24 KB = 136 KB.
constexpr int kLds3 = 2 * kSlab + 3 * kTok;
template <bool MFMA>
__global__ __launch_bounds__(256, 1) void probe3(const float *x, const unsigned char *cb, float *out, int tokens_per_wg)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    const float *xb = x + (size_t)blockIdx.x * tokens_per_wg * D;
    // token copies: instruction i of wave w covers tokens 8 (6 w + i) .. + 7; lane -> (token, piece ^ swizzle)
    const int trow = lane >> 3, pslot = lane & 7;
    unsigned tv[SETS];
#pragma unroll
    for (int i = 0; i < SETS; ++i) {
        const int t = 8 * (SETS * wid + i) + trow;
        tv[i] = (unsigned)((t * D) * 4 + ((pslot ^ ((t >> 1) & 7)) * 16));
    }
    const unsigned av = (unsigned)(wid * 8 * 1024 + lane * 16);
    auto issue_tok = [&](int c) {
        const unsigned slot = (unsigned)(c % 3);
#pragma unroll
        for (int i = 0; i < SETS; ++i) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 2 * kSlab + slot * kTok + (SETS * wid + i) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(tv[i] + (unsigned)c * 128u), "s"(xb), "s"(dst) : "memory");
        }
    };
    auto issue_slab = [&](int c) {
        const unsigned slot = (unsigned)(c & 1);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * kSlab + (wid * 8 + i) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(av + (unsigned)(c * kSlab + i * 1024)), "s"(cb), "s"(dst) : "memory");
        }
    };
    f32x16 acc[SETS][NTW];
#pragma unroll
    for (int s = 0; s < SETS; ++s)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[s][t][r] = 0.0f;
    // issue order: slab(0), tok(0), tok(1); per iteration c: slab(c + 1), tok(c + 2)
    issue_slab(0); issue_tok(0); issue_tok(1);
    const int tok = lane & 31, kh = lane >> 5;
    for (int c = 0; c < CH; ++c) {
        // outstanding, oldest first: [tok(c), slab(c), tok(c + 1)] (c == 0: slab(0), tok(0), tok(1)): everything but tok(c + 1)
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(SETS) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (c + 1 < CH) issue_slab(c + 1);
        if (c + 2 < CH) issue_tok(c + 2);
        if (MFMA) {
            const unsigned char *slab = smem + (c & 1) * kSlab + wid * NTW * 2048 + lane * 16;
            const unsigned char *raw = smem + 2 * kSlab + (c % 3) * kTok;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 a[NTW];
#pragma unroll
                for (int t = 0; t < NTW; ++t) a[t] = *reinterpret_cast<const half8 *>(slab + t * 2048 + ks * 1024);
#pragma unroll
                for (int s = 0; s < SETS; ++s) {
                    // token 32 s + tok, floats 16 ks + 8 kh .. + 7 of the chunk = pieces 4 ks + 2 kh, + 1 (swizzled slots)
                    const int t = 32 * s + tok;
                    const int p0 = 4 * ks + 2 * kh;
                    const unsigned char *row = raw + t * 128;
                    const f32x4 lo = *reinterpret_cast<const f32x4 *>(row + ((p0 ^ ((t >> 1) & 7)) << 4));
                    const f32x4 hi = *reinterpret_cast<const f32x4 *>(row + (((p0 + 1) ^ ((t >> 1) & 7)) << 4));
                    half8 b;
                    b[0] = (_Float16)lo.x; b[1] = (_Float16)lo.y; b[2] = (_Float16)lo.z; b[3] = (_Float16)lo.w;
                    b[4] = (_Float16)hi.x; b[5] = (_Float16)hi.y; b[6] = (_Float16)hi.z; b[7] = (_Float16)hi.w;
#pragma unroll
                    for (int t2 = 0; t2 < NTW; ++t2) {
                        if (s < 4) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[s][t2]) : "v"(a[t2]), "v"(b));
                        else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[s][t2]) : "v"(a[t2]), "v"(b));
                    }
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float sum = 0.0f;
#pragma unroll
    for (int s = 0; s < SETS; ++s)
#pragma unroll
        for (int t = 0; t < NTW; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[s][t][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

template <bool M>
static int run3(const char *name, float **xs, unsigned char *cb, float *out)
{
    CK(hipFuncSetAttribute((const void *)probe3<M>, hipFuncAttributeMaxDynamicSharedMemorySize, kLds3));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 4; ++i) hipLaunchKernelGGL((probe3<M>), dim3(256), dim3(256), kLds3, 0, xs[i & 3], cb, out, 196);
    CK(hipDeviceSynchronize());
    float best = 1e9f, tot = 0.0f;
    for (int i = 0; i < 20; ++i) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe3<M>), dim3(256), dim3(256), kLds3, 0, xs[i & 3], cb, out, 196);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best; tot += ms;
    }
    printf("%-28s best %.1f us  mean %.1f us\n", name, best * 1e3f, tot / 20 * 1e3f);
    return 0;
}

__global__ void empty_kernel(float *o) { if (o == nullptr) o[0] = 0.0f; }

int main()
{
    const size_t n_tok = 50176, xbytes = n_tok * D * 4;
    float *xs[4]; unsigned char *cb; float *out;
    std::vector<float> h(n_tok * D);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)(rand() % 1000) * 1e-3f;
    for (int i = 0; i < 4; ++i) { CK(hipMalloc(&xs[i], xbytes + 4096)); CK(hipMemcpy(xs[i], h.data(), xbytes, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&cb, CH * kSlab));
    { std::vector<unsigned short> hc(CH * kSlab / 2, 0x3400); CK(hipMemcpy(cb, hc.data(), CH * kSlab, hipMemcpyHostToDevice)); }
    CK(hipMalloc(&out, 256 * 256 * 4));
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e9f;
        for (int i = 0; i < 20; ++i) { CK(hipEventRecord(e0)); hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(256), 0, 0, out); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best; }
        printf("empty kernel (event floor)   best %.1f us\n", best * 1e3f);
    }
    if (run3<true>("v3: all", xs, cb, out)) return 1;
    if (run3<false>("v3: intake only", xs, cb, out)) return 1;
    if (run<true, true, true>("tokens + codebook + MFMA", xs, cb, out)) return 1;
    if (run<true, true, false>("tokens + codebook", xs, cb, out)) return 1;
    if (run<true, false, false>("tokens only", xs, cb, out)) return 1;
    if (run<false, true, false>("codebook only", xs, cb, out)) return 1;
    if (run<false, false, true>("MFMA only", xs, cb, out)) return 1;
    if (run<false, true, true>("codebook + MFMA", xs, cb, out)) return 1;
    if (run<true, false, true>("tokens + MFMA", xs, cb, out)) return 1;
    return 0;
}
