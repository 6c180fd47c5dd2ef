// Feasibility probe for a second form of the split-fp16 GCN product: one workgroup of four 512-register waves per CU,
// a 256 x 256 tile of C (each wave 128 x 128: sixteen 32 x 32 accumulators = the 256 AGPRs), k stages of 16 through a four-slot
// LDS ring filled by LDS-DMA (blocked planes: 1 KiB = 32 rows x 16 k, hi and lo planes of A and B: 32 KiB per stage), per stage and
// wave 16 ds_read_b128 and 48 MFMAs (hi.hi + hi.lo + lo.hi).  The shipped kernel's waves hold 64 x 128 and read 512 B of LDS per MFMA
// with 8 waves per CU - the LDS's whole bandwidth at full matrix rate; this form reads 341 B per MFMA with 4 waves.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pg tools/proto_gemm256.hip && /tmp/pg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kStage = 32 * 1024, kRing = 4;

// planes: A [G][m/32][k/16][1 KiB] hi and lo; B [G][n/32][k/16][1 KiB] hi and lo.  C [G][m][n] fp32.
__global__ __launch_bounds__(256, 1) void gemm256(const unsigned char *a_hi, const unsigned char *a_lo, const unsigned char *b_hi, const unsigned char *b_lo,
                                                  float *c, int m, int n, int k, int tiles_m, int tiles_n, int store)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wm = wid >> 1, wn = wid & 1;
    const int tile = blockIdx.x % (tiles_m * tiles_n), g = blockIdx.x / (tiles_m * tiles_n);
    const int tm = tile / tiles_n, tn = tile % tiles_n;
    const int kb = k / 16;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // copies of a stage: 32 blocks of 1 KiB: A hi rows 0..7 (row blocks of the tile), A lo 8..15, B hi 16..23, B lo 24..31; wave w copies 8 w .. + 7
    const unsigned char *src[8];
    unsigned voff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int blk = wid * 8 + i, which = blk >> 3, rb = blk & 7;
        const unsigned char *base = which == 0 ? a_hi : (which == 1 ? a_lo : (which == 2 ? b_hi : b_lo));
        const size_t rows_blocks = which < 2 ? (size_t)(m / 32) : (size_t)(n / 32);
        const size_t rbg = which < 2 ? (size_t)tm * 8 + rb : (size_t)tn * 8 + rb;
        src[i] = base + ((size_t)g * rows_blocks + rbg) * kb * 1024;
        voff[i] = (unsigned)(lane * 16);
    }
    auto issue = [&](int s) {
        const unsigned slot = (unsigned)(s % kRing);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + slot * kStage + (wid * 8 + i) * 1024);
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff[i] + (unsigned)s * 1024u), "s"(src[i]), "s"(dst) : "memory");
        }
    };
    f32x16 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    issue(0); issue(1); issue(2);
    const unsigned a_lane = (unsigned)((wm * 4) * 1024 + lane * 16), b_lane = (unsigned)((16 + wn * 4) * 1024 + lane * 16);
    for (int s = 0; s < kb; ++s) {
        // outstanding (oldest first): stage s, s + 1, s + 2 (8 copies each)
        if (s + 2 < kb) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (s + 1 < kb) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (s + 3 < kb) issue(s + 3);
        const unsigned char *st = smem + (s % kRing) * kStage;
        half8 ah[4], al[4], bh[4], bl[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ah[i] = *reinterpret_cast<const half8 *>(st + a_lane + i * 1024);
            al[i] = *reinterpret_cast<const half8 *>(st + a_lane + (8 + i) * 1024);
            bh[i] = *reinterpret_cast<const half8 *>(st + b_lane + i * 1024);
            bl[i] = *reinterpret_cast<const half8 *>(st + b_lane + (8 + i) * 1024);
        }
        // (the three products of a tile are a dependent chain: all sixteen tiles of one product first)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(ah[i]), "v"(bh[j]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(ah[i]), "v"(bl[j]));
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(al[i]), "v"(bh[j]));
        asm volatile("s_nop 7" :: "v"(ah[0]), "v"(ah[1]), "v"(ah[2]), "v"(ah[3]), "v"(al[0]), "v"(al[1]), "v"(al[2]), "v"(al[3]));
        asm volatile("s_nop 7" :: "v"(bh[0]), "v"(bh[1]), "v"(bh[2]), "v"(bh[3]), "v"(bl[0]), "v"(bl[1]), "v"(bl[2]), "v"(bl[3]));
    }
    for (int i = 0; i < 6; ++i) asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float *cg = c + (size_t)g * m * n;
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = tm * 256 + wm * 128 + i * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, col = tn * 256 + wn * 128 + j * 32 + r;
                if (store || acc[i][j][q] == 12345.678f) cg[(size_t)row * n + col] = acc[i][j][q];
            }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    const int G = 500, m = 512, n = 1024, k = 512;
    const size_t abytes = (size_t)G * m * k * 2, bbytes = (size_t)G * n * k * 2, cbytes = (size_t)G * m * n * 4;
    unsigned char *ah, *al, *bh, *bl; float *c;
    CK(hipMalloc(&ah, abytes)); CK(hipMalloc(&al, abytes)); CK(hipMalloc(&bh, bbytes)); CK(hipMalloc(&bl, bbytes)); CK(hipMalloc(&c, cbytes));
    {
        std::vector<unsigned short> h(bbytes / 2);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x7FF));      // fp16 in [0.125, 0.25)
        CK(hipMemcpy(ah, h.data(), abytes, hipMemcpyHostToDevice)); CK(hipMemcpy(al, h.data(), abytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(bh, h.data(), bbytes, hipMemcpyHostToDevice)); CK(hipMemcpy(bl, h.data(), bbytes, hipMemcpyHostToDevice));
    }
    const int lds = kRing * kStage;
    CK(hipFuncSetAttribute((const void *)gemm256, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int tiles_m = m / 256, tiles_n = n / 256;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(gemm256, dim3(G * tiles_m * tiles_n), dim3(256), lds, 0, ah, al, bh, bl, c, m, n, k, tiles_m, tiles_n, 1);
    CK(hipDeviceSynchronize());
    for (int store = 1; store >= 0; --store) {
        float best = 1e9f;
        for (int i = 0; i < 5; ++i) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(gemm256, dim3(G * tiles_m * tiles_n), dim3(256), lds, 0, ah, al, bh, bl, c, m, n, k, tiles_m, tiles_n, store);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        const double issued = 3.0 * 2.0 * G * (double)m * n * k;
        printf("G=%d m=%d n=%d k=%d store=%d: %.3f ms, issued %.2f PFLOP/s (useful %.2f)\n", G, m, n, k, store, best, issued / (best * 1e-3) / 1e15, issued / 3 / (best * 1e-3) / 1e15);
    }
    return 0;
}
