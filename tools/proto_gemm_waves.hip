// Probe (round 5): the stage loop of the split-fp16 GCN product (csrc/sn_gcn.hip: 128 x 256 tile, k stages of 16 through a three-slot
// LDS ring filled by LDS-DMA, two workgroups per CU) with FOUR waves per workgroup (each 64 x 128 = 2 x 4 accumulators: the shipped
// shape) against EIGHT (each 64 x 64 = 2 x 2 accumulators, 128 registers: four waves per SIMD instead of two), on the shape of the
// bench's class product (100 graphs, 512 x 512 . 512 x 256) and of config [3]'s (1000 graphs).  Main loop + plain fp32 store, no
// epilogue.  The shipped loop runs as the SUM of its copy skeleton and its MFMA phase: does more waves per SIMD overlap them?
// Measured (one MI355X): G = 100: 51.0 us with four waves, 50.2 with eight (0.79 / 0.80 PFLOP/s issued; 46.8 / 46.4 without the
// stores); G = 1000: 437 against 486 us (0.92 / 0.83), without the stores 447 / 399.  Bit-identical results.  NO: occupancy is not the lever.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/pgw tools/proto_gemm_waves.hip && /tmp/pgw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kStage = 24 * 1024, kRing = 3;      // A: 4 row blocks x (hi, lo) = 8 KiB; B: 8 column blocks x (hi, lo) = 16 KiB

// planes: A [G][m/32][k/16][1 KiB] hi and lo; B [G][n/32][k/16][1 KiB] hi and lo.  C [G][m][n] fp32.  One workgroup = rows tm*128.., all 256 columns.
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void gemm_waves(const unsigned char *a_hi, const unsigned char *a_lo, const unsigned char *b_hi,
                                                            const unsigned char *b_lo, float *c, int G, int m, int n, int k, int tiles_m, int store)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int WN = WAVES / 2, NJ = 8 / WN, kDma = 24 / WAVES;          // column tiles per wave; copies per wave and stage
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int wm = wid / WN, wn = wid % WN;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int g = (slot / tiles_m) * 8 + xcd, tm = slot % tiles_m;
    const int kb = k / 16;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem;
    // copies of a stage: block c < 8: A row block c >> 1, plane c & 1; c >= 8: B column block (c - 8) >> 1, plane c & 1
    const unsigned char *src[kDma];
#pragma unroll
    for (int i = 0; i < kDma; ++i) {
        const int cc = wid * kDma + i;
        const bool is_b = cc >= 8;
        const int t = (is_b ? cc - 8 : cc) >> 1, plane = cc & 1;
        const unsigned char *base = is_b ? (plane ? b_lo : b_hi) : (plane ? a_lo : a_hi);
        const size_t rows_blocks = is_b ? (size_t)(n / 32) : (size_t)(m / 32);
        const size_t rb = is_b ? (size_t)t : (size_t)tm * 4 + t;
        src[i] = base + ((size_t)g * rows_blocks + rb) * kb * 1024 + lane * 16;
    }
    auto issue = [&](int s) {
        const unsigned sl = (unsigned)(s % kRing);
#pragma unroll
        for (int i = 0; i < kDma; ++i) {
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + sl * kStage + (wid * kDma + i) * 1024);
            unsigned keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src[i] + (size_t)s * 1024), "s"(dst) : "memory");
        }
    };
    f32x16 acc[2][NJ];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    if (g < G) {
        issue(0);
        if (kb > 1) issue(1);
        for (int s = 0; s < kb; ++s) {
            if (s + 1 < kb) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kDma) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (s + 2 < kb) issue(s + 2);
            const unsigned char *sa = smem + (s % kRing) * kStage, *sb = sa + 8 * 1024;
            half8 ah[2], al[2], bh[NJ], bl[NJ];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const half8 *>(sa + ((2 * wm + i) * 2 + 0) * 1024 + lane * 16);
                al[i] = *reinterpret_cast<const half8 *>(sa + ((2 * wm + i) * 2 + 1) * 1024 + lane * 16);
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                bh[j] = *reinterpret_cast<const half8 *>(sb + ((NJ * wn + j) * 2 + 0) * 1024 + lane * 16);
                bl[j] = *reinterpret_cast<const half8 *>(sb + ((NJ * wn + j) * 2 + 1) * 1024 + lane * 16);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
        float *cg = c + (size_t)g * m * n;
        const int r = lane & 31, h = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int row = tm * 128 + (2 * wm + i) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h, col = (NJ * wn + j) * 32 + r;
                    if (store || acc[i][j][q] == 12345.678f) cg[(size_t)row * n + col] = acc[i][j][q];
                }
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int WAVES>
static int run(int G, int m, int n, int k, const unsigned char *ah, const unsigned char *al, const unsigned char *bh, const unsigned char *bl, float *c)
{
    const int lds = kRing * kStage, tiles_m = m / 128;
    CK(hipFuncSetAttribute((const void *)gemm_waves<WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const unsigned grid = 8u * ((G + 7) / 8) * tiles_m;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(gemm_waves<WAVES>, dim3(grid), dim3(WAVES * 64), lds, 0, ah, al, bh, bl, c, G, m, n, k, tiles_m, 1);
    CK(hipDeviceSynchronize());
    for (int store = 1; store >= 0; --store) {
        float best = 1e9f;
        for (int i = 0; i < 7; ++i) {
            CK(hipEventRecord(e0));
            hipLaunchKernelGGL(gemm_waves<WAVES>, dim3(grid), dim3(WAVES * 64), lds, 0, ah, al, bh, bl, c, G, m, n, k, tiles_m, store);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
        }
        const double issued = 3.0 * 2.0 * G * (double)m * n * k;
        printf("%d waves per workgroup: G=%d m=%d n=%d k=%d store=%d: %.1f us, issued %.2f PFLOP/s\n", WAVES, G, m, n, k, store, best * 1e3, issued / (best * 1e-3) / 1e15);
    }
    return 0;
}

int main()
{
    const int Gmax = 1000, m = 512, n = 256, k = 512;
    const size_t abytes = (size_t)Gmax * m * k * 2, bbytes = (size_t)Gmax * n * k * 2, cbytes = (size_t)Gmax * m * n * 4;
    unsigned char *ah, *al, *bh, *bl; float *c, *c2;
    CK(hipMalloc(&ah, abytes)); CK(hipMalloc(&al, abytes)); CK(hipMalloc(&bh, bbytes)); CK(hipMalloc(&bl, bbytes)); CK(hipMalloc(&c, cbytes)); CK(hipMalloc(&c2, cbytes));
    {
        std::vector<unsigned short> h(abytes / 2);
        for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x7FF));      // fp16 in [0.125, 0.25)
        CK(hipMemcpy(ah, h.data(), abytes, hipMemcpyHostToDevice)); CK(hipMemcpy(al, h.data(), abytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(bh, h.data(), bbytes, hipMemcpyHostToDevice)); CK(hipMemcpy(bl, h.data() + 12345, bbytes, hipMemcpyHostToDevice));
    }
    for (int G : {100, 1000}) {
        if (run<4>(G, m, n, k, ah, al, bh, bl, c)) return 1;
        if (run<8>(G, m, n, k, ah, al, bh, bl, c2)) return 1;
        std::vector<float> x((size_t)G * m * n), y(x.size());
        CK(hipMemcpy(x.data(), c, x.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(y.data(), c2, y.size() * 4, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < x.size(); ++i) bad += x[i] != y[i];
        printf("G=%d: %zu of %zu elements differ between the two forms; c[0] = %g\n", G, bad, x.size(), (double)x[0]);
    }
    return 0;
}
