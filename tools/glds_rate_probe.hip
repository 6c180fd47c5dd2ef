// Probe: sustained LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction) and plain
// global_load_dwordx4 throughput per CU from an L2-resident buffer, 4 or 8 waves per CU, whole chip.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: LDS-DMA, own M0 per instruction; 1: LDS-DMA, one M0 + 4 immediate offsets; 2: global_load_dwordx4 to VGPRs
__global__ __launch_bounds__(512) void probe(const unsigned char *src, unsigned long long *out, int iters, float *sink)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char *)smem + wid * 8192;
    const unsigned char *p = src + ((size_t)(blockIdx.x * 8 + wid) % 64) * 65536 + lane * 16;   // 4 MiB window, L2 resident
    f32x4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const unsigned char *q = p + (it & 7) * 8192;
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(q + j * 1024), "s"(lds_base + j * 1024) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; j += 4) {
                unsigned keep;
                asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                             "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %1, off offset:2048\n\t"
                             "global_load_lds_dwordx4 %1, off offset:3072\n\ts_mov_b32 m0, %0"
                             : "=&s"(keep) : "v"(q + j * 1024), "s"(lds_base + j * 1024) : "memory");
            }
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        } else {
            f32x4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = *reinterpret_cast<const f32x4 *>(q + j * 1024);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += v[j];
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (acc.x == 123.0f) sink[0] = acc.y + acc.z + acc.w;
    if (lane == 0) out[blockIdx.x * 8 + wid] = t1 - t0;
}

template <int MODE>
void run(const unsigned char *src, unsigned long long *d, float *sink, int threads, int grid, const char *what)
{
    const int iters = 400;
    static unsigned long long h[4096];
    (void)hipFuncSetAttribute((const void *)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    probe<MODE><<<grid, threads, 65536>>>(src, d, 20, sink);
    probe<MODE><<<grid, threads, 65536>>>(src, d, iters, sink);
    (void)hipMemcpy(h, d, sizeof(unsigned long long) * grid * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 0;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < threads / 64; ++w) { sum += (double)h[b * 8 + w]; ++n; }
    const double cyc = sum / n;                     // cycles per wave for iters * 8 KiB
    printf("%-44s %d waves/CU grid %3d: %.1f cycles per 1 KiB instruction per wave -> %.1f B/clk/CU\n", what, threads / 64, grid,
           cyc / (iters * 8.0), (threads / 64) * iters * 8192.0 / cyc);
}

int main()
{
    unsigned char *src; unsigned long long *d; float *sink;
    (void)hipMalloc(&src, 8 << 20); (void)hipMemset(src, 1, 8 << 20);
    (void)hipMalloc(&d, 4096 * 8); (void)hipMalloc(&sink, 16);
    for (int grid = 1; grid <= 256; grid *= 256)
        for (int threads = 256; threads <= 512; threads += 256) {
            run<0>(src, d, sink, threads, grid, "LDS-DMA, M0 per instruction");
            run<1>(src, d, sink, threads, grid, "LDS-DMA, M0 per 4 (immediate offsets)");
            run<2>(src, d, sink, threads, grid, "global_load_dwordx4 -> VGPR");
        }
    return 0;
}
