// Probe: cycles per v_mfma_f32_32x32x16_f16 on one SIMD when consecutive MFMAs use DIFFERENT accumulators (a K-outer kernel that
// keeps NACC accumulator tiles per wave), accumulators in architectural VGPRs or in AGPRs, one or two waves per SIMD.
// hipcc --offload-arch=gfx950 -O3 -o mfma_acc_probe tools/mfma_acc_probe.hip && ./mfma_acc_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, bool AG, int SAMEB>
__global__ __launch_bounds__(512, 2) void probe(float *out, unsigned long long *cyc, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
    half8 a[4], b[3];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.001f * (threadIdx.x + i + e));
    for (int i = 0; i < 3; ++i) for (int e = 0; e < 8; ++e) b[i][e] = (_Float16)(0.002f * (threadIdx.x + 2 * i + e));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (AG) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a[i & 3]), "v"(b[SAMEB ? 0 : (i / 4) % 3]));
            else asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a[i & 3]), "v"(b[SAMEB ? 0 : (i / 4) % 3]));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC, bool AG, int SAMEB>
static void run(const char *name, int threads)
{
    float *out; unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<NACC, AG, SAMEB>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[2048];
    hipMemcpy(h, cyc, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 256 * (threads / 64);
    for (int i = 0; i < n; ++i) sum += (double)h[i];
    const double per_wave = sum / n / ((double)iters * NACC);
    printf("%-44s threads %3d: %.1f cycles per MFMA per wave = %.1f per SIMD\n", name, threads, per_wave, per_wave / (threads / 256.0));
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<12, false, 0>("12 accumulators, VGPR", 256);
    run<12, false, 0>("12 accumulators, VGPR", 512);
    run<12, true, 0>("12 accumulators, AGPR", 256);
    run<12, true, 0>("12 accumulators, AGPR", 512);
    run<1, false, 0>("1 accumulator (dependent chain), VGPR", 256);
    run<1, false, 0>("1 accumulator (dependent chain), VGPR", 512);
    run<2, false, 0>("2 accumulators, VGPR", 256);
    run<2, false, 0>("2 accumulators, VGPR", 512);
    run<4, false, 0>("4 accumulators, VGPR", 512);
    run<4, true, 0>("4 accumulators, AGPR", 512);
    return 0;
}
