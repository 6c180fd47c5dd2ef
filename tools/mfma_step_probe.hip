// Probe: one "step" of the K-outer S1 screen - request the next raw rows from LDS (2 x ds_read_b128), four 32 x 32 x 16 MFMAs on
// four different accumulators, then 13 VALU (squares + fp32 -> fp16) on the rows - in a loop, no barriers, no copies:
// cycles per step for one and for two waves per SIMD, with the VALU block, the LDS reads or both removed.
// hipcc --offload-arch=gfx950 -O3 -o mfma_step_probe.bin tools/mfma_step_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>      // bit 0: LDS reads, bit 1: VALU block, bit 2: reload of the A fragments from LDS in every third step
__global__ __launch_bounds__(512, 2) void probe(float *out, unsigned long long *cyc, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[64 * 1024];
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) reinterpret_cast<float *>(smem)[i] = 0.001f * (float)(i & 1023);
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    f32x16 acc[12];
    for (int i = 0; i < 12; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = (float)(threadIdx.x + i + r);
    half8 a[4], b;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 8; ++e) a[i][e] = (_Float16)(0.001f * (threadIdx.x + i + e));
    for (int e = 0; e < 8; ++e) b[e] = (_Float16)(0.002f * (threadIdx.x + e));
    float sumsq = 0.0f;
    // (rows of 128 bytes, piece slot = piece ^ ((row >> 1) & 7): the screen's conflict-free layout; MODE bit 3: unswizzled = 8-way conflicts)
    const int r_ = lane & 31, h_ = lane >> 5, sw_ = (MODE & 8) ? 0 : (r_ >> 1) & 7;
    const unsigned base = (unsigned)(r_ * 128 + (((4 * h_) ^ sw_) << 4) + (w & 3) * 4096);
    const unsigned abase = (unsigned)(lane * 16 + 32768 + (w & 3) * 4096);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int st = 0; st < 3; ++st) {
            f32x4 lo = {0, 0, 0, 0}, hi = {0, 0, 0, 0};
            if constexpr (MODE & 1) {
                lo = *reinterpret_cast<const f32x4 *>(smem + (base ^ (unsigned)((st & 1) * 32)) + (st >> 1) * 8192);
                hi = *reinterpret_cast<const f32x4 *>(smem + (base ^ (unsigned)((st & 1) * 32 + 16)) + (st >> 1) * 8192);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                acc[4 * st + v] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[v], b, acc[4 * st + v], 0, 0, 0);
                if constexpr (MODE & 4) { if (st == 2) a[v] = *reinterpret_cast<const half8 *>(smem + abase + v * 1024); }
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (MODE & 2) {
                float q = lo.x * lo.x;
                q = fmaf(lo.y, lo.y, q); q = fmaf(lo.z, lo.z, q); q = fmaf(lo.w, lo.w, q);
                q = fmaf(hi.x, hi.x, q); q = fmaf(hi.y, hi.y, q); q = fmaf(hi.z, hi.z, q); q = fmaf(hi.w, hi.w, q);
                sumsq += q;
                asm volatile("" : "+v"(sumsq));
                b[0] = (_Float16)lo.x; b[1] = (_Float16)lo.y; b[2] = (_Float16)lo.z; b[3] = (_Float16)lo.w;
                b[4] = (_Float16)hi.x; b[5] = (_Float16)hi.y; b[6] = (_Float16)hi.z; b[7] = (_Float16)hi.w;
            } else if constexpr (MODE & 1) {
                b = __builtin_bit_cast(half8, lo);
                asm volatile("" :: "v"(hi));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = sumsq;
    for (int i = 0; i < 12; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + w] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int threads)
{
    float *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&cyc, 256 * 8 * 8);
    const int iters = 100;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((probe<MODE>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
    (void)hipDeviceSynchronize();
    unsigned long long h[2048];
    (void)hipMemcpy(h, cyc, 256 * (threads / 64) * 8, hipMemcpyDeviceToHost);
    double sum = 0; int n = 256 * (threads / 64);
    for (int i = 0; i < n; ++i) sum += (double)h[i];
    const double per_step = sum / n / ((double)iters * 3);
    printf("%-52s %d waves/SIMD: %.0f cycles per step per wave (4 MFMAs = 128 of matrix pipe) -> pipe busy %.0f %%\n", name, threads / 256, per_step,
           100.0 * 128.0 * (threads / 256) / per_step);
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0>("MFMAs only", 256); run<0>("MFMAs only", 512);
    run<1>("+ LDS rows (no VALU)", 256); run<1>("+ LDS rows (no VALU)", 512);
    run<2>("+ VALU block (no LDS)", 256); run<2>("+ VALU block (no LDS)", 512);
    run<3>("+ LDS rows + VALU block", 256); run<3>("+ LDS rows + VALU block", 512);
    run<7>("+ LDS rows + VALU block + A reload every 3rd step", 256); run<7>("+ LDS rows + VALU block + A reload every 3rd step", 512);
    run<11>("LDS rows with 8-way bank conflicts + VALU block", 256); run<11>("LDS rows with 8-way bank conflicts + VALU block", 512);
    return 0;
}
