// Register-allocation probe for the codebook-stationary S1 screen: 384 VGPR/AGPRs of A fragments
// per wave, tokens streamed.  Compile only:
//   hipcc --offload-arch=gfx950 -O3 -c tools/proto_screen2.hip -Rpass-analysis=kernel-resource-usage
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NT, int KS, int NAGPR>
__global__ __launch_bounds__(256, 1) void proto(const unsigned char *packed, const unsigned char *hnp, unsigned *out, int n_sets)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    half8 A[NT][KS];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int j = 0; j < KS; ++j)
            A[a][j] = *reinterpret_cast<const half8 *>(packed + ((size_t)((wid * NT + a) * KS + j)) * 1024 + lane * 16);
    f32x4 hn[NT];
#pragma unroll
    for (int a = 0; a < NT; ++a) hn[a] = *reinterpret_cast<const f32x4 *>(hnp + (size_t)(wid * NT + a) * 1024 + lane * 16);
    // pin the register file split: NA fragments + the half norms in AGPRs, the rest in VGPRs
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            if (a * KS + j < NAGPR) asm volatile("" : "+a"(A[a][j]));
            else asm volatile("" : "+v"(A[a][j]));
        }
#pragma unroll
    for (int a = 0; a < NT; ++a) asm volatile("" : "+a"(hn[a]));
    float m1 = 3e38f, m2 = 3e38f, m3 = 3e38f;
    unsigned keymask = 0xFFFFFF00u;
    asm volatile("" : "+v"(keymask));
    for (int s = 0; s < n_sets; ++s) {
        const unsigned char *bbase = smem + (s & 1) * (KS * 1024) + lane * 16;
#pragma unroll
        for (int q = 0; q < NT / 4; ++q) {
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = hn[q * 4 + t];
#pragma unroll
            for (int j = 0; j < KS; ++j) {
                const half8 b = *reinterpret_cast<const half8 *>(bbase + j * 1024);
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(A[q * 4 + t][j], b, acc[t], 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned code = (unsigned)((q * 4 + t) << 2 | r);
                    const float k = __uint_as_float((__float_as_uint(acc[t][r]) & keymask) | code);
                    m3 = __builtin_amdgcn_fmed3f(k, m2, m3);
                    m2 = __builtin_amdgcn_fmed3f(k, m1, m2);
                    m1 = fminf(k, m1);
                }
        }
        __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = __float_as_uint(m1) ^ __float_as_uint(m2) ^ __float_as_uint(m3);
}

template __global__ void proto<8, 12, 52>(const unsigned char *, const unsigned char *, unsigned *, int);
