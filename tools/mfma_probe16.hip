// Probe: how does v_mfma_f32_16x16x32_f16 round?  Compares the device result of a chain of
// NCHAIN MFMAs (fp16 inputs, fp32 accumulate, C initialised to a large value) with the exact
// real sum (long double; inputs chosen so it is exact) and with two rounding models.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe16.hip -o gpurun_out/mfma_probe16
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int NCHAIN = 12;

__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D)
{   // A [NCHAIN][16][32], B [NCHAIN][32][16], C/D [16][16]
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f32x4 acc;
    for (int reg = 0; reg < 4; ++reg) acc[reg] = C[(4 * g + reg) * 16 + r];
    for (int s = 0; s < NCHAIN; ++s) {
        half8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = A[(s * 16 + r) * 32 + 8 * g + j]; b[j] = B[(s * 32 + 8 * g + j) * 16 + r]; }
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 4; ++reg) D[(4 * g + reg) * 16 + r] = acc[reg];
}

static double rnd() { return (double)rand() / RAND_MAX; }

int main()
{
    const int nA = NCHAIN * 16 * 32, nB = NCHAIN * 32 * 16;
    for (int test = 0; test < 4; ++test) {
        std::vector<_Float16> A(nA), B(nB);
        std::vector<float> C(256), D(256);
        const double cscale = test == 0 ? 0.0 : (test == 1 ? 600.0 : (test == 2 ? 60000.0 : 1.0));
        srand(1234 + test);
        for (auto &v : A) v = (_Float16)((float)((rnd() * 2 - 1) * (test == 3 ? 30.0 : 3.0)));
        for (auto &v : B) v = (_Float16)((float)((rnd() * 2 - 1) * (test == 3 ? 30.0 : 3.0)));
        for (auto &v : C) v = (float)(cscale * (0.5 + rnd()));
        _Float16 *dA, *dB; float *dC, *dD;
        hipMalloc(&dA, nA * 2); hipMalloc(&dB, nB * 2); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), nA * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), nB * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        double worst_ulp = 0, worst_rel = 0, worst_chain = 0, worst_instr = 0;
        for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
            long double exact = C[i * 16 + j];
            long double maxpart = fabsl(exact);
            float chain = C[i * 16 + j];      // model 1: fmaf chain, one rounding per product
            float instr = C[i * 16 + j];      // model 2: exact 32-product sum + C, one rounding per MFMA
            for (int s = 0; s < NCHAIN; ++s) {
                long double blk = 0;
                for (int kk = 0; kk < 32; ++kk) {
                    const double p = (double)(float)A[(s * 16 + i) * 32 + kk] * (double)(float)B[(s * 32 + kk) * 16 + j];
                    exact += p; blk += p;
                    chain = fmaf((float)A[(s * 16 + i) * 32 + kk], (float)B[(s * 32 + kk) * 16 + j], chain);
                    if (fabsl(exact) > maxpart) maxpart = fabsl(exact);
                }
                instr = (float)((long double)instr + blk);
            }
            const double d = D[i * 16 + j];
            const double ulp = ldexp(1.0, ilogb((double)fabsl(exact)) - 23);
            worst_ulp = fmax(worst_ulp, fabs(d - (double)exact) / ulp);
            worst_rel = fmax(worst_rel, fabs(d - (double)exact) / ((double)maxpart * ldexp(1.0, -24)));
            worst_chain = fmax(worst_chain, fabs(d - chain) / ulp);
            worst_instr = fmax(worst_instr, fabs(d - instr) / ulp);
        }
        printf("test %d (C~%g): max |dev-exact| = %.3f ulp(result) = %.3f x 2^-24*max|partial| ; vs fmaf-chain model %.3f ulp ; "
               "vs one-rounding-per-MFMA model %.3f ulp  [%d products]\n",
               test, cscale, worst_ulp, worst_rel, worst_chain, worst_instr, NCHAIN * 32);
    }
    return 0;
}
