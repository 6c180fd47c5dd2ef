// Probe: how does v_mfma_f32_32x32x16_f16 round?  Compares the device result of a chain of
// NCHAIN MFMAs (fp16 inputs, fp32 accumulate, C initialised to a large value) with the exact
// real sum (long double; inputs chosen so it is exact) and with two rounding models.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o gpurun_out/mfma_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int NCHAIN = 24;

__global__ void k(const _Float16 *A, const _Float16 *B, const float *C, float *D)
{   // A [NCHAIN][32][16], B [NCHAIN][16][32], C/D [32][32]
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    f32x16 acc;
    for (int reg = 0; reg < 16; ++reg) acc[reg] = C[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r];
    for (int s = 0; s < NCHAIN; ++s) {
        half8 a, b;
        for (int j = 0; j < 8; ++j) { a[j] = A[(s * 32 + r) * 16 + 8 * h + j]; b[j] = B[(s * 16 + 8 * h + j) * 32 + r]; }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    }
    for (int reg = 0; reg < 16; ++reg) D[((reg & 3) + 8 * (reg >> 2) + 4 * h) * 32 + r] = acc[reg];
}

static double rnd() { return (double)rand() / RAND_MAX; }

int main()
{
    const int nA = NCHAIN * 32 * 16, nB = NCHAIN * 16 * 32;
    for (int test = 0; test < 4; ++test) {
        std::vector<_Float16> A(nA), B(nB);
        std::vector<float> C(1024), D(1024);
        const double cscale = test == 0 ? 0.0 : (test == 1 ? 600.0 : (test == 2 ? 60000.0 : 1.0));
        srand(1234 + test);
        for (auto &v : A) v = (_Float16)((float)((rnd() * 2 - 1) * (test == 3 ? 30.0 : 3.0)));
        for (auto &v : B) v = (_Float16)((float)((rnd() * 2 - 1) * (test == 3 ? 30.0 : 3.0)));
        for (auto &v : C) v = (float)(cscale * (0.5 + rnd()));
        _Float16 *dA, *dB; float *dC, *dD;
        hipMalloc(&dA, nA * 2); hipMalloc(&dB, nB * 2); hipMalloc(&dC, 4096); hipMalloc(&dD, 4096);
        hipMemcpy(dA, A.data(), nA * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), nB * 2, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 4096, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        double worst_ulp = 0, worst_rel = 0, worst_chain = 0, worst_instr = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            long double exact = C[i * 32 + j];
            long double maxpart = fabsl(exact);
            float chain = C[i * 32 + j];      // model 1: fmaf chain, one rounding per product
            float instr = C[i * 32 + j];      // model 2: exact 16-product sum + C, one rounding per MFMA
            for (int s = 0; s < NCHAIN; ++s) {
                long double blk = 0;
                for (int kk = 0; kk < 16; ++kk) {
                    const double p = (double)(float)A[(s * 32 + i) * 16 + kk] * (double)(float)B[(s * 16 + kk) * 32 + j];
                    exact += p; blk += p;
                    chain = fmaf((float)A[(s * 32 + i) * 16 + kk], (float)B[(s * 16 + kk) * 32 + j], chain);
                    if (fabsl(exact) > maxpart) maxpart = fabsl(exact);
                }
                instr = (float)((long double)instr + blk);
            }
            const double d = D[i * 32 + j];
            const double ulp = ldexp(1.0, ilogb((double)fabsl(exact)) - 23);
            worst_ulp = fmax(worst_ulp, fabs(d - (double)exact) / ulp);
            worst_rel = fmax(worst_rel, fabs(d - (double)exact) / ((double)maxpart * ldexp(1.0, -24)));
            worst_chain = fmax(worst_chain, fabs(d - chain) / ulp);
            worst_instr = fmax(worst_instr, fabs(d - instr) / ulp);
        }
        printf("test %d (C~%g): max |dev-exact| = %.3f ulp(result) = %.3f x 2^-24*max|partial| ; vs fmaf-chain model %.3f ulp ; "
               "vs one-rounding-per-MFMA model %.3f ulp  [%d products]\n",
               test, cscale, worst_ulp, worst_rel, worst_chain, worst_instr, NCHAIN * 16);
    }
    return 0;
}
