// Does v_mfma_f32_32x32x16_f16 flush fp16 subnormal inputs?  A = 2^-20 (subnormal), B = 2^10.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float av, float bv)
{
    half8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)0.0f; b[j] = (_Float16)0.0f; }
    a[0] = (_Float16)av; b[0] = (_Float16)bv;
    f32x16 acc;
    for (int j = 0; j < 16; ++j) acc[j] = 0.0f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; }
}
int main()
{
    float *d, h[2];
    (void)hipMalloc(&d, 8);
    const float as[4] = {9.5367431640625e-07f /*2^-20*/, 5.9604644775390625e-08f /*2^-24*/, 6.103515625e-05f /*2^-14 normal*/, 3.0517578125e-05f /*2^-15*/};
    for (int i = 0; i < 4; ++i) {
        k<<<1, 64>>>(d, as[i], 1024.0f);
        (void)hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a = %g (as fp16 %g)  x 1024 (two lanes' k=0 and k=8 contribute): mfma = %g, expected %g\n", as[i], h[1], h[0], 2.0 * as[i] * 1024.0);
    }
    return 0;
}
