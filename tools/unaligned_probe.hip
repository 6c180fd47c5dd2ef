// GPU probe: are global_load_dwordx4 / global_store_dwordx4 at addresses that are only 4-byte aligned served correctly on gfx950?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float *src, float *dst, int off, int n4)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    f32x4 v;
    const float *p = src + off + 4 * i;
    asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    dst[4 * i + 0] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
}
int main()
{
    const int n4 = 1 << 16, n = 4 * n4 + 16;
    std::vector<float> h(n);
    for (int i = 0; i < n; ++i) h[i] = (float)i;
    float *s, *d;
    hipMalloc(&s, n * 4); hipMalloc(&d, n * 4);
    hipMemcpy(s, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int off = 0; off < 4; ++off) {
        hipMemset(d, 0, n * 4);
        hipLaunchKernelGGL(k, dim3(n4 / 256), dim3(256), 0, 0, s, d, off, n4);
        hipError_t e = hipDeviceSynchronize();
        std::vector<float> o(n);
        hipMemcpy(o.data(), d, n * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 4 * n4; ++i) bad += o[i] != (float)(i + off);
        printf("offset %d floats: %s, %d mismatches\n", off, hipGetErrorString(e), bad);
    }
    return 0;
}
