// Shared host/device helpers for libschemanet_hip.so (gfx950 only; wave = 64).
#pragma once

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "schemanet_hip.h"

#define SN_WAVE 64

// ---------------------------------------------------------------- host: error reporting
void sn_set_error(const char *fmt, ...);

#define SN_REQUIRE(cond, code, ...)                                                         \
    do {                                                                                    \
        if (!(cond)) {                                                                      \
            sn_set_error(__VA_ARGS__);                                                      \
            return (code);                                                                  \
        }                                                                                   \
    } while (0)

#define SN_CHECK_LAUNCH(name)                                                               \
    do {                                                                                    \
        hipError_t e_ = hipGetLastError();                                                  \
        if (e_ != hipSuccess) {                                                             \
            sn_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));             \
            return SN_ERR_LAUNCH;                                                           \
        }                                                                                   \
    } while (0)

// ---------------------------------------------------------------- host: per-kernel event timing
void sn_prof_start(int kernel_id, hipStream_t st);
void sn_prof_stop(int kernel_id, hipStream_t st);

// ---------------------------------------------------------------- device: wave reductions
// xor-butterfly: every lane ends with the same value (fp add / max are commutative).
__device__ __forceinline__ float sn_wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, SN_WAVE);
    return v;
}

__device__ __forceinline__ double sn_wave_sum_f64(double v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, SN_WAVE);
    return v;
}

__device__ __forceinline__ float sn_wave_max(float v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, SN_WAVE));
    return v;
}

__device__ __forceinline__ int sn_wave_sum_i(int v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, SN_WAVE);
    return v;
}

// at::nan_to_num_(x, 0): nan -> 0, +-inf -> +-FLT_MAX  (graph/utils.py:12, large_scale_feat_to_v.cpp:124)
__device__ __forceinline__ float sn_nan_to_num(float v)
{
    if (v != v) return 0.0f;
    if (v == INFINITY) return 3.402823466e+38f;
    if (v == -INFINITY) return -3.402823466e+38f;
    return v;
}
