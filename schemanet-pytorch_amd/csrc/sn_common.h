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

// ---------------------------------------------------------------- host: per-device state
// A process may drive several GPUs: function attributes and device properties are kept per device (and the tables
// behind these two are guarded by a mutex: entry points may be called from several host threads).
// Raises the dynamic-LDS limit of kernel `fn` to `bytes` on the current device, once per (device, kernel, size class).
int sn_ensure_dynamic_lds(const void *fn, size_t bytes, const char *name);
// Compute units of the current device.
int sn_device_cus(void);
// Zero `bytes` (a multiple of 4) at a 4-byte aligned device address, as a KERNEL on `st`.  Used instead of
// hipMemsetAsync wherever a launch sequence may be captured into a hipGraph: a captured memset node was seen NOT to clear
// its 32 bytes on replay (ROCm 7.2, a graph captured after graphs of another topology: the S1 work counters kept the
// allocator's leftovers and the re-rank walked a garbage overflow list - a GPU memory fault); a kernel node is ordered
// and executed like every other node of the sequence.
int sn_zero_async(void *ptr, size_t bytes, hipStream_t st);

// ---------------------------------------------------------------- host: per-kernel event timing
void sn_prof_start(int kernel_id, hipStream_t st);
void sn_prof_stop(int kernel_id, hipStream_t st);

// ---------------------------------------------------------------- device: wave reductions
// 64-lane all-reduce on the DPP network: four row-local steps (row_mirror, row_half_mirror,
// quad mirror, quad xor-1) leave every lane with its 16-lane row total, v_readlane combines the
// four rows.  ~11 instructions and no LDS crossbar (ds_bpermute) round trips.  Every lane ends
// with the same value.  (The fp64 re-rank of S1 keeps the xor butterfly below: its summation
// order is part of the oracle contract.)
#define SN_DPP_F32(v, ctrl) __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), (ctrl), 0xF, 0xF, false))
#define SN_READLANE_F32(v, l) __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (l)))   // the builtin is int -> int

__device__ __forceinline__ float sn_wave_sum(float v)
{
    v += SN_DPP_F32(v, 0x140);      // row_mirror        i <-> 15 - i
    v += SN_DPP_F32(v, 0x141);      // row_half_mirror   i <-> 7 - i inside each half row
    v += SN_DPP_F32(v, 0x1B);       // quad_perm [3,2,1,0]
    v += SN_DPP_F32(v, 0xB1);       // quad_perm [1,0,3,2]
    const float r0 = SN_READLANE_F32(v, 0), r1 = SN_READLANE_F32(v, 16);
    const float r2 = SN_READLANE_F32(v, 32), r3 = SN_READLANE_F32(v, 48);
    return (r0 + r1) + (r2 + r3);
}

// LayerNorm + optional ReLU of one row held by a wave: v[k] = element lane + 64 k of the row (0 beyond E; all 0 for a
// masked row, reference gnn.py:43-46) -> normalised values in place.  ONE definition for every kernel that normalises
// rows (sn_mask_layernorm_act and the fused forms that never store the result): same sums in the same order, bit for bit.
constexpr int SN_LN_MAX = 16;           // E <= 64 * 16
// gm / bt: gamma and beta at the lane's columns (sn_layernorm_coeffs), loaded once per wave.
__device__ __forceinline__ void sn_layernorm_coeffs(float (&gm)[SN_LN_MAX], float (&bt)[SN_LN_MAX], int lane, int E, const float *gamma,
                                                    const float *beta)
{
#pragma unroll
    for (int k = 0; k < SN_LN_MAX; ++k) {
        const int c = lane + SN_WAVE * k;
        gm[k] = c < E ? gamma[c] : 0.0f;
        bt[k] = c < E ? beta[c] : 0.0f;
    }
}

__device__ __forceinline__ void sn_layernorm_row(float (&v)[SN_LN_MAX], int lane, int E, const float (&gm)[SN_LN_MAX],
                                                 const float (&bt)[SN_LN_MAX], float eps, int relu)
{
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < SN_LN_MAX; ++k) s += v[k];
    const float mean = sn_wave_sum(s) / (float)E;
    float q = 0.0f;
#pragma unroll
    for (int k = 0; k < SN_LN_MAX; ++k) {
        const int c = lane + SN_WAVE * k;
        const float d = (c < E) ? v[k] - mean : 0.0f;
        q += d * d;
    }
    const float rstd = 1.0f / sqrtf(sn_wave_sum(q) / (float)E + eps);
#pragma unroll
    for (int k = 0; k < SN_LN_MAX; ++k) {
        const int c = lane + SN_WAVE * k;
        if (c < E) {
            float y = (v[k] - mean) * rstd * gm[k] + bt[k];
            if (relu) y = fmaxf(y, 0.0f);
            v[k] = y;
        }
    }
}

// fp64 xor butterfly (off = 32, 16, 8, 4, 2, 1): the summation order of the oracle's sno_dot64.
// Same values as six __shfl_xor steps, but moved with v_permlane32/16_swap and DPP instead of
// ds_bpermute (checked bit for bit on MI355X: the S1 parity tests compare every index with the oracle).  xor 8 / xor 4 use
// row_ror:8 / row_ror:4: after the previous step the values have period 16 / 8 inside a row, so
// the rotated lane holds exactly the value of the xor lane.
__device__ __forceinline__ double sn_f64_from(unsigned lo, unsigned hi) { return __hiloint2double((int)hi, (int)lo); }

#define SN_DPP_U32(v, ctrl) ((unsigned)__builtin_amdgcn_mov_dpp((int)(v), (ctrl), 0xF, 0xF, false))

__device__ __forceinline__ double sn_wave_sum_f64(double v)
{
    const int lane = threadIdx.x & 63;
    unsigned lo = (unsigned)__double2loint(v), hi = (unsigned)__double2hiint(v);
    {
        const auto a = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
        v += sn_f64_from(lane < 32 ? a[1] : a[0], lane < 32 ? b[1] : b[0]);
    }
    lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
    {
        const auto a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
        const auto b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
        v += sn_f64_from((lane & 16) ? a[0] : a[1], (lane & 16) ? b[0] : b[1]);
    }
    lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
    v += sn_f64_from(SN_DPP_U32(lo, 0x128), SN_DPP_U32(hi, 0x128));      // row_ror:8  == xor 8
    lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
    v += sn_f64_from(SN_DPP_U32(lo, 0x124), SN_DPP_U32(hi, 0x124));      // row_ror:4  == xor 4
    lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
    v += sn_f64_from(SN_DPP_U32(lo, 0x4E), SN_DPP_U32(hi, 0x4E));        // quad_perm [2,3,0,1] == xor 2
    lo = (unsigned)__double2loint(v); hi = (unsigned)__double2hiint(v);
    v += sn_f64_from(SN_DPP_U32(lo, 0xB1), SN_DPP_U32(hi, 0xB1));        // quad_perm [1,0,3,2] == xor 1
    return v;
}

__device__ __forceinline__ float sn_wave_max(float v)
{
    v = fmaxf(v, SN_DPP_F32(v, 0x140));
    v = fmaxf(v, SN_DPP_F32(v, 0x141));
    v = fmaxf(v, SN_DPP_F32(v, 0x1B));
    v = fmaxf(v, SN_DPP_F32(v, 0xB1));
    const float r0 = SN_READLANE_F32(v, 0), r1 = SN_READLANE_F32(v, 16);
    const float r2 = SN_READLANE_F32(v, 32), r3 = SN_READLANE_F32(v, 48);
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

__device__ __forceinline__ float sn_wave_min(float v)
{
    v = fminf(v, SN_DPP_F32(v, 0x140));
    v = fminf(v, SN_DPP_F32(v, 0x141));
    v = fminf(v, SN_DPP_F32(v, 0x1B));
    v = fminf(v, SN_DPP_F32(v, 0xB1));
    const float r0 = SN_READLANE_F32(v, 0), r1 = SN_READLANE_F32(v, 16);
    const float r2 = SN_READLANE_F32(v, 32), r3 = SN_READLANE_F32(v, 48);
    return fminf(fminf(r0, r1), fminf(r2, r3));
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
