// Sparsity terms of the training loss: row entropies of the normalised atlas.
// replaces: entropy(p) = -sum(p * log(p + 1e-7), dim=-1)   schema_inference/loss/schema_inference_loss.py:51-58
// (called on class_vertices [K, n] and class_edges [K, n, n]: 105 MB at K = 100, n = 512; the reference's
// torch expression makes three passes over it and keeps two temporaries for the backward).
// One wave per row: forward reads p once; the backward recomputes log(p + eps) and touches only the rows
// whose upstream gradient is not zero (the loss takes a max over rows, so all but K of the K n rows are).
#include "sn_common.h"

namespace {

__global__ __launch_bounds__(256) void row_entropy_kernel(const float *p, int64_t rows, int n, float eps, float *ent)
{
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    const bool vec = (n & 3) == 0 && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += n_waves) {
        const float *row = p + r * n;
        float acc = 0.0f;
        if (vec) {
            for (int j = lane * 4; j < n; j += SN_WAVE * 4) {
                const float4 v = *reinterpret_cast<const float4 *>(row + j);
                acc += v.x * logf(v.x + eps);
                acc += v.y * logf(v.y + eps);
                acc += v.z * logf(v.z + eps);
                acc += v.w * logf(v.w + eps);
            }
        } else {
            for (int j = lane; j < n; j += SN_WAVE) acc += row[j] * logf(row[j] + eps);
        }
        acc = sn_wave_sum(acc);
        if (lane == 0) ent[r] = -acc;
    }
}

// grad_p[r][j] = -g[r] * (log(p + eps) + p / (p + eps))
__global__ __launch_bounds__(256) void row_entropy_backward_kernel(const float *p, const float *g, int64_t rows, int n, float eps, float *grad_p)
{
    const int lane = threadIdx.x & 63;
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6); r < rows; r += n_waves) {
        const float gr = g[r];
        const float *row = p + r * n;
        float *out = grad_p + r * n;
        if (gr == 0.0f) {                                            // wave-uniform
            for (int j = lane; j < n; j += SN_WAVE) out[j] = 0.0f;
        } else {
            for (int j = lane; j < n; j += SN_WAVE) {
                const float v = row[j], q = v + eps;
                out[j] = -gr * (logf(q) + v / q);
            }
        }
    }
}

}  // namespace

extern "C" int sn_row_entropy(const float *p, int64_t rows, int n, float eps, float *entropy, void *stream)
{
    SN_REQUIRE(rows >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_row_entropy: bad rows=%lld n=%d", (long long)rows, n);
    if (rows == 0) return SN_OK;
    SN_REQUIRE(p && entropy, SN_ERR_BAD_ARG, "sn_row_entropy: NULL pointer");
    const int64_t blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(row_entropy_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, p, rows, n, eps, entropy);
    SN_CHECK_LAUNCH("sn_row_entropy");
    return SN_OK;
}

extern "C" int sn_row_entropy_backward(const float *p, const float *grad_entropy, int64_t rows, int n, float eps, float *grad_p, void *stream)
{
    SN_REQUIRE(rows >= 0 && n > 0, SN_ERR_BAD_ARG, "sn_row_entropy_backward: bad rows=%lld n=%d", (long long)rows, n);
    if (rows == 0) return SN_OK;
    SN_REQUIRE(p && grad_entropy && grad_p, SN_ERR_BAD_ARG, "sn_row_entropy_backward: NULL pointer");
    const int64_t blocks = (rows + 3) / 4;
    hipLaunchKernelGGL(row_entropy_backward_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, (hipStream_t)stream, p,
                       grad_entropy, rows, n, eps, grad_p);
    SN_CHECK_LAUNCH("sn_row_entropy_backward");
    return SN_OK;
}
