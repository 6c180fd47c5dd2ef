// ABI housekeeping: version, thread-local error string, device probe.
#include "sn_common.h"

#include <string.h>

#include <mutex>
#include <vector>

static thread_local char g_err[512] = "";

void sn_set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int sn_abi_version(void) { return 12; }

extern "C" const char *sn_last_error(void) { return g_err; }

extern "C" int sn_device_ok(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        (void)hipGetLastError();
        return 0;
    }
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return strncmp(prop.gcnArchName, "gfx950", 6) == 0 ? 1 : 0;
}

// ------------------------------------------------------------------------------------------
// per-device state: dynamic-LDS opt-in of a kernel, CU count
// ------------------------------------------------------------------------------------------
namespace {
std::mutex g_dev_mutex;
struct LdsKey { int dev; const void *fn; size_t bytes; };
std::vector<LdsKey> g_lds_done;
int g_cus[64] = {0};
}  // namespace

int sn_ensure_dynamic_lds(const void *fn, size_t bytes, const char *name)
{
    if (bytes > 160 * 1024) {
        sn_set_error("%s: needs %zu bytes of LDS (> 160 KiB)", name, bytes);
        return SN_ERR_UNSUPPORTED;
    }
    if (bytes <= 64 * 1024) return SN_OK;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) { sn_set_error("%s: hipGetDevice failed", name); return SN_ERR_LAUNCH; }
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    for (const LdsKey &k : g_lds_done)
        if (k.dev == dev && k.fn == fn && k.bytes >= bytes) return SN_OK;
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) {
        sn_set_error("%s: cannot raise dynamic LDS to %zu: %s", name, bytes, hipGetErrorString(e));
        return SN_ERR_LAUNCH;
    }
    g_lds_done.push_back(LdsKey{dev, fn, bytes});
    return SN_OK;
}

namespace {
__global__ void zero_words_kernel(unsigned *p, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace

int sn_zero_async(void *ptr, size_t bytes, hipStream_t st)
{
    if (bytes == 0) return SN_OK;
    if (!ptr || (bytes & 3) != 0 || (reinterpret_cast<uintptr_t>(ptr) & 3) != 0) {
        sn_set_error("sn_zero_async: pointer / size not 4-byte aligned");
        return SN_ERR_BAD_ARG;
    }
    const size_t n_words = bytes / 4;
    const unsigned blocks = (unsigned)((n_words + 255) / 256 < 1024 ? (n_words + 255) / 256 : 1024);
    hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, st, (unsigned *)ptr, n_words);
    SN_CHECK_LAUNCH("sn_zero_async");
    return SN_OK;
}

// ------------------------------------------------------------------------------------------
// A captured memset node is not reliable on this ROCm (see sn_common.h, sn_zero_async): in a graph that PyTorch
// captured (train.GraphedTrainIter: forward + backward + optimizer) the library's own memsets - the semaphores of its
// multi-block reductions, the zero fill under embedding_dense_backward - are memset nodes.  This walks a captured
// hipGraph_t BEFORE it is instantiated and puts a kernel node with the same predecessors and successors in the place of
// every one-dimensional memset node.
// ------------------------------------------------------------------------------------------
namespace {
__global__ void fill_words_kernel(unsigned *p, unsigned pattern, size_t n_words)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (size_t)gridDim.x * blockDim.x) p[i] = pattern;
}
__global__ void fill_elems_kernel(unsigned char *p, unsigned value, unsigned elem_size, size_t n_elems)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * blockDim.x)
        for (unsigned b = 0; b < elem_size; ++b) p[i * elem_size + b] = (unsigned char)(value >> (8 * b));
}
}  // namespace

extern "C" int sn_graph_replace_memsets(void *graph, int *n_replaced, int *n_left)
{
    SN_REQUIRE(graph, SN_ERR_BAD_ARG, "sn_graph_replace_memsets: NULL graph");
    hipGraph_t g = (hipGraph_t)graph;
    size_t n = 0;
    hipError_t e = hipGraphGetNodes(g, nullptr, &n);
    SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphGetNodes: %s", hipGetErrorString(e));
    std::vector<hipGraphNode_t> nodes(n);
    if (n) {
        e = hipGraphGetNodes(g, nodes.data(), &n);
        SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphGetNodes: %s", hipGetErrorString(e));
    }
    int done = 0, left = 0;
    for (size_t i = 0; i < n; ++i) {
        hipGraphNodeType type;
        e = hipGraphNodeGetType(nodes[i], &type);
        SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphNodeGetType: %s", hipGetErrorString(e));
        if (type != hipGraphNodeTypeMemset) continue;
        hipMemsetParams mp;
        e = hipGraphMemsetNodeGetParams(nodes[i], &mp);
        SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphMemsetNodeGetParams: %s", hipGetErrorString(e));
        if (mp.height > 1 || (mp.elementSize != 1 && mp.elementSize != 2 && mp.elementSize != 4) || !mp.dst) { ++left; continue; }
        if (mp.width == 0) { ++left; continue; }
        size_t n_in = 0, n_out = 0;
        e = hipGraphNodeGetDependencies(nodes[i], nullptr, &n_in);
        SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphNodeGetDependencies: %s", hipGetErrorString(e));
        std::vector<hipGraphNode_t> in(n_in);
        if (n_in) {
            e = hipGraphNodeGetDependencies(nodes[i], in.data(), &n_in);
            SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphNodeGetDependencies: %s", hipGetErrorString(e));
        }
        e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &n_out);
        SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphNodeGetDependentNodes: %s", hipGetErrorString(e));
        std::vector<hipGraphNode_t> out(n_out);
        if (n_out) {
            e = hipGraphNodeGetDependentNodes(nodes[i], out.data(), &n_out);
            SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphNodeGetDependentNodes: %s", hipGetErrorString(e));
        }
        const size_t bytes = mp.width * mp.elementSize;
        unsigned pattern = mp.value;
        if (mp.elementSize == 1) { pattern &= 0xFFu; pattern |= pattern << 8; pattern |= pattern << 16; }
        else if (mp.elementSize == 2) { pattern &= 0xFFFFu; pattern |= pattern << 16; }
        hipKernelNodeParams kp;
        memset(&kp, 0, sizeof(kp));
        void *dst = mp.dst;
        size_t count;
        unsigned value = mp.value, esz = mp.elementSize;
        void *args_w[3] = {&dst, &pattern, &count};
        void *args_e[4] = {&dst, &value, &esz, &count};
        if ((reinterpret_cast<uintptr_t>(dst) & 3) == 0 && (bytes & 3) == 0) {
            count = bytes / 4;
            kp.func = (void *)fill_words_kernel;
            kp.kernelParams = args_w;
        } else {
            count = mp.width;
            kp.func = (void *)fill_elems_kernel;
            kp.kernelParams = args_e;
        }
        const size_t blocks = (count + 255) / 256;
        kp.gridDim = dim3((unsigned)(blocks < 2048 ? blocks : 2048));
        kp.blockDim = dim3(256);
        hipGraphNode_t kn;
        e = hipGraphAddKernelNode(&kn, g, n_in ? in.data() : nullptr, n_in, &kp);
        SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphAddKernelNode: %s", hipGetErrorString(e));
        for (size_t j = 0; j < n_out; ++j) {
            e = hipGraphAddDependencies(g, &kn, &out[j], 1);
            SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphAddDependencies: %s", hipGetErrorString(e));
        }
        e = hipGraphDestroyNode(nodes[i]);
        SN_REQUIRE(e == hipSuccess, SN_ERR_LAUNCH, "sn_graph_replace_memsets: hipGraphDestroyNode: %s", hipGetErrorString(e));
        ++done;
    }
    if (n_replaced) *n_replaced = done;
    if (n_left) *n_left = left;
    return SN_OK;
}

int sn_device_cus(void)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    std::lock_guard<std::mutex> lock(g_dev_mutex);
    if (g_cus[dev] == 0) {
        hipDeviceProp_t prop;
        g_cus[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    }
    return g_cus[dev];
}

// ------------------------------------------------------------------------------------------
// per-kernel event timing (bench.py roofline): events are recorded on the SAME stream as the
// kernel, immediately before / after its launch.
// ------------------------------------------------------------------------------------------
namespace {
struct Prof {
    int cap = 0;
    int n_start[SN_PROF_KERNELS] = {0};
    int n_stop[SN_PROF_KERNELS] = {0};
    hipEvent_t *ev[SN_PROF_KERNELS][2] = {{nullptr}};
} g_prof;
}  // namespace

void sn_prof_start(int k, hipStream_t st)
{
    if (g_prof.cap == 0 || g_prof.n_start[k] >= g_prof.cap) return;
    (void)hipEventRecord(g_prof.ev[k][0][g_prof.n_start[k]++], st);
}

void sn_prof_stop(int k, hipStream_t st)
{
    if (g_prof.cap == 0 || g_prof.n_stop[k] >= g_prof.cap) return;
    (void)hipEventRecord(g_prof.ev[k][1][g_prof.n_stop[k]++], st);
}

extern "C" int sn_profile_enable(int max_samples)
{
    SN_REQUIRE(max_samples >= 0 && max_samples <= (1 << 20), SN_ERR_BAD_ARG, "sn_profile_enable: max_samples=%d", max_samples);
    for (int k = 0; k < SN_PROF_KERNELS; ++k) {
        for (int s = 0; s < 2; ++s) {
            if (g_prof.ev[k][s]) {
                for (int i = 0; i < g_prof.cap; ++i) (void)hipEventDestroy(g_prof.ev[k][s][i]);
                delete[] g_prof.ev[k][s];
                g_prof.ev[k][s] = nullptr;
            }
        }
        g_prof.n_start[k] = g_prof.n_stop[k] = 0;
    }
    g_prof.cap = 0;
    if (max_samples == 0) return SN_OK;
    for (int k = 0; k < SN_PROF_KERNELS; ++k)
        for (int s = 0; s < 2; ++s) {
            g_prof.ev[k][s] = new hipEvent_t[max_samples];
            for (int i = 0; i < max_samples; ++i)
                if (hipEventCreate(&g_prof.ev[k][s][i]) != hipSuccess) {
                    sn_set_error("sn_profile_enable: hipEventCreate failed");
                    return SN_ERR_LAUNCH;
                }
        }
    g_prof.cap = max_samples;
    return SN_OK;
}

extern "C" int sn_profile_count(int kernel_id)
{
    if (kernel_id < 0 || kernel_id >= SN_PROF_KERNELS) return 0;
    return g_prof.n_stop[kernel_id];
}

extern "C" int sn_profile_elapsed_ms(int kernel_id, float *out_ms_host, int n)
{
    SN_REQUIRE(kernel_id >= 0 && kernel_id < SN_PROF_KERNELS && out_ms_host, SN_ERR_BAD_ARG, "sn_profile_elapsed_ms: bad argument");
    SN_REQUIRE(n >= 0 && n <= g_prof.n_stop[kernel_id], SN_ERR_BAD_ARG, "sn_profile_elapsed_ms: only %d samples", g_prof.n_stop[kernel_id]);
    for (int i = 0; i < n; ++i) {
        if (hipEventSynchronize(g_prof.ev[kernel_id][1][i]) != hipSuccess ||
            hipEventElapsedTime(&out_ms_host[i], g_prof.ev[kernel_id][0][i], g_prof.ev[kernel_id][1][i]) != hipSuccess) {
            sn_set_error("sn_profile_elapsed_ms: event query failed");
            return SN_ERR_LAUNCH;
        }
    }
    return SN_OK;
}
