// Probe 2: the S1 main loop in isolation (no DMA, no barrier): 24-MFMA chains alternating between two
// accumulators, A fragments through a ring of 8 ds_read_b128, B fragments in 96 registers, one key
// (4 VALU) per MFMA gap on the idle accumulator.  MODE selects what is switched off.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MODE bit0: keys on; bit1: keys read the idle accumulator (else a plain register); bit2: A from LDS ring
// (else constant); bit3: distinct B per step (else constant); bit4: re-initialise the idle accumulator
template <int MODE>
__global__ __launch_bounds__(512, 2) void loop_probe(unsigned long long *out, int tiles, float seed)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 25600 / 4; i += blockDim.x) ((unsigned *)lds)[i] = 0x2e002e00u + (i & 255);
    __syncthreads();
    half8 b[24];
    for (int s = 0; s < 24; ++s)
        for (int j = 0; j < 8; ++j) b[s][j] = (_Float16)(seed * (lane + j + s));
    f32x16 accA, accB;
    for (int j = 0; j < 16; ++j) { accA[j] = seed; accB[j] = seed + j; }
    unsigned m1[4], m2[4], m3[4];
    for (int g = 0; g < 4; ++g) m1[g] = m2[g] = m3[g] = 0xFFFFFFFFu;
    unsigned keymask = 0xFFFFFF00u;
    asm volatile("" : "+v"(keymask));
    float plain = seed * lane;
    asm volatile("" : "+v"(plain));
    half8 ar[8];
    auto frag_at = [&](int step) { return *reinterpret_cast<const half8 *>(lds + step * 1024 + lane * 16); };
    for (int q = 0; q < 8; ++q) ar[q] = frag_at(q);
    auto key_insert = [&](float v, unsigned code, int g) {
        unsigned k;
        asm volatile("v_and_or_b32 %0, %4, %5, %6\n\tv_med3_u32 %3, %0, %2, %3\n\tv_med3_u32 %2, %0, %1, %2\n\tv_min_u32 %1, %0, %1"
                     : "=&v"(k), "+v"(m1[g]), "+v"(m2[g]), "+v"(m3[g]) : "v"(v), "v"(keymask), "s"(code));
    };
    auto tile_step = [&](int w, f32x16 &cur, f32x16 &oth) {
        const unsigned code0 = ((unsigned)(w - 1) & 63u) << 2;
#pragma unroll
        for (int s = 0; s < 24; ++s) {
            cur = __builtin_amdgcn_mfma_f32_32x32x16_f16(ar[(MODE & 4) ? s % 8 : 0], b[(MODE & 8) ? s : 0], cur, 0, 0, 0);
            if (MODE & 4) ar[s % 8] = frag_at((s + 8) % 24);
            if ((MODE & 1) && s >= 2 && s < 18) key_insert((MODE & 2) ? oth[s - 2] : plain, code0 | (unsigned)(s & 3), (s - 2) >> 2);
            if ((MODE & 16) && s >= 16 && s < 20) {
                const float4 c4 = *reinterpret_cast<const float4 *>(lds + 24 * 1024 + ((s - 16) * 2 + (lane >> 5)) * 16);
                oth[4 * (s - 16) + 0] = c4.x + plain; oth[4 * (s - 16) + 1] = c4.y + plain;
                oth[4 * (s - 16) + 2] = c4.z + plain; oth[4 * (s - 16) + 3] = c4.w + plain;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int w = 0; w < tiles; w += 2) {
        tile_step(w, accA, accB);
        tile_step(w + 1, accB, accA);
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sum = 0.0f;
    for (int j = 0; j < 16; ++j) sum += accA[j] + accB[j];
    unsigned x = 0;
    for (int g = 0; g < 4; ++g) x ^= m1[g] ^ m2[g] ^ m3[g];
    if (sum == 123.456f || x == 0x12345u) out[4096] = 1;
    if (lane == 0) out[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

static int g_grid = 1;
template <int MODE>
void run(unsigned long long *d, const char *what)
{
    const int tiles = 512;
    unsigned long long h[8];
    (void)hipFuncSetAttribute((const void *)loop_probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 26 * 1024);
    loop_probe<MODE><<<g_grid, 512, 26 * 1024>>>(d, 16, 0.001f);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    loop_probe<MODE><<<g_grid, 512, 26 * 1024>>>(d, tiles, 0.001f);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    unsigned long long mx = 0;
    for (int w = 0; w < 8; ++w) mx = h[w] > mx ? h[w] : mx;
    printf("grid %3d mode %2d (%s): %.1f ticks per MFMA per wave (floor 64); kernel %.1f us -> %.2f G ticks/s\n", g_grid, MODE, what,
           (double)mx / (tiles * 24.0), ms * 1e3, (double)mx / (ms * 1e6));
}

int main()
{
    unsigned long long *d;
    (void)hipMalloc(&d, 8192 * 8);
    for (g_grid = 1; g_grid <= 256; g_grid *= 16) {
        run<0>(d, "MFMA only, constant operands");
        run<12>(d, "A ring from LDS, distinct B");
        run<15>(d, "+ keys on the idle accumulator");
        run<31>(d, "+ idle accumulator re-initialised (full loop)");
    }
    return 0;
}
