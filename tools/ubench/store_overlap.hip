// Micro-benchmark: do vector stores overlap with VALU work on gfx950, or do their costs add?
// Every wavefront repeats ITER times: VALU_N dependent-free integer ops on 8 register chains, then STORES 16-byte stores
// per lane (1 KiB per wave-instruction, consecutive lanes -> consecutive addresses).  Three kernels: VALU only, stores
// only, both.  5 waves per SIMD resident (launch bounds + grid), like k_enc_pframe.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 64;

template <bool VALU, bool STORE, int VALU_N, int STORES>
__global__ __launch_bounds__(256) void k(uint4 *out, unsigned a, size_t per_wave_u4)
{
    unsigned r[8];
    for (int j = 0; j < 8; j++) r[j] = threadIdx.x * (j + 1) + a;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    uint4 *dst = out + wave * per_wave_u4 + (threadIdx.x & 63);
    for (int it = 0; it < ITER; it++) {
        if (VALU) {
#pragma unroll
            for (int n = 0; n < VALU_N / 8; n++)
#pragma unroll
                for (int j = 0; j < 8; j++) r[j] = __builtin_amdgcn_udot4(r[j], a, r[(j + 1) & 7], false);
        }
        if (STORE) {
#pragma unroll
            for (int s = 0; s < STORES; s++)
                dst[(size_t)(it * STORES + s) * 64] = make_uint4(r[0], r[1], r[2], r[3] + s);
        }
    }
    if (!STORE) { unsigned s = 0; for (int j = 0; j < 8; j++) s += r[j]; if (s == 0x12345) out[wave].x = s; }
}
template <bool V, bool S, int VN, int SN> static float run(uint4 *out, int blocks, size_t per_wave)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<V, S, VN, SN>), dim3(blocks), dim3(256), 0, 0, out, 3u, per_wave);
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<V, S, VN, SN>), dim3(blocks), dim3(256), 0, 0, out, 3u, per_wave);
    CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1000.0f;
}
template <int VN, int SN> static void experiment()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 5 * 8;        // 8 rounds of 5 workgroups (20 waves) per CU
    const size_t per_wave = (size_t)ITER * SN * 64;
    uint4 *out; CHECK(hipMalloc(&out, (size_t)blocks * 4 * per_wave * 16 + 4096));
    float v = run<true, false, VN, SN>(out, blocks, per_wave), s = run<false, true, VN, SN>(out, blocks, per_wave),
          b = run<true, true, VN, SN>(out, blocks, per_wave);
    double mb = (double)blocks * 4 * per_wave * 16 / 1e6;
    printf("VALU %4d ops + %d stores per iteration: valu %.0f us, stores %.0f us (%.0f MB, %.2f TB/s), both %.0f us  (sum %.0f, max %.0f)\n",
           VN, SN, v, s, mb, mb / s / 1e6 * 1e6 / 1e6, b, v + s, v > s ? v : s);
    CHECK(hipFree(out));
}
int main()
{
    experiment<64, 1>(); experiment<128, 1>(); experiment<256, 1>(); experiment<512, 1>(); experiment<256, 2>(); experiment<1024, 4>();
    return 0;
}
