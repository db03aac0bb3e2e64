// Micro-benchmark 2: achieved issue rate of the real idct8 butterfly (compiler-scheduled) at several
// occupancies.  Prints ns per wave-iteration per SIMD; combine with the static instruction count of
// the loop body (tools: hipcc -S) for cycles/instruction.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
struct TDiv {
    int x; unsigned sb;
    __device__ __forceinline__ explicit TDiv(int v) : x(v), sb((unsigned)v >> 31) {}
    __device__ __forceinline__ int d2() const { return (int)((unsigned)x + sb) >> 1; }
    __device__ __forceinline__ int d4() const { return (int)((unsigned)x + sb * 3u) >> 2; }
    __device__ __forceinline__ int d16() const { return (int)((unsigned)x + sb * 15u) >> 4; }
};
__device__ __forceinline__ void idct8(int (&v)[8])
{
    int c0 = v[0], d4 = v[1], d6 = v[3], c1 = v[4], d5 = v[5], d7 = v[7];
    TDiv c2(v[2]), c3(v[6]);
    int c4 = d4, c5 = wadd(d5, d6), c7 = wsub(d5, d6), c6 = d7;
    TDiv b4(wadd(c4, c5)), b5(wsub(c4, c5)), b6(wadd(c6, c7)), b7(wsub(c6, c7));
    int b0 = wadd(c0, c1), b1 = wsub(c0, c1);
    int b2 = wadd(wadd(c2.x, c2.d4()), c3.d2());
    int b3 = wsub(wsub(c2.d2(), c3.x), c3.d4());
    int b4q = b4.d4(), b7q = b7.d4();
    int a4 = wsub(wadd(wadd(b7q, b4.x), b4q), b4.d16());
    int a7 = wadd(wsub(wsub(b4q, b7.x), b7q), b7.d16());
    int a5 = wadd(wadd(wsub(b5.x, b6.x), b6.d4()), b6.d16());
    int a6 = wsub(wsub(wadd(b6.x, b5.x), b5.d4()), b5.d16());
    int a0 = wadd(b0, b2), a1 = wadd(b1, b3), a2 = wsub(b1, b3), a3 = wsub(b0, b2);
    v[0] = wadd(a0, a4); v[1] = wadd(a1, a5); v[2] = wadd(a2, a6); v[3] = wadd(a3, a7);
    v[4] = wsub(a3, a7); v[5] = wsub(a2, a6); v[6] = wsub(a1, a5); v[7] = wsub(a0, a4);
}
constexpr int ITER = 4000;
template <int NARR, int LDSPAD>
__global__ __launch_bounds__(256) void k_bfly(int *out, int seed)
{
    __shared__ int pad[LDSPAD];
    if (seed == 12345) pad[threadIdx.x] = seed;   // keep the allocation alive
    int v[NARR][8];
    for (int s = 0; s < NARR; s++) for (int k = 0; k < 8; k++) v[s][k] = (threadIdx.x * 17 + s * 5 + k * 3 + seed) & 0xffff;
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int s = 0; s < NARR; s++) {
            idct8(v[s]);
#pragma unroll
            for (int k = 0; k < 8; k++) v[s][k] >>= 1;   // keep the values bounded (8 cheap VOP2 ops)
        }
    }
    int acc = 0;
    for (int s = 0; s < NARR; s++) for (int k = 0; k < 8; k++) acc += v[s][k];
    out[blockIdx.x * 256 + threadIdx.x] = acc + (seed == 12345 ? pad[0] : 0);
}
template <int NARR, int LDSPAD>
void run(const char *name, int *out, int cus, int blocks_per_cu)
{
    int blocks = cus * blocks_per_cu;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_bfly<NARR, LDSPAD>), dim3(blocks), dim3(256), 0, 0, out, 1);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((k_bfly<NARR, LDSPAD>), dim3(blocks), dim3(256), 0, 0, out, 1);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
    double idct_per_simd = (double)ITER * NARR * blocks_per_cu;   // 4 waves per block -> 1 wave per SIMD per block
    printf("%-28s waves/SIMD %d  %8.3f ms  %7.2f ns per idct8(+8 shifts) per SIMD = %6.1f cycles @2.4GHz\n", name, blocks_per_cu,
           ms, ms * 1e6 / idct_per_simd, ms * 1e6 / idct_per_simd * 2.4);
}
int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    int *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    run<1, 64>("1 array/lane", out, cus, 8);
    run<2, 64>("2 arrays/lane", out, cus, 8);
    run<4, 64>("4 arrays/lane", out, cus, 8);
    run<2, 64>("2 arrays/lane", out, cus, 4);
    run<2, 64>("2 arrays/lane", out, cus, 2);
    run<2, 64>("2 arrays/lane", out, cus, 1);
    run<4, 64>("4 arrays/lane", out, cus, 1);
    return 0;
}
