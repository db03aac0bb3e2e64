// Micro-benchmark: do LDS reads at byte addresses that are not multiples of the access size work on gfx950, and at what
// cost?  Each lane reads `W` bytes at  lane * stride + off  (off = 0..3) in a dependent-free loop; prints correctness
// and ns per wave-instruction per CU relative to the aligned case.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 4000;

template <int W>
__global__ __launch_bounds__(256) void k_read(unsigned *out, int off, int stride)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 48 + 64];
    for (int i = threadIdx.x; i < (int)sizeof(lds); i += 256) lds[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned addr = threadIdx.x * stride + off;
    unsigned acc = 0;
    for (int it = 0; it < ITER; it++) {
        unsigned a = addr + (it & 1) * 16;   // two alternating addresses so the loads are not hoisted
        if (W == 4) {
            unsigned v;
            asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v;
        } else if (W == 8) {
            unsigned long long v;
            asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += (unsigned)v + (unsigned)(v >> 32);
        } else {
            uint4 v;
            asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(a) : "memory");
            acc += v.x + v.y + v.z + v.w;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}
// one read each, results written out for checking
template <int W>
__global__ void k_check(unsigned char *out, int off, int stride)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[256 * 48 + 64];
    for (int i = threadIdx.x; i < (int)sizeof(lds); i += 256) lds[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const unsigned a = threadIdx.x * stride + off;
    unsigned v[4] = {0, 0, 0, 0};
    if (W == 4) asm volatile("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v[0]) : "v"(a) : "memory");
    else if (W == 8) { unsigned long long t; asm volatile("ds_read_b64 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a) : "memory"); v[0] = (unsigned)t; v[1] = (unsigned)(t >> 32); }
    else { uint4 t; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(a) : "memory"); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    memcpy(out + threadIdx.x * 16, v, 16);
}
template <int W> static void run(int stride)
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * 4;
    unsigned *out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    unsigned char *chk; CHECK(hipMalloc(&chk, 256 * 16));
    unsigned char host[256 * 16];
    for (int off = 0; off < 4; off++) {
        hipLaunchKernelGGL(k_check<W>, dim3(1), dim3(256), 0, 0, chk, off, stride);
        CHECK(hipMemcpy(host, chk, sizeof host, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int t = 0; t < 256; t++)
            for (int b = 0; b < W; b++)
                bad += host[t * 16 + b] != (unsigned char)((t * stride + off + b) * 7 + 3);
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k_read<W>, dim3(blocks), dim3(256), 0, 0, out, off, stride);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k_read<W>, dim3(blocks), dim3(256), 0, 0, out, off, stride);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        // per CU: 4 blocks x 4 waves x ITER wave-instructions
        printf("W=%2d stride=%2d off=%d  wrong_bytes=%d  %.2f ns per wave-instruction per CU\n", W, stride, off, bad, ms * 1e6 / (16.0 * ITER));
    }
    CHECK(hipFree(out)); CHECK(hipFree(chk));
}
int main()
{
    run<4>(4); run<4>(44); run<8>(8); run<8>(44); run<16>(16); run<16>(44);
    return 0;
}
