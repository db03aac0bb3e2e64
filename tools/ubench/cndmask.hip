// Micro-benchmark: what does v_cndmask_b32 cost on gfx950, alone and in the v_cmp + v_cndmask pairs real code has?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 2000;
#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define BODY(str) asm volatile(REP8(str) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b), "s"(mask) : "vcc")
#define KERNEL(name, S, N) \
__global__ __launch_bounds__(256) void name(unsigned *out, unsigned a, unsigned b, unsigned long long mask) { \
    unsigned r[8]; for (int k = 0; k < 8; k++) r[k] = threadIdx.x * (k + 1) + a; \
    asm volatile("v_cmp_gt_u32 vcc, %0, %1" :: "v"(a), "v"(r[0]) : "vcc"); \
    for (int it = 0; it < ITER; it++) { BODY(S); BODY(S); BODY(S); BODY(S); BODY(S); BODY(S); BODY(S); BODY(S); } \
    unsigned s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; } \
static const int n_##name = N;
#define S_E32(n) "v_cndmask_b32_e32 %" #n ", %" #n ", %8, vcc\n"
#define S_E64(n) "v_cndmask_b32_e64 %" #n ", %" #n ", %8, %10\n"
#define S_CMPSEL(n) "v_cmp_gt_u32_e32 vcc, %9, %" #n "\n v_cndmask_b32_e32 %" #n ", %" #n ", %8, vcc\n"
#define S_CMPSEL64(n) "v_cmp_gt_u32_e64 s[20:21], %9, %" #n "\n v_cndmask_b32_e64 %" #n ", %" #n ", %8, s[20:21]\n"
#define S_MINMAX(n) "v_min_u32_e32 %" #n ", %" #n ", %8\n v_max_u32_e32 %" #n ", %" #n ", %9\n"
#define S_ADD(n) "v_add_u32_e32 %" #n ", %" #n ", %8\n"
KERNEL(k_e32, S_E32, 1) KERNEL(k_e64, S_E64, 1) KERNEL(k_cmpsel, S_CMPSEL, 2) KERNEL(k_minmax, S_MINMAX, 2) KERNEL(k_add, S_ADD, 1)
__global__ __launch_bounds__(256) void k_cmpsel64(unsigned *out, unsigned a, unsigned b, unsigned long long mask) {
    unsigned r[8]; for (int k = 0; k < 8; k++) r[k] = threadIdx.x * (k + 1) + a;
    for (int it = 0; it < ITER; it++) {
#define B64 asm volatile(REP8(S_CMPSEL64) : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b), "s"(mask) : "s20", "s21")
        B64; B64; B64; B64; B64; B64; B64; B64;
    }
    unsigned s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
typedef void (*kfn)(unsigned *, unsigned, unsigned, unsigned long long);
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int blocks = prop.multiProcessorCount * 8;
    unsigned *out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    struct { const char *name; kfn f; int n; } ks[] = {{"v_add_u32 (reference)", k_add, 1}, {"v_cndmask_b32_e32 (vcc)", k_e32, 1},
        {"v_cndmask_b32_e64 (sgpr pair)", k_e64, 1}, {"v_cmp_e32 + v_cndmask_e32", k_cmpsel, 2}, {"v_cmp_e64 + v_cndmask_e64", k_cmpsel64, 2},
        {"v_min_u32 + v_max_u32", k_minmax, 2}};
    for (auto &k : ks) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, 0x5555aaaa5555aaaaull);
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u, 0x5555aaaa5555aaaaull);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double per = ms * 1e6 / ((double)ITER * 64 * 8);   // 8 waves per SIMD, 64 sequences per iteration
        printf("%-34s %.3f ns per sequence per SIMD = %.2f cycles @2.4GHz (%d instructions)\n", k.name, per, per * 2.4, k.n);
    }
    return 0;
}
