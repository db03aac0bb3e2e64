// Micro-benchmark (round 2): WHEN do the "fast" wave64 VALU instructions (v_add/sub/and/xor/mov/ashr/lshr, ~2.3 cycles alone) keep
// their rate on gfx950?  Varies (a) the number of independent register chains inside one wavefront (1 = every instruction depends
// on the previous one), (b) the run length of fast instructions between slow ones (v_dot4_u32_u8, ~4.2 cycles), with W wavefronts
// resident per SIMD.  Output: SIMD cycles per wave-instruction at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/issue_patterns.hip -o /tmp/ip && /tmp/ip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 500;

#define F(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define G(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n"
#define S(n) "v_dot4_u32_u8 %" #n ", %8, %9, %" #n "\n"
#define ASM(body) asm volatile(body : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b))
#define KERNEL(name, body, count) \
__global__ __launch_bounds__(256) void name(unsigned *out, unsigned a, unsigned b) { \
    unsigned r[8]; for (int k = 0; k < 8; k++) r[k] = threadIdx.x * (k + 1) + a; \
    for (int it = 0; it < ITER; it++) { ASM(body); ASM(body); ASM(body); ASM(body); } \
    unsigned s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; } \
constexpr int name##_n = 4 * (count);

// (a) 16 fast instructions over 1, 2, 4, 8 independent chains
KERNEL(k_dep1, F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0) F(0), 16)
KERNEL(k_dep2, F(0) F(1) F(0) F(1) F(0) F(1) F(0) F(1) F(0) F(1) F(0) F(1) F(0) F(1) F(0) F(1), 16)
KERNEL(k_dep4, F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3) F(0) F(1) F(2) F(3), 16)
KERNEL(k_dep8, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7), 16)
KERNEL(k_sdep1, S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0) S(0), 16)
KERNEL(k_sdep8, S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7), 16)
// mixed fast kinds, butterfly-like dependence (pairs): add, shift of the result, add ...
KERNEL(k_bfly, F(0) G(0) F(1) G(1) F(2) G(2) F(3) G(3) F(4) G(4) F(5) G(5) F(6) G(6) F(7) G(7), 16)
// (b) run length of fast instructions between slow ones (independent chains)
KERNEL(k_f1s1, F(0) S(1) F(2) S(3) F(4) S(5) F(6) S(7) F(0) S(1) F(2) S(3) F(4) S(5) F(6) S(7), 16)
KERNEL(k_f2s2, F(0) F(1) S(2) S(3) F(4) F(5) S(6) S(7) F(0) F(1) S(2) S(3) F(4) F(5) S(6) S(7), 16)
KERNEL(k_f4s4, F(0) F(1) F(2) F(3) S(4) S(5) S(6) S(7) F(0) F(1) F(2) F(3) S(4) S(5) S(6) S(7), 16)
KERNEL(k_f8s8, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7), 16)
KERNEL(k_f3s1, F(0) F(1) F(2) S(3) F(4) F(5) F(6) S(7) F(0) F(1) F(2) S(3) F(4) F(5) F(6) S(7), 16)
KERNEL(k_f7s1, F(0) F(1) F(2) F(3) F(4) F(5) F(6) S(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) S(7), 16)
KERNEL(k_f15s1, F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7) F(0) F(1) F(2) F(3) F(4) F(5) F(6) S(7), 16)

typedef void (*kfn)(unsigned *, unsigned, unsigned);
struct K { const char *name; kfn f; int n; double expect; };
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("%d CUs; cycles per wave64 instruction per SIMD at 2.4 GHz; 'sum' = what 2.4 (fast) / 4.2 (slow) cycles per instruction would give\n", cus);
    unsigned *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    K ks[] = {
        {"16 fast, 1 chain (all dependent)", k_dep1, k_dep1_n, 2.4}, {"16 fast, 2 chains", k_dep2, k_dep2_n, 2.4}, {"16 fast, 4 chains", k_dep4, k_dep4_n, 2.4},
        {"16 fast, 8 chains", k_dep8, k_dep8_n, 2.4}, {"16 slow, 1 chain", k_sdep1, k_sdep1_n, 4.2}, {"16 slow, 8 chains", k_sdep8, k_sdep8_n, 4.2},
        {"add -> ashr pairs, 8 chains", k_bfly, k_bfly_n, 2.4},
        {"1 fast : 1 slow alternating", k_f1s1, k_f1s1_n, 3.3}, {"2 fast, 2 slow", k_f2s2, k_f2s2_n, 3.3}, {"4 fast, 4 slow", k_f4s4, k_f4s4_n, 3.3},
        {"8 fast, 8 slow", k_f8s8, k_f8s8_n, 3.3}, {"3 fast, 1 slow", k_f3s1, k_f3s1_n, 2.85}, {"7 fast, 1 slow", k_f7s1, k_f7s1_n, 2.625},
        {"15 fast, 1 slow", k_f15s1, k_f15s1_n, 2.5125}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-36s %7s %7s %7s %7s %7s %7s\n", "pattern", "W=8", "W=5", "W=3", "W=2", "W=1", "sum");
    for (auto &k : ks) {
        printf("%-36s", k.name);
        for (int w : {8, 5, 3, 2, 1}) {
            int blocks = cus * w;
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %7.2f", ms * 1e6 / ((double)ITER * k.n * w) * 2.4);
        }
        printf(" %7.2f\n", k.expect);
    }
    return 0;
}
