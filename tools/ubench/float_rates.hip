// Micro-benchmark (round 2): issue cost of the f32 instructions a float formulation of the integer DCT would use on gfx950
// (packed VOP3P f32 ops work on two values per instruction).  Same method as valu_rates2.hip: 8 independent chains, W wavefronts
// resident per SIMD; cycles per wave64 instruction per SIMD at 2.4 GHz.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench/float_rates.hip -o /tmp/fr && /tmp/fr
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 1000;
typedef float float2_t __attribute__((ext_vector_type(2)));
#define OP8(str) asm volatile( \
    str(0) str(1) str(2) str(3) str(4) str(5) str(6) str(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b))
#define KERNEL(name, S) \
__global__ __launch_bounds__(256) void name(float *out, float fa, float fb) { \
    float r[8]; float a = fa, b = fb; for (int k = 0; k < 8; k++) r[k] = (float)(threadIdx.x * (k + 1)) + fa; \
    for (int it = 0; it < ITER; it++) { OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); } \
    float s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
// packed: 8 chains of 64-bit register pairs
#define OP8P(str) asm volatile( \
    str(0) str(1) str(2) str(3) str(4) str(5) str(6) str(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b))
#define KERNELP(name, S) \
__global__ __launch_bounds__(256) void name(float *out, float fa, float fb) { \
    float2_t r[8]; float2_t a = {fa, fa}, b = {fb, fb}; for (int k = 0; k < 8; k++) { r[k].x = (float)(threadIdx.x * (k + 1)) + fa; r[k].y = r[k].x + 1.0f; } \
    for (int it = 0; it < ITER; it++) { OP8P(S); OP8P(S); OP8P(S); OP8P(S); OP8P(S); OP8P(S); OP8P(S); OP8P(S); } \
    float s = 0; for (int k = 0; k < 8; k++) s += r[k].x + r[k].y; out[blockIdx.x * 256 + threadIdx.x] = s; }

#define S_ADDF(n) "v_add_f32 %" #n ", %" #n ", %8\n"
#define S_SUBF(n) "v_sub_f32 %" #n ", %" #n ", %8\n"
#define S_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define S_FMAF(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define S_MINF(n) "v_min_f32 %" #n ", %" #n ", %8\n"
#define S_MAXF(n) "v_max_f32 %" #n ", %" #n ", %8\n"
#define S_TRUNC(n) "v_trunc_f32 %" #n ", %" #n "\n"
#define S_FLOOR(n) "v_floor_f32 %" #n ", %" #n "\n"
#define S_CVTUB(n) "v_cvt_f32_ubyte1 %" #n ", %" #n "\n"
#define S_CVTPKU8(n) "v_cvt_pk_u8_f32 %" #n ", %8, 1, %" #n "\n"
#define S_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define S_CVTF(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define S_MED3F(n) "v_med3_f32 %" #n ", %" #n ", %8, %9\n"
#define S_PKADD(n) "v_pk_add_f32 %" #n ", %" #n ", %8\n"
#define S_PKMUL(n) "v_pk_mul_f32 %" #n ", %" #n ", %8\n"
#define S_PKFMA(n) "v_pk_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define S_PKADDNEG(n) "v_pk_add_f32 %" #n ", %" #n ", %8 neg_lo:[0,1] neg_hi:[0,1]\n"

KERNEL(k_addf, S_ADDF) KERNEL(k_subf, S_SUBF) KERNEL(k_mulf, S_MULF) KERNEL(k_fmaf, S_FMAF) KERNEL(k_minf, S_MINF) KERNEL(k_maxf, S_MAXF)
KERNEL(k_trunc, S_TRUNC) KERNEL(k_floor, S_FLOOR) KERNEL(k_cvtub, S_CVTUB) KERNEL(k_cvtpku8, S_CVTPKU8) KERNEL(k_cvti, S_CVTI) KERNEL(k_cvtf, S_CVTF)
KERNEL(k_med3f, S_MED3F)
KERNELP(k_pkadd, S_PKADD) KERNELP(k_pkmul, S_PKMUL) KERNELP(k_pkfma, S_PKFMA) KERNELP(k_pkaddneg, S_PKADDNEG)

typedef void (*kfn)(float *, float, float);
struct K { const char *name; kfn f; };
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount;
    printf("%d CUs; cycles per wave64 instruction per SIMD at 2.4 GHz (packed instructions process two values each)\n", cus);
    float *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    K ks[] = {{"v_add_f32", k_addf}, {"v_sub_f32", k_subf}, {"v_mul_f32", k_mulf}, {"v_fma_f32", k_fmaf}, {"v_min_f32", k_minf}, {"v_max_f32", k_maxf},
              {"v_med3_f32", k_med3f}, {"v_trunc_f32", k_trunc}, {"v_floor_f32", k_floor}, {"v_cvt_f32_ubyte1", k_cvtub}, {"v_cvt_pk_u8_f32", k_cvtpku8},
              {"v_cvt_i32_f32", k_cvti}, {"v_cvt_f32_i32", k_cvtf}, {"v_pk_add_f32", k_pkadd}, {"v_pk_add_f32 (neg)", k_pkaddneg},
              {"v_pk_mul_f32", k_pkmul}, {"v_pk_fma_f32", k_pkfma}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-24s %8s %8s %8s\n", "instruction", "W=8", "W=4", "W=1");
    for (auto &k : ks) {
        printf("%-24s", k.name);
        for (int w : {8, 4, 1}) {
            int blocks = cus * w;
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3.0f, 0.5f);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3.0f, 0.5f);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf(" %8.2f", ms * 1e6 / ((double)ITER * 64 * w) * 2.4);
        }
        printf("\n");
    }
    return 0;
}
