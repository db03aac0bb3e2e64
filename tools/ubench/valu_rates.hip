// Micro-benchmark: issue rate of the integer VALU ops the PFV kernels are made of (gfx950).
// Each kernel runs ITER x 64 copies of one instruction on 8 independent register chains per wave,
// 8 waves per SIMD resident; reports ns per wave-instruction per SIMD (x clock = cycles).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 2000;
#define OP8(str) asm volatile( \
    str(0) str(1) str(2) str(3) str(4) str(5) str(6) str(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b))
#define KERNEL(name, S) \
__global__ __launch_bounds__(256) void name(unsigned *out, unsigned a, unsigned b) { \
    unsigned r[8]; for (int k = 0; k < 8; k++) r[k] = threadIdx.x * (k + 1) + a; \
    for (int it = 0; it < ITER; it++) { OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); } \
    unsigned s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }

#define S_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define S_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define S_MUL24(n) "v_mul_i32_i24 %" #n ", %" #n ", %8\n"
#define S_MAD24(n) "v_mad_u32_u24 %" #n ", %" #n ", %8, %9\n"
#define S_MULHI24(n) "v_mul_hi_i32_i24 %" #n ", %" #n ", %8\n"
#define S_DOT4(n) "v_dot4_u32_u8 %" #n ", %8, %9, %" #n "\n"
#define S_ALIGN(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, %9\n"
#define S_ASHR(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n"
#define S_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define S_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 8, %8\n"
#define S_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define S_CVTF(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define S_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define S_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define S_DPP(n) "v_add_u32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define S_BFE(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define S_SDWA(n) "v_sub_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define S_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define S_PKADD(n) "v_pk_add_u16 %" #n ", %" #n ", %8\n"
#define S_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define S_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define S_MADU64(n) "v_mul_hi_u32 %" #n ", %" #n ", %8\n"

KERNEL(k_add, S_ADD) KERNEL(k_mullo, S_MULLO) KERNEL(k_mul24, S_MUL24) KERNEL(k_mad24, S_MAD24) KERNEL(k_mulhi24, S_MULHI24)
KERNEL(k_dot4, S_DOT4) KERNEL(k_align, S_ALIGN) KERNEL(k_ashr, S_ASHR) KERNEL(k_med3, S_MED3) KERNEL(k_lshlor, S_LSHLOR)
KERNEL(k_add3, S_ADD3) KERNEL(k_cvtf, S_CVTF) KERNEL(k_cvti, S_CVTI) KERNEL(k_mulf, S_MULF) KERNEL(k_dpp, S_DPP)
KERNEL(k_bfe, S_BFE) KERNEL(k_sdwa, S_SDWA) KERNEL(k_perm, S_PERM) KERNEL(k_pkadd, S_PKADD) KERNEL(k_lshladd, S_LSHLADD)
KERNEL(k_cndmask, S_CNDMASK) KERNEL(k_mulhi, S_MADU64)

typedef void (*kfn)(unsigned *, unsigned, unsigned);
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount; double ghz = prop.clockRate * 1e-6;
    printf("device %s, %d CUs, clockRate %.2f GHz\n", prop.name, cus, ghz);
    int blocks = cus * 8;   // 8 blocks of 4 waves per CU = 8 waves per SIMD
    unsigned *out; CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    struct { const char *name; kfn f; } ks[] = {
        {"v_add_u32", k_add}, {"v_mul_lo_u32", k_mullo}, {"v_mul_i32_i24", k_mul24}, {"v_mad_u32_u24", k_mad24},
        {"v_mul_hi_i32_i24", k_mulhi24}, {"v_dot4_u32_u8", k_dot4}, {"v_alignbyte_b32", k_align}, {"v_ashrrev_i32", k_ashr},
        {"v_med3_i32", k_med3}, {"v_lshl_or_b32", k_lshlor}, {"v_add3_u32", k_add3}, {"v_cvt_f32_i32", k_cvtf},
        {"v_cvt_i32_f32", k_cvti}, {"v_mul_f32", k_mulf}, {"v_add_u32_dpp", k_dpp}, {"v_bfe_u32", k_bfe},
        {"v_sub_u32_sdwa", k_sdwa}, {"v_perm_b32", k_perm}, {"v_pk_add_u16", k_pkadd}, {"v_lshl_add_u32", k_lshladd},
        {"v_cndmask_b32", k_cndmask}, {"v_mul_hi_u32", k_mulhi}};
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (auto &k : ks) {
        hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        double instr_per_simd = (double)ITER * 64 * 8;   // per wave x 8 waves per SIMD
        double ns = ms * 1e6 / instr_per_simd;
        printf("%-18s %8.3f ms  %6.3f ns/wave-instr/SIMD  = %5.2f cycles @2.4GHz\n", k.name, ms, ns, ns * 2.4);
    }
    return 0;
}
