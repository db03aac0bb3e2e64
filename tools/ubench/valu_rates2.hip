// Micro-benchmark (round 2): issue cost of the wave64 VALU instructions the PFV kernels use or could use (gfx950), by
// encoding class (VOP1/VOP2 vs VOP3 vs VOP3P vs DPP/SDWA).  Each kernel runs ITER x 64 copies of one instruction on 8
// independent register chains per wavefront; the grid keeps W wavefronts resident per SIMD (W = 8, 4, 1).  Printed:
// SIMD cycles per wave-instruction at the 2.4 GHz peak clock (the sustained clock is lower, so true figures are a bit smaller).
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O2 tools/ubench/valu_rates2.hip -o /tmp/vr2 && /tmp/vr2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
constexpr int ITER = 1000;
#define OP8(str) asm volatile( \
    str(0) str(1) str(2) str(3) str(4) str(5) str(6) str(7) \
    : "+v"(r[0]), "+v"(r[1]), "+v"(r[2]), "+v"(r[3]), "+v"(r[4]), "+v"(r[5]), "+v"(r[6]), "+v"(r[7]) : "v"(a), "v"(b))
#define KERNEL(name, S) \
__global__ __launch_bounds__(256) void name(unsigned *out, unsigned a, unsigned b) { \
    unsigned r[8]; for (int k = 0; k < 8; k++) r[k] = threadIdx.x * (k + 1) + a; \
    for (int it = 0; it < ITER; it++) { OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); OP8(S); } \
    unsigned s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }
// two different instructions alternating (same count of each): is the cost additive?
#define KERNEL2(name, S, T) \
__global__ __launch_bounds__(256) void name(unsigned *out, unsigned a, unsigned b) { \
    unsigned r[8]; for (int k = 0; k < 8; k++) r[k] = threadIdx.x * (k + 1) + a; \
    for (int it = 0; it < ITER; it++) { OP8(S); OP8(T); OP8(S); OP8(T); OP8(S); OP8(T); OP8(S); OP8(T); } \
    unsigned s = 0; for (int k = 0; k < 8; k++) s += r[k]; out[blockIdx.x * 256 + threadIdx.x] = s; }

#define I2(op) "v_" op " %"
#define S_ADD(n) "v_add_u32 %" #n ", %" #n ", %8\n"
#define S_SUB(n) "v_sub_u32 %" #n ", %" #n ", %8\n"
#define S_AND(n) "v_and_b32 %" #n ", %" #n ", %8\n"
#define S_XOR(n) "v_xor_b32 %" #n ", %" #n ", %8\n"
#define S_MOV(n) "v_mov_b32 %" #n ", %8\n"
#define S_ASHR(n) "v_ashrrev_i32 %" #n ", 1, %" #n "\n"
#define S_LSHR(n) "v_lshrrev_b32 %" #n ", 1, %" #n "\n"
#define S_LSHL(n) "v_lshlrev_b32 %" #n ", 1, %" #n "\n"
#define S_MIN(n) "v_min_i32 %" #n ", %" #n ", %8\n"
#define S_MAX(n) "v_max_i32 %" #n ", %" #n ", %8\n"
#define S_MINU(n) "v_min_u32 %" #n ", %" #n ", %8\n"
#define S_MUL24(n) "v_mul_i32_i24 %" #n ", %" #n ", %8\n"
#define S_MULU24(n) "v_mul_u32_u24 %" #n ", %" #n ", %8\n"
#define S_MULLO(n) "v_mul_lo_u32 %" #n ", %" #n ", %8\n"
#define S_MAD24(n) "v_mad_i32_i24 %" #n ", %" #n ", %8, %9\n"
#define S_DOT4(n) "v_dot4_u32_u8 %" #n ", %8, %9, %" #n "\n"
#define S_DOT4I(n) "v_dot4_i32_i8 %" #n ", %8, %9, %" #n "\n"
#define S_DOT4C(n) "v_dot4c_i32_i8 %" #n ", %8, %9\n"
#define S_DOT2C(n) "v_dot2c_i32_i16 %" #n ", %8, %9\n"
#define S_ALIGNBYTE(n) "v_alignbyte_b32 %" #n ", %" #n ", %8, %9\n"
#define S_ALIGNBIT(n) "v_alignbit_b32 %" #n ", %" #n ", %8, 8\n"
#define S_MED3(n) "v_med3_i32 %" #n ", %" #n ", %8, %9\n"
#define S_LSHLOR(n) "v_lshl_or_b32 %" #n ", %" #n ", 8, %8\n"
#define S_ANDOR(n) "v_and_or_b32 %" #n ", %" #n ", %8, %9\n"
#define S_ADD3(n) "v_add3_u32 %" #n ", %" #n ", %8, %9\n"
#define S_LSHLADD(n) "v_lshl_add_u32 %" #n ", %" #n ", 1, %8\n"
#define S_ADDLSHL(n) "v_add_lshl_u32 %" #n ", %" #n ", %8, 1\n"
#define S_BFEU(n) "v_bfe_u32 %" #n ", %" #n ", 8, 8\n"
#define S_BFEI(n) "v_bfe_i32 %" #n ", %" #n ", 8, 8\n"
#define S_PERM(n) "v_perm_b32 %" #n ", %" #n ", %8, %9\n"
#define S_CVTF(n) "v_cvt_f32_i32 %" #n ", %" #n "\n"
#define S_CVTI(n) "v_cvt_i32_f32 %" #n ", %" #n "\n"
#define S_MULF(n) "v_mul_f32 %" #n ", %" #n ", %8\n"
#define S_FMA(n) "v_fma_f32 %" #n ", %" #n ", %8, %9\n"
#define S_CNDMASK(n) "v_cndmask_b32 %" #n ", %" #n ", %8, vcc\n"
#define S_DPPADD(n) "v_add_u32_dpp %" #n ", %" #n ", %" #n " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define S_DPPADDM(n) "v_add_u32_dpp %" #n ", %" #n ", %" #n " row_half_mirror row_mask:0xf bank_mask:0xf\n"
#define S_DPPMOV(n) "v_mov_b32_dpp %" #n ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define S_DPPMIN(n) "v_min_u32_dpp %" #n ", %" #n ", %" #n " quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n"
#define S_SDWASUB(n) "v_sub_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_2\n"
#define S_SDWAADDW(n) "v_add_u32_sdwa %" #n ", %" #n ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_0\n"
#define S_PKADD(n) "v_pk_add_i16 %" #n ", %" #n ", %8\n"
#define S_PKSUB(n) "v_pk_sub_i16 %" #n ", %" #n ", %8\n"
#define S_PKASHR(n) "v_pk_ashrrev_i16 %" #n ", 1, %" #n " op_sel_hi:[0,1]\n"
#define S_PKMAD(n) "v_pk_mad_i16 %" #n ", %" #n ", %8, %9\n"
#define S_PKMIN(n) "v_pk_min_i16 %" #n ", %" #n ", %8\n"
#define S_PKMAX(n) "v_pk_max_i16 %" #n ", %" #n ", %8\n"
#define S_PKMUL(n) "v_pk_mul_lo_u16 %" #n ", %" #n ", %8\n"
#define S_SATPK(n) "v_sat_pk_u8_i16 %" #n ", %" #n "\n"
#define S_CVTPKI16(n) "v_cvt_pk_i16_i32 %" #n ", %" #n ", %8\n"
#define S_CVTPKU8(n) "v_cvt_pk_u8_f32 %" #n ", %8, 1, %" #n "\n"
#define S_SAD(n) "v_sad_u8 %" #n ", %8, %9, %" #n "\n"
#define S_PKADDF(n) "v_pk_add_f32 %[r" #n "], %[r" #n "], %[r" #n "]\n"

#define LIST(X) \
 X(add_u32, S_ADD) X(sub_u32, S_SUB) X(and_b32, S_AND) X(xor_b32, S_XOR) X(mov_b32, S_MOV) X(ashrrev_i32, S_ASHR) X(lshrrev_b32, S_LSHR) \
 X(lshlrev_b32, S_LSHL) X(min_i32, S_MIN) X(max_i32, S_MAX) X(min_u32, S_MINU) X(mul_i32_i24, S_MUL24) X(mul_u32_u24, S_MULU24) \
 X(mul_lo_u32, S_MULLO) X(mad_i32_i24, S_MAD24) X(dot4_u32_u8, S_DOT4) X(dot4_i32_i8, S_DOT4I) X(dot4c_i32_i8, S_DOT4C) X(dot2c_i32_i16, S_DOT2C) \
 X(alignbyte_b32, S_ALIGNBYTE) X(alignbit_b32, S_ALIGNBIT) X(med3_i32, S_MED3) X(lshl_or_b32, S_LSHLOR) X(and_or_b32, S_ANDOR) \
 X(add3_u32, S_ADD3) X(lshl_add_u32, S_LSHLADD) X(add_lshl_u32, S_ADDLSHL) X(bfe_u32, S_BFEU) X(bfe_i32, S_BFEI) X(perm_b32, S_PERM) \
 X(cvt_f32_i32, S_CVTF) X(cvt_i32_f32, S_CVTI) X(mul_f32, S_MULF) X(fma_f32, S_FMA) X(cndmask_b32_vcc, S_CNDMASK) \
 X(add_u32_dpp_quad, S_DPPADD) X(add_u32_dpp_half_mirror, S_DPPADDM) X(mov_b32_dpp_quad, S_DPPMOV) X(min_u32_dpp_quad, S_DPPMIN) \
 X(sub_u32_sdwa_bytes, S_SDWASUB) X(add_u32_sdwa_words, S_SDWAADDW) \
 X(pk_add_i16, S_PKADD) X(pk_sub_i16, S_PKSUB) X(pk_ashrrev_i16, S_PKASHR) X(pk_mad_i16, S_PKMAD) X(pk_min_i16, S_PKMIN) X(pk_max_i16, S_PKMAX) \
 X(pk_mul_lo_u16, S_PKMUL) X(sat_pk_u8_i16, S_SATPK) X(cvt_pk_i16_i32, S_CVTPKI16) X(cvt_pk_u8_f32, S_CVTPKU8) X(sad_u8, S_SAD)

#define DEF(name, S) KERNEL(k_##name, S)
LIST(DEF)
KERNEL2(k2_add_dot4, S_ADD, S_DOT4) KERNEL2(k2_add_dppadd, S_ADD, S_DPPADD) KERNEL2(k2_add_pkadd, S_ADD, S_PKADD) KERNEL2(k2_add_ashr, S_ADD, S_ASHR)
KERNEL2(k2_dot4_alignbyte, S_DOT4, S_ALIGNBYTE) KERNEL2(k2_add_mul24, S_ADD, S_MUL24) KERNEL2(k2_add_cvtf, S_ADD, S_CVTF)

typedef void (*kfn)(unsigned *, unsigned, unsigned);
struct K { const char *name; kfn f; };
#define ENT(name, S) {"v_" #name, k_##name},
int main() {
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    int cus = prop.multiProcessorCount; double ghz = prop.clockRate * 1e-6;
    printf("device %s, %d CUs, clockRate %.2f GHz; cycles per wave64 instruction per SIMD at 2.4 GHz, W wavefronts resident per SIMD\n", prop.name, cus, ghz);
    unsigned *out; CHECK(hipMalloc(&out, (size_t)cus * 8 * 256 * 4));
    K ks[] = { LIST(ENT)
        {"mix v_add_u32 + v_dot4_u32_u8", k2_add_dot4}, {"mix v_add_u32 + v_add_u32_dpp", k2_add_dppadd}, {"mix v_add_u32 + v_pk_add_i16", k2_add_pkadd},
        {"mix v_add_u32 + v_ashrrev_i32", k2_add_ashr}, {"mix v_dot4_u32_u8 + v_alignbyte_b32", k2_dot4_alignbyte}, {"mix v_add_u32 + v_mul_i32_i24", k2_add_mul24},
        {"mix v_add_u32 + v_cvt_f32_i32", k2_add_cvtf} };
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    printf("%-36s %8s %8s %8s\n", "instruction", "W=8", "W=4", "W=1");
    for (auto &k : ks) {
        printf("%-36s", k.name);
        for (int w : {8, 4, 1}) {
            int blocks = cus * w;   // w blocks of 4 wavefronts per CU = w wavefronts per SIMD
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k.f, dim3(blocks), dim3(256), 0, 0, out, 3u, 5u);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            double instr_per_simd = (double)ITER * 64 * w;
            printf(" %8.2f", ms * 1e6 / instr_per_simd * 2.4);
        }
        printf("\n");
    }
    return 0;
}
