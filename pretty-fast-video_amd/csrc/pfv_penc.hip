// pfv_penc.hip -- launcher of the p-frame encode kernels (k_enc_pframe<FLT>, k_enc_pframe16<FLT>; pfv_kernels.hip).
//
// In the product build this file is its OWN translation unit, compiled with `-mllvm -amdgpu-sched-strategy=max-ilp`
// (__graft_entry__.build_hip): k_enc_pframe is the one kernel of the path that answers to instruction SCHEDULING -- the ILP-first
// strategy hoists its LDS reads further ahead of their uses and runs it 2.6 % faster (profiles/r04_sched_strategies.txt: 574 -> 559 us
// per 96 x 1080p launch, no spills at 95 VGPRs) -- while the same strategy costs k_enc_iframe 9 % and the decoders 1 %, and LLVM offers
// the choice per compilation, not per function.  Everything else stays in pfv_capi.hip's unit, which then only DECLARES the launcher
// (-DPFV_SPLIT_PENC).  Without that macro pfv_capi.hip includes this file and the library is one translation unit as before (the CPU
// emulator build of the tests, the single-command variant builds of tools/).
#ifndef PFV_CAPI_TU
#define PFV_PENC_TU
#include "pfv_kernels.hip"
#endif

namespace pfv {

void launch_enc_pframe_kernels(hipStream_t stream, bool flt, bool small, int compact_max, const FrameGeom &g, unsigned blocks, const uint8_t *src,
                               const uint8_t *ref, int8_t *mv, uint8_t *has, int16_t *coef, uint8_t *recon, const QTab *qt, float min_err)
{
    if (small) {
        if (flt) hipLaunchKernelGGL(k_enc_pframe16<true>, dim3(blocks), dim3(kThreads16), 0, stream, g, src, ref, mv, has, coef, recon, qt, min_err, -2, kQuantMagic);
        else hipLaunchKernelGGL(k_enc_pframe16<false>, dim3(blocks), dim3(kThreads16), 0, stream, g, src, ref, mv, has, coef, recon, qt, min_err, -2, kQuantMagic);
    } else if (compact_max == kPencSplit) {   // search kernel + transform kernel (PFV_OPT_TILE_COMPACTION = 2)
        // the search kernel only: the transform kernel is launched by the caller, from the main translation unit (default scheduling strategy)
        hipLaunchKernelGGL(k_pf_search<true>, dim3(blocks), dim3(kThreads), 0, stream, g, src, ref, mv, has, coef, recon, min_err, -2);
    } else {
        if (flt) hipLaunchKernelGGL(k_enc_pframe<true>, dim3(blocks), dim3(kThreads), 0, stream, g, src, ref, mv, has, coef, recon, qt, min_err, -2, kQuantMagic, compact_max);
        else hipLaunchKernelGGL(k_enc_pframe<false>, dim3(blocks), dim3(kThreads), 0, stream, g, src, ref, mv, has, coef, recon, qt, min_err, -2, kQuantMagic, compact_max);
    }
}

}  // namespace pfv

#ifdef PFV_KPROF         // tools/kprof.py, tools/kprof_simd.py: the stamp rows of the last k_enc_pframe launch (the array lives where the kernel does)
extern "C" __attribute__((visibility("default"))) int pfv_debug_kprof(unsigned long long *out, int n_rows)
{
    (void)hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pfv::pfv_kprof), sizeof(unsigned long long) * 16 * (size_t)n_rows) == hipSuccess ? 0 : -1;
}
#endif
