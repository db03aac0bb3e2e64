// pfv_prof_host.h -- host side of the experiment builds (see pfv_prof.h): the timestamp rows of the last launch.
// Included at the end of pfv_capi.hip; empty in the shipped build.
#pragma once
#ifdef PFV_ENT_PROFILE   // tools/ent_profile.py: the rows of kernel `kern` (0 scan, 1 pack)
extern "C" __attribute__((visibility("default"))) int pfv_debug_ent_profile(int kern, unsigned long long *out, int n_groups)
{
    hipDeviceSynchronize();
    const size_t row = sizeof(unsigned long long) * 16;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pfv::ent_prof), row * (size_t)n_groups, row * pfv::kEntProfGroups * (size_t)kern) != hipSuccess) return -1;
    return 0;
}
#endif
// (pfv_debug_kprof, the k_enc_pframe stamp rows, is in pfv_penc.hip: the array lives in the translation unit that holds the kernel)
