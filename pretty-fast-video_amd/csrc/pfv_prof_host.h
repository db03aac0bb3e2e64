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
#ifdef PFV_KPROF         // tools/kprof.py: the rows of the last k_enc_pframe launch
extern "C" __attribute__((visibility("default"))) int pfv_debug_kprof(unsigned long long *out, int n_rows)
{
    hipDeviceSynchronize();
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(pfv::pfv_kprof), sizeof(unsigned long long) * 16 * (size_t)n_rows) == hipSuccess ? 0 : -1;
}
#endif
