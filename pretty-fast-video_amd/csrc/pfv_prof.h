// pfv_prof.h -- in-kernel phase timestamps for the experiment builds (-DPFV_KPROF: k_enc_pframe, tools/kprof.py;
// -DPFV_ENT_PROFILE: k_ent_scan / k_ent_pack, tools/ent_profile.py).  The kernels carry only the marks below; in the shipped build
// every one of them is empty.  A mark stores clock64() of the workgroup's first thread in a row of 16 per workgroup.
#pragma once
#include <hip/hip_runtime.h>

namespace pfv {

#if defined(PFV_KPROF) && defined(PFV_SPLIT_PENC) && !defined(PFV_PENC_TU)
// split build: the marks (and the array they write) belong to the p-frame encoder's own translation unit
#define KMARK(i) do {} while (0)
#define KMARK_WHERE() do {} while (0)
#elif defined(PFV_KPROF) && PFV_KPROF >= 2
// -DPFV_KPROF=2: one row per WAVEFRONT (row = workgroup * 4 + wavefront), written by the wavefront's first lane -- what each SIMD's
// resident wavefronts were doing at any time can then be reconstructed (tools/kprof_simd.py: when could NO resident wavefront issue?)
constexpr int kProfRows = 1 << 18;
__device__ unsigned long long pfv_kprof[kProfRows][16];
#define KPROF_ROW() ((int)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)))
#define KMARK(i) do { if ((threadIdx.x & 63) == 0 && KPROF_ROW() < kProfRows) pfv_kprof[KPROF_ROW()][i] = clock64(); } while (0)
#define KMARK_WHERE() do { if ((threadIdx.x & 63) == 0 && KPROF_ROW() < kProfRows) { \
        pfv_kprof[KPROF_ROW()][12] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
        pfv_kprof[KPROF_ROW()][13] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); } } while (0)
#elif defined(PFV_KPROF)
constexpr int kProfRows = 1 << 16;
__device__ unsigned long long pfv_kprof[kProfRows][16];
#define KMARK(i) do { if (threadIdx.x == 0 && blockIdx.x < kProfRows) pfv_kprof[blockIdx.x][i] = clock64(); } while (0)
// where the workgroup runs: HW_ID (cu / sh / se ids) and XCC_ID
#define KMARK_WHERE() do { if (threadIdx.x == 0 && blockIdx.x < kProfRows) { \
        pfv_kprof[blockIdx.x][12] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); \
        pfv_kprof[blockIdx.x][13] = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (31 << 11)); } } while (0)
#else
#define KMARK(i) do {} while (0)
#define KMARK_WHERE() do {} while (0)
#endif

#ifdef PFV_ENT_PROFILE
constexpr int kEntProfGroups = 1 << 15;
__device__ unsigned long long ent_prof[2][kEntProfGroups][16];
#define ENT_MARK(kern, i) do { if (threadIdx.x == 0 && prof_row_ < kEntProfGroups) ent_prof[kern][prof_row_][i] = clock64(); } while (0)
#define ENT_MARK0() const unsigned prof_row_ = blockIdx.y * gridDim.x + blockIdx.x
#else
#define ENT_MARK(kern, i) do {} while (0)
#define ENT_MARK0() do {} while (0)
#endif

}  // namespace pfv

