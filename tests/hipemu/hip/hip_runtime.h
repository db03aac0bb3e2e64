// tests/hipemu/hip/hip_runtime.h -- TEST INFRASTRUCTURE.
//
// A tiny single-threaded emulator of the subset of the HIP runtime + gfx950 builtins that
// pretty-fast-video_amd/csrc uses, so that the *unmodified* kernel and C-ABI sources can be
// compiled with g++ and exercised on a machine without a GPU (the build container).  Every
// GPU thread of a workgroup is a ucontext fiber; __syncthreads and every wavefront-level
// exchange (DPP, shuffles, votes, the LDS hand-off barrier) are rendezvous points of the
// fiber scheduler.  Wavefront = 64 lanes.  This checks indexing, LDS layout, cross-lane
// logic and integer arithmetic; it says nothing about performance or hardware memory
// ordering -- the `-m gpu` tests on a real MI355X remain the parity gate.
//
// Never linked into libpfv_hip.so.
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <ucontext.h>

#include <cstdio>
#include <functional>
#include <vector>

#define PFV_HIPEMU 1
#define address_space(n)
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __constant__ static
#define __launch_bounds__(...)

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct int4 { int x, y, z, w; };
struct float4 { float x, y, z, w; };
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
static inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
static inline unsigned long long min(unsigned long long a, unsigned long long b) { return a < b ? a : b; }
static inline unsigned long long max(unsigned long long a, unsigned long long b) { return a > b ? a : b; }

namespace hipemu {

struct Idx { unsigned x, y, z; };
extern Idx g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

void launch(dim3 grid, dim3 block, const std::function<void()> &body);
void block_barrier();
void wave_barrier();
int wave_exchange(int value, int src_lane);   // returns `value` of lane src_lane of the caller's wavefront
bool wave_any(bool pred);

}  // namespace hipemu

#define threadIdx hipemu::g_threadIdx
#define blockIdx hipemu::g_blockIdx
#define blockDim hipemu::g_blockDim
#define gridDim hipemu::g_gridDim

static inline void __syncthreads() { hipemu::block_barrier(); }
static inline int __shfl_xor(int v, int mask) { return hipemu::wave_exchange(v, (int)((threadIdx.x & 63) ^ (unsigned)mask)); }
static inline bool __any(bool p) { return hipemu::wave_any(p); }
static inline unsigned long long __ballot(bool p)
{
    unsigned long long m = 0;
    for (int l = 0; l < 64; l++) m |= (unsigned long long)(hipemu::wave_exchange(p ? 1 : 0, l) & 1) << l;
    return m;
}
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __mul24(int a, int b)   // v_mul_i32_i24: low 32 bits of the product of the operands' low 24 bits, sign-extended
{
    const long long x = (int)((unsigned)a << 8) >> 8, y = (int)((unsigned)b << 8) >> 8;   // (no left shift of a negative value: UBSan runs over this file too)
    return (int)(unsigned)(unsigned long long)(x * y);
}
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline int __shfl(int v, int src_lane) { return hipemu::wave_exchange(v, src_lane); }
static inline int atomicOr(int *p, int v) { int o = *p; *p = o | v; return o; }
static inline unsigned atomicOr(unsigned *p, unsigned v) { unsigned o = *p; *p = o | v; return o; }
static inline unsigned atomicMin(unsigned *p, unsigned v) { unsigned o = *p; if (v < o) *p = v; return o; }
static inline int atomicAdd(int *p, int v) { int o = *p; *p = o + v; return o; }
static inline unsigned atomicAdd(unsigned *p, unsigned v) { unsigned o = *p; *p = o + v; return o; }
static inline unsigned long long atomicAdd(unsigned long long *p, unsigned long long v) { unsigned long long o = *p; *p = o + v; return o; }

// ---- gfx950 builtins used by the kernels
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_wave_barrier() hipemu::wave_barrier()
static inline unsigned hipemu_udot4(unsigned a, unsigned b, unsigned c, bool)
{
    for (int i = 0; i < 4; i++) c += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
    return c;
}
#define __builtin_amdgcn_udot4 hipemu_udot4
template <class V>
static inline unsigned hipemu_udot2(V a, V b, unsigned c, bool)
{
    unsigned short x[2], y[2];
    static_assert(sizeof(V) == 4, "two 16-bit values");
    __builtin_memcpy(x, &a, 4);
    __builtin_memcpy(y, &b, 4);
    return c + (unsigned)x[0] * y[0] + (unsigned)x[1] * y[1];
}
#define __builtin_amdgcn_udot2 hipemu_udot2
static inline unsigned hipemu_alignbyte(unsigned hi, unsigned lo, unsigned sh)
{
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (unsigned)(v >> (8 * (sh & 3)));
}
#define __builtin_amdgcn_alignbyte hipemu_alignbyte
static inline unsigned hipemu_alignbit(unsigned hi, unsigned lo, unsigned sh)
{
    uint64_t v = ((uint64_t)hi << 32) | lo;
    return (unsigned)(v >> (sh & 31));
}
#define __builtin_amdgcn_alignbit hipemu_alignbit
static inline int hipemu_update_dpp(int old, int src, int ctrl, int row_mask, int, bool bound_ctrl)
{
    int lane = (int)(threadIdx.x & 63), from = lane;
    bool valid = true;   // an invalid source lane leaves `old` in the destination (bound_ctrl: 0 instead)
    if (ctrl >= 0 && ctrl <= 0xff) from = (lane & ~3) | ((ctrl >> (2 * (lane & 3))) & 3);   // quad_perm
    else if (ctrl >= 0x111 && ctrl <= 0x11f) {                                              // row_shr:n
        const int n = ctrl & 15;
        valid = (lane & 15) >= n;
        from = valid ? lane - n : lane;
    }
    else if (ctrl >= 0x121 && ctrl <= 0x12f) from = (lane & ~15) | (((lane & 15) - (ctrl & 15)) & 15);   // row_ror:n (rotate right inside 16 lanes)
    else if (ctrl == 0x140) from = (lane & ~15) | (15 - (lane & 15));                       // row_mirror
    else if (ctrl == 0x141) from = (lane & ~7) | (7 - (lane & 7));                          // row_half_mirror
    else if (ctrl == 0x142) { valid = lane >= 16; from = valid ? ((lane & ~15) - 1) : lane; }   // row_bcast:15
    else if (ctrl == 0x143) { valid = lane >= 32; from = valid ? 31 : lane; }                   // row_bcast:31
    else abort();
    const int got = hipemu::wave_exchange(src, from);
    if (!((row_mask >> (lane >> 4)) & 1)) return old;
    if (!valid) return bound_ctrl ? 0 : old;
    return got;
}
#define __builtin_amdgcn_update_dpp hipemu_update_dpp
// global_load_lds: per-lane copy global -> LDS (the emulator has no lane-linear restriction; the kernels keep to it)
static inline void hipemu_global_load_lds(const void *g, void *l, unsigned size, int offset, int) { memcpy((char *)l + offset, (const char *)g + offset, size); }
#define __builtin_amdgcn_global_load_lds hipemu_global_load_lds
static inline int hipemu_ds_bpermute(int byte_addr, int v) { return hipemu::wave_exchange(v, (byte_addr >> 2) & 63); }
#define __builtin_amdgcn_ds_bpermute hipemu_ds_bpermute
// v_cvt_pk_u8_f32: byte `idx` of `old` replaced by the float converted to u8 (round to nearest even, saturated to 0..255)
static inline unsigned hipemu_cvt_pk_u8_f32(float v, unsigned idx, unsigned old)
{
    float r = __builtin_rintf(v);
    unsigned b = !(r > 0.0f) ? 0u : (r >= 255.0f ? 255u : (unsigned)r);
    return (old & ~(0xffu << (8 * (idx & 3)))) | (b << (8 * (idx & 3)));
}
#define __builtin_amdgcn_cvt_pk_u8_f32 hipemu_cvt_pk_u8_f32
// v_mbcnt_lo / _hi: the number of set mask bits BELOW the lane (lo: lanes 0..31 of the mask, hi: lanes 32..63), added to `v`
static inline unsigned hipemu_mbcnt_lo(unsigned mask, unsigned v)
{
    const unsigned lane = threadIdx.x & 63;
    return v + (unsigned)__builtin_popcount(mask & (lane >= 32 ? 0xffffffffu : ((1u << lane) - 1u)));
}
static inline unsigned hipemu_mbcnt_hi(unsigned mask, unsigned v)
{
    const unsigned lane = threadIdx.x & 63;
    return v + (lane > 32 ? (unsigned)__builtin_popcount(mask & ((1u << (lane - 32)) - 1u)) : 0u);
}
#define __builtin_amdgcn_mbcnt_lo hipemu_mbcnt_lo
#define __builtin_amdgcn_mbcnt_hi hipemu_mbcnt_hi
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_readlane(v, l) hipemu::wave_exchange((v), (l))   // every lane of the wavefront executes it (the source lane is uniform)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_nontemporal_load(p) (*(p))
#define __builtin_nontemporal_store(v, p) (*(p) = (v))

// ---- runtime API subset
typedef int hipError_t;
typedef void *hipStream_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice };
enum { hipStreamNonBlocking = 1, hipHostMallocDefault = 0 };
static inline const char *hipGetErrorString(hipError_t) { return "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t, int) { *v = 3; return hipSuccess; }   // tiny "GPU": forces persistent loops to iterate
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (void *)1; return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int *least, int *greatest) { *least = 0; *greatest = -1; return hipSuccess; }
static inline hipError_t hipLaunchHostFunc(hipStream_t, void (*fn)(void *), void *arg) { fn(arg); return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t *s, unsigned, int) { *s = (void *)1; return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
typedef void *hipEvent_t;
enum { hipEventDefault = 0, hipEventDisableTiming = 2 };
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipDeviceGetPCIBusId(char *out, int len, int) { snprintf(out, (size_t)len, "0000:00:00.0"); return hipSuccess; }
static inline hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }   // the emulator runs everything synchronously
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { *e = (void *)1; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipMalloc(void **p, size_t n) { *p = aligned_alloc(256, (n + 255) & ~(size_t)255); return *p ? hipSuccess : hipErrorOutOfMemory; }
static inline hipError_t hipHostMalloc(void **p, size_t n, unsigned) { return hipMalloc(p, n); }
static inline hipError_t hipFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *f, size_t *t) { *f = (size_t)64 << 30; *t = (size_t)64 << 30; return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }

// ---- stream capture / graphs: while a capture is open, kernel launches are recorded (not executed); hipGraphLaunch runs them
namespace hipemu {
struct Graph { std::vector<std::function<void()>> nodes; };
extern Graph *g_capture;
}
typedef hipemu::Graph *hipGraph_t;
typedef hipemu::Graph *hipGraphExec_t;
typedef void *hipGraphNode_t;
enum hipStreamCaptureMode { hipStreamCaptureModeGlobal, hipStreamCaptureModeThreadLocal, hipStreamCaptureModeRelaxed };
static inline hipError_t hipStreamBeginCapture(hipStream_t, hipStreamCaptureMode) { hipemu::g_capture = new hipemu::Graph(); return hipSuccess; }
static inline hipError_t hipStreamEndCapture(hipStream_t, hipGraph_t *g) { *g = hipemu::g_capture; hipemu::g_capture = nullptr; return hipSuccess; }
static inline hipError_t hipGraphInstantiate(hipGraphExec_t *e, hipGraph_t g, hipGraphNode_t *, char *, size_t) { *e = new hipemu::Graph(*g); return hipSuccess; }
static inline hipError_t hipGraphLaunch(hipGraphExec_t e, hipStream_t) { for (auto &n : e->nodes) n(); return hipSuccess; }
static inline hipError_t hipGraphExecDestroy(hipGraphExec_t e) { delete e; return hipSuccess; }
static inline hipError_t hipGraphDestroy(hipGraph_t g) { delete g; return hipSuccess; }

template <class K, class... A>
static inline void hipLaunchKernelGGL(K kernel, dim3 grid, dim3 block, size_t, hipStream_t, A... args)
{
    if (hipemu::g_capture) {
        hipemu::g_capture->nodes.push_back([=]() { hipemu::launch(grid, block, [=]() { kernel(args...); }); });
        return;
    }
    hipemu::launch(grid, block, [=]() { kernel(args...); });
}
