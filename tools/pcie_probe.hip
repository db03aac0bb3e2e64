// Measurement tool (not part of the library): device-to-host copy rates into page-locked memory on this box -- one hipMemcpyAsync, the same bytes in
// 25 MB pieces, the pieces split over two streams, and a kernel that stores to the mapped host buffer -- each repeated, fresh and reused buffers.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/pcie_probe tools/pcie_probe.hip && /tmp/pcie_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k_push(const u32x4 *__restrict__ src, u32x4 *__restrict__ dst, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) __builtin_nontemporal_store(src[i], dst + i);
}
__global__ void k_busy(float *p, int iters)
{
    float a = p[threadIdx.x];
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    p[threadIdx.x] = a;
}

int main(int argc, char **argv)
{
    const size_t total = (size_t)380 << 20, piece = (size_t)25 << 20;
    uint8_t *dev = nullptr;
    CK(hipMalloc((void **)&dev, total));
    CK(hipMemset(dev, 1, total));
    hipStream_t s0, s1, sk;
    CK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sk, hipStreamNonBlocking));
    float *busy = nullptr;
    CK(hipMalloc((void **)&busy, 4096));
    for (int fresh = 0; fresh < 3; fresh++) {
        uint8_t *host = nullptr;
        const unsigned flags = fresh == 2 ? hipHostMallocNonCoherent : hipHostMallocDefault;
        double t0 = now();
        CK(hipHostMalloc((void **)&host, total, flags));
        printf("buffer %d (%s): hipHostMalloc %.1f ms\n", fresh, fresh == 2 ? "non-coherent" : "default", (now() - t0) * 1e3);
        for (int rep = 0; rep < 3; rep++) {
            if (rep == 2) memset(host, 0, total);     // the CPU wrote the buffer (dirty lines in its caches)
            t0 = now();
            CK(hipMemcpyAsync(host, dev, total, hipMemcpyDeviceToHost, s0));
            CK(hipStreamSynchronize(s0));
            const double a = now() - t0;
            t0 = now();
            for (size_t o = 0; o < total; o += piece) CK(hipMemcpyAsync(host + o, dev + o, std::min(piece, total - o), hipMemcpyDeviceToHost, s0));
            CK(hipStreamSynchronize(s0));
            const double b = now() - t0;
            t0 = now();
            for (size_t o = 0; o < total; o += piece) {
                const size_t n = std::min(piece, total - o), h = n / 2;
                CK(hipMemcpyAsync(host + o, dev + o, h, hipMemcpyDeviceToHost, s0));
                CK(hipMemcpyAsync(host + o + h, dev + o + h, n - h, hipMemcpyDeviceToHost, s1));
            }
            CK(hipStreamSynchronize(s0));
            CK(hipStreamSynchronize(s1));
            const double c = now() - t0;
            double d[3];
            const int grids[3] = {32, 128, 512};
            for (int g = 0; g < 3; g++) {
                t0 = now();
                hipLaunchKernelGGL(k_push, dim3(grids[g]), dim3(256), 0, s0, (const u32x4 *)dev, (u32x4 *)host, total / 16);
                CK(hipStreamSynchronize(s0));
                d[g] = now() - t0;
            }
            // with the device busy on another stream
            hipLaunchKernelGGL(k_busy, dim3(4096), dim3(256), 0, sk, busy, 400000);
            t0 = now();
            CK(hipMemcpyAsync(host, dev, total, hipMemcpyDeviceToHost, s0));
            CK(hipStreamSynchronize(s0));
            const double e = now() - t0;
            t0 = now();
            hipLaunchKernelGGL(k_push, dim3(128), dim3(256), 0, s0, (const u32x4 *)dev, (u32x4 *)host, total / 16);
            CK(hipStreamSynchronize(s0));
            const double f = now() - t0;
            CK(hipStreamSynchronize(sk));
            const double gb = total / 1e9;
            printf("  rep %d%s: one copy %.1f GB/s, 25 MB pieces %.1f, pieces over two streams %.1f, kernel stores (32/128/512 workgroups) %.1f / %.1f / %.1f; device busy: copy %.1f, kernel %.1f\n",
                   rep, rep == 2 ? " (after a CPU memset)" : "", gb / a, gb / b, gb / c, gb / d[0], gb / d[1], gb / d[2], gb / e, gb / f);
        }
        // host-to-device for reference
        t0 = now();
        CK(hipMemcpyAsync(dev, host, total, hipMemcpyHostToDevice, s0));
        CK(hipStreamSynchronize(s0));
        printf("  host-to-device one copy %.1f GB/s\n", total / 1e9 / (now() - t0));
        CK(hipHostFree(host));
    }
    return 0;
}
