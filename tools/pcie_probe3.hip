// Measurement tool: device-to-host rate by STREAM -- streams created one after another (kept, or destroyed and re-created), each timed alone.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double rate(uint8_t *dst, const uint8_t *src, size_t total, hipStream_t s, hipMemcpyKind kind)
{
    const size_t piece = (size_t)25 << 20;
    CK(hipMemcpyAsync(dst, src, piece, kind, s));    // first use of the stream
    CK(hipStreamSynchronize(s));
    const double t0 = now();
    for (size_t o = 0; o < total; o += piece) CK(hipMemcpyAsync(dst + o, src + o, std::min(piece, total - o), kind, s));
    CK(hipStreamSynchronize(s));
    return total / 1e9 / (now() - t0);
}
int main()
{
    const size_t total = (size_t)380 << 20;
    uint8_t *dev = nullptr, *host = nullptr;
    CK(hipMalloc((void **)&dev, total));
    CK(hipMemset(dev, 1, total));
    CK(hipHostMalloc((void **)&host, total, hipHostMallocDefault));
    int least = 0, greatest = 0;
    CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
    printf("kept streams, normal priority (device-to-host | host-to-device GB/s):\n");
    std::vector<hipStream_t> keep;
    for (int i = 0; i < 12; i++) {
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        keep.push_back(s);
        printf("  stream %2d: %.1f | %.1f\n", i, rate(host, dev, total, s, hipMemcpyDeviceToHost), rate(dev, host, total, s, hipMemcpyHostToDevice));
    }
    printf("again, the same streams:\n");
    for (int i = 0; i < 12; i++) printf("  stream %2d: %.1f\n", i, rate(host, dev, total, keep[(size_t)i], hipMemcpyDeviceToHost));
    for (hipStream_t s : keep) CK(hipStreamDestroy(s));
    printf("created, used, destroyed -- one at a time:\n");
    for (int i = 0; i < 8; i++) {
        hipStream_t s;
        CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        printf("  stream %2d: %.1f\n", i, rate(host, dev, total, s, hipMemcpyDeviceToHost));
        CK(hipStreamDestroy(s));
    }
    printf("greatest priority, kept:\n");
    keep.clear();
    for (int i = 0; i < 6; i++) {
        hipStream_t s;
        CK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, greatest));
        keep.push_back(s);
        printf("  stream %2d: %.1f\n", i, rate(host, dev, total, s, hipMemcpyDeviceToHost));
    }
    // two streams at once: does the second take from the first?
    {
        const double t0 = now();
        const size_t piece = (size_t)25 << 20;
        for (size_t o = 0; o < total; o += piece) {
            CK(hipMemcpyAsync(host + o, dev + o, piece / 2, hipMemcpyDeviceToHost, keep[0]));
            CK(hipMemcpyAsync(host + o + piece / 2, dev + o + piece / 2, std::min(piece, total - o) - piece / 2, hipMemcpyDeviceToHost, keep[1]));
        }
        CK(hipStreamSynchronize(keep[0]));
        CK(hipStreamSynchronize(keep[1]));
        printf("  streams 0 and 1 together: %.1f\n", total / 1e9 / (now() - t0));
    }
    return 0;
}
