// How does v_cvt_pk_u8_f32 round?  The encoders pack reconstructed pixels with it; today every input is an integer-valued float, so
// only the saturation matters.  Feeding it x / 256 + 128 directly (dropping a v_floor_f32 per pixel of the i-frame closed loop)
// needs its rounding rule for fractional inputs.  Prints the byte for inputs around the integers, halves and the range ends, and
// classifies the rule.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
__global__ void k(const float *in, unsigned *out, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 0u, 0u);
}
int main()
{
    const int N = (258 + 4) * 256 + 1;
    float *h = (float *)malloc(N * sizeof(float));
    for (int i = 0; i < N; i++) h[i] = -3.0f + i / 256.0f;        // -3 .. 259 in steps of 1/256
    float *d; unsigned *o;
    hipMalloc(&d, N * sizeof(float)); hipMalloc(&o, N * sizeof(unsigned));
    hipMemcpy(d, h, N * sizeof(float), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3((N + 255) / 256), dim3(256), 0, 0, d, o, N);
    unsigned *r = (unsigned *)malloc(N * sizeof(unsigned));
    hipMemcpy(r, o, N * sizeof(unsigned), hipMemcpyDeviceToHost);
    long bad_rne = 0, bad_trunc = 0, bad_floor = 0, bad_halfup = 0;
    auto sat = [](float v) { return v <= 0.0f ? 0u : (v >= 255.0f ? 255u : (unsigned)v); };
    for (int i = 0; i < N; i++) {
        float x = h[i];
        bad_rne += r[i] != sat(nearbyintf(x));
        bad_trunc += r[i] != sat(truncf(x));
        bad_floor += r[i] != sat(floorf(x));
        bad_halfup += r[i] != sat(floorf(x + 0.5f));
    }
    printf("v_cvt_pk_u8_f32 over %d inputs in [-3, 259], step 1/256: mismatches vs round-to-nearest-even %ld, truncate %ld, floor %ld, round-half-up %ld\n",
           N, bad_rne, bad_trunc, bad_floor, bad_halfup);
    const float probes[] = {-0.75f, -0.5f, -0.25f, 0.25f, 0.5f, 0.75f, 1.5f, 2.5f, 3.5f, 254.5f, 254.75f, 255.25f, 255.5f, 256.0f, 1e9f, -1e9f};
    for (float p : probes) {
        int i = (int)lrintf((p + 3.0f) * 256.0f);
        if (i >= 0 && i < N && h[i] == p) printf("  %8.2f -> %u\n", p, r[i]);
    }
    // the formula the i-frame encoder would use: byte(x / 256 + 127.501953125) == clamp(floor(x / 256) + 128, 0, 255) for every integer |x| < 2^17
    return 0;
}
