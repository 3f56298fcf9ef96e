// What does the fp32 matrix pipe deliver in WALL-CLOCK terms when its operands are not zeros?  mfma_f32_rate.hip showed
// that the cycle count per MFMA and the clock trade against each other (48 cycles at 1.71 GHz == 64 cycles at 2.35 GHz ==
// 157 TFLOP/s): the part regulates to a power / throughput target, so "cycles per MFMA" is not the ceiling -- TFLOP/s on
// realistic data is.  Pure v_mfma_f32_32x32x2_f32 loops, 2 waves per SIMD on all 256 CUs, 8 independent A/B register
// pairs, 4 accumulators, operands: (z) all zero, (c) one constant, (r) random N(0,1)-like per lane and per register,
// (h) random with half the lanes zero.  Prints TFLOP/s from hipEvents and the clock implied by clock64().
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k(const float *__restrict__ src, float *out, long long *cyc, int iters)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = src[(size_t)tid * 16 + i];
        b[i] = src[(size_t)tid * 16 + 8 + i];
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], b[(u + i) & 7], acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[tid] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}

static float gauss()
{
    float s = 0;
    for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX;
    return s - 6.f;
}

int main()
{
    const int blocks = 256, threads = 512, iters = 4000;
    const size_t n = (size_t)blocks * threads * 16;
    float *h = (float *)malloc(n * 4), *src, *out;
    long long *cyc, hc[2048];
    hipMalloc(&src, n * 4);
    hipMalloc(&out, blocks * threads * 4);
    hipMalloc(&cyc, 2048 * 8);
    const char *names[] = {"(z) all-zero operands", "(c) one constant (1.0)", "(s) small random, |x| ~ 1e-3", "(r) random N(0,1)",
                           "(h) random N(0,1), half the lanes zero", "(w) A = N(0, 1/16) weights, B = N(0,1) activations"};
    for (int mode = 0; mode < 6; ++mode) {
        srand(1);
        for (size_t i = 0; i < n; ++i) {
            const bool is_a = (i & 15) < 8;
            float v = 0.f;
            if (mode == 1) v = 1.0f;
            if (mode == 2) v = 1e-3f * gauss();
            if (mode == 3) v = gauss();
            if (mode == 4) v = ((i >> 4) & 1) ? gauss() : 0.f;
            if (mode == 5) v = is_a ? gauss() / 16.f : gauss();
            h[i] = v;
        }
        hipMemcpy(src, h, n * 4, hipMemcpyHostToDevice);
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0);
        for (int rep = 0; rep < 5; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, src, out, cyc, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        ms /= 5;
        hipMemcpy(hc, cyc, 2048 * 8, hipMemcpyDeviceToHost);
        double mean = 0;
        for (int i = 0; i < 2048; ++i) mean += (double)hc[i];
        mean /= 2048;
        const double flop = (double)blocks * 8 * iters * 32 * 4096.0;
        printf("%-52s %.3f ms  %.1f TFLOP/s (%.3f of 157.3)  %.1f cycles per MFMA per SIMD, clock ~%.2f GHz\n", names[mode], ms,
               flop / ms / 1e9, flop / ms / 1e9 / 157.3, mean / (iters * 32.0) / 2, mean / (ms * 1e6));
    }
    return 0;
}
