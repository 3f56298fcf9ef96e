// Issue rate of the f32-input MFMAs from ONE wave per SIMD (and two) as a function of the number of independent
// accumulators: cycles per 2048-MAC unit (one 32x32x2, or two 16x16x4).  hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k32(float *out, long long *cyc, int iters)
{
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int NACC>
__global__ void k16(float *out, long long *cyc, int iters)
{
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.0f + threadIdx.x * 1e-4f;
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <class K>
static void run(const char *name, K kern, int threads, double units_per_iter)
{
    const int blocks = 256, iters = 2000;
    float *out;
    long long *cyc, h[256 * 8];
    hipMalloc(&out, blocks * threads * 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, blocks * (threads / 64) * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int i = 0; i < blocks * (threads / 64); ++i) mean += (double)h[i];
    mean /= blocks * (threads / 64);
    const int waves_per_simd = threads / 256;
    printf("%-34s waves/SIMD %d  cycles per 2048-MAC unit per wave %.1f  per SIMD %.1f   wall %.3f ms  (clock ~%.2f GHz)\n", name,
           waves_per_simd, mean / (iters * units_per_iter), mean / (iters * units_per_iter) / waves_per_simd, ms,
           mean / (ms * 1e6));
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int th : {256, 512}) {
        run("32x32x2  1 accumulator", k32<1>, th, 16);
        run("32x32x2  2 accumulators", k32<2>, th, 16);
        run("32x32x2  4 accumulators", k32<4>, th, 16);
        run("32x32x2  8 accumulators", k32<8>, th, 16);
        run("16x16x4  1 accumulator", k16<1>, th, 16);
        run("16x16x4  2 accumulators", k16<2>, th, 16);
        run("16x16x4  4 accumulators", k16<4>, th, 16);
        run("16x16x4  8 accumulators", k16<8>, th, 16);
        run("16x16x4 16 accumulators", k16<16>, th, 16);
    }
    return 0;
}
