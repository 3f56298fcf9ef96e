// What costs a single wave its 64-cycle MFMA cadence?  One k-group of the denoiser's loop = 8 x v_mfma_f32_32x32x2_f32 on
// 2 accumulators with (a) operands fixed, (b) a fresh A/B register per MFMA, (c) + 2 ds_read2_b32 per group,
// (d) + 2 global_load_dwordx4 per group, (e) c + d.  One wave per SIMD, 256 workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(512) void k(const f32x4 *__restrict__ w, float *out, long long *cyc, int iters)
{
    __shared__ float lds[256 * 36];
    for (int i = threadIdx.x; i < 256 * 36; i += blockDim.x) lds[i] = i * 1e-4f;
    __syncthreads();
    const int lane = threadIdx.x & 63, hh = lane >> 5, c32 = lane & 31;
    f32x16 acc[2];
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    f32x4 ring[4][2];
    float bb[2][4];
    const f32x4 *ap0 = w + lane, *ap1 = w + 96 * 64 + lane;
    for (int s = 0; s < 4; ++s) {
        ring[s][0] = ap0[s * 64];
        ring[s][1] = ap1[s * 64];
    }
    const float *T = lds + hh * 36 + c32;
    for (int e = 0; e < 4; ++e) bb[0][e] = bb[1][e] = T[2 * e * 36];
    long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = (it * 4 + u) & 63;
            if (MODE == 3 || MODE >= 4) {
                ring[(u + 3) & 3][0] = ap0[(size_t)q * 64];
                ring[(u + 3) & 3][1] = ap1[(size_t)q * 64];
            }
            if (MODE == 2 || MODE >= 4) {
                const float *Tx = T + (q & 31) * 8 * 36;
#pragma unroll
                for (int e = 0; e < 4; ++e) bb[(u + 1) & 1][e] = Tx[2 * e * 36];
            }
            if (MODE < 5) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a0 = MODE == 0 ? ring[0][0][0] : ring[u][0][e], a1 = MODE == 0 ? ring[0][1][0] : ring[u][1][e];
                const float b = MODE == 0 ? bb[0][0] : bb[u & 1][e];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b, acc[1], 0, 0, 0);
                if (MODE < 5) __builtin_amdgcn_sched_barrier(0);
            }
            if (MODE >= 5) {
                // one memory instruction behind each of the first MFMAs of the group, in the shadow of its 64 cycles
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                     // 1 MFMA
                    if (g < 2) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);          // 1 VMEM read
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);               // 1 DS read
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 2; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 8 + threadIdx.x / 64] = t1 - t0;
}

template <class K>
static void run(const char *name, K kern, const f32x4 *w, int threads)
{
    const int blocks = 256, iters = 1000, nw = threads / 64;
    float *out;
    long long *cyc, h[2048];
    hipMalloc(&out, blocks * 512 * 4);
    hipMalloc(&cyc, blocks * 8 * 8);
    hipMemset(cyc, 0, blocks * 8 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, w, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
    double mean = 0;
    for (int b = 0; b < blocks; ++b)
        for (int i = 0; i < nw; ++i) mean += (double)h[b * 8 + i];
    mean /= blocks * nw;
    printf("%d waves/SIMD  %-58s %.1f cycles per MFMA per SIMD\n", nw / 4, name, mean / (iters * 32.0) / (nw / 4));
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    f32x4 *w;
    hipMalloc(&w, 4 << 20);
    hipMemset(w, 0, 4 << 20);
    for (int th : {256, 512}) {
        run("(a) fixed operands", k<0>, w, th);
        run("(b) a fresh A / B register per MFMA", k<1>, w, th);
        run("(c) b + 2 ds_read2_b32 per 8 MFMAs", k<2>, w, th);
        run("(d) b + 2 global_load_dwordx4 per 8 MFMAs", k<3>, w, th);
        run("(e) b + both", k<4>, w, th);
        run("(f) e, one memory instruction behind each of 4 MFMAs", k<5>, w, th);
    }
    return 0;
}
