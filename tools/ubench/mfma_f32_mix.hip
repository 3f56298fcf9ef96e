// Wall-clock cost of the denoiser loop's memory instructions next to its MFMAs (mfma_f32_rate / _power showed that cycle
// counts and clock trade against each other on this part: only TFLOP/s from hipEvents is comparable across variants).
// One workgroup of 8 waves per CU (2 per SIMD), each wave runs k-groups of the single-launch kernel's inner loop:
// NMB x NNB accumulators of v_mfma_f32_32x32x2_f32, per k-group (4 k-steps) NMB weight float4s from global memory
// (a [layers][rows] stream walked linearly, 2.75 MB per "layer" like the real one, shared by all workgroups -> L2 hits)
// and NNB B-fragment float4s from a k-interleaved LDS tile.  Variants select which loads exist and where they sit.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

enum { LD_NONE = 0, LD_LDS = 1, LD_GLB = 2, LD_BOTH = 3 };
enum { PL_TOP = 0, PL_SPREAD = 1, PL_FREE = 2 };

template <int NMB, int NNB, int LD, int PL>
__global__ __launch_bounds__(512) void k(const f32x4 *__restrict__ w, const float *__restrict__ init, float *out, int groups,
                                         int wrap_groups)
{
    __shared__ __attribute__((aligned(16))) float lds[256 * 66];
    for (int i = threadIdx.x; i < 256 * 66; i += blockDim.x) lds[i] = init[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hh = lane >> 5, c32 = lane & 31;
    f32x16 acc[NMB][NNB];
    for (int i = 0; i < NMB; ++i)
        for (int j = 0; j < NNB; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    f32x4 ring[4][NMB], bb[2][NNB];
    const f32x4 *ap[NMB];
    for (int i = 0; i < NMB; ++i) ap[i] = w + ((size_t)(wv * NMB + i) * wrap_groups) * 64 + lane;
    for (int s = 0; s < 4; ++s)
        for (int i = 0; i < NMB; ++i) ring[s][i] = ap[i][(size_t)s * 64];
    const float *T = lds + c32 * 8 + hh * 4;
    for (int j = 0; j < NNB; ++j) bb[0][j] = bb[1][j] = *reinterpret_cast<const f32x4 *>(T + 32 * 8 * j);
    int q = 0;
#pragma unroll 1
    for (int g = 0; g < groups; g += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            q = q + 1 < wrap_groups ? q + 1 : 0;
            if (LD & LD_GLB) {
#pragma unroll
                for (int i = 0; i < NMB; ++i) ring[(u + 2) & 3][i] = ap[i][(size_t)q * 64];
            }
            if (LD & LD_LDS) {
                const float *Tx = T + (q & 31) * (66 * 8);
#pragma unroll
                for (int j = 0; j < NNB; ++j) bb[(u + 1) & 1][j] = *reinterpret_cast<const f32x4 *>(Tx + 32 * 8 * j);
            }
            if (PL == PL_TOP) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < NMB; ++i)
#pragma unroll
                    for (int j = 0; j < NNB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][i][e], bb[u & 1][j][e], acc[i][j], 0, 0, 0);
            }
            if (PL == PL_SPREAD) {
                // one memory instruction behind each of the first MFMAs, the rest of the MFMAs after them
                constexpr int NG = ((LD & LD_GLB) ? NMB : 0), NL = ((LD & LD_LDS) ? NNB : 0);
#pragma unroll
                for (int m = 0; m < NG + NL; ++m) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    if (m < NG) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, 4 * NMB * NNB - 2 * (NG + NL), 0);
            }
            if (PL != PL_FREE) __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int i = 0; i < NMB; ++i)
        for (int j = 0; j < NNB; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float gauss()
{
    float s = 0;
    for (int i = 0; i < 12; ++i) s += (float)rand() / RAND_MAX;
    return s - 6.f;
}

static f32x4 *g_w;
static float *g_init, *g_out;

template <int NMB, int NNB, int LD, int PL>
static void run(const char *name, int wrap_groups)
{
    const int blocks = 256, groups = 8000;
    auto kern = k<NMB, NNB, LD, PL>;
    // every wave reads f32x4 indices < (8 waves * NMB rows) * wrap_groups * 64: must stay inside the 64 Mi-float buffer
    if ((size_t)8 * NMB * wrap_groups * 64 * 4 > ((size_t)64 << 20) || wrap_groups < 4) {
        printf("%dx%d %s: stream does not fit the buffer, skipped\n", NMB, NNB, name);
        return;
    }
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, g_w, g_init, g_out, groups, wrap_groups);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, g_w, g_init, g_out, groups, wrap_groups);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 3;
    const double flop = (double)blocks * 8 * groups * (4.0 * NMB * NNB) * 4096.0;
    printf("%dx%d  %-58s %7.3f ms  %6.1f TFLOP/s  %.3f of 157.3\n", NMB, NNB, name, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3);
}

int main()
{
    const size_t wfloats = (size_t)64 << 20;   // 256 MB of "weights"
    float *h = (float *)malloc(wfloats * 4);
    srand(3);
    for (size_t i = 0; i < wfloats; ++i) h[i] = (i < (1u << 22)) ? gauss() * 0.0625f : h[i & ((1u << 22) - 1)];
    hipMalloc(&g_w, wfloats * 4);
    hipMemcpy(g_w, h, wfloats * 4, hipMemcpyHostToDevice);
    for (int i = 0; i < 256 * 66; ++i) h[i] = gauss();
    hipMalloc(&g_init, 256 * 66 * 4);
    hipMemcpy(g_init, h, 256 * 66 * 4, hipMemcpyHostToDevice);
    hipMalloc(&g_out, 256 * 512 * 4);
    // wrap_groups: k-groups of one row block before its stream wraps.  64 = 64 KB per wave (L1 / L2 resident);
    // 4096 = 4 MB per row block, 16-32 row blocks = 64-128 MB walked by every workgroup: L2 / MALL traffic like the real stream
    const int W_SMALL = 64, W_BIG = 4096;
    run<2, 2, LD_NONE, PL_TOP>("no loads", W_SMALL);
    run<2, 2, LD_LDS, PL_TOP>("2 ds_read_b128 per 16 MFMAs, top", W_SMALL);
    run<2, 2, LD_GLB, PL_TOP>("2 global_load_x4 per 16 MFMAs, top, 64 KB streams", W_SMALL);
    run<2, 2, LD_GLB, PL_TOP>("2 global_load_x4 per 16 MFMAs, top, 4 MB streams", W_BIG);
    run<2, 2, LD_BOTH, PL_TOP>("both, top (the kernel's GEMM 2 / 3), 64 KB streams", W_SMALL);
    run<2, 2, LD_BOTH, PL_TOP>("both, top, 4 MB streams", W_BIG);
    run<2, 2, LD_BOTH, PL_SPREAD>("both, one load behind every 2nd MFMA, 4 MB streams", W_BIG);
    run<2, 2, LD_BOTH, PL_FREE>("both, compiler's placement, 4 MB streams", W_BIG);
    run<1, 2, LD_BOTH, PL_TOP>("both, top (GEMM 1: 8 MFMAs per group), 4 MB", W_BIG);
    run<1, 2, LD_BOTH, PL_SPREAD>("both, spread (GEMM 1), 4 MB", W_BIG);
    run<4, 2, LD_BOTH, PL_TOP>("both, top (4x2: 32 MFMAs per group), 4 MB", W_BIG);
    run<4, 2, LD_BOTH, PL_SPREAD>("both, spread (4x2), 4 MB", W_BIG);
    run<2, 1, LD_BOTH, PL_TOP>("both, top (the 32-frame form), 4 MB", W_BIG);
    return 0;
}
