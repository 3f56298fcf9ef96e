// Ablation micro-benchmark for resblock_split_kernel (diagnostic only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../../mixgan-tts_amd/csrc/resblock_split.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
int main(int argc, char **argv)
{
    const int B = 16, L = 1000, C = 256;
    const size_t act = (size_t)B * C * L;
    float *xa, *xb, *skip, *vec, *bias;
    __bf16 *condS, *w;
    CK(hipMalloc(&condS, act * 4)); CK(hipMalloc(&xa, act * 4)); CK(hipMalloc(&xb, act * 4)); CK(hipMalloc(&skip, act * 4));
    const size_t wel = (size_t)(8 * 16 + 16 * 48 + 16 * 16) * 1024;  // bf16 elements per layer
    CK(hipMalloc(&w, wel * 2 * 20)); CK(hipMalloc(&vec, B * C * 4)); CK(hipMalloc(&bias, 4096 * 4));
    std::vector<unsigned short> h(act * 2);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned short)(0x3c00 + ((i * 2654435761u >> 9) & 0x3ff)) ^ ((i & 1) << 15);
    CK(hipMemcpy(condS, h.data(), act * 4, hipMemcpyHostToDevice));
    std::vector<unsigned short> hw(wel * 20);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = (unsigned short)(0x3a00 + ((i * 40503u >> 5) & 0x1ff)) ^ (((i >> 3) & 1) << 15);
    CK(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(xa, 0, act * 4)); CK(hipMemset(skip, 0, act * 4)); CK(hipMemset(vec, 0, B * C * 4)); CK(hipMemset(bias, 0, 4096 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < iters; ++it) {
            const int l = it % 20;
            __bf16 *lw = w + (size_t)l * wel;
            ResSplitArgs a;
            a.condS = condS; a.x_in = (it & 1) ? xb : xa; a.x_out = (it & 1) ? xa : xb; a.skip = skip;
            a.wc = lw; a.w3 = lw + 8 * 16 * 1024; a.wo = lw + (8 * 16 + 16 * 48) * 1024;
            a.bc = bias; a.b3 = bias + 256; a.bo = bias + 1024; a.hvec = vec; a.dvec = vec;
            a.L = L; a.tiles_per_b = (L + 63) / 64; a.first = (l == 0);
            hipLaunchKernelGGL(resblock_split_kernel, dim3(a.tiles_per_b * B), dim3(512), 0, 0, a);
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.1f us/launch  (%.1f TFLOP/s algorithmic)\n", argv[0], ms * 1e3 / iters, 1179648.0 * B * L / (ms * 1e-3 / iters) / 1e12);
    }
    return 0;
}
