// Ablation micro-benchmark for resblock_fused_kernel (diagnostic only; not part of the library).
// Build variants with -DRB_ABLATE_A (no weight stream) / -DRB_ABLATE_B (no LDS operand reads).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstdlib>
#include "../../mixgan-tts_amd/csrc/resblock_fused.h"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main(int argc, char **argv)
{
    const int B = 16, L = 1000, C = 256;
    const size_t act = (size_t)B * C * L;
    float *cond, *xa, *xb, *skip, *w, *vec;
    const int layout = argc > 3 ? atoi(argv[3]) : 0;
    CK(hipMalloc(&cond, act * 4));
    if (layout == 0) {
        CK(hipMalloc(&xa, act * 4)); CK(hipMalloc(&xb, act * 4)); CK(hipMalloc(&skip, act * 4));
    } else {  // the library's workspace carve-up: one allocation, x | skip | y back to back after the small vectors
        float *ws; CK(hipMalloc(&ws, (106496 + 3 * act + 1024) * 4));
        xa = ws + 106496; skip = xa + act; xb = skip + act;
    }
    const size_t wfl = 8 * 32 * 256 + 16 * 96 * 256 + 16 * 32 * 256 + 4096;
    CK(hipMalloc(&w, wfl * 4 * 20)); CK(hipMalloc(&vec, B * C * 4));
    std::vector<float> h(act);
    const int mode = argc > 1 ? atoi(argv[1]) : 0;
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (float)((st >> 11) & 0xffffff) / 16777216.f; };
    auto gauss = [&]() { float u = rnd() + 1e-7f, v = rnd(); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); };
    for (size_t i = 0; i < act; ++i) h[i] = mode == 0 ? (float)((i * 2654435761u >> 8) & 0xffff) / 65536.f - 0.5f : (mode == 1 ? gauss() : 0.f);
    CK(hipMemcpy(cond, h.data(), act * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(xa, h.data(), act * 4, hipMemcpyHostToDevice));
    CK(hipMemset(skip, 0, act * 4));
    std::vector<float> hw(wfl * 20);
    for (size_t i = 0; i < hw.size(); ++i) hw[i] = mode == 0 ? ((float)((i * 40503u >> 4) & 0xfff) / 4096.f - 0.5f) * 0.06f : (mode == 1 ? gauss() * 0.04f : 0.f);
    CK(hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(vec, 0, B * C * 4));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = argc > 2 ? atoi(argv[2]) : 200;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        for (int it = 0; it < iters; ++it) {
            const int l = it % 20;
            float *lw = w + (size_t)l * wfl;
            ResArgs a;
            a.cond = cond; a.x_in = (it & 1) ? xb : xa; a.x_out = (it & 1) ? xa : xb; a.skip = skip;
            a.wc = lw; a.w3 = lw + 8 * 32 * 256; a.wo = lw + 8 * 32 * 256 + 16 * 96 * 256;
            a.bc = lw + wfl - 4096; a.b3 = a.bc + 256; a.bo = a.bc + 1024;
            a.hvec = vec; a.dvec = vec; a.h_save = a.sig_save = a.tnh_save = a.g_save = nullptr;
            a.L = L; a.tiles_per_b = (L + 63) / 64; a.first = (l == 0);
            hipLaunchKernelGGL((resblock_fused_kernel<true, false>), dim3(a.tiles_per_b * B), dim3(512), 0, 0, a);
        }
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%s: %.1f us/launch  (%.1f TFLOP/s algorithmic)\n", argv[0], ms * 1e3 / iters,
               1179648.0 * B * L / (ms * 1e-3 / iters) / 1e12);
    }
    return 0;
}
