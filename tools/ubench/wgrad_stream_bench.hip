// Stand-alone timing of the streaming weight-gradient kernel (mixgan-tts_amd/csrc/wgrad_stream.h) on the residual
// stack's shapes at B=8, L=1000, 20 layers, with per-phase cycle counts of every wave (WS_TIMING).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DWS_TIMING tools/ubench/wgrad_stream_bench.hip -o tools/ubench/ws
#include "../../mixgan-tts_amd/csrc/wgrad_stream.h"
#include <cstdio>
#include <cstdlib>
#include <vector>

template <class C>
static void run(const char *name, int G, int B, int Co, int Ci, int L, long dy_bs, long dy_gs, long x_bs, long x_gs,
                const float *dy, const float *x, float *dw, float *scr)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w)
        wgrad_stream_launch_k<C>(dy, x, dw, scr, G, B, Co, Ci, L, dy_bs, x_bs, dy_gs, x_gs, (long)Co * Ci * C::K, 1.f, 0, 0);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w)
        wgrad_stream_launch_k<C>(dy, x, dw, scr, G, B, Co, Ci, L, dy_bs, x_bs, dy_gs, x_gs, (long)Co * Ci * C::K, 1.f, 0, 0);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = 2.0 * G * Co * Ci * C::K * (double)B * L;
    std::vector<long long> h(256 * 8 * 4);
    hipMemcpy(h.data(), ws_timing_buffer, h.size() * 8, hipMemcpyDeviceToHost);
    double t[4] = {0, 0, 0, 0};
    for (size_t i = 0; i < h.size(); ++i) t[i & 3] += (double)h[i];
    const double tot = t[0] + t[1] + t[2] + t[3];
    printf("%s: %.1f us (kernel + finalize)  %.1f TFLOP/s  phases: load-issue %.1f%%  mfma %.1f%%  store+flush %.1f%%  barrier %.1f%%"
           "  (mean cycles per wave %.0f)\n",
           name, ms * 1e3, flop / ms * 1e-9, 100 * t[0] / tot, 100 * t[1] / tot, 100 * t[2] / tot, 100 * t[3] / tot,
           tot / (256 * 8));
}

int main(int argc, char **argv)
{
    const float amp = argc > 1 ? (float)atof(argv[1]) : 1.f;   // 0: all-zero operands (no data toggling)
    const int NL = 20, B = 8, C = 256, L = 1000;
    const size_t CL = (size_t)C * L;
    float *dz, *h, *dw, *scr;
    const size_t ndz = (size_t)B * NL * 2 * CL, nh = (size_t)NL * B * CL;
    hipMalloc(&dz, ndz * 4);
    hipMalloc(&h, nh * 4);
    hipMalloc(&dw, (size_t)NL * 2 * C * C * 3 * 4);
    hipMalloc(&scr, wgrad_stream_scratch_floats(2 * C, C, 3, NL) * 4 + (128 << 20));
    hipMalloc(&ws_timing_buffer, 256 * 8 * 4 * 8);
    std::vector<float> r(1 << 20);
    for (auto &v : r) v = amp * ((float)rand() / RAND_MAX - 0.5f);
    for (size_t o = 0; o < ndz; o += r.size()) hipMemcpy(dz + o, r.data(), std::min(r.size(), ndz - o) * 4, hipMemcpyHostToDevice);
    for (size_t o = 0; o < nh; o += r.size()) hipMemcpy(h + o, r.data(), std::min(r.size(), nh - o) * 4, hipMemcpyHostToDevice);
    // k=3 conv of the 20 layers: dy = dz_all [B][NL*2C][L], x = h_all [NL][B][C][L]
    run<WsCfgK3>("dW3  G=20 512x256x3", NL, B, 2 * C, C, L, (long)NL * 2 * CL, (long)2 * CL, (long)CL, (long)B * CL, dz, h, dw, scr);
    // output conv, top rows: dy slots of C rows, x = g_all
    run<WsCfgK1>("dWo  G=20 256x256x1", NL, B, C, C, L, (long)NL * 2 * CL, (long)2 * CL, (long)CL, (long)B * CL, dz, h, dw, scr);
    run<WsCfgK1Wide>("dWo  G=20 256x256x1 wide", NL, B, C, C, L, (long)NL * 2 * CL, (long)2 * CL, (long)CL, (long)B * CL, dz, h, dw, scr);
    // conditioner projections of all layers as one gradient: [NL*C, H]
    run<WsCfgK1>("dWc  G=1 5120x256x1", 1, B, NL * C, C, L, (long)NL * 2 * CL, 0, (long)CL, 0, dz, h, dw, scr);
    run<WsCfgK1Wide>("dWc  G=1 5120x256x1 wide", 1, B, NL * C, C, L, (long)NL * 2 * CL, 0, (long)CL, 0, dz, h, dw, scr);
    return 0;
}
