// Head and tail of Denoiser.forward as two small fused kernels on the fp32 MFMA (same tiling idea as
// resblock_fused.h: a workgroup owns 64 frames and keeps the whole channel column in LDS):
//   head:  x0 = relu(W_in x_t + b_in)                                   model/modules.py:430-431
//   tail:  out = W_out relu(W_skip (sum skips)/sqrt(NL) + b_skip) + b_out   model/modules.py:441-444
// The generic conv kernel spends ~45 us per launch on these short-K GEMMs (latency-bound chunk
// loop); here the input tile is resident, so each is one staging pass plus a short MFMA loop, and
// the tail's intermediate never touches HBM.
#pragma once
#include "common.h"
#include "resblock_fused.h"

#define PW_RS 68  // LDS row stride of the fp32 tiles (64 frames + 4 pad)

struct HeadArgs {
    const float *x_t;   // [B, M, L]
    const float *w;     // packed PLAIN [256 rows, K = 96 (M = 80 padded to 3 chunks of 32)]
    const float *bias;  // [256]
    float *x0;          // [B, 256, L]
    int M, L, tiles_per_b;
};

__global__ __launch_bounds__(512, 2) void denoiser_head_kernel(HeadArgs a)
{
    __shared__ float xT[96 * PW_RS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hh = lane >> 5, c32 = lane & 31;
    const int b = blockIdx.x / a.tiles_per_b;
    const int l0 = (blockIdx.x - b * a.tiles_per_b) * RB_NT;
    const int L = a.L;
    const float *xb = a.x_t + (size_t)b * a.M * L;
#pragma unroll
    for (int k = 0; k < 12; ++k) {  // 96 rows x 64 frames = 12 x 512
        const int idx = tid + k * 512;
        const int row = idx >> 6, c = idx & 63;
        const int f = l0 + c;
        const float v = xb[(size_t)min(row, a.M - 1) * L + min(f, L - 1)];
        xT[row * PW_RS + c] = (row < a.M && f < L) ? v : 0.f;
    }
    f32x16 acc[1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = a.bias[w * 32 + 8 * (r >> 2) + 4 * hh + (r & 3)];
        acc[0][0][r] = bv;
        acc[0][1][r] = bv;
    }
    __syncthreads();
    rb_mfma_loop<1, 2, 1, PW_RS, 3>(acc, reinterpret_cast<const f32x4 *>(a.w) + (size_t)w * 12 * 64 + lane, 0,
                                    xT + hh * PW_RS, c32);
    float *ob = a.x0 + (size_t)b * RB_C * L;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = w * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int f = l0 + 32 * j + c32;
            if (f < L) ob[(size_t)row * L + f] = fmaxf(acc[0][j][r], 0.f);
        }
    }
}

struct TailArgs {
    const float *skip;   // [B, 256, L] sum of the layers' skip outputs
    const float *wskip;  // packed PLAIN [256 rows, K = 256]
    const float *bskip;  // [256]
    const float *wout;   // packed PLAIN [M rows padded to 128, K = 256]
    const float *bout;   // [M]
    float *y_save;       // optional [B, 256, L]: relu output, kept for the backward's ReLU mask
    float *out;          // [B, M, L]
    float alpha;         // 1/sqrt(NL)
    int M, L, tiles_per_b;
};

__global__ __launch_bounds__(512, 2) void denoiser_tail_kernel(TailArgs a)
{
    __shared__ float lds[2 * RB_C * PW_RS];
    float *sT = lds, *yT = lds + RB_C * PW_RS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hh = lane >> 5, c32 = lane & 31;
    const int b = blockIdx.x / a.tiles_per_b;
    const int l0 = (blockIdx.x - b * a.tiles_per_b) * RB_NT;
    const int L = a.L;
    const float *sb = a.skip + (size_t)b * RB_C * L;
#pragma unroll
    for (int k = 0; k < 32; ++k) {  // 256 rows x 64 frames = 32 x 512
        const int idx = tid + k * 512;
        const int row = idx >> 6, c = idx & 63;
        const int f = l0 + c;
        const float v = sb[(size_t)row * L + min(f, L - 1)];
        sT[row * PW_RS + c] = f < L ? v * a.alpha : 0.f;
    }
    f32x16 acc[1][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float bv = a.bskip[w * 32 + 8 * (r >> 2) + 4 * hh + (r & 3)];
        acc[0][0][r] = bv;
        acc[0][1][r] = bv;
    }
    __syncthreads();
    rb_mfma_loop<1, 2, 1, PW_RS>(acc, reinterpret_cast<const f32x4 *>(a.wskip) + (size_t)w * 32 * 64 + lane, 0,
                                 sT + hh * PW_RS, c32);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = w * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float v = fmaxf(acc[0][j][r], 0.f);
            yT[row * PW_RS + 32 * j + c32] = v;
            const int f = l0 + 32 * j + c32;
            if (a.y_save && f < L) a.y_save[((size_t)b * RB_C + row) * L + f] = v;
        }
    }
    __syncthreads();
    // output projection: M rows in 32-row blocks; (row block, 32-frame block) pairs spread over the waves
    const int mblocks = (a.M + 31) / 32;
    for (int job = w; job < mblocks * 2; job += 8) {
        const int mb = job >> 1, nb = job & 1;
        f32x16 o[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mb * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
            o[0][0][r] = row < a.M ? a.bout[row] : 0.f;
        }
        rb_mfma_loop<1, 1, 1, PW_RS>(o, reinterpret_cast<const f32x4 *>(a.wout) + (size_t)mb * 32 * 64 + lane, 0,
                                     yT + hh * PW_RS, 32 * nb + c32);
        const int f = l0 + 32 * nb + c32;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = mb * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
            if (row < a.M && f < L) a.out[((size_t)b * a.M + row) * L + f] = o[0][0][r];
        }
    }
}
