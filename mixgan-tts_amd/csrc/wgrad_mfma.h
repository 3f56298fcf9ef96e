// Weight gradient of Conv1d / Linear on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   dW[co, ci, k] = alpha * sum_{b, l} dY[b, co, l] * (X[b, ci, l*stride + k - pad] + xvec[b, ci])
//
// Both operands are channel-major with the reduction axis (frames) contiguous, so the GEMM is
// "NT": A[m = co][kk = frame], B[kk = frame][n = ci].  A workgroup owns a 128(co) x 128(ci) tile
// of one tap and walks 64-frame chunks of the (batch, frame) axis; chunk tiles are staged in LDS
// with an odd row stride (65) so that the 32 lanes of an MFMA fragment -- 32 different rows, same
// column -- hit 32 different banks.  The frame axis is split across `nsplit` workgroups per tile
// (the output is tiny compared with the reduction: 393k outputs vs 16k frames); every split writes its
// partial tile with plain 256-byte row stores into scratch [nsplit][K][Co][Ci] and the finalize kernel
// sums the splits in a fixed order while it transposes to the [Co, Ci, K] parameter layout.  (The first
// version combined the partials with fp32 atomics into one [K][Co][Ci] buffer: 8.4 M atomics per launch,
// 15-30 us of a 60-100 us kernel, a memset in front, and a summation order that changed run to run.)
//
// Measured and rejected (round 2): frame-interleaved tiles (stride 68, position (f & 1) * 4 + ((f >> 1) & 3) inside every
// 8 frames) read with one ds_read_b128 per operand block and 4 k-steps -- 4 LDS reads per 16 MFMAs instead of 16.  The
// reads were conflict-free, but the staging stores of that layout are 4-way bank-conflicted (2-way here), and they sit
// between the two barriers of a chunk where nothing overlaps them: 253-259 us per launch against 221 us.
#pragma once
#include "common.h"
#include "wgrad_stream.h"
#include <cstdlib>

#define WG_FT 64
#define WG_RSA 65   // odd strides: the 32 lanes of a fragment read (32 rows, one column) hit 32 banks
#define WG_RSB 73

struct WgradArgs {
    const float *dy;    // [B, Co, Ldy] (batch stride dy_bs, row stride Ldy)
    const float *x;     // [B, Ci, Lx]  (batch stride x_bs,  row stride Lx)
    const float *xvec;  // optional [B, Ci]
    float *scratch;     // [nsplit][G][K][Co][Ci] partial sums
    long dy_bs, x_bs;
    int B, Co, Ci, Ldy, Lx, K, stride, pad;
    int chunks_per_b, nchunks, nsplit, ci_tiles;
    // G independent gradients of the same shape in one launch (the 20 residual layers of the denoiser): group g
    // reads dy + g*dy_gs and x + g*x_gs (a stride of 0 shares the operand); blockIdx.z = g*K + tap
    int G;
    long dy_gs, x_gs;
};

// VEC: 16-byte staging loads (needs stride 1, Ldy % 4 == 0, Lx % 4 == 0, 16-byte aligned bases, no xvec):
// the dY tile is 16 float4 per row; the X tile is read as the aligned 17-float4 window that contains
// the tap-shifted 64 frames and scattered into the LDS tile with the shift applied.
template <bool VEC>
__global__ __launch_bounds__(256, 2) void wgrad_mfma_kernel(WgradArgs a)
{
    // [A: 128 rows x 64 frames, stride 65 | B: 128 rows x 72 frames, stride 73], single-buffered (70.7 KB -> two
    // workgroups per CU).  The B tile holds the ALIGNED 72-frame window that contains the tap-shifted 64 frames; the
    // shift is applied when the fragments are read (an address offset), not when the tile is written.
    __shared__ float lds[128 * WG_RSA + 128 * WG_RSB];
    float *ldsA = lds, *ldsB = lds + 128 * WG_RSA;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int hh = lane >> 5, c32 = lane & 31;
    const int split = blockIdx.x;
    const int co0 = (blockIdx.y / a.ci_tiles) * 128;
    const int ci0 = (blockIdx.y % a.ci_tiles) * 128;
    const int grp = blockIdx.z / a.K;
    const int tap = blockIdx.z - grp * a.K;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    f32x4 va[VEC ? 8 : 1], vb[VEC ? 9 : 1];
    float sa[VEC ? 1 : 8], sb[VEC ? 1 : 8];  // scalar path stages in 4 batches of 8 (no prefetch, no spill)
    const int shift = tap - a.pad;
    const int shift4 = shift & ~3, rsh = VEC ? (shift & 3) : 0;  // aligned window start and residual (VEC)
    // Everything about a staging element that does not depend on the chunk is computed ONCE here: the staging code
    // used to redo ~30 integer instructions per element per chunk (divisions by 17, clamps, a branch per scattered
    // store), as many issue cycles as the chunk's MFMAs -- which is what held this kernel at half of the MFMA peak.
    const int c4A = tid & 15;
    int gA[VEC ? 8 : 1], gB[VEC ? 9 : 1], lB[VEC ? 9 : 1], c4B[VEC ? 9 : 1];
    unsigned okA = 0, okB = 0;   // bit k: the element's row exists (and, for B, the element is inside the 128 x 18 window)
    if (VEC) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int row = (tid >> 4) + 16 * k;
            gA[k] = min(co0 + row, a.Co - 1) * a.Ldy;
            if (co0 + row < a.Co) okA |= 1u << k;
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int idx = tid + k * 256;   // 128 rows x 18 float4 = 2304
            const int row = min(idx / 18, 127), c4 = idx - (idx / 18) * 18;
            gB[k] = min(ci0 + row, a.Ci - 1) * a.Lx;
            lB[k] = row * WG_RSB + 4 * c4;
            c4B[k] = c4;
            if (idx < 128 * 18 && ci0 + row < a.Ci) okB |= 1u << k;
        }
    }
    auto load_stage = [&](int chunk) {
        const int b = chunk / a.chunks_per_b;
        const int f0 = (chunk - b * a.chunks_per_b) * WG_FT;
        const float *dyb = a.dy + (size_t)grp * a.dy_gs + (size_t)b * a.dy_bs;
        const float *xb = a.x + (size_t)grp * a.x_gs + (size_t)b * a.x_bs;
        if (VEC) {
            const int f = f0 + 4 * c4A;
            const bool fok = f < a.Ldy;
            const int fcl = min(f, a.Ldy - 4);
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(dyb + gA[k] + fcl);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                va[k] = (fok && ((okA >> k) & 1u)) ? v : z;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int xf = f0 + shift4 + 4 * c4B[k];
                const bool ok = ((okB >> k) & 1u) && xf >= 0 && xf < a.Lx;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(xb + gB[k] + min(max(xf, 0), a.Lx - 4));
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                vb[k] = ok ? v : z;
            }
        } else {
#pragma unroll 1
            for (int kb = 0; kb < 32; kb += 8) {
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int idx = tid + (kb + k) * 256;  // 128 rows x 64 frames
                    const int row = idx >> 6, c = idx & 63;
                    const int f = f0 + c;
                    {
                        const int co = co0 + row;
                        const bool ok = co < a.Co && f < a.Ldy;
                        const float v = dyb[(size_t)min(co, a.Co - 1) * a.Ldy + min(f, a.Ldy - 1)];
                        sa[k] = ok ? v : 0.f;
                    }
                    {
                        const int ci = ci0 + row;
                        const int xf = f * a.stride + shift;
                        const bool ok = ci < a.Ci && f < a.Ldy && xf >= 0 && xf < a.Lx;
                        const int cic = min(ci, a.Ci - 1);
                        float v = xb[(size_t)cic * a.Lx + min(max(xf, 0), a.Lx - 1)];
                        if (a.xvec) v += a.xvec[(size_t)b * a.Ci + cic];
                        sb[k] = ok ? v : 0.f;
                    }
                }
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int idx = tid + (kb + k) * 256;
                    const int row = idx >> 6, c = idx & 63;
                    ldsA[row * WG_RSA + c] = sa[k];
                    ldsB[row * WG_RSB + c] = sb[k];
                }
            }
        }
    };
    auto store_stage = [&]() {
        if (VEC) {
            float *dA = ldsA + (tid >> 4) * WG_RSA + 4 * c4A;
#pragma unroll
            for (int k = 0; k < 8; ++k)
#pragma unroll
                for (int j = 0; j < 4; ++j) dA[k * 16 * WG_RSA + j] = va[k][j];   // immediate offsets off one base
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                if (k < 8 || tid + 8 * 256 < 128 * 18) {   // k = 8: only the first 256 of the 2304 window elements
#pragma unroll
                    for (int j = 0; j < 4; ++j) ldsB[lB[k] + j] = vb[k][j];
                }
            }
        }
    };

    int chunk = split;
    if (chunk < a.nchunks) {
        load_stage(chunk);  // the scalar path writes LDS itself
        store_stage();
    }
    __syncthreads();
    const float *A = ldsA + (wm * 64 + c32) * WG_RSA + hh;
    const float *Bt = ldsB + (wn * 64 + c32) * WG_RSB + hh + rsh;
    for (; chunk < a.nchunks; chunk += a.nsplit) {
        const bool more = chunk + a.nsplit < a.nchunks;
        if (VEC && more) load_stage(chunk + a.nsplit);  // global loads fly behind the MFMAs below
#pragma unroll 8
        for (int s = 0; s < WG_FT / 2; ++s) {
            const float a0 = A[2 * s], a1 = A[32 * WG_RSA + 2 * s];
            const float b0 = Bt[2 * s], b1 = Bt[32 * WG_RSB + 2 * s];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();  // every wave is done reading the tiles
        if (more) {
            if (VEC) store_stage();
            else load_stage(chunk + a.nsplit);
        }
        __syncthreads();
    }

    // acc[i][j][r]: co = co0 + wm*64 + i*32 + 8*(r>>2) + 4*hh + (r&3),  ci = ci0 + wn*64 + j*32 + c32
    float *dst = a.scratch + (((size_t)split * a.G + grp) * a.K + tap) * a.Co * a.Ci;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
            if (co >= a.Co) continue;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int ci = ci0 + wn * 64 + j * 32 + c32;
                if (ci < a.Ci) dst[(size_t)co * a.Ci + ci] = acc[i][j][r];
            }
        }
}

// scratch [nsplit][G][K][Co][Ci] -> dw[g] [Co][Ci][K] (= or +=), scaled.  A block owns 32 consecutive scratch elements;
// its 8 thread groups each add every 8th split (the small gradients of the discriminator run 25-128 splits: one
// thread walking them all is a chain of that many dependent-latency loads, 15-35 us for a few hundred KB), then the
// 8 sums are added in group order: a fixed summation order whatever the launch shape.
__global__ __launch_bounds__(256) void wgrad_finalize_kernel(const float *__restrict__ scratch, float *__restrict__ dw,
                                                             int Co, int Ci, int K, int G, long dw_gs, int nsplit,
                                                             float alpha, int accumulate)
{
    __shared__ float part[8][32];
    const size_t per = (size_t)Co * Ci * K, n = per * G;
    const int lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (size_t base = (size_t)blockIdx.x * 32; base < n; base += (size_t)gridDim.x * 32) {
        const size_t i = base + lane;
        float v0 = 0.f, v1 = 0.f;
        if (i < n) {
            int s = grp;
            for (; s + 8 < nsplit; s += 16) {
                v0 += scratch[(size_t)s * n + i];
                v1 += scratch[(size_t)(s + 8) * n + i];
            }
            if (s < nsplit) v0 += scratch[(size_t)s * n + i];
        }
        part[grp][lane] = v0 + v1;
        __syncthreads();
        if (grp == 0 && i < n) {
            float v = part[0][lane];
#pragma unroll
            for (int q = 1; q < 8; ++q) v += part[q][lane];
            v *= alpha;
            // scratch element (g, k, co, ci) -> dw[g][co][ci][k]: coalesced reads of every split, writes strided by K
            const size_t g = i / per, r = i - g * per;
            const size_t cc = r % ((size_t)Co * Ci);
            const int k = (int)(r / ((size_t)Co * Ci));
            float *o = dw + g * dw_gs + cc * K + k;
            *o = accumulate ? *o + v : v;
        }
        __syncthreads();
    }
}

// number of frame splits for a [Co, Ci, K] gradient: two workgroups per CU are resident (66.5 KB LDS each), keep
// the grid within ONE round of 512 workgroups -- 560 workgroups take two rounds, i.e. twice the time of 504
static inline int wgrad_nsplit(int Co, int Ci, int K, int G = 1)
{
    const int tiles = mg_cdiv(Co, 128) * mg_cdiv(Ci, 128) * K * G;
    // measured sweep (B=8, L=1000): 512 workgroups is the optimum for every shape with more than 8 tiles; the small
    // k=1 gradients (<= 8 tiles: 512x256, 256x256) are 10-25 % faster with 256 -- their finalize pass, which reads
    // nsplit copies of the output, is as long as the GEMM itself
    const int target = tiles <= 8 ? 256 : 512;
    const int n = target / tiles;
    return n < 1 ? 1 : n;
}
static inline size_t wgrad_scratch_floats(int Co, int Ci, int K, int G = 1)
{
    // either kernel may run (the streaming one needs aligned, stride-1 operands): size for both
    const size_t split = (size_t)wgrad_nsplit(Co, Ci, K, G) * G * Co * Ci * K;
    const size_t stream = wgrad_stream_scratch_floats(Co, Ci, K, G);
    return split > stream ? split : stream;
}

struct WgradShape {
    int B, Co, Ci, Ldy, Lx, K, stride, pad;
    long dy_bs, x_bs;  // 0 -> dense
    int G = 1;         // groups (see WgradArgs)
    long dy_gs = 0, x_gs = 0, dw_gs = 0;
    // optional bias gradient db[g][co] = alpha * sum_{b,l} dY_g[b, co, l] (group stride db_gs, 0 -> Co): the streaming
    // kernel produces it on the way; *db_done tells the caller whether it did (otherwise: a row-sum launch)
    float *db = nullptr;
    long db_gs = 0;
    bool *db_done = nullptr;
};

static int wgrad_launch(const WgradShape &s, const float *dy, const float *x, const float *xvec, float *dw,
                        float *scratch, float alpha, int accumulate, hipStream_t st)
{
    if (s.B <= 0 || s.Co <= 0 || s.Ci <= 0 || s.Ldy <= 0 || s.Lx <= 0 || s.K <= 0 || s.G <= 0) return MG_ERR_SHAPE;
    WgradArgs a;
    a.G = s.G;
    a.dy_gs = s.dy_gs;
    a.x_gs = s.x_gs;
    a.dy = dy;
    a.x = x;
    a.xvec = xvec;
    a.scratch = scratch;
    a.dy_bs = s.dy_bs ? s.dy_bs : (long)s.Co * s.Ldy;
    a.x_bs = s.x_bs ? s.x_bs : (long)s.Ci * s.Lx;
    a.B = s.B;
    a.Co = s.Co;
    a.Ci = s.Ci;
    a.Ldy = s.Ldy;
    a.Lx = s.Lx;
    a.K = s.K;
    a.stride = s.stride;
    a.pad = s.pad;
    a.chunks_per_b = mg_cdiv(s.Ldy, WG_FT);
    a.nchunks = a.chunks_per_b * s.B;
    a.ci_tiles = mg_cdiv(s.Ci, 128);
    int nsplit = wgrad_nsplit(s.Co, s.Ci, s.K, s.G);
    if (nsplit > a.nchunks) nsplit = a.nchunks;
    a.nsplit = nsplit;
    const size_t n = (size_t)s.Co * s.Ci * s.K * s.G;
    dim3 grid(nsplit, mg_cdiv(s.Co, 128) * a.ci_tiles, s.K * s.G);
    {   // big stride-1 "same" gradients: the streaming kernel (wgrad_stream.h); MG_WGRAD_STREAM=0 keeps the split kernel
        const char *env = std::getenv("MG_WGRAD_STREAM");
        const bool stream_on = !(env && env[0] == '0');
        const long dybs = a.dy_bs, xbs = a.x_bs;
        const WsShape w = wgrad_stream_shape(s.Co, s.Ci, s.K);
        const long long units = w.cfg ? (long long)wgrad_stream_tiles(w, s.Co, s.Ci, s.G) * s.B * mg_cdiv(s.Ldy, w.FT) : 0;
        const bool ok = stream_on && w.cfg && s.stride == 1 && !xvec && s.pad == (s.K - 1) / 2 && s.Ldy == s.Lx &&
                        s.Ldy % 4 == 0 && s.Ldy >= 4 && dybs % 4 == 0 && xbs % 4 == 0 && s.dy_gs % 4 == 0 &&
                        s.x_gs % 4 == 0 && ((((uintptr_t)dy | (uintptr_t)x) & 15) == 0) && units * 2 < (1ll << 31) &&
                        // every CU flushes a whole partial tile per run: worth it from ~8 units per CU on; smaller
                        // gradients (the skip / input / output projections) stay on the split kernel
                        units >= 8 * WS_NW;
        if (ok) {
            if (s.db_done) *s.db_done = s.db != nullptr;
            return wgrad_stream_launch(w, dy, x, dw, scratch, s.G, s.B, s.Co, s.Ci, s.Ldy, dybs, xbs, s.dy_gs, s.x_gs,
                                       s.dw_gs ? s.dw_gs : (long)s.Co * s.Ci * s.K, alpha, accumulate, st, s.db, s.db_gs);
        }
    }
    const bool vec = s.stride == 1 && !xvec && (s.Ldy % 4 == 0) && (s.Lx % 4 == 0) && (a.dy_bs % 4 == 0) &&
                     (a.x_bs % 4 == 0) && (a.dy_gs % 4 == 0) && (a.x_gs % 4 == 0) && ((((uintptr_t)dy | (uintptr_t)x) & 15) == 0) && s.Ldy >= 4 && s.Lx >= 4;
    if (vec) hipLaunchKernelGGL(wgrad_mfma_kernel<true>, grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL(wgrad_mfma_kernel<false>, grid, dim3(256), 0, st, a);
    MG_LAUNCH_CHECK();
    const int blocks = (int)((n + 31) / 32 < 8192 ? (n + 31) / 32 : 8192);
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(blocks), dim3(256), 0, st, scratch, dw, s.Co, s.Ci, s.K, s.G,
                       s.dw_gs ? s.dw_gs : (long)s.Co * s.Ci * s.K, nsplit, alpha, accumulate);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
