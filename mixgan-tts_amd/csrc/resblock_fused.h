// One gated residual layer of the Denoiser (model/blocks.py:1157-1176) as ONE kernel, gfx950.
//
//   h = Wc cond + bc + x + (Wd s [+ Wp spk])                     GEMM 1   M=256  K=256
//   g = sigmoid(z[:C]) * tanh(z[C:]),  z = W3 (*) h + b3          GEMM 2   M=512  K=768 (k=3)
//   o = Wo g + bo;  x' = (o[:C] + x + Wd s)/sqrt2;  skip += o[C:] GEMM 3   M=512  K=256
//
// A workgroup (512 threads = 8 waves, 2 per SIMD) owns 64 frames of one utterance and keeps the
// whole 256-channel column of that tile in LDS across the three GEMMs, so h and g never touch
// HBM: per frame and layer the kernel reads cond, x, skip and writes x', skip (5 KB instead of
// the 10 KB of the three-launch form).  The k=3 convolution needs h on a 1-frame halo each side;
// GEMM 1 therefore produces 66 columns (the third 32-column MFMA block carries 2 useful columns:
// 5 % extra MFMAs per layer, no inter-workgroup exchange).  x is double-buffered across layers
// (x_in -> x_out) because neighbouring tiles read each other's halo frames.
//
// LDS (143,488 B of the CU's 160 KiB, one workgroup per CU):
//   condT [256][72] (+32 pad)  col j <-> frame l0-4+j   (16-byte aligned rows for float4 staging)
//   hT    [256][68]            col j <-> frame l0-1+j
//   gT    [256][64]            aliases condT once GEMM 1 is done
// MFMA operands: weights stream global/L2 -> VGPR in fragment order (conv_mfma.h packing);
// activations come from the LDS tiles with one conflict-free ds_read_b32 per 32x32x2 MFMA; the
// tap shift of the k=3 conv is an address offset into hT.
//
// The epilogue addends of GEMM 3 (x, skip, biases, step vector) are loaded before its k-loop and
// consumed after it, so their latency hides behind 512 MFMAs per wave; GEMM 1's accumulators
// are initialised with x + bias + step vector while the cond tile is being staged.
#pragma once
#include "common.h"

#define RB_C 256
#define RB_NT 64
#define RB_RSC 72
#define RB_RSH 68
#define RB_RSG 64
#define RB_COND_FLOATS (RB_C * RB_RSC + 32)
#define RB_H_FLOATS (RB_C * RB_RSH)
#define RB_LDS_FLOATS (RB_COND_FLOATS + RB_H_FLOATS)

// Narrow-tile variants for launches whose 64-frame tiling would leave most CUs idle (one utterance, small
// batches, the 8-per-GPU training shard): a workgroup owns 30 or 32 output frames, every GEMM-2/3 has a single
// 32-column n-block and the LDS footprint is 78 KB.
//   NTU = 30: the halo frame on each side + 30 outputs = exactly one 32-column block in GEMM 1 too (6.7 % of
//             the columns are halo); hT columns 32, 33 (read by the two scratch output columns) are zeroed.
//   NTU = 32: 34 h columns -> GEMM 1 runs a second n-block for 2 columns (+20 % MFMAs), but 32 divides the
//             frame axis 6 % finer: B=8, L=1000 is exactly 256 workgroups = one per CU instead of 272.
//   condT [256][40] (+32)  col 3 + j <-> frame l0-1+j      hT [256][36]  col j <-> frame l0-1+j
//   gT    [256][32]        col j <-> frame l0+j
template <int NTU_>
struct RbTile {
    static_assert(NTU_ == 64 || NTU_ == 30 || NTU_ == 32, "tile widths: 64 (default), 30, 32");
    static constexpr int NTU = NTU_;
    static constexpr int NB = NTU_ == 64 ? 2 : 1;                        // n-blocks of GEMM 2 / 3
    static constexpr int NB1 = NTU_ == 64 ? 3 : (NTU_ == 32 ? 2 : 1);    // n-blocks of GEMM 1
    static constexpr int COLS1 = NTU_ + 2;                               // h columns produced
    static constexpr int RSC = NTU_ == 64 ? RB_RSC : 40;
    static constexpr int RSH = NTU_ == 64 ? RB_RSH : 36;
    static constexpr int RSG = NTU_ == 64 ? RB_RSG : 32;
    static constexpr int COND_FLOATS = RB_C * RSC + 32;
    static constexpr int LDS_FLOATS = COND_FLOATS + RB_C * RSH;
};

struct ResArgs {
    const float *cond;  // [B, 256, L]
    const float *x_in;  // [B, 256, L]
    float *x_out;       // [B, 256, L]
    float *skip;        // [B, 256, L] running sum
    const float *wc;    // packed PLAIN [256 rows, K=256]
    const float *w3;    // packed GATE  [512 rows, K=256 x 3 taps]
    const float *wo;    // packed PLAIN [512 rows, K=256]
    const float *bc, *b3, *bo;
    const float *hvec;  // [B, 256]  Wd s (+ Wp spk): enters h only
    const float *dvec;  // [B, 256]  Wd s: enters the residual
    float *h_save, *sig_save, *tnh_save, *g_save;  // [B, 256, L] each, SAVE only
    int L, tiles_per_b, first;
};

// k-loop of one GEMM phase.  Rows of the B tile are channels; `bcol` is this lane's first column.
// Software pipeline in registers with static indices (k-groups of 8 channels, unrolled by 4):
// the weight float4s run 2 k-groups ahead (ring of 4), the B fragments one k-group ahead (double
// buffer); each k-group's prefetches are issued at its top and pinned there with sched_barrier (left
// alone, the scheduler sinks loads next to their use and the MFMAs wait on the round trip).
#define RB_DIST 2   // weight prefetch distance in k-groups

// The first RB_DIST weight k-groups of a GEMM phase, loaded early: issued before the previous phase's epilogue and
// barrier (or, for GEMM 1, before the cond tile is staged) so that their L2 round trip is not exposed at the top
// of the k-loop three times per layer.
template <int NMB>
__device__ __forceinline__ void rb_preload(f32x4 (&pre)[RB_DIST][NMB], const f32x4 *ap0, int qstride_mb)
{
#pragma unroll
    for (int s = 0; s < RB_DIST; ++s)
#pragma unroll
        for (int i = 0; i < NMB; ++i) pre[s][i] = ap0[(size_t)i * qstride_mb + (size_t)s * 64];
}

template <int NMB, int NNB, int KW, int RS, int NCH = 8>
__device__ __forceinline__ void rb_mfma_loop(f32x16 (&acc)[NMB][NNB], const f32x4 *ap0, int qstride_mb,
                                             const float *__restrict__ tile, int bcol,
                                             const f32x4 (&pre)[RB_DIST][NMB])
{
    constexpr int TC = NCH * KW;     // (chunk, tap) pairs; 4 k-groups each
    constexpr int Q = TC * 4;
    constexpr int DIST = RB_DIST;
    const f32x4 *ap[NMB];
#pragma unroll
    for (int i = 0; i < NMB; ++i) ap[i] = ap0 + (size_t)i * qstride_mb;
    f32x4 ring[4][NMB];
    float bb[2][4][NNB];
    const float *T0 = tile + bcol;
#pragma unroll
    for (int s = 0; s < DIST; ++s)
#pragma unroll
        for (int i = 0; i < NMB; ++i) ring[s][i] = pre[s][i];
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < NNB; ++j) bb[0][e][j] = T0[(2 * e) * RS + 32 * j];
    for (int tc = 0; tc < TC; ++tc) {
        const int tcn = tc + 1 < TC ? tc + 1 : TC - 1;
        const int off_cur = (tc / KW) * (32 * RS) + (tc % KW);
        const int off_nxt = (tcn / KW) * (32 * RS) + (tcn % KW);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int q = tc * 4 + u;
            const int qa = q + DIST < Q ? q + DIST : Q - 1;
#ifdef RB_ABLATE_A  // diagnostic builds only (tools/ubench): drop the weight stream
#pragma unroll
            for (int i = 0; i < NMB; ++i) ring[(u + DIST) & 3][i] = ring[u][i];
            (void)qa;
#else
#pragma unroll
            for (int i = 0; i < NMB; ++i) ring[(u + DIST) & 3][i] = ap[i][(size_t)qa * 64];
#endif
            {
                // next k-group: g = u+1 in the same (chunk, tap), or g = 0 of the next one
                const float *Tn = T0 + (u < 3 ? off_cur + ((u + 1) * 8) * RS : off_nxt);
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int j = 0; j < NNB; ++j) {
#ifdef RB_ABLATE_B  // diagnostic builds only: drop the LDS operand reads
                        bb[(u + 1) & 1][e][j] = ring[u][0][(e + j) & 3];
                        (void)Tn;
#else
                        bb[(u + 1) & 1][e][j] = Tn[(2 * e) * RS + 32 * j];
#endif
                    }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NMB; ++i)
#pragma unroll
                    for (int j = 0; j < NNB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][i][e], bb[u & 1][e][j], acc[i][j], 0, 0, 0);
#ifdef RB_INTERLEAVE  // measured slower for the fp32 MFMA (170.8 vs 163.6 us); kept for the record
            // issue order inside the k-group: [1 global load][DS reads][MFMAs] repeated
            constexpr int NM = 4 * NMB * NNB, ND = 4 * NNB;
#pragma unroll
            for (int g = 0; g < NMB; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);              // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x100, ND / NMB, 0);       // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, NM / NMB, 0);       // MFMA
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// same loop, first weight k-groups loaded at the call (head / tail kernels: one short GEMM each)
template <int NMB, int NNB, int KW, int RS, int NCH = 8>
__device__ __forceinline__ void rb_mfma_loop(f32x16 (&acc)[NMB][NNB], const f32x4 *ap0, int qstride_mb,
                                             const float *__restrict__ tile, int bcol)
{
    f32x4 pre[RB_DIST][NMB];
    rb_preload<NMB>(pre, ap0, qstride_mb);
    rb_mfma_loop<NMB, NNB, KW, RS, NCH>(acc, ap0, qstride_mb, tile, bcol, pre);
}

template <bool VEC4, bool SAVE, int NTILE = 64>
__global__ __launch_bounds__(512, 2) void resblock_fused_kernel(ResArgs a)
{
    using TL = RbTile<NTILE>;
    static_assert(NTILE == 64 || !VEC4, "narrow tiles use the scalar staging path");
    constexpr int NB = TL::NB, NB1 = TL::NB1, COLS1 = TL::COLS1, NTU = TL::NTU, RSC = TL::RSC, RSH = TL::RSH,
                  RSG = TL::RSG;
    __shared__ __attribute__((aligned(16))) float lds[TL::LDS_FLOATS];
    float *condT = lds;
    float *hT = lds + TL::COND_FLOATS;
    float *gT = lds;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int hh = lane >> 5, c32 = lane & 31;
    const int b = blockIdx.x / a.tiles_per_b;
    const int l0 = (blockIdx.x - b * a.tiles_per_b) * NTU;
    const int L = a.L;
    const size_t bbase = (size_t)b * RB_C * L;

    const f32x4 *ap1 = reinterpret_cast<const f32x4 *>(a.wc) + (size_t)w * 32 * 64 + lane;
    const f32x4 *ap2 = reinterpret_cast<const f32x4 *>(a.w3) + (size_t)(2 * w) * 96 * 64 + lane;
    const f32x4 *ap3 = reinterpret_cast<const f32x4 *>(a.wo) + (size_t)(2 * w) * 32 * 64 + lane;
    f32x4 pre1[RB_DIST][1];
    rb_preload<1>(pre1, ap1, 0);

    // ---------------------------------------------------------------- stage cond tile -> LDS
    {
        const float *cb = a.cond + bbase;
        if (VEC4) {
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int idx = tid + k * 512;  // 256 rows x 18 float4
                const int row = idx / 18, c4 = idx - row * 18;
                const int f0 = l0 - 4 + 4 * c4;
                // unconditional load from a clamped address + select: a branch around each load would
                // serialise them behind vmcnt(0) (cdna_hip_programming.md, "three .s-level traps" (c))
                const bool ok = f0 >= 0 && f0 < L;
                const int fc = min(max(f0, 0), L - 4);
                f32x4 v = *reinterpret_cast<const f32x4 *>(cb + (size_t)row * L + fc);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                *reinterpret_cast<f32x4 *>(condT + row * RSC + 4 * c4) = ok ? v : z;
            }
        } else {
#pragma unroll
            for (int k = 0; k < COLS1 / 2; ++k) {  // 256 rows x COLS1 frames = (COLS1 / 2) x 512
                const int idx = tid + k * 512;
                const int row = idx / COLS1, cc = idx - row * COLS1;
                const int f = l0 - 1 + cc;
                const float v = cb[(size_t)row * L + min(max(f, 0), L - 1)];
                condT[row * RSC + 3 + cc] = (f >= 0 && f < L) ? v : 0.f;
            }
        }
    }

    // ---------------------------------------------------------------- GEMM 1: h (66 columns)
    f32x16 acc1[1][NB1];
    {
        const float *xb = a.x_in + bbase;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = w * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
            const float add = a.bc[row] + a.hvec[(size_t)b * RB_C + row];
#pragma unroll
            for (int j = 0; j < NB1; ++j) {
                const int f = l0 - 1 + 32 * j + c32;
                const bool ok = f >= 0 && f < L && 32 * j + c32 < COLS1;
                const float v = xb[(size_t)row * L + min(max(f, 0), L - 1)];
                acc1[0][j][r] = ok ? v + add : 0.f;
            }
        }
    }
    __syncthreads();
    rb_mfma_loop<1, NB1, 1, RSC>(acc1, ap1, 0, condT + hh * RSC, 3 + c32, pre1);
    f32x4 pre2[RB_DIST][2];
    rb_preload<2>(pre2, ap2, 96 * 64);   // lands behind the h write-back and the barrier below
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = w * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
#pragma unroll
        for (int j = 0; j < NB1; ++j) {
            const int col = 32 * j + c32;
            if (col < COLS1) {
                const int f = l0 - 1 + col;
                const bool ok = f >= 0 && f < L;  // zero padding of the k=3 conv applies to h
                const float v = ok ? acc1[0][j][r] : 0.f;
                hT[row * RSH + col] = v;
                if (SAVE && ok && col >= 1 && col <= NTU) a.h_save[bbase + (size_t)row * L + f] = v;
            }
        }
    }
    if (NTU == 30) {   // columns 32, 33 of hT are read (taps 1, 2 of the two scratch output columns): keep them finite
        if (tid < RB_C) hT[tid * RSH + 32] = hT[tid * RSH + 33] = 0.f;
    }

    __syncthreads();  // hT complete; condT free

    // ---------------------------------------------------------------- GEMM 2: z = W3 (*) h, gate
    f32x16 acc2[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    rb_mfma_loop<2, NB, 3, RSH>(acc2, ap2, 96 * 64, hT + hh * RSH, c32, pre2);
    f32x4 pre3[RB_DIST][2];
    rb_preload<2>(pre3, ap3, 32 * 64);   // lands behind the gate epilogue, the addend loads and the barrier
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = w * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
        const float bg = a.b3[ch], bf = a.b3[RB_C + ch];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const float s = mg_sigmoid(acc2[0][j][r] + bg);
            const float t = mg_tanh(acc2[1][j][r] + bf);
            const int col = 32 * j + c32;
            gT[ch * RSG + col] = s * t;
            if (SAVE) {
                const int f = l0 + col;
                if (f < L && col < NTU) {
                    const size_t o = bbase + (size_t)ch * L + f;
                    a.sig_save[o] = s;
                    a.tnh_save[o] = t;
                    a.g_save[o] = s * t;
                }
            }
        }
    }
    // ---------------------------------------------------------------- GEMM 3's addends: loaded here (acc2 is dead), consumed after GEMM 3
    // wave w owns rows 64w..64w+63 of o: w < 4 -> x rows, w >= 4 -> skip rows
    f32x16 add3[2][NB];
    {
        const bool xrows = w < 4;
        const float *src = xrows ? a.x_in + bbase : a.skip + bbase;
        const float *vec = a.dvec + (size_t)b * RB_C;
        const float use_src = (xrows || !a.first) ? 1.f : 0.f;  // layer 0 starts the skip sum
        const float use_vec = xrows ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = w * 64 + i * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
                const int ch = row & (RB_C - 1);
                const float add = a.bo[row] + use_vec * vec[ch];
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int f = min(l0 + 32 * j + c32, L - 1);  // frames >= L are never stored
                    add3[i][j][r] = add + use_src * src[(size_t)ch * L + f];
                }
            }
        }
    }
    __syncthreads();  // gT complete

    // ---------------------------------------------------------------- GEMM 3: o = Wo g, residual / skip
    f32x16 acc3[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.f;
    rb_mfma_loop<2, NB, 1, RSG>(acc3, ap3, 32 * 64, gT + hh * RSG, c32, pre3);
    {
        const bool xrows = w < 4;
        float *dst = xrows ? a.x_out + bbase : a.skip + bbase;
        const float sc = xrows ? 0.70710678118654752440f : 1.0f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = (w * 64 + i * 32 + 8 * (r >> 2) + 4 * hh + (r & 3)) & (RB_C - 1);
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const int f = l0 + 32 * j + c32;
                    if (f < L && 32 * j + c32 < NTU) dst[(size_t)ch * L + f] = (acc3[i][j][r] + add3[i][j][r]) * sc;
                }
            }
        }
    }
}
