// The fused residual layer of resblock_fused.h with its three GEMMs on the bf16 MFMA as
// error-compensated ("split-fp32") products:
//
//     a * b  ~=  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi ,    x_hi = bf16(x),  x_lo = bf16(x - x_hi)
//
// Each operand carries 16 mantissa bits across its hi/lo pair, products are exact in the fp32
// accumulator, and only the lo*lo term (2^-16 relative) is dropped: fp32-grade results at three
// v_mfma_f32_32x32x16_bf16 per 16-deep k-step instead of eight v_mfma_f32_32x32x2_f32 -- 96 vs
// 512 matrix-pipe cycles.  Everything outside the products (accumulation, bias/residual/skip
// arithmetic, gate transcendentals, x / skip / output tensors) stays fp32.
//
// What changes relative to the fp32 kernel:
//   * weights are packed as [mb][kg][lane]{8 x bf16 hi, 8 x bf16 lo} (32 B per lane per 16-deep
//     k-group: the same bytes as fp32), k order = tap-major, channel-minor;
//   * the conditioner is pre-split once per call into frame-major bf16 planes
//     condS [B][L][2][256] so a tile stages with straight 16-byte copies;
//   * LDS tiles are frame-major [frame][channel] bf16 hi/lo planes with a 528-byte row pitch
//     (ds_read_b128 B fragments: 8 consecutive channels of one frame; 16 consecutive rows tile the 64
//     banks exactly once), written by the GEMM epilogues as 8-byte packed quads.
#pragma once
#include "common.h"
#include "resblock_fused.h"  // ResArgs, RB_C, RB_NT

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

#define RS_ROW 264                       // bf16 elements per LDS row (256 channels + 8 pad = 528 B)
#define RS_HROWS 66                      // h tile: frames l0-1 .. l0+64
#define RS_PLANE (RS_HROWS * RS_ROW)     // elements per plane
#define RS_TILE (2 * RS_PLANE)           // hi + lo planes
#define RS_LDS_ELEMS (2 * RS_TILE)       // cond/g tile + h tile
#ifndef RS_DIST12
#define RS_DIST12 2                      // weight prefetch distance (k-groups) in GEMM 1 / 2
#endif
#ifndef RS_DIST3
#define RS_DIST3 1                       // ... in GEMM 3 (its 64 addend registers are live)
#endif

struct ResSplitArgs {
    const __bf16 *condS;  // [B][L][2][256] frame-major hi/lo planes of the conditioner
    const float *x_in;    // [B, 256, L]
    float *x_out;
    float *skip;
    const __bf16 *wc;     // split-packed [8 mb][16 kg][64 lanes][16]
    const __bf16 *w3;     // split-packed GATE rows [16 mb][48 kg][64][16]
    const __bf16 *wo;     // split-packed [16 mb][16 kg][64][16]
    const float *bc, *b3, *bo;
    const float *hvec, *dvec;
    int L, tiles_per_b, first;
};

struct SplitFrag {
    bf16x8 hi, lo;
};

// One GEMM phase.  acc[i][j] += A(mb_i) * B(n-block j) over KG 16-deep k-groups; for the k=3 conv
// (TAPS == 3) k-group kg = tap*16 + cg reads tile rows shifted by `tap`.
template <int NMB, int NNB, int TAPS, int DIST>
__device__ __forceinline__ void rs_mfma_loop(f32x16 (&acc)[NMB][NNB], const __bf16 *ap0, int mb_stride,
                                             const __bf16 *tile_hi, const int (&rows)[NNB], int hh, int rot)
{
    // `rot` rotates the order in which the k-groups are visited (a function of the tile's position in
    // its utterance, so results do not depend on batch composition): workgroups then stream
    // different parts of the shared weight blob at any instant instead of all hitting the same L2
    // lines in lockstep.
    // Software pipeline, all stages in registers with static indices (unroll by 4):
    //   weights (global/L2 -> VGPR) run DIST k-groups ahead in a ring of 4 stages,
    //   B fragments (LDS -> VGPR) run one k-group ahead in a double buffer.
    // Both prefetches are issued at the top of a k-group and pinned there with sched_barrier: left
    // alone, the scheduler sinks loads next to their use and every k-group stalls on the round trip.
    constexpr int KG = TAPS * 16;
    static_assert(DIST >= 1 && DIST <= 3, "ring of 4 stages");
#ifdef RS_ABLATE_MFMA  // diagnostic builds only: prologue/epilogue cost
    return;
#endif
    const bf16x8 *ap[NMB];
#pragma unroll
    for (int i = 0; i < NMB; ++i) ap[i] = reinterpret_cast<const bf16x8 *>(ap0 + (size_t)i * mb_stride);
    SplitFrag ring[4][NMB];
    SplitFrag bb[2][NNB];
    const __bf16 *bbase[NNB];
#pragma unroll
    for (int j = 0; j < NNB; ++j) bbase[j] = tile_hi + rows[j] * RS_ROW + 8 * hh;
#pragma unroll
    for (int s = 0; s < DIST; ++s) {
        const int k0 = (s + rot) % KG;
#pragma unroll
        for (int i = 0; i < NMB; ++i) {
            ring[s][i].hi = ap[i][(size_t)k0 * 128];  // 64 lanes x 2 vectors per k-group
            ring[s][i].lo = ap[i][(size_t)k0 * 128 + 1];
        }
    }
    {
        const int k0 = rot % KG;
        const int off = (k0 >> 4) * RS_ROW + (k0 & 15) * 16;
#pragma unroll
        for (int j = 0; j < NNB; ++j) {
            bb[0][j].hi = *reinterpret_cast<const bf16x8 *>(bbase[j] + off);
            bb[0][j].lo = *reinterpret_cast<const bf16x8 *>(bbase[j] + off + RS_PLANE);
        }
    }
    for (int kg4 = 0; kg4 < KG; kg4 += 4) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int kg = kg4 + u;
#ifdef RS_ABLATE_L1  // diagnostic: weight loads confined to an L1-resident window
            const int ka = (kg + DIST) & 3;
#else
            const int ka = ((kg + DIST < KG ? kg + DIST : KG - 1) + rot) % KG;
#endif
            const int kb = ((kg + 1 < KG ? kg + 1 : KG - 1) + rot) % KG;
#ifdef RS_ABLATE_A  // diagnostic builds only (tools/ubench): no weight stream
#pragma unroll
            for (int i = 0; i < NMB; ++i) ring[(u + DIST) & 3][i] = ring[u][i];
            (void)ka;
#else
#pragma unroll
            for (int i = 0; i < NMB; ++i) {
                ring[(u + DIST) & 3][i].hi = ap[i][(size_t)ka * 128];
                ring[(u + DIST) & 3][i].lo = ap[i][(size_t)ka * 128 + 1];
            }
#endif
            {
                const int off = (kb >> 4) * RS_ROW + (kb & 15) * 16;  // tap shift = one row; 16 channels per k-group
#pragma unroll
                for (int j = 0; j < NNB; ++j) {
#ifdef RS_ABLATE_B  // diagnostic builds only: no LDS operand reads
                    bb[(u + 1) & 1][j].hi = ring[u][0].hi;
                    bb[(u + 1) & 1][j].lo = ring[(u + 1 + j) & 3][0].lo;
                    (void)off;
#else
                    bb[(u + 1) & 1][j].hi = *reinterpret_cast<const bf16x8 *>(bbase[j] + off);
                    bb[(u + 1) & 1][j].lo = *reinterpret_cast<const bf16x8 *>(bbase[j] + off + RS_PLANE);
#endif
                }
            }
#ifdef RS_SCHED_BURST
            __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
            for (int i = 0; i < NMB; ++i)
#pragma unroll
                for (int j = 0; j < NNB; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[u][i].hi, bb[u & 1][j].hi, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[u][i].hi, bb[u & 1][j].lo, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ring[u][i].lo, bb[u & 1][j].hi, acc[i][j], 0, 0, 0);
                }
#ifndef RS_SCHED_BURST
            // interleave this k-group's prefetches with its MFMAs: one global load + one LDS read per three
            // MFMAs, instead of a burst of loads that blocks the wave at issue while the queues are full
#pragma unroll
            for (int g = 0; g < 2 * NMB; ++g) {
                __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // VMEM read
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
                __builtin_amdgcn_sched_group_barrier(0x008, (3 * NMB * NNB) / (2 * NMB), 0);  // MFMA
            }
#endif
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// write 4 consecutive channels (fp32) of one frame row as bf16 hi / lo quads
__device__ __forceinline__ void rs_store_quad(__bf16 *tile_hi, int row, int ch, float v0, float v1, float v2, float v3)
{
    bf16x4 h, l;
    h[0] = (__bf16)v0; h[1] = (__bf16)v1; h[2] = (__bf16)v2; h[3] = (__bf16)v3;
    l[0] = (__bf16)(v0 - (float)h[0]);
    l[1] = (__bf16)(v1 - (float)h[1]);
    l[2] = (__bf16)(v2 - (float)h[2]);
    l[3] = (__bf16)(v3 - (float)h[3]);
    __bf16 *p = tile_hi + row * RS_ROW + ch;
    *reinterpret_cast<bf16x4 *>(p) = h;
    *reinterpret_cast<bf16x4 *>(p + RS_PLANE) = l;
}

__global__ __launch_bounds__(512, 2) void resblock_split_kernel(ResSplitArgs a)
{
    __shared__ __attribute__((aligned(16))) __bf16 lds[RS_LDS_ELEMS];
    __bf16 *condT = lds;            // [2][66][264]; g tile aliases it after GEMM 1
    __bf16 *hT = lds + RS_TILE;     // [2][66][264]
    __bf16 *gT = lds;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int w = tid >> 6;
    const int hh = lane >> 5, c32 = lane & 31;
    const int b = blockIdx.x / a.tiles_per_b;
    const int l0 = (blockIdx.x - b * a.tiles_per_b) * RB_NT;
    const int L = a.L;
    const size_t bbase = (size_t)b * RB_C * L;
#ifdef RS_NO_ROT
    const int rot = 0;
#else
    const int rot = (blockIdx.x - b * a.tiles_per_b) & 15;  // position of the tile in its utterance
#endif

    // ---------------------------------------------------------------- stage cond tile (66 frames x 2 planes x 512 B)
    {
        const __bf16 *cb = a.condS + (size_t)b * L * 512;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            const int idx = tid + k * 512;            // 66 rows x 2 planes x 32 vectors of 16 B = 4224
            const int row = idx >> 6, rem = idx & 63;
            const int plane = rem >> 5, v = rem & 31;
            const int f = l0 - 1 + row;
            const bool ok = idx < RS_HROWS * 64 && f >= 0 && f < L;
            const int fc = min(max(f, 0), L - 1);
            bf16x8 val = *reinterpret_cast<const bf16x8 *>(cb + ((size_t)fc * 2 + plane) * 256 + v * 8);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) val[j] = (__bf16)0.f;
            }
            if (idx < RS_HROWS * 64)
                *reinterpret_cast<bf16x8 *>(condT + plane * RS_PLANE + row * RS_ROW + v * 8) = val;
        }
    }

    // ---------------------------------------------------------------- GEMM 1: h on 66 frames
    f32x16 acc1[1][3];
    {
        const float *xb = a.x_in + bbase;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = w * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
            const float add = a.bc[row] + a.hvec[(size_t)b * RB_C + row];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int f = l0 - 1 + 32 * j + c32;
                const bool ok = f >= 0 && f < L && (j < 2 || c32 < 2);
                const float v = xb[(size_t)row * L + min(max(f, 0), L - 1)];
                acc1[0][j][r] = ok ? v + add : 0.f;
            }
        }
    }
    __syncthreads();
    {
        const int rows1[3] = {c32, 32 + c32, min(64 + c32, RS_HROWS - 1)};  // third block: 2 useful columns
        rs_mfma_loop<1, 3, 1, RS_DIST12>(acc1, a.wc + ((size_t)w * 16 * 64 + lane) * 16, 0, condT, rows1, hh, rot);
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (j < 2 || c32 < 2) {
            const int col = 32 * j + c32;
            const int f = l0 - 1 + col;
            const bool ok = f >= 0 && f < L;  // the k=3 conv zero-pads h, not x
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int ch = w * 32 + 8 * q + 4 * hh;
                rs_store_quad(hT, col, ch, ok ? acc1[0][j][4 * q + 0] : 0.f, ok ? acc1[0][j][4 * q + 1] : 0.f,
                              ok ? acc1[0][j][4 * q + 2] : 0.f, ok ? acc1[0][j][4 * q + 3] : 0.f);
            }
        }
    }
    __syncthreads();  // hT complete; condT free

    // ---------------------------------------------------------------- GEMM 2: z = W3 (*) h, gate
    f32x16 acc2[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.f;
    {
        const int rows2[2] = {c32, 32 + c32};
        rs_mfma_loop<2, 2, 3, RS_DIST12>(acc2, a.w3 + ((size_t)(2 * w) * 48 * 64 + lane) * 16, 48 * 64 * 16, hT, rows2, hh, 3 * rot);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int col = 32 * j + c32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ch = w * 32 + 8 * q + 4 * hh;
            float g4[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float s = mg_sigmoid(acc2[0][j][4 * q + e] + a.b3[ch + e]);
                const float t = mg_tanh(acc2[1][j][4 * q + e] + a.b3[RB_C + ch + e]);
                g4[e] = s * t;
            }
            rs_store_quad(gT, col, ch, g4[0], g4[1], g4[2], g4[3]);
        }
    }
    // ---------------------------------------------------------------- GEMM 3's addends
    f32x16 add3[2][2];
    {
        const bool xrows = w < 4;
        const float *src = xrows ? a.x_in + bbase : a.skip + bbase;
        const float *vec = a.dvec + (size_t)b * RB_C;
        const float use_src = (xrows || !a.first) ? 1.f : 0.f;
        const float use_vec = xrows ? 1.f : 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = w * 64 + i * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
                const int ch = row & (RB_C - 1);
                const float add = a.bo[row] + use_vec * vec[ch];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int f = min(l0 + 32 * j + c32, L - 1);
                    add3[i][j][r] = add + use_src * src[(size_t)ch * L + f];
                }
            }
        }
    }
    __syncthreads();  // gT complete

    // ---------------------------------------------------------------- GEMM 3: o = Wo g
    f32x16 acc3[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.f;
    {
        const int rows3[2] = {c32, 32 + c32};
        rs_mfma_loop<2, 2, 1, RS_DIST3>(acc3, a.wo + ((size_t)(2 * w) * 16 * 64 + lane) * 16, 16 * 64 * 16, gT, rows3, hh, rot);
    }
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
                for (int j = 0; j < 2; ++j) {
                    const int f = l0 + 32 * j + c32;
                    if (f < L) dst[(size_t)ch * L + f] = (acc3[i][j][r] + add3[i][j][r]) * sc;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// split packing:  w [Co][Ci][K] fp32 -> [mb][kg][lane]{hi[8], lo[8]},  k = tap*CiP + ci  (CiP multiple of 16)
// gate != 0: rows interleaved as in MG_PACK_GATE.
// ---------------------------------------------------------------------------------------------
__global__ void pack_split_kernel(const float *__restrict__ w, __bf16 *__restrict__ wp, int Co, int Ci, int K, int MB,
                                  int gate)
{
    const int KG = K * (Ci / 16);
    const size_t total = (size_t)MB * KG * 64 * 8;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx & 7);
        const int lane = (int)((idx >> 3) & 63);
        const size_t g = idx >> 9;
        const int kg = (int)(g % KG);
        const int mb = (int)(g / KG);
        const int tap = kg / (Ci / 16);
        const int ci = (kg - tap * (Ci / 16)) * 16 + 8 * (lane >> 5) + j;
        const int r = lane & 31;
        int row;
        bool ok;
        if (gate) {
            const int half = mb & 1, rr = (mb >> 1) * 32 + r;
            row = half * (Co / 2) + rr;
            ok = rr < Co / 2;
        } else {
            row = mb * 32 + r;
            ok = row < Co;
        }
        const float v = ok ? w[((size_t)row * Ci + ci) * K + tap] : 0.f;
        const __bf16 h = (__bf16)v;
        const size_t o = (g * 64 + lane) * 16 + j;
        wp[o] = h;
        wp[o + 8] = (__bf16)(v - (float)h);
    }
}

// cond [B][256][L] fp32 -> condS [B][L][2][256] bf16 hi/lo (LDS-tiled transpose, 64 frames per block;
// reads are 256-byte row segments, writes are 16-byte vectors of 8 channels)
__global__ __launch_bounds__(256) void cond_split_kernel(const float *__restrict__ cond, __bf16 *__restrict__ out, int L)
{
    __shared__ float tile[64][257];
    const int b = blockIdx.y, l0 = blockIdx.x * 64;
    const int nl = min(64, L - l0);
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
        const int idx = threadIdx.x + k * 256;
        const int c = idx >> 6, l = idx & 63;
        const float v = cond[((size_t)b * 256 + c) * L + min(l0 + l, L - 1)];
        tile[l][c] = v;
    }
    __syncthreads();
    // 64 frames x 32 channel-octets = 2048 work items
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int idx = threadIdx.x + k * 256;
        const int l = idx >> 5, c0 = (idx & 31) * 8;
        if (l >= nl) continue;
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = tile[l][c0 + j];
            hi[j] = (__bf16)v;
            lo[j] = (__bf16)(v - (float)hi[j]);
        }
        __bf16 *o = out + (((size_t)b * L + l0 + l) * 2) * 256 + c0;
        *reinterpret_cast<bf16x8 *>(o) = hi;
        *reinterpret_cast<bf16x8 *>(o + 256) = lo;
    }
}
