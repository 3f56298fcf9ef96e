// Data gradients of the 20 gated residual layers (what torch.autograd derives for model/blocks.py:1157-1176) as ONE
// launch on gfx950: the backward counterpart of denoiser_persist.h.  Per layer l (top to bottom), with
// dout = [dx_{l+1}/sqrt2 ; dskip] (dskip = gradient of the skip sum: the same for every layer):
//   dg  = Wo^T dout                                  GEMM A   M = 256  K = 512
//   dz  = [dg * tanh * sig (1 - sig) ; dg * sig (1 - tanh^2)]         (saved sigmoid / tanh of the forward)
//   dh  = W3^T (*) dz   (k = 3, taps flipped)        GEMM B   M = 256  K = 1536
//   dx_l / sqrt2 = (dh + dx_{l+1}/sqrt2) / sqrt2     -> next layer's dout
// The launch-per-layer form ran these as two generic conv launches per layer (39 + 77 us at B=8, L=1000: 64x64 output
// tiles that re-stage the 512-channel slab once per 64 output rows, 1.25 memory instructions per MFMA).  Here a
// workgroup (8 waves, wave w owns channels 32w..32w+31) keeps 32 frames of one utterance for all layers: dskip is
// staged once and stays in LDS, dx stays in registers (and in LDS as GEMM A's operand), dz never leaves LDS between
// the two GEMMs, the k=3 halo of dz is handed between neighbouring workgroups (tagged granules, as in the forward),
// and the weights stream once per workgroup.  HBM per layer: sigmoid / tanh in; dz, dh, dx out -- they feed the
// grouped weight-gradient GEMMs that follow (mg_denoiser_bwd).
// LDS (132 KB, one workgroup per CU), both tiles k-interleaved (dp_at):
//   douT 512 channels x 32 columns   rows 0..255 dx/sqrt2 (rewritten every layer), rows 256..511 dskip (persistent)
//   dzT  512 channels x 34 columns   col j <-> frame l0-1+j
#pragma once
#include "denoiser_persist.h"

struct BwdPersistArgs {
    float *dout;             // [B, 2C, L]: rows >= C hold dskip on entry; rows < C receive dx_0 / sqrt2 on exit
    const float *sig, *tnh;  // [NL][B, C, L] saved by the forward (layer stride act_stride)
    size_t act_stride;
    const float *blayers;    // first layer's backward record: DGRAD packs
    size_t blayer_stride, bl_woT, bl_w3T;
    float *dz_all;           // [B][NL*2C][L]
    float *dx_all;           // [B][(NL+1)*C][L]: slot l receives the dx layer l produces (/sqrt2)
    float *dh_all;           // [B][NL*C][L]
    dp_u64 *gran;            // [2 parity][tiles][2 sides][512]
    unsigned *sync;          // as in the forward: [0] ticket, [1] error (sticky), [2] launches completed, [3] done
    unsigned *host_err;      // pinned host word (or NULL)
    unsigned spin_limit;
    int B, L, NL, tiles_per_b;
};

template <int NCH>
struct DpIterK1N {   // 1x1 over NCH chunks of 32 channels
    static constexpr int N = NCH, KW = 1;
    static __device__ __forceinline__ int chunk(int it) { return it; }
    static __device__ __forceinline__ int tap(int) { return 0; }
};
struct DpIterCentre16 {
    static constexpr int N = 16, KW = 3;
    static __device__ __forceinline__ int chunk(int it) { return it; }
    static __device__ __forceinline__ int tap(int) { return 1; }
};
struct DpIterOuter16 {
    static constexpr int N = 32, KW = 3;
    static __device__ __forceinline__ int chunk(int it) { return it >> 1; }
    static __device__ __forceinline__ int tap(int it) { return (it & 1) * 2; }
};

template <bool VEC4>
__global__ __launch_bounds__(512, 2) void denoiser_bwd_persist_kernel(BwdPersistArgs a)
{
    constexpr int NT = 32, ND = NT, NZ = NT + 2;
    __shared__ __attribute__((aligned(16))) float lds[2 * RB_C * (ND + NZ)];
    __shared__ unsigned s_tile, s_dead, s_launch;
    float *douT = lds;                       // 512 channels x ND columns
    float *dzT = lds + 2 * RB_C * ND;        // 512 channels x NZ columns

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hh = lane >> 5, c32 = lane & 31;
    const int L = a.L, NL = a.NL;
    const int n_tiles = a.tiles_per_b * a.B;
    if (tid == 0) {
        s_tile = __hip_atomic_fetch_add(a.sync, 1u, DP_RLX_AGENT);   // tickets in START order
        s_launch = __hip_atomic_load(a.sync + 2, DP_RLX_AGENT);     // advanced only after every workgroup has exited
        s_dead = 0u;
    }
    __syncthreads();
    const int tile = (int)(s_tile % (unsigned)n_tiles);
    const unsigned launch_no = s_launch;
    const int b = tile / a.tiles_per_b, jt = tile - b * a.tiles_per_b;
    const int l0 = jt * NT;
    const bool has_left = jt > 0, has_right = jt + 1 < a.tiles_per_b;
    const int f = l0 + c32;
    const bool fvalid = f < L;
    const int fc = min(f, L - 1);
    const int ch0 = 32 * w;   // this wave's 32 channels
    auto row_of = [&](int r) { return ch0 + 8 * (r >> 2) + 4 * hh + (r & 3); };
    const size_t CL = (size_t)RB_C * L;

    // ---------------------------------------------------------------- stage dskip (rows 256..511 of douT), zero dx
    {
        const float *db = a.dout + (size_t)b * 2 * CL + CL;
        if (VEC4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // 256 rows x 8 float4
                const int idx = tid + k * 512;
                const int row = idx >> 3, c4 = idx & 7;
                const int f0 = l0 + 4 * c4;
                const bool ok = f0 < L;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(db + (size_t)row * L + min(f0, L - 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) douT[dp_at<ND>(RB_C + row, 4 * c4 + e)] = ok ? v[e] : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {   // 256 rows x 32 frames
                const int idx = tid + k * 512;
                const int row = idx >> 5, cc = idx & 31;
                const float v = db[(size_t)row * L + min(l0 + cc, L - 1)];
                douT[dp_at<ND>(RB_C + row, cc)] = l0 + cc < L ? v : 0.f;
            }
        }
    }
    f32x16 dxs;   // dx_{l+1} / sqrt2 of this wave's 32 channels (0 above the last layer)
#pragma unroll
    for (int r = 0; r < 16; ++r) dxs[r] = 0.f;
    dp_store_block<ND>(douT, ch0, c32, hh, [&](int) { return 0.f; });

    dp_gu64 *const gran = (dp_gu64 *)a.gran;
    const size_t dz_bs = (size_t)NL * 2 * CL, dx_bs = (size_t)(NL + 1) * CL, dh_bs = (size_t)NL * CL;

    for (int l = NL - 1; l >= 0; --l) {
        const float *bp = a.blayers + (size_t)l * a.blayer_stride;
        const unsigned epoch = launch_no * ((unsigned)NL + 1u) + (unsigned)(NL - 1 - l) + 1u;   // never repeats on a workspace
        const int par = l & 1;
        // saved sigmoid / tanh of this wave's channels: loaded now, used after GEMM A
        float sg[16], th[16];
        {
            const float *sp = a.sig + (size_t)l * a.act_stride + (size_t)b * CL, *tp = a.tnh + (size_t)l * a.act_stride + (size_t)b * CL;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                sg[r] = sp[(size_t)row_of(r) * L + fc];
                th[r] = tp[(size_t)row_of(r) * L + fc];
            }
        }
        __syncthreads();   // douT rows < 256 hold this layer's dx/sqrt2; every wave is past the previous layer's GEMM B

        // ------------------------------------------------------------ GEMM A: dg = Wo^T [dx/sqrt2 ; dskip]
        f32x16 dg[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) dg[0][0][r] = 0.f;
        {
            const f32x4 *const ap[1] = {reinterpret_cast<const f32x4 *>(bp + a.bl_woT) + (size_t)w * 64 * 64 + lane};
            dp_mfma_loop<1, 1, ND, DpIterK1N<16>>(dg, ap, douT + c32 * 8 + hh * 4);
        }
        // gate derivative (model/blocks.py:1170-1171) -> dz rows ch (sigmoid branch) and 256 + ch (tanh branch)
        float zg[16], zf[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float g = fvalid ? dg[0][0][r] : 0.f;   // frames >= L: zero padding of the convolution
            zg[r] = g * th[r] * sg[r] * (1.f - sg[r]);
            zf[r] = g * sg[r] * (1.f - th[r] * th[r]);
        }
        dp_store_block<NZ>(dzT, ch0, 1 + c32, hh, [&](int r) { return zg[r]; });
        dp_store_block<NZ>(dzT, RB_C + ch0, 1 + c32, hh, [&](int r) { return zf[r]; });
        if (fvalid) {
            float *zb = a.dz_all + (size_t)b * dz_bs + (size_t)l * 2 * CL;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                zb[(size_t)row_of(r) * L + f] = zg[r];
                zb[(size_t)(RB_C + row_of(r)) * L + f] = zf[r];
            }
        }
        __syncthreads();   // interior of dzT complete

        // ------------------------------------------------------------ hand the edge columns of dz to the neighbours
        if (w < 2) {
            const bool go = w == 0 ? has_left : has_right;
            if (go) {
                const int dst_tile = w == 0 ? tile - 1 : tile + 1;
                const int col = w == 0 ? 1 : NT;
                dp_gu64 *g = gran + (((size_t)par * n_tiles + dst_tile) * 2 + (w == 0 ? 1 : 0)) * (2 * RB_C);
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int row = lane + 64 * k;
                    const dp_u64 v = ((dp_u64)epoch << 32) | (dp_u64)__float_as_uint(dzT[dp_at<NZ>(row, col)]);
                    __hip_atomic_store(g + row, v, DP_RLX_AGENT);
                }
            }
        }

        // ------------------------------------------------------------ GEMM B: dh = W3^T (*) dz, centre tap first
        f32x16 dh[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[0][0][r] = 0.f;
        const f32x4 *const apb[1] = {reinterpret_cast<const f32x4 *>(bp + a.bl_w3T) + (size_t)w * 192 * 64 + lane};
        dp_mfma_loop<1, 1, NZ, DpIterCentre16>(dh, apb, dzT + c32 * 8 + hh * 4);

        if (w < 2) {   // receive the halo columns
            const bool from = w == 0 ? has_left : has_right;
            const int col = w == 0 ? 0 : NT + 1;
            unsigned v[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
            if (from && s_dead == 0u) {
                dp_gu64 *g = gran + (((size_t)par * n_tiles + tile) * 2 + w) * (2 * RB_C);
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 8; ++k) {
                        const dp_u64 x = __hip_atomic_load(g + lane + 64 * k, DP_RLX_AGENT);
                        v[k] = (unsigned)x;
                        ok &= (unsigned)(x >> 32) == epoch;
                    }
                    if (__all(ok)) break;
                    if (++spins > a.spin_limit) {
                        if (lane == 0) {
                            dp_fail(a.sync, a.host_err, 0x100u + (unsigned)l);
                            s_dead = 1u;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) dzT[dp_at<NZ>(lane + 64 * k, col)] = from ? __uint_as_float(v[k]) : 0.f;
        }
        __syncthreads();   // halo columns in place
        dp_mfma_loop<1, 1, NZ, DpIterOuter16>(dh, apb, dzT + c32 * 8 + hh * 4);

        // ------------------------------------------------------------ dx_l/sqrt2 = (dh + dx_{l+1}/sqrt2)/sqrt2
#pragma unroll
        for (int r = 0; r < 16; ++r) dxs[r] = (dh[0][0][r] + dxs[r]) * 0.70710678118654752440f;
        dp_store_block<ND>(douT, ch0, c32, hh, [&](int r) { return dxs[r]; });   // GEMM A of every wave is long done
        if (fvalid) {
            float *hb = a.dh_all + (size_t)b * dh_bs + (size_t)l * CL;
            float *xb = a.dx_all + (size_t)b * dx_bs + (size_t)l * CL;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                hb[(size_t)row_of(r) * L + f] = dh[0][0][r];
                xb[(size_t)row_of(r) * L + f] = dxs[r];
            }
        }
    }
    // a hand-off timed out in this launch: poison what the input-projection backward and d x_t are computed from
    // (and with them the gradient norm), so that the failed backward cannot pass for a result
    const bool bad = dp_failed(a.sync);
    if (fvalid) {   // what the input-projection backward reads: dout rows < C = dx_0 / sqrt2
        float *ob = a.dout + (size_t)b * 2 * CL;
#pragma unroll
        for (int r = 0; r < 16; ++r) ob[(size_t)row_of(r) * L + f] = bad ? __builtin_nanf("") : dxs[r];
    }
    if (tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(a.sync + 3, 1u, DP_RLX_AGENT);
        if (done == (unsigned)n_tiles - 1u) {
            __hip_atomic_store(a.sync + 3, 0u, DP_RLX_AGENT);
            __hip_atomic_store(a.sync, 0u, DP_RLX_AGENT);
            __hip_atomic_fetch_add(a.sync + 2, 1u, DP_RLX_AGENT);
        }
    }
}
