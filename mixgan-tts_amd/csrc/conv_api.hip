// Weight packing + the generic mg_conv1d_fwd entry point.
#include "conv_mfma.h"
#include "wgrad_mfma.h"

// ---------------------------------------------------------------------------------------------
// pack: [Co, Ci, K] fp32 -> Wp[mb][q][lane][e]  (layout: conv_mfma.h header)
// ---------------------------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float *__restrict__ w, float *__restrict__ wp, PackDesc d)
{
    const size_t total = (size_t)d.MB * (d.CiP * d.K / 8) * 256;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        size_t dst;
        const float v = mg_pack_element(w, d, idx, &dst);
        wp[dst] = v;
    }
}

extern "C" size_t mg_conv_packed_floats(int Co, int Ci, int K, int mode)
{
    int Mrows, Kin, MB;
    if (pack_dims(Co, Ci, K, mode, &Mrows, &Kin, &MB) != MG_OK) return 0;
    const int CiP = mg_round_up(Kin, mg_conv_ck(K));
    return (size_t)MB * (CiP * K / 8) * 256;
}

// Packs one [Co, Ci, K] weight as k-groups [q0, q0 + Q) of a packed buffer that holds Qtot k-groups
// per 32-row block: lets several weights that share their rows be concatenated along the
// reduction axis (e.g. all 20 conditioner projections as one K = 20*256 data-gradient GEMM).
extern "C" int mg_conv_pack_at(const float *w, float *packed, int Co, int Ci, int K, int mode, int q0, int Qtot,
                               void *stream)
{
    if (!w || !packed) return MG_ERR_ARG;
    int Mrows, Kin, MB;
    MG_TRY(pack_dims(Co, Ci, K, mode, &Mrows, &Kin, &MB));
    PackDesc pd;
    size_t total;
    MG_TRY(mg_pack_desc(Co, Ci, K, mode, q0, Qtot, &pd, &total));
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, pd);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_conv_pack(const float *w, float *packed, int Co, int Ci, int K, int mode, void *stream)
{
    int Mrows, Kin, MB;
    MG_TRY(pack_dims(Co, Ci, K, mode, &Mrows, &Kin, &MB));
    const int Q = mg_round_up(Kin, mg_conv_ck(K)) * K / 8;
    return mg_conv_pack_at(w, packed, Co, Ci, K, mode, 0, Q, stream);
}

// ConvTranspose1d(Ci -> Co, kernel 2u, stride u, padding u/2) -- hifigan/models.py:121-127 -- as a polyphase
// GEMM: no zero insertion, 3 taps instead of 2u on the matrix cores.
static bool tpose_ok(int Ci, int Co, int u) { return Ci > 0 && Co > 0 && (u == 2 || u == 4 || u == 8) && (Co * u) % 4 == 0; }

extern "C" size_t mg_conv_transpose_packed_floats(int Ci, int Co, int u)
{
    if (!tpose_ok(Ci, Co, u)) return 0;
    return (size_t)mg_conv_mblocks(Co * u) * (mg_round_up(Ci, mg_conv_ck(3)) * 3 / 8) * 256;
}

extern "C" int mg_conv_transpose_pack(const float *w, float *packed, int Ci, int Co, int u, void *stream)
{
    if (!w || !packed) return MG_ERR_ARG;
    if (!tpose_ok(Ci, Co, u)) return MG_ERR_SHAPE;
    const int CK = mg_conv_ck(3), CiP = mg_round_up(Ci, CK), MB = mg_conv_mblocks(Co * u), Q = CiP * 3 / 8;
    const size_t total = (size_t)MB * Q * 256;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    const PackDesc pd{Co * u, Ci, 3, CK, CiP, MB, MG_PACK_TPOSE, 0, Q, u};
    hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, pd);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

template <>
struct EpiWide<EpiBiasAct> { static constexpr bool value = true; };

extern "C" int mg_conv_transpose1d_fwd(const float *in, const float *packed, const float *bias, float *out, int B, int Ci,
                                       int Lin, int Co, int u, float in_slope, float alpha, void *stream)
{
    if (!in || !packed || !out) return MG_ERR_ARG;
    if (B <= 0 || Lin <= 0 || !tpose_ok(Ci, Co, u)) return MG_ERR_SHAPE;
    ConvShape s{B, Ci, Lin, Lin, 3, 1, 1, Co * u, 0, 0, 1, in_slope};
    EpiBiasAct::Params ep{out, bias, nullptr, alpha, Co, MG_ACT_NONE, 0, 0, nullptr, 0.f, u};
    return conv_launch<EpiBiasAct>(s, in, nullptr, packed, ep, (hipStream_t)stream);
}

extern "C" int mg_conv1d_fwd_split(const float *in, const float *in_vec, const float *packed, const float *bias,
                                   const float *add, float *out, int B, int Ci, int Lin, int Co, int Lout, int K,
                                   int stride, int pad, int dil, float in_slope, int act, float act_slope, float alpha,
                                   int accumulate, float *scratch, size_t scratch_floats, void *stream)
{
    if (!in || !packed || !out) return MG_ERR_ARG;
    if (act < 0 || act > MG_ACT_LRELU) return MG_ERR_ARG;
    if (B <= 0 || Ci <= 0 || Co <= 0 || Lin <= 0 || Lout <= 0 || pad < 0 || dil < 1) return MG_ERR_SHAPE;
    ConvShape s{B, Ci, Lin, Lout, K, stride, pad, Co, 0, 0, dil, in_slope, scratch, scratch ? scratch_floats : 0};
    EpiBiasAct::Params ep{out, bias, add, alpha, Co, act, accumulate, 0, nullptr, act_slope};
    return conv_launch<EpiBiasAct>(s, in, in_vec, packed, ep, (hipStream_t)stream);
}

extern "C" int mg_conv1d_fwd_ex(const float *in, const float *in_vec, const float *packed, const float *bias,
                                const float *add, float *out, int B, int Ci, int Lin, int Co, int Lout, int K,
                                int stride, int pad, int dil, float in_slope, int act, float act_slope, float alpha,
                                int accumulate, void *stream)
{
    return mg_conv1d_fwd_split(in, in_vec, packed, bias, add, out, B, Ci, Lin, Co, Lout, K, stride, pad, dil, in_slope, act,
                               act_slope, alpha, accumulate, nullptr, 0, stream);
}

extern "C" int mg_conv1d_fwd(const float *in, const float *in_vec, const float *packed, const float *bias,
                             const float *add, float *out, int B, int Ci, int Lin, int Co, int Lout, int K, int stride,
                             int pad, int act, float alpha, int accumulate, void *stream)
{
    return mg_conv1d_fwd_ex(in, in_vec, packed, bias, add, out, B, Ci, Lin, Co, Lout, K, stride, pad, 1, 1.f, act, 0.f,
                            alpha, accumulate, stream);
}

// ---------------------------------------------------------------------------------------------
// weight gradient + row sums (bias gradients, per-sample channel sums)
// ---------------------------------------------------------------------------------------------
extern "C" size_t mg_conv1d_wgrad_scratch_floats(int Co, int Ci, int K)
{
    if (Co <= 0 || Ci <= 0 || K <= 0) return 0;
    return wgrad_scratch_floats(Co, Ci, K);
}

extern "C" int mg_conv1d_wgrad_strided(const float *dy, long dy_bs, const float *x, long x_bs, const float *x_vec,
                                       float *dw, float *scratch, int B, int Co, int Ci, int Ldy, int Lx, int K,
                                       int stride, int pad, float alpha, int accumulate, void *stream)
{
    if (!dy || !x || !dw || !scratch) return MG_ERR_ARG;
    if (stride < 1 || pad < 0) return MG_ERR_SHAPE;
    WgradShape s{B, Co, Ci, Ldy, Lx, K, stride, pad, dy_bs, x_bs};
    return wgrad_launch(s, dy, x, x_vec, dw, scratch, alpha, accumulate, (hipStream_t)stream);
}

extern "C" size_t mg_conv1d_wgrad_grouped_scratch_floats(int Co, int Ci, int K, int G)
{
    if (Co <= 0 || Ci <= 0 || K <= 0 || G <= 0) return 0;
    return wgrad_scratch_floats(Co, Ci, K, G);
}

extern "C" int mg_conv1d_wgrad_grouped(const float *dy, long dy_bs, long dy_gs, const float *x, long x_bs, long x_gs,
                                       float *dw, long dw_gs, float *scratch, int G, int B, int Co, int Ci, int Ldy, int Lx,
                                       int K, int stride, int pad, float alpha, int accumulate, void *stream)
{
    if (!dy || !x || !dw || !scratch) return MG_ERR_ARG;
    if (stride < 1 || pad < 0 || G < 1) return MG_ERR_SHAPE;
    WgradShape s{B, Co, Ci, Ldy, Lx, K, stride, pad, dy_bs, x_bs, G, dy_gs, x_gs, dw_gs};
    return wgrad_launch(s, dy, x, nullptr, dw, scratch, alpha, accumulate, (hipStream_t)stream);
}

extern "C" int mg_rowsum(const float *in, long in_bs, int B, int R, int L, float *out_r, float *out_br, float alpha,
                         int accumulate, void *stream);

extern "C" int mg_conv1d_wgrad_grouped_bias(const float *dy, long dy_bs, long dy_gs, const float *x, long x_bs, long x_gs,
                                            float *dw, long dw_gs, float *db, long db_gs, float *scratch, int G, int B,
                                            int Co, int Ci, int Ldy, int Lx, int K, int stride, int pad, float alpha,
                                            int accumulate, void *stream)
{
    if (!dy || !x || !dw || !db || !scratch) return MG_ERR_ARG;
    if (stride < 1 || pad < 0 || G < 1) return MG_ERR_SHAPE;
    bool done = false;
    WgradShape s{B, Co, Ci, Ldy, Lx, K, stride, pad, dy_bs, x_bs, G, dy_gs, x_gs, dw_gs, db, db_gs, &done};
    MG_TRY(wgrad_launch(s, dy, x, nullptr, dw, scratch, alpha, accumulate, (hipStream_t)stream));
    if (!done) {   // the split kernel ran (small or unaligned shapes): one row-sum launch per group
        const long bs = dy_bs ? dy_bs : (long)Co * Ldy;
        for (int g = 0; g < G; ++g)
            MG_TRY(mg_rowsum(dy + (size_t)g * dy_gs, bs, B, Co, Ldy, db + (size_t)g * (db_gs ? db_gs : Co), nullptr, alpha,
                             accumulate, stream));
    }
    return MG_OK;
}

extern "C" int mg_conv1d_wgrad(const float *dy, const float *x, const float *x_vec, float *dw, float *scratch, int B,
                               int Co, int Ci, int Ldy, int Lx, int K, int stride, int pad, float alpha,
                               int accumulate, void *stream)
{
    return mg_conv1d_wgrad_strided(dy, 0, x, 0, x_vec, dw, scratch, B, Co, Ci, Ldy, Lx, K, stride, pad, alpha, accumulate,
                                   stream);
}

// in [B, R, L] (batch stride in_bs, 0 -> R*L).  out_r[r] (+)= alpha * sum_{b,l};  out_br[b*R + r] = alpha * sum_l.
// One workgroup per row.  Without per-batch outputs every thread first accumulates its share of all B*L samples
// (16-byte loads when the rows are 16-byte aligned) and the workgroup reduces once; the per-batch form reduces
// after every batch element (two barriers each), which is 2-3x slower and only used for the [B, R] vector grads.
__global__ __launch_bounds__(256) void rowsum_kernel(const float *__restrict__ in, long in_bs, int B, int R, int L,
                                                     float *__restrict__ out_r, float *__restrict__ out_br,
                                                     float alpha, int accumulate, int vec4)
{
    __shared__ float red[4];
    const int r = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (!out_br) {
        float v = 0.f;
        if (vec4) {
            const int L4 = L >> 2;
            for (int b = 0; b < B; ++b) {
                const f32x4 *p = reinterpret_cast<const f32x4 *>(in + (size_t)b * in_bs + (size_t)r * L);
                for (int l = threadIdx.x; l < L4; l += 256) {
                    const f32x4 q = p[l];
                    v += (q[0] + q[1]) + (q[2] + q[3]);
                }
            }
        } else {
            for (int b = 0; b < B; ++b) {
                const float *p = in + (size_t)b * in_bs + (size_t)r * L;
                for (int l = threadIdx.x; l < L; l += 256) v += p[l];
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) red[wave] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float tot = alpha * (red[0] + red[1] + red[2] + red[3]);
            out_r[r] = accumulate ? out_r[r] + tot : tot;
        }
        return;
    }
    float tot = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *p = in + (size_t)b * in_bs + (size_t)r * L;
        float v = 0.f;
        for (int l = threadIdx.x; l < L; l += 256) v += p[l];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if (lane == 0) red[wave] = v;
        __syncthreads();
        const float s = red[0] + red[1] + red[2] + red[3];
        if (threadIdx.x == 0) out_br[(size_t)b * R + r] = alpha * s;
        tot += s;
    }
    if (out_r && threadIdx.x == 0) out_r[r] = accumulate ? out_r[r] + alpha * tot : alpha * tot;
}

extern "C" int mg_rowsum(const float *in, long in_bs, int B, int R, int L, float *out_r, float *out_br, float alpha,
                         int accumulate, void *stream)
{
    if (!in || (!out_r && !out_br)) return MG_ERR_ARG;
    if (B <= 0 || R <= 0 || L <= 0) return MG_ERR_SHAPE;
    const long bs = in_bs ? in_bs : (long)R * L;
    const int vec4 = (L % 4 == 0) && (bs % 4 == 0) && (((uintptr_t)in & 15) == 0);
    hipLaunchKernelGGL(rowsum_kernel, dim3(R), dim3(256), 0, (hipStream_t)stream, in, bs, B, R, L, out_r, out_br, alpha,
                       accumulate, vec4);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
