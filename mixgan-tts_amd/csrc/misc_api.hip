// Small entry points around the GEMM kernels: activation derivatives, zero-insertion for strided
// data gradients, the step-embedding MLP (model/blocks.py:899-913 + LinearNorm/Mish/LinearNorm as
// used by Denoiser model/modules.py:399-403 and JCUDiscriminator model/mixgantts.py:204-208) and
// bias-free per-sample linears (speaker projections), forward and backward.
#include "denoiser_common.h"

// dpre = dy * act'(pre) expressed through the saved OUTPUT y = act(pre)
__global__ void act_bwd_kernel(const float *__restrict__ dy, const float *__restrict__ y, float *__restrict__ out, int act,
                               size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float g = dy[i], v = y[i];
        float r = g;
        if (act == MG_ACT_RELU) r = v > 0.f ? g : 0.f;
        else if (act == MG_ACT_LRELU02) r = v > 0.f ? g : 0.2f * g;
        else if (act == MG_ACT_TANH) r = g * (1.f - v * v);
        out[i] = r;
    }
}

extern "C" int mg_act_bwd(const float *dy, const float *y, float *out, int act, size_t n, void *stream)
{
    if (!dy || !y || !out) return MG_ERR_ARG;
    if (act < 0 || act > MG_ACT_TANH) return MG_ERR_ARG;
    if (n == 0) return MG_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(act_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, y, out, act, n);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// out[r, j] = (j % s == 0 && j/s < Lin) ? lrelu(in[r, j/s], slope) : 0,   j < Lup   (slope 1 = identity)
__global__ void upsample_zero_kernel(const float *__restrict__ in, float *__restrict__ out, int Lin, int s, int Lup,
                                     float slope, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / Lup;
        const int j = (int)(i - r * Lup);
        const int q = j / s;
        float v = 0.f;
        if (j - q * s == 0 && q < Lin) {
            v = in[r * Lin + q];
            v = v > 0.f ? v : v * slope;
        }
        out[i] = v;
    }
}

extern "C" int mg_upsample_zero_act(const float *in, float *out, int rows, int Lin, int stride, int Lup, float slope,
                                    void *stream)
{
    if (!in || !out) return MG_ERR_ARG;
    if (rows <= 0 || Lin <= 0 || stride < 1 || Lup <= 0) return MG_ERR_SHAPE;
    const size_t n = (size_t)rows * Lup;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    hipLaunchKernelGGL(upsample_zero_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, Lin, stride, Lup,
                       slope, n);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_upsample_zero(const float *in, float *out, int rows, int Lin, int stride, int Lup, void *stream)
{
    return mg_upsample_zero_act(in, out, rows, Lin, stride, Lup, 1.f, stream);
}

// ---------------------------------------------------------------------------------------------
// step MLP: out = W2 mish(W0 emb(t))        emb [B,D0], pre/h [B,D1], out [B,D2]
// ---------------------------------------------------------------------------------------------
extern "C" int mg_step_mlp_fwd(const int64_t *t, const float *freq, const float *W0, const float *W2, float *emb,
                               float *pre, float *h, float *out, int B, int D0, int D1, int D2, void *stream)
{
    if (!t || !freq || !W0 || !W2 || !emb || !pre || !h || !out) return MG_ERR_ARG;
    if (B <= 0 || D0 <= 0 || D0 % 2 || D1 <= 0 || D2 <= 0) return MG_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(step_embed_kernel, dim3(mg_cdiv(B * (D0 / 2), 256)), dim3(256), 0, st, t, freq, emb, B, D0);
    MG_LAUNCH_CHECK();
    MG_TRY(small_linear(W0, 0, emb, h, 0, nullptr, 0, pre, B, D1, D0, 1, 1, st));
    MG_TRY(small_linear(W2, 0, h, out, 0, nullptr, 0, nullptr, B, D2, D1, 1, 0, st));
    return MG_OK;
}

static int outer(const float *a, long a_zs, long a_bs, const float *c, float *out, int Z, int B, int N, int K,
                 hipStream_t st)
{
    const size_t n = (size_t)Z * N * K;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(small_outer_kernel, dim3(blocks), dim3(256), 0, st, a, a_zs, a_bs, c, out, Z, B, N, K);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

static int linear_t(const float *W, const float *a, float *out, int B, int N, int K, hipStream_t st)
{
    hipLaunchKernelGGL(small_linear_t_kernel, dim3(mg_cdiv(K, 64), B, 1), dim3(256), 0, st, W, 0, a, 0, (long)N, out, 1, B,
                       N, K);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// g_out [B,D2] -> dW2 [D2,D1], dW0 [D1,D0]; scratch: 2*B*D1 floats
extern "C" int mg_step_mlp_bwd(const float *g_out, const float *emb, const float *pre, const float *h, const float *W2,
                               float *dW0, float *dW2, float *scratch, int B, int D0, int D1, int D2, void *stream)
{
    if (!g_out || !emb || !pre || !h || !W2 || !dW0 || !dW2 || !scratch) return MG_ERR_ARG;
    if (B <= 0 || D0 <= 0 || D1 <= 0 || D2 <= 0) return MG_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    float *dm = scratch, *da = scratch + (size_t)B * D1;
    MG_TRY(outer(g_out, 0, D2, h, dW2, 1, B, D2, D1, st));
    MG_TRY(linear_t(W2, g_out, dm, B, D2, D1, st));
    hipLaunchKernelGGL(mish_bwd_kernel, dim3(mg_cdiv(B * D1, 256)), dim3(256), 0, st, dm, pre, da, B * D1);
    MG_LAUNCH_CHECK();
    MG_TRY(outer(da, 0, D1, emb, dW0, 1, B, D1, D0, st));
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// bias-free per-sample linear: out [B,N] = x [B,K] W^T,  W [N,K]
// ---------------------------------------------------------------------------------------------
extern "C" int mg_linear_small_fwd(const float *x, const float *W, float *out, int B, int N, int K, void *stream)
{
    if (!x || !W || !out) return MG_ERR_ARG;
    if (B <= 0 || N <= 0 || K <= 0) return MG_ERR_SHAPE;
    return small_linear(W, 0, x, out, 0, nullptr, 0, nullptr, B, N, K, 1, 0, (hipStream_t)stream);
}

extern "C" int mg_linear_small_bwd(const float *g, const float *x, const float *W, float *dx, float *dW, int B, int N,
                                   int K, void *stream)
{
    if (!g || !x || !W || (!dx && !dW)) return MG_ERR_ARG;
    if (B <= 0 || N <= 0 || K <= 0) return MG_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    if (dW) MG_TRY(outer(g, 0, N, x, dW, 1, B, N, K, st));
    if (dx) MG_TRY(linear_t(W, g, dx, B, N, K, st));
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// stand-alone pieces of the denoiser's small modules (model/blocks.py:894-913, :1170-1171)
// ---------------------------------------------------------------------------------------------
// dz[b, c, l] = dg * tanh * sig (1 - sig), dz[b, C + c, l] = dg * sig (1 - tanh^2): derivative of
// g = sigmoid(z[:C]) * tanh(z[C:]) through the values the forward saved
__global__ void gate_bwd_kernel(const float *__restrict__ dg, const float *__restrict__ sig, const float *__restrict__ tnh,
                                float *__restrict__ dz, int CL, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / CL, r = i - b * CL;
        const float g = dg[i], s = sig[i], t = tnh[i];
        dz[b * 2 * CL + r] = g * t * s * (1.f - s);
        dz[b * 2 * CL + CL + r] = g * s * (1.f - t * t);
    }
}

extern "C" int mg_gate_bwd(const float *dg, const float *sig, const float *tnh, float *dz, int B, int C, int L, void *stream)
{
    if (!dg || !sig || !tnh || !dz) return MG_ERR_ARG;
    if (B <= 0 || C <= 0 || L <= 0) return MG_ERR_SHAPE;
    const size_t n = (size_t)B * C * L;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(gate_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dg, sig, tnh, dz, C * L, n);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

__global__ void mish_fwd_kernel(const float *__restrict__ x, float *__restrict__ y, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float v = x[i];
        y[i] = v * tanhf(mg_softplus(v));
    }
}

extern "C" int mg_mish_fwd(const float *x, float *y, size_t n, void *stream)
{
    if (!x || !y) return MG_ERR_ARG;
    if (n == 0) return MG_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(mish_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, n);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_mish_bwd(const float *gy, const float *x, float *gx, size_t n, void *stream)
{
    if (!gy || !x || !gx) return MG_ERR_ARG;
    if (n == 0) return MG_OK;
    if (n > 0x7fffffffu) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(mish_bwd_kernel, dim3(mg_cdiv((int)n, 256)), dim3(256), 0, (hipStream_t)stream, gy, x, gx, (int)n);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// DiffusionEmbedding.forward (model/blocks.py:906-913): emb [B, D] = [sin(t f) | cos(t f)], f [D/2] from the host
extern "C" int mg_step_embed(const int64_t *t, const float *freq, float *emb, int B, int D, void *stream)
{
    if (!t || !freq || !emb) return MG_ERR_ARG;
    if (B <= 0 || D <= 0 || D % 2) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(step_embed_kernel, dim3(mg_cdiv(B * (D / 2), 256)), dim3(256), 0, (hipStream_t)stream, t, freq, emb,
                       B, D);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
