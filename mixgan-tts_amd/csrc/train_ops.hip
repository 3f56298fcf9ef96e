// Train-mode normalisation layers of the coarse-mel decoder / PostNet with their backward passes, for aux
// pre-training: post-LayerNorm with the dropout of transformer/SubLayers.py:54-55,
// 90-91 fused in front of it, and BatchNorm1d (batch statistics) + tanh + dropout of transformer/Layers.py:
// 131-134.  All tensors channel-major [B, C, L]; every kernel is HBM-bound (one read of each input, one write
// of each output), dropout keep-masks are uint8 [B, C, L] made by the caller (so tests can inject them).
#include "common.h"

#define MG_LN_SLOTS 32

// ---------------------------------------------------------------------------------------------
// pre = a * keep * drop_scale + res;  out = pad ? 0 : LN_c(pre) * gamma + beta          (C == 256)
// 256 threads = 32 frames x 8 channel groups, as the inference kernel (attention.hip).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_train_fwd_kernel(const float *__restrict__ a,
                                                                  const uint8_t *__restrict__ keep, float drop_scale,
                                                                  const float *__restrict__ res,
                                                                  const float *__restrict__ gamma,
                                                                  const float *__restrict__ beta,
                                                                  const uint8_t *__restrict__ pad,
                                                                  float *__restrict__ pre, float *__restrict__ out, int L,
                                                                  float eps)
{
    constexpr int C = 256, G = 8, PER = C / G;
    __shared__ float red[G][32];
    const int f = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int b = blockIdx.y;
    const int l = blockIdx.x * 32 + f;
    const int lc = min(l, L - 1);
    const size_t base = (size_t)b * C * L + lc;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const size_t o = base + (size_t)(g * PER + i) * L;
        float x = a[o];
        if (keep) x = keep[o] ? x * drop_scale : 0.f;
        v[i] = x + (res ? res[o] : 0.f);
        s += v[i];
    }
    red[g][f] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) mean += red[k][f];
    mean *= (1.f / C);
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    red[g][f] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) var += red[k][f];
    const float rstd = rsqrtf(var * (1.f / C) + eps);
    if (l < L) {
        const bool z = pad && pad[(size_t)b * L + l];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int c = g * PER + i;
            const size_t o = (size_t)b * C * L + (size_t)c * L + l;
            if (pre) pre[o] = v[i];
            out[o] = z ? 0.f : (v[i] - mean) * rstd * gamma[c] + beta[c];
        }
    }
}

// Backward of the above from the saved `pre`:  with xh = (pre - mean) * rstd and g = dy * gamma (0 on padded
// frames), d_pre = rstd * (g - mean_c(g) - xh * mean_c(g * xh));  d_a = d_pre * keep * drop_scale;
// dgamma[c] += sum dy * xh, dbeta[c] += sum dy over the non-padded frames (fp32 atomics, one per channel
// per workgroup).
__global__ __launch_bounds__(256) void layernorm_train_bwd_kernel(const float *__restrict__ pre,
                                                                  const float *__restrict__ dy,
                                                                  const float *__restrict__ gamma,
                                                                  const uint8_t *__restrict__ pad,
                                                                  const uint8_t *__restrict__ keep, float drop_scale,
                                                                  float *__restrict__ d_pre, float *__restrict__ d_a,
                                                                  float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                  int L, float eps)
{
    constexpr int C = 256, G = 8, PER = C / G;
    __shared__ float red[2][G][32];
    const int f = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int b = blockIdx.y;
    const int l = blockIdx.x * 32 + f;
    const int lc = min(l, L - 1);
    const size_t base = (size_t)b * C * L + lc;
    const bool live = l < L && !(pad && pad[(size_t)b * L + lc]);
    float v[PER], gy[PER];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const size_t o = base + (size_t)(g * PER + i) * L;
        v[i] = pre[o];
        gy[i] = live ? dy[o] : 0.f;
        s += v[i];
    }
    red[0][g][f] = s;
    __syncthreads();
    float mean = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) mean += red[0][k][f];
    mean *= (1.f / C);
    __syncthreads();
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    red[0][g][f] = q;
    __syncthreads();
    float var = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) var += red[0][k][f];
    const float rstd = rsqrtf(var * (1.f / C) + eps);
    __syncthreads();
    // per-frame means of g and g * xh over the channels
    float sg = 0.f, sgx = 0.f;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const float xh = (v[i] - mean) * rstd;
        const float gg = gy[i] * gamma[g * PER + i];
        v[i] = xh;
        sg += gg;
        sgx += gg * xh;
    }
    red[0][g][f] = sg;
    red[1][g][f] = sgx;
    __syncthreads();
    float mg = 0.f, mgx = 0.f;
#pragma unroll
    for (int k = 0; k < G; ++k) {
        mg += red[0][k][f];
        mgx += red[1][k][f];
    }
    mg *= (1.f / C);
    mgx *= (1.f / C);
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int c = g * PER + i;
        const size_t o = (size_t)b * C * L + (size_t)c * L + l;
        const float d = rstd * (gy[i] * gamma[c] - mg - v[i] * mgx);
        if (l < L) {
            d_pre[o] = d;
            if (d_a) d_a[o] = keep ? (keep[o] ? d * drop_scale : 0.f) : d;
        }
        // channel sums over this workgroup's 32 frames: the 32 lanes of a half-wave share g
        float pg = gy[i] * v[i], pb = gy[i];
#pragma unroll
        for (int o2 = 16; o2 > 0; o2 >>= 1) {
            pg += __shfl_xor(pg, o2, 64);
            pb += __shfl_xor(pb, o2, 64);
        }
        if (f == 0) {   // MG_LN_SLOTS copies of the two vectors: 1/32 of the same-address contention
            const int slot = (blockIdx.x + blockIdx.y) % MG_LN_SLOTS;
            atomicAdd(dgamma + slot * C + c, pg);
            atomicAdd(dbeta + slot * C + c, pb);
        }
    }
}

extern "C" int mg_layernorm_cm_train_fwd(const float *a, const uint8_t *keep, float drop_scale, const float *res,
                                         const float *gamma, const float *beta, const uint8_t *pad, float *pre, float *out,
                                         int B, int C, int L, float eps, void *stream)
{
    if (!a || !gamma || !beta || !out) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || C != 256) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(layernorm_train_fwd_kernel, dim3(mg_cdiv(L, 32), B), dim3(256), 0, (hipStream_t)stream, a, keep,
                       drop_scale, res, gamma, beta, pad, pre, out, L, eps);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// dgamma / dbeta are [MG_LN_SLOTS = 32][C] partial sums, ACCUMULATED into (the caller zeroes them and adds the 32
// rows: one same-address atomic per channel per workgroup costs 5x the kernel's HBM time otherwise).
extern "C" int mg_layernorm_cm_bwd(const float *pre, const float *dy, const float *gamma, const uint8_t *pad,
                                   const uint8_t *keep, float drop_scale, float *d_pre, float *d_a, float *dgamma,
                                   float *dbeta, int B, int C, int L, float eps, void *stream)
{
    if (!pre || !dy || !gamma || !d_pre || !dgamma || !dbeta) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || C != 256) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(layernorm_train_bwd_kernel, dim3(mg_cdiv(L, 32), B), dim3(256), 0, (hipStream_t)stream, pre, dy, gamma,
                       pad, keep, drop_scale, d_pre, d_a, dgamma, dbeta, L, eps);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm1d in training mode on [B, C, L]: statistics over (B, L) per channel.
// ---------------------------------------------------------------------------------------------
static __device__ __forceinline__ float block_sum(float v, float *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// mean[c], var[c] (biased) over (B, L); moments are taken around a per-channel shift (the channel's first
// sample) so that E[d^2] - E[d]^2 stays well conditioned in fp32.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float *__restrict__ x, int B, int C, int L,
                                                       float *__restrict__ mean, float *__restrict__ var)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    const float shift = x[(size_t)c * L];
    float s = 0.f, q = 0.f;
    for (int b = 0; b < B; ++b) {
        const float *p = x + ((size_t)b * C + c) * L;
        for (int l = threadIdx.x; l < L; l += 256) {
            const float d = p[l] - shift;
            s += d;
            q = fmaf(d, d, q);
        }
    }
    s = block_sum(s, red);
    q = block_sum(q, red);
    if (threadIdx.x == 0) {
        const float n = (float)B * (float)L;
        const float m = s / n;
        mean[c] = shift + m;
        var[c] = fmaxf(q / n - m * m, 0.f);   // biased (normalisation) variance
    }
}

// y = act((x - mean) * invstd * gamma + beta)  (act: MG_ACT_NONE | MG_ACT_TANH);  out = y * keep * drop_scale
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float *__restrict__ x, const float *__restrict__ mean,
                                                         const float *__restrict__ invstd, const float *__restrict__ gamma,
                                                         const float *__restrict__ beta, const uint8_t *__restrict__ keep,
                                                         float drop_scale, int act, float *__restrict__ y,
                                                         float *__restrict__ out, int C, int L, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i / L) % C);
        float v = (x[i] - mean[c]) * invstd[c] * gamma[c] + beta[c];
        if (act == MG_ACT_TANH) v = mg_tanh(v);
        if (y) y[i] = v;
        out[i] = keep ? (keep[i] ? v * drop_scale : 0.f) : v;
    }
}

// per channel: dbeta = sum dpre, dgamma = sum dpre * xh, with dpre = dout * keep * drop_scale * act'(y)
__global__ __launch_bounds__(256) void bn_act_bwd_reduce_kernel(const float *__restrict__ dout,
                                                                const uint8_t *__restrict__ keep, float drop_scale,
                                                                const float *__restrict__ y, const float *__restrict__ x,
                                                                const float *__restrict__ mean,
                                                                const float *__restrict__ invstd, int act, int B, int C,
                                                                int L, float *__restrict__ dgamma,
                                                                float *__restrict__ dbeta)
{
    __shared__ float red[4];
    const int c = blockIdx.x;
    const float m = mean[c], is = invstd[c];
    float sb = 0.f, sg = 0.f;
    for (int b = 0; b < B; ++b) {
        const size_t o = ((size_t)b * C + c) * L;
        for (int l = threadIdx.x; l < L; l += 256) {
            float d = dout[o + l];
            if (keep) d = keep[o + l] ? d * drop_scale : 0.f;
            if (act == MG_ACT_TANH) {
                const float yy = y[o + l];
                d *= 1.f - yy * yy;
            }
            sb += d;
            sg = fmaf(d, (x[o + l] - m) * is, sg);
        }
    }
    sb = block_sum(sb, red);
    sg = block_sum(sg, red);
    if (threadIdx.x == 0) {
        dbeta[c] = sb;
        dgamma[c] = sg;
    }
}

// dx = gamma * invstd * (dpre - dbeta / N - xh * dgamma / N);   N = count (B*L, or the global count when the
// statistics were all-reduced over ranks: then dbeta / dgamma passed in are the all-reduced sums)
__global__ __launch_bounds__(256) void bn_act_bwd_apply_kernel(const float *__restrict__ dout,
                                                               const uint8_t *__restrict__ keep, float drop_scale,
                                                               const float *__restrict__ y, const float *__restrict__ x,
                                                               const float *__restrict__ mean,
                                                               const float *__restrict__ invstd,
                                                               const float *__restrict__ gamma,
                                                               const float *__restrict__ dgamma,
                                                               const float *__restrict__ dbeta, float inv_count, int act,
                                                               float *__restrict__ dx, int C, int L, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int c = (int)((i / L) % C);
        float d = dout[i];
        if (keep) d = keep[i] ? d * drop_scale : 0.f;
        if (act == MG_ACT_TANH) {
            const float yy = y[i];
            d *= 1.f - yy * yy;
        }
        const float xh = (x[i] - mean[c]) * invstd[c];
        dx[i] = gamma[c] * invstd[c] * (d - dbeta[c] * inv_count - xh * dgamma[c] * inv_count);
    }
}

static unsigned ew_blocks(size_t n) { return (unsigned)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192); }

extern "C" int mg_bn_stats(const float *x, float *mean, float *var, int B, int C, int L, void *stream)
{
    if (!x || !mean || !var) return MG_ERR_ARG;
    if (B <= 0 || C <= 0 || L <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(bn_stats_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, x, B, C, L, mean, var);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_bn_act_fwd(const float *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                             const uint8_t *keep, float drop_scale, int act, float *y, float *out, int B, int C, int L,
                             void *stream)
{
    if (!x || !mean || !invstd || !gamma || !beta || !out) return MG_ERR_ARG;
    if (act != MG_ACT_NONE && act != MG_ACT_TANH) return MG_ERR_ARG;
    if (act == MG_ACT_TANH && !y) return MG_ERR_ARG;
    if (B <= 0 || C <= 0 || L <= 0) return MG_ERR_SHAPE;
    const size_t n = (size_t)B * C * L;
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, x, mean, invstd, gamma, beta,
                       keep, drop_scale, act, y, out, C, L, n);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_bn_act_bwd_reduce(const float *dout, const uint8_t *keep, float drop_scale, const float *y, const float *x,
                                    const float *mean, const float *invstd, int act, float *dgamma, float *dbeta, int B,
                                    int C, int L, void *stream)
{
    if (!dout || !x || !mean || !invstd || !dgamma || !dbeta) return MG_ERR_ARG;
    if (act != MG_ACT_NONE && act != MG_ACT_TANH) return MG_ERR_ARG;
    if (act == MG_ACT_TANH && !y) return MG_ERR_ARG;
    if (B <= 0 || C <= 0 || L <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(bn_act_bwd_reduce_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dout, keep, drop_scale, y, x, mean,
                       invstd, act, B, C, L, dgamma, dbeta);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_bn_act_bwd_apply(const float *dout, const uint8_t *keep, float drop_scale, const float *y, const float *x,
                                   const float *mean, const float *invstd, const float *gamma, const float *dgamma,
                                   const float *dbeta, float inv_count, int act, float *dx, int B, int C, int L,
                                   void *stream)
{
    if (!dout || !x || !mean || !invstd || !gamma || !dgamma || !dbeta || !dx) return MG_ERR_ARG;
    if (act != MG_ACT_NONE && act != MG_ACT_TANH) return MG_ERR_ARG;
    if (act == MG_ACT_TANH && !y) return MG_ERR_ARG;
    if (B <= 0 || C <= 0 || L <= 0) return MG_ERR_SHAPE;
    const size_t n = (size_t)B * C * L;
    hipLaunchKernelGGL(bn_act_bwd_apply_kernel, dim3(ew_blocks(n)), dim3(256), 0, (hipStream_t)stream, dout, keep, drop_scale,
                       y, x, mean, invstd, gamma, dgamma, dbeta, inv_count, act, dx, C, L, n);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// Optimizer step on flat buffers (train.py:75-85: clip_grad_norm_ + Adam.step per optimizer)
// ---------------------------------------------------------------------------------------------
// The gradients of one optimizer already live in one flat buffer (distributed.py GradBucket); with the parameters and
// both moments flat as well, clip + Adam is two launches that move 4 + 28 bytes per parameter: a fixed-order
// sum of squares, then one pass that scales the gradient by the clip factor on the fly and updates p, m, v.
#define GN_BLOCKS 1024

__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float *__restrict__ g, size_t n, float *__restrict__ partial)
{
    __shared__ float red[4];
    const size_t n4 = n >> 2, stride = (size_t)gridDim.x * 256;
    float s = 0.f;
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = g4[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const float v = g[(n4 << 2) + threadIdx.x];
        s += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// out[0] = ||g||_2, out[1] = min(1, max_norm / (||g|| + 1e-6))  (torch.nn.utils.clip_grad_norm_)
__global__ __launch_bounds__(256) void grad_norm_finish_kernel(const float *__restrict__ partial, int nblocks, float max_norm,
                                                               float *__restrict__ out)
{
    __shared__ double red[4];
    double s = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)partial[i];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt((red[0] + red[1]) + (red[2] + red[3]));
        out[0] = norm;
        const float c = max_norm / (norm + 1e-6f);
        out[1] = max_norm > 0.f ? (c < 1.f ? c : 1.f) : 1.f;
    }
}

struct AdamArgs {
    float *p, *m, *v;
    const float *g;
    const float *scale;   // device scalar multiplied into g (the clip factor), or null
    size_t n;
    float lr_over_bc1, inv_sqrt_bc2, beta1, beta2, eps, weight_decay;
    const float *hyper;   // optional device pair {lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)} read instead of the two above:
                          // a captured hipGraph replays this launch with the step count and lr of the moment
};

__device__ __forceinline__ void adam_one(float &p, float &m, float &v, float g, const AdamArgs &a)
{
    if (a.weight_decay != 0.f) g += a.weight_decay * p;          // L2 form of torch.optim.Adam
    m += (1.f - a.beta1) * (g - m);                              // lerp, as torch's _single_tensor_adam
    v = a.beta2 * v + (1.f - a.beta2) * g * g;
    const float denom = sqrtf(v) * a.inv_sqrt_bc2 + a.eps;
    p -= a.lr_over_bc1 * (m / denom);
}

__global__ __launch_bounds__(256) void adam_flat_kernel(AdamArgs a)
{
    if (a.hyper) {
        a.lr_over_bc1 = a.hyper[0];
        a.inv_sqrt_bc2 = a.hyper[1];
    }
    const float sc = a.scale ? a.scale[0] : 1.f;
    const size_t n4 = a.n >> 2, stride = (size_t)gridDim.x * 256;
    f32x4 *p4 = reinterpret_cast<f32x4 *>(a.p), *m4 = reinterpret_cast<f32x4 *>(a.m), *v4 = reinterpret_cast<f32x4 *>(a.v);
    const f32x4 *g4 = reinterpret_cast<const f32x4 *>(a.g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        f32x4 p = p4[i], m = m4[i], v = v4[i];
        const f32x4 g = g4[i];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float pj = p[j], mj = m[j], vj = v[j];
            adam_one(pj, mj, vj, g[j] * sc, a);
            p[j] = pj;
            m[j] = mj;
            v[j] = vj;
        }
        p4[i] = p;
        m4[i] = m;
        v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const size_t i = (n4 << 2) + threadIdx.x;
        adam_one(a.p[i], a.m[i], a.v[i], a.g[i] * sc, a);
    }
}

extern "C" size_t mg_grad_norm_scratch_floats(void) { return GN_BLOCKS; }

extern "C" int mg_grad_norm(const float *g, size_t n, float max_norm, float *scratch, float *out, void *stream)
{
    if (!g || !scratch || !out) return MG_ERR_ARG;
    if (n == 0) return MG_ERR_SHAPE;
    if (((uintptr_t)g & 15) != 0) return MG_ERR_ARG;
    const size_t want = (n / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > GN_BLOCKS ? GN_BLOCKS : want));
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, n, scratch);
    MG_LAUNCH_CHECK();
    hipLaunchKernelGGL(grad_norm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, scratch, blocks, max_norm, out);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_adam_flat(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2,
                            float eps, float weight_decay, long step, const float *grad_scale, void *stream)
{
    return mg_adam_flat_dev(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale, nullptr, stream);
}

extern "C" int mg_adam_flat_dev(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2,
                                float eps, float weight_decay, long step, const float *grad_scale, const float *hyper,
                                void *stream)
{
    if (!p || !g || !m || !v) return MG_ERR_ARG;
    if (n == 0 || step < 1) return MG_ERR_SHAPE;
    if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return MG_ERR_ARG;
    AdamArgs a;
    a.p = p;
    a.m = m;
    a.v = v;
    a.g = g;
    a.scale = grad_scale;
    a.hyper = hyper;
    a.n = n;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    a.lr_over_bc1 = (float)((double)lr / bc1);
    a.inv_sqrt_bc2 = (float)(1.0 / sqrt(bc2));
    a.beta1 = beta1;
    a.beta2 = beta2;
    a.eps = eps;
    a.weight_decay = weight_decay;
    const size_t want = (n / 4 + 255) / 256;
    const int blocks = (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    hipLaunchKernelGGL(adam_flat_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
