// FFT-block pieces for the shallow / aux coarse-mel decoder (transformer/SubLayers.py:8-93,
// transformer/Modules.py:6-25): streaming-softmax multi-head attention on the fp32 MFMA and the
// post-LayerNorm, both on the channel-major [B, C, L] layout used by the conv kernels.
//
// Attention never materialises the [n_head*B, L, L] score tensor of the reference
// (transformer/Modules.py:16-22: 2 GB per layer at B=16, L=4000).  Per (batch, head, 128-query
// tile) a workgroup of 4 waves walks 64-key tiles of K and V staged in LDS:
//   S^T = K_tile^T-major MFMA (A = K[key][d], B = Q[d][query]): the accumulator then has the QUERY
//         on the lane and the KEYS in its 16 registers, so the softmax row statistics are per-lane
//         scalars (one cross-half shuffle) -- no cross-lane reductions;
//   O^T += V P^T reuses that accumulator directly as the MFMA B operand: k-step (a,b) takes key
//         8a + 4h + b for lane half h, which is exactly register 4a+b of every lane, and V is read
//         from LDS with the same key permutation (odd row stride 65 -> conflict-free).
#include "common.h"

#define AT_D 128   // head dim (d_k = d_v = decoder_hidden / decoder_head = 128)
#define AT_KT 64   // keys per LDS tile
#define AT_RS 65   // LDS row stride (odd)
#define AT_NEG (-1.0e30f)

__global__ __launch_bounds__(256, 2) void attention_fwd_kernel(const float *__restrict__ qkv,
                                                               const uint8_t *__restrict__ key_pad,
                                                               float *__restrict__ out, int L, int n_head, float scale)
{
    __shared__ float Kt[AT_D * AT_RS];
    __shared__ float Vt[AT_D * AT_RS];
    __shared__ float kmask[AT_KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, r = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int HD = n_head * AT_D;
    const float *Q = qkv + ((size_t)b * 3 * HD + head * AT_D) * L;
    const float *K = Q + (size_t)HD * L;
    const float *V = K + (size_t)HD * L;

    // Q fragments for all 64 k-steps (d = 2s + h), pre-scaled by 1/temperature
    float qf[AT_D / 2];
    {
        const int q = min(q0 + r, L - 1);
#pragma unroll
        for (int s = 0; s < AT_D / 2; ++s) qf[s] = Q[(size_t)(2 * s + hh) * L + q] * scale;
    }
    f32x16 O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[i][j] = 0.f;
    float m_run = AT_NEG, l_run = 0.f;

    for (int kt0 = 0; kt0 < L; kt0 += AT_KT) {
        __syncthreads();  // previous tile fully consumed
#pragma unroll
        for (int k = 0; k < (AT_D * AT_KT) / 256; ++k) {
            const int idx = tid + k * 256;
            const int d = idx >> 6, j = idx & 63;
            const int key = kt0 + j;
            const bool ok = key < L;
            const size_t off = (size_t)d * L + min(key, L - 1);
            const float kv = K[off], vv = V[off];
            Kt[d * AT_RS + j] = ok ? kv : 0.f;
            Vt[d * AT_RS + j] = ok ? vv : 0.f;
        }
        if (tid < AT_KT) {
            const int key = kt0 + tid;
            const bool masked = key >= L || (key_pad && key_pad[(size_t)b * L + min(key, L - 1)]);
            kmask[tid] = masked ? 1.f : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < AT_KT / 32; ++kb) {
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) S[j] = 0.f;
            const float *Kp = Kt + hh * AT_RS + kb * 32 + r;
#pragma unroll
            for (int s = 0; s < AT_D / 2; ++s)
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(Kp[(2 * s) * AT_RS], qf[s], S, 0, 0, 0);
            // S[p] = score(key = kb*32 + 8(p>>2) + 4h + (p&3), query = r)
            float mk[16];
            float mx = AT_NEG;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                mk[p] = kmask[kb * 32 + 8 * (p >> 2) + 4 * hh + (p & 3)];
                const float v = mk[p] != 0.f ? AT_NEG : S[p];
                S[p] = v;
                mx = fmaxf(mx, v);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);
            const float corr = __expf(m_run - m_new);
            float ps = 0.f;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const float e = mk[p] != 0.f ? 0.f : __expf(S[p] - m_new);
                S[p] = e;
                ps += e;
            }
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * corr + ps;
            m_run = m_new;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) O[i][j] *= corr;
            // O^T[d][query] += V[d][key] P^T[key][query]; k-step p <-> key 8(p>>2) + 4h + (p&3)
            const float *Vp = Vt + r * AT_RS + kb * 32 + 4 * hh;
#pragma unroll
            for (int p = 0; p < 16; ++p) {
                const int kc = 8 * (p >> 2) + (p & 3);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    O[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(Vp[(i * 32) * AT_RS + kc], S[p], O[i], 0, 0, 0);
            }
        }
    }
    const int q = q0 + r;
    if (q < L) {
        const float inv = 1.f / l_run;
        float *o = out + ((size_t)b * HD + head * AT_D) * L + q;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int d = i * 32 + 8 * (j >> 2) + 4 * hh + (j & 3);
                o[(size_t)d * L] = O[i][j] * inv;
            }
    }
}

extern "C" int mg_attention_fwd(const float *qkv, const uint8_t *key_pad, float *out, int B, int L, int n_head,
                                int d_head, float scale, void *stream)
{
    if (!qkv || !out) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || n_head <= 0 || d_head != AT_D) return MG_ERR_SHAPE;
    dim3 grid(mg_cdiv(L, 128), n_head, B);
    hipLaunchKernelGGL(attention_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, key_pad, out, L, n_head, scale);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// out[b,c,l] = pad[b,l] ? 0 : LN_c(a[b,:,l] + res[b,:,l]) * gamma[c] + beta[c]     (C == 256)
// 256 threads = 32 frames x 8 channel groups; two-pass mean / variance in registers.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_cm_kernel(const float *__restrict__ a, const float *__restrict__ res,
                                                           const float *__restrict__ gamma,
                                                           const float *__restrict__ beta,
                                                           const uint8_t *__restrict__ pad, float *__restrict__ out,
                                                           int L, float eps)
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
        v[i] = a[o] + (res ? res[o] : 0.f);
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
            out[(size_t)b * C * L + (size_t)c * L + l] = z ? 0.f : (v[i] - mean) * rstd * gamma[c] + beta[c];
        }
    }
}

extern "C" int mg_layernorm_cm_fwd(const float *a, const float *res, const float *gamma, const float *beta,
                                   const uint8_t *pad, float *out, int B, int C, int L, float eps, void *stream)
{
    if (!a || !gamma || !beta || !out) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || C != 256) return MG_ERR_SHAPE;
    dim3 grid(mg_cdiv(L, 32), B);
    hipLaunchKernelGGL(layernorm_cm_kernel, grid, dim3(256), 0, (hipStream_t)stream, a, res, gamma, beta, pad, out, L, eps);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
