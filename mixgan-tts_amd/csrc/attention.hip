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

// ---------------------------------------------------------------------------------------------
// Same streaming-softmax attention with fp16 operands on v_mfma_f32_32x32x16_f16 (fp32 accumulate, fp32
// softmax statistics, fp32 I/O) -- the "fp16 MFMA attention path" of BASELINE configs[4] (L = 4000 long-form:
// 262 GFLOP per layer at B=16).  Q (pre-scaled), K, V and the probabilities are rounded to fp16 when they
// become MFMA operands; one instruction covers 16 k-steps of the fp32 kernel.
//   S^T block: A = K[key i][d = 16s + 8h + j] (LDS, key-major rows), B = Q[d][query] (registers, 8 x half8)
//   O^T:       B = the lane's own 8 accumulator registers 8t..8t+7 converted to fp16 -- k index 8h + j of
//              PV step t is key 16t + 8(j>>2) + 4h + (j&3) of the 32-key block -- and A = V[dv][those keys]:
//              two 8-byte LDS reads from a dv-major row.
// ---------------------------------------------------------------------------------------------
typedef _Float16 mg_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 mg_half4 __attribute__((ext_vector_type(4)));
typedef _Float16 mg_half2 __attribute__((ext_vector_type(2)));
#define AH_KRS 132   // halves per key row of Kt (66 dwords: 2-way write conflicts, 8-byte aligned reads)
#define AH_VRS 72    // halves per dv row of Vt

__global__ __launch_bounds__(256, 2) void attention_fwd_f16_kernel(const float *__restrict__ qkv,
                                                                   const uint8_t *__restrict__ key_pad,
                                                                   float *__restrict__ out, int L, int n_head,
                                                                   float scale)
{
    __shared__ __attribute__((aligned(16))) _Float16 Kt[AT_KT * AH_KRS];   // [key][d]
    __shared__ __attribute__((aligned(16))) _Float16 Vt[AT_D * AH_VRS];    // [dv][key]
    __shared__ float kmask[AT_KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, r = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    const int q0 = blockIdx.x * 128 + wave * 32;
    const int HD = n_head * AT_D;
    const float *Q = qkv + ((size_t)b * 3 * HD + head * AT_D) * L;
    const float *K = Q + (size_t)HD * L;
    const float *V = K + (size_t)HD * L;

    mg_half8 qf[AT_D / 16];
    {
        const int q = min(q0 + r, L - 1);
#pragma unroll
        for (int s = 0; s < AT_D / 16; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = (_Float16)(Q[(size_t)(16 * s + 8 * hh + j) * L + q] * scale);
    }
    f32x16 O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[i][j] = 0.f;
    float m_run = AT_NEG, l_run = 0.f;

    for (int kt0 = 0; kt0 < L; kt0 += AT_KT) {
        __syncthreads();  // previous tile fully consumed
        // K: thread <-> (channel pair, key): two coalesced row reads, one 4-byte LDS write into the key's row
#pragma unroll
        for (int k = 0; k < (AT_D / 2 * AT_KT) / 256; ++k) {
            const int idx = tid + k * 256;
            const int d2 = idx >> 6, j = idx & 63;
            const int key = kt0 + j;
            const bool ok = key < L;
            const size_t off = (size_t)(2 * d2) * L + min(key, L - 1);
            const float k0 = K[off], k1 = K[off + L];
            mg_half2 hv;
            hv[0] = (_Float16)(ok ? k0 : 0.f);
            hv[1] = (_Float16)(ok ? k1 : 0.f);
            *reinterpret_cast<mg_half2 *>(Kt + j * AH_KRS + 2 * d2) = hv;
        }
        // V: thread <-> (channel, key pair)
#pragma unroll
        for (int k = 0; k < (AT_D * AT_KT / 2) / 256; ++k) {
            const int idx = tid + k * 256;
            const int d = idx >> 5, j2 = idx & 31;
            const int key = kt0 + 2 * j2;
            const size_t row = (size_t)d * L;
            const float v0 = V[row + min(key, L - 1)], v1 = V[row + min(key + 1, L - 1)];
            mg_half2 hv;
            hv[0] = (_Float16)(key < L ? v0 : 0.f);
            hv[1] = (_Float16)(key + 1 < L ? v1 : 0.f);
            *reinterpret_cast<mg_half2 *>(Vt + d * AH_VRS + 2 * j2) = hv;
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
            const _Float16 *Kp = Kt + (kb * 32 + r) * AH_KRS + 8 * hh;
#pragma unroll
            for (int s = 0; s < AT_D / 16; ++s) {
                const mg_half4 lo = *reinterpret_cast<const mg_half4 *>(Kp + 16 * s);
                const mg_half4 hi = *reinterpret_cast<const mg_half4 *>(Kp + 16 * s + 4);
                const mg_half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                S = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, qf[s], S, 0, 0, 0);
            }
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
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                mg_half8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (_Float16)S[8 * t + j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16 *Vp = Vt + (i * 32 + r) * AH_VRS + kb * 32 + 16 * t + 4 * hh;
                    const mg_half4 lo = *reinterpret_cast<const mg_half4 *>(Vp);
                    const mg_half4 hi = *reinterpret_cast<const mg_half4 *>(Vp + 8);
                    const mg_half8 a = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    O[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, pb, O[i], 0, 0, 0);
                }
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

extern "C" int mg_attention_fwd_f16(const float *qkv, const uint8_t *key_pad, float *out, int B, int L, int n_head,
                                    int d_head, float scale, void *stream)
{
    if (!qkv || !out) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || n_head <= 0 || d_head != AT_D) return MG_ERR_SHAPE;
    dim3 grid(mg_cdiv(L, 128), n_head, B);
    hipLaunchKernelGGL(attention_fwd_f16_kernel, grid, dim3(256), 0, (hipStream_t)stream, qkv, key_pad, out, L, n_head, scale);
    MG_LAUNCH_CHECK();
    return MG_OK;
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
