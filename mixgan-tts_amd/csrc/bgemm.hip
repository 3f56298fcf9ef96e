// Training-mode attention of the FFT blocks (transformer/Modules.py:16-23, SubLayers.py:29-57) and its
// backward, for aux pre-training.
//
// Inference streams the softmax (attention.hip) and never forms the score matrix.  Training needs the
// probabilities again in the backward pass, and at training sizes (B=8, L<=1500, 2 heads) they are 64-144 MB
// per layer -- 0.05 % of HBM -- so here they are simply kept: every contraction of the forward and the
// backward is then one strided, batched fp32 GEMM on the MFMA over operands in the layouts they already have
// (channel-major [B, C, L] activations, row-major [B*H, L, L] probabilities):
//     S[q,k]   = sum_d Q[d,q] K[d,k]          P = softmax_k(scale * S + key mask)      O[d,q] = sum_k V[d,k] P[q,k]
//     dV[d,k]  = sum_q dO[d,q] P[q,k]         dP[q,k] = sum_d dO[d,q] V[d,k]
//     dS       = scale * P o (dP - rowsum(dP o P))
//     dQ[d,q]  = sum_k K[d,k] dS[q,k]         dK[d,k] = sum_q Q[d,q] dS[q,k]
//
// mg_bgemm: C[z][m][n] (+)= alpha * sum_k A[z](m,k) B[z](k,n), z = (b, h) with separate batch / head strides,
// arbitrary element strides for A and B (one of the two strides of each operand must be 1), C row-major.
// Workgroup tile 128 x 128 x 16, 4 waves as 2 x 2, wave tile 64 x 64 = 2 x 2 accumulators of
// v_mfma_f32_32x32x2_f32 (64 x 64 / 1 x 1 when that under-fills the chip); operand tiles are staged k-major in
// LDS ([16][tile + 1]) through registers with the next
// tile's global loads in flight behind the current tile's 32 MFMAs per wave.
#include "common.h"

struct BgemmArgs {
    const float *A, *B;
    float *C;
    int M, N, K, H;
    long a_ms, a_ks, a_bs, a_hs;
    long b_ks, b_ns, b_bs, b_hs;
    long c_ms, c_bs, c_hs;
    float alpha;
    int accumulate;
};

#define BG_BK 16

// MC / NC: the M (resp. N) index is the contiguous one in memory, otherwise K is.
// TI x TJ accumulators per wave: workgroup tile (64 TI) x (64 TJ); the 64 x 64 tile is used when the 128 x 128
// one would leave the chip under-filled (the d = 128 row contractions O / dV / dQ / dK have M = 128).
template <bool A_MC, bool B_NC, int TI, int TJ>
__global__ __launch_bounds__(256, 2) void bgemm_kernel(BgemmArgs a)
{
    constexpr int TM = 64 * TI, TN = 64 * TJ, RSA = TM + 1, RSB = TN + 1;
    constexpr int NA = TM * BG_BK / 256, NB = TN * BG_BK / 256;
    __shared__ float As[2][BG_BK * RSA];
    __shared__ float Bs[2][BG_BK * RSB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, hh = lane >> 5, r = lane & 31;
    const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
    const int zb = blockIdx.z / a.H, zh = blockIdx.z - zb * a.H;
    const float *A = a.A + (size_t)zb * a.a_bs + (size_t)zh * a.a_hs;
    const float *B = a.B + (size_t)zb * a.b_bs + (size_t)zh * a.b_hs;
    float *C = a.C + (size_t)zb * a.c_bs + (size_t)zh * a.c_hs;

    float ra[NA], rb[NB];
    auto gload = [&](int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * 256;
            const int am = A_MC ? (idx % TM) : (idx >> 4), ak = A_MC ? (idx / TM) : (idx & 15);
            const int gm = min(m0 + am, a.M - 1), gka = min(kt * BG_BK + ak, a.K - 1);
            ra[i] = A[(size_t)gm * a.a_ms + (size_t)gka * a.a_ks];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * 256;
            const int bn = B_NC ? (idx % TN) : (idx >> 4), bk = B_NC ? (idx / TN) : (idx & 15);
            const int gn = min(n0 + bn, a.N - 1), gkb = min(kt * BG_BK + bk, a.K - 1);
            rb[i] = B[(size_t)gkb * a.b_ks + (size_t)gn * a.b_ns];
        }
    };
    auto sstore = [&](int buf, int kt) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int idx = tid + i * 256;
            const int am = A_MC ? (idx % TM) : (idx >> 4), ak = A_MC ? (idx / TM) : (idx & 15);
            As[buf][ak * RSA + am] = (m0 + am < a.M && kt * BG_BK + ak < a.K) ? ra[i] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int idx = tid + i * 256;
            const int bn = B_NC ? (idx % TN) : (idx >> 4), bk = B_NC ? (idx / TN) : (idx & 15);
            Bs[buf][bk * RSB + bn] = (n0 + bn < a.N && kt * BG_BK + bk < a.K) ? rb[i] : 0.f;
        }
    };

    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int nk = (a.K + BG_BK - 1) / BG_BK;
    gload(0);
    sstore(0, 0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int ktn = kt + 1 < nk ? kt + 1 : kt;
        gload(ktn);
        const float *Ac = As[kt & 1] + wm * 32 * TI + r, *Bc = Bs[kt & 1] + wn * 32 * TJ + r;
#pragma unroll
        for (int s = 0; s < BG_BK / 2; ++s) {
            float av[TI], bv[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) av[i] = Ac[(2 * s + hh) * RSA + 32 * i];
#pragma unroll
            for (int j = 0; j < TJ; ++j) bv[j] = Bc[(2 * s + hh) * RSB + 32 * j];
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (kt + 1 < nk) sstore((kt + 1) & 1, kt + 1);
        __syncthreads();
    }

    int nc[TJ];
    bool nok[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        const int n = n0 + wn * 32 * TJ + j * 32 + r;
        nok[j] = n < a.N;
        nc[j] = nok[j] ? n : a.N - 1;
    }
#pragma unroll
    for (int i = 0; i < TI; ++i) {
        float old[16][TJ];
        size_t ro[16];
        bool rok[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int m = m0 + wm * 32 * TI + i * 32 + 8 * (e >> 2) + 4 * hh + (e & 3);
            rok[e] = m < a.M;
            ro[e] = (size_t)(rok[e] ? m : a.M - 1) * a.c_ms;
        }
        if (a.accumulate) {
#pragma unroll
            for (int e = 0; e < 16; ++e)
#pragma unroll
                for (int j = 0; j < TJ; ++j) old[e][j] = C[ro[e] + nc[j]];
        }
#pragma unroll
        for (int e = 0; e < 16; ++e)
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                float v = acc[i][j][e] * a.alpha;
                if (a.accumulate) v += old[e][j];
                if (rok[e] && nok[j]) C[ro[e] + nc[j]] = v;
            }
    }
}

extern "C" int mg_bgemm(const float *A, const float *B, float *C, int M, int N, int K, int batch, int heads, long a_ms,
                        long a_ks, long a_bs, long a_hs, long b_ks, long b_ns, long b_bs, long b_hs, long c_ms, long c_bs,
                        long c_hs, float alpha, int accumulate, void *stream)
{
    if (!A || !B || !C) return MG_ERR_ARG;
    if (M <= 0 || N <= 0 || K <= 0 || batch <= 0 || heads <= 0) return MG_ERR_SHAPE;
    if ((a_ms != 1 && a_ks != 1) || (b_ks != 1 && b_ns != 1) || c_ms < N) return MG_ERR_SHAPE;
    BgemmArgs a{A, B, C, M, N, K, heads, a_ms, a_ks, a_bs, a_hs, b_ks, b_ns, b_bs, b_hs, c_ms, c_bs, c_hs, alpha, accumulate};
    const bool amc = a_ms == 1, bnc = b_ns == 1;
    hipStream_t st = (hipStream_t)stream;
    const bool small = (long)mg_cdiv(N, 128) * mg_cdiv(M, 128) * batch * heads < 512;
#define BG_LAUNCH(AM, BN)                                                                                     \
    do {                                                                                                      \
        if (small)                                                                                            \
            hipLaunchKernelGGL((bgemm_kernel<AM, BN, 1, 1>), dim3(mg_cdiv(N, 64), mg_cdiv(M, 64), batch * heads), \
                               dim3(256), 0, st, a);                                                          \
        else                                                                                                  \
            hipLaunchKernelGGL((bgemm_kernel<AM, BN, 2, 2>), dim3(mg_cdiv(N, 128), mg_cdiv(M, 128), batch * heads), \
                               dim3(256), 0, st, a);                                                          \
    } while (0)
    if (amc && bnc) BG_LAUNCH(true, true);
    else if (amc) BG_LAUNCH(true, false);
    else if (bnc) BG_LAUNCH(false, true);
    else BG_LAUNCH(false, false);
#undef BG_LAUNCH
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// Row softmax of the score matrix, in place: S [B*H, L, L] -> P = softmax_k(scale * S[q, k] | key k of
// batch b padded -> -inf).  One wave per row; the row is read twice from L2 (L <= a few thousand floats).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_rows_kernel(float *__restrict__ S, const uint8_t *__restrict__ key_pad,
                                                           int L, int H, float scale, size_t rows)
{
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = (int)(row / ((size_t)H * L));
    float *s = S + row * L;
    const uint8_t *pad = key_pad ? key_pad + (size_t)b * L : nullptr;
    float m = -INFINITY;
    for (int k = lane; k < L; k += 64) {
        const float v = (pad && pad[k]) ? -INFINITY : s[k] * scale;
        m = fmaxf(m, v);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float sum = 0.f;
    for (int k = lane; k < L; k += 64) {
        const float v = (pad && pad[k]) ? -INFINITY : s[k] * scale;
        sum += __expf(v - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.f / sum;
    for (int k = lane; k < L; k += 64) {
        const float v = (pad && pad[k]) ? -INFINITY : s[k] * scale;
        s[k] = __expf(v - m) * inv;
    }
}

// dS = scale * P o (dP - sum_k dP o P), in place on dP.
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float *__restrict__ P, float *__restrict__ dP, int L,
                                                               float scale, size_t rows)
{
    const int lane = threadIdx.x & 63;
    const size_t row = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float *p = P + row * L;
    float *d = dP + row * L;
    float dot = 0.f;
    for (int k = lane; k < L; k += 64) dot = fmaf(p[k], d[k], dot);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    for (int k = lane; k < L; k += 64) d[k] = scale * p[k] * (d[k] - dot);
}

extern "C" int mg_softmax_rows_fwd(float *S, const uint8_t *key_pad, int B, int H, int L, float scale, void *stream)
{
    if (!S) return MG_ERR_ARG;
    if (B <= 0 || H <= 0 || L <= 0) return MG_ERR_SHAPE;
    const size_t rows = (size_t)B * H * L;
    hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, S, key_pad,
                       L, H, scale, rows);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_softmax_rows_bwd(const float *P, float *dP, int B, int H, int L, float scale, void *stream)
{
    if (!P || !dP) return MG_ERR_ARG;
    if (B <= 0 || H <= 0 || L <= 0) return MG_ERR_SHAPE;
    const size_t rows = (size_t)B * H * L;
    hipLaunchKernelGGL(softmax_rows_bwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP, L,
                       scale, rows);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
