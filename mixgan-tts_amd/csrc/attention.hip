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
//         from LDS with the same key permutation.
#include "common.h"
#include <cstdlib>

#define AT_D 128   // head dim (d_k = d_v = decoder_hidden / decoder_head = 128)
#define AT_KT 64   // keys per LDS tile
#define AT_RSK 132 // K tile [key][d], d interleaved inside every 8 (position (d & 1) * 4 + ((d >> 1) & 3)): the fragments of
                   // four k-steps (d = 2s + h) of a lane are 16 contiguous bytes -> one ds_read_b128 per 4 MFMAs
#define AT_RSV 68  // V tile [d][key]: the four keys 8a + 4h + {0..3} of PV k-steps 4a .. 4a+3 are contiguous -> ds_read_b128
                   // (both strides = 4 mod 64 words: the 16-lane groups of a b128 read land on 16 distinct 4-bank slots)
#define AT_NEG (-1.0e30f)

// Round 2: one LDS read per FOUR MFMAs (was one ds_read_b32 per MFMA), the next key tile's global loads in flight behind
// the current tile's MFMAs (was: loaded after the barrier, fully exposed), the key mask as one ballot word per 32 keys
// (was 16 LDS reads per block), the accumulator rescale skipped while the running maximum does not move.
// KSPLIT: a 64-query workgroup whose wave pairs share the queries and split the keys of every tile (wave = 2 * query
// group + key half), merged through LDS at the end.  Twice the workgroups for short sequences: B=16, L=1000, 2 heads is
// 256 workgroups of 128 queries -- one per CU, one wave per SIMD, nothing to hide the staging behind.
// NWQ: query waves per workgroup (4: 128 queries; 8: 256 queries -- every staged key tile then serves twice the queries:
// half the K/V re-reads from L2 and half the staging work per FLOP, for sequences long enough to still fill the chip).
template <bool VEC, bool KSPLIT, int NWQ = 4>
__global__ __launch_bounds__(NWQ * 64, 2) void attention_fwd_kernel(const float *__restrict__ qkv,
                                                               const uint8_t *__restrict__ key_pad,
                                                               float *__restrict__ out, int L, int n_head, float scale)
{
    __shared__ __attribute__((aligned(16))) float Kt[AT_KT * AT_RSK];
    __shared__ __attribute__((aligned(16))) float Vt[AT_D * AT_RSV];
    __shared__ unsigned kmask[2];   // bit j of word kb: key kb*32 + j of the tile is masked
    __shared__ float Pk[KSPLIT && NWQ == 8 ? 2 * 64 * 66 : 1];   // 8-wave key split: merge space of query groups 2, 3

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, r = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    constexpr int KCH = AT_D * 64 / (NWQ * 64);   // K channels per staging thread: 32 (4 waves) or 16 (8)
    constexpr int NVR = 2048 / (NWQ * 64);        // V float4s per staging thread: 8 or 4
    const int q0 = KSPLIT ? blockIdx.x * (NWQ * 16) + (wave >> 1) * 32 : blockIdx.x * (NWQ * 32) + wave * 32;
    const int HD = n_head * AT_D;
    const float *Q = qkv + ((size_t)b * 3 * HD + head * AT_D) * L;
    const float *K = Q + (size_t)HD * L;
    const float *V = K + (size_t)HD * L;

    // Q fragments for all 64 k-steps (d = 2s + h), pre-scaled by 1/temperature
    float qf[AT_D / 2];
    {
        const int q = min(q0 + r, L - 1);
#pragma unroll
        for (int s = 0; s < AT_D / 2; ++s) qf[s] = Q[(size_t)(2 * s + hh) * L + q] * (scale * 1.44269504088896341f);
    }
    f32x16 O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[i][j] = 0.f;
    float m_run = AT_NEG, l_run = 0.f;

    // staging roles.  K: thread <-> (key, 32 channels): 32 coalesced row reads, 8 ds_write_b128 into the key's row.
    //                 V: thread <-> (channel, 4 keys) x 8: one 16-byte read, one ds_write_b128.
    const int skey = tid & 63, sdq = tid >> 6;
    float kreg[KCH];
    f32x4 vreg[NVR];
    bool kmasked = false;
    auto load_tile = [&](int kt0) {
        const int key = kt0 + skey;
        const int kc = min(key, L - 1);
        const float *kp = K + (size_t)(KCH * sdq) * L + kc;
#pragma unroll
        for (int i = 0; i < KCH; ++i) kreg[i] = kp[(size_t)i * L];
#pragma unroll
        for (int k = 0; k < NVR; ++k) {
            const int d = (tid >> 4) + (NWQ * 4) * k, k4 = tid & 15;
            const int f0 = kt0 + 4 * k4;
            if (VEC) {
                vreg[k] = *reinterpret_cast<const f32x4 *>(V + (size_t)d * L + min(f0, L - 4));
                if (f0 >= L) vreg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = V[(size_t)d * L + min(f0 + e, L - 1)];
                    vreg[k][e] = f0 + e < L ? v : 0.f;
                }
            }
        }
        if (tid < AT_KT) kmasked = key >= L || (key_pad && key_pad[(size_t)b * L + kc]);
        if (key >= L) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) kreg[i] = 0.f;
        }
    };
    auto store_tile = [&]() {
        float *kd = Kt + skey * AT_RSK + KCH * sdq;
#pragma unroll
        for (int sg = 0; sg < KCH / 8; ++sg) {
            const f32x4 even = {kreg[8 * sg], kreg[8 * sg + 2], kreg[8 * sg + 4], kreg[8 * sg + 6]};
            const f32x4 odd = {kreg[8 * sg + 1], kreg[8 * sg + 3], kreg[8 * sg + 5], kreg[8 * sg + 7]};
            *reinterpret_cast<f32x4 *>(kd + 8 * sg) = even;
            *reinterpret_cast<f32x4 *>(kd + 8 * sg + 4) = odd;
        }
#pragma unroll
        for (int k = 0; k < NVR; ++k)
            *reinterpret_cast<f32x4 *>(Vt + ((tid >> 4) + (NWQ * 4) * k) * AT_RSV + 4 * (tid & 15)) = vreg[k];
        if (tid < AT_KT) {
            const unsigned long long bal = __ballot(kmasked);   // wave 0: lane j <-> key j of the tile
            if (tid == 0) {
                kmask[0] = (unsigned)bal;
                kmask[1] = (unsigned)(bal >> 32);
            }
        }
    };

    load_tile(0);
    for (int kt0 = 0; kt0 < L; kt0 += AT_KT) {
        __syncthreads();  // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (kt0 + AT_KT < L) load_tile(kt0 + AT_KT);   // flies behind the 256 MFMAs below
#pragma unroll
        for (int kb = KSPLIT ? (wave & 1) : 0; kb < (KSPLIT ? (wave & 1) + 1 : AT_KT / 32); ++kb) {
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) S[j] = 0.f;
            const float *Kp = Kt + (kb * 32 + r) * AT_RSK + 4 * hh;
#pragma unroll
            for (int sg = 0; sg < AT_D / 8; ++sg) {
                const f32x4 a = *reinterpret_cast<const f32x4 *>(Kp + 8 * sg);
#pragma unroll
                for (int e = 0; e < 4; ++e) S = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], qf[4 * sg + e], S, 0, 0, 0);
            }
            // S[p] = score(key = kb*32 + 8(p>>2) + 4h + (p&3), query = r)
            // scores are in log2 units (Q carries scale * log2 e): the exponentials are bare v_exp_f32.  Blocks without a
            // masked key -- all but the last tile of an unpadded utterance -- skip the per-element mask arithmetic, which
            // is as many VALU cycles as the softmax itself.
            const unsigned kw = __builtin_amdgcn_readfirstlane(kmask[kb]);
            float mx = AT_NEG, ps = 0.f, m_new, corr;
            if (kw == 0u) {
#pragma unroll
                for (int p = 0; p < 16; ++p) mx = fmaxf(mx, S[p]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                m_new = fmaxf(m_run, mx);
                corr = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    S[p] = __builtin_amdgcn_exp2f(S[p] - m_new);
                    ps += S[p];
                }
            } else {
                const unsigned mw = kw >> (4 * hh);
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const bool masked = (mw >> (8 * (p >> 2) + (p & 3))) & 1u;
                    const float v = masked ? AT_NEG : S[p];
                    S[p] = v;
                    mx = fmaxf(mx, v);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                m_new = fmaxf(m_run, mx);
                corr = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const bool masked = (mw >> (8 * (p >> 2) + (p & 3))) & 1u;
                    const float e = masked ? 0.f : __builtin_amdgcn_exp2f(S[p] - m_new);
                    S[p] = e;
                    ps += e;
                }
            }
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * corr + ps;
            if (__any(m_new != m_run)) {   // wave-uniform: once the running maxima have settled this is skipped
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) O[i][j] *= corr;
            }
            m_run = m_new;
            // O^T[d][query] += V[d][key] P^T[key][query]; k-step p <-> key 8(p>>2) + 4h + (p&3)
            const float *Vp = Vt + r * AT_RSV + kb * 32 + 4 * hh;
#pragma unroll
            for (int aa = 0; aa < 4; ++aa) {
                f32x4 va[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) va[i] = *reinterpret_cast<const f32x4 *>(Vp + (i * 32) * AT_RSV + 8 * aa);
#pragma unroll
                for (int bq = 0; bq < 4; ++bq)
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        O[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(va[i][bq], S[4 * aa + bq], O[i], 0, 0, 0);
            }
        }
    }
    if (KSPLIT) {
        // merge the two key halves of a query group: the odd wave parks (m, l, O) in the K tile's storage
        // (2 query groups x 64 lanes x 66 floats = the tile's 8448 floats exactly)
        __syncthreads();   // the last tile is consumed
        const int qg = wave >> 1;
        float *park = (qg < 2 ? Kt + (qg * 64 + lane) * 66 : Pk + ((qg - 2) * 64 + lane) * 66);
        if (wave & 1) {
            park[0] = m_run;
            park[1] = l_run;
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) park[2 + 16 * i + j] = O[i][j];
        }
        __syncthreads();
        if (wave & 1) return;
        const float m1 = park[0], l1 = park[1];
        const float m = fmaxf(m_run, m1);
        const float c0 = __builtin_amdgcn_exp2f(m_run - m), c1 = __builtin_amdgcn_exp2f(m1 - m);   // log2 units
        l_run = l_run * c0 + l1 * c1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) O[i][j] = O[i][j] * c0 + park[2 + 16 * i + j] * c1;
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
#define AH_KRS 136   // halves per key row of Kt (272 B: 16-byte aligned rows -> the 8 halves of a fragment are one ds_read_b128)
#define AH_VRS 72    // halves per dv row of Vt; inside every 16 keys, key k sits at position 8 ((k >> 2) & 1) + 4 (k >> 3) + (k & 3):
                     // the 8 keys of a PV fragment (16t + 8 (j >> 2) + 4h + (j & 3)) are then contiguous -> one ds_read_b128

// Round 2: as the fp32 kernel -- the next key tile's global loads fly behind the current tile's work, fragments are single
// 16-byte LDS reads, the key mask is a ballot word, the accumulator rescale is skipped while the running maximum rests.
template <bool VEC, int NWQ = 4>
__global__ __launch_bounds__(NWQ * 64, 2) void attention_fwd_f16_kernel(const float *__restrict__ qkv,
                                                                   const uint8_t *__restrict__ key_pad,
                                                                   float *__restrict__ out, int L, int n_head,
                                                                   float scale)
{
    __shared__ __attribute__((aligned(16))) _Float16 Kt[AT_KT * AH_KRS];   // [key][d]
    __shared__ __attribute__((aligned(16))) _Float16 Vt[AT_D * AH_VRS];    // [dv][key, permuted inside 16]
    __shared__ unsigned kmask[2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hh = lane >> 5, r = lane & 31;
    const int b = blockIdx.z, head = blockIdx.y;
    constexpr int KCH = AT_D * 64 / (NWQ * 64), NVR = 2048 / (NWQ * 64);   // staging shares, as in the fp32 kernel
    const int q0 = blockIdx.x * (NWQ * 32) + wave * 32;
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
            for (int j = 0; j < 8; ++j) qf[s][j] = (_Float16)(Q[(size_t)(16 * s + 8 * hh + j) * L + q] * (scale * 1.44269504088896341f));
    }
    f32x16 O[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 16; ++j) O[i][j] = 0.f;
    float m_run = AT_NEG, l_run = 0.f;

    const int skey = tid & 63, sdq = tid >> 6;
    float kreg[KCH];
    f32x4 vreg[NVR];
    bool kmasked = false;
    auto load_tile = [&](int kt0) {
        const int key = kt0 + skey;
        const int kc = min(key, L - 1);
        const float *kp = K + (size_t)(KCH * sdq) * L + kc;
#pragma unroll
        for (int i = 0; i < KCH; ++i) kreg[i] = kp[(size_t)i * L];
#pragma unroll
        for (int k = 0; k < NVR; ++k) {
            const int d = (tid >> 4) + (NWQ * 4) * k, k4 = tid & 15;
            const int f0 = kt0 + 4 * k4;
            if (VEC) {
                vreg[k] = *reinterpret_cast<const f32x4 *>(V + (size_t)d * L + min(f0, L - 4));
                if (f0 >= L) vreg[k] = f32x4{0.f, 0.f, 0.f, 0.f};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = V[(size_t)d * L + min(f0 + e, L - 1)];
                    vreg[k][e] = f0 + e < L ? v : 0.f;
                }
            }
        }
        if (tid < AT_KT) kmasked = key >= L || (key_pad && key_pad[(size_t)b * L + kc]);
        if (key >= L) {
#pragma unroll
            for (int i = 0; i < KCH; ++i) kreg[i] = 0.f;
        }
    };
    auto store_tile = [&]() {
        _Float16 *kd = Kt + skey * AH_KRS + KCH * sdq;
#pragma unroll
        for (int c = 0; c < KCH / 8; ++c) {
            mg_half8 hv;
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[j] = (_Float16)kreg[8 * c + j];
            *reinterpret_cast<mg_half8 *>(kd + 8 * c) = hv;
        }
#pragma unroll
        for (int k = 0; k < NVR; ++k) {
            const int d = (tid >> 4) + (NWQ * 4) * k, k4 = tid & 15;
            // keys 4 k4 .. 4 k4 + 3: group of 16 = k4 >> 2, inside it quad (k4 & 3): h = quad & 1, hi = quad >> 1
            const int pos = 16 * (k4 >> 2) + 8 * (k4 & 1) + 4 * ((k4 >> 1) & 1);
            mg_half4 hv;
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = (_Float16)vreg[k][e];
            *reinterpret_cast<mg_half4 *>(Vt + d * AH_VRS + pos) = hv;
        }
        if (tid < AT_KT) {
            const unsigned long long bal = __ballot(kmasked);
            if (tid == 0) {
                kmask[0] = (unsigned)bal;
                kmask[1] = (unsigned)(bal >> 32);
            }
        }
    };

    load_tile(0);
    for (int kt0 = 0; kt0 < L; kt0 += AT_KT) {
        __syncthreads();  // previous tile fully consumed
        store_tile();
        __syncthreads();
        if (kt0 + AT_KT < L) load_tile(kt0 + AT_KT);
#pragma unroll
        for (int kb = 0; kb < AT_KT / 32; ++kb) {
            f32x16 S;
#pragma unroll
            for (int j = 0; j < 16; ++j) S[j] = 0.f;
            const _Float16 *Kp = Kt + (kb * 32 + r) * AH_KRS + 8 * hh;
#pragma unroll
            for (int s = 0; s < AT_D / 16; ++s)
                S = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const mg_half8 *>(Kp + 16 * s), qf[s], S, 0, 0, 0);
            // S[p] = score(key = kb*32 + 8(p>>2) + 4h + (p&3), query = r)
            // scores are in log2 units (Q carries scale * log2 e): the exponentials are bare v_exp_f32.  Blocks without a
            // masked key -- all but the last tile of an unpadded utterance -- skip the per-element mask arithmetic, which
            // is as many VALU cycles as the softmax itself.
            const unsigned kw = __builtin_amdgcn_readfirstlane(kmask[kb]);
            float mx = AT_NEG, ps = 0.f, m_new, corr;
            if (kw == 0u) {
#pragma unroll
                for (int p = 0; p < 16; ++p) mx = fmaxf(mx, S[p]);
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                m_new = fmaxf(m_run, mx);
                corr = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    S[p] = __builtin_amdgcn_exp2f(S[p] - m_new);
                    ps += S[p];
                }
            } else {
                const unsigned mw = kw >> (4 * hh);
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const bool masked = (mw >> (8 * (p >> 2) + (p & 3))) & 1u;
                    const float v = masked ? AT_NEG : S[p];
                    S[p] = v;
                    mx = fmaxf(mx, v);
                }
                mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
                m_new = fmaxf(m_run, mx);
                corr = __builtin_amdgcn_exp2f(m_run - m_new);
#pragma unroll
                for (int p = 0; p < 16; ++p) {
                    const bool masked = (mw >> (8 * (p >> 2) + (p & 3))) & 1u;
                    const float e = masked ? 0.f : __builtin_amdgcn_exp2f(S[p] - m_new);
                    S[p] = e;
                    ps += e;
                }
            }
            ps += __shfl_xor(ps, 32, 64);
            l_run = l_run * corr + ps;
            if (__any(m_new != m_run)) {
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 16; ++j) O[i][j] *= corr;
            }
            m_run = m_new;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                mg_half8 pb;
#pragma unroll
                for (int j = 0; j < 8; ++j) pb[j] = (_Float16)S[8 * t + j];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const _Float16 *Vp = Vt + (i * 32 + r) * AH_VRS + kb * 32 + 16 * t + 8 * hh;
                    O[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const mg_half8 *>(Vp), pb, O[i], 0, 0, 0);
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

// 256-query workgroups (8 waves) when they still number two per CU; MG_ATTENTION_WIDE=0/1 pins the choice (tests)
static bool attention_wide(int L, int n_head, int B)
{
    const char *e = std::getenv("MG_ATTENTION_WIDE");
    if (e) return e[0] == '1';
    return (long)mg_cdiv(L, 256) * n_head * B >= 512;
}

extern "C" int mg_attention_fwd_f16(const float *qkv, const uint8_t *key_pad, float *out, int B, int L, int n_head,
                                    int d_head, float scale, void *stream)
{
    if (!qkv || !out) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || n_head <= 0 || d_head != AT_D) return MG_ERR_SHAPE;
    const bool vec = (L % 4 == 0) && (((uintptr_t)qkv & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    if (attention_wide(L, n_head, B)) {   // 256-query workgroups
        dim3 grid(mg_cdiv(L, 256), n_head, B);
        if (vec) hipLaunchKernelGGL((attention_fwd_f16_kernel<true, 8>), grid, dim3(512), 0, st, qkv, key_pad, out, L, n_head, scale);
        else hipLaunchKernelGGL((attention_fwd_f16_kernel<false, 8>), grid, dim3(512), 0, st, qkv, key_pad, out, L, n_head, scale);
    } else {
        dim3 grid(mg_cdiv(L, 128), n_head, B);
        if (vec) hipLaunchKernelGGL((attention_fwd_f16_kernel<true, 4>), grid, dim3(256), 0, st, qkv, key_pad, out, L, n_head, scale);
        else hipLaunchKernelGGL((attention_fwd_f16_kernel<false, 4>), grid, dim3(256), 0, st, qkv, key_pad, out, L, n_head, scale);
    }
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_attention_fwd(const float *qkv, const uint8_t *key_pad, float *out, int B, int L, int n_head,
                                int d_head, float scale, void *stream)
{
    if (!qkv || !out) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || n_head <= 0 || d_head != AT_D) return MG_ERR_SHAPE;
    const bool vec = (L % 4 == 0) && (((uintptr_t)qkv & 15) == 0);
    // fewer than two 128-query workgroups per CU: 64-query workgroups that split the keys between wave pairs
    const char *ke = std::getenv("MG_ATTENTION_KSPLIT");   // tests pin each form
    // MG_ATTENTION_KSPLIT: 0 = never, 1 = 64-query workgroups (4 waves), 2 = 128-query workgroups (8 waves: four query
    // groups x two key halves -- two waves per SIMD from ONE workgroup per CU, at the staging cost of the plain form)
    const int ksplit = ke ? ke[0] - '0' : ((long)mg_cdiv(L, 128) * n_head * B < 512 ? 2 : 0);
    dim3 grid(mg_cdiv(L, ksplit == 1 ? 64 : 128), n_head, B);
    hipStream_t st = (hipStream_t)stream;
    if (!ksplit && attention_wide(L, n_head, B)) {
        dim3 wgrid(mg_cdiv(L, 256), n_head, B);
        if (vec) hipLaunchKernelGGL((attention_fwd_kernel<true, false, 8>), wgrid, dim3(512), 0, st, qkv, key_pad, out, L, n_head, scale);
        else hipLaunchKernelGGL((attention_fwd_kernel<false, false, 8>), wgrid, dim3(512), 0, st, qkv, key_pad, out, L, n_head, scale);
    } else if (ksplit == 2) {
        if (vec) hipLaunchKernelGGL((attention_fwd_kernel<true, true, 8>), grid, dim3(512), 0, st, qkv, key_pad, out, L, n_head, scale);
        else hipLaunchKernelGGL((attention_fwd_kernel<false, true, 8>), grid, dim3(512), 0, st, qkv, key_pad, out, L, n_head, scale);
    } else if (ksplit == 1) {
        if (vec) hipLaunchKernelGGL((attention_fwd_kernel<true, true>), grid, dim3(256), 0, st, qkv, key_pad, out, L, n_head, scale);
        else hipLaunchKernelGGL((attention_fwd_kernel<false, true>), grid, dim3(256), 0, st, qkv, key_pad, out, L, n_head, scale);
    } else {
        if (vec) hipLaunchKernelGGL((attention_fwd_kernel<true, false>), grid, dim3(256), 0, st, qkv, key_pad, out, L, n_head, scale);
        else hipLaunchKernelGGL((attention_fwd_kernel<false, false>), grid, dim3(256), 0, st, qkv, key_pad, out, L, n_head, scale);
    }
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
