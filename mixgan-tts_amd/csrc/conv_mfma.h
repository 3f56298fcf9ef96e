// Conv1d / Linear as an implicit GEMM on the fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
// Data layout.  Activations are channel-major [B, C, L] with the frame axis contiguous -- the
// reference's own Conv1d layout -- so a [CK channels x TW frames] slab of the input is a set of
// contiguous row segments: coalesced HBM reads straight into an LDS tile, halo included.  With
// the GEMM written as   out[co, l] = sum_{(tap,ci)} W[co, (tap,ci)] * in[ci, l*stride + tap - pad]
// the MFMA B operand of k-step (ci, ci+1) is "32 consecutive frames of row ci / ci+1" of that
// LDS tile: one conflict-free ds_read_b32 per MFMA, no im2col, and the tap shift is an address
// offset.  The A operand (weights) is pre-packed in MFMA fragment order and streamed global/L2 ->
// VGPR as one float4 per lane per 4 k-steps (1 KiB per wave-instruction); every workgroup reads
// the same few MB of weights, which stay resident in each XCD's 4 MiB L2.
//
// Tiling.  Workgroup = 256 threads = 4 waves as 2(M) x 2(N); wave tile (32*WM) x 64, i.e.
// WM x 2 accumulator tiles of 32x32 (f32x16 each).  Workgroup tile MT = 64*WM output channels x
// NT = 128 output frames of one batch element.  K loop: chunks of CK input channels staged in a
// double-buffered LDS tile (register-staged prefetch of chunk c+1 behind the MFMAs of chunk c,
// one barrier per chunk).
//
// Packed weight layout (mg_conv_pack):  Wp[mb][q][lane][e], mb = 32-row block of M, q = k-group
// counter in loop order (chunk, tap, 8-channel group), lane 0..63, e 0..3:
//     ci  = chunk*CK + g*8 + 2*e + (lane >> 5),   row = rowmap(mb, lane & 31)
// so element e of the float4 is the A fragment (A[i = lane&31][k = lane>>5]) of k-step e.
#pragma once
#include "common.h"


struct ConvArgs {
    const float *in;      // [B, Ci, Lin]
    const float *in_vec;  // optional [B, Ci]: added to in-range input samples
    const float *wp;      // packed weights
    long in_bs;           // batch stride of `in` (floats)
    int in_rs;            // row (channel) stride of `in` (floats), normally Lin
    int B, Ci, CiP, Lin, Lout, pad;
    int ntiles_per_b, ntiles_total, mtiles;
    int dil;              // dilation (<= DILMAX of the instantiation)
    float in_slope;       // leaky-ReLU slope applied to the input samples while staging (1 = identity)
    // Split reduction (ksplit > 1): workgroup (tile, ks) reduces input-channel chunks [ks * cps, (ks + 1) * cps) and
    // leaves its accumulators in part[tile][ks]; conv_split_finalize_kernel (a second launch: the kernel boundary is
    // what makes partial tiles written on one XCD visible on another) adds the ksplit partial tiles in split order
    // and runs the epilogue.  For convolutions whose output is a few dozen tiles with a reduction thousands deep (the
    // JCU discriminator's 1/4-rate tail: 512 x 5 taps into 128 channels at L/4 frames, model/mixgantts.py:219-248),
    // where whole-reduction tiles leave most CUs idle.
    int ksplit, cps;
    float *part;          // [tiles][ksplit][WM * NNB * 16][256]
};

// One element of the packed weight stream (layout above): destination index `idx` inside a [MB][Q][64][4] block
// -> value, and the float offset it is stored at when the block is a q-slice [q0, q0 + Q) of a wider stream.
// Shared by pack_conv_kernel (conv_api.hip) and the single-launch table pack of mg_denoiser_pack.
struct PackDesc {
    int Co, Ci, K, CK, CiP, MB, mode, q0, Qtot, u;
};
__device__ __forceinline__ float mg_pack_element(const float *__restrict__ w, const PackDesc &d, size_t idx, size_t *dst)
{
    const int Q = d.CiP * d.K / 8;
    const int e = (int)(idx & 3);
    const int lane = (int)((idx >> 2) & 63);
    const size_t gq = idx >> 8;
    const int q = (int)(gq % Q);
    const int mb = (int)(gq / Q);
    const int qc = d.K * (d.CK / 8);
    const int chunk = q / qc;
    const int rem = q - chunk * qc;
    const int tap = rem / (d.CK / 8);
    const int g = rem - tap * (d.CK / 8);
    int ci = chunk * d.CK + g * 8 + 2 * e + (lane >> 5);
    int r = lane & 31;
    const int Co = d.Co, Ci = d.Ci, K = d.K;
    float v = 0.f;
    int mode = d.mode;
    if (mode == MG_PACK_PLAIN16 || mode == MG_PACK_GATE16) {
        // v_mfma_f32_16x16x4_f32 fragments (A[row = lane & 15][k = lane >> 4]) in the same container: the 256 floats of
        // (32-row block mb, 8-channel group g) hold 16-row block (g & 1) of mb for the 16-channel group (g >> 1) of
        // the chunk; element e of a lane is k-step e: channel 16 (g >> 1) + 4 e + (lane >> 4)
        r = 16 * (g & 1) + (lane & 15);
        ci = chunk * d.CK + 16 * (g >> 1) + 4 * e + (lane >> 4);
        mode = mode == MG_PACK_PLAIN16 ? MG_PACK_PLAIN : MG_PACK_GATE;
    }
    if (mode == MG_PACK_PLAIN) {
        const int row = mb * 32 + r;
        if (row < Co && ci < Ci) v = w[((size_t)row * Ci + ci) * K + tap];
    } else if (mode == MG_PACK_GATE) {
        const int half = mb & 1, rr = (mb >> 1) * 32 + r;
        if (rr < Co / 2 && ci < Ci) v = w[((size_t)(half * (Co / 2) + rr) * Ci + ci) * K + tap];
    } else if (mode == MG_PACK_TPOSE) {
        // ConvTranspose1d weight [Ci, Co', 2u] (stride u, padding u/2) as the 3-tap polyphase GEMM:
        // row = co*u + phase; output u*m + phase reads x[m + c0] with tap rho and x[m + c0 - 1] with
        // tap rho + u, where rho = (phase + u/2) % u, c0 = (phase + u/2) / u; here `Co` = Co' * u.
        const int row = mb * 32 + r, u = d.u;
        if (row < Co && ci < Ci) {
            const int co = row / u, ph = row - co * u;
            const int rho = (ph + u / 2) % u, c0 = (ph + u / 2) / u;
            const int t = tap == c0 + 1 ? rho : (tap == c0 ? rho + u : -1);
            if (t >= 0) v = w[((size_t)ci * (Co / u) + co) * (2 * u) + t];
        }
    } else {  // MG_PACK_DGRAD: rows = source Ci, reduction = source Co, taps flipped
        const int row = mb * 32 + r;
        if (row < Ci && ci < Co) v = w[((size_t)ci * Ci + row) * K + (K - 1 - tap)];
    }
    *dst = (((size_t)mb * d.Qtot + d.q0 + q) << 8) + (idx & 255);
    return v;
}

// number of k-groups (8 channels x 1 tap) per 32-row block
static inline int mg_conv_qcount(int CiP, int K) { return CiP * K / 8; }
static inline int mg_conv_ck(int K) { return K <= 3 ? 32 : 16; }
static inline int mg_round_up(int a, int b) { return (a + b - 1) / b * b; }

// ---------------------------------------------------------------------------------------------
// Epilogues.  acc[i][j][r] of the wave at (mrow0, l0w) is the output element
//     row = mrow0 + i*32 + 8*(r>>2) + 4*(lane>>5) + (r&3),   frame = l0w + j*32 + (lane&31).
// ---------------------------------------------------------------------------------------------
struct EpiBiasAct {
    struct Params {
        float *out;        // [B, Co, Lout]
        const float *bias; // [Co] or null
        const float *add;  // [B, Co, Lout] or null
        float alpha;
        int Co;
        int act;           // MG_ACT_*
        int accumulate;    // out += result
        long out_bs;       // batch stride of out (0 -> Co*Lout): lets the output be a channel slice
        const float *mask; // optional [B, Co, Lout] dense: result *= (mask > 0)   (ReLU backward)
        float act_slope;   // slope of MG_ACT_LRELU
        int phase_u;       // > 0: polyphase transposed conv, rows = Co * phase_u (see run_phases)
    };
    // Loads (residual add / ReLU mask / accumulate target) of a whole 32-row block are issued at clamped
    // addresses before any of them is used, and only the store is predicated: a load inside a divergent
    // `if (l < Lout)` makes hipcc fence every single one with s_waitcnt vmcnt(0), i.e. 16*WM*NNB serial
    // HBM round trips per wave.
    template <int WM, int NNB>
    static __device__ __forceinline__ void run(const Params &p, f32x16 (&acc)[WM][NNB], int b, int mrow0, int l0w,
                                               int lane, int Lout)
    {
        const int h = lane >> 5, c = lane & 31;
        if (p.phase_u > 0) {
            run_phases<WM, NNB>(p, acc, b, mrow0, l0w, lane, Lout);
            return;
        }
        int lc[NNB];
        bool lok[NNB];
#pragma unroll
        for (int j = 0; j < NNB; ++j) {
            const int l = l0w + j * 32 + c;
            lok[j] = l < Lout;
            lc[j] = lok[j] ? l : Lout - 1;
        }
        const bool has_add = p.add != nullptr, has_mask = p.mask != nullptr, acc_out = p.accumulate != 0;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
            if (mrow0 + i * 32 >= p.Co) continue;   // wave-uniform
#pragma unroll
            for (int half = 0; half < 2; ++half) {   // 8 rows x NNB frames per batch of loads
                float av[8][NNB], mv[8][NNB], ov[8][NNB], bv[8];
                size_t dense[8], orow[8];
                bool rok[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int rr = half * 8 + r;
                    const int row = mrow0 + i * 32 + 8 * (rr >> 2) + 4 * h + (rr & 3);
                    rok[r] = row < p.Co;
                    const int rc = rok[r] ? row : p.Co - 1;
                    dense[r] = ((size_t)b * p.Co + rc) * Lout;
                    orow[r] = p.out_bs ? (size_t)b * p.out_bs + (size_t)rc * Lout : dense[r];
                    bv[r] = p.bias ? p.bias[rc] : 0.f;
                }
                if (has_add) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
#pragma unroll
                        for (int j = 0; j < NNB; ++j) av[r][j] = p.add[dense[r] + lc[j]];
                }
                if (has_mask) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
#pragma unroll
                        for (int j = 0; j < NNB; ++j) mv[r][j] = p.mask[dense[r] + lc[j]];
                }
                if (acc_out) {
#pragma unroll
                    for (int r = 0; r < 8; ++r)
#pragma unroll
                        for (int j = 0; j < NNB; ++j) ov[r][j] = p.out[orow[r] + lc[j]];
                }
#pragma unroll
                for (int r = 0; r < 8; ++r) {
#pragma unroll
                    for (int j = 0; j < NNB; ++j) {
                        float v = acc[i][j][half * 8 + r] * p.alpha + bv[r];
                        switch (p.act) {
                        case MG_ACT_RELU: v = mg_act<MG_ACT_RELU>(v); break;
                        case MG_ACT_LRELU02: v = mg_act<MG_ACT_LRELU02>(v); break;
                        case MG_ACT_TANH: v = mg_act<MG_ACT_TANH>(v); break;
                        case MG_ACT_LRELU: v = v > 0.f ? v : p.act_slope * v; break;
                        default: break;
                        }
                        if (has_add) v += av[r][j];
                        if (has_mask) v = mv[r][j] > 0.f ? v : 0.f;
                        if (acc_out) v += ov[r][j];
                        if (rok[r] && lok[j]) p.out[orow[r] + lc[j]] = v;
                    }
                }
            }
        }
    }

    // Polyphase transposed convolution (mg_conv_transpose1d_fwd): GEMM row = co*u + r is phase r of
    // output channel co, GEMM column m is the input frame; element -> out[b, co, u*m + r].  The 4
    // consecutive rows a lane holds are 4 consecutive phases (u >= 4: one 16-byte store; the 2 x 32
    // lanes of a wave then write one contiguous run per channel) or 2 phases of 2 channels (u = 2).
    template <int WM, int NNB>
    static __device__ __forceinline__ void run_phases(const Params &p, f32x16 (&acc)[WM][NNB], int b, int mrow0,
                                                      int l0w, int lane, int Lm)
    {
        const int h = lane >> 5, c = lane & 31, u = p.phase_u;
        const int Mrows = p.Co * u;
        const size_t Lfull = (size_t)Lm * u;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                const int row0 = mrow0 + i * 32 + 8 * rq + 4 * h;
                if (row0 >= Mrows) continue;
#pragma unroll
                for (int j = 0; j < NNB; ++j) {
                    const int m = l0w + j * 32 + c;
                    if (m >= Lm) continue;
                    if (u >= 4) {
                        const int co = row0 / u, r0 = row0 - co * u;
                        const float bv = p.bias ? p.bias[co] : 0.f;
                        f32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = acc[i][j][rq * 4 + e] * p.alpha + bv;
                        *reinterpret_cast<f32x4 *>(p.out + ((size_t)b * p.Co + co) * Lfull + (size_t)m * u + r0) = v;
                    } else {   // u == 2
#pragma unroll
                        for (int e2 = 0; e2 < 2; ++e2) {
                            const int co = (row0 >> 1) + e2;
                            const float bv = p.bias ? p.bias[co] : 0.f;
                            float2 v;
                            v.x = acc[i][j][rq * 4 + 2 * e2] * p.alpha + bv;
                            v.y = acc[i][j][rq * 4 + 2 * e2 + 1] * p.alpha + bv;
                            *reinterpret_cast<float2 *>(p.out + ((size_t)b * p.Co + co) * Lfull + (size_t)m * 2) = v;
                        }
                    }
                }
            }
        }
    }
};

// ---------------------------------------------------------------------------------------------
template <int KW, int STRIDE, int CK, int DILMAX, int MW, int WM, int NNB, class Epi>
__global__ __launch_bounds__(256, 2) void conv_mfma_kernel(ConvArgs a, typename Epi::Params ep)
{
    static_assert(MW == 2 || (MW == 1 && WM == 1), "wave layouts: 2(M) x 2(N), or 1 x 4 for <= 32 output rows");
    constexpr int NW = 4 / MW;         // waves along the frame axis
    constexpr int NT = NW * 32 * NNB;  // output frames per workgroup
    constexpr int TW = NT * STRIDE + (KW - 1) * DILMAX;  // input frames per tile row (taps at multiples of a.dil)
    constexpr int TILE = CK * TW;
    constexpr int NLD = (TILE + 255) / 256;
    static_assert(NLD <= 64, "slab too large for the validity bit mask");
    constexpr int QC = KW * (CK / 8);  // k-groups per chunk

    __shared__ float lds[2][TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = MW == 2 ? wave >> 1 : 0, wn = MW == 2 ? wave & 1 : wave;
    const int h = lane >> 5, c32 = lane & 31;

    const int ks = a.ksplit > 1 ? (int)(blockIdx.x % (unsigned)a.ksplit) : 0;   // the splits of a tile are neighbours in the grid
    const int tile_id = a.ksplit > 1 ? (int)(blockIdx.x / (unsigned)a.ksplit) : (int)blockIdx.x;
    const int nt = tile_id % a.ntiles_total;
    const int mt = tile_id / a.ntiles_total;
    const int b = nt / a.ntiles_per_b;
    const int l0 = (nt % a.ntiles_per_b) * NT;
    const int mb0 = mt * (MW * WM) + wm * WM;  // first 32-row block of this wave

    const int nchunks = a.CiP / CK;
    const int Q = nchunks * QC;
    const int ch0 = a.ksplit > 1 ? ks * a.cps : 0;                                        // this workgroup's chunks
    const int ch1 = a.ksplit > 1 ? (ch0 + a.cps < nchunks ? ch0 + a.cps : nchunks) : nchunks;   // (never empty: launcher)
    const int Qe = ch1 * QC;

    // A (weight) stream: one float4 per lane per k-group, sequential in q.
    const f32x4 *ap[WM];
#pragma unroll
    for (int i = 0; i < WM; ++i) ap[i] = reinterpret_cast<const f32x4 *>(a.wp) + ((size_t)(mb0 + i) * Q) * 64 + lane;

    f32x16 acc[WM][NNB];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < NNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const float *inb = a.in + (size_t)b * a.in_bs;
    const int lbase = l0 * STRIDE - a.pad;

    // Register-staged prefetch of one [CK x TW] input slab.  All NLD loads of a chunk are issued
    // back to back at clamped (always valid) addresses; out-of-range samples are zeroed, and the
    // optional per-channel vector / leaky ReLU applied, only when the registers go to LDS after the
    // MFMAs of the previous chunk -- so the loads fly behind the whole chunk of matrix work.
    float stage[NLD], svec[NLD];
    unsigned long long lmask = 0;  // bit k: element k is inside the tile and its frame inside [0, Lin)
#pragma unroll
    for (int k = 0; k < NLD; ++k) {
        const int idx = tid + k * 256;
        const int row = idx / TW;
        const int l = lbase + (idx - row * TW);
        if (idx < TILE && l >= 0 && l < a.Lin) lmask |= 1ull << k;
    }
    const bool has_vec = a.in_vec != nullptr;
    const float *vecb = has_vec ? a.in_vec + (size_t)b * a.Ci : nullptr;
    // element k of the slab of `chunk`: one clamped global load (+ the per-channel vector)
    auto load_elem = [&](int chunk, int k) {
        const int idx = tid + k * 256, row = idx / TW;
        const int ci = chunk * CK + row, l = lbase + (idx - row * TW);
        const int cic = ci < a.Ci ? ci : a.Ci - 1;
        const int lcl = l < 0 ? 0 : (l >= a.Lin ? a.Lin - 1 : l);   // clamped: always a valid address
        stage[k] = inb[(size_t)cic * a.in_rs + lcl];
        if (has_vec) svec[k] = vecb[cic];
    };
    auto store_stage = [&](int buf, int chunk) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = tid + k * 256;
            const int ci = chunk * CK + idx / TW;
            float v = stage[k];
            if (has_vec) v += svec[k];
            v = v > 0.f ? v : v * a.in_slope;
            v = ((lmask >> k) & 1ull) && ci < a.Ci ? v : 0.f;
            if (idx < TILE) lds[buf][idx] = v;
        }
    };

    // Operand pipeline (all indices static, pinned with sched_barrier so hipcc cannot sink the
    // prefetches next to their use): weights AD k-groups ahead in a shifting register queue, the B
    // fragments of the next k-group read from LDS while the 4*WM*NNB MFMAs of the current one issue,
    // and the NLD slab loads of the next chunk spread evenly over this chunk's k-groups -- vmcnt
    // retires in order, so a burst of HBM-latency slab loads in front of the L2-latency weight loads
    // would stall the weight queue.
    constexpr int AD = 3;
    f32x4 aq[AD + 1][WM];
#pragma unroll
    for (int d = 0; d < AD; ++d)
#pragma unroll
        for (int i = 0; i < WM; ++i) aq[d][i] = ap[i][(size_t)(ch0 * QC + d < Qe ? ch0 * QC + d : Qe - 1) * 64];

#pragma unroll
    for (int k = 0; k < NLD; ++k) load_elem(ch0, k);
    store_stage(0, ch0);
    __syncthreads();

    int q = ch0 * QC;
    const int boff = (wn * 32 * NNB + c32) * STRIDE + h * TW;
    auto read_b = [&](const float *L, int gi, float (&bv)[4][NNB]) {
        const int tap = gi / (CK / 8), g = gi % (CK / 8);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
                bv[e][j] = L[boff + (2 * (g * 4 + e)) * TW + (DILMAX > 1 ? tap * a.dil : tap) + 32 * j * STRIDE];
    };
    for (int ch = ch0; ch < ch1; ++ch) {
        const float *L = lds[(ch - ch0) & 1];
        const int chn = ch + 1 < ch1 ? ch + 1 : ch;   // last chunk: harmless reload of itself
        float bc[4][NNB], bn[4][NNB];
        read_b(L, 0, bc);
#pragma unroll
        for (int gi = 0; gi < QC; ++gi) {
            const int qn = q + AD < Qe ? q + AD : Qe - 1;
#pragma unroll
            for (int i = 0; i < WM; ++i) aq[AD][i] = ap[i][(size_t)qn * 64];
#pragma unroll
            for (int k = gi * NLD / QC; k < (gi + 1) * NLD / QC; ++k) load_elem(chn, k);
            if (gi + 1 < QC) read_b(L, gi + 1, bn);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < WM; ++i)
#pragma unroll
                    for (int j = 0; j < NNB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[0][i][e], bc[e][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < AD; ++d)
#pragma unroll
                for (int i = 0; i < WM; ++i) aq[d][i] = aq[d + 1][i];
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int j = 0; j < NNB; ++j) bc[e][j] = bn[e][j];
            ++q;
        }
        if (ch + 1 < ch1) store_stage((ch + 1 - ch0) & 1, ch + 1);
        __syncthreads();
    }

    if (a.ksplit > 1) {   // the partial tile of this split: conv_split_finalize_kernel adds them up and runs the epilogue
        float *pp = a.part + ((size_t)tile_id * a.ksplit + ks) * (size_t)(WM * NNB * 16 * 256) + tid;
#pragma unroll
        for (int i = 0; i < WM; ++i)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) pp[((i * NNB + j) * 16 + r) * 256] = acc[i][j][r];
        return;
    }
    Epi::template run<WM, NNB>(ep, acc, b, (mt * MW + wm) * (32 * WM), l0 + wn * 32 * NNB, lane, a.Lout);
}

// One wave per 32 x 32 block of a split convolution's output: sum the ksplit partial blocks in split order (four
// splits' loads in flight at a time), then the epilogue of the unsplit kernel on that block.
template <int MW, int WM, int NNB, class Epi>
__global__ __launch_bounds__(256) void conv_split_finalize_kernel(ConvArgs a, typename Epi::Params ep)
{
    constexpr int NW = 4 / MW, NT = NW * 32 * NNB, BLK = WM * NNB;
    const int tid = threadIdx.x, lane = tid & 63;
    const long unit = (long)blockIdx.x * 4 + (tid >> 6);   // (tile, wave of the conv workgroup, block of that wave)
    const long n_units = (long)a.ntiles_total * a.mtiles * 4 * BLK;
    if (unit >= n_units) return;
    const int blk = (int)(unit % BLK), wave = (int)((unit / BLK) % 4);
    const int tile_id = (int)(unit / (4 * BLK));
    const int i = blk / NNB, j = blk - i * NNB;
    const int wm = MW == 2 ? wave >> 1 : 0, wn = MW == 2 ? wave & 1 : wave;
    const int nt = tile_id % a.ntiles_total, mt = tile_id / a.ntiles_total;
    const int b = nt / a.ntiles_per_b, l0 = (nt % a.ntiles_per_b) * NT;
    const size_t split_stride = (size_t)BLK * 16 * 256;
    const float *pt = a.part + (size_t)tile_id * a.ksplit * split_stride + (size_t)blk * 16 * 256 + wave * 64 + lane;
    f32x16 acc[1][1];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = 0.f;
    int sp = 0;
    for (; sp + 4 <= a.ksplit; sp += 4) {
        float v[4][16];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) v[u][r] = pt[(size_t)(sp + u) * split_stride + r * 256];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][0][r] += v[u][r];
    }
    for (; sp < a.ksplit; ++sp)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[0][0][r] += pt[(size_t)sp * split_stride + r * 256];
    Epi::template run<1, 1>(ep, acc, b, (mt * MW + wm) * (32 * WM) + 32 * i, l0 + wn * 32 * NNB + 32 * j, lane, a.Lout);
}

// ---------------------------------------------------------------------------------------------
// Host-side launcher.  Mrows = number of GEMM rows in the packed weight (before any pairing).
// ---------------------------------------------------------------------------------------------
struct ConvShape {
    int B, Ci, Lin, Lout, K, stride, pad, Mrows;
    long in_bs;  // 0 -> Ci*Lin
    int in_rs;   // 0 -> Lin
    int dil = 1;
    float in_slope = 1.f;
    float *scratch = nullptr;      // split-reduction scratch (mg_conv1d_fwd_split): the partial tiles
    size_t scratch_floats = 0;
};



template <int KW, int STRIDE, int CK, int DILMAX, int MW, int WM, int NNB, class Epi>
static int conv_launch_t(const ConvShape &s, const float *in, const float *in_vec, const float *wp,
                         const typename Epi::Params &ep, hipStream_t st, int ksplit = 1)
{
    ConvArgs a;
    a.in = in;
    a.in_vec = in_vec;
    a.wp = wp;
    a.in_bs = s.in_bs ? s.in_bs : (long)s.Ci * s.Lin;
    a.in_rs = s.in_rs ? s.in_rs : s.Lin;
    a.B = s.B;
    a.Ci = s.Ci;
    a.CiP = mg_round_up(s.Ci, CK);
    a.Lin = s.Lin;
    a.Lout = s.Lout;
    a.pad = s.pad;
    a.dil = s.dil;
    a.in_slope = s.in_slope;
    a.ntiles_per_b = mg_cdiv(s.Lout, (4 / MW) * 32 * NNB);
    a.ntiles_total = a.ntiles_per_b * s.B;
    const int mtiles = mg_cdiv(s.Mrows, 32 * MW * WM);
    a.mtiles = mtiles;
    a.ksplit = 1;
    a.cps = 0;
    a.part = nullptr;
    if (ksplit > 1) {
        const int nchunks = a.CiP / CK;
        const long tiles = (long)a.ntiles_total * mtiles;
        a.cps = mg_cdiv(nchunks, ksplit);
        ksplit = mg_cdiv(nchunks, a.cps);   // no empty split
        const size_t need = (size_t)tiles * ksplit * (WM * NNB * 16 * 256);
        if (ksplit > 1 && s.scratch && need <= s.scratch_floats) {
            a.ksplit = ksplit;
            a.part = s.scratch;
        }
    }
    dim3 grid((unsigned)(a.ntiles_total * mtiles * a.ksplit));
    hipLaunchKernelGGL((conv_mfma_kernel<KW, STRIDE, CK, DILMAX, MW, WM, NNB, Epi>), grid, dim3(256), 0, st, a, ep);
    MG_LAUNCH_CHECK();
    if (a.ksplit > 1) {
        const long n_units = (long)a.ntiles_total * mtiles * 4 * WM * NNB;
        hipLaunchKernelGGL((conv_split_finalize_kernel<MW, WM, NNB, Epi>), dim3((unsigned)((n_units + 3) / 4)), dim3(256), 0, st, a, ep);
        MG_LAUNCH_CHECK();
    }
    return MG_OK;
}

// Tile choice: the 128x128 workgroup tile unless that leaves most of the 256 CUs idle (short sequences,
// few output channels: the JCU tail runs at L/4 frames with 128 or 1 output channels) -- then halve the
// frame tile and/or the channel tile until there are at least 2 workgroups per CU or nothing is left to halve.
template <class Epi>
struct EpiNeedsWM2 { static constexpr bool value = false; };

template <int KW, int STRIDE, int CK, int DILMAX, class Epi>
static int conv_launch_k(const ConvShape &s, const float *in, const float *in_vec, const float *wp,
                         const typename Epi::Params &ep, hipStream_t st)
{
    const bool can_wm1 = !EpiNeedsWM2<Epi>::value;
    // the packed form of <= 64 rows holds 2 blocks (WM = 1 only); wider ones are padded to 128 rows
    const bool must_wm1 = s.Mrows <= 64;
    auto wgs = [&](int wm, int nnb) { return (long)mg_cdiv(s.Mrows, 64 * wm) * mg_cdiv(s.Lout, 64 * nnb) * s.B; };
    if (s.Mrows <= 32 && can_wm1) {   // one 32-row block: all 4 waves along the frame axis
        // 256-frame tiles unless the slab would not fit the register staging (> 32 elements per thread)
        constexpr bool big_slab = CK * (256 * STRIDE + (KW - 1) * DILMAX) > 32 * 256;
        if constexpr (!big_slab) {
            if ((long)mg_cdiv(s.Lout, 256) * s.B >= 512)
                return conv_launch_t<KW, STRIDE, CK, DILMAX, 1, 1, 2, Epi>(s, in, in_vec, wp, ep, st);
        }
        return conv_launch_t<KW, STRIDE, CK, DILMAX, 1, 1, 1, Epi>(s, in, in_vec, wp, ep, st);
    }
    int wm = must_wm1 ? 1 : 2, nnb = 2;
    // Few tiles, deep reduction, scratch given: keep the large tile (16 MFMAs per k-group and wave instead of 4) and
    // split the reduction so that about two workgroups land on every CU; each split keeps >= 2 chunks.
    if (s.scratch && wgs(wm, nnb) < 256) {
        const int nchunks = mg_round_up(s.Ci, CK) / CK;
        int ksplit = (int)((384 + wgs(wm, nnb) - 1) / wgs(wm, nnb));
        if (ksplit > nchunks / 4) ksplit = nchunks / 4;   // >= 4 chunks per split: the pipeline fill is paid per split
        if (ksplit > 8) ksplit = 8;
        if (ksplit >= 2) {
            if (wm == 2) return conv_launch_t<KW, STRIDE, CK, DILMAX, 2, 2, 2, Epi>(s, in, in_vec, wp, ep, st, ksplit);
            return conv_launch_t<KW, STRIDE, CK, DILMAX, 2, 1, 2, Epi>(s, in, in_vec, wp, ep, st, ksplit);
        }
    }
    if (wgs(wm, nnb) < 512) nnb = 1;
    if (wgs(wm, nnb) < 512 && wm == 2 && can_wm1) wm = 1;
    if (wm == 2)
        return nnb == 2 ? conv_launch_t<KW, STRIDE, CK, DILMAX, 2, 2, 2, Epi>(s, in, in_vec, wp, ep, st)
                        : conv_launch_t<KW, STRIDE, CK, DILMAX, 2, 2, 1, Epi>(s, in, in_vec, wp, ep, st);
    return nnb == 2 ? conv_launch_t<KW, STRIDE, CK, DILMAX, 2, 1, 2, Epi>(s, in, in_vec, wp, ep, st)
                    : conv_launch_t<KW, STRIDE, CK, DILMAX, 2, 1, 1, Epi>(s, in, in_vec, wp, ep, st);
}

// Dispatch on (K, stride, dilation); CK must match mg_conv_ck().  The path's own convolutions (K in
// {1,3,5,9}, dilation 1) are always instantiated; the vocoder shapes (K in {7,11,16,4}, dilations up to 5:
// hifigan/models.py:19-95,112-143) only for epilogues that opt in (EpiWide), to keep build time down.
template <class Epi>
struct EpiWide { static constexpr bool value = false; };

template <class Epi>
static int conv_launch(const ConvShape &s, const float *in, const float *in_vec, const float *wp,
                       const typename Epi::Params &ep, hipStream_t st)
{
    if (s.B <= 0 || s.Lout <= 0 || s.Ci <= 0 || s.Mrows <= 0 || s.dil < 1) return MG_ERR_SHAPE;
#define MG_CONV_CASE(KW_, ST_, CK_, DM_) \
    if (s.K == KW_ && s.stride == ST_ && s.dil <= DM_) return conv_launch_k<KW_, ST_, CK_, DM_, Epi>(s, in, in_vec, wp, ep, st);
    MG_CONV_CASE(1, 1, 32, 1)
    MG_CONV_CASE(3, 1, 32, 1)
    MG_CONV_CASE(5, 1, 16, 1)
    MG_CONV_CASE(9, 1, 16, 1)
    MG_CONV_CASE(5, 2, 16, 1)
    if constexpr (EpiWide<Epi>::value) {
        MG_CONV_CASE(3, 1, 32, 5)
        MG_CONV_CASE(7, 1, 16, 5)
        MG_CONV_CASE(11, 1, 16, 5)
        MG_CONV_CASE(16, 1, 16, 1)
        MG_CONV_CASE(4, 1, 16, 1)
    }
#undef MG_CONV_CASE
    return MG_ERR_SHAPE;
}

// number of 32-row blocks the packed form holds (padded to the workgroup M tile)
static inline int mg_conv_mblocks(int Mrows) { return Mrows > 64 ? mg_round_up(Mrows, 128) / 32 : 2; }

// Row / reduction bookkeeping of a pack: GEMM rows, reduction channels and 32-row blocks held by the packed form.
static inline int pack_dims(int Co, int Ci, int K, int mode, int *Mrows, int *Kin, int *MB)
{
    if (Co <= 0 || Ci <= 0 || !(K == 1 || K == 3 || K == 4 || K == 5 || K == 7 || K == 9 || K == 11 || K == 16))
        return MG_ERR_SHAPE;
    if ((mode == MG_PACK_PLAIN16 || mode == MG_PACK_GATE16) && K > 3) return MG_ERR_SHAPE;   // 32-channel chunks only
    if (mode == MG_PACK_PLAIN || mode == MG_PACK_PLAIN16) {
        *Mrows = Co;
        *Kin = Ci;
        *MB = mg_conv_mblocks(Co);
    } else if (mode == MG_PACK_GATE || mode == MG_PACK_GATE16) {
        if (Co % 2) return MG_ERR_SHAPE;
        *Mrows = Co;
        *Kin = Ci;
        *MB = mg_round_up(2 * mg_cdiv(Co / 2, 32), 4);
    } else if (mode == MG_PACK_DGRAD) {
        *Mrows = Ci;
        *Kin = Co;
        *MB = mg_conv_mblocks(Ci);
    } else
        return MG_ERR_ARG;
    return MG_OK;
}

// PackDesc + element count of packing [Co, Ci, K] as k-groups [q0, q0 + Q) of a stream holding Qtot per block
// (Qtot <= 0: a stand-alone pack).
static inline int mg_pack_desc(int Co, int Ci, int K, int mode, int q0, int Qtot, PackDesc *pd, size_t *total)
{
    int Mrows, Kin, MB;
    const int rc = pack_dims(Co, Ci, K, mode, &Mrows, &Kin, &MB);
    if (rc != MG_OK) return rc;
    const int CK = mg_conv_ck(K);
    const int CiP = mg_round_up(Kin, CK);
    const int Q = CiP * K / 8;
    if (Qtot <= 0) {
        q0 = 0;
        Qtot = Q;
    }
    if (q0 < 0 || q0 + Q > Qtot) return MG_ERR_SHAPE;
    *pd = PackDesc{Co, Ci, K, CK, CiP, MB, mode, q0, Qtot, 0};
    *total = (size_t)MB * Q * 256;
    return MG_OK;
}
