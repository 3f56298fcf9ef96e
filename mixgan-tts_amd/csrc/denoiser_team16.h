// The single-launch Denoiser.forward / p_sample for ONE utterance (or a handful of short ones): a 16-frame tile is
// computed by a TEAM of four (or two) workgroups, each owning 64 (128) of the 256 residual channels.
//
// Why: with a whole 256-channel tile per workgroup (denoiser_persist16.h) one 1000-frame utterance is 63 workgroups on 256
// CUs, and every one of them streams all 55 MB of weights through its own L1 and runs all 23.8 MFLOP per frame on one CU:
// 0.87 ms per reverse step, 27 TFLOP/s.  synthesize.py-style serving is exactly this shape (batch 1).  Here workgroup
// (tile, m) computes the output rows of its channels in all three GEMMs of a layer -- a quarter of the weight stream and of
// the MFMAs per CU, 252 workgroups -- and the team all-gathers what a layer's next GEMM reduces over (all 256 channels of
// h, then of the gate product g) through tagged 8-byte granules in global memory, the same "the data is the flag" hand-off
// the wider kernels use for their halo columns (cdna_hip_programming.md section 6, Guideline 16, form R2).  The k=3 halo
// rides on the same exchange: a workgroup writes the edge columns of its h rows straight into the neighbouring tiles'
// gather buffers.
//
// Per layer and workgroup: 2 publishes (its channels x 16 columns) and 2 gathers (256 x 18, 256 x 16 granules) against
// its share of a tile's MFMAs.  x and the skip sum of the workgroup's channels stay in registers for all layers (one
// 16-row block per wave); cond, h and g tiles in LDS (50 KB).  Fragments, packs and LDS layout as in
// denoiser_persist16.h (v_mfma_f32_16x16x4_f32); every GEMM reduces in the order of the wider kernels, so an utterance
// computed by teams is bit-identical to the same utterance inside a large batch.
//
// Forward progress: the workgroups of a team and the teams of neighbouring tiles wait for each other inside the launch,
// so the whole grid must be co-resident: the launcher uses this kernel only when tiles x team size <= the CU count.
// Every wait is bounded (dp_fail: sticky error word, host-visible word, NaN output), as in the wider kernels.
#pragma once
#include "denoiser_persist16.h"

// Team sizes: 4 (64 channels and 4 waves per workgroup) while tiles x 4 <= CUs, else 2 (128 channels, 8 waves) while
// tiles x 2 <= CUs -- two utterances of 1000 frames, one of 2000.  (Teams of 8 were measured slower than 4.)

// (GEMM 2 walks its reduction in the order of the wider kernels -- centre tap of every chunk, then taps 0 and 2 -- so that
// an utterance computed by teams is bit-identical to the same utterance inside a batch that runs on 32- or 64-frame tiles.)

// k loop for the teams' one- or two-block wave tiles.  A wave here issues 4 or 8 MFMAs per 16-channel step -- 128 / 256
// cycles -- so the two-steps-ahead weight prefetch of d16_mfma_loop (sized for four blocks per wave) covers a fraction of
// an L2 round trip, and with one wave per SIMD nothing else hides it: weights run DIST = 6 steps ahead in a ring of 8,
// and the first DIST steps are requested by the caller BEFORE it waits for the exchange in front of the GEMM
// (d16_preload), so the phase opens with its operands in flight.
template <int NRB, class IT, int DIST = 6>
__device__ __forceinline__ void d16_preload(f32x4 (&ring)[8][NRB], const f32x4 *const (&ap)[NRB])
{
    auto qstep = [](int t) { return 2 * ((IT::chunk(t >> 1) * IT::KW + IT::tap(t >> 1)) * 2 + (t & 1)); };
#pragma unroll
    for (int s = 0; s < DIST; ++s)
#pragma unroll
        for (int i = 0; i < NRB; ++i) ring[s][i] = ap[i][(size_t)qstep(s) * 64];
}

template <int NRB, int NTC, class IT, int DIST = 6>
__device__ __forceinline__ void d16_mfma_loop_deep(f32x4 (&acc)[NRB], const f32x4 *const (&ap)[NRB], const float *__restrict__ tile,
                                                   f32x4 (&ring)[8][NRB])
{
    constexpr int RS = 8, NS = IT::N * 2;   // steps of 16 channels
    static_assert(NS % RS == 0 && DIST < RS - 1, "whole rings; a slot is rewritten only after its MFMAs");
    auto qstep = [](int t) { return 2 * ((IT::chunk(t >> 1) * IT::KW + IT::tap(t >> 1)) * 2 + (t & 1)); };   // 8-channel groups
    auto boff = [](int t) { return ((IT::chunk(t >> 1) * 2 + (t & 1)) * NTC + IT::tap(t >> 1)) * 16; };
    f32x4 bb[2];
    bb[0] = *reinterpret_cast<const f32x4 *>(tile + boff(0));
#pragma unroll 1
    for (int t0 = 0; t0 < NS; t0 += RS) {
#pragma unroll
        for (int u = 0; u < RS; ++u) {
            const int t = t0 + u;
            const int ta = t + DIST < NS ? t + DIST : NS - 1, tb = t + 1 < NS ? t + 1 : NS - 1;   // last steps: harmless reloads
#pragma unroll
            for (int i = 0; i < NRB; ++i) ring[(u + DIST) % RS][i] = ap[i][(size_t)qstep(ta) * 64];
            bb[(u + 1) & 1] = *reinterpret_cast<const f32x4 *>(tile + boff(tb));
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < NRB; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(ring[u][i][e], bb[u & 1][e], acc[i], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Gather NCOL columns x 256 channels of tagged granules into a k-interleaved16 LDS tile (column c of the buffer ->
// column c of the tile).  Columns `skip_lo` / `skip_hi` (the halo columns of an utterance's first / last tile) are not
// waited for and read as zero.  Returns false after a timeout (the caller marks the launch failed).
template <int NCOL, int NTC, int NTHR>
__device__ __forceinline__ bool dt_gather(const dp_gu64 *buf, float *T, unsigned epoch, int tid, int skip_lo, int skip_hi,
                                          unsigned spin_limit)
{
    constexpr int PER = NCOL * 256 / NTHR;   // 256 * NCOL granules over the workgroup's threads
    static_assert(PER <= 32 && (NCOL * 256) % NTHR == 0, "one bit per granule");
    unsigned vals[PER];
    unsigned have = 0;          // bit k: granule k of this thread has arrived (or is not waited for)
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int col = (tid + NTHR * k) % NCOL;
        if (col == skip_lo || col == skip_hi) have |= 1u << k;
    }
    unsigned spins = 0;
    bool good = true;
    for (;;) {
        dp_u64 x[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k)   // only what is still missing: all loads of a round in flight together
            if (!((have >> k) & 1u)) x[k] = __hip_atomic_load(buf + tid + NTHR * k, DP_RLX_AGENT);
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (!((have >> k) & 1u) && (unsigned)(x[k] >> 32) == epoch) {
                vals[k] = (unsigned)x[k];
                have |= 1u << k;
            }
        if (__all(have == (1u << PER) - 1u)) break;
        if (++spins > spin_limit) {
            good = false;
            break;
        }
        __builtin_amdgcn_s_sleep(2);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int idx = tid + NTHR * k;
        const int row = idx / NCOL, col = idx - row * NCOL;
        const bool real = col != skip_lo && col != skip_hi && ((have >> k) & 1u);
        T[d16_at<NTC>(row, col)] = real ? __uint_as_float(vals[k]) : 0.f;
    }
    return good;
}

template <bool VEC4, int TEAM>
__global__ __launch_bounds__(1024 / TEAM, TEAM == 4 ? 1 : 2) void denoiser_team16_kernel(PersistArgs a)
{
    static_assert(TEAM == 2 || TEAM == 4, "128 or 64 channels per workgroup");
    constexpr int NWV = 16 / TEAM, NTHR = 64 * NWV;   // one 16-row block of the 256 channels per wave
    constexpr int NT = 16, NC = NT, NH = NT + 2, NG = NT;
    __shared__ __attribute__((aligned(16))) float lds[RB_C * (NC + NH + NG)];
    __shared__ unsigned s_slot, s_dead, s_launch;
    float *condT = lds;                    // col j <-> frame l0+j
    float *hT = lds + RB_C * NC;           // col j <-> frame l0-1+j
    float *gT = lds + RB_C * (NC + NH);    // col j <-> frame l0+j (g, skip sum)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int L = a.L;
    const int n_tiles = a.tiles_per_b * a.B;
    const int n_slots = n_tiles * TEAM;
    if (tid == 0) {
        s_slot = __hip_atomic_fetch_add(a.sync, 1u, DP_RLX_AGENT);   // tickets in START order
        s_launch = __hip_atomic_load(a.sync + 2, DP_RLX_AGENT);
        s_dead = 0u;
    }
    __syncthreads();
    // ticket -> (tile, member).  Consecutive workgroups go to consecutive XCDs (8 of them): the members of a tile are
    // tickets t, t+8, ... of a run of 8 * TEAM, so that a team shares one XCD when dispatch follows ticket order.
    // Speed only: any bijection is correct.
    const int slot = (int)(s_slot % (unsigned)n_slots);
    const int full = (n_tiles / 8) * (8 * TEAM);
    int tile, member;
    if (slot < full) {
        tile = (slot / (8 * TEAM)) * 8 + (slot & 7);
        member = (slot >> 3) % TEAM;
    } else {
        tile = (n_tiles / 8) * 8 + (slot - full) / TEAM;
        member = (slot - full) % TEAM;
    }
    const unsigned launch_no = s_launch;
    const int b = tile / a.tiles_per_b, jt = tile - b * a.tiles_per_b;
    const int l0 = jt * NT;
    const bool has_left = jt > 0, has_right = jt + 1 < a.tiles_per_b;
    const int f = l0 + c16;
    const bool fvalid = f < L;
    const int cb = NWV * member + w;   // this wave's 16-row block of the 256 channels
    const int ch0 = 16 * cb;
    auto row_of = [&](int r) { return ch0 + 4 * g + r; };

    // ---------------------------------------------------------------- stage the cond tile (all 256 channels) and x_t
    {
        const float *cbp = a.cond + (size_t)b * RB_C * L;
        if (a.cproj) {
            // (the conditioner enters only through its precomputed projections: no tile to stage)
        } else if (VEC4) {
#pragma unroll
            for (int k = 0; k < 1024 / NTHR; ++k) {   // 256 rows x 4 float4
                const int idx = tid + k * NTHR;
                const int row = idx >> 2, c4 = idx & 3;
                const int f0 = l0 + 4 * c4;
                const bool ok = f0 < L;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(cbp + (size_t)row * L + min(f0, L - 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) condT[d16_at<NC>(row, 4 * c4 + e)] = ok ? v[e] : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4096 / NTHR; ++k) {   // 256 rows x 16 frames
                const int idx = tid + k * NTHR;
                const int row = idx >> 4, cc = idx & 15;
                const float v = cbp[(size_t)row * L + min(l0 + cc, L - 1)];
                condT[d16_at<NC>(row, cc)] = l0 + cc < L ? v : 0.f;
            }
        }
        const float *xb = a.x_t + (size_t)b * a.M * L;
#pragma unroll
        for (int k = 0; k < 1536 / NTHR; ++k) {   // 96 rows (M = 80 padded) x 16 frames -> hT rows 0..95, col c <-> frame l0+c
            const int idx = tid + k * NTHR;
            const int row = idx >> 4, c = idx & 15;
            const float v = xb[(size_t)min(row, a.M - 1) * L + min(l0 + c, L - 1)];
            hT[d16_at<NH>(row, c)] = (row < a.M && l0 + c < L) ? v : 0.f;
        }
    }
    f32x4 XS[2];   // [0] residual stream, [1] skip sum of this wave's 16 channels
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        XS[0][r] = a.in_b[row_of(r)];
        XS[1][r] = 0.f;
    }
    f32x4 ringA[8][1], ringB[8][2];   // weight rings: GEMM 1 / skip projection, and GEMM 2 / GEMM 3
    __syncthreads();
    {   // input projection + ReLU (model/modules.py:430-431): K = 96 -> 6 steps of 16 channels
        const f32x4 *wi = reinterpret_cast<const f32x4 *>(a.in_w);   // 12 8-channel groups per 32-row block
#pragma unroll 1
        for (int s = 0; s < 6; ++s) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(hT + (s * NH + c16) * 16 + g * 4);
            const f32x4 av = wi[((size_t)(cb >> 1) * 12 + 2 * s + (cb & 1)) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) XS[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], XS[0], 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) XS[0][r] = fmaxf(XS[0][r], 0.f);
    }
    __syncthreads();   // x_t has been read out of hT: the first gather may overwrite it

    dp_gu64 *const team = (dp_gu64 *)a.team;
    const size_t h_par = (size_t)n_tiles * RB_C * NH, g_par = (size_t)n_tiles * RB_C * NG;
    dp_gu64 *const Hbuf = team;                    // [2][tiles][256][18]
    dp_gu64 *const Gbuf = team + 2 * h_par;        // [2][tiles][256][16]
    auto blk = [&](const float *base, int rb, int Q) {
        return reinterpret_cast<const f32x4 *>(base) + ((size_t)(rb >> 1) * Q + (rb & 1)) * 64 + lane;
    };
    auto tagged = [](unsigned epoch, float v) { return ((dp_u64)epoch << 32) | (dp_u64)__float_as_uint(v); };
    {   // the first layer's GEMM 1 weights
        const f32x4 *const ap0[1] = {blk(a.p16layers + a.p_wc, cb, 32)};
        d16_preload<1, D16IterK1>(ringA, ap0);
    }
    // after a timeout: keep going without waiting (the output is poisoned at the end), never hang
    auto failed = [&](unsigned code) {   // called by every lane of the wave whose wait gave up
        if (lane == 0) {
            dp_fail(a.sync, a.host_err, code);
            s_dead = 1u;
        }
    };

    for (int l = 0; l < a.NL; ++l) {
        const float *lp = a.layers + (size_t)l * a.layer_stride;        // biases live in the base layer record
        const float *pp = a.p16layers + (size_t)l * a.p16layer_stride;  // 16-row packs
        const size_t vrows = a.vec_rows ? (size_t)a.vec_rows : (size_t)a.B;
        const float *hv = a.hvec + ((size_t)l * vrows + b) * RB_C;
        const float *dv = a.dvec + ((size_t)l * vrows + b) * RB_C;
        const unsigned epoch = launch_no * ((unsigned)a.NL + 1u) + (unsigned)l + 1u;
        const int par = l & 1;

        // ------------------------------------------------------------ GEMM 1: h = Wc cond + bc + x + (Wd s [+ Wp spk])
        f32x4 acc1[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[0][r] = lp[a.l_bc + row_of(r)];   // the accumulators start at bc
        if (a.cproj) {   // (Wc cond + bc) precomputed for the whole sampling loop: see denoiser_persist.h
            const float *cp = a.cproj + ((size_t)b * a.NL + l) * RB_C * L + min(f, L - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1[0][r] = cp[(size_t)row_of(r) * L];
        } else {
            const f32x4 *const ap[1] = {blk(pp + a.p_wc, cb, 32)};
            d16_mfma_loop_deep<1, NC, D16IterK1>(acc1, ap, condT + c16 * 16 + g * 4, ringA);   // preloaded a phase ago
            if (a.cproj_out && f < L) {
                float *co = a.cproj_out + ((size_t)b * a.NL + l) * RB_C * L + f;
#pragma unroll
                for (int r = 0; r < 4; ++r) co[(size_t)row_of(r) * L] = acc1[0][r];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc1[0][r] += XS[0][r] + hv[row_of(r)];   // fl(P + fl(x + vec)) either way
        // GEMM 2's first weights: requested now, needed behind the h exchange
        const int c32 = cb >> 1, half = cb & 1;   // GATE16 packs: per 32 channels, blocks {gate lo, gate hi, filter lo, filter hi}
        const f32x4 *const ap2[2] = {blk(pp + a.p_w3, 4 * c32 + half, 96), blk(pp + a.p_w3, 4 * c32 + 2 + half, 96)};
        d16_preload<2, DpIterCentre>(ringB, ap2);
        // GEMM 2's accumulators start at the conv bias (the loads fly during the exchange): [0] gate rows, [1] filter rows
        f32x4 acc2[2];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc2[0][r] = lp[a.l_b3 + row_of(r)];
            acc2[1][r] = lp[a.l_b3 + RB_C + row_of(r)];
        }
        // ------------------------------------------------------------ publish h (zero beyond the utterance: conv padding)
        {
            dp_gu64 *mine = Hbuf + (size_t)par * h_par + (size_t)tile * RB_C * NH;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const dp_u64 v = tagged(epoch, fvalid ? acc1[0][r] : 0.f);
                __hip_atomic_store(mine + (size_t)row_of(r) * NH + 1 + c16, v, DP_RLX_AGENT);
                // (test hook DP_F_WITHHOLD: one wave of the second tile never sends its left edge)
                if (c16 == 0 && has_left && !((a.flags & DP_F_WITHHOLD) && jt == 1 && cb == 0))   // my frame l0 is the right halo column of the tile on the left
                    __hip_atomic_store(mine - (size_t)RB_C * NH + (size_t)row_of(r) * NH + NH - 1, v, DP_RLX_AGENT);
                if (c16 == NT - 1 && has_right)    // my frame l0+15 is the left halo column of the tile on the right
                    __hip_atomic_store(mine + (size_t)RB_C * NH + (size_t)row_of(r) * NH, v, DP_RLX_AGENT);
            }
        }
        // ------------------------------------------------------------ gather all 256 channels of h incl. the halo columns
        if (s_dead == 0u) {
            if (!dt_gather<NH, NH, NTHR>(Hbuf + (size_t)par * h_par + (size_t)tile * RB_C * NH, hT, epoch, tid, has_left ? -1 : 0,
                                   has_right ? -1 : NH - 1, a.spin_limit))
                failed(1u + (unsigned)l);
        }
        __syncthreads();   // hT complete

        // ------------------------------------------------------------ GEMM 2 (all three taps); gate
        d16_mfma_loop_deep<2, NH, DpIterCentre>(acc2, ap2, hT + c16 * 16 + g * 4, ringB);
        d16_preload<2, DpIterOuter>(ringB, ap2);
        d16_mfma_loop_deep<2, NH, DpIterOuter>(acc2, ap2, hT + c16 * 16 + g * 4, ringB);
        {
            dp_gu64 *mine = Gbuf + (size_t)par * g_par + (size_t)tile * RB_C * NG;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gv = mg_sigmoid(acc2[0][r]) * mg_tanh(acc2[1][r]);
                __hip_atomic_store(mine + (size_t)row_of(r) * NG + c16, tagged(epoch, gv), DP_RLX_AGENT);
            }
        }
        // GEMM 3's accumulators start as its addends: x + bo + Wd s and skip + bo (model/blocks.py:1166,1174-1176);
        // XS[0] = x rows, XS[1] = skip rows of this wave's channels: one loop, the g fragments read once for both
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            XS[0][r] += lp[a.l_bo + row_of(r)] + dv[row_of(r)];
            XS[1][r] += lp[a.l_bo + RB_C + row_of(r)];
        }
        // the first weights of GEMM 3 and of the next layer's GEMM 1 (or of the skip projection): behind the g exchange
        const f32x4 *const ap3[2] = {blk(pp + a.p_wo, cb, 32), blk(pp + a.p_wo, 16 + cb, 32)};
        d16_preload<2, D16IterK1>(ringB, ap3);
        {
            const f32x4 *const apn[1] = {l + 1 < a.NL ? blk(a.p16layers + (size_t)(l + 1) * a.p16layer_stride + a.p_wc, cb, 32)
                                                       : blk(a.skip_w, cb, 32)};
            d16_preload<1, D16IterK1>(ringA, apn);
        }
        if (s_dead == 0u) {
            if (!dt_gather<NG, NG, NTHR>(Gbuf + (size_t)par * g_par + (size_t)tile * RB_C * NG, gT, epoch, tid, -1, -1, a.spin_limit))
                failed(1u + (unsigned)l);
        }
        __syncthreads();   // gT complete (and every wave is done with hT: the next layer's gather may overwrite it)

        // ------------------------------------------------------------ GEMM 3: x rows, skip rows
        d16_mfma_loop_deep<2, NG, D16IterK1>(XS, ap3, gT + c16 * 16 + g * 4, ringB);
#pragma unroll
        for (int r = 0; r < 4; ++r) XS[0][r] *= 0.70710678118654752440f;
        // (the next write into gT is the next layer's g gather, behind that layer's hT barrier: every wave has left
        // this GEMM 3 by then)
    }

    // ---------------------------------------------------------------- tail: sum(skip)/sqrt(NL) -> skip_projection -> ReLU -> output_projection
    const unsigned epochT = launch_no * ((unsigned)a.NL + 1u) + (unsigned)a.NL + 1u;   // one tag for both tail exchanges
    const int parT = a.NL & 1;
    {
        dp_gu64 *mine = Gbuf + (size_t)parT * g_par + (size_t)tile * RB_C * NG;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __hip_atomic_store(mine + (size_t)row_of(r) * NG + c16, tagged(epochT, XS[1][r] * a.rsNL), DP_RLX_AGENT);
    }
    __syncthreads();   // last GEMM 3 done reading gT
    if (s_dead == 0u) {
        if (!dt_gather<NG, NG, NTHR>(Gbuf + (size_t)parT * g_par + (size_t)tile * RB_C * NG, gT, epochT, tid, -1, -1, a.spin_limit))
            failed(1u + (unsigned)a.NL);
    }
    __syncthreads();
    {
        f32x4 acc[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][r] = a.skip_b[row_of(r)];
        const f32x4 *const ap[1] = {blk(a.skip_w, cb, 32)};
        d16_mfma_loop_deep<1, NG, D16IterK1>(acc, ap, gT + c16 * 16 + g * 4, ringA);   // preloaded behind the last layer's g exchange
        dp_gu64 *mine = Hbuf + (size_t)parT * h_par + (size_t)tile * RB_C * NH;   // y -> the h buffer, columns 1..16
#pragma unroll
        for (int r = 0; r < 4; ++r)
            __hip_atomic_store(mine + (size_t)row_of(r) * NH + 1 + c16, tagged(epochT, fmaxf(acc[0][r], 0.f)), DP_RLX_AGENT);
    }
    if (s_dead == 0u) {   // (hT was last read in GEMM 2 of the last layer: free)
        if (!dt_gather<NH, NH, NTHR>(Hbuf + (size_t)parT * h_par + (size_t)tile * RB_C * NH, hT, epochT, tid, 0, NH - 1, a.spin_limit))
            failed(1u + (unsigned)a.NL);
    }
    __syncthreads();
    // output projection: M rows in 16-row blocks dealt over the team's 16 waves (block rb -> member rb % TEAM, wave rb / TEAM)
    const int nrb = (a.M + 15) / 16;
    const int rb = member + TEAM * w;
    if (rb < nrb) {
        f32x4 o[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rb + 4 * g + r;
            o[0][r] = row < a.M ? a.out_b[row] : 0.f;
        }
        const f32x4 *const ap[1] = {blk(a.out_w, rb, 32)};
        d16_preload<1, D16IterK1>(ringA, ap);
        d16_mfma_loop_deep<1, NH, D16IterK1>(o, ap, hT + (1 + c16) * 16 + g * 4, ringA);
        const size_t bo = (size_t)b * a.M * L;
        const bool bad = dp_failed(a.sync);   // a hand-off timed out: no tile of this launch may look like a result
        const float poison = __builtin_nanf("");
        if (!a.post) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + 4 * g + r;
                if (row < a.M && fvalid) a.out[bo + (size_t)row * L + f] = bad ? poison : o[0][r];
            }
        } else {   // p_sample tail (model/diffusion.py:113-129)
            long tb = (long)a.t[b];
            tb = tb < 0 ? 0 : (tb >= a.n_steps ? a.n_steps - 1 : tb);
            const float c1 = a.coef1[tb], c2 = a.coef2[tb];
            const float sg = tb == 0 ? 0.f : __expf(0.5f * a.logvar[tb]);
            const unsigned long long seed = a.seed, off = (a.noise_stream << 32) | (unsigned long long)launch_no;
            const int fc = min(f, L - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * rb + 4 * g + r;
                const size_t e = bo + (size_t)min(row, a.M - 1) * L + fc;
                const float xt = a.x_t[e];
                const float nz = a.noise ? a.noise[e] : dp_normal(seed, off, e);
                if (row < a.M && fvalid) {
                    float x0 = o[0][r];
                    if (a.x0_out) a.x0_out[e] = x0;
                    if (a.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
                    a.out[e] = bad ? poison : fmaf(sg, nz, fmaf(c1, x0, c2 * xt));
                }
            }
        }
    }
    if (tid == 0) {   // last workgroup out re-arms the tickets for the next launch
        const unsigned done = __hip_atomic_fetch_add(a.sync + 3, 1u, DP_RLX_AGENT);
        if (done == (unsigned)n_slots - 1u) {
            __hip_atomic_store(a.sync + 3, 0u, DP_RLX_AGENT);
            __hip_atomic_store(a.sync, 0u, DP_RLX_AGENT);
            __hip_atomic_store(a.sync + 16, 0u, DP_RLX_AGENT);
            __hip_atomic_fetch_add(a.sync + 2, 1u, DP_RLX_AGENT);
        }
    }
}
