// The single-launch Denoiser.forward / p_sample (denoiser_persist.h) at a 16-frame tile width, on
// v_mfma_f32_16x16x4_f32: for single utterances and small batches, where 32-frame tiles leave most of the chip idle
// (B=1, L=1000: 32 workgroups on 256 CUs).  Half the frames per workgroup = twice the workgroups, and the 16x16x4 MFMA
// has the same FLOP rate as the 32x32x2 one, so a workgroup's layer takes half the time; the weight stream per workgroup
// is unchanged (2.36 MB per layer), which small launches can afford.
//
// Same structure as the wider kernel: 4 waves, wave w owns channels 64w..64w+63 (four 16-row blocks) of x and of the skip
// sum in registers for all layers (f32x4 per block: row 4 (lane >> 4) + reg, column lane & 15); the conditioner tile
// stays in LDS; the edge columns of h go to the neighbouring workgroups as tagged granules while the centre tap of the
// k=3 GEMM runs; tiles are dealt by ticket.  Fragments: A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15];
// weights come from the 16-row packs (MG_PACK_PLAIN16 / MG_PACK_GATE16: conv_mfma.h), LDS tiles are k-interleaved over
// 16 channels: T[q][col][16] with channel 4e + g at position 4g + e, so a lane's B fragments of a k-group (4 k-steps of
// 4 channels) are one ds_read_b128.
#pragma once
#include "denoiser_persist.h"

__device__ __forceinline__ int d16_pos(int ch) { return ((ch & 3) << 2) | ((ch >> 2) & 3); }
template <int NTC>
__device__ __forceinline__ int d16_at(int ch, int col) { return (((ch >> 4) * NTC + col) << 4) + d16_pos(ch); }

struct D16IterK1 {   // 1x1 over 8 chunks of 32 channels
    static constexpr int N = 8, KW = 1;
    static __device__ __forceinline__ int chunk(int it) { return it; }
    static __device__ __forceinline__ int tap(int) { return 0; }
};

// k loop for NRB 16-row blocks x one 16-column block.  One step = 16 channels (4 MFMAs per block); a (chunk, tap)
// iteration of IT is two steps.  Weights DIST = 2 steps ahead in a ring of 4, B fragments one step ahead.
//   ap[i]: this lane's float4 of 16-row block i at k-group 0; step s of (chunk, tap) sits at ap[i][(2 * ((chunk * KW + tap) * 2 + s)) * 64]
//   tile:  LDS tile + (this lane's column for tap 0) * 16 + (lane >> 4) * 4;  NTC: its columns
template <int NRB, int NTC, class IT>
__device__ __forceinline__ void d16_mfma_loop(f32x4 (&acc)[NRB], const f32x4 *const (&ap)[NRB], const float *__restrict__ tile)
{
    static_assert(IT::N % 2 == 0, "the loop body covers two (chunk, tap) iterations");
    f32x4 ring[4][NRB];
    f32x4 bb[2];
    auto qstep = [](int it, int s) { return 2 * ((IT::chunk(it) * IT::KW + IT::tap(it)) * 2 + s); };   // in 8-channel groups
    auto boff = [](int it, int s) { return ((IT::chunk(it) * 2 + s) * NTC + IT::tap(it)) * 16; };
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < NRB; ++i) ring[s][i] = ap[i][(size_t)qstep(0, s) * 64];
    bb[0] = *reinterpret_cast<const f32x4 *>(tile + boff(0, 0));
#pragma unroll 1
    for (int it = 0; it < IT::N; it += 2) {
        const int itn = it + 2 < IT::N ? it + 2 : it;   // last body: harmless reload
#pragma unroll
        for (int u = 0; u < 4; ++u) {   // steps (it, 0), (it, 1), (it + 1, 0), (it + 1, 1)
            const int ia = u < 2 ? it + 1 : itn, sa = u & 1;            // two steps ahead
            const int ib = u == 0 ? it : (u == 3 ? itn : it + 1), sb = (u + 1) & 1;   // one step ahead
#pragma unroll
            for (int i = 0; i < NRB; ++i) ring[(u + 2) & 3][i] = ap[i][(size_t)qstep(ia, sa) * 64];
            bb[(u + 1) & 1] = *reinterpret_cast<const f32x4 *>(tile + boff(ib, sb));
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

// this lane's 4 registers of a 16-row block (channels ch0 + 4g .. + 3, column col) into a k-interleaved16 tile
template <int NTC, class F>
__device__ __forceinline__ void d16_store_block(float *T, int ch0, int col, int g, F val)
{
    float *base = T + ((((ch0 >> 4) * NTC) + col) << 4) + g;   // channel 4g + reg -> position 4 reg + g
#pragma unroll
    for (int r = 0; r < 4; ++r) base[4 * r] = val(r);
}

// SOLO: built for one workgroup per CU (one wave per SIMD, up to 512 registers: 298 used, nothing spilled) -- for launches of
// at most one tile per CU whose utterances' chains fit in a quarter of those slots; otherwise the two-per-CU build.
template <bool VEC4, bool SOLO = false>
__global__ __launch_bounds__(256, SOLO ? 1 : 2) void denoiser_persist16_kernel(PersistArgs a)
{
    constexpr int NT = 16, NC = NT, NH = NT + 2;
    __shared__ __attribute__((aligned(16))) float lds[RB_C * (NC + NH)];
    __shared__ unsigned s_tile, s_dead, s_launch;
    float *condT = lds;            // col j <-> frame l0+j
    float *hT = lds + RB_C * NC;   // col j <-> frame l0-1+j (h), or frame l0+j (x_t, g, skip sum)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, c16 = lane & 15;
    const int L = a.L;
    const int n_tiles = a.tiles_per_b * a.B;
    if (tid == 0) {
        s_tile = __hip_atomic_fetch_add(a.sync, 1u, DP_RLX_AGENT);   // tickets in START order
        s_launch = __hip_atomic_load(a.sync + 2, DP_RLX_AGENT);
        s_dead = 0u;
    }
    __syncthreads();
    const int tile = (int)(s_tile % (unsigned)n_tiles);
    const unsigned launch_no = s_launch;   // launches completed on this workspace: hand-off tags and the noise offset
    const int b = tile / a.tiles_per_b, jt = tile - b * a.tiles_per_b;
    const int l0 = jt * NT;
    const bool has_left = jt > 0, has_right = jt + 1 < a.tiles_per_b;
    const int f = l0 + c16;
    const bool fvalid = f < L;
    const int rbase = 64 * w;
    auto row_of = [&](int i, int r) { return rbase + 16 * i + 4 * g + r; };

    // ---------------------------------------------------------------- stage the cond tile (once) and the x_t tile
    {
        const float *cb = a.cond + (size_t)b * RB_C * L;
        if (a.cproj) {
            // (the conditioner enters only through its precomputed projections: no tile to stage)
        } else if (VEC4) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {   // 256 rows x 4 float4
                const int idx = tid + k * 256;
                const int row = idx >> 2, c4 = idx & 3;
                const int f0 = l0 + 4 * c4;
                const bool ok = f0 < L;
                const f32x4 v = *reinterpret_cast<const f32x4 *>(cb + (size_t)row * L + min(f0, L - 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) condT[d16_at<NC>(row, 4 * c4 + e)] = ok ? v[e] : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) {   // 256 rows x 16 frames
                const int idx = tid + k * 256;
                const int row = idx >> 4, cc = idx & 15;
                const float v = cb[(size_t)row * L + min(l0 + cc, L - 1)];
                condT[d16_at<NC>(row, cc)] = l0 + cc < L ? v : 0.f;
            }
        }
        const float *xb = a.x_t + (size_t)b * a.M * L;
#pragma unroll
        for (int k = 0; k < 6; ++k) {   // 96 rows (M = 80 padded) x 16 frames
            const int idx = tid + k * 256;
            const int row = idx >> 4, c = idx & 15;
            const float v = xb[(size_t)min(row, a.M - 1) * L + min(l0 + c, L - 1)];
            hT[d16_at<NH>(row, c)] = (row < a.M && l0 + c < L) ? v : 0.f;
        }
    }

    f32x4 X[4], S[4];   // residual stream and skip sum: 16-row blocks of this wave's 64 channels
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            X[i][r] = a.in_b[row_of(i, r)];
            S[i][r] = 0.f;
        }
    __syncthreads();
    {   // input projection + ReLU: K = 96 -> 6 steps of 16 channels, not worth a pipeline
        const f32x4 *wi = reinterpret_cast<const f32x4 *>(a.in_w);   // 12 8-channel groups per 32-row block
#pragma unroll 1
        for (int s = 0; s < 6; ++s) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(hT + (s * NH + c16) * 16 + g * 4);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 av = wi[((size_t)(2 * w + (i >> 1)) * 12 + 2 * s + (i & 1)) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; ++e) X[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], X[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[i][r] = fmaxf(X[i][r], 0.f);
    }

    dp_gu64 *const gran = (dp_gu64 *)a.gran;
    // 16-row block rb (of a 512- or 256-row matrix) at 8-channel group 0: container block rb >> 1, sub-block rb & 1
    auto blk = [&](const float *base, int rb, int Q) {
        return reinterpret_cast<const f32x4 *>(base) + ((size_t)(rb >> 1) * Q + (rb & 1)) * 64 + lane;
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
        f32x4 acc1[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1[i][r] = lp[a.l_bc + row_of(i, r)];   // the accumulators start at bc
        if (a.cproj) {   // (Wc cond + bc) precomputed for the whole sampling loop: see denoiser_persist.h
            const float *cp = a.cproj + ((size_t)b * a.NL + l) * RB_C * L + min(f, L - 1);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc1[i][r] = cp[(size_t)row_of(i, r) * L];
        } else {
            const f32x4 *const ap[4] = {blk(pp + a.p_wc, 4 * w, 32), blk(pp + a.p_wc, 4 * w + 1, 32),
                                        blk(pp + a.p_wc, 4 * w + 2, 32), blk(pp + a.p_wc, 4 * w + 3, 32)};
            d16_mfma_loop<4, NC, D16IterK1>(acc1, ap, condT + c16 * 16 + g * 4);
            if (a.cproj_out && f < L) {
                float *co = a.cproj_out + ((size_t)b * a.NL + l) * RB_C * L + f;
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) co[(size_t)row_of(i, r) * L] = acc1[i][r];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc1[i][r] += X[i][r] + hv[row_of(i, r)];   // fl(P + fl(x + vec)) either way
        // GEMM 2's accumulators start at the conv bias.  Pass p covers channels 64w + 32p .. +31:
        // acc2[p][0..1] = gate rows (two 16-row blocks), acc2[p][2..3] = filter rows
        f32x4 acc2[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = rbase + 32 * p + 16 * i + 4 * g + r;
                    acc2[p][i][r] = lp[a.l_b3 + ch];
                    acc2[p][2 + i][r] = lp[a.l_b3 + RB_C + ch];
                }
        __syncthreads();   // every wave is past the previous layer's GEMM 3 (or the head GEMM): hT may be rewritten
#pragma unroll
        for (int i = 0; i < 4; ++i)
            d16_store_block<NH>(hT, rbase + 16 * i, 1 + c16, g, [&](int r) { return fvalid ? acc1[i][r] : 0.f; });
        __syncthreads();   // interior columns of hT complete

        if (w < 2) {   // hand the edge columns to the neighbours
            bool go = w == 0 ? has_left : has_right;
            if ((a.flags & DP_F_WITHHOLD) && w == 0 && jt == 1) go = false;
            if (go) {
                const int dst_tile = w == 0 ? tile - 1 : tile + 1;
                const int col = w == 0 ? 1 : NT;
                dp_gu64 *gq = gran + (((size_t)par * n_tiles + dst_tile) * 2 + (w == 0 ? 1 : 0)) * RB_C;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = lane + 64 * k;
                    const dp_u64 v = ((dp_u64)epoch << 32) | (dp_u64)__float_as_uint(hT[d16_at<NH>(row, col)]);
                    __hip_atomic_store(gq + row, v, DP_RLX_AGENT);
                }
            }
        }

        // ------------------------------------------------------------ GEMM 2, centre tap
        const f32x4 *ap2[2][4];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 4; ++i) ap2[p][i] = blk(pp + a.p_w3, 4 * (2 * w + p) + i, 96);
#pragma unroll
        for (int p = 0; p < 2; ++p) d16_mfma_loop<4, NH, DpIterCentre>(acc2[p], ap2[p], hT + c16 * 16 + g * 4);

        if (w < 2) {   // receive the halo columns
            const bool from = w == 0 ? has_left : has_right;
            const int col = w == 0 ? 0 : NT + 1;
            unsigned v[4] = {0u, 0u, 0u, 0u};
            if (from && s_dead == 0u) {
                dp_gu64 *gq = gran + (((size_t)par * n_tiles + tile) * 2 + w) * RB_C;
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const dp_u64 x = __hip_atomic_load(gq + lane + 64 * k, DP_RLX_AGENT);
                        v[k] = (unsigned)x;
                        ok &= (unsigned)(x >> 32) == epoch;
                    }
                    if (__all(ok)) break;
                    if (++spins > a.spin_limit) {
                        if (lane == 0) {
                            dp_fail(a.sync, a.host_err, 1u + (unsigned)l);
                            s_dead = 1u;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(8);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) hT[d16_at<NH>(lane + 64 * k, col)] = from ? __uint_as_float(v[k]) : 0.f;
        }
        __syncthreads();   // halo columns in place

        // ------------------------------------------------------------ GEMM 2, taps 0 and 2; gate
#pragma unroll
        for (int p = 0; p < 2; ++p) d16_mfma_loop<4, NH, DpIterOuter>(acc2[p], ap2[p], hT + c16 * 16 + g * 4);
        __syncthreads();   // every wave has read hT for the last time: g may overwrite it
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                d16_store_block<NH>(hT, rbase + 32 * p + 16 * i, c16, g,
                                    [&](int r) { return mg_sigmoid(acc2[p][i][r]) * mg_tanh(acc2[p][2 + i][r]); });
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row_of(i, r);
                X[i][r] += lp[a.l_bo + row] + dv[row];
                S[i][r] += lp[a.l_bo + RB_C + row];
            }
        __syncthreads();   // g complete

        // ------------------------------------------------------------ GEMM 3: x rows, then skip rows
        {
            const f32x4 *const apx[4] = {blk(pp + a.p_wo, 4 * w, 32), blk(pp + a.p_wo, 4 * w + 1, 32),
                                         blk(pp + a.p_wo, 4 * w + 2, 32), blk(pp + a.p_wo, 4 * w + 3, 32)};
            const f32x4 *const aps[4] = {blk(pp + a.p_wo, 16 + 4 * w, 32), blk(pp + a.p_wo, 17 + 4 * w, 32),
                                         blk(pp + a.p_wo, 18 + 4 * w, 32), blk(pp + a.p_wo, 19 + 4 * w, 32)};
            d16_mfma_loop<4, NH, D16IterK1>(X, apx, hT + c16 * 16 + g * 4);
            d16_mfma_loop<4, NH, D16IterK1>(S, aps, hT + c16 * 16 + g * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) X[i][r] *= 0.70710678118654752440f;
    }

    // ---------------------------------------------------------------- tail
    __syncthreads();   // last GEMM 3 done reading hT
#pragma unroll
    for (int i = 0; i < 4; ++i) d16_store_block<NH>(hT, rbase + 16 * i, c16, g, [&](int r) { return S[i][r] * a.rsNL; });
    __syncthreads();
    {
        f32x4 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][r] = a.skip_b[row_of(i, r)];
        const f32x4 *const ap[4] = {blk(a.skip_w, 4 * w, 32), blk(a.skip_w, 4 * w + 1, 32), blk(a.skip_w, 4 * w + 2, 32),
                                    blk(a.skip_w, 4 * w + 3, 32)};
        d16_mfma_loop<4, NH, D16IterK1>(acc, ap, hT + c16 * 16 + g * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            d16_store_block<NC>(condT, rbase + 16 * i, c16, g, [&](int r) { return fmaxf(acc[i][r], 0.f); });   // cond is dead now
    }
    __syncthreads();
    const int nrb = (a.M + 15) / 16;   // output projection: M rows in 16-row blocks, dealt over the waves
    for (int rb = w; rb < nrb; rb += 4) {
        f32x4 o[1];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * rb + 4 * g + r;
            o[0][r] = row < a.M ? a.out_b[row] : 0.f;
        }
        const f32x4 *const ap[1] = {blk(a.out_w, rb, 32)};
        d16_mfma_loop<1, NC, D16IterK1>(o, ap, condT + c16 * 16 + g * 4);
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
    if (tid == 0) {
        const unsigned done = __hip_atomic_fetch_add(a.sync + 3, 1u, DP_RLX_AGENT);
        if (done == (unsigned)n_tiles - 1u) {
            __hip_atomic_store(a.sync + 3, 0u, DP_RLX_AGENT);
            __hip_atomic_store(a.sync, 0u, DP_RLX_AGENT);
            __hip_atomic_store(a.sync + 16, 0u, DP_RLX_AGENT);
            __hip_atomic_fetch_add(a.sync + 2, 1u, DP_RLX_AGENT);
        }
    }
}
