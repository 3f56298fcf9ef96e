// Weight gradient of a stride-1 "same" Conv1d (k = 1 or 3), streaming form, fp32 MFMA (v_mfma_f32_32x32x2_f32), gfx950.
//
//   dW[g][co, ci, k] = alpha * sum_{b, l} dY_g[b, co, l] * X_g[b, ci, l + k - pad]
//
// What wgrad_mfma_kernel (wgrad_mfma.h) leaves on the table for the big gradients of the residual stack:
//   * one workgroup per tap: the three taps of the k=3 conv stage the same dY and X tiles three times;
//   * single-buffered tiles with two barriers per 64-frame chunk and a staging phase nothing overlaps;
//   * whole tiles per workgroup: 160 (tile, layer) pairs on 256 CUs leave the grid either under-filled or split
//     into rounds.
// Here one 8-wave workgroup per CU walks a contiguous run of (tile, frame-chunk) UNITS -- every workgroup gets the
// same number of units, whatever the tile count -- with all taps of a tile accumulated from ONE staged pair of
// tiles (the X tile holds the aligned window that covers the three shifts; a tap is an address offset at read
// time).  Tiles are double-buffered in LDS with one barrier per unit; the staging of unit u+1 is spread over the
// k-steps of unit u (one global load per k-step at the front of the loop, one element to the other buffer per k-step in
// its second half).  When the run crosses into the next tile the accumulators are flushed to a partial tile; the
// finalize kernel adds the (at most `maxseg` per workgroup) partials of a tile in workgroup order -- a fixed summation
// order, like the split kernel -- and transposes to [Co][Ci][K].  The row sums of dY (bias gradients) are collected
// from the staging registers on the way (rs_part).  Measured: tools/ubench/wgrad_stream_bench.hip,
// profiles/r02_e_wgrad_stream_ubench.txt (0.71 of the nominal fp32 MFMA peak on real data -- the clock, not the
// schedule, is what gives: the same cycle count runs 15 % faster on all-zero operands).
#pragma once
#include "common.h"

#define WS_NW 256   // workgroups = CUs of one MI355X; fewer when there are fewer units

struct WgradStreamArgs {
    const float *dy;   // [G][B][Co][L]  (group stride dy_gs, batch stride dy_bs, row stride L)
    const float *x;    // [G][B][Ci][L]
    float *part;       // [nw][maxseg][K][TM][TN] partial tiles
    long dy_bs, x_bs, dy_gs, x_gs;
    int L;
    int co_tiles, ci_tiles, chunks_per_b, nchunks;
    unsigned units;    // tiles * nchunks
    int maxseg;
    float *rs_part;    // optional [nw][maxseg][TM]: partial row sums of dY (bias gradients) from the ci-tile-0 tiles
#ifdef WS_TIMING
    long long *dbg;    // [nw][8 waves][4]: cycles in (load issue, MFMA loop, LDS store + flush, barrier); tools/ubench only
#endif
};

#ifdef WS_TIMING
#define WS_T(k) do { const long long t__ = clock64(); tacc[k] += t__ - tlast; tlast = t__; } while (0)
#else
#define WS_T(k) do { } while (0)
#endif

static __device__ __forceinline__ unsigned ws_start(unsigned w, unsigned units, unsigned nw)
{
    return (unsigned)(((unsigned long long)w * units) / nw);
}

// K taps; FT frames per chunk; TM x TN tile of (co, ci); the 8 waves are WM (co) x 8/WM (ci), a wave owns
// NI x NJ blocks of 32 x 32 for each tap.
template <int K, int FT, int TM, int TN, int WM, int NI, int NJ>
__global__ __launch_bounds__(512) void wgrad_stream_kernel(WgradStreamArgs a)
{
    constexpr int WN = 8 / WM;
    static_assert(WM * NI * 32 == TM && WN * NJ * 32 == TN, "wave layout does not cover the tile");
    constexpr int W0 = K == 3 ? -4 : 0;          // first frame of the X window relative to the chunk (aligned)
    constexpr int BW = FT + (K == 3 ? 8 : 0);    // window frames
    constexpr int RSA = FT + 1, RSB = BW + 1;    // odd strides: a fragment read (32 rows, one column) hits 32 banks
    constexpr int BUF = TM * RSA + TN * RSB;
    constexpr int A4 = FT / 4, B4 = BW / 4;
    constexpr int NA = TM * A4 / 512, NB = (TN * B4 + 511) / 512;
    static_assert(TM * A4 % 512 == 0, "A tile must divide over the threads");
    __shared__ float lds[2 * BUF];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int hh = lane >> 5, c32 = lane & 31;
    const unsigned nw = gridDim.x;
    const unsigned u0 = ws_start(blockIdx.x, a.units, nw), u1 = ws_start(blockIdx.x + 1, a.units, nw);
    if (u0 >= u1) return;

    f32x16 acc[NI][NJ][K];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < NI; ++i)
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < K; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][t][r] = 0.f;
    };
    zero_acc();

    // staging elements of this thread, relative to the tile's first row: fixed for the whole run.  A: element k is
    // row tid / A4 + k * (512 / A4), float4 column tid % A4 (the same for every k)
    static_assert(512 % A4 == 0, "A rows per pass");
    constexpr int RA = 512 / A4;
    const int c4A = tid % A4;
    const int gA0 = (tid / A4) * a.L, lA0 = (tid / A4) * RSA + 4 * c4A;
    int gB[NB], lB[NB], fB[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        const int idx = tid + k * 512, row = min(idx / B4, TN - 1), c4 = idx - (idx / B4) * B4;
        gB[k] = row * a.L;
        lB[k] = TM * RSA + row * RSB + 4 * c4;
        fB[k] = idx < TN * B4 ? W0 + 4 * c4 : (1 << 28);   // past the tile: never inside [0, L)
    }
    f32x4 va[NA], vb[NB];
    unsigned okA = 0, okB = 0;   // which staged elements are inside [0, L): applied when the registers go to LDS
    const int tiles_per_g = a.co_tiles * a.ci_tiles;
    // Bias gradients ride along: every dY element passes through this thread's registers on its way to LDS, so the
    // row sums of dY over (batch, frames) cost four adds per staged float4.  Tiles with ci-tile 0 count (the others
    // stage the same rows again); the sums follow the STAGED stream, which runs one unit ahead of the MFMAs.
    float rs[NA];
#pragma unroll
    for (int k = 0; k < NA; ++k) rs[k] = 0.f;
    bool rs_on = false;   // the staged tile contributes row sums
    int rs_seg = 0;       // segment (tile index within this workgroup's run) of the staged tile
    auto rs_flush = [&]() {
        if (!a.rs_part) return;
        float *dst = a.rs_part + ((size_t)blockIdx.x * a.maxseg + rs_seg) * TM;
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            float v = rs[k];
#pragma unroll
            for (int o = A4 / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);   // the A4 lanes that share a row
            if (c4A == 0) dst[tid / A4 + k * RA] = rs_on ? v : 0.f;
            rs[k] = 0.f;
        }
    };

    // The staging of unit u+1 is spread over the k-steps of unit u: ONE global load per k-step at the front of the
    // loop (a burst of all of them right after the barrier, from all eight waves at once, backs up the memory pipe and
    // the waves sit at the issue -- 20 % of the kernel when measured), ONE element to LDS per k-step in its second half.
    const float *dyt = nullptr, *xt = nullptr;   // rows of the tile being staged (batch item 0)
    const float *dyb = nullptr, *xb = nullptr;   // ... of the batch item being staged
    int f0n = 0;
    auto set_tile = [&](int tile) {   // wave-uniform; the divisions run once per tile, not once per unit
        const int g = tile / tiles_per_g, r = tile - g * tiles_per_g;
        const int cot = r / a.ci_tiles, cit = r - cot * a.ci_tiles;
        rs_on = cit == 0;
        dyt = a.dy + (size_t)g * a.dy_gs + (size_t)cot * TM * a.L;
        xt = a.x + (size_t)g * a.x_gs + (size_t)cit * TN * a.L;
    };
    auto set_chunk = [&](int b, int c) {
        f0n = c * FT;
        dyb = dyt + (size_t)b * a.dy_bs + gA0;
        xb = xt + (size_t)b * a.x_bs;
        okA = f0n + 4 * c4A < a.L ? 1u : 0u;
        okB = 0;
    };
    auto load_elem = [&](int e) {   // e: compile-time after unrolling
        if (e < NA) {
            va[e] = *reinterpret_cast<const f32x4 *>(dyb + min(f0n + 4 * c4A, a.L - 4) + (size_t)e * RA * a.L);
        } else {
            const int k = e - NA, f = f0n + fB[k];
            vb[k] = *reinterpret_cast<const f32x4 *>(xb + gB[k] + min(max(f, 0), a.L - 4));
            if (f >= 0 && f < a.L) okB |= 1u << k;
        }
    };
    auto store_elem = [&](float *buf, int e, bool count) {
        if (e < NA) {
#pragma unroll
            for (int j = 0; j < 4; ++j) buf[lA0 + e * RA * RSA + j] = okA ? va[e][j] : 0.f;
            if (count && okA) rs[e] += (va[e][0] + va[e][1]) + (va[e][2] + va[e][3]);
        } else {
            const int k = e - NA;
            if (k + 1 < NB || TN * B4 % 512 == 0 || tid + k * 512 < TN * B4) {
                const bool ok = (okB >> k) & 1u;
#pragma unroll
                for (int j = 0; j < 4; ++j) buf[lB[k] + j] = ok ? vb[k][j] : 0.f;
            }
        }
    };
    auto flush = [&](int seg) {
        // acc[i][j][t][r]: co = wm*NI*32 + i*32 + 8*(r>>2) + 4*hh + (r&3),  ci = wn*NJ*32 + j*32 + c32
        float *dst = a.part + ((size_t)blockIdx.x * a.maxseg + seg) * K * TM * TN;
#pragma unroll
        for (int t = 0; t < K; ++t)
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int co = wm * NI * 32 + i * 32 + 8 * (r >> 2) + 4 * hh + (r & 3);
#pragma unroll
                    for (int j = 0; j < NJ; ++j)
                        dst[((size_t)t * TM + co) * TN + wn * NJ * 32 + j * 32 + c32] = acc[i][j][t][r];
                }
    };

    constexpr int NS = FT / 2, NE = NA + NB;
    static_assert(NE <= NS / 2, "one staging element per k-step");
    int tile = (int)(u0 / (unsigned)a.nchunks), chunk = (int)(u0 - (unsigned)tile * a.nchunks), seg = 0, p = 0;
    int sb = chunk / a.chunks_per_b, sc = chunk - sb * a.chunks_per_b;   // batch item, chunk inside it: what is staged
    set_tile(tile);
    set_chunk(sb, sc);
#pragma unroll
    for (int e = 0; e < NE; ++e) load_elem(e);
#pragma unroll
    for (int e = 0; e < NE; ++e) store_elem(lds, e, true);
    __syncthreads();
    const int offA = (wm * NI * 32 + c32) * RSA + hh;
    const int offB = TM * RSA + (wn * NJ * 32 + c32) * RSB + hh + (K == 3 ? 3 : 0);
#ifdef WS_TIMING
    long long tacc[4] = {0, 0, 0, 0}, tlast = clock64();
#endif
    for (unsigned u = u0; u < u1; ++u) {
        const bool more = u + 1 < u1;
        int ntile = tile, nchunk = chunk + 1;
        if (nchunk == a.nchunks) {
            nchunk = 0;
            ++ntile;
        }
        // the last unit of the run stages its predecessor's successor again, i.e. itself (into the buffer nobody
        // reads): no branches in the loop
        if (more) {
            if (++sc == a.chunks_per_b) {
                sc = 0;
                if (++sb == a.nchunks / a.chunks_per_b) {
                    sb = 0;
                    rs_flush();   // every unit of the previous tile has been staged
                    ++rs_seg;
                    set_tile(ntile);
                }
            }
        }
        set_chunk(sb, sc);
        WS_T(0);
        const float *A = lds + p * BUF + offA;
        const float *Bt = lds + p * BUF + offB;
        float *nbuf = lds + (p ^ 1) * BUF;   // last read two units ago; every wave has passed a barrier since
        // fragments of k-step s+1 are read while the MFMAs of k-step s run
        float av[2][NI], bv[2][NJ][K];
        auto frags = [&](int s, int q) {
#pragma unroll
            for (int i = 0; i < NI; ++i) av[q][i] = A[i * 32 * RSA + 2 * s];
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int t = 0; t < K; ++t) bv[q][j][t] = Bt[j * 32 * RSB + 2 * s + t];
        };
        frags(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int q = s & 1;
            if (s + 1 < NS) frags(s + 1, q ^ 1);
            if (s < NE) load_elem(s);
            if (s >= NS / 2 && s - NS / 2 < NE) store_elem(nbuf, s - NS / 2, more);   // !more: a dummy restage
#pragma unroll
            for (int t = 0; t < K; ++t)
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int i = 0; i < NI; ++i)
                        acc[i][j][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q][i], bv[q][j][t], acc[i][j][t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        WS_T(1);
        if (!more || ntile != tile) {
            flush(seg);
            ++seg;
            zero_acc();
        }
        WS_T(2);
        __syncthreads();
        WS_T(3);
        p ^= 1;
        tile = ntile;
        chunk = nchunk;
    }
    rs_flush();
#ifdef WS_TIMING
    if (lane == 0)
        for (int k = 0; k < 4; ++k) a.dbg[((size_t)blockIdx.x * 8 + wave) * 4 + k] = tacc[k];
#endif
}

// partial tiles -> dw[g] [Co][Ci][K] (= or +=), scaled.  One block per (tile, tap, 16-row slab); the workgroups that
// touched the tile follow from the unit arithmetic of the kernel above and are added in index order.
template <int K, int TM, int TN>
__global__ __launch_bounds__(256) void wgrad_stream_finalize_kernel(const float *__restrict__ part, float *__restrict__ dw,
                                                                    int Ci, int co_tiles, int ci_tiles, int nchunks,
                                                                    unsigned units, unsigned nw, int maxseg, long dw_gs,
                                                                    float alpha, int accumulate)
{
    constexpr int SL = TM / 16, NE = 16 * TN / 256;
    int bid = blockIdx.x;
    const int slab = bid % SL;
    bid /= SL;
    const int k = bid % K, tile = bid / K;
    const int tiles_per_g = co_tiles * ci_tiles;
    const int g = tile / tiles_per_g, r = tile - g * tiles_per_g;
    const int cot = r / ci_tiles, cit = r - cot * ci_tiles;
    const unsigned a0 = (unsigned)tile * nchunks, a1 = a0 + nchunks;
    unsigned w = (unsigned)(((unsigned long long)a0 * nw) / units);
    while (w + 1 < nw && ws_start(w + 1, units, nw) <= a0) ++w;
    float v[NE];
#pragma unroll
    for (int e = 0; e < NE; ++e) v[e] = 0.f;
    for (; w < nw; ++w) {
        const unsigned s0 = ws_start(w, units, nw);
        if (s0 >= a1) break;
        const int seg = tile - (int)(s0 / (unsigned)nchunks);
        const float *p = part + (((size_t)w * maxseg + seg) * K + k) * TM * TN + (size_t)slab * 16 * TN;
#pragma unroll
        for (int e = 0; e < NE; ++e) v[e] += p[threadIdx.x + e * 256];
    }
#pragma unroll
    for (int e = 0; e < NE; ++e) {
        const int idx = threadIdx.x + e * 256;
        const int co = cot * TM + slab * 16 + idx / TN, ci = cit * TN + idx % TN;
        float *o = dw + (size_t)g * dw_gs + ((size_t)co * Ci + ci) * K + k;
        const float val = v[e] * alpha;
        *o = accumulate ? *o + val : val;
    }
}

#ifdef WS_TIMING
static long long *ws_timing_buffer = nullptr;
#endif

// partial row sums -> db[g][co] (= or +=), scaled: one thread per output row, the contributing workgroups (those whose
// run touched the row's ci-tile-0 tile) added in index order
template <int TM>
__global__ __launch_bounds__(256) void wgrad_stream_rowsum_finalize_kernel(const float *__restrict__ rs_part,
                                                                           float *__restrict__ db, int Co, int G,
                                                                           int co_tiles, int ci_tiles, int nchunks,
                                                                           unsigned units, unsigned nw, int maxseg,
                                                                           long db_gs, float alpha, int accumulate)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= G * Co) return;
    const int g = i / Co, co = i - g * Co;
    const int tile = (g * co_tiles + co / TM) * ci_tiles;
    const unsigned a0 = (unsigned)tile * nchunks, a1 = a0 + nchunks;
    unsigned w = (unsigned)(((unsigned long long)a0 * nw) / units);
    while (w + 1 < nw && ws_start(w + 1, units, nw) <= a0) ++w;
    float v = 0.f;
    for (; w < nw; ++w) {
        const unsigned s0 = ws_start(w, units, nw);
        if (s0 >= a1) break;
        const int seg = tile - (int)(s0 / (unsigned)nchunks);
        v += rs_part[((size_t)w * maxseg + seg) * TM + co % TM];
    }
    float *o = db + (size_t)g * db_gs + co;
    *o = accumulate ? *o + alpha * v : alpha * v;
}

// Tile shapes.  k=3: 128 x 128 x 3 taps, 64-frame chunks (wave: 64 co x 32 ci x 3 taps = 6 accumulators).
// 1x1: 128 x 256, 32-frame chunks (wave 64 x 64); when both channel counts are multiples of 256, 256 x 256 (wave
// 64 x 128: 8 accumulators, 6 fragment reads per 8 MFMAs instead of 4 per 4, units twice as long per barrier).
struct WsCfgK3 { static constexpr int K = 3, FT = 64, TM = 128, TN = 128, WM = 2, NI = 2, NJ = 1; };
struct WsCfgK1 { static constexpr int K = 1, FT = 32, TM = 128, TN = 256, WM = 2, NI = 2, NJ = 2; };
struct WsCfgK1Wide { static constexpr int K = 1, FT = 32, TM = 256, TN = 256, WM = 4, NI = 2, NJ = 4; };

struct WsShape {
    int cfg;           // 0: not a streaming shape, 1: k=3, 2: 1x1, 3: 1x1 wide
    int K, FT, TM, TN;
};
static inline WsShape wgrad_stream_shape(int Co, int Ci, int K)
{
    if (K == 3 && Co % WsCfgK3::TM == 0 && Ci % WsCfgK3::TN == 0) return {1, 3, WsCfgK3::FT, WsCfgK3::TM, WsCfgK3::TN};
    if (K == 1 && Co % WsCfgK1Wide::TM == 0 && Ci % WsCfgK1Wide::TN == 0)
        return {3, 1, WsCfgK1Wide::FT, WsCfgK1Wide::TM, WsCfgK1Wide::TN};
    if (K == 1 && Co % WsCfgK1::TM == 0 && Ci % WsCfgK1::TN == 0) return {2, 1, WsCfgK1::FT, WsCfgK1::TM, WsCfgK1::TN};
    return {0, K, 0, 0, 0};
}
static inline int wgrad_stream_tiles(const WsShape &w, int Co, int Ci, int G) { return w.cfg ? G * (Co / w.TM) * (Ci / w.TN) : 0; }
// a run of ceil(units / nw) units touches at most tiles / nw + 2 tiles
static inline int wgrad_stream_maxseg(int tiles) { return tiles / WS_NW + 2; }
static inline size_t wgrad_stream_scratch_floats(int Co, int Ci, int K, int G)
{
    const WsShape w = wgrad_stream_shape(Co, Ci, K);
    if (!w.cfg) return 0;
    // partial tiles + partial row sums (the optional bias gradients)
    return (size_t)WS_NW * wgrad_stream_maxseg(wgrad_stream_tiles(w, Co, Ci, G)) * ((size_t)w.K * w.TM * w.TN + w.TM);
}

template <class C>
static int wgrad_stream_launch_k(const float *dy, const float *x, float *dw, float *scratch, int G, int B, int Co, int Ci,
                                 int L, long dy_bs, long x_bs, long dy_gs, long x_gs, long dw_gs, float alpha,
                                 int accumulate, hipStream_t st, float *db = nullptr, long db_gs = 0)
{
    constexpr int K = C::K;
    WgradStreamArgs a;
    a.dy = dy;
    a.x = x;
    a.part = scratch;
    a.dy_bs = dy_bs;
    a.x_bs = x_bs;
    a.dy_gs = dy_gs;
    a.x_gs = x_gs;
    a.L = L;
    a.co_tiles = Co / C::TM;
    a.ci_tiles = Ci / C::TN;
    a.chunks_per_b = mg_cdiv(L, C::FT);
    a.nchunks = a.chunks_per_b * B;
    const int tiles = G * a.co_tiles * a.ci_tiles;
    a.units = (unsigned)tiles * (unsigned)a.nchunks;
    a.maxseg = wgrad_stream_maxseg(tiles);
    const unsigned nw = a.units < WS_NW ? a.units : WS_NW;
    a.rs_part = db ? scratch + (size_t)WS_NW * a.maxseg * K * C::TM * C::TN : nullptr;
#ifdef WS_TIMING
    a.dbg = ws_timing_buffer;
#endif
    hipLaunchKernelGGL((wgrad_stream_kernel<K, C::FT, C::TM, C::TN, C::WM, C::NI, C::NJ>), dim3(nw), dim3(512), 0, st, a);
    MG_LAUNCH_CHECK();
    hipLaunchKernelGGL((wgrad_stream_finalize_kernel<K, C::TM, C::TN>), dim3((unsigned)tiles * K * (C::TM / 16)), dim3(256),
                       0, st, scratch, dw, Ci, a.co_tiles, a.ci_tiles, a.nchunks, a.units, nw, a.maxseg, dw_gs, alpha,
                       accumulate);
    MG_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL((wgrad_stream_rowsum_finalize_kernel<C::TM>), dim3((unsigned)mg_cdiv(G * Co, 256)), dim3(256), 0,
                           st, a.rs_part, db, Co, G, a.co_tiles, a.ci_tiles, a.nchunks, a.units, nw, a.maxseg,
                           db_gs ? db_gs : (long)Co, alpha, accumulate);
        MG_LAUNCH_CHECK();
    }
    return MG_OK;
}

static int wgrad_stream_launch(const WsShape &w, const float *dy, const float *x, float *dw, float *scratch, int G, int B,
                               int Co, int Ci, int L, long dy_bs, long x_bs, long dy_gs, long x_gs, long dw_gs, float alpha,
                               int accumulate, hipStream_t st, float *db = nullptr, long db_gs = 0)
{
    switch (w.cfg) {
    case 1: return wgrad_stream_launch_k<WsCfgK3>(dy, x, dw, scratch, G, B, Co, Ci, L, dy_bs, x_bs, dy_gs, x_gs, dw_gs, alpha, accumulate, st, db, db_gs);
    case 2: return wgrad_stream_launch_k<WsCfgK1>(dy, x, dw, scratch, G, B, Co, Ci, L, dy_bs, x_bs, dy_gs, x_gs, dw_gs, alpha, accumulate, st, db, db_gs);
    case 3: return wgrad_stream_launch_k<WsCfgK1Wide>(dy, x, dw, scratch, G, B, Co, Ci, L, dy_bs, x_bs, dy_gs, x_gs, dw_gs, alpha, accumulate, st, db, db_gs);
    default: return MG_ERR_SHAPE;
    }
}
