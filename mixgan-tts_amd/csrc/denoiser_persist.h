// Denoiser.forward (model/modules.py:420-446) as ONE launch on gfx950: input projection, the 20 gated residual
// layers (model/blocks.py:1157-1176), skip / output projections and (optionally) the clamp + posterior sample of
// p_sample (model/diffusion.py:113-129) run in a single persistent kernel.
//
// Why: the one-launch-per-layer kernel (resblock_fused.h) re-stages the conditioner tile, reads and writes x / skip and
// recomputes the halo columns of h (5.3 % of its MFMAs) in every layer, and nothing overlaps its prologue and epilogue:
// 163 us per layer against a 123 us MFMA floor.  Here a workgroup owns NT frames of one utterance for ALL layers:
//   * the conditioner tile is staged ONCE (it is the same for every layer) and stays in LDS;
//   * the residual stream x and the running skip sum never leave registers: each wave owns a fixed set of channels of
//     both (they are the accumulators of GEMM 3, initialised in place), so a layer touches HBM / L2 only for its weights;
//   * the k=3 convolution's halo is not recomputed: after GEMM 1 each workgroup hands the two edge columns of h to its
//     neighbours (tagged 8-byte granules, write-through stores: cdna_hip_programming.md section 6 Guideline 16, form
//     R2, the data is the flag) and runs the centre tap of GEMM 2 while they travel.  A workgroup executes exactly the
//     algorithmic MFMAs (no halo MFMAs).
// Two tile widths (same code, template parameter NT):
//   NT = 64: 4 waves, wave w owns channels 64w..64w+63 x 64 frames -- ONE wave per SIMD with the whole 512-entry register
//            file (x, the skip sum and GEMM 2's accumulators alone are 384 registers); one workgroup per CU (139 KB of
//            LDS); the weight stream is read once per 64 frames.  Used when it fills the chip (B x ceil(L/64) > 128
//            workgroups).  (Until round 3: 8 waves of 32 channels, two per SIMD; still there as MG_PERSIST_NT=864.  The
//            second wave of every SIMD ran each GEMM phase 12-15 % slower than the first and the workgroup waited for it
//            at every barrier: 350 -> 367 steps/s on the headline when each SIMD has one wave and nothing to wait for.)
//   NT = 32: 4 waves, wave w owns channels 64w..64w+63 x 32 frames; two workgroups per CU (2 x 74 KB); twice the
//            workgroups for small batches / single utterances, at twice the weight stream per frame.
// LDS: condT 256 channels x NT columns    col j <-> frame l0+j     lives all layers
//      hT    256 channels x NT+2 columns  col j <-> frame l0-1+j   h of the current layer incl. the two halo columns;
//                         overwritten in place by g = sigmoid * tanh (col j <-> frame l0+j) once GEMM 2 has read it
//      both k-interleaved (dp_at): a lane's B fragments of a k-group are one ds_read_b128
// Forward progress: tiles are handed out by atomic tickets in START order, so a workgroup only ever waits for workgroups
// that have started or will start as soon as a slot frees; an utterance's tiles are consecutive tickets, and the launcher
// uses this kernel only when an utterance's chain fits in a quarter of the chip's slots.  Every spin is bounded: a
// timeout sets the workspace's sticky error word and a host-visible one (pinned memory, mg_persist_error), the kernel
// drains, and every workgroup that finishes with the error word set writes NaN instead of its output tile -- a failed
// launch cannot be mistaken for a result.
// Hand-off tags and the Philox offset derive from the launch counter kept IN the workspace (sync[2]), so a captured
// graph's replays get fresh tags and fresh noise without any host-side argument changing.
#pragma once
#include "common.h"
#include "resblock_fused.h"

#define DP_SPIN_LIMIT (1u << 21)   // x ~2 us per poll: seconds, then the error word is set and the kernel drains
#define DP_F_ROLES 1               // NT = 32: two ticket queues by hardware wave slot (see the kernel)
#define DP_F_WITHHOLD 2            // test hook: the second tile of every utterance never sends its left edge column

typedef unsigned long long dp_u64;
typedef __attribute__((address_space(1))) dp_u64 dp_gu64;
typedef __attribute__((address_space(1))) unsigned dp_gu32;

struct PersistArgs {
    const float *x_t;    // [B, M, L]
    const float *cond;   // [B, 256, L]
    const float *cproj;  // optional [B, NL * 256, L]: Wc_l cond + bc_l of every layer, precomputed (mg_denoiser_cond_project
                         // or an earlier launch's cproj_out): GEMM 1 is skipped
    float *cproj_out;    // optional, same layout (cproj == NULL): this launch stores its GEMM 1 results there
    const float *in_w, *in_b;             // packed PLAIN [256 rows, K = 96], [256]
    const float *layers;                  // first layer record
    size_t layer_stride, l_wc, l_w3, l_wo, l_bc, l_b3, l_bo;
    const float *p16layers;               // 16-row packs (denoiser_persist16.h): first layer record, stride, offsets
    size_t p16layer_stride, p_wc, p_w3, p_wo;
    const float *skip_w, *skip_b, *out_w, *out_b;
    const float *hvec, *dvec;             // [NL][vec_rows][256], this launch's utterances in rows 0 .. B-1
    int vec_rows;                         // 0: B (the launch's own vectors); n B: a slice of a sampling loop's n steps
    float *out;                           // [B, M, L]  predicted x_0 (pre-clamp), or x_{t-1} when post != 0
    // fused p_sample tail (post != 0): out = c1[t] clamp(x0) + c2[t] x_t + (t > 0) exp(0.5 lv[t]) noise
    const int64_t *t;                     // [B]
    const float *coef1, *coef2, *logvar;  // [T]
    const float *noise;                   // [B, M, L] or NULL: then Philox4x32-10, key = seed,
    unsigned long long seed;              //   counter = (element index, noise_stream << 32 | launches on this workspace)
    unsigned long long noise_stream;      // unique per workspace instance (host-assigned)
    float *x0_out;                        // optional [B, M, L]: the pre-clamp x_0 when post != 0
    dp_u64 *gran;                         // [2 parity][tiles][2 sides][256] {tag << 32 | float bits}
    dp_u64 *team;                         // denoiser_team16.h: the teams' gather buffers (same granules), or NULL
    unsigned *sync;                       // [0] / [16] tickets, [1] error (sticky), [2] launches completed, [3] workgroups done;
                                          // zero once at allocation: the last workgroup out re-arms [0], [16] and [3]
    unsigned *host_err;                   // pinned host word (or NULL): receives the error code at system scope
    unsigned spin_limit;                  // polls before a hand-off wait gives up (DP_SPIN_LIMIT; tests shrink it)
    // SAVE instantiation (training forward): what mg_denoiser_bwd consumes, all [B, 256, L]
    float *x0_save, *y_save, *skip_save;  // ReLU outputs of the input / skip projections, raw skip sum
    float *h_save, *g_save, *sig_save, *tnh_save;   // per layer (stride act_stride floats): h, gate product, sigmoid, tanh
    size_t act_stride;
    unsigned long long *dbg;              // TIMING instantiation only: [tiles][NL + 2][12] cycle stamps of lane 0
    int dbg_wave;                         //   ... of wave dbg_wave (MG_PERSIST_DBG_WAVE, default 0)
    int B, L, M, NL, tiles_per_b, post, clip, n_steps;
    int flags;                            // DP_F_*
    float rsNL;
    // Two problems in one grid (mg_denoiser_fwd_pair; b_split > 0): utterances [0, b_split) are problem 1 (x_t, out,
    // hvec, dvec; nothing saved), [b_split, B) problem 2 (x_t2, out2, hvec2, dvec2; the SAVE stores, indexed from 0).
    // Same weights; problem 2 reads its own conditioner cond2 (utterance b - b_split uses row b - b_split; the caller
    // passes cond again when both phases share it).  post == 0.
    int b_split;
    const float *x_t2, *hvec2, *dvec2, *cond2;
    float *out2;
};

// (chunk, tap) iteration orders of the k loops.  A "chunk" is 32 reduction channels, a k-group 8 of them.
struct DpIterK1 {   // 1x1: 8 chunks
    static constexpr int N = 8, KW = 1;
    static __device__ __forceinline__ int chunk(int it) { return it; }
    static __device__ __forceinline__ int tap(int) { return 0; }
};
struct DpIterHead {   // input projection: K = 96 (80 mel bins padded)
    static constexpr int N = 3, KW = 1;
    static __device__ __forceinline__ int chunk(int it) { return it; }
    static __device__ __forceinline__ int tap(int) { return 0; }
};
struct DpIterCentre {   // k=3 conv, centre tap only: needs no halo column
    static constexpr int N = 8, KW = 3;
    static __device__ __forceinline__ int chunk(int it) { return it; }
    static __device__ __forceinline__ int tap(int) { return 1; }
};
struct DpIterOuter {   // k=3 conv, taps 0 and 2
    static constexpr int N = 16, KW = 3;
    static __device__ __forceinline__ int chunk(int it) { return it >> 1; }
    static __device__ __forceinline__ int tap(int it) { return (it & 1) * 2; }
};

// LDS tile layout ("k-interleaved"): T[g][col][8] floats, g = channel / 8, and inside the 8: channel k = 2e + hh sits at
// position hh * 4 + e.  The B fragments of one k-group (the four 32x32x2 MFMAs' k = 2e + hh, e = 0..3) of a lane are
// then 16 contiguous bytes: ONE ds_read_b128 per k-group and 32-column block (4 LDS cycles) where the row-major tile
// took two ds_read2_b32 (16 LDS cycles).  Measured (tools/ubench/mfma_f32_loop.hip): LDS read instructions, not the
// weight loads, are what pushes the f32 MFMA off its 64-cycle cadence.
__device__ __forceinline__ int dp_pos(int ch) { return ((ch & 1) << 2) | ((ch >> 1) & 3); }
template <int NTC>
__device__ __forceinline__ int dp_at(int ch, int col) { return (((ch >> 3) * NTC + col) << 3) + dp_pos(ch); }

// k loop of one GEMM phase for NMB 32-row blocks x NNB 32-column blocks.  Register pipeline as in rb_mfma_loop
// (resblock_fused.h): weight float4s DIST k-groups ahead in a ring of 4, LDS B fragments one k-group ahead; the
// prefetches are pinned at the TOP of each k-group with sched_barrier (left alone, hipcc sinks them to the end of the
// group: one k-group of latency cover instead of DIST) and the MFMAs keep the written order, which never puts two MFMAs
// on one accumulator back to back.  The iteration order over (chunk, tap) comes from IT.
//   ap[i]: packed weights of block i (+ lane); k-group q of (chunk, tap) = (chunk * KW + tap) * 4 + g sits at ap[i][q * 64]
//   tile:  k-interleaved LDS tile + (this lane's column of n-block 0 for tap 0) * 8 + hh * 4;  NTC: its columns
// DIST: how many k-groups ahead the weight fragments are requested.  3 everywhere since round 3: per-wave stamps
// (MG_PERSIST_DBG_WAVE) show the second wave of every SIMD (waves 4-7) 14 % slower than the first in every GEMM phase
// -- its requests queue behind the older wave's -- and the whole workgroup waits for it at the next barrier.
#ifndef DP_DIST_BIG
#define DP_DIST_BIG 3
#endif
template <int NMB, int NNB, int NTC, class IT, int DIST = (NMB * NNB >= 4 ? DP_DIST_BIG : 3)>
__device__ __forceinline__ void dp_mfma_loop(f32x16 (&acc)[NMB][NNB], const f32x4 *const (&ap)[NMB], const float *__restrict__ tile)
{
    static_assert(DIST >= 1 && DIST <= 3, "ring of 4 slots");
    f32x4 ring[4][NMB];
    f32x4 bb[2][NNB];
    auto qbase = [](int it) { return (IT::chunk(it) * IT::KW + IT::tap(it)) * 4; };
    auto boff = [](int it) { return IT::chunk(it) * (4 * NTC * 8) + IT::tap(it) * 8; };
    {
        const int q0 = qbase(0);
#pragma unroll
        for (int s = 0; s < DIST; ++s)
#pragma unroll
            for (int i = 0; i < NMB; ++i) ring[s][i] = ap[i][(size_t)(q0 + s) * 64];
        const float *T0 = tile + boff(0);
#pragma unroll
        for (int j = 0; j < NNB; ++j) bb[0][j] = *reinterpret_cast<const f32x4 *>(T0 + 32 * 8 * j);
    }
#pragma unroll 1   // keep the layer's code inside the instruction cache
    for (int it = 0; it < IT::N; ++it) {
        const int itn = it + 1 < IT::N ? it + 1 : IT::N - 1;
        const int qc = qbase(it), qn = qbase(itn);
        const float *Tc = tile + boff(it), *Tn = tile + boff(itn);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int qa = u + DIST < 4 ? qc + u + DIST : qn + u + DIST - 4;   // DIST k-groups ahead
#pragma unroll
            for (int i = 0; i < NMB; ++i) ring[(u + DIST) & 3][i] = ap[i][(size_t)qa * 64];
            {
                const float *Tx = u < 3 ? Tc + (u + 1) * (NTC * 8) : Tn;
#pragma unroll
                for (int j = 0; j < NNB; ++j) bb[(u + 1) & 1][j] = *reinterpret_cast<const f32x4 *>(Tx + 32 * 8 * j);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int i = 0; i < NMB; ++i)
#pragma unroll
                    for (int j = 0; j < NNB; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ring[u][i][e], bb[u & 1][j][e], acc[i][j], 0, 0, 0);
                if (NMB * NNB < 4) __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// write the 16 accumulator registers of one 32x32 block (rows ch0 + 8 (r >> 2) + 4 hh + (r & 3), this lane's column)
// into a k-interleaved tile: registers (0, 2) and (1, 3) of every quad are adjacent there -> 8 ds_write_b64
template <int NTC, class F>
__device__ __forceinline__ void dp_store_block(float *T, int ch0, int col, int hh, F val)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float *base = T + ((((ch0 >> 3) + q) * NTC + col) << 3);
        const f32x2 lo = {val(4 * q + 0), val(4 * q + 2)}, hi = {val(4 * q + 1), val(4 * q + 3)};
        *reinterpret_cast<f32x2 *>(base + 2 * hh) = lo;       // channels 4hh, 4hh+2  -> positions 2hh, 2hh+1
        *reinterpret_cast<f32x2 *>(base + 4 + 2 * hh) = hi;   // channels 4hh+1, 4hh+3 -> positions 4+2hh, 5+2hh
    }
}

// Philox4x32-10 (Salmon et al. 2011), counter = (idx_lo, idx_hi, offset_lo, offset_hi), key = seed: 4 x 32 random bits
__device__ __forceinline__ void dp_philox(unsigned long long seed, unsigned long long offset, unsigned long long idx,
                                          unsigned (&r)[4])
{
    unsigned c0 = (unsigned)idx, c1 = (unsigned)(idx >> 32), c2 = (unsigned)offset, c3 = (unsigned)(offset >> 32);
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int i = 0; i < 10; ++i) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1,
                       n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

// one N(0,1) draw for element `idx` of the launch's noise tensor (Box-Muller on two of the four Philox words)
__device__ __forceinline__ float dp_normal(unsigned long long seed, unsigned long long offset, unsigned long long idx)
{
    unsigned r[4];
    dp_philox(seed, offset, idx, r);
    const float u1 = ((float)r[0] + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]
    const float u2 = (float)r[1] * 2.3283064365386963e-10f;            // [0, 1)
    return sqrtf(-2.0f * __logf(u1)) * __cosf(6.283185307179586f * u2);
}

#define DP_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// A hand-off wait gave up (lane 0 of the waiting wave): set the workspace's sticky error word and the host-visible one,
// and make both visible before this workgroup sends anything computed from the halo it never got.
__device__ __forceinline__ void dp_fail(unsigned *sync, unsigned *host_err, unsigned code)
{
    __hip_atomic_store(sync + 1, code, DP_RLX_AGENT);
    if (host_err) __hip_atomic_store(host_err, code, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
}
// End of a workgroup: has this launch (or an earlier one on this workspace) failed?  Every workgroup whose tile
// depends on a dead neighbour's columns received them after that neighbour's dp_fail, so it sees the word set here.
__device__ __forceinline__ bool dp_failed(unsigned *sync)
{
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    return __hip_atomic_load(sync + 1, DP_RLX_AGENT) != 0u;
}
#define DP_STAMP(k) do { if (TIMING && tid == 64 * a.dbg_wave) a.dbg[((size_t)tile * (a.NL + 2) + stamp_row) * 12 + (k)] = clock64(); } while (0)

// A reading step's conditioner projections (PersistArgs.cproj) come through LDS: wave w fetches the NT columns of its own
// RW rows of layer l's [256, L] block straight into the (otherwise unused) conditioner tile region with LDS-direct loads
// -- no registers, issued a phase ahead -- 64 lanes x 16 bytes = 1 KB = 1024 / (4 NT) whole rows per instruction, plain
// row-major with stride NT.  Only the issuing wave reads what it fetched: its own vmcnt(0) is the only synchronisation.
// L % 4 == 0 and a 16-byte aligned base (the launcher checks); a lane past the utterance's end re-reads the row's last
// vector (those columns are never used).  Written as inline assembly on purpose: the compiler orders every later LDS
// access behind an LDS-direct load it knows of (s_waitcnt vmcnt(0) in front of the gate's first ds_write), which would
// put the whole memory latency back on the critical path.
template <int NT, int RW>
__device__ __forceinline__ void dp_cproj_fetch(const float *cp, float *cl, int w, int lane, int l0, int L)
{
    constexpr int LPR = NT / 4;     // lanes per row
    constexpr int RPI = 64 / LPR;   // rows per instruction
    const int rsub = lane / LPR, c4 = lane - rsub * LPR;
    // one 32-bit byte offset per lane, recomputed at every call (the opaque asm: nothing address-like stays live across the
    // layer loop), on top of a uniform base per instruction
    unsigned off = (unsigned)(((RW * w + rsub) * L + min(l0 + 4 * c4, L - 4)) * 4);
    asm volatile("" : "+v"(off));
    float *dst = cl + (size_t)RW * w * NT;
    const unsigned at = (unsigned)(size_t)(__attribute__((address_space(3))) void *)dst;   // LDS byte address, wave-uniform
#pragma unroll
    for (int it = 0; it < RW / RPI; ++it) {
        const float *gi = cp + (size_t)it * RPI * L;   // uniform
        const unsigned m = __builtin_amdgcn_readfirstlane(at + (unsigned)(it * RPI * NT * 4));
        // (m0 is reserved: the compiler loads it in front of each of its own uses)
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m), "v"(off), "s"(gi) : "memory");
    }
}

// CPM: what the launch does with the conditioner projections -- 0: computes them, 1: computes and stores them
// (a.cproj_out), 2: reads them (a.cproj): its own instantiation without GEMM 1 and the conditioner tile.  Compile-time, so
// that no kind of step carries another's registers (as run-time branches the stores cost the plain step 34 spilled
// registers and 2 % of its time).
// SOLO (NT = 32, 4 waves): compiled for ONE workgroup per CU -- one wave per SIMD with the 512-entry register file -- for
// launches of at most one 32-frame tile per CU (B = 8, L = 1000), where the second workgroup slot stays empty anyway.
template <int NT, bool VEC4, bool TIMING = false, bool SAVE = false, int NWV = NT / 8, int CPM = 0, bool SOLO = false>
__global__ __launch_bounds__(NWV * 64, ((NT == 64 && NWV == 4) || SOLO) ? 1 : 2) void denoiser_persist_kernel(PersistArgs a)
{
    static_assert(!SOLO || (NT == 32 && NWV == 4), "SOLO is the one-per-CU build of the 4-wave 32-frame form");
    static_assert(!(CPM && SAVE), "the saving forward computes its projections and keeps none");
    constexpr bool READP = CPM == 2, WRITEP = CPM == 1;
    // weight fragments of the layer's 2 x 2-block (and larger) loops: three k-groups ahead where two waves share a SIMD
    // (dp_mfma_loop), two where one wave has it alone (requests in flight beyond what hides the latency only queue: a
    // ring of eight with 4 / 5 / 7 ahead measured 359 / 354 / 343 steps/s against 366 at 3 and 369 at 2)
#ifndef DP_DIST_SOLO
#define DP_DIST_SOLO 2
#endif
    constexpr int DBIG = ((NT == 64 && NWV == 4) || SOLO) ? DP_DIST_SOLO : DP_DIST_BIG;
#ifndef DP_ONEPASS_WHEN
#define DP_ONEPASS_WHEN (MB == 2 && ((NT == 64 && NWV == 4) || SOLO))
#endif
    static_assert((NT == 32 && (NWV == 4 || NWV == 8)) || (NT == 64 && (NWV == 8 || NWV == 4)),
                  "tile widths: 32 frames (4 waves, two workgroups per CU; or 8 waves, one per CU) or 64 (8 waves, one)");
    constexpr int NTHR = NWV * 64, NW = NWV;        // threads, waves
    constexpr bool TWO_PER_CU = NT == 32 && NWV == 4;
    constexpr int MB = 8 / NW;                      // 32-row blocks of the 256 channels per wave: 2 (4 waves) or 1
    constexpr bool ONEPASS = DP_ONEPASS_WHEN;       // GEMM 2 over all of the wave's row blocks in one loop
    constexpr int NNB = NT / 32;                    // 32-column blocks per tile
    constexpr int NC = NT, NH = NT + 2;             // columns of the cond tile and of the h tile (k-interleaved: dp_at)
    __shared__ __attribute__((aligned(16))) float lds[RB_C * (NC + NH)];
    __shared__ unsigned s_tile, s_dead, s_launch;
    float *condT = lds;                             // col j <-> frame l0+j
    float *hT = lds + RB_C * NC;                    // col j <-> frame l0-1+j (h), or frame l0+j (x_t, g, skip sum)

    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hh = lane >> 5, c32 = lane & 31;
    const int L = a.L;

    // NT = 32: the two workgroups of a CU sit in different hardware wave slots; the matrix pipe serves the older wave
    // first, so one of them runs ahead and the other fills its gaps.  A workgroup that waits for a neighbour which is
    // the starved partner on ITS CU stalls both (measured: 18 % of the kernel): tiles are therefore dealt from two
    // queues, utterances [0, B/2) to even slots and [B/2, B) to odd ones, so that a chain of neighbours shares one role
    // and advances in step.  Speed only: any assignment of tiles to workgroups is correct.
    const int n_tiles = a.tiles_per_b * a.B;
    const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);   // HW_REG_HW_ID; WAVE_ID = bits 3:0
    const int role = (TWO_PER_CU && (a.flags & DP_F_ROLES)) ? (int)(hw_id & 1u) : 0;
    if (tid == 0) {
        const int nA = (TWO_PER_CU && (a.flags & DP_F_ROLES)) ? (a.B / 2) * a.tiles_per_b : n_tiles;   // queue 0: tiles [0, nA)
        const int n_mine = role == 0 ? nA : n_tiles - nA;
        unsigned tk = __hip_atomic_fetch_add(a.sync + (role ? 16 : 0), 1u, DP_RLX_AGENT);   // tickets in START order
        int tl;
        if ((int)tk < n_mine) tl = (role == 0 ? 0 : nA) + (int)tk;
        else {   // this role's queue is empty: the other one has exactly as many tiles left as such workgroups
            tk = __hip_atomic_fetch_add(a.sync + (role ? 0 : 16), 1u, DP_RLX_AGENT);
            tl = (role == 0 ? nA : 0) + (int)tk;
        }
        s_tile = (unsigned)tl;
        s_launch = __hip_atomic_load(a.sync + 2, DP_RLX_AGENT);      // advanced only after every workgroup has exited
        s_dead = 0u;
    }
    __syncthreads();
    const int tile = min((int)s_tile, n_tiles - 1);
    const unsigned launch_no = s_launch;   // launches completed on this workspace: hand-off tags and the noise offset
    const int bg = tile / a.tiles_per_b, jt = tile - bg * a.tiles_per_b;   // bg: utterance in the grid
    const bool second = a.b_split > 0 && bg >= a.b_split;                   // wave-uniform (workgroup-uniform)
    const int b = second ? bg - a.b_split : bg;                             // utterance inside its problem
    const int Bp = a.b_split > 0 ? (second ? a.B - a.b_split : a.b_split) : a.B;
    const bool do_save = SAVE && (a.b_split == 0 || second);
    const float *const x_in = second ? a.x_t2 : a.x_t;
    float *const x_out = second ? a.out2 : a.out;
    const int l0 = jt * NT;
    const bool has_left = jt > 0, has_right = jt + 1 < a.tiles_per_b;
    int stamp_row = 0;
    DP_STAMP(0);
    if (TIMING && tid == 0) {
        a.dbg[((size_t)tile * (a.NL + 2)) * 12 + 2] = hw_id;
        a.dbg[((size_t)tile * (a.NL + 2)) * 12 + 3] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
    }

    // ---------------------------------------------------------------- stage the cond tile (once) and the x_t tile
    {
        const float *cb = (second ? a.cond2 : a.cond) + (size_t)b * RB_C * L;
        if (READP) {
            // the conditioner enters only through its precomputed projections: no tile to stage; layer 0's are fetched
            if (VEC4) dp_cproj_fetch<NT, 32 * MB>(a.cproj + (size_t)b * a.NL * RB_C * L, lds, w, lane, l0, L);
        } else if (VEC4) {
#pragma unroll
            for (int k = 0; k < 64 * NT / NTHR; ++k) {   // 256 rows x NT/4 float4 (frames l0 .. l0+NT-1)
                const int idx = tid + k * NTHR;
                const int row = idx / (NT / 4), c4 = idx - row * (NT / 4);
                const int f0 = l0 + 4 * c4;
                const bool ok = f0 < L;   // L % 4 == 0: a float4 is inside or outside as a whole
                const f32x4 v = *reinterpret_cast<const f32x4 *>(cb + (size_t)row * L + min(f0, L - 4));
#pragma unroll
                for (int e = 0; e < 4; ++e) condT[dp_at<NC>(row, 4 * c4 + e)] = ok ? v[e] : 0.f;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 256 * NT / NTHR; ++k) {   // 256 rows x NT frames
                const int idx = tid + k * NTHR;
                const int row = idx / NT, cc = idx - row * NT;
                const int f = l0 + cc;
                const float v = cb[(size_t)row * L + min(f, L - 1)];
                condT[dp_at<NC>(row, cc)] = f < L ? v : 0.f;
            }
        }
        const float *xb = x_in + (size_t)b * a.M * L;
#pragma unroll
        for (int k = 0; k < 96 * NT / NTHR; ++k) {   // 96 rows (M = 80 padded) x NT frames -> hT channels 0..95, col c <-> frame l0+c
            const int idx = tid + k * NTHR;
            const int row = idx / NT, c = idx - row * NT;
            const int f = l0 + c;
            const float v = xb[(size_t)min(row, a.M - 1) * L + min(f, L - 1)];
            hT[dp_at<NH>(row, c)] = (row < a.M && f < L) ? v : 0.f;
        }
    }

    // residual stream and skip sum of this wave's 32*MB channels x NT frames: st[0..MB-1] = x, st[MB..2MB-1] = skip sum
    f32x16 st[2 * MB][NNB];
    const int rbase = 32 * MB * w;   // first channel of this wave
    auto row_of = [&](int i, int r) { return rbase + 32 * i + 8 * (r >> 2) + 4 * hh + (r & 3); };
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NNB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                st[i][j][r] = a.in_b[row_of(i, r)];
                st[MB + i][j][r] = 0.f;
            }
    __syncthreads();
    {   // input projection + ReLU (model/modules.py:430-431)
        f32x16 acc[MB][NNB];
        const f32x4 *ap[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            ap[i] = reinterpret_cast<const f32x4 *>(a.in_w) + (size_t)(MB * w + i) * 12 * 64 + lane;
#pragma unroll
            for (int j = 0; j < NNB; ++j) acc[i][j] = st[i][j];
        }
        dp_mfma_loop<MB, NNB, NH, DpIterHead>(acc, ap, hT + c32 * 8 + hh * 4);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[i][j][r] = fmaxf(acc[i][j][r], 0.f);
    }
    (void)0;
    DP_STAMP(1);
    bool fvalid[NNB];   // this lane's frame of n-block j exists
    const size_t bbase = (size_t)b * RB_C * L;
    // SAVE: one 32x32 block (channels ch0.., this lane's frame of n-block j) to a [B, 256, L] tensor
    auto save_block = [&](float *dst, int ch0, int j, auto val) {
        const int f = l0 + 32 * j + c32;
        if (f < L) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dst[bbase + (size_t)(ch0 + 8 * (r >> 2) + 4 * hh + (r & 3)) * L + f] = val(r);
        }
    };
#pragma unroll
    for (int j = 0; j < NNB; ++j) fvalid[j] = l0 + 32 * j + c32 < L;
    if (SAVE && do_save) {
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NNB; ++j) save_block(a.x0_save, rbase + 32 * i, j, [&](int r) { return st[i][j][r]; });
    }
    dp_gu64 *const gran = (dp_gu64 *)a.gran;

    for (int l = 0; l < a.NL; ++l) {
        const float *lp = a.layers + (size_t)l * a.layer_stride;
        const size_t vrows = a.vec_rows ? (size_t)a.vec_rows : (size_t)Bp;
        const float *hv = (second ? a.hvec2 : a.hvec) + ((size_t)l * vrows + b) * RB_C;
        const float *dv = (second ? a.dvec2 : a.dvec) + ((size_t)l * vrows + b) * RB_C;
        const unsigned epoch = launch_no * ((unsigned)a.NL + 1u) + (unsigned)l + 1u;   // never repeats on a workspace
        const int par = l & 1;
        stamp_row = l + 1;
        DP_STAMP(0);

        // ------------------------------------------------------------ GEMM 1: h = (Wc cond + bc) + (x + (Wd s [+ Wp spk]))
        // The first bracket does not depend on x_t: inside a T-step sampling loop it is the same in every step, and the
        // caller may hand it over precomputed for all layers (a.cproj, mg_denoiser_cond_project: one GEMM per loop
        // instead of one per layer and step).  Both ways evaluate fl(P + fl(x + vec)), P = bc with the products added onto
        // it in channel order (the accumulators START at bc: P is a function of cond alone) -- bit-identical results.
        f32x16 acc1[MB][NNB];
        if (READP && VEC4) {
            // fetched into LDS a phase ago by this wave itself (dp_cproj_fetch): wait for its own loads, read, and the
            // region is free for the next layer's
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NNB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][j][r] = lds[row_of(i, r) * NT + 32 * j + c32];
        } else if (READP) {   // L % 4 != 0: straight from global memory (uniform base + lane offsets as for the stores below)
            const float *cp = a.cproj + ((size_t)b * a.NL + l) * RB_C * L;
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NNB; ++j) {
                    unsigned lo = (unsigned)((rbase + 4 * hh) * L + min(l0 + 32 * j + c32, L - 1));
                    asm volatile("" : "+v"(lo));
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][j][r] = cp[lo + (unsigned)((32 * i + 8 * (r >> 2) + (r & 3)) * L)];
                }
        } else {
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NNB; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc1[i][j][r] = lp[a.l_bc + row_of(i, r)];
            const f32x4 *wc = reinterpret_cast<const f32x4 *>(lp + a.l_wc);
            const f32x4 *ap[MB];
#pragma unroll
            for (int i = 0; i < MB; ++i) ap[i] = wc + (size_t)(MB * w + i) * 32 * 64 + lane;
            dp_mfma_loop<MB, NNB, NC, DpIterK1, (MB * NNB >= 4 ? DBIG : 3)>(acc1, ap, condT + c32 * 8 + hh * 4);
            if (WRITEP) {   // the first step of a sampling loop leaves the projections for the steps behind it
                // uniform base + a 32-bit lane offset that is recomputed every layer (the opaque asm): kept live across
                // the layer loop, 32 store addresses cost 34 spilled registers
                float *co = a.cproj_out + ((size_t)b * a.NL + l) * RB_C * L;
                unsigned lo = (unsigned)((rbase + 4 * hh) * L + l0 + c32);
                asm volatile("" : "+v"(lo));
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NNB; ++j)
                        if (l0 + 32 * j + c32 < L) {
#pragma unroll
                            for (int r = 0; r < 16; ++r)
                                co[lo + (unsigned)((32 * i + 8 * (r >> 2) + (r & 3)) * L + 32 * j)] = acc1[i][j][r];
                        }
            }
        }
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[i][j][r] += st[i][j][r] + hv[row_of(i, r)];
        DP_STAMP(1);
        // GEMM 2's accumulators start at the conv bias: these loads fly during the barrier and the h write-back.
        // acc2[p][0] = gate rows, acc2[p][1] = filter rows of channels 32*(MB*w + p) .. +31
        f32x16 acc2[MB][2][NNB];
#pragma unroll
        for (int p = 0; p < MB; ++p)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = row_of(p, r);
                const float bg = lp[a.l_b3 + ch], bf = lp[a.l_b3 + RB_C + ch];
#pragma unroll
                for (int j = 0; j < NNB; ++j) {
                    acc2[p][0][j][r] = bg;
                    acc2[p][1][j][r] = bf;
                }
            }
        __syncthreads();   // every wave is past the previous layer's GEMM 3 (or the head GEMM): hT may be rewritten
        DP_STAMP(2);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
            {
                dp_store_block<NH>(hT, rbase + 32 * i, 1 + 32 * j + c32, hh,   // zero padding of the conv applies to h
                                   [&](int r) { return fvalid[j] ? acc1[i][j][r] : 0.f; });
                if (SAVE && do_save) save_block(a.h_save + (size_t)l * a.act_stride, rbase + 32 * i, j, [&](int r) { return acc1[i][j][r]; });
            }
        // The next layer's conditioner projections: on their way for the rest of this layer.  Issued HERE because vector
        // loads return in order -- whatever load is waited for next also waits for these: behind the h stores (which
        // need everything loaded so far, so no earlier wait can sink below the fetch) the next one is GEMM 2's first
        // weight fragment, a barrier, the halo publish and an L2 round trip away.
        if (READP && VEC4 && l + 1 < a.NL)
            dp_cproj_fetch<NT, 32 * MB>(a.cproj + ((size_t)b * a.NL + l + 1) * RB_C * L, lds, w, lane, l0, L);
        __syncthreads();   // interior columns of hT complete
        DP_STAMP(3);

        // ------------------------------------------------------------ hand the edge columns to the neighbours
        // wave 0: my frame l0 -> right halo of tile-1;  wave 1: my frame l0+NT-1 -> left halo of tile+1
        if (w < 2) {
            bool go = w == 0 ? has_left : has_right;
            if ((a.flags & DP_F_WITHHOLD) && w == 0 && jt == 1) go = false;
            if (go) {
                const int dst_tile = w == 0 ? tile - 1 : tile + 1;
                const int col = w == 0 ? 1 : NT;
                dp_gu64 *g = gran + (((size_t)par * n_tiles + dst_tile) * 2 + (w == 0 ? 1 : 0)) * RB_C;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = lane + 64 * k;
                    const dp_u64 v = ((dp_u64)epoch << 32) | (dp_u64)__float_as_uint(hT[dp_at<NH>(row, col)]);
                    __hip_atomic_store(g + row, v, DP_RLX_AGENT);   // one aligned 8-byte write-through store per granule
                }
            }
        }

        // ------------------------------------------------------------ GEMM 2, centre tap (needs no halo)
        // one pass per 32 channels (two 32-row blocks: gate + filter rows): with all of a wave's blocks in one pass the
        // weight ring alone is 64 VGPRs next to 64 of accumulators and the 64 of x / skip, and the kernel spills
        const f32x4 *w3 = reinterpret_cast<const f32x4 *>(lp + a.l_w3);
        const f32x4 *ap2[MB][2];
#pragma unroll
        for (int p = 0; p < MB; ++p) {
            ap2[p][0] = w3 + (size_t)(2 * (MB * w + p)) * 96 * 64 + lane;
            ap2[p][1] = w3 + (size_t)(2 * (MB * w + p) + 1) * 96 * 64 + lane;
        }
        // ONEPASS (one wave per SIMD, 512 registers): all four row blocks of the wave in one loop -- half the B-fragment
        // reads and one pipeline fill per phase instead of two
        if (ONEPASS) {
            dp_mfma_loop<2 * MB, NNB, NH, DpIterCentre, DBIG>(reinterpret_cast<f32x16 (&)[2 * MB][NNB]>(acc2),
                                                              reinterpret_cast<const f32x4 *const (&)[2 * MB]>(ap2),
                                                              hT + c32 * 8 + hh * 4);
        } else {
#pragma unroll
            for (int p = 0; p < MB; ++p)
                dp_mfma_loop<2, NNB, NH, DpIterCentre, (2 * NNB >= 4 ? DBIG : 3)>(acc2[p], ap2[p], hT + c32 * 8 + hh * 4);
        }
        DP_STAMP(4);

        // ------------------------------------------------------------ receive the halo columns
        if (w < 2) {
            const bool from = w == 0 ? has_left : has_right;
            const int col = w == 0 ? 0 : NT + 1;
            unsigned v[4] = {0u, 0u, 0u, 0u};
            if (from && s_dead == 0u) {   // after a timeout: keep going (the output is poisoned at the end), never hang
                dp_gu64 *g = gran + (((size_t)par * n_tiles + tile) * 2 + w) * RB_C;
                unsigned spins = 0;
                for (;;) {
                    bool ok = true;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const dp_u64 x = __hip_atomic_load(g + lane + 64 * k, DP_RLX_AGENT);
                        v[k] = (unsigned)x;
                        ok &= (unsigned)(x >> 32) == epoch;
                    }
                    if (__all(ok)) break;
                    if (++spins > a.spin_limit) {   // uniform across the wave
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
            for (int k = 0; k < 4; ++k) hT[dp_at<NH>(lane + 64 * k, col)] = from ? __uint_as_float(v[k]) : 0.f;
        }
        DP_STAMP(5);
        __syncthreads();   // halo columns in place
        DP_STAMP(6);

        // ------------------------------------------------------------ GEMM 2, taps 0 and 2; gate
        if (ONEPASS) {
            dp_mfma_loop<2 * MB, NNB, NH, DpIterOuter, DBIG>(reinterpret_cast<f32x16 (&)[2 * MB][NNB]>(acc2),
                                                             reinterpret_cast<const f32x4 *const (&)[2 * MB]>(ap2),
                                                             hT + c32 * 8 + hh * 4);
        } else {
#pragma unroll
            for (int p = 0; p < MB; ++p)
                dp_mfma_loop<2, NNB, NH, DpIterOuter, (2 * NNB >= 4 ? DBIG : 3)>(acc2[p], ap2[p], hT + c32 * 8 + hh * 4);
        }
        DP_STAMP(7);
        __syncthreads();   // every wave has read hT for the last time: g may overwrite it
        DP_STAMP(8);
#pragma unroll
        for (int p = 0; p < MB; ++p)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
            {
                if (SAVE && do_save) {   // the backward's gate derivative needs sigmoid and tanh themselves
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        acc2[p][0][j][r] = mg_sigmoid(acc2[p][0][j][r]);
                        acc2[p][1][j][r] = mg_tanh(acc2[p][1][j][r]);
                    }
                    const size_t lo = (size_t)l * a.act_stride;
                    save_block(a.sig_save + lo, rbase + 32 * p, j, [&](int r) { return acc2[p][0][j][r]; });
                    save_block(a.tnh_save + lo, rbase + 32 * p, j, [&](int r) { return acc2[p][1][j][r]; });
                    save_block(a.g_save + lo, rbase + 32 * p, j, [&](int r) { return acc2[p][0][j][r] * acc2[p][1][j][r]; });
                    dp_store_block<NH>(hT, rbase + 32 * p, 32 * j + c32, hh, [&](int r) { return acc2[p][0][j][r] * acc2[p][1][j][r]; });
                } else {
                    dp_store_block<NH>(hT, rbase + 32 * p, 32 * j + c32, hh,
                                       [&](int r) { return mg_sigmoid(acc2[p][0][j][r]) * mg_tanh(acc2[p][1][j][r]); });
                }
            }
        // GEMM 3's accumulators start as its addends: x + bo + Wd s and skip + bo (model/blocks.py:1166,1174-1176)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = row_of(i, r);
                    st[i][j][r] += lp[a.l_bo + row] + dv[row];
                    st[MB + i][j][r] += lp[a.l_bo + RB_C + row];
                }
        DP_STAMP(9);
        __syncthreads();   // g complete
        DP_STAMP(10);

        // ------------------------------------------------------------ GEMM 3: o = Wo g; x' = (o[:C] + x + Wd s)/sqrt2; skip += o[C:]
        {
            const f32x4 *wo = reinterpret_cast<const f32x4 *>(lp + a.l_wo);
            const f32x4 *ap[2 * MB];
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                ap[i] = wo + (size_t)(MB * w + i) * 32 * 64 + lane;            // x rows
                ap[MB + i] = wo + (size_t)(8 + MB * w + i) * 32 * 64 + lane;   // skip rows
            }
            dp_mfma_loop<2 * MB, NNB, NH, DpIterK1, (2 * MB * NNB >= 4 ? DBIG : 3)>(st, ap, hT + c32 * 8 + hh * 4);
        }
        DP_STAMP(11);
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) st[i][j][r] *= 0.70710678118654752440f;
    }

    // ---------------------------------------------------------------- tail: sum(skip)/sqrt(NL) -> skip_projection -> ReLU -> output_projection
    stamp_row = a.NL + 1;
    DP_STAMP(0);
    __syncthreads();   // last GEMM 3 done reading hT
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NNB; ++j)
            {
                dp_store_block<NH>(hT, rbase + 32 * i, 32 * j + c32, hh, [&](int r) { return st[MB + i][j][r] * a.rsNL; });
                if (SAVE && do_save) save_block(a.skip_save, rbase + 32 * i, j, [&](int r) { return st[MB + i][j][r]; });
            }
    __syncthreads();
    {
        f32x16 acc[MB][NNB];
        const f32x4 *ap[MB];
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            ap[i] = reinterpret_cast<const f32x4 *>(a.skip_w) + (size_t)(MB * w + i) * 32 * 64 + lane;
#pragma unroll
            for (int j = 0; j < NNB; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = a.skip_b[row_of(i, r)];
        }
        dp_mfma_loop<MB, NNB, NH, DpIterK1>(acc, ap, hT + c32 * 8 + hh * 4);
        // y -> the cond tile's storage (dead now), col c <-> frame l0+c
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NNB; ++j)
            {
                dp_store_block<NC>(condT, rbase + 32 * i, 32 * j + c32, hh, [&](int r) { return fmaxf(acc[i][j][r], 0.f); });
                if (SAVE && do_save) save_block(a.y_save, rbase + 32 * i, j, [&](int r) { return fmaxf(acc[i][j][r], 0.f); });
            }
    }
    __syncthreads();
    const int mblocks = (a.M + 31) / 32;
    // output projection: (32-row block, 32-column block) tasks dealt over the waves (M <= 96: 3 NNB tasks; one each, two for
    // the first waves of the 4-wave 64-frame form)
    for (int task = w; task < mblocks * NNB; task += NW) {
        const int mb = task / NNB, nb = task - mb * NNB;
        f32x16 o[1][1];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 32 * mb + 8 * (r >> 2) + 4 * hh + (r & 3);
            o[0][0][r] = row < a.M ? a.out_b[row] : 0.f;
        }
        const f32x4 *const ap[1] = {reinterpret_cast<const f32x4 *>(a.out_w) + (size_t)mb * 32 * 64 + lane};
        dp_mfma_loop<1, 1, NC, DpIterK1>(o, ap, condT + (32 * nb + c32) * 8 + hh * 4);
        const int f = l0 + 32 * nb + c32;
        const size_t bo = (size_t)b * a.M * L;
        // a hand-off timed out somewhere in this launch (or an earlier one on this workspace whose error nobody has
        // cleared): no tile may look like a result
        const bool bad = dp_failed(a.sync);
        const float poison = __builtin_nanf("");
        if (!a.post) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mb + 8 * (r >> 2) + 4 * hh + (r & 3);
                if (row < a.M && f < L) x_out[bo + (size_t)row * L + f] = bad ? poison : o[0][0][r];
            }
        } else {
            // p_sample tail (model/diffusion.py:113-129): clamp, posterior mean, + sigma * noise unless t == 0
            long tb = (long)a.t[b];
            tb = tb < 0 ? 0 : (tb >= a.n_steps ? a.n_steps - 1 : tb);
            const float c1 = a.coef1[tb], c2 = a.coef2[tb];
            const float sg = tb == 0 ? 0.f : __expf(0.5f * a.logvar[tb]);
            const unsigned long long seed = a.seed, off = (a.noise_stream << 32) | (unsigned long long)launch_no;
            const int fc = min(f, L - 1);
            float xt[16], nz[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {   // loads first (clamped addresses), math and predicated stores after
                const int row = min(32 * mb + 8 * (r >> 2) + 4 * hh + (r & 3), a.M - 1);
                const size_t e = bo + (size_t)row * L + fc;
                xt[r] = a.x_t[e];
                nz[r] = a.noise ? a.noise[e] : dp_normal(seed, off, e);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * mb + 8 * (r >> 2) + 4 * hh + (r & 3);
                if (row < a.M && f < L) {
                    const size_t e = bo + (size_t)row * L + f;
                    float x0 = o[0][0][r];
                    if (a.x0_out) a.x0_out[e] = x0;
                    if (a.clip) x0 = fminf(fmaxf(x0, -1.f), 1.f);
                    a.out[e] = bad ? poison : fmaf(sg, nz[r], fmaf(c1, x0, c2 * xt[r]));
                }
            }
        }
    }
    DP_STAMP(1);
    // ---------------------------------------------------------------- last workgroup out re-arms the tickets for the next launch
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
