// Denoiser.forward (model/modules.py:420-446) on gfx950: step-embedding MLP + 20 gated residual
// blocks (model/blocks.py:1157-1176) + skip/output projections, as fp32-MFMA implicit GEMMs with
// every elementwise op fused into a GEMM epilogue.
//
// Per residual layer (3 launches, all on conv_mfma_kernel):
//   (1) h = Wc * cond + bc + x + (Wd s [+ Wp spk])            k=1,  EpiCond
//   (2) g = sigmoid(z[:C]) * tanh(z[C:]),  z = W3 (*) h + b3   k=3,  EpiGate   (rows gate-interleaved)
//   (3) o = Wo g + bo;  x <- (o[:C] + x + Wd s)/sqrt2;  skip += o[C:]     k=1, EpiResSkip
// The 20 skip tensors are never materialised (the reference stacks them, model/modules.py:441);
// a running fp32 sum is kept instead.
#include "denoiser_common.h"
#include "resblock_fused.h"
#include "resblock_split.h"
#include "pointwise_fused.h"
#include "denoiser_persist.h"
#include "denoiser_persist16.h"
#include "denoiser_team16.h"
#include <atomic>
#include <cstdlib>
#include <cstring>

// ------------------------------------------------------------------------------------------ epilogues
struct EpiCond {
    struct Params {
        float *out;         // h [B, C, L]
        const float *bias;  // [C]
        const float *x;     // [B, C, L]
        const float *vec;   // [B, C]  (Wd s [+ Wp spk])
        int C;
    };
    template <int WM, int NNB>
    static __device__ __forceinline__ void run(const Params &p, f32x16 (&acc)[WM][NNB], int b, int mrow0, int l0w,
                                               int lane, int Lout)
    {
        const int h = lane >> 5, c = lane & 31;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mrow0 + i * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (row >= p.C) continue;
                const float add = p.bias[row] + p.vec[(size_t)b * p.C + row];
                const size_t ro = ((size_t)b * p.C + row) * Lout;
#pragma unroll
                for (int j = 0; j < NNB; ++j) {
                    const int l = l0w + j * 32 + c;
                    if (l < Lout) p.out[ro + l] = acc[i][j][r] + add + p.x[ro + l];
                }
            }
        }
    }
};

struct EpiGate {
    struct Params {
        float *out;         // g [B, C, L]
        const float *bias;  // [2C]: gate rows then filter rows (reference order, model/blocks.py:1170)
        float *sig;         // optional saves for backward: sigmoid(gate), tanh(filter)  [B, C, L]
        float *tnh;
        int C;
    };
    template <int WM, int NNB>
    static __device__ __forceinline__ void run(const Params &p, f32x16 (&acc)[WM][NNB], int b, int mrow0, int l0w,
                                               int lane, int Lout)
    {
        // needs the gate/filter 32-row block pair in one wave: only the WM == 2 tiling is launched
        if constexpr (WM != 2) return;
        const int h = lane >> 5, c = lane & 31;
        const int ch0 = (mrow0 / 64) * 32;  // packed block pair -> first output channel
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = ch0 + 8 * (r >> 2) + 4 * h + (r & 3);
            if (ch >= p.C) continue;
            const float bg = p.bias[ch], bf = p.bias[p.C + ch];
            const size_t ro = ((size_t)b * p.C + ch) * Lout;
#pragma unroll
            for (int j = 0; j < NNB; ++j) {
                const int l = l0w + j * 32 + c;
                if (l < Lout) {
                    const float s = mg_sigmoid(acc[0][j][r] + bg);
                    const float t = mg_tanh(acc[WM - 1][j][r] + bf);
                    p.out[ro + l] = s * t;
                    if (p.sig) {
                        p.sig[ro + l] = s;
                        p.tnh[ro + l] = t;
                    }
                }
            }
        }
    }
};

template <>
struct EpiNeedsWM2<EpiGate> { static constexpr bool value = true; };

struct EpiResSkip {
    struct Params {
        float *x;           // [B, C, L] in/out
        float *skip;        // [B, C, L] running sum
        const float *bias;  // [2C]
        const float *dvec;  // [B, C]  Wd s
        int C;
        int first;          // layer 0: skip = ..., else skip += ...
    };
    template <int WM, int NNB>
    static __device__ __forceinline__ void run(const Params &p, f32x16 (&acc)[WM][NNB], int b, int mrow0, int l0w,
                                               int lane, int Lout)
    {
        const int h = lane >> 5, c = lane & 31;
        const float rs2 = 0.70710678118654752440f;
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = mrow0 + i * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (row >= 2 * p.C) continue;
                const float bv = p.bias[row];
                if (row < p.C) {
                    const float dv = p.dvec[(size_t)b * p.C + row];
                    const size_t ro = ((size_t)b * p.C + row) * Lout;
#pragma unroll
                    for (int j = 0; j < NNB; ++j) {
                        const int l = l0w + j * 32 + c;
                        if (l < Lout) p.x[ro + l] = (acc[i][j][r] + bv + (p.x[ro + l] + dv)) * rs2;
                    }
                } else {
                    const size_t ro = ((size_t)b * p.C + (row - p.C)) * Lout;
#pragma unroll
                    for (int j = 0; j < NNB; ++j) {
                        const int l = l0w + j * 32 + c;
                        if (l < Lout) {
                            const float v = acc[i][j][r] + bv;
                            p.skip[ro + l] = p.first ? v : p.skip[ro + l] + v;
                        }
                    }
                }
            }
        }
    }
};

// ------------------------------------------------------------------------------------------ packing
extern "C" size_t mg_denoiser_packed_floats(const mg_denoiser_dims *d, int flags)
{
    if (den_check(d) != MG_OK) return 0;
    if ((flags & MG_DEN_SPLIT) && (d->channels != RB_C || d->cond_channels != RB_C)) return 0;
    return den_layout(d, flags).total;
}

static int copy_d2d(float *dst, const float *src, size_t n, hipStream_t st)
{
    if (!src) return MG_ERR_ARG;
    hipError_t e = hipMemcpyAsync(dst, src, n * sizeof(float), hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? MG_OK : (int)e;
}

// ---------------------------------------------------------------------------------------------
// Packing.  A training step repacks every weight after the optimizer step: issued one matrix at a time that is
// ~230 tiny launches and device-to-device copies (0.5 ms + 0.5 ms of a 17 ms step).  Here the whole job is a table
// of (source pointer, destination offset, pack descriptor) entries, uploaded into a reserved slice of the packed
// buffer and executed by ONE kernel; a workgroup finds its entry by the prefix sums of the per-entry block counts.
// ---------------------------------------------------------------------------------------------
struct PackJob {
    const float *src;
    unsigned long long dst;     // float offset into `packed`
    unsigned long long total;   // elements to produce
    PackDesc d;                 // d.mode < 0: plain copy of `total` floats
    unsigned first_block, nblocks;
};

__global__ __launch_bounds__(256) void pack_table_kernel(const PackJob *__restrict__ jobs, int njobs,
                                                         float *__restrict__ packed)
{
    __shared__ int which;
    if (threadIdx.x == 0) {
        int lo = 0, hi = njobs - 1;   // last job whose first_block <= blockIdx.x
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (jobs[mid].first_block <= blockIdx.x) lo = mid;
            else hi = mid - 1;
        }
        which = lo;
    }
    __syncthreads();
    const PackJob j = jobs[which];
    const size_t stride = (size_t)j.nblocks * 256;
    float *out = packed + j.dst;
    for (size_t idx = (size_t)(blockIdx.x - j.first_block) * 256 + threadIdx.x; idx < j.total; idx += stride) {
        if (j.d.mode < 0) {
            out[idx] = j.src[idx];
        } else {
            size_t dst;
            const float v = mg_pack_element(j.src, j.d, idx, &dst);
            out[dst] = v;
        }
    }
}

#define MG_DEN_MAX_JOBS 768
static_assert(sizeof(PackJob) <= 96, "den_layout reserves 96 bytes per pack job");

extern "C" int mg_denoiser_pack(const mg_denoiser_dims *d, const float *const *w, const float *freq, float *packed,
                                int flags, void *stream)
{
    MG_TRY(den_check(d));
    if (!w || !packed || !freq) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int C = d->channels, H = d->cond_channels, M = d->mel_bins;
    const int with_backward = flags & MG_DEN_BACKWARD;
    const DenLayout o = den_layout(d, flags);
    for (int i = 0; i < MG_DEN_HEAD_PTRS; ++i)
        if (!w[i]) return MG_ERR_ARG;

    static thread_local PackJob jobs[MG_DEN_MAX_JOBS];
    int n = 0;
    unsigned blocks = 0;
    int rc = MG_OK;
    auto push = [&](const float *src, size_t dst, size_t total, const PackDesc &pd) {
        if (!src || n >= MG_DEN_MAX_JOBS) {
            rc = MG_ERR_ARG;
            return;
        }
        unsigned nb = (unsigned)((total + 255) / 256);
        if (nb > 512) nb = 512;   // grid-stride inside the job beyond that
        if (nb < 1) nb = 1;
        jobs[n++] = PackJob{src, dst, total, pd, blocks, nb};
        blocks += nb;
    };
    auto copy = [&](size_t dst, const float *src, size_t count) {
        PackDesc pd{};
        pd.mode = -1;
        push(src, dst, count, pd);
    };
    auto pack = [&](const float *src, size_t dst, int Co, int Ci, int K, int mode, int q0 = 0, int Qtot = 0) {
        PackDesc pd;
        size_t total;
        const int r = mg_pack_desc(Co, Ci, K, mode, q0, Qtot, &pd, &total);
        if (r != MG_OK) {
            rc = r;
            return;
        }
        push(src, dst, total, pd);
    };
    copy(o.freq, freq, C / 2);
    pack(w[0], o.in_w, C, M, 1, MG_PACK_PLAIN);
    copy(o.in_b, w[1], C);
    copy(o.mlp0, w[2], (size_t)4 * C * C);
    copy(o.mlp2, w[3], (size_t)4 * C * C);
    pack(w[4], o.skip_w, C, C, 1, MG_PACK_PLAIN);
    copy(o.skip_b, w[5], C);
    pack(w[6], o.out_w, M, C, 1, MG_PACK_PLAIN);
    copy(o.out_b, w[7], M);
    for (int l = 0; l < d->n_layers; ++l) {
        const float *const *lw = w + MG_DEN_HEAD_PTRS + (size_t)l * MG_DEN_LAYER_PTRS;
        const size_t lp = o.layers + (size_t)l * o.layer_stride;
        for (int j = 0; j < 7; ++j)
            if (!lw[j]) return MG_ERR_ARG;
        pack(lw[0], lp + o.l_w3, 2 * C, C, 3, MG_PACK_GATE);
        copy(lp + o.l_b3, lw[1], 2 * C);
        copy(lp + o.l_wd, lw[2], (size_t)C * C);
        pack(lw[3], lp + o.l_wc, C, H, 1, MG_PACK_PLAIN);
        copy(lp + o.l_bc, lw[4], C);
        pack(lw[5], lp + o.l_wo, 2 * C, C, 1, MG_PACK_PLAIN);
        copy(lp + o.l_bo, lw[6], 2 * C);
        if (d->multi_speaker) copy(lp + o.l_wp, lw[7], (size_t)C * H);
    }
    if (with_backward) {
        // data-gradient (transposed, tap-flipped) packs for mg_denoiser_bwd
        pack(w[0], o.in_wT, C, M, 1, MG_PACK_DGRAD);
        pack(w[4], o.skip_wT, C, C, 1, MG_PACK_DGRAD);
        pack(w[6], o.out_wT, M, C, 1, MG_PACK_DGRAD);
        const int Qtot = d->n_layers * C / 8;
        for (int l = 0; l < d->n_layers; ++l) {
            const float *const *lw = w + MG_DEN_HEAD_PTRS + (size_t)l * MG_DEN_LAYER_PTRS;
            const size_t bp = o.blayers + (size_t)l * o.blayer_stride;
            pack(lw[3], o.wc_allT, C, H, 1, MG_PACK_DGRAD, l * (C / 8), Qtot);
            pack(lw[0], bp + o.bl_w3T, 2 * C, C, 3, MG_PACK_DGRAD);
            pack(lw[5], bp + o.bl_woT, 2 * C, C, 1, MG_PACK_DGRAD);
        }
    }
    if (flags & MG_DEN_P16) {
        if (C != RB_C || H != RB_C) return MG_ERR_SHAPE;
        pack(w[0], o.in_w16, C, M, 1, MG_PACK_PLAIN16);
        pack(w[4], o.skip_w16, C, C, 1, MG_PACK_PLAIN16);
        pack(w[6], o.out_w16, M, C, 1, MG_PACK_PLAIN16);
        for (int l = 0; l < d->n_layers; ++l) {
            const float *const *lw = w + MG_DEN_HEAD_PTRS + (size_t)l * MG_DEN_LAYER_PTRS;
            const size_t pp = o.p16layers + (size_t)l * o.p16layer_stride;
            pack(lw[3], pp + o.p_wc, C, H, 1, MG_PACK_PLAIN16);
            pack(lw[0], pp + o.p_w3, 2 * C, C, 3, MG_PACK_GATE16);
            pack(lw[5], pp + o.p_wo, 2 * C, C, 1, MG_PACK_PLAIN16);
            // the [NL * C, H] matrix of all conditioner projections: 32-row blocks are the outermost index of a pack,
            // so the layers' own packs, one behind the other, ARE its pack (C is a multiple of the 128-row tile)
            pack(lw[3], o.wc_all + (size_t)l * mg_conv_packed_floats(C, H, 1, MG_PACK_PLAIN), C, H, 1, MG_PACK_PLAIN);
            copy(o.bc_all + (size_t)l * C, lw[4], C);
        }
    }
    if (rc != MG_OK) return rc;
    {
        // the table travels in the reserved tail of the packed buffer (host memory is pageable: the runtime stages
        // the 30 KB copy, after which `jobs` may be reused)
        PackJob *dev_jobs = reinterpret_cast<PackJob *>(packed + o.jobs);
        if (!(flags & MG_DEN_JOBS_RESIDENT)) {   // (a caller repacking the same tensors into the same buffer skips the copy)
            hipError_t e = hipMemcpyAsync(dev_jobs, jobs, (size_t)n * sizeof(PackJob), hipMemcpyHostToDevice, st);
            if (e != hipSuccess) return (int)e;
        }
        hipLaunchKernelGGL(pack_table_kernel, dim3(blocks), dim3(256), 0, st, dev_jobs, n, packed);
        MG_LAUNCH_CHECK();
    }
    if (flags & MG_DEN_SPLIT) {
        if (C != RB_C || H != RB_C) return MG_ERR_SHAPE;
        for (int l = 0; l < d->n_layers; ++l) {
            const float *const *lw = w + MG_DEN_HEAD_PTRS + (size_t)l * MG_DEN_LAYER_PTRS;
            float *sp = packed + o.slayers + (size_t)l * o.slayer_stride;
            auto pk = [&](const float *src, size_t off, int Co, int Ci, int K, int gate) {
                const int MB = Co / 32;
                const size_t total = (size_t)MB * K * (Ci / 16) * 64 * 8;
                const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
                hipLaunchKernelGGL(pack_split_kernel, dim3(blocks), dim3(256), 0, st, src,
                                   reinterpret_cast<__bf16 *>(sp + off), Co, Ci, K, MB, gate);
            };
            pk(lw[3], o.sl_wc, C, H, 1, 0);
            pk(lw[0], o.sl_w3, 2 * C, C, 3, 1);
            pk(lw[5], o.sl_wo, 2 * C, C, 1, 0);
            MG_LAUNCH_CHECK();
        }
    }
    return MG_OK;
}

extern "C" size_t mg_denoiser_workspace_floats(const mg_denoiser_dims *d, int B, int L, int save_for_backward)
{
    if (den_check(d) != MG_OK || B <= 0 || L <= 0) return 0;
    return den_ws(d, B, L, save_for_backward).total;
}

// ------------------------------------------------------------------------------------------ profiling
// HIP-event brackets around the dominant kernel (the k=3 gated conv), recorded on the launch
// stream inside mg_denoiser_fwd while a profile session is open.  Host-side state only.
#include <vector>
static thread_local std::vector<hipEvent_t> g_prof_ev;  // 2 per bracket
static thread_local int g_prof_used = 0, g_prof_cap = 0, g_prof_every = 1, g_prof_seen = 0, g_prof_open = 0;

// every: bracket only every `every`-th launch of the dominant kernel (1 = all).  An odd stride that is
// coprime with the layer count samples every layer over a few steps while keeping the event
// traffic (and its ~4 % perturbation of the timed region) negligible.
extern "C" int mg_profile_begin_sampled(int max_brackets, int every)
{
    if (max_brackets <= 0 || every <= 0) return MG_ERR_ARG;
    const int rc = mg_profile_begin(max_brackets);
    g_prof_every = every;
    return rc;
}

extern "C" int mg_profile_begin(int max_brackets)
{
    if (max_brackets <= 0) return MG_ERR_ARG;
    g_prof_every = 1;
    g_prof_seen = 0;
    g_prof_open = 0;
    for (hipEvent_t e : g_prof_ev) (void)hipEventDestroy(e);
    g_prof_ev.assign((size_t)2 * max_brackets, nullptr);
    for (auto &e : g_prof_ev) {
        hipError_t rc = hipEventCreate(&e);
        if (rc != hipSuccess) return (int)rc;
    }
    g_prof_cap = max_brackets;
    g_prof_used = 0;
    return MG_OK;
}

// Waits for the recorded events, writes one elapsed time (ms) per bracket; returns the count (<0: error).
extern "C" int mg_profile_end(float *ms_out, int max_out)
{
    int n = g_prof_used < max_out ? g_prof_used : max_out;
    for (int i = 0; i < n; ++i) {
        hipError_t rc = hipEventSynchronize(g_prof_ev[2 * i + 1]);
        if (rc == hipSuccess) rc = hipEventElapsedTime(&ms_out[i], g_prof_ev[2 * i], g_prof_ev[2 * i + 1]);
        if (rc != hipSuccess) { n = -(int)rc; break; }
    }
    for (hipEvent_t e : g_prof_ev) (void)hipEventDestroy(e);
    g_prof_ev.clear();
    g_prof_cap = g_prof_used = 0;
    return n;
}

static inline void prof_mark(hipStream_t st, int which)
{
    if (g_prof_used >= g_prof_cap) return;
    if (which == 0) {
        g_prof_open = (g_prof_seen++ % g_prof_every) == 0;
        if (g_prof_open) (void)hipEventRecord(g_prof_ev[2 * g_prof_used], st);
    } else if (g_prof_open) {
        (void)hipEventRecord(g_prof_ev[2 * g_prof_used + 1], st);
        ++g_prof_used;
        g_prof_open = 0;
    }
}

// ------------------------------------------------------------------------------------------ one fused layer
// Tile width: 64 frames unless that leaves more than ~1/4 of the 256 CUs without a workgroup AND a
// narrow tiling still fits one workgroup per CU (a second round would cost more than it gains).
static void launch_res_layer(ResArgs &a, int B, int L, bool save, bool vec4, hipStream_t st)
{
    const long w64 = (long)mg_cdiv(L, RB_NT) * B, w30 = (long)mg_cdiv(L, 30) * B, w32 = (long)mg_cdiv(L, 32) * B;
    int ntile = w64 > 192 ? 64 : (w30 <= 256 ? 30 : (w32 <= 256 ? 32 : 64));
    if (const char *force = std::getenv("MG_RB_TILE")) {   // tests pin each tile width against the fixtures
        const int f = std::atoi(force);
        if (f == 64 || f == 30 || f == 32) ntile = f;
    }
    a.tiles_per_b = mg_cdiv(L, ntile);
    dim3 grid((unsigned)(a.tiles_per_b * B));
#define MG_RB_LAUNCH(V, S, N) hipLaunchKernelGGL((resblock_fused_kernel<V, S, N>), grid, dim3(512), 0, st, a)
    if (ntile == 64) {
        if (save) {
            if (vec4) MG_RB_LAUNCH(true, true, 64);
            else MG_RB_LAUNCH(false, true, 64);
        } else {
            if (vec4) MG_RB_LAUNCH(true, false, 64);
            else MG_RB_LAUNCH(false, false, 64);
        }
    } else if (ntile == 30) {
        if (save) MG_RB_LAUNCH(false, true, 30);
        else MG_RB_LAUNCH(false, false, 30);
    } else {
        if (save) MG_RB_LAUNCH(false, true, 32);
        else MG_RB_LAUNCH(false, false, 32);
    }
#undef MG_RB_LAUNCH
}

// ResidualBlock.forward stand-alone (model/blocks.py:1157-1176): the same fused kernel on one layer's packed
// weights.  hvec = Wd s (+ Wp spk), dvec = Wd s (computed by the caller with mg_linear_small_fwd); x_out and skip
// are written (not accumulated); the four saves are all given (training) or all NULL.
extern "C" int mg_resblock_fwd(const float *x, const float *cond, const float *wc_packed, const float *w3_packed,
                               const float *wo_packed, const float *bc, const float *b3, const float *bo,
                               const float *hvec, const float *dvec, float *x_out, float *skip, float *h_save,
                               float *g_save, float *sig_save, float *tnh_save, int B, int C, int H, int L, void *stream)
{
    if (!x || !cond || !wc_packed || !w3_packed || !wo_packed || !bc || !b3 || !bo || !hvec || !dvec || !x_out || !skip)
        return MG_ERR_ARG;
    const int nsave = (h_save != nullptr) + (g_save != nullptr) + (sig_save != nullptr) + (tnh_save != nullptr);
    if (nsave != 0 && nsave != 4) return MG_ERR_ARG;
    if (x_out == x) return MG_ERR_ARG;   // neighbouring tiles read each other's halo frames of x
    if (B <= 0 || L <= 0 || C != RB_C || H != RB_C) return MG_ERR_SHAPE;
    ResArgs a;
    a.cond = cond;
    a.x_in = x;
    a.x_out = x_out;
    a.skip = skip;
    a.wc = wc_packed;
    a.w3 = w3_packed;
    a.wo = wo_packed;
    a.bc = bc;
    a.b3 = b3;
    a.bo = bo;
    a.hvec = hvec;
    a.dvec = dvec;
    a.h_save = h_save;
    a.g_save = g_save;
    a.sig_save = sig_save;
    a.tnh_save = tnh_save;
    a.L = L;
    a.first = 1;
    const bool vec4 = (L % 4 == 0) && (((uintptr_t)cond & 15) == 0);
    launch_res_layer(a, B, L, nsave == 4, vec4, (hipStream_t)stream);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ------------------------------------------------------------------------------------------ forward
// the clamp + posterior sample of p_sample (model/diffusion.py:113-129), fused behind the forward when given
struct PostSample {
    const float *coef1, *coef2, *logvar;   // [n_steps] posterior_mean_coef1/2, posterior_log_variance_clipped
    const float *noise;                    // [B, M, L] or NULL (in-kernel Philox)
    unsigned long long seed, noise_stream;
    float *x0_out;                         // optional pre-clamp x_0
    int n_steps, clip;
    const float *cproj;                    // optional precomputed conditioner projections (mg_denoiser_cond_project)
    float *cproj_out;                      // ... or where this launch leaves them for the next ones
    const float *step_vectors;             // optional mg_denoiser_step_vectors output of step_count steps;
    int step_index, step_count;            //   this launch is step step_index of them
};

// ---- host side of the persistent kernels' failure reporting and slot accounting (declared in denoiser_common.h)
static unsigned *g_host_err = nullptr, *g_host_err_dev = nullptr;
static std::atomic<int> g_host_err_state{0};   // 0 untried, 1 ready, -1 unavailable

unsigned *mg_host_err_device_ptr()
{
    int s = g_host_err_state.load();
    if (s == 0) {
        static std::atomic_flag busy = ATOMIC_FLAG_INIT;
        if (busy.test_and_set()) return nullptr;   // another thread is allocating: this launch reports through the workspace only
        void *h = nullptr, *dptr = nullptr;
        // not a stream operation; under stream capture it fails (cleanly) and is retried on the next launch
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) == hipSuccess && h) {
            std::memset(h, 0, 64);
            if (hipHostGetDevicePointer(&dptr, h, 0) == hipSuccess && dptr) {
                g_host_err = (unsigned *)h;
                g_host_err_dev = (unsigned *)dptr;
                g_host_err_state.store(1);
            } else {
                (void)hipHostFree(h);
            }
        }
        (void)hipGetLastError();   // a failed attempt must not surface as a later launch's error
        busy.clear();
        s = g_host_err_state.load();
    }
    return s == 1 ? g_host_err_dev : nullptr;
}

extern "C" unsigned mg_persist_error(int clear)
{
    if (g_host_err_state.load() != 1) {
        (void)mg_host_err_device_ptr();
        if (g_host_err_state.load() != 1) return 0u;
    }
    const unsigned v = __atomic_load_n(g_host_err, __ATOMIC_ACQUIRE);
    if (clear && v) __atomic_store_n(g_host_err, 0u, __ATOMIC_RELEASE);
    return v;
}

int mg_device_cus()
{
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load();
    if (n == 0) {
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus[dev].store(n);
    }
    return n;
}

unsigned mg_persist_spin_limit()
{
    if (const char *e = std::getenv("MG_PERSIST_SPIN_LIMIT")) {   // read per call: a test shrinks it for one launch
        const long v = std::atol(e);
        if (v > 0) return (unsigned)v;
    }
    return DP_SPIN_LIMIT;
}
// tools only (tools/persist_timeline.py): when set, launches with float4 staging run the TIMING instantiation, which
// writes lane 0's cycle stamps [tiles][NL + 2][12] here.  Not declared in the public header.
static unsigned long long *g_persist_dbg = nullptr;
extern "C" void mg_debug_persist_stamps(unsigned long long *device_buffer) { g_persist_dbg = device_buffer; }

// step embedding -> MLP -> per-layer projections (model/modules.py:433-434, blocks.py:1159): fills ws.emb / h1pre / h1 / s
// (kept for the backward in a save workspace) and the per-layer vectors dvec [NL][B][C] (hvec: + speaker projection)
// `rows` = B for one forward; n B for the n steps of a sampling loop at once (t [n][B], the B speaker rows repeating)
struct StepVecAt {
    size_t emb, h1pre, h1, s, dvec, hvec, total;
};
static inline StepVecAt den_stepvec_layout(const mg_denoiser_dims *d, size_t rows)
{
    const size_t C = d->channels, NL = d->n_layers;
    StepVecAt w;
    size_t p = 0;
    auto take = [&](size_t n) {
        size_t at = p;
        p += mg_align_up(n, 64);
        return at;
    };
    w.emb = take(rows * C);
    w.h1pre = take(rows * 4 * C);
    w.h1 = take(rows * 4 * C);
    w.s = take(rows * C);
    w.dvec = take(NL * rows * C);
    w.hvec = d->multi_speaker ? take(NL * rows * C) : w.dvec;
    w.total = p;
    return w;
}

template <class At>
static int den_step_vectors(const mg_denoiser_dims *d, const DenLayout &o, const float *packed, const int64_t *t,
                            const float *spk, float *ws, const At &w, int B, hipStream_t st, int spk_rows = 0)
{
    const int C = d->channels, H = d->cond_channels, NL = d->n_layers;
    const float *lay0 = packed + o.layers;
    hipLaunchKernelGGL(step_embed_kernel, dim3(mg_cdiv(B * (C / 2), 256)), dim3(256), 0, st, t, packed + o.freq,
                       ws + w.emb, B, C);
    MG_LAUNCH_CHECK();
    MG_TRY(small_linear(packed + o.mlp0, 0, ws + w.emb, ws + w.h1, 0, nullptr, 0, ws + w.h1pre, B, 4 * C, C, 1, 1, st));
    MG_TRY(small_linear(packed + o.mlp2, 0, ws + w.h1, ws + w.s, 0, nullptr, 0, nullptr, B, C, 4 * C, 1, 0, st));
    MG_TRY(small_linear(lay0 + o.l_wd, (long)o.layer_stride, ws + w.s, ws + w.dvec, (long)B * C, nullptr, 0, nullptr, B,
                        C, C, NL, 0, st));
    if (d->multi_speaker)
        MG_TRY(small_linear(lay0 + o.l_wp, (long)o.layer_stride, spk, ws + w.hvec, (long)B * C, ws + w.dvec,
                            (long)B * C, nullptr, B, C, H, NL, 0, st, spk_rows));
    return MG_OK;
}

extern "C" size_t mg_denoiser_step_vectors_floats(const mg_denoiser_dims *d, int n, int B)
{
    if (den_check(d) != MG_OK || n <= 0 || B <= 0) return 0;
    return den_stepvec_layout(d, (size_t)n * B).total;
}

// The step-dependent vectors of Denoiser.forward for the n steps of a sampling loop in one set of launches
extern "C" int mg_denoiser_step_vectors(const mg_denoiser_dims *d, const float *packed, const int64_t *t, const float *spk,
                                        float *vectors, size_t vectors_floats, int n, int B, void *stream)
{
    MG_TRY(den_check(d));
    if (!packed || !t || !vectors) return MG_ERR_ARG;
    if (d->multi_speaker && !spk) return MG_ERR_ARG;
    if (n <= 0 || B <= 0) return MG_ERR_SHAPE;
    const StepVecAt w = den_stepvec_layout(d, (size_t)n * B);
    if (vectors_floats < w.total) return MG_ERR_WORKSPACE;
    const DenLayout o = den_layout(d, 0);
    return den_step_vectors(d, o, packed, t, spk, vectors, w, n * B, (hipStream_t)stream, B);
}

static int denoiser_forward(const mg_denoiser_dims *d, const float *packed, const float *x_t, const int64_t *t,
                            const float *cond, const float *spk, float *out, float *ws, size_t ws_floats, int B,
                            int L, int mode, const PostSample *post, void *stream);

extern "C" int mg_denoiser_fwd(const mg_denoiser_dims *d, const float *packed, const float *x_t, const int64_t *t,
                               const float *cond, const float *spk, float *out, float *ws, size_t ws_floats, int B,
                               int L, int mode, void *stream)
{
    return denoiser_forward(d, packed, x_t, t, cond, spk, out, ws, ws_floats, B, L, mode, nullptr, stream);
}

extern "C" int mg_denoiser_psample(const mg_denoiser_dims *d, const float *packed, const float *x_t, const int64_t *t,
                                   const float *cond, const float *spk, const float *coef1, const float *coef2,
                                   const float *logvar, int n_steps, const float *noise, unsigned long long seed,
                                   unsigned long long noise_stream, int clip, float *x_prev, float *x0_out,
                                   const mg_sampling_loop *loop, float *ws, size_t ws_floats, int B, int L, int mode,
                                   void *stream)
{
    const float *cproj = loop ? loop->cproj : nullptr;
    float *cproj_out = loop ? loop->cproj_out : nullptr;
    const float *step_vectors = loop ? loop->step_vectors : nullptr;
    if (step_vectors && (loop->step_count <= 0 || loop->step_index < 0 || loop->step_index >= loop->step_count))
        return MG_ERR_ARG;
    if (!coef1 || !coef2 || !logvar || n_steps <= 0 || !x_prev) return MG_ERR_ARG;
    if (mode & MG_FWD_SAVE) return MG_ERR_ARG;
    if (x_prev == x_t) return MG_ERR_ARG;   // the posterior reads x_t after other tiles have written x_prev
    // the projections belong to the fp32 single-launch kernels (the inference packs); reading and writing exclude each other
    if ((cproj || cproj_out) && (!(mode & MG_FWD_P16) || (mode & MG_FWD_SPLIT))) return MG_ERR_ARG;
    if (cproj && cproj_out) return MG_ERR_ARG;
    const PostSample ps{coef1, coef2, logvar, noise, seed, noise_stream, x0_out, n_steps, clip, cproj, cproj_out, step_vectors,
                        loop ? loop->step_index : 0, loop ? loop->step_count : 0};
    return denoiser_forward(d, packed, x_t, t, cond, spk, x_prev, ws, ws_floats, B, L, mode, &ps, stream);
}

// Wc_l cond + bc_l for all layers as one GEMM: [NL * C, H] x [H, B * L] -> cproj [B, NL * C, L]
extern "C" int mg_denoiser_cond_project(const mg_denoiser_dims *d, const float *packed, const float *cond, float *cproj, int B,
                                        int L, void *stream)
{
    MG_TRY(den_check(d));
    if (!packed || !cond || !cproj) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || d->channels % 128) return MG_ERR_SHAPE;
    const DenLayout o = den_layout(d, MG_DEN_P16);
    const int C = d->channels, H = d->cond_channels, NL = d->n_layers;
    ConvShape s{B, H, L, L, 1, 1, 0, NL * C, 0, 0};
    EpiBiasAct::Params ep{cproj, packed + o.bc_all, nullptr, 1.f, NL * C, MG_ACT_NONE, 0, 0, nullptr, 0.f};
    return conv_launch<EpiBiasAct>(s, cond, nullptr, packed + o.wc_all, ep, (hipStream_t)stream);
}

extern "C" int mg_denoiser_persist_status(const mg_denoiser_dims *d, const float *ws, int B, int L, unsigned *host_out4,
                                          void *stream)
{
    if (den_check(d) != MG_OK || !ws || !host_out4 || B <= 0 || L <= 0) return MG_ERR_ARG;
    const DenWs w = den_ws(d, B, L, 0);
    hipError_t e = hipMemcpyAsync(host_out4, ws + w.sync, 4 * sizeof(unsigned), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return e == hipSuccess ? MG_OK : (int)e;
}

// Philox normal fill + posterior for the launch-per-layer path (same generator and element indexing as the fused tail)
__global__ void psample_tail_kernel(const float *__restrict__ x0, const float *__restrict__ x_t, const int64_t *__restrict__ t,
                                    const float *__restrict__ coef1, const float *__restrict__ coef2,
                                    const float *__restrict__ logvar, const float *__restrict__ noise,
                                    unsigned long long seed, unsigned long long noise_stream,
                                    const unsigned *__restrict__ launch_ctr, float *__restrict__ out,
                                    float *__restrict__ x0_out, int n_steps, int clip, size_t per_sample, size_t n)
{
    const unsigned long long off = (noise_stream << 32) | (launch_ctr ? (unsigned long long)launch_ctr[0] : 0ull);
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (size_t)gridDim.x * blockDim.x) {
        const size_t b = e / per_sample;
        long tb = (long)t[b];
        tb = tb < 0 ? 0 : (tb >= n_steps ? n_steps - 1 : tb);
        const float sg = tb == 0 ? 0.f : __expf(0.5f * logvar[tb]);
        float v = x0[e];
        if (x0_out && x0_out != x0) x0_out[e] = v;
        if (clip) v = fminf(fmaxf(v, -1.f), 1.f);
        const float nz = noise ? noise[e] : dp_normal(seed, off, e);
        out[e] = fmaf(sg, nz, fmaf(coef1[tb], v, coef2[tb] * x_t[e]));
    }
}

__global__ void bump_counter_kernel(unsigned *ctr) { ctr[0] += 1u; }

static int psample_tail(const PostSample &ps, const float *x0, const float *x_t, const int64_t *t, float *out,
                        unsigned *sync, int B, int M, int L, hipStream_t st)
{
    const size_t per = (size_t)M * L, n = per * B;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(psample_tail_kernel, dim3(blocks), dim3(256), 0, st, x0, x_t, t, ps.coef1, ps.coef2, ps.logvar,
                       ps.noise, ps.seed, ps.noise_stream, sync + 2, out, ps.x0_out, ps.n_steps, ps.clip, per, n);
    MG_LAUNCH_CHECK();
    if (!ps.noise) {   // the launch counter is the Philox offset: one fresh stream per call
        hipLaunchKernelGGL(bump_counter_kernel, dim3(1), dim3(1), 0, st, sync + 2);
        MG_LAUNCH_CHECK();
    }
    return MG_OK;
}

extern "C" int mg_denoiser_cond_project(const mg_denoiser_dims *d, const float *packed, const float *cond, float *cproj, int B,
                                        int L, void *stream);

static int denoiser_forward(const mg_denoiser_dims *d, const float *packed, const float *x_t, const int64_t *t,
                            const float *cond, const float *spk, float *out, float *ws, size_t ws_floats, int B,
                            int L, int mode, const PostSample *post, void *stream)
{
    const int save = mode & MG_FWD_SAVE;
    const int split = mode & MG_FWD_SPLIT;
    const int has_p16 = mode & MG_FWD_P16;
    if (split && save) return MG_ERR_ARG;  // the backward consumes fp32 activations
    MG_TRY(den_check(d));
    if (!packed || !x_t || !t || !cond || !out || !ws) return MG_ERR_ARG;
    if (d->multi_speaker && !spk) return MG_ERR_ARG;
    if (B <= 0 || L <= 0) return MG_ERR_SHAPE;
    const DenWs w = den_ws(d, B, L, save);
    if (ws_floats < w.total) return MG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const int C = d->channels, H = d->cond_channels, M = d->mel_bins, NL = d->n_layers;
    const DenLayout o = den_layout(d, (split ? MG_DEN_SPLIT : 0) | (has_p16 ? MG_DEN_P16 : 0));
    const float *lay0 = packed + o.layers;

    static const bool force_generic = std::getenv("MG_DENOISER_GENERIC") != nullptr;
    const bool fused = !force_generic && C == RB_C && H == RB_C;
    // ---- single-launch forward (denoiser_persist.h): inference in exact fp32; an utterance's chain of 32-frame tiles
    // must fit in a quarter of the chip's 512 slots (forward progress with another process on the GPU), and the tag
    // scheme needs >= 3 layers.  MG_DENOISER_PERSIST=0 keeps the launch-per-layer kernels.
    const char *pe = std::getenv("MG_DENOISER_PERSIST");   // read per call: tests pin each path
    const bool no_persist = pe && pe[0] == '0';
    // tile width: 64 frames (8 waves, one workgroup per CU, weights streamed once per 64 frames) when that still gives
    // more than 128 workgroups, else 32 frames (4 waves, two per CU: twice the workgroups for small launches)
    // ... and 16 frames (v_mfma_f32_16x16x4_f32, denoiser_persist16.h) when even the 32-frame tiling leaves half the
    // CUs idle: single utterances, the configs[0] shape.  Inference only; needs the 16-row packs (MG_FWD_P16).
    int nt = (long)mg_cdiv(L, 64) * B > 128 ? 64 : 32;
    if (nt == 32 && has_p16 && !save && (long)mg_cdiv(L, 32) * B <= 128) nt = 16;
    // the SAVING forward (training) on 32-frame tiles that number at most one per CU -- the per-GPU training shard,
    // B=8, L=1000 -- runs as 8 waves of 32 channels instead of 4 of 64: the activation stores of one wave hide behind
    // the MFMAs of its SIMD partner (measured at that shape: -0.2 ms; the non-saving forward is 8 % SLOWER that way --
    // half the MFMAs per fragment read -- and keeps 4 waves)
    bool wide32 = nt == 32 && save && (long)mg_cdiv(L, 32) * B <= 256;
    bool eight64 = false;
    if (const char *ne = std::getenv("MG_PERSIST_NT")) {   // tests pin each width: 16, 32, 64, 328 = 32 frames x 8 waves
        const int f = std::atoi(ne);
        if (f == 32 || f == 64 || f == 328 || (f == 16 && has_p16 && !save)) {
            nt = f == 328 ? 32 : f;
            wide32 = f == 328;
        }
        if (f == 864) {   // 64-frame tiles as eight waves of 32 channels
            nt = 64;
            eight64 = true;
        }
    }
    if (nt == 16 && mg_cdiv(L, 16) > 128) nt = 32;
    if (nt != 32) wide32 = false;
    // 32-frame tiles, not saving, at most one per CU: the one-workgroup-per-CU build (MG_PERSIST_SOLO=0: the two-per-CU one)
    // (and an utterance's chain within a quarter of the slots THAT build has: one per CU)
    bool solo32 = nt == 32 && !wide32 && !save && (long)mg_cdiv(L, 32) * B <= mg_device_cus() &&
                  mg_cdiv(L, 32) <= mg_device_cus() / 4;
    if (const char *se = std::getenv("MG_PERSIST_SOLO")) solo32 = solo32 && se[0] != '0';
    // ... and when even the 16-frame tiles number no more than a quarter of the CUs (one utterance of up to 1024 frames,
    // the configs[0] shape B=4, L<=256): four workgroups per tile, each owning 64 channels (denoiser_team16.h).  The
    // whole grid must be co-resident, one workgroup per CU.  MG_PERSIST_TEAM=0 keeps one workgroup per tile.
    // Teams of 4 while tiles x 4 <= CUs, else of 2 while tiles x 2 <= CUs.  MG_PERSIST_TEAM=0 / 2 / 4: none / pin a size.
    int team = 0;
    if (nt == 16 && !save && w.team != 0) {
        const char *te = std::getenv("MG_PERSIST_TEAM");
        const long tiles16 = (long)mg_cdiv(L, 16) * B;
        const int pin = te ? std::atoi(te) : -1;
        if (pin != 0) {
            if (tiles16 * 4 <= mg_device_cus() && pin != 2) team = 4;
            else if (tiles16 * 2 <= mg_device_cus() && pin != 4) team = 2;
        }
    }
    // 16-frame tiles, one workgroup per tile: the one-workgroup-per-CU build (512 registers per wave) when the launch has
    // at most one tile per CU and an utterance's chain fits in a quarter of THOSE slots, else the two-per-CU build
    bool solo16 = nt == 16 && team == 0 && (long)mg_cdiv(L, 16) * B <= mg_device_cus() && mg_cdiv(L, 16) <= mg_device_cus() / 4;
    if (const char *se = std::getenv("MG_PERSIST_SOLO")) solo16 = solo16 && se[0] != '0';
    const int tiles_per_b = mg_cdiv(L, nt);
    // a quarter of the chip's workgroup slots: one per CU for the builds with one wave per SIMD (64-frame tiles, the
    // 8-wave 32-frame form, 16-frame tiles without teams, the one-per-CU 32-frame build), two for the 4-wave 32-frame form
    const int chain_cap = (nt == 64 || wide32 || solo32 || solo16) ? mg_device_cus() / 4 : mg_device_cus() / 2;
    const bool persist = fused && !no_persist && !split && M <= 96 && NL >= 3 && tiles_per_b <= chain_cap;
    // the step-dependent vectors: this launch's own, unless the caller computed them for its whole sampling loop
    // (mg_denoiser_step_vectors; read in place by the single-launch kernels only)
    const bool own_vectors = !(persist && post && post->step_vectors);
    if (own_vectors) MG_TRY(den_step_vectors(d, o, packed, t, spk, ws, w, B, st));
    if (persist) {
        PersistArgs a;
        a.b_split = 0;
        a.x_t2 = a.hvec2 = a.dvec2 = a.cond2 = nullptr;
        a.out2 = nullptr;
        a.x_t = x_t;
        a.cond = cond;
        a.cproj = post ? post->cproj : nullptr;
        a.cproj_out = post ? post->cproj_out : nullptr;
        a.in_w = packed + o.in_w;
        a.in_b = packed + o.in_b;
        a.layers = lay0;
        a.layer_stride = o.layer_stride;
        a.l_wc = o.l_wc;
        a.l_w3 = o.l_w3;
        a.l_wo = o.l_wo;
        a.l_bc = o.l_bc;
        a.l_b3 = o.l_b3;
        a.l_bo = o.l_bo;
        a.skip_w = packed + o.skip_w;
        a.skip_b = packed + o.skip_b;
        a.out_w = packed + o.out_w;
        a.out_b = packed + o.out_b;
        a.p16layers = packed + o.p16layers;
        a.p16layer_stride = o.p16layer_stride;
        a.p_wc = o.p_wc;
        a.p_w3 = o.p_w3;
        a.p_wo = o.p_wo;
        if (nt == 16) {   // the 16x16x4 forms of the head / tail projections
            a.in_w = packed + o.in_w16;
            a.skip_w = packed + o.skip_w16;
            a.out_w = packed + o.out_w16;
        }
        a.hvec = ws + w.hvec;
        a.dvec = ws + w.dvec;
        a.vec_rows = 0;
        if (!own_vectors) {
            const StepVecAt v = den_stepvec_layout(d, (size_t)post->step_count * B);
            a.hvec = post->step_vectors + v.hvec + (size_t)post->step_index * B * C;
            a.dvec = post->step_vectors + v.dvec + (size_t)post->step_index * B * C;
            a.vec_rows = post->step_count * B;
        }
        a.out = out;
        a.t = t;
        a.coef1 = post ? post->coef1 : nullptr;
        a.coef2 = post ? post->coef2 : nullptr;
        a.logvar = post ? post->logvar : nullptr;
        a.noise = post ? post->noise : nullptr;
        a.seed = post ? post->seed : 0ull;
        a.noise_stream = post ? post->noise_stream : 0ull;
        a.x0_out = post ? post->x0_out : nullptr;
        a.gran = reinterpret_cast<dp_u64 *>(ws + w.gran);
        a.team = team ? reinterpret_cast<dp_u64 *>(ws + w.team) : nullptr;
        a.sync = reinterpret_cast<unsigned *>(ws + w.sync);
        a.host_err = mg_host_err_device_ptr();
        a.spin_limit = mg_persist_spin_limit();
        a.B = B;
        a.L = L;
        a.M = M;
        a.NL = NL;
        a.tiles_per_b = tiles_per_b;
        a.post = post ? 1 : 0;
        a.clip = post ? post->clip : 0;
        a.n_steps = post ? post->n_steps : 1;
        a.rsNL = 1.0f / sqrtf((float)NL);
        a.dbg = g_persist_dbg;
        {
            const char *we = g_persist_dbg ? std::getenv("MG_PERSIST_DBG_WAVE") : nullptr;
            a.dbg_wave = we ? std::atoi(we) & 7 : 0;
            if (a.dbg_wave >= (nt == 32 && !wide32 ? 4 : 8)) a.dbg_wave = 0;
        }
        a.x0_save = save ? ws + w.x0 : nullptr;
        a.y_save = save ? ws + w.y : nullptr;
        a.skip_save = save ? ws + w.skip : nullptr;
        a.h_save = save ? ws + w.h : nullptr;
        a.g_save = save ? ws + w.g : nullptr;
        a.sig_save = save ? ws + w.sig : nullptr;
        a.tnh_save = save ? ws + w.tnh : nullptr;
        a.act_stride = w.act_stride;
        {
            const char *fe = std::getenv("MG_PERSIST_FLAGS");
            a.flags = fe ? std::atoi(fe) : DP_F_ROLES;
        }
        // (16-byte rows for the float4 staging of cond and for the LDS-direct fetches of its projections)
        const bool vec4 = (L % 4 == 0) && (((uintptr_t)cond & 15) == 0) && (((uintptr_t)a.cproj & 15) == 0);
        dim3 grid((unsigned)(tiles_per_b * B));
        prof_mark(st, 0);
#define MG_DP_LAUNCH(NT, V, T, S) hipLaunchKernelGGL((denoiser_persist_kernel<NT, V, T, S>), grid, dim3(NT * 8), 0, st, a)
// a step that stores (CPM 1: a.cproj_out) or reads (CPM 2: a.cproj) its conditioner projections: own instantiations
#define MG_DP_LAUNCH_C(NT, V, T, NWV, CPM) \
    hipLaunchKernelGGL((denoiser_persist_kernel<NT, V, T, false, NWV, CPM>), grid, dim3(NWV * 64), 0, st, a)
#define MG_DP_LAUNCH_R(NT, V, T, NWV)                  \
    do {                                               \
        if (a.cproj) MG_DP_LAUNCH_C(NT, V, T, NWV, 2); \
        else MG_DP_LAUNCH_C(NT, V, T, NWV, 1);         \
    } while (0)
        const bool readp = (a.cproj != nullptr || a.cproj_out != nullptr) && !save;   // (the launch has a loop's buffer)
        if (nt == 16 && team == 4) {
            const dim3 tgrid((unsigned)(tiles_per_b * B * 4));
            if (vec4) hipLaunchKernelGGL((denoiser_team16_kernel<true, 4>), tgrid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((denoiser_team16_kernel<false, 4>), tgrid, dim3(256), 0, st, a);
        } else if (nt == 16 && team == 2) {
            const dim3 tgrid((unsigned)(tiles_per_b * B * 2));
            if (vec4) hipLaunchKernelGGL((denoiser_team16_kernel<true, 2>), tgrid, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((denoiser_team16_kernel<false, 2>), tgrid, dim3(512), 0, st, a);
        } else if (nt == 16 && solo16) {
            if (vec4) hipLaunchKernelGGL((denoiser_persist16_kernel<true, true>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((denoiser_persist16_kernel<false, true>), grid, dim3(256), 0, st, a);
        } else if (nt == 16) {
            if (vec4) hipLaunchKernelGGL((denoiser_persist16_kernel<true, false>), grid, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((denoiser_persist16_kernel<false, false>), grid, dim3(256), 0, st, a);
        } else if (nt == 64 && !eight64) {
            // 64-frame tiles as FOUR waves of 64 channels: one wave per SIMD with the whole 512-entry register file
            // (nobody to share the matrix pipe with, nobody's operands queueing behind an older wave's)
#define MG_DP_LAUNCH4(V, T, S, CPM) \
    hipLaunchKernelGGL((denoiser_persist_kernel<64, V, T, S, 4, CPM>), grid, dim3(256), 0, st, a)
#define MG_DP_LAUNCH4V(T, S, CPM)             \
    do {                                      \
        if (vec4) MG_DP_LAUNCH4(true, T, S, CPM); \
        else MG_DP_LAUNCH4(false, T, S, CPM);     \
    } while (0)
            const bool stamps = g_persist_dbg && vec4;
            if (save) MG_DP_LAUNCH4V(false, true, 0);
            else if (a.cproj) {
                if (stamps) MG_DP_LAUNCH4(true, true, false, 2);
                else MG_DP_LAUNCH4V(false, false, 2);
            } else if (a.cproj_out) MG_DP_LAUNCH4V(false, false, 1);
            else if (stamps) MG_DP_LAUNCH4(true, true, false, 0);
            else MG_DP_LAUNCH4V(false, false, 0);
#undef MG_DP_LAUNCH4V
#undef MG_DP_LAUNCH4
        } else if (nt == 64) {   // MG_PERSIST_NT=864: eight waves of 32 channels (the form up to round 3; A/B and tests)
            if (save) {
                if (vec4) MG_DP_LAUNCH(64, true, false, true);
                else MG_DP_LAUNCH(64, false, false, true);
            } else if (readp) {
                if (vec4) MG_DP_LAUNCH_R(64, true, false, 8);
                else MG_DP_LAUNCH_R(64, false, false, 8);
            } else if (g_persist_dbg && vec4) MG_DP_LAUNCH(64, true, true, false);
            else if (vec4) MG_DP_LAUNCH(64, true, false, false);
            else MG_DP_LAUNCH(64, false, false, false);
        } else if (wide32) {
#define MG_DP_LAUNCH8(V, S) hipLaunchKernelGGL((denoiser_persist_kernel<32, V, false, S, 8>), grid, dim3(512), 0, st, a)
            if (save) {
                if (vec4) MG_DP_LAUNCH8(true, true);
                else MG_DP_LAUNCH8(false, true);
            } else if (readp) {
                if (vec4) MG_DP_LAUNCH_R(32, true, false, 8);
                else MG_DP_LAUNCH_R(32, false, false, 8);
            } else if (vec4) MG_DP_LAUNCH8(true, false);
            else MG_DP_LAUNCH8(false, false);
#undef MG_DP_LAUNCH8
        } else if (solo32) {
            // at most one 32-frame tile per CU: the 4-wave form built for one workgroup per CU (512 registers per wave)
#define MG_DP_LAUNCH_S(V, CPM) \
    hipLaunchKernelGGL((denoiser_persist_kernel<32, V, false, false, 4, CPM, true>), grid, dim3(256), 0, st, a)
            if (a.cproj) {
                if (vec4) MG_DP_LAUNCH_S(true, 2);
                else MG_DP_LAUNCH_S(false, 2);
            } else if (a.cproj_out) {
                if (vec4) MG_DP_LAUNCH_S(true, 1);
                else MG_DP_LAUNCH_S(false, 1);
            } else if (vec4) MG_DP_LAUNCH_S(true, 0);
            else MG_DP_LAUNCH_S(false, 0);
#undef MG_DP_LAUNCH_S
        } else {
            if (save) {
                if (vec4) MG_DP_LAUNCH(32, true, false, true);
                else MG_DP_LAUNCH(32, false, false, true);
            } else if (readp) {
                if (vec4) MG_DP_LAUNCH_R(32, true, false, 4);
                else MG_DP_LAUNCH_R(32, false, false, 4);
            } else if (g_persist_dbg && vec4) MG_DP_LAUNCH(32, true, true, false);
            else if (vec4) MG_DP_LAUNCH(32, true, false, false);
            else MG_DP_LAUNCH(32, false, false, false);
        }
#undef MG_DP_LAUNCH_R
#undef MG_DP_LAUNCH_C
#undef MG_DP_LAUNCH
        prof_mark(st, 1);
        MG_LAUNCH_CHECK();
        return MG_OK;
    }
    // launch-per-layer path: x_0 first (into `out`), the posterior as one more elementwise launch at the end
    if (post && post->cproj_out)   // its kernels project per layer: the projections a later launch may read come from the GEMM
        MG_TRY(mg_denoiser_cond_project(d, packed, cond, post->cproj_out, B, L, stream));
    float *const final_out = out;
    if (post && post->x0_out) out = post->x0_out;   // otherwise x_0 is produced in `out` and overwritten in place
    // input projection + ReLU (model/modules.py:430-431; the second relu is idempotent)
    if (fused && M <= 96) {
        HeadArgs ha{x_t, packed + o.in_w, packed + o.in_b, ws + w.x0, M, L, mg_cdiv(L, RB_NT)};
        hipLaunchKernelGGL(denoiser_head_kernel, dim3((unsigned)(ha.tiles_per_b * B)), dim3(512), 0, st, ha);
        MG_LAUNCH_CHECK();
    } else {
        ConvShape s{B, M, L, L, 1, 1, 0, C, 0, 0};
        EpiBiasAct::Params ep{ws + w.x0, packed + o.in_b, nullptr, 1.f, C, MG_ACT_RELU, 0, 0, nullptr, 0.f};
        MG_TRY(conv_launch<EpiBiasAct>(s, x_t, nullptr, packed + o.in_w, ep, st));
    }
    if (split) {
        if (C != RB_C || H != RB_C) return MG_ERR_SHAPE;
        // conditioner -> frame-major bf16 hi/lo planes (once per call; every layer's tile stages from it)
        __bf16 *condS = reinterpret_cast<__bf16 *>(ws + w.conds);
        hipLaunchKernelGGL(cond_split_kernel, dim3(mg_cdiv(L, 64), B), dim3(256), 0, st, cond, condS, L);
        MG_LAUNCH_CHECK();
        float *xa = ws + w.x0, *xb = ws + w.y, *xc = ws + w.x;
        for (int l = 0; l < NL; ++l) {
            const float *lp = lay0 + (size_t)l * o.layer_stride;
            const float *sp = packed + o.slayers + (size_t)l * o.slayer_stride;
            ResSplitArgs a;
            a.condS = condS;
            a.x_in = xa;
            a.x_out = xb;
            a.skip = ws + w.skip;
            a.wc = reinterpret_cast<const __bf16 *>(sp + o.sl_wc);
            a.w3 = reinterpret_cast<const __bf16 *>(sp + o.sl_w3);
            a.wo = reinterpret_cast<const __bf16 *>(sp + o.sl_wo);
            a.bc = lp + o.l_bc;
            a.b3 = lp + o.l_b3;
            a.bo = lp + o.l_bo;
            a.hvec = ws + w.hvec + (size_t)l * B * C;
            a.dvec = ws + w.dvec + (size_t)l * B * C;
            a.L = L;
            a.tiles_per_b = mg_cdiv(L, RB_NT);
            a.first = (l == 0);
            prof_mark(st, 0);
            hipLaunchKernelGGL(resblock_split_kernel, dim3((unsigned)(a.tiles_per_b * B)), dim3(512), 0, st, a);
            prof_mark(st, 1);
            MG_LAUNCH_CHECK();
            float *tmp = (l == 0) ? xc : xa;
            xa = xb;
            xb = tmp;
        }
    } else if (fused) {
        // one launch per layer (resblock_fused.h); x ping-pongs between ws.x and ws.y
        const bool vec4 = (L % 4 == 0) && (((uintptr_t)cond & 15) == 0);
        float *xa = ws + w.x0, *xb = ws + w.y, *xc = ws + w.x;  // x0 is preserved when saving
        for (int l = 0; l < NL; ++l) {
            const float *lp = lay0 + (size_t)l * o.layer_stride;
            ResArgs a;
            a.cond = cond;
            a.x_in = xa;
            a.x_out = xb;
            a.skip = ws + w.skip;
            a.wc = lp + o.l_wc;
            a.w3 = lp + o.l_w3;
            a.wo = lp + o.l_wo;
            a.bc = lp + o.l_bc;
            a.b3 = lp + o.l_b3;
            a.bo = lp + o.l_bo;
            a.hvec = ws + w.hvec + (size_t)l * B * C;
            a.dvec = ws + w.dvec + (size_t)l * B * C;
            a.h_save = save ? ws + w.h + (size_t)l * w.act_stride : nullptr;
            a.g_save = save ? ws + w.g + (size_t)l * w.act_stride : nullptr;
            a.sig_save = save ? ws + w.sig + (size_t)l * w.act_stride : nullptr;
            a.tnh_save = save ? ws + w.tnh + (size_t)l * w.act_stride : nullptr;
            a.L = L;
            a.first = (l == 0);
            prof_mark(st, 0);
            launch_res_layer(a, B, L, save != 0, vec4, st);
            prof_mark(st, 1);
            MG_LAUNCH_CHECK();
            float *tmp = (l == 0) ? xc : xa;  // after layer 0 the pair is (ws.y, ws.x)
            xa = xb;
            xb = tmp;
        }
    } else {
    if (w.x0 != w.x) MG_TRY(copy_d2d(ws + w.x, ws + w.x0, (size_t)B * C * L, st));
    for (int l = 0; l < NL; ++l) {
        const float *lp = lay0 + (size_t)l * o.layer_stride;
        float *hbuf = ws + w.h + (size_t)l * w.act_stride;
        float *gbuf = ws + w.g + (size_t)l * w.act_stride;
        {
            ConvShape s{B, H, L, L, 1, 1, 0, C, 0, 0};
            EpiCond::Params ep{hbuf, lp + o.l_bc, ws + w.x, ws + w.hvec + (size_t)l * B * C, C};
            MG_TRY(conv_launch<EpiCond>(s, cond, nullptr, lp + o.l_wc, ep, st));
        }
        {
            ConvShape s{B, C, L, L, 3, 1, 1, 2 * C, 0, 0};
            EpiGate::Params ep{gbuf, lp + o.l_b3, save ? ws + w.sig + (size_t)l * w.act_stride : nullptr,
                               save ? ws + w.tnh + (size_t)l * w.act_stride : nullptr, C};
            prof_mark(st, 0);
            MG_TRY(conv_launch<EpiGate>(s, hbuf, nullptr, lp + o.l_w3, ep, st));
            prof_mark(st, 1);
        }
        {
            ConvShape s{B, C, L, L, 1, 1, 0, 2 * C, 0, 0};
            EpiResSkip::Params ep{ws + w.x, ws + w.skip, lp + o.l_bo, ws + w.dvec + (size_t)l * B * C, C, l == 0};
            MG_TRY(conv_launch<EpiResSkip>(s, gbuf, nullptr, lp + o.l_wo, ep, st));
        }
    }
    }
    // sum(skips)/sqrt(NL) -> skip_projection -> ReLU -> output_projection (model/modules.py:441-444)
    if (fused && M <= 128) {
        TailArgs ta{ws + w.skip, packed + o.skip_w, packed + o.skip_b, packed + o.out_w, packed + o.out_b,
                    save ? ws + w.y : nullptr, out, 1.0f / sqrtf((float)NL), M, L, mg_cdiv(L, RB_NT)};
        hipLaunchKernelGGL(denoiser_tail_kernel, dim3((unsigned)(ta.tiles_per_b * B)), dim3(512), 0, st, ta);
        MG_LAUNCH_CHECK();
        return post ? psample_tail(*post, out, x_t, t, final_out, reinterpret_cast<unsigned *>(ws + w.sync), B, M, L, st) : MG_OK;
    }
    {
        ConvShape s{B, C, L, L, 1, 1, 0, C, 0, 0};
        EpiBiasAct::Params ep{ws + w.y, packed + o.skip_b, nullptr, 1.0f / sqrtf((float)NL), C, MG_ACT_RELU, 0, 0, nullptr, 0.f};
        MG_TRY(conv_launch<EpiBiasAct>(s, ws + w.skip, nullptr, packed + o.skip_w, ep, st));
    }
    {
        ConvShape s{B, C, L, L, 1, 1, 0, M, 0, 0};
        EpiBiasAct::Params ep{out, packed + o.out_b, nullptr, 1.f, M, MG_ACT_NONE, 0, 0, nullptr, 0.f};
        MG_TRY(conv_launch<EpiBiasAct>(s, ws + w.y, nullptr, packed + o.out_w, ep, st));
    }
    return post ? psample_tail(*post, out, x_t, t, final_out, reinterpret_cast<unsigned *>(ws + w.sync), B, M, L, st) : MG_OK;
}

// Both generator forwards of a GAN training step (train.py:133 and :153: same weights, different t / noise) as ONE
// launch: problem A = the D phase's no-grad forward, problem B = the G phase's saving forward, Bh utterances each over
// the same conditioner.  wsA: a workspace for (2 Bh, L, no save) -- tickets, halo granules, A's step vectors; wsB: one for
// (Bh, L, save) -- exactly what mg_denoiser_fwd(..., MG_FWD_SAVE) would have filled, so mg_denoiser_bwd consumes it
// unchanged.  64-frame tiles: 2 Bh ceil(L/64) workgroups.  Returns MG_ERR_SHAPE when the single-launch kernel does
// not take the shape (the caller then runs the two forwards separately).
extern "C" int mg_denoiser_fwd_pair(const mg_denoiser_dims *d, const float *packed, const float *x_tA, const int64_t *tA,
                                    const float *x_tB, const int64_t *tB, const float *cond, const float *spk,
                                    const float *condB, const float *spkB, float *outA, float *outB, float *wsA,
                                    size_t wsA_floats, float *wsB, size_t wsB_floats, int Bh, int L, void *stream)
{
    if (!condB) condB = cond;
    if (!spkB) spkB = spk;
    MG_TRY(den_check(d));
    if (!packed || !x_tA || !tA || !x_tB || !tB || !cond || !outA || !outB || !wsA || !wsB) return MG_ERR_ARG;
    if (d->multi_speaker && !spk) return MG_ERR_ARG;
    if (Bh <= 0 || L <= 0) return MG_ERR_SHAPE;
    const int C = d->channels, H = d->cond_channels, M = d->mel_bins, NL = d->n_layers;
    const int tiles_per_b = mg_cdiv(L, 64);
    const char *pe = std::getenv("MG_DENOISER_PERSIST");
    if ((pe && pe[0] == '0') || C != RB_C || H != RB_C || M > 96 || NL < 3 || tiles_per_b > mg_device_cus() / 4) return MG_ERR_SHAPE;
    const DenWs wA = den_ws(d, 2 * Bh, L, 0), wB = den_ws(d, Bh, L, 1);
    if (wsA_floats < wA.total || wsB_floats < wB.total) return MG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const DenLayout o = den_layout(d, 0);
    MG_TRY(den_step_vectors(d, o, packed, tA, spk, wsA, wA, Bh, st));
    MG_TRY(den_step_vectors(d, o, packed, tB, spkB, wsB, wB, Bh, st));
    PersistArgs a;
    a.x_t = x_tA;
    a.x_t2 = x_tB;
    a.out = outA;
    a.out2 = outB;
    a.hvec = wsA + wA.hvec;
    a.dvec = wsA + wA.dvec;
    a.vec_rows = 0;
    a.hvec2 = wsB + wB.hvec;
    a.dvec2 = wsB + wB.dvec;
    a.b_split = Bh;
    a.cond = cond;
    a.cproj = nullptr;
    a.cproj_out = nullptr;
    a.cond2 = condB;
    a.in_w = packed + o.in_w;
    a.in_b = packed + o.in_b;
    a.layers = packed + o.layers;
    a.layer_stride = o.layer_stride;
    a.l_wc = o.l_wc;
    a.l_w3 = o.l_w3;
    a.l_wo = o.l_wo;
    a.l_bc = o.l_bc;
    a.l_b3 = o.l_b3;
    a.l_bo = o.l_bo;
    a.skip_w = packed + o.skip_w;
    a.skip_b = packed + o.skip_b;
    a.out_w = packed + o.out_w;
    a.out_b = packed + o.out_b;
    a.p16layers = nullptr;
    a.p16layer_stride = 0;
    a.p_wc = a.p_w3 = a.p_wo = 0;
    a.t = nullptr;
    a.coef1 = a.coef2 = a.logvar = a.noise = nullptr;
    a.seed = a.noise_stream = 0ull;
    a.x0_out = nullptr;
    a.gran = reinterpret_cast<dp_u64 *>(wsA + wA.gran);
    a.team = nullptr;
    a.sync = reinterpret_cast<unsigned *>(wsA + wA.sync);
    a.host_err = mg_host_err_device_ptr();
    a.spin_limit = mg_persist_spin_limit();
    a.B = 2 * Bh;
    a.L = L;
    a.M = M;
    a.NL = NL;
    a.tiles_per_b = tiles_per_b;
    a.post = 0;
    a.clip = 0;
    a.n_steps = 1;
    a.rsNL = 1.0f / sqrtf((float)NL);
    a.dbg = nullptr;
    a.dbg_wave = 0;
    a.x0_save = wsB + wB.x0;
    a.y_save = wsB + wB.y;
    a.skip_save = wsB + wB.skip;
    a.h_save = wsB + wB.h;
    a.g_save = wsB + wB.g;
    a.sig_save = wsB + wB.sig;
    a.tnh_save = wsB + wB.tnh;
    a.act_stride = wB.act_stride;
    a.flags = 0;
    const bool vec4 = (L % 4 == 0) && (((uintptr_t)cond & 15) == 0) && (((uintptr_t)condB & 15) == 0);
    dim3 grid((unsigned)(tiles_per_b * 2 * Bh));
    const char *ne = std::getenv("MG_PERSIST_NT");
    if (ne && std::atoi(ne) == 864) {   // eight waves of 32 channels (see denoiser_forward)
        if (vec4) hipLaunchKernelGGL((denoiser_persist_kernel<64, true, false, true>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((denoiser_persist_kernel<64, false, false, true>), grid, dim3(512), 0, st, a);
    } else if (vec4) hipLaunchKernelGGL((denoiser_persist_kernel<64, true, false, true, 4>), grid, dim3(256), 0, st, a);
    else hipLaunchKernelGGL((denoiser_persist_kernel<64, false, false, true, 4>), grid, dim3(256), 0, st, a);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
