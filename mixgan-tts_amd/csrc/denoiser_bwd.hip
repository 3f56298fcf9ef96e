// Backward of Denoiser.forward (model/modules.py:420-446; what torch.autograd derives for the
// reference) on gfx950.  Every contraction is an fp32-MFMA GEMM:
//   data gradients  = the forward conv kernel on transposed, tap-flipped packs (MG_PACK_DGRAD);
//   weight gradients = frames are the reduction axis: wgrad_stream_kernel for the residual stack (all taps from one
//                      staged tile pair, bias row sums folded in), wgrad_mfma_kernel for the small projections;
// gate / ReLU / residual derivatives are fused into the data-gradient epilogues.  The 20
// conditioner projections share their input, so their data gradient (K = 20*256) and their weight
// gradient (M = 20*256) run as ONE GEMM each over the stacked per-layer dh.
//
// Per residual layer l (top to bottom), with dout = [dx_{l+1}/sqrt2 ; dS] kept in one [B,2C,L] buffer:
//   dWo = dout (x) g_l                      dz  = gate'(Wo^T dout)        (EpiGateBwd)
//   dW3 = dz (x) shift(h_l)                 dh  = W3^T (*) dz             (EpiDhBwd: dx_l = dh + dx_{l+1}/sqrt2)
#include "denoiser_common.h"
#include "denoiser_bwd_persist.h"
#include <atomic>
#include <cstdlib>

// ------------------------------------------------------------------------------------------ epilogues
struct EpiGateBwd {
    struct Params {
        float *dz;         // [B, 2C, L] slice of dz_all (batch stride dz_bs)
        const float *sig;  // [B, C, L] sigmoid(gate) saved by the forward
        const float *tnh;  // [B, C, L] tanh(filter)
        int C;
        long dz_bs;        // floats between batch elements of dz
    };
    template <int WM, int NNB>
    static __device__ __forceinline__ void run(const Params &p, f32x16 (&acc)[WM][NNB], int b, int mrow0, int l0w,
                                               int lane, int Lout)
    {
        // loads of 8 rows x NNB frames first (clamped addresses), then the math and the predicated stores
        const int h = lane >> 5, c = lane & 31;
        int lc[NNB];
        bool lok[NNB];
#pragma unroll
        for (int j = 0; j < NNB; ++j) {
            const int l = l0w + j * 32 + c;
            lok[j] = l < Lout;
            lc[j] = lok[j] ? l : Lout - 1;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float sv[8][NNB], tv[8][NNB];
                size_t zo[8];
                bool rok[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int rr = half * 8 + r;
                    const int ch = mrow0 + i * 32 + 8 * (rr >> 2) + 4 * h + (rr & 3);
                    rok[r] = ch < p.C;
                    const int cc = rok[r] ? ch : p.C - 1;
                    const size_t so = ((size_t)b * p.C + cc) * Lout;
                    zo[r] = (size_t)b * p.dz_bs + (size_t)cc * Lout;
#pragma unroll
                    for (int j = 0; j < NNB; ++j) {
                        sv[r][j] = p.sig[so + lc[j]];
                        tv[r][j] = p.tnh[so + lc[j]];
                    }
                }
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < NNB; ++j) {
                        const float dg = acc[i][j][half * 8 + r], sg = sv[r][j], t = tv[r][j];
                        if (rok[r] && lok[j]) {
                            p.dz[zo[r] + lc[j]] = dg * t * sg * (1.f - sg);                         // d/d gate pre-activation
                            p.dz[zo[r] + (size_t)p.C * Lout + lc[j]] = dg * sg * (1.f - t * t);   // d/d filter pre-activation
                        }
                    }
            }
        }
    }
};

struct EpiDhBwd {
    struct Params {
        float *dh;       // this layer's [C, L] slice of dh_all [B, NL*C, L]
        long dh_bs;      // NL*C*L
        float *dout;     // [B, 2C, L]; rows < C hold dx_{l+1}/sqrt2 on entry, dx_l/sqrt2 on exit
        int C;
        float *xsave;    // copy of the exit value into this layer's slot of dx_all (batch stride xsave_bs)
        long xsave_bs;
    };
    template <int WM, int NNB>
    static __device__ __forceinline__ void run(const Params &p, f32x16 (&acc)[WM][NNB], int b, int mrow0, int l0w,
                                               int lane, int Lout)
    {
        const int h = lane >> 5, c = lane & 31;
        const float rs2 = 0.70710678118654752440f;
        int lc[NNB];
        bool lok[NNB];
#pragma unroll
        for (int j = 0; j < NNB; ++j) {
            const int l = l0w + j * 32 + c;
            lok[j] = l < Lout;
            lc[j] = lok[j] ? l : Lout - 1;
        }
#pragma unroll
        for (int i = 0; i < WM; ++i) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float xv[8][NNB];
                size_t ho[8], xo[8], so[8];
                bool rok[8];
#pragma unroll
                for (int r = 0; r < 8; ++r) {
                    const int rr = half * 8 + r;
                    const int ch = mrow0 + i * 32 + 8 * (rr >> 2) + 4 * h + (rr & 3);
                    rok[r] = ch < p.C;
                    const int cc = rok[r] ? ch : p.C - 1;
                    ho[r] = (size_t)b * p.dh_bs + (size_t)cc * Lout;
                    xo[r] = ((size_t)b * 2 * p.C + cc) * Lout;
                    so[r] = (size_t)b * p.xsave_bs + (size_t)cc * Lout;
#pragma unroll
                    for (int j = 0; j < NNB; ++j) xv[r][j] = p.dout[xo[r] + lc[j]];
                }
#pragma unroll
                for (int r = 0; r < 8; ++r)
#pragma unroll
                    for (int j = 0; j < NNB; ++j) {
                        const float v = acc[i][j][half * 8 + r];
                        if (rok[r] && lok[j]) {
                            const float nx = (v + xv[r][j]) * rs2;
                            p.dh[ho[r] + lc[j]] = v;
                            p.dout[xo[r] + lc[j]] = nx;
                            p.xsave[so[r] + lc[j]] = nx;
                        }
                    }
            }
        }
    }
};

// thin shim over the exported weight-gradient entry point (kernel lives in conv_api.hip)
struct WgradShape {
    int B, Co, Ci, Ldy, Lx, K, stride, pad;
    long dy_bs, x_bs;
};
static int wgrad_launch(const WgradShape &s, const float *dy, const float *x, const float *xvec, float *dw,
                        float *scratch, float alpha, int accumulate, hipStream_t st)
{
    return mg_conv1d_wgrad_strided(dy, s.dy_bs, x, s.x_bs, xvec, dw, scratch, s.B, s.Co, s.Ci, s.Ldy, s.Lx, s.K, s.stride,
                                   s.pad, alpha, accumulate, st);
}

// db (optional): the bias gradients [G][db_gs] from the same pass over dy
static int wgrad_grouped(int G, const float *dy, long dy_bs, long dy_gs, const float *x, long x_bs, long x_gs, float *dw,
                         long dw_gs, float *scratch, int B, int Co, int Ci, int L, int K, int pad, hipStream_t st,
                         float *db = nullptr, long db_gs = 0)
{
    if (db)
        return mg_conv1d_wgrad_grouped_bias(dy, dy_bs, dy_gs, x, x_bs, x_gs, dw, dw_gs, db, db_gs, scratch, G, B, Co, Ci, L, L,
                                            K, 1, pad, 1.f, 0, st);
    return mg_conv1d_wgrad_grouped(dy, dy_bs, dy_gs, x, x_bs, x_gs, dw, dw_gs, scratch, G, B, Co, Ci, L, L, K, 1, pad, 1.f,
                                   0, st);
}

// out[l][0:C] = top[l*C + c], out[l][C:2C] = bottom[c]   (output-conv bias gradients of all layers)
static __global__ void bias_scatter_kernel(const float *__restrict__ top, const float *__restrict__ bottom,
                                           float *__restrict__ out, int NL, int C)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= NL * 2 * C) return;
    const int l = i / (2 * C), r = i - l * 2 * C;
    out[i] = r < C ? top[l * C + r] : bottom[r - C];
}

static int rowsum(const float *in, long in_bs, int B, int R, int L, float *out_r, float *out_br, float alpha,
                  hipStream_t st)
{
    return mg_rowsum(in, in_bs, B, R, L, out_r, out_br, alpha, 0, st);
}

// ------------------------------------------------------------------------------------------ API
extern "C" size_t mg_denoiser_bwd_workspace_floats(const mg_denoiser_dims *d, int B, int L)
{
    if (den_check(d) != MG_OK || B <= 0 || L <= 0) return 0;
    return den_bws(d, B, L).total;
}

extern "C" int mg_denoiser_bwd(const mg_denoiser_dims *d, const float *packed, const float *g_out, const float *x_t,
                               const float *cond, const float *spk, float *ws, float *bws, size_t bws_floats,
                               float *const *grads, float *d_x_t, float *d_cond, float *d_spk, int B, int L,
                               void *stream)
{
    return mg_denoiser_bwd_staged(d, packed, g_out, x_t, cond, spk, ws, bws, bws_floats, grads, d_x_t, d_cond, d_spk, B, L,
                                  nullptr, stream);
}

extern "C" int mg_denoiser_bwd_staged(const mg_denoiser_dims *d, const float *packed, const float *g_out, const float *x_t,
                                      const float *cond, const float *spk, float *ws, float *bws, size_t bws_floats,
                                      float *const *grads, float *d_x_t, float *d_cond, float *d_spk, int B, int L,
                                      void *conv3_grads_done, void *stream)
{
    MG_TRY(den_check(d));
    if (!packed || !g_out || !x_t || !cond || !ws || !bws || !grads) return MG_ERR_ARG;
    if (d->multi_speaker && !spk) return MG_ERR_ARG;
    if (B <= 0 || L <= 0) return MG_ERR_SHAPE;
    const int C = d->channels, H = d->cond_channels, M = d->mel_bins, NL = d->n_layers;
    const DenLayout o = den_layout(d, MG_DEN_BACKWARD);
    const DenWs w = den_ws(d, B, L, 1);
    const DenBws bw = den_bws(d, B, L);
    if (bws_floats < bw.total) return MG_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *const *lg0 = grads + MG_DEN_HEAD_PTRS;
    auto LG = [&](int l, int j) { return lg0[(size_t)l * MG_DEN_LAYER_PTRS + j]; };
    // batched outputs must be contiguous across layers
    for (int l = 1; l < NL; ++l) {
        if (LG(0, 0) && LG(l, 0) != LG(0, 0) + (size_t)l * 2 * C * C * 3) return MG_ERR_ARG;
        if (LG(0, 1) && LG(l, 1) != LG(0, 1) + (size_t)l * 2 * C) return MG_ERR_ARG;
        if (LG(0, 2) && LG(l, 2) != LG(0, 2) + (size_t)l * C * C) return MG_ERR_ARG;
        if (LG(0, 3) && LG(l, 3) != LG(0, 3) + (size_t)l * C * H) return MG_ERR_ARG;
        if (LG(0, 4) && LG(l, 4) != LG(0, 4) + (size_t)l * C) return MG_ERR_ARG;
        if (LG(0, 5) && LG(l, 5) != LG(0, 5) + (size_t)l * 2 * C * C) return MG_ERR_ARG;
        if (LG(0, 6) && LG(l, 6) != LG(0, 6) + (size_t)l * 2 * C) return MG_ERR_ARG;
        if (d->multi_speaker && LG(0, 7) && LG(l, 7) != LG(0, 7) + (size_t)l * C * H) return MG_ERR_ARG;
    }
    const size_t CL = (size_t)C * L;
    float *dout = bws + bw.dout, *dz_all = bws + bw.dz_all, *dx_all = bws + bw.dx_all, *dh_all = bws + bw.dh_all;
    float *dy = bws + bw.dy;
    float *dx0 = bws + bw.dx0, *scr = bws + bw.scratch;
    const long dz_bs = (long)((size_t)NL * 2 * CL), dx_bs = (long)((size_t)(NL + 1) * CL);
    const float rsNL = 1.0f / sqrtf((float)NL);

    // ---- head: output_projection, ReLU, skip_projection (model/modules.py:441-444) ------------
    {
        ConvShape s{B, M, L, L, 1, 1, 0, C, 0, 0};
        EpiBiasAct::Params ep{dy, nullptr, nullptr, 1.f, C, MG_ACT_NONE, 0, 0, ws + w.y, 0.f};
        MG_TRY(conv_launch<EpiBiasAct>(s, g_out, nullptr, packed + o.out_wT, ep, st));
    }
    if (grads[6]) {
        WgradShape s{B, M, C, L, L, 1, 1, 0, 0, 0};
        MG_TRY(wgrad_launch(s, g_out, ws + w.y, nullptr, grads[6], scr, 1.f, 0, st));
    }
    if (grads[7]) MG_TRY(rowsum(g_out, 0, B, M, L, grads[7], nullptr, 1.f, st));
    {
        ConvShape s{B, C, L, L, 1, 1, 0, C, 0, 0};
        EpiBiasAct::Params ep{dout + CL, nullptr, nullptr, rsNL, C, MG_ACT_NONE, 0, (long)(2 * CL), nullptr, 0.f};
        MG_TRY(conv_launch<EpiBiasAct>(s, dy, nullptr, packed + o.skip_wT, ep, st));
    }
    if (grads[4]) {
        WgradShape s{B, C, C, L, L, 1, 1, 0, 0, 0};
        MG_TRY(wgrad_launch(s, dy, ws + w.skip, nullptr, grads[4], scr, rsNL, 0, st));
    }
    if (grads[5]) MG_TRY(rowsum(dy, 0, B, C, L, grads[5], nullptr, 1.f, st));
    {   // the last layer's x output is unused: dx_NL = 0 (in dout's top half and in slot NL of dx_all)
        hipError_t e = hipMemset2DAsync(dout, 2 * CL * sizeof(float), 0, CL * sizeof(float), B, st);
        if (e != hipSuccess) return (int)e;
        e = hipMemset2DAsync(dx_all + (size_t)NL * CL, (size_t)dx_bs * sizeof(float), 0, CL * sizeof(float), B, st);
        if (e != hipSuccess) return (int)e;
    }

    // ---- residual layers, top to bottom: data gradients only ------------------------------------
    // dout = [dx_l / sqrt2 ; dskip]; slot l+1 of dx_all holds the dx that ENTERS layer l, slot l the one it produces
    // One persistent launch (denoiser_bwd_persist.h) when the shapes allow; MG_DENOISER_PERSIST=0 keeps the two
    // generic conv launches per layer.
    const char *pe = std::getenv("MG_DENOISER_PERSIST");
    const int tiles_per_b = mg_cdiv(L, 32);
    const bool persist = !(pe && pe[0] == '0') && C == RB_C && NL >= 3 && tiles_per_b <= mg_device_cus() / 4;
    if (persist) {
        BwdPersistArgs pa;
        pa.dout = dout;
        pa.sig = ws + w.sig;
        pa.tnh = ws + w.tnh;
        pa.act_stride = w.act_stride;
        pa.blayers = packed + o.blayers;
        pa.blayer_stride = o.blayer_stride;
        pa.bl_woT = o.bl_woT;
        pa.bl_w3T = o.bl_w3T;
        pa.dz_all = dz_all;
        pa.dx_all = dx_all;
        pa.dh_all = dh_all;
        pa.gran = reinterpret_cast<dp_u64 *>(bws + bw.gran);
        pa.sync = reinterpret_cast<unsigned *>(bws + bw.sync);
        pa.host_err = mg_host_err_device_ptr();
        pa.spin_limit = mg_persist_spin_limit();
        pa.B = B;
        pa.L = L;
        pa.NL = NL;
        pa.tiles_per_b = tiles_per_b;
        const bool vec4 = (L % 4 == 0) && (((uintptr_t)dout & 15) == 0);
        dim3 grid((unsigned)(tiles_per_b * B));
        if (vec4) hipLaunchKernelGGL(denoiser_bwd_persist_kernel<true>, grid, dim3(512), 0, st, pa);
        else hipLaunchKernelGGL(denoiser_bwd_persist_kernel<false>, grid, dim3(512), 0, st, pa);
        MG_LAUNCH_CHECK();
    }
    for (int l = persist ? -1 : NL - 1; l >= 0; --l) {
        const float *bp = packed + o.blayers + (size_t)l * o.blayer_stride;
        const float *sig_l = ws + w.sig + (size_t)l * w.act_stride;
        const float *tnh_l = ws + w.tnh + (size_t)l * w.act_stride;
        float *dz_l = dz_all + (size_t)l * 2 * CL;
        {
            ConvShape s{B, 2 * C, L, L, 1, 1, 0, C, 0, 0};
            EpiGateBwd::Params ep{dz_l, sig_l, tnh_l, C, dz_bs};
            MG_TRY(conv_launch<EpiGateBwd>(s, dout, nullptr, bp + o.bl_woT, ep, st));
        }
        {
            ConvShape s{B, 2 * C, L, L, 3, 1, 1, C, dz_bs, 0};
            EpiDhBwd::Params ep{dh_all + (size_t)l * CL, (long)((size_t)NL * CL), dout, C, dx_all + (size_t)l * CL, dx_bs};
            MG_TRY(conv_launch<EpiDhBwd>(s, dz_l, nullptr, bp + o.bl_w3T, ep, st));
        }
    }

    // ---- weight / bias gradients of all residual layers, grouped over the layer axis ----------------------------
    {
        const float *h_all = ws + w.h, *g_all = ws + w.g;
        const long act_gs = (long)w.act_stride;
        // The bias gradients (row sums of dz, of the dx slots and of dskip over batch and frames) come out of the
        // weight-gradient launches that read those rows anyway (mg_conv1d_wgrad_grouped_bias).
        const bool fold3 = LG(0, 0) && LG(0, 1), foldo = LG(0, 5) && LG(0, 6);
        if (LG(0, 0))   // k=3 conv: dW3_l = dz_l (*) h_l;  db3_l = rowsum(dz_l)
            MG_TRY(wgrad_grouped(NL, dz_all, dz_bs, (long)(2 * CL), h_all, (long)CL, act_gs, LG(0, 0), (long)2 * C * C * 3, scr,
                                 B, 2 * C, C, L, 3, 1, st, fold3 ? LG(0, 1) : nullptr, (long)(2 * C)));
        if (LG(0, 1) && !fold3) MG_TRY(rowsum(dz_all, 0, B, NL * 2 * C, L, LG(0, 1), nullptr, 1.f, st));
        if (conv3_grads_done) {   // the largest gradient array is final from here on: its exchange may start
            const hipError_t ee = hipEventRecord((hipEvent_t)conv3_grads_done, st);
            if (ee != hipSuccess) return (int)ee;
        }
        if (LG(0, 5)) {   // output conv: rows < C see dx_l (slot l+1), rows >= C the layer-independent dskip
            MG_TRY(wgrad_grouped(NL, dx_all + CL, dx_bs, (long)CL, g_all, (long)CL, act_gs, LG(0, 5), (long)2 * C * C, scr, B, C,
                                 C, L, 1, 0, st, foldo ? LG(0, 6) : nullptr, (long)(2 * C)));
            MG_TRY(wgrad_grouped(NL, dout + CL, (long)(2 * CL), 0, g_all, (long)CL, act_gs, LG(0, 5) + (size_t)C * C,
                                 (long)2 * C * C, scr, B, C, C, L, 1, 0, st, foldo ? LG(0, 6) + C : nullptr, (long)(2 * C)));
        }
        if (LG(0, 6) && !foldo) {
            MG_TRY(rowsum(dx_all + CL, dx_bs, B, NL * C, L, bws + bw.btop, nullptr, 1.f, st));
            MG_TRY(rowsum(dout + CL, (long)(2 * CL), B, C, L, bws + bw.bbot, nullptr, 1.f, st));
            hipLaunchKernelGGL(bias_scatter_kernel, dim3(mg_cdiv(NL * 2 * C, 256)), dim3(256), 0, st, bws + bw.btop,
                               bws + bw.bbot, LG(0, 6), NL, C);
            MG_LAUNCH_CHECK();
        }
        // d(Wd s)_l = sqrt2 * sum_frames of the dx layer l produced (slot l): [B][NL*C] per-sample sums
        // (the step vector enters h and the residual, model/blocks.py:1166)
        MG_TRY(rowsum(dx_all, dx_bs, B, NL * C, L, nullptr, bws + bw.dd_all, 1.41421356237309504880f, st));
    }

    // ---- input projection + ReLU (model/modules.py:430-431) -----------------------------------
    {
        const size_t n = (size_t)B * CL;
        const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
        hipLaunchKernelGGL(scale_mask_kernel, dim3(blocks), dim3(256), 0, st, dout, (long)(2 * CL), ws + w.x0, dx0,
                           1.41421356237309504880f, (int)CL, n);
        MG_LAUNCH_CHECK();
    }
    if (grads[0]) {
        WgradShape s{B, C, M, L, L, 1, 1, 0, 0, 0};
        MG_TRY(wgrad_launch(s, dx0, x_t, nullptr, grads[0], scr, 1.f, 0, st));
    }
    if (grads[1]) MG_TRY(rowsum(dx0, 0, B, C, L, grads[1], nullptr, 1.f, st));
    if (d_x_t) {
        ConvShape s{B, C, L, L, 1, 1, 0, M, 0, 0};
        EpiBiasAct::Params ep{d_x_t, nullptr, nullptr, 1.f, M, MG_ACT_NONE, 0, 0, nullptr, 0.f};
        MG_TRY(conv_launch<EpiBiasAct>(s, dx0, nullptr, packed + o.in_wT, ep, st));
    }

    // ---- conditioner projections, all layers at once -----------------------------------------
    if (d_cond) {
        ConvShape s{B, NL * C, L, L, 1, 1, 0, H, 0, 0};
        EpiBiasAct::Params ep{d_cond, nullptr, nullptr, 1.f, H, MG_ACT_NONE, 0, 0, nullptr, 0.f};
        MG_TRY(conv_launch<EpiBiasAct>(s, dh_all, nullptr, packed + o.wc_allT, ep, st));
    }
    // single-speaker: the bias gradients rowsum(dh) ride on the weight-gradient launch; multi-speaker needs the
    // per-sample sums too (the speaker vector enters h), which stay one row-sum launch that yields both
    const bool foldc = LG(0, 3) && LG(0, 4) && !d->multi_speaker;
    if (LG(0, 3)) {
        if (foldc) {
            MG_TRY(mg_conv1d_wgrad_grouped_bias(dh_all, 0, 0, cond, 0, 0, LG(0, 3), 0, LG(0, 4), 0, scr, 1, B, NL * C, H, L, L, 1,
                                                1, 0, 1.f, 0, st));
        } else {
            WgradShape s{B, NL * C, H, L, L, 1, 1, 0, 0, 0};
            MG_TRY(wgrad_launch(s, dh_all, cond, nullptr, LG(0, 3), scr, 1.f, 0, st));
        }
    }
    if ((LG(0, 4) && !foldc) || d->multi_speaker)
        MG_TRY(rowsum(dh_all, 0, B, NL * C, L, LG(0, 4), d->multi_speaker ? bws + bw.dhv_all : nullptr, 1.f, st));

    // ---- step-embedding MLP and per-layer step / speaker projections (tiny, per sample) --------
    const float *lay0 = packed + o.layers;
    auto outer = [&](const float *a, long a_zs, long a_bs, const float *c, float *out, int Z, int N, int K) {
        const size_t n = (size_t)Z * N * K;
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        hipLaunchKernelGGL(small_outer_kernel, dim3(blocks), dim3(256), 0, st, a, a_zs, a_bs, c, out, Z, B, N, K);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? MG_OK : (int)e;
    };
    auto linear_t = [&](const float *W, long w_zs, const float *a, long a_zs, long a_bs, float *out, int Z, int N,
                        int K) {
        if (Z > 1) {
            const hipError_t em = hipMemsetAsync(out, 0, (size_t)B * K * sizeof(float), st);
            if (em != hipSuccess) return (int)em;
        }
        hipLaunchKernelGGL(small_linear_t_kernel, dim3(mg_cdiv(K, 64), B, Z), dim3(256), 0, st, W, w_zs, a, a_zs, a_bs,
                           out, Z, B, N, K);
        const hipError_t e = hipGetLastError();
        return e == hipSuccess ? MG_OK : (int)e;
    };
    const float *dd_all = bws + bw.dd_all;
    if (LG(0, 2)) MG_TRY(outer(dd_all, C, (long)NL * C, ws + w.s, LG(0, 2), NL, C, C));
    MG_TRY(linear_t(lay0 + o.l_wd, (long)o.layer_stride, dd_all, C, (long)NL * C, bws + bw.ds, NL, C, C));
    if (d->multi_speaker) {
        const float *dhv = bws + bw.dhv_all;  // [B, NL*C]
        if (LG(0, 7)) MG_TRY(outer(dhv, C, (long)NL * C, spk, LG(0, 7), NL, C, H));
        if (d_spk) MG_TRY(linear_t(lay0 + o.l_wp, (long)o.layer_stride, dhv, C, (long)NL * C, d_spk, NL, C, H));
    }
    if (grads[3]) MG_TRY(outer(bws + bw.ds, 0, C, ws + w.h1, grads[3], 1, C, 4 * C));
    MG_TRY(linear_t(packed + o.mlp2, 0, bws + bw.ds, 0, C, bws + bw.dm, 1, C, 4 * C));
    hipLaunchKernelGGL(mish_bwd_kernel, dim3(mg_cdiv(B * 4 * C, 256)), dim3(256), 0, st, bws + bw.dm, ws + w.h1pre,
                       bws + bw.da, B * 4 * C);
    MG_LAUNCH_CHECK();
    if (grads[2]) MG_TRY(outer(bws + bw.da, 0, 4 * C, ws + w.emb, grads[2], 1, 4 * C, C));
    return MG_OK;
}
