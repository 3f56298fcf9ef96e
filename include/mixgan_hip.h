/*
 * mixgan_hip.h -- C ABI of libmixgan_hip.so: the MI355X (gfx950) implementation of the
 * MixGAN-TTS diffusion hot path.
 *
 * The reference has no FFI for this path: its boundary is the Python nn.Module surface
 * (SURVEY.md section 8b).  Each entry point below names the reference function it stands in
 * for (file:line relative to the reference tree); the host-side mirror classes under
 * mixgan-tts_amd/ keep the reference's names, signatures and state_dict keys and call these
 * through ctypes.  INTEGRATION.md shows the binding a reference maintainer would add.
 *
 * Conventions
 *   - all tensors are dense fp32 (timesteps int64), caller-owned device memory;
 *   - the library never allocates device memory, never synchronises the host and enqueues
 *     only on `stream` (a hipStream_t passed as void*), so every call is hipGraph-capturable;
 *   - return 0 on success, <0 for argument/shape errors (MG_ERR_*), >0 = hipError_t;
 *   - sequence tensors are channel-major [B, C, L] (frames contiguous) unless stated.
 */
#ifndef MIXGAN_HIP_H
#define MIXGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_ERR_ARG (-1)       /* null pointer / bad enum */
#define MG_ERR_SHAPE (-2)     /* dimension the kernels do not support */
#define MG_ERR_WORKSPACE (-3) /* workspace too small */

#define MG_VERSION 100

int mg_version(void);
/* Human-readable text for a return code (negative: ours; positive: hipGetErrorString). */
const char *mg_error_string(int code);

/* ------------------------------------------------------------------ activations */
#define MG_ACT_NONE 0
#define MG_ACT_RELU 1
#define MG_ACT_LRELU02 2 /* F.leaky_relu(x, 0.2): model/mixgantts.py:273,282,286 */
#define MG_ACT_TANH 3
#define MG_ACT_LRELU 4 /* leaky ReLU with a caller-given slope (HiFi-GAN: hifigan/models.py:7,95,150,161) */

/* ------------------------------------------------------------------ Conv1d / Linear as MFMA GEMM
 * Stands in for every nn.Conv1d / nn.Linear on the path (ConvNorm model/blocks.py:364-371,
 * LinearNorm model/blocks.py:289-291, transformer/SubLayers.py:67-80, transformer/Layers.py:67-137).
 *
 * Weights are consumed in a packed, MFMA-fragment-ordered layout that is a *derived cache* of
 * the stored [Co, Ci, K] parameter (checkpoints keep the reference layout).
 */
#define MG_PACK_PLAIN 0  /* rows in natural order */
#define MG_PACK_GATE 1   /* rows interleaved so that channel c and c+Co/2 share a lane (GLU gate) */
#define MG_PACK_DGRAD 2  /* transposed + tap-flipped: the data-gradient convolution (stride 1) */
#define MG_PACK_TPOSE 3  /* internal to mg_conv_transpose_pack (polyphase ConvTranspose1d); rejected by mg_conv_pack */
#define MG_PACK_PLAIN16 4 /* as PLAIN, fragments of v_mfma_f32_16x16x4_f32 (16-row blocks, 16-channel k-groups); K <= 3 */
#define MG_PACK_GATE16 5  /* as GATE for the 16x16x4 form: per 32 channels two gate 16-row blocks, then two filter ones */

/* Number of floats of the packed form of a [Co, Ci, K] weight. */
size_t mg_conv_packed_floats(int Co, int Ci, int K, int mode);
/* w: [Co, Ci, K] fp32 -> packed. */
int mg_conv_pack(const float *w, float *packed, int Co, int Ci, int K, int mode, void *stream);

/* out[b, co, l] = act(alpha * sum_{ci,k} w[co,ci,k] * (in[b, ci, l*stride + k - pad] + in_vec[b,ci])
 *                     + bias[co]) (+ add[b,co,l]) (+ out_old if accumulate)
 * in: [B, Ci, Lin]; out/add: [B, Co, Lout]; bias may be NULL; in_vec ([B, Ci], added to in-range
 * input positions only, i.e. before zero padding) may be NULL.  K in {1,3,5,9}, stride in {1,2}. */
int mg_conv1d_fwd(const float *in, const float *in_vec, const float *packed, const float *bias,
                  const float *add, float *out, int B, int Ci, int Lin, int Co, int Lout, int K,
                  int stride, int pad, int act, float alpha, int accumulate, void *stream);

/* As mg_conv_pack, but writes k-groups [q0, q0+Q) of a packed buffer holding Qtot k-groups per
 * 32-row block: concatenates weights that share their rows along the reduction axis. */
int mg_conv_pack_at(const float *w, float *packed, int Co, int Ci, int K, int mode, int q0, int Qtot,
                    void *stream);

/* Weight gradient of the same convolution (autograd of nn.Conv1d / nn.Linear weights):
 *   dw[co,ci,k] (+)= alpha * sum_{b,l} dy[b,co,l] * (x[b,ci,l*stride+k-pad] + x_vec[b,ci])
 * dy [B,Co,Ldy], x [B,Ci,Lx], dw [Co,Ci,K]; scratch: mg_conv1d_wgrad_scratch_floats() floats
 * (per-split partial tiles, at most 512 x 128 x 128 floats = 33.5 MB; written then summed in a fixed order, so
 * the result is reproducible run to run; needs no initialisation). */
size_t mg_conv1d_wgrad_scratch_floats(int Co, int Ci, int K);
int mg_conv1d_wgrad(const float *dy, const float *x, const float *x_vec, float *dw, float *scratch,
                    int B, int Co, int Ci, int Ldy, int Lx, int K, int stride, int pad, float alpha,
                    int accumulate, void *stream);

/* Same with explicit batch strides (floats; 0 = dense) so dy / x may be channel slices of wider tensors. */
int mg_conv1d_wgrad_strided(const float *dy, long dy_bs, const float *x, long x_bs, const float *x_vec,
                            float *dw, float *scratch, int B, int Co, int Ci, int Ldy, int Lx, int K,
                            int stride, int pad, float alpha, int accumulate, void *stream);

/* G weight gradients of one shape in a single launch: group g uses dy + g*dy_gs, x + g*x_gs (a group stride of 0
 * shares that operand) and writes dw + g*dw_gs (0 = Co*Ci*K).  mg_denoiser_bwd computes the gradients of all
 * 20 residual layers' k=3 and output convolutions this way once their inputs are complete: with G*tiles >= 512
 * workgroups no frame split is needed and the combine pass degenerates to the layout transpose. */
size_t mg_conv1d_wgrad_grouped_scratch_floats(int Co, int Ci, int K, int G);
int mg_conv1d_wgrad_grouped(const float *dy, long dy_bs, long dy_gs, const float *x, long x_bs, long x_gs, float *dw,
                            long dw_gs, float *scratch, int G, int B, int Co, int Ci, int Ldy, int Lx, int K, int stride,
                            int pad, float alpha, int accumulate, void *stream);
/* ... and the bias gradients with it: db[g][co] (+)= alpha * sum_{b,l} dy_g[b,co,l] (group stride db_gs, 0 -> Co).
 * The streaming kernel (large stride-1 "same" shapes) sums the dY rows while it stages them; other shapes fall back
 * to one mg_rowsum launch per group.  Same scratch as mg_conv1d_wgrad_grouped. */
int mg_conv1d_wgrad_grouped_bias(const float *dy, long dy_bs, long dy_gs, const float *x, long x_bs, long x_gs, float *dw,
                                 long dw_gs, float *db, long db_gs, float *scratch, int G, int B, int Co, int Ci, int Ldy,
                                 int Lx, int K, int stride, int pad, float alpha, int accumulate, void *stream);

/* Row sums of in [B,R,L] (batch stride in_bs floats, 0 = dense): bias gradients and per-sample
 * channel sums.  out_r[r] (+)= alpha*sum_{b,l} (may be NULL); out_br[b,r] = alpha*sum_l (may be NULL). */
int mg_rowsum(const float *in, long in_bs, int B, int R, int L, float *out_r, float *out_br,
              float alpha, int accumulate, void *stream);

/* Extended form for the vocoder (hifigan/models.py): K additionally in {4,7,11,16},
 * dilation dil <= 5 (input sample l*stride + k*dil - pad), a leaky ReLU of slope in_slope applied to the
 * input samples while they are staged (1 = identity; pre-activation of ResBlock convs), and
 * act = MG_ACT_LRELU with slope act_slope. */
int mg_conv1d_fwd_ex(const float *in, const float *in_vec, const float *packed, const float *bias,
                     const float *add, float *out, int B, int Ci, int Lin, int Co, int Lout, int K,
                     int stride, int pad, int dil, float in_slope, int act, float act_slope, float alpha,
                     int accumulate, void *stream);
/* Same, with a split reduction for outputs of few tiles and deep reductions (the JCU discriminator's tail and its data
 * gradients, model/mixgantts.py:219-248: 512 channels x 5 taps into 128 channels at L/4 frames is 32 output tiles on 256
 * CUs): when `scratch` is given and the 128 x 128-tile grid would leave most CUs idle, the input-channel chunks of a tile
 * are dealt to up to 8 workgroups, which leave partial tiles in the scratch; a second, small launch adds them in split
 * order and applies bias / activation / add.  scratch: fp32, one per stream in flight; 12 Mi floats cover every shape of
 * the path.  Without scratch, or when a split does not pay, identical to mg_conv1d_fwd_ex. */
int mg_conv1d_fwd_split(const float *in, const float *in_vec, const float *packed, const float *bias,
                        const float *add, float *out, int B, int Ci, int Lin, int Co, int Lout, int K, int stride,
                        int pad, int dil, float in_slope, int act, float act_slope, float alpha, int accumulate,
                        float *scratch, size_t scratch_floats, void *stream);

/* ------------------------------------------------------------------ diffusion algebra (HBM-bound)
 * Schedule tables are the fp32 buffers of GaussianDiffusion (model/diffusion.py:60-83). */

/* diffuse_fn + q_sample (model/diffusion.py:177-185, 147-153), fused with norm_spec (:228),
 * the [B,L,M] -> [B,1,M,L] transpose and the mask multiply of :206-207.
 *   mel [B, L, M]; t int64 [B] (t<0 rows return the clean normalised mel); noise [B, M, L];
 *   keep uint8 [B, L] (1 = valid frame) or NULL; out [B, M, L].                                   */
int mg_diffuse_fwd(const float *mel, const int64_t *t, const float *noise, const uint8_t *keep,
                   const float *spec_min, const float *spec_max, const float *sqrt_ac,
                   const float *sqrt_1mac, float *out, int B, int L, int M, int T, void *stream);

/* q_posterior_sample (model/diffusion.py:104-119) with the clamp_ of :126-127 / :211-212 and the
 * mask multiply of :220 fused:  x0c = clamp(x0 (* keep), -1, 1) if clip;
 *   out = (coef1[t] x0c + coef2[t] x_t + [t != 0] exp(0.5 logvar[t]) noise) (* keep).
 * If x0_clamped_out != NULL the masked+clamped x0 is written there (training returns it).
 * Tensors [B, M, L]; keep uint8 [B, L] or NULL (p_sample applies no mask). */
int mg_posterior_sample_fwd(const float *x0, const float *x_t, const int64_t *t, const float *noise,
                            const uint8_t *keep, const float *coef1, const float *coef2,
                            const float *logvar, float *out, float *x0_clamped_out, int clip, int B,
                            int L, int M, int T, void *stream);

/* Gradient of mg_posterior_sample_fwd w.r.t. x0 (everything else on the path is data):
 *   g_x0 = keep * 1[-1 <= x0*keep <= 1 or !clip] * (g_x0c + coef1[t] * keep * g_xpp).
 * g_x0c or g_xpp may be NULL (not both). */
int mg_posterior_sample_bwd(const float *x0, const int64_t *t, const uint8_t *keep,
                            const float *coef1, const float *g_x0c, const float *g_xpp, float *g_x0,
                            int clip, int B, int L, int M, int T, void *stream);

/* norm_spec / denorm_spec (model/diffusion.py:228-232) on a flat [n/M, M] tensor:
 * mode 1 norm, 2 denorm, 3 gradient of norm (g * 2/(max-min)), 4 gradient of denorm. */
int mg_spec_affine(const float *in, float *out, const float *spec_min, const float *spec_max,
                   int mode, size_t n, int M, void *stream);

/* [B, M, L] <-> [B, L, M] transposes with optional norm/denorm_spec (model/diffusion.py:228-232):
 * mode 0 plain, 1 norm_spec on the way in (BLM->BML), 2 denorm_spec on the way out (BML->BLM). */
int mg_transpose_bml(const float *in, float *out, const float *spec_min, const float *spec_max,
                     const uint8_t *keep, int to_blm, int mode, int B, int L, int M, void *stream);
/* Same, with the [B,M,L] side being an M-channel slice of a wider tensor (batch stride bml_bs floats):
 * builds torch.cat([x_t_prevs, x_ts], -1).transpose(1,2) (model/mixgantts.py:262-264) without a copy. */
int mg_transpose_bml_strided(const float *in, float *out, const float *spec_min, const float *spec_max,
                             const uint8_t *keep, int to_blm, int mode, int B, int L, int M, long bml_bs,
                             void *stream);

/* ------------------------------------------------------------------ Denoiser (model/modules.py:382-446)
 * Weight pointer table order for mg_denoiser_pack (names are the reference state_dict keys under
 * `diffusion.denoise_fn.`):
 *   [0] input_projection.0.conv.weight [C,M,1]   [1] input_projection.0.conv.bias [C]
 *   [2] mlp.0.linear.weight [4C,C]               [3] mlp.2.linear.weight [C,4C]
 *   [4] skip_projection.conv.weight [C,C,1]      [5] skip_projection.conv.bias [C]
 *   [6] output_projection.conv.weight [M,C,1]    [7] output_projection.conv.bias [M]
 *   then per residual layer i (8 + 9*i + j):
 *   j=0 conv_layer.conv.weight [2C,C,3]          j=1 conv_layer.conv.bias [2C]
 *   j=2 diffusion_projection.linear.weight [C,C] j=3 conditioner_projection.conv.weight [C,H,1]
 *   j=4 conditioner_projection.conv.bias [C]     j=5 output_projection.conv.weight [2C,C,1]
 *   j=6 output_projection.conv.bias [2C]         j=7 speaker_projection.linear.weight [C,H] or NULL
 *   j=8 reserved (NULL)
 */
typedef struct {
    int32_t n_layers;      /* model.denoiser.residual_layers (20) */
    int32_t channels;      /* C: residual_channels (256)          */
    int32_t cond_channels; /* H: transformer.encoder_hidden (256) */
    int32_t mel_bins;      /* M: n_mel_channels (80)              */
    int32_t multi_speaker; /* 0/1                                 */
} mg_denoiser_dims;

#define MG_DEN_HEAD_PTRS 8
#define MG_DEN_LAYER_PTRS 9

/* flags for mg_denoiser_packed_floats / mg_denoiser_pack */
#define MG_DEN_BACKWARD 1 /* also pack the transposed (data-gradient) forms mg_denoiser_bwd consumes */
#define MG_DEN_SPLIT 2    /* also pack hi/lo bf16 pairs for the split-precision forward */
#define MG_DEN_P16 4      /* also pack the 16x16x4-MFMA forms: the 16-frame tile width of the single-launch forward */
#define MG_DEN_JOBS_RESIDENT 8 /* mg_denoiser_pack only: the previous call had the same weight pointers, flags and `packed`
                                * buffer -- its job table is still in the buffer's tail, skip the host-to-device copy
                                * (the one host transfer of a training step: without it the step is hipGraph-capturable) */
/* flags for mg_denoiser_fwd's `mode` */
#define MG_FWD_SAVE 1     /* keep per-layer activations for mg_denoiser_bwd (fp32 path only) */
#define MG_FWD_SPLIT 2    /* residual-layer GEMMs as 3-term bf16-split MFMA products (fp32-grade, ~1e-5) */
#define MG_FWD_P16 4      /* `packed` was built with MG_DEN_P16: small launches may use the 16-frame tile width */
size_t mg_denoiser_packed_floats(const mg_denoiser_dims *d, int flags);
/* freq: the C/2 step-embedding frequencies exp(-i ln(1e4)/(C/2-1)) (model/blocks.py:909-910),
 * computed by the host exactly as the reference does and cached in the packed blob. */
int mg_denoiser_pack(const mg_denoiser_dims *d, const float *const *weights, const float *freq,
                     float *packed, int flags, void *stream);
/* Workspace (floats) for a forward of batch B, L frames.  save_for_backward additionally keeps the
 * per-layer activations that mg_denoiser_bwd consumes. */
size_t mg_denoiser_workspace_floats(const mg_denoiser_dims *d, int B, int L, int save_for_backward);

/* Denoiser.forward: x_t [B, M, L] (the reference's [B,1,M,L]), t int64 [B], cond [B, H, L],
 * spk [B, H] or NULL -> out [B, M, L].  */
int mg_denoiser_fwd(const mg_denoiser_dims *d, const float *packed, const float *x_t,
                    const int64_t *t, const float *cond, const float *spk, float *out,
                    float *workspace, size_t workspace_floats, int B, int L, int mode, void *stream);

/* The workspace's first use must find its 64-float counter block zeroed (allocate it zero-filled once; the kernels
 * re-arm the counters themselves, also under hipGraph replay).
 *
 * p_sample (model/diffusion.py:121-129) as one call: x_0 = Denoiser.forward(x_t, t, cond, spk); clamp to [-1, 1] when
 * clip; x_prev = coef1[t] x_0 + coef2[t] x_t + (t > 0) exp(0.5 logvar[t]) * noise  (q_posterior + q_posterior_sample,
 * :104-119).  coef1 / coef2 / logvar: the posterior_mean_coef1 / posterior_mean_coef2 / posterior_log_variance_clipped
 * buffers [n_steps].  noise [B, M, L], or NULL: N(0,1) from Philox4x32-10 (the reference draws torch.randn_like per
 * step, model/diffusion.py:32-35,118): key = `seed`, counter = (element index, noise_stream << 32 | launches completed
 * on this workspace).  The launch count lives in the workspace, so every call and every replay of a captured graph
 * draws fresh noise; `noise_stream` (its low 32 bits) must be unique per workspace instance within a process -- the
 * caller numbers its workspaces -- so that two workspaces (another shape, a re-allocated one, another captured graph)
 * never walk the same stream; ranks use different seeds.  x_prev [B, M, L] must not alias x_t; x0_out (optional)
 * receives the pre-clamp x_0.  On the fp32 inference path this is ONE kernel launch.
 *
 *
 * loop (optional): what the T steps of one sampling loop (model/diffusion.py:133-147) share -- see mg_sampling_loop.
 * Results are bit-identical to loop == NULL. */
typedef struct mg_sampling_loop {
    /* cproj / cproj_out (fp32 MG_FWD_P16 packs only, at most one of them): the conditioner projections of all layers,
     * [B, n_layers * channels, L].  The steps of a loop call the denoiser with the SAME cond, and
     * conditioner_projection(cond) (model/blocks.py:1160) depends on neither x_t nor t: the first step passes cproj_out
     * and leaves what it computed there, the steps behind it pass that buffer as cproj and skip the projections -- 11 %
     * of a step's multiply-adds.  The kernels form fl(P + fl(x + step)) either way, P = b_c with the products W_c cond
     * added onto it in channel order -- a function of cond alone.  (mg_denoiser_cond_project fills the same buffer without
     * a step; it adds b_c behind the sum, so its P may differ from a kernel's in the last bit.) */
    const float *cproj;
    float *cproj_out;
    /* step_vectors: mg_denoiser_step_vectors' output for all step_count steps of the loop (a loop knows its t values
     * in advance); this call is step step_index of them and reads its slice in place instead of launching the step
     * embedding, its MLP and the per-layer diffusion / speaker projections (model/modules.py:433-435,
     * model/blocks.py:1159-1163) itself.  `t` must still be this step's t (the posterior reads it). */
    const float *step_vectors;
    int step_index, step_count;
} mg_sampling_loop;
int mg_denoiser_psample(const mg_denoiser_dims *d, const float *packed, const float *x_t, const int64_t *t,
                        const float *cond, const float *spk, const float *coef1, const float *coef2,
                        const float *logvar, int n_steps, const float *noise, unsigned long long seed,
                        unsigned long long noise_stream, int clip, float *x_prev, float *x0_out,
                        const mg_sampling_loop *loop, float *workspace, size_t workspace_floats, int B, int L, int mode,
                        void *stream);
/* The step-dependent vectors of Denoiser.forward for n steps at once: t [n, B] (step-major), spk [B, H] (the same
 * utterances in every step; NULL unless multi_speaker).  For each step: diffusion_embedding(t) -> mlp
 * (model/modules.py:433-434), then per residual layer diffusion_projection(.) and, multi-speaker,
 * + speaker_projection(spk) (model/blocks.py:1159-1163).  `vectors`: mg_denoiser_step_vectors_floats(d, n, B) floats,
 * opaque; pass it to each step's mg_denoiser_psample through mg_sampling_loop.  Same values bit for bit as the vectors
 * a step computes for itself. */
size_t mg_denoiser_step_vectors_floats(const mg_denoiser_dims *d, int n, int B);
int mg_denoiser_step_vectors(const mg_denoiser_dims *d, const float *packed, const int64_t *t, const float *spk,
                             float *vectors, size_t vectors_floats, int n, int B, void *stream);
/* cproj[b, l * channels + c, :] = conditioner_projection_l(cond[b])[c, :] (model/blocks.py:1150,1160: Conv1d(H, C, 1) with
 * bias) for every residual layer l, as one [n_layers * channels, H] x [H, B * L] product.  `packed` must have been built
 * with MG_DEN_P16 (channels == cond_channels == 256). */
int mg_denoiser_cond_project(const mg_denoiser_dims *d, const float *packed, const float *cond, float *cproj, int B,
                             int L, void *stream);
/* Both generator forwards of a GAN training step (train.py:133 and :153 -- same weights, different t / noise) as ONE
 * launch of 64-frame tiles: problem A (x_tA, tA -> outA; nothing kept) is the D phase's no-grad forward, problem B
 * (x_tB, tB -> outB) the G phase's: its layer activations land in wsB exactly as mg_denoiser_fwd(MG_FWD_SAVE) leaves
 * them, so mg_denoiser_bwd runs on (x_tB, tB, wsB) unchanged.  Bh utterances per problem; problem A reads cond [Bh, H, L]
 * and spk, problem B condB / spkB (NULL = the same as A's: the reference runs its train-mode linguistic encoder once per
 * model call, so the two phases see different conditioners unless the encoder is deterministic).  wsA: mg_denoiser_workspace_floats(dims, 2 * Bh, L, 0) floats (zero before its first use, like every
 * forward workspace); wsB: mg_denoiser_workspace_floats(dims, Bh, L, 1).  packed: with the backward packs or without.
 * MG_ERR_SHAPE when the single-launch kernel does not take the shape -- run the two forwards separately then. */
int mg_denoiser_fwd_pair(const mg_denoiser_dims *dims, const float *packed, const float *x_tA, const int64_t *tA,
                         const float *x_tB, const int64_t *tB, const float *cond, const float *spk, const float *condB,
                         const float *spkB, float *outA, float *outB, float *wsA, size_t wsA_floats, float *wsB,
                         size_t wsB_floats, int Bh, int L, void *stream);
/* Failure reporting of the single-launch kernels (mg_denoiser_fwd / _psample / _fwd_pair / _bwd).  Their workgroups
 * hand halo columns to each other; every wait is bounded, and when one gives up (another tenant holding the GPU's
 * workgroup slots for seconds) the launch drains instead of hanging, and
 *   - the workspace's sticky error word is set: that launch, and every later one on the same workspace, writes NaN
 *     instead of its output (forward: out / x_prev; backward: d x_t and the input-projection gradients);
 *   - a process-wide word in pinned host memory receives the code (forward: 1 + layer, backward: 0x100 + layer).
 * mg_persist_error(clear) returns that word without synchronising anything (0 = no failure so far; after a stream
 * synchronisation it is exact for all work before it) and resets it when `clear`.  A caller that sees it non-zero must
 * discard the workspaces in use (their sticky words stay set).  Test hooks, read per call: MG_PERSIST_SPIN_LIMIT (polls
 * before giving up), MG_PERSIST_FLAGS bit 1 (a tile withholds one hand-off). */
unsigned mg_persist_error(int clear);
/* Copies the single-launch forward's counter words {ticket, error, launches, workgroups done} of a (B, L, no-save)
 * workspace to host_out4 and synchronises the stream: error != 0 means a neighbour hand-off timed out (the launch
 * drained instead of hanging; its output is NaN). */
int mg_denoiser_persist_status(const mg_denoiser_dims *d, const float *workspace, int B, int L, unsigned *host_out4,
                               void *stream);

/* Backward of Denoiser.forward (what torch.autograd does for the reference).  `workspace` is the
 * forward's workspace of a save_for_backward call on the same inputs; `bwd_workspace` has
 * mg_denoiser_bwd_workspace_floats() floats.  g_out [B, M, L] is dL/d(out).
 * grads: pointer table in the order of mg_denoiser_pack's weight table; every non-NULL entry
 * receives dL/dW (overwritten, not accumulated).  Every per-layer entry (j=0/1 conv_layer weight/bias, j=2
 * diffusion_projection, j=3/4 conditioner_projection weight/bias, j=5/6 output_projection weight/bias, j=7
 * speaker_projection) must be the layer's slice of one contiguous [n_layers, ...] buffer: the layer loop only
 * runs the two data-gradient GEMMs per layer and keeps their results (0.5 GB of the workspace at B=8, L=1000);
 * all weight and bias gradients are then produced by a few launches grouped over the layer axis.
 * d_x_t [B,M,L], d_cond [B,H,L], d_spk [B,H] may be NULL when not needed. */
size_t mg_denoiser_bwd_workspace_floats(const mg_denoiser_dims *d, int B, int L);
int mg_denoiser_bwd(const mg_denoiser_dims *d, const float *packed, const float *g_out,
                    const float *x_t, const float *cond, const float *spk, float *workspace,
                    float *bwd_workspace, size_t bwd_workspace_floats, float *const *grads,
                    float *d_x_t, float *d_cond, float *d_spk, int B, int L, void *stream);
/* Same; conv3_grads_done (a hipEvent_t, or NULL) is recorded on `stream` right behind the launches that produce the
 * conv_layer weight and bias gradients of all layers (entries j=0/1: a third of the generator's gradient bytes), 0.6 ms
 * of GPU work before the last launch of the backward at B=8, L=1000 -- a data-parallel caller starts the all-reduce of
 * that slice behind the event while the remaining gradients are still being computed (train.py has no such exchange:
 * the reference is single-device). */
int mg_denoiser_bwd_staged(const mg_denoiser_dims *d, const float *packed, const float *g_out,
                           const float *x_t, const float *cond, const float *spk, float *workspace,
                           float *bwd_workspace, size_t bwd_workspace_floats, float *const *grads,
                           float *d_x_t, float *d_cond, float *d_spk, int B, int L, void *conv3_grads_done,
                           void *stream);

/* ------------------------------------------------------------------ pieces of autograd around the GEMMs
 * dpre = dy * act'(.) written through the saved OUTPUT y = act(pre) (ReLU / LeakyReLU(0.2) / tanh). */
int mg_act_bwd(const float *dy, const float *y, float *out, int act, size_t n, void *stream);
/* Zero insertion out[r, j] = in[r, j/stride] if stride | j else 0 (j < Lup): turns the data gradient
 * of a strided Conv1d (model/mixgantts.py:219-228, strides [1,2,2]) into a stride-1 convolution. */
int mg_upsample_zero(const float *in, float *out, int rows, int Lin, int stride, int Lup, void *stream);
/* Same with a leaky ReLU (slope) on the kept samples: lrelu + ConvTranspose1d of hifigan/models.py:150-151 as
 * zero insertion followed by a stride-1 convolution on the MG_PACK_DGRAD pack of the [Cin, Cout, K] weight. */
int mg_upsample_zero_act(const float *in, float *out, int rows, int Lin, int stride, int Lup, float slope,
                         void *stream);
/* ConvTranspose1d(Ci, Co, kernel 2u, stride u, padding u/2) of hifigan/models.py:121-127,151 without the zeros:
 * output phase r = n mod u is a 2-tap convolution of the input, so all u phases together are one 3-tap
 * implicit GEMM with Co*u rows (3/2 of the useful MACs instead of u times them with zero insertion) whose
 * epilogue interleaves the phases back into out [B, Co, u*Lin] with 16-byte stores.  u in {2,4,8}.
 * w is the PyTorch ConvTranspose1d weight [Ci, Co, 2u]; out = alpha * convT(lrelu(in, in_slope)) + bias. */
size_t mg_conv_transpose_packed_floats(int Ci, int Co, int u);
int mg_conv_transpose_pack(const float *w, float *packed, int Ci, int Co, int u, void *stream);
int mg_conv_transpose1d_fwd(const float *in, const float *packed, const float *bias, float *out, int B, int Ci,
                            int Lin, int Co, int u, float in_slope, float alpha, void *stream);

/* Step-embedding MLP: out = W2 mish(W0 [sin|cos](t * freq)).  Denoiser: model/modules.py:398-403,434;
 * JCUDiscriminator: model/mixgantts.py:203-208,265.  emb [B,D0], pre/h [B,D1] are saved for backward. */
int mg_step_mlp_fwd(const int64_t *t, const float *freq, const float *W0, const float *W2, float *emb,
                    float *pre, float *h, float *out, int B, int D0, int D1, int D2, void *stream);
/* g_out [B,D2] -> dW0 [D1,D0], dW2 [D2,D1]; scratch 2*B*D1 floats. */
int mg_step_mlp_bwd(const float *g_out, const float *emb, const float *pre, const float *h,
                    const float *W2, float *dW0, float *dW2, float *scratch, int B, int D0, int D1, int D2,
                    void *stream);
/* Bias-free per-sample Linear (LinearNorm on [B,K] vectors: speaker projections, spk_mlp). */
int mg_linear_small_fwd(const float *x, const float *W, float *out, int B, int N, int K, void *stream);
int mg_linear_small_bwd(const float *g, const float *x, const float *W, float *dx, float *dW, int B,
                        int N, int K, void *stream);

/* Stand-alone ResidualBlock.forward (model/blocks.py:1157-1176) on one layer's packs (mg_conv_pack: conditioner 1x1
 * MG_PACK_PLAIN, k=3 conv MG_PACK_GATE, output 1x1 MG_PACK_PLAIN), the same fused kernel mg_denoiser_fwd launches per
 * layer.  hvec [B,C] = Wd s (+ Wp spk) enters h, dvec [B,C] = Wd s the residual.  x_out [B,C,L] (must differ from x)
 * and skip [B,C,L] are written.  h/g/sig/tnh_save [B,C,L]: all four (training) or all NULL.  C = H = 256. */
int mg_resblock_fwd(const float *x, const float *cond, const float *wc_packed, const float *w3_packed,
                    const float *wo_packed, const float *bc, const float *b3, const float *bo, const float *hvec,
                    const float *dvec, float *x_out, float *skip, float *h_save, float *g_save, float *sig_save,
                    float *tnh_save, int B, int C, int H, int L, void *stream);
/* Derivative of the GLU gate g = sigmoid(z[:C]) * tanh(z[C:]) (model/blocks.py:1170-1171) through the saved
 * sigmoid / tanh values: dg, sig, tnh [B,C,L] -> dz [B,2C,L]. */
int mg_gate_bwd(const float *dg, const float *sig, const float *tnh, float *dz, int B, int C, int L, void *stream);
/* Mish.forward (model/blocks.py:894-896): y = x tanh(softplus(x)), and its derivative gx = gy * mish'(x). */
int mg_mish_fwd(const float *x, float *y, size_t n, void *stream);
int mg_mish_bwd(const float *gy, const float *x, float *gx, size_t n, void *stream);
/* DiffusionEmbedding.forward (model/blocks.py:906-913): emb [B,D] = [sin(t f) | cos(t f)], f [D/2] host table. */
int mg_step_embed(const int64_t *t, const float *freq, float *emb, int B, int D, void *stream);

/* Gradient of diffuse_trace (model/diffusion.py:167-175) w.r.t. x_start [B,L,M]: g is the stack [T+1,B,L,M] of the
 * gradients of the T+1 trace entries (entry 0 = clamped normalised x_start, entry t+1 = q_sample at step t); keep
 * uint8 [B,L] (0 = padded frame) or NULL; sqrt_alphas_cumprod [T]. */
int mg_diffuse_trace_bwd(const float *g, const float *x_start, const float *spec_min, const float *spec_max,
                         const uint8_t *keep, const float *sqrt_alphas_cumprod, float *d_x, int T, int B, int L, int M,
                         void *stream);

/* ------------------------------------------------------------------ aux pre-training
 * Strided batched fp32 GEMM on the MFMA: C[z](m,n) (+)= alpha * sum_k A[z](m,k) B[z](k,n), z = b*heads + h,
 * operand z at base + b*x_bs + h*x_hs; A(m,k) at m*a_ms + k*a_ks, B(k,n) at k*b_ks + n*b_ns (one stride of each
 * operand must be 1), C row-major with row stride c_ms.  The six contractions of train-mode attention and its
 * backward (transformer/Modules.py:16-23) are calls of this on the channel-major q/k/v and the kept [B*H,L,L]
 * probabilities. */
int mg_bgemm(const float *A, const float *B, float *C, int M, int N, int K, int batch, int heads, long a_ms, long a_ks,
             long a_bs, long a_hs, long b_ks, long b_ns, long b_bs, long b_hs, long c_ms, long c_bs, long c_hs,
             float alpha, int accumulate, void *stream);
/* In place: S [B*H, L, L] -> softmax over keys of scale*S with padded keys (key_pad [B, L] != 0) at -inf. */
int mg_softmax_rows_fwd(float *S, const uint8_t *key_pad, int B, int H, int L, float scale, void *stream);
/* In place on dP: dS = scale * P o (dP - rowsum(dP o P)). */
int mg_softmax_rows_bwd(const float *P, float *dP, int B, int H, int L, float scale, void *stream);
/* Train-mode post-LayerNorm (transformer/SubLayers.py:54-55,90-91): pre = a * keep * drop_scale + res (keep: uint8
 * [B,C,L] dropout keep-mask or NULL), out = pad ? 0 : LN_c(pre) * gamma + beta; `pre` is saved for the backward. */
int mg_layernorm_cm_train_fwd(const float *a, const uint8_t *keep, float drop_scale, const float *res,
                              const float *gamma, const float *beta, const uint8_t *pad, float *pre, float *out, int B,
                              int C, int L, float eps, void *stream);
/* d_pre (= d res), d_a = d_pre * keep * drop_scale (optional); dgamma / dbeta are [32][C] partial sums ACCUMULATED
 * by atomics (the caller zeroes them and sums the 32 rows). */
int mg_layernorm_cm_bwd(const float *pre, const float *dy, const float *gamma, const uint8_t *pad, const uint8_t *keep,
                        float drop_scale, float *d_pre, float *d_a, float *dgamma, float *dbeta, int B, int C, int L,
                        float eps, void *stream);
/* BatchNorm1d with batch statistics + tanh + dropout of PostNet (transformer/Layers.py:131-134) on [B,C,L]:
 * mg_bn_stats -> per-channel mean and biased variance over (B,L);  mg_bn_act_fwd: y = act((x-mean)*invstd*gamma+beta)
 * (act MG_ACT_NONE|MG_ACT_TANH; y saved when tanh), out = y * keep * drop_scale;  backward in two passes:
 * _reduce -> dbeta = sum dpre, dgamma = sum dpre*xhat (dpre = dout*keep*drop_scale*act'(y)), then _apply ->
 * dx = gamma*invstd*(dpre - dbeta*inv_count - xhat*dgamma*inv_count)  (all-reduce the two vectors between the
 * passes for cross-rank statistics). */
int mg_bn_stats(const float *x, float *mean, float *var, int B, int C, int L, void *stream);
int mg_bn_act_fwd(const float *x, const float *mean, const float *invstd, const float *gamma, const float *beta,
                  const uint8_t *keep, float drop_scale, int act, float *y, float *out, int B, int C, int L, void *stream);
int mg_bn_act_bwd_reduce(const float *dout, const uint8_t *keep, float drop_scale, const float *y, const float *x,
                         const float *mean, const float *invstd, int act, float *dgamma, float *dbeta, int B, int C, int L,
                         void *stream);
int mg_bn_act_bwd_apply(const float *dout, const uint8_t *keep, float drop_scale, const float *y, const float *x,
                        const float *mean, const float *invstd, const float *gamma, const float *dgamma,
                        const float *dbeta, float inv_count, int act, float *dx, int B, int C, int L, void *stream);

/* ------------------------------------------------------------------ optimizer step on flat buffers
 * The update of train.py:75-85 per optimizer -- nn.utils.clip_grad_norm_(params, clip) then Adam.step()
 * (utils/model.py:32-40 builds torch.optim.Adam(lr, betas)) -- with gradients, parameters and both moments each in
 * ONE flat fp32 buffer of n elements (16-byte aligned):
 * mg_grad_norm: out[0] = ||g||_2 (fixed summation order), out[1] = min(1, max_norm / (out[0] + 1e-6)), the factor
 *   clip_grad_norm_ multiplies into every gradient; scratch holds mg_grad_norm_scratch_floats() floats.
 * mg_adam_flat: one Adam step (torch.optim.Adam's rule: L2 weight decay, m lerp, bias corrections from `step` >= 1,
 *   denom = sqrt(v)/sqrt(1-beta2^step) + eps) on g * grad_scale[0] (device scalar, e.g. out + 1 above; NULL = 1).
 *   g is not modified. */
size_t mg_grad_norm_scratch_floats(void);
int mg_grad_norm(const float *g, size_t n, float max_norm, float *scratch, float *out, void *stream);
int mg_adam_flat(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2, float eps,
                 float weight_decay, long step, const float *grad_scale, void *stream);
/* Same; hyper (device, 2 floats, or NULL) = {lr / (1 - beta1^step), 1 / sqrt(1 - beta2^step)} is read by the kernel
 * instead of the values derived from the lr / step arguments -- a captured hipGraph of a training step replays this launch
 * with the numbers the host wrote there before the replay (HotPathTrainer.capture). */
int mg_adam_flat_dev(float *p, const float *g, float *m, float *v, size_t n, float lr, float beta1, float beta2, float eps,
                     float weight_decay, long step, const float *grad_scale, const float *hyper, void *stream);

/* ------------------------------------------------------------------ losses on the path (model/loss.py)
 * mg_loss_sum: out[0] = sum (a-c)^2 (mode 0: F.mse_loss against a constant label, loss.py:14-19)
 *              or sum |a-b| (mode 1: F.l1_loss numerator, loss.py:221-227); the caller divides by n.
 * mg_loss_grad: da = g[0] * coef * {2(a-c) | sign(a-b)}; g is a device scalar (no host sync). */
int mg_loss_sum(const float *a, const float *b, float c, int mode, size_t n, float *out, void *stream);
int mg_loss_grad(const float *a, const float *b, float c, int mode, const float *g, float coef, size_t n,
                 float *da, void *stream);
/* A weighted sum of mean-reduced terms in one launch: total = sum_k weight_k * mean_k, mean_k = mean (a-c)^2
 * (mode 0) or mean |a-b| (mode 1) over n elements -- the LSGAN pair of model/loss.py:12-30 and the feature-matching
 * sum of model/loss.py:221-227 are such sums.  out[0] = total, out[1+q] = the weighted subtotal of the terms with
 * group == q (q < MG_LOSS_GROUPS; e.g. adversarial and feature-matching parts for logging),
 * out[1+MG_LOSS_GROUPS+k] = mean_k; fixed summation order.  scratch:
 * mg_multi_loss_scratch_floats() floats, ZERO before the first call (the kernel keeps a ticket in it and re-arms it),
 * private to one stream.  mg_multi_loss_bwd: da_k = g[0] * weight_k / n_k * {2(a-c) | sign(a-b)} for every term
 * whose da is not NULL. */
#define MG_LOSS_MAX_TERMS 16
#define MG_LOSS_GROUPS 4
typedef struct MgLossTerm {
    const float *a;
    const float *b;   /* mode 1 only */
    float *da;        /* backward only; NULL = no gradient wanted */
    size_t n;
    float c;          /* mode 0 only */
    float weight;
    int32_t mode;
    int32_t group;
} MgLossTerm;
size_t mg_multi_loss_scratch_floats(void);
int mg_multi_loss_fwd(const MgLossTerm *terms, int nterms, float *scratch, float *out, void *stream);
int mg_multi_loss_bwd(const MgLossTerm *terms, int nterms, const float *g, void *stream);
/* Masked mel L1 (loss.py:229-242,255-259) over rows = B*L frames of M bins, pad uint8 [rows] (1 = pad):
 * out2 = {sum |p-t| over counted rows, M * #counted rows}; the loss is out2[0]/out2[1]. */
int mg_mel_l1_fwd(const float *pred, const float *targ, const uint8_t *pad, int rows, int M, float *out2,
                  void *stream);
int mg_mel_l1_bwd(const float *pred, const float *targ, const uint8_t *pad, int rows, int M,
                  const float *g, const float *den, float *dpred, void *stream);

/* ------------------------------------------------------------------ FFT blocks (shallow / aux coarse mel)
 * Multi-head self-attention of transformer/SubLayers.py:29-57 + Modules.py:16-23 without the
 * [n_head*B, L, L] score tensor (streaming softmax, fp32 MFMA).  qkv [B, 3*n_head*d_head, L]
 * channel-major, rows = [Q heads | K heads | V heads], head-major like the reference's .view();
 * key_pad uint8 [B, L] (1 = padded key, masked with -inf for every query) or NULL;
 * out [B, n_head*d_head, L]; scale = 1/temperature = 1/sqrt(d_k).  d_head must be 128. */
int mg_attention_fwd(const float *qkv, const uint8_t *key_pad, float *out, int B, int L, int n_head,
                     int d_head, float scale, void *stream);
/* The same function with fp16 MFMA operands (q*scale, k, v and the probabilities rounded to fp16; fp32
 * accumulation, softmax statistics and I/O): BASELINE configs[4]'s "fp16 MFMA attention path" for L = 4000. */
int mg_attention_fwd_f16(const float *qkv, const uint8_t *key_pad, float *out, int B, int L, int n_head,
                     int d_head, float scale, void *stream);
/* Post-LayerNorm on the channel-major layout (SubLayers.py:55,91 + Layers.py:25,28):
 * out[b,c,l] = pad[b,l] ? 0 : LN_c(a[b,:,l] + res[b,:,l]) * gamma[c] + beta[c];  C must be 256. */
int mg_layernorm_cm_fwd(const float *a, const float *res, const float *gamma, const float *beta,
                        const uint8_t *pad, float *out, int B, int C, int L, float eps, void *stream);

/* ------------------------------------------------------------------ linguistic-encoder index ops
 * Device versions of the four functions of the LinguisticEncoder that loop over the batch in
 * Python with one .item() per phoneme.  Durations / counts are int64 like the reference's LongTensors.
 * LengthRegulator (model/linguistic_encoder.py:383-416): x [B,Tw,H], dur [B,Tw] (negative = 0) ->
 * out [B,Lmax,H] (rows past the expanded length are zero, longer expansions are cropped), mel_len [B]. */
int mg_length_regulate_fwd(const float *x, const int64_t *dur, float *out, int64_t *mel_len, int B,
                           int Tw, int H, int Lmax, void *stream);
int mg_length_regulate_bwd(const float *dout, const int64_t *dur, float *dx, int B, int Tw, int H,
                           int Lmax, void *stream);
/* word_level_pooling (utils/tools.py:394-413): src [B,Tp,H], wb [B,Tw] phones per word,
 * src_w_len [B] -> out [B,Wout,H] (sum, or mean when mean != 0). */
int mg_word_pool_fwd(const float *src, const int64_t *wb, const int64_t *src_w_len, float *out, int B,
                     int Tp, int Tw, int H, int Wout, int mean, void *stream);
int mg_word_pool_bwd(const float *dout, const int64_t *wb, const int64_t *src_w_len, float *dsrc, int B,
                     int Tp, int Tw, int H, int Wout, int mean, void *stream);
/* get_mapping_mask (model/linguistic_encoder.py:185-199): out uint8 [B,Lq,Lkv], 1 inside the
 * (frames of word i) x (phonemes of word i) blocks. */
int mg_mapping_mask(const int64_t *dur_w, const int64_t *wb, const int64_t *src_w_len, uint8_t *out,
                    int B, int Tw, int Lq, int Lkv, void *stream);
/* get_rel_coef (model/linguistic_encoder.py:222-236): mask uint8 [B,Lout] (1 = valid) -> out [B,Lout]. */
int mg_rel_coef(const int64_t *dur, const int64_t *dur_len, const uint8_t *mask, float *out, int B, int T,
                int Lout, void *stream);

/* ------------------------------------------------------------------ measurement hooks (bench.py)
 * While a session is open, mg_denoiser_fwd brackets each launch of its dominant kernel (the k=3
 * gated convolution of a residual layer) with HIP events recorded on the launch stream.
 * mg_profile_end waits for them and returns the number of brackets written to ms_out. */
int mg_profile_begin(int max_brackets);
/* As above, bracketing only every `every`-th launch (keeps the events' own cost out of the timing). */
int mg_profile_begin_sampled(int max_brackets, int every);
int mg_profile_end(float *ms_out, int max_out);

#ifdef __cplusplus
}
#endif
#endif /* MIXGAN_HIP_H */
