// Shared between denoiser.hip (forward, packing) and denoiser_bwd.hip (backward): the packed
// weight blob layout, the forward/backward workspace carve-ups and the tiny per-sample kernels
// of the step-embedding MLP.
#pragma once
#include "conv_mfma.h"

// ------------------------------------------------------------------------------------------ single-launch kernels: host side
// (defined in denoiser.hip; used by the forward and backward launchers)
// Compute units of the current device (the forward-progress argument of the persistent kernels counts slots).
int mg_device_cus();
// Device-visible address of the library's host-pinned error word, or NULL when it cannot be allocated (then only the
// workspace's own sticky word and the NaN poison report a failure).  64 bytes of pinned HOST memory, allocated on first
// use -- the only allocation the library makes.
unsigned *mg_host_err_device_ptr();
// polls a hand-off wait makes before it gives up: DP_SPIN_LIMIT, or MG_PERSIST_SPIN_LIMIT (tests)
unsigned mg_persist_spin_limit();

// ------------------------------------------------------------------------------------------ packed blob
struct DenLayout {
    // offsets in floats into the packed blob
    size_t freq, in_w, in_b, mlp0, mlp2, skip_w, skip_b, out_w, out_b, layers, layer_stride;
    size_t l_wc, l_w3, l_wo, l_bc, l_b3, l_bo, l_wd, l_wp;  // offsets inside a layer record
    // backward section (data-gradient packs), present when with_backward
    size_t bw, in_wT, skip_wT, out_wT, wc_allT, blayers, blayer_stride;
    size_t bl_w3T, bl_woT;
    // split-bf16 section (resblock_split.h), present when flags & MG_DEN_SPLIT
    size_t slayers, slayer_stride, sl_wc, sl_w3, sl_wo;
    // 16x16x4-MFMA packs (denoiser_persist16.h), present when flags & MG_DEN_P16
    size_t in_w16, skip_w16, out_w16, p16layers, p16layer_stride, p_wc, p_w3, p_wo;
    // ... and with them (inference packs): every layer's conditioner projection as ONE [NL * C, H] forward GEMM with its
    // biases (mg_denoiser_cond_project: the x_t-independent part of all GEMM 1s of a sampling loop)
    size_t wc_all, bc_all;
    size_t jobs;   // scratch slice for the pack job table (mg_denoiser_pack)
    size_t total;
};

static inline DenLayout den_layout(const mg_denoiser_dims *d, int flags)
{
    const int with_backward = flags & MG_DEN_BACKWARD;
    const int C = d->channels, H = d->cond_channels, M = d->mel_bins, NL = d->n_layers;
    DenLayout o;
    size_t p = 0;
    auto take = [&](size_t n) {
        size_t at = p;
        p += mg_align_up(n, 64);
        return at;
    };
    o.freq = take(C / 2);
    o.in_w = take(mg_conv_packed_floats(C, M, 1, MG_PACK_PLAIN));
    o.in_b = take(C);
    o.mlp0 = take((size_t)4 * C * C);
    o.mlp2 = take((size_t)4 * C * C);
    o.skip_w = take(mg_conv_packed_floats(C, C, 1, MG_PACK_PLAIN));
    o.skip_b = take(C);
    o.out_w = take(mg_conv_packed_floats(M, C, 1, MG_PACK_PLAIN));
    o.out_b = take(M);
    o.layers = p;
    size_t q = 0;
    auto ltake = [&](size_t n) {
        size_t at = q;
        q += mg_align_up(n, 64);
        return at;
    };
    // The three MFMA weight streams of a layer come first, contiguous and in the order the fused
    // kernel walks them: measured 4.17 -> 3.57 ms/step against the "w3, b3, wd, wc, ..." order
    // (identical ISA; the difference is only where the streams sit in the address space).
    o.l_wc = ltake(mg_conv_packed_floats(C, H, 1, MG_PACK_PLAIN));
    o.l_w3 = ltake(mg_conv_packed_floats(2 * C, C, 3, MG_PACK_GATE));
    o.l_wo = ltake(mg_conv_packed_floats(2 * C, C, 1, MG_PACK_PLAIN));
    o.l_bc = ltake(C);
    o.l_b3 = ltake(2 * C);
    o.l_bo = ltake(2 * C);
    o.l_wd = ltake((size_t)C * C);
    o.l_wp = ltake(d->multi_speaker ? (size_t)C * H : 0);
    o.layer_stride = q;
    p += q * NL;
    o.bw = p;
    o.in_wT = o.skip_wT = o.out_wT = o.wc_allT = o.blayers = o.blayer_stride = o.bl_w3T = o.bl_woT = 0;
    if (with_backward) {
        o.in_wT = take(mg_conv_packed_floats(C, M, 1, MG_PACK_DGRAD));
        o.skip_wT = take(mg_conv_packed_floats(C, C, 1, MG_PACK_DGRAD));
        o.out_wT = take(mg_conv_packed_floats(M, C, 1, MG_PACK_DGRAD));
        // all conditioner projections as one data-gradient GEMM: rows = H, reduction = NL*C
        o.wc_allT = take((size_t)mg_conv_mblocks(H) * ((size_t)NL * C / 8) * 256);
        o.blayers = p;
        size_t r = 0;
        auto btake = [&](size_t n) {
            size_t at = r;
            r += mg_align_up(n, 64);
            return at;
        };
        o.bl_w3T = btake(mg_conv_packed_floats(2 * C, C, 3, MG_PACK_DGRAD));
        o.bl_woT = btake(mg_conv_packed_floats(2 * C, C, 1, MG_PACK_DGRAD));
        o.blayer_stride = r;
        p += r * NL;
    }
    o.slayers = o.slayer_stride = o.sl_wc = o.sl_w3 = o.sl_wo = 0;
    if (flags & MG_DEN_SPLIT) {
        // hi/lo bf16 pairs: 32 B per lane per 16-deep k-group = MB * KG * 512 floats
        o.slayers = p;
        size_t r = 0;
        auto stake = [&](size_t n) {
            size_t at = r;
            r += mg_align_up(n, 64);
            return at;
        };
        o.sl_wc = stake((size_t)(C / 32) * (H / 16) * 512);
        o.sl_w3 = stake((size_t)(2 * C / 32) * (3 * C / 16) * 512);
        o.sl_wo = stake((size_t)(2 * C / 32) * (C / 16) * 512);
        o.slayer_stride = r;
        p += r * NL;
    }
    o.in_w16 = o.skip_w16 = o.out_w16 = o.p16layers = o.p16layer_stride = o.p_wc = o.p_w3 = o.p_wo = 0;
    o.wc_all = o.bc_all = 0;
    if (flags & MG_DEN_P16) {
        o.in_w16 = take(mg_conv_packed_floats(C, M, 1, MG_PACK_PLAIN16));
        o.skip_w16 = take(mg_conv_packed_floats(C, C, 1, MG_PACK_PLAIN16));
        o.out_w16 = take(mg_conv_packed_floats(M, C, 1, MG_PACK_PLAIN16));
        o.p16layers = p;
        size_t r = 0;
        auto ptake = [&](size_t n) {
            size_t at = r;
            r += mg_align_up(n, 64);
            return at;
        };
        o.p_wc = ptake(mg_conv_packed_floats(C, H, 1, MG_PACK_PLAIN16));
        o.p_w3 = ptake(mg_conv_packed_floats(2 * C, C, 3, MG_PACK_GATE16));
        o.p_wo = ptake(mg_conv_packed_floats(2 * C, C, 1, MG_PACK_PLAIN16));
        o.p16layer_stride = r;
        p += r * NL;
        o.wc_all = take((size_t)NL * mg_conv_packed_floats(C, H, 1, MG_PACK_PLAIN));   // C % 128 == 0: row blocks concatenate
        o.bc_all = take((size_t)NL * C);
    }
    // reserved for the job table of mg_denoiser_pack (768 entries x 96 B, 16-byte aligned)
    p = mg_align_up(p, 64);
    o.jobs = p;
    p += 768 * 96 / 4;
    o.total = p;
    return o;
}

static inline int den_check(const mg_denoiser_dims *d)
{
    if (!d) return MG_ERR_ARG;
    if (d->n_layers <= 0 || d->n_layers > 256 || d->channels <= 0 || d->channels % 64 || d->cond_channels <= 0 ||
        d->cond_channels % 32 || d->mel_bins <= 0)
        return MG_ERR_SHAPE;
    return MG_OK;
}

// ------------------------------------------------------------------------------------------ forward workspace
struct DenWs {
    size_t emb, h1pre, h1, s, dvec, hvec, x, skip, y, x0, h, g, sig, tnh, conds, total;
    size_t act_stride;  // per-layer stride of h/g/sig/tnh (0 when not saving)
    // single-launch forward (denoiser_persist.h): 64 floats of counters (must be ZERO when the workspace is first used;
    // the kernel re-arms them itself) and the halo hand-off granules [2][tiles][2][C] x 8 bytes
    size_t sync, gran;
    size_t team;   // denoiser_team16.h: gather buffers of the workgroup teams (small launches only; zero at first use)
};

// 32-frame tiles of the single-launch forward
static inline size_t den_persist_tiles(int B, int L) { return (size_t)B * ((L + 15) / 16); }

static inline DenWs den_ws(const mg_denoiser_dims *d, int B, int L, int save)
{
    const size_t C = d->channels, NL = d->n_layers;
    const size_t act = mg_align_up((size_t)B * C * L, 64);
    DenWs w;
    size_t p = 0;
    auto take = [&](size_t n) {
        size_t at = p;
        p += mg_align_up(n, 64);
        return at;
    };
    w.emb = take(B * C);
    w.h1pre = take(B * 4 * C);
    w.h1 = take(B * 4 * C);
    w.s = take(B * C);
    w.dvec = take(NL * B * C);
    w.hvec = d->multi_speaker ? take(NL * B * C) : w.dvec;
    w.x = take(act);
    w.skip = take(act);
    w.y = take(act);
    w.x0 = save ? take(act) : w.x;  // input-projection output, kept for its ReLU mask
    w.act_stride = save ? act : 0;
    const size_t nact = save ? NL : 1;
    w.h = take(act * nact);
    w.g = take(act * nact);
    w.sig = save ? take(act * nact) : 0;
    w.tnh = save ? take(act * nact) : 0;
    w.conds = take(act);  // frame-major bf16 hi/lo planes of the conditioner (split-precision path)
    w.sync = take(64);
    w.gran = take(2 * den_persist_tiles(B, L) * 2 * C * 2);
    // [2 parities][tiles][256][18 + 16] granules when the launch is small enough for the team kernel to be considered
    w.team = take(den_persist_tiles(B, L) <= 128 && C == 256 ? 2 * den_persist_tiles(B, L) * 256 * (18 + 16) * 2 : 0);
    w.total = p;
    return w;
}

// ------------------------------------------------------------------------------------------ backward workspace
struct DenBws {
    // dz_all [B][NL*2C][L] and dx_all [B][(NL+1)*C][L] keep every layer's gate-pre-activation gradient and residual
    // gradient so that the weight / bias gradients of all layers are computed in a handful of grouped launches at
    // the end (0.5 GB at B=8, L=1000 -- 0.2 % of HBM) instead of 5 small launches per layer.
    size_t dout, dz_all, dx_all, dh_all, dy, dx0, scratch, dd_all, dhv_all, ds, dm, da, btop, bbot, total;
    // single-launch data gradients (denoiser_bwd_persist.h): counters (ZERO at first use) + dz hand-off granules
    size_t sync, gran;
};

static inline DenBws den_bws(const mg_denoiser_dims *d, int B, int L)
{
    const size_t C = d->channels, NL = d->n_layers, H = d->cond_channels, M = d->mel_bins;
    const size_t act = mg_align_up((size_t)B * C * L, 64);
    DenBws w;
    size_t p = 0;
    auto take = [&](size_t n) {
        size_t at = p;
        p += mg_align_up(n, 64);
        return at;
    };
    w.dout = take(2 * act);
    w.dz_all = take(2 * NL * act);
    w.dx_all = take((NL + 1) * act);
    w.dh_all = take(NL * act);
    w.dy = take(act);
    w.dx0 = take(act);
    // partial-tile scratch of the largest weight gradient (mg_conv1d_wgrad_scratch_floats of each shape run by
    // mg_denoiser_bwd: out 1x1, k=3, cond 1x1 of all layers at once, input / skip / output projections)
    size_t sc = 0;
    const int c = (int)C, h = (int)H, m = (int)M, nl = (int)NL;
    const int shapes[6][3] = {{2 * c, c, 3}, {2 * c, c, 1}, {nl * c, h, 1}, {c, m, 1}, {c, c, 1}, {m, c, 1}};
    for (const auto &sh : shapes) {
        const size_t need = mg_conv1d_wgrad_scratch_floats(sh[0], sh[1], sh[2]);
        if (need > sc) sc = need;
    }
    {   // the grouped gradients of the residual layers: k=3 conv [2C, C, 3] and the two row halves of the output conv
        const size_t g3 = mg_conv1d_wgrad_grouped_scratch_floats(2 * c, c, 3, nl);
        const size_t go = mg_conv1d_wgrad_grouped_scratch_floats(c, c, 1, nl);
        if (g3 > sc) sc = g3;
        if (go > sc) sc = go;
    }
    w.scratch = take(sc);
    w.btop = take((size_t)NL * C);
    w.bbot = take(C);
    w.dd_all = take(NL * B * C);
    w.dhv_all = take(NL * B * C);
    w.ds = take(B * C);
    w.dm = take(B * 4 * C);
    w.da = take(B * 4 * C);
    w.sync = take(64);
    w.gran = take(2 * den_persist_tiles(B, L) * 2 * (2 * C) * 2);
    w.total = p;
    return w;
}

// ------------------------------------------------------------------------------------------ tiny kernels
// step embedding (model/blocks.py:906-913): emb[b] = [sin(t f_i) | cos(t f_i)], f from the host table
static __global__ void step_embed_kernel(const int64_t *__restrict__ t, const float *__restrict__ freq,
                                  float *__restrict__ emb, int B, int C)
{
    const int half = C / 2;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * half) return;
    const int b = idx / half, i = idx - b * half;
    // the reference rounds the product to fp32 and takes sin / cos of THAT number; at t ~ 1000 rad the fp32 device
    // sinf is off by about one ulp of the ANGLE (6e-5), so evaluate in double (B * C/2 values per call: free)
    const double ang = (double)((float)t[b] * freq[i]);
    emb[(size_t)b * C + i] = (float)sin(ang);
    emb[(size_t)b * C + half + i] = (float)cos(ang);
}

__device__ __forceinline__ float mg_softplus(float v) { return v > 20.f ? v : log1pf(expf(v)); }  // F.softplus defaults

// out[z][b][j] = act(sum_i W[z][j][i] * in[b % in_rows][i]) (+ add[z][b][j]);  one wave per output row j.
// mish: out = x tanh(softplus(x)) (model/blocks.py:894-896); pre (optional) receives x itself.
// in_rows < B: the input rows repeat (the speaker embeddings of a batch under the steps of a sampling loop).
template <int BC>
__global__ __launch_bounds__(256) void small_linear_kernel(const float *__restrict__ W, long w_zs,
                                                           const float *__restrict__ in, float *__restrict__ out,
                                                           long out_zs, const float *__restrict__ add, long add_zs,
                                                           float *__restrict__ pre, int B, int N, int K, int mish,
                                                           int in_rows)
{
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int b0 = blockIdx.y * BC;
    const int z = blockIdx.z;
    if (j >= N) return;
    const float *w = W + (size_t)z * w_zs + (size_t)j * K;
    float acc[BC];
#pragma unroll
    for (int b = 0; b < BC; ++b) acc[b] = 0.f;
    if ((K & 255) == 0) {
        // 16-byte loads, all of an iteration's loads independent (the scalar loop below is a chain of
        // dependent L2 round trips: 17-35 us per launch for a few hundred kFLOP)
        for (int i = lane * 4; i < K; i += 256) {
            const f32x4 wv = *reinterpret_cast<const f32x4 *>(w + i);
            f32x4 xv[BC];
#pragma unroll
            for (int b = 0; b < BC; ++b)
                xv[b] = *reinterpret_cast<const f32x4 *>(in + (size_t)(min(b0 + b, B - 1) % in_rows) * K + i);
            // explicit fma chain: the same rounding for every batch slot (a free-form expression lets the
            // SLP vectoriser pick packed mul+add for some slots and fma for others, which breaks bitwise
            // batch independence)
#pragma unroll
            for (int b = 0; b < BC; ++b) {
                float t = acc[b];
                t = fmaf(wv[0], xv[b][0], t);
                t = fmaf(wv[1], xv[b][1], t);
                t = fmaf(wv[2], xv[b][2], t);
                t = fmaf(wv[3], xv[b][3], t);
                acc[b] = t;
            }
        }
    } else {
        for (int i = lane; i < K; i += 64) {
            const float wv = w[i];
#pragma unroll
            for (int b = 0; b < BC; ++b)
                if (b0 + b < B) acc[b] = fmaf(wv, in[(size_t)((b0 + b) % in_rows) * K + i], acc[b]);
        }
    }
#pragma unroll
    for (int b = 0; b < BC; ++b) {
        float v = acc[b];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0 && b0 + b < B) {
            if (pre) pre[(size_t)(b0 + b) * N + j] = v;
            if (mish) v = v * tanhf(mg_softplus(v));
            const size_t o = (size_t)z * out_zs + (size_t)(b0 + b) * N + j;
            if (add) v += add[(size_t)z * add_zs + (size_t)(b0 + b) * N + j];
            out[o] = v;
        }
    }
}

static inline int small_linear(const float *W, long w_zs, const float *in, float *out, long out_zs, const float *add,
                               long add_zs, float *pre, int B, int N, int K, int Z, int mish, hipStream_t st,
                               int in_rows = 0)
{
    constexpr int BC = 8;
    dim3 grid(mg_cdiv(N, 4), mg_cdiv(B, BC), Z);
    hipLaunchKernelGGL(small_linear_kernel<BC>, grid, dim3(256), 0, st, W, w_zs, in, out, out_zs, add, add_zs, pre, B, N,
                       K, mish, in_rows > 0 ? in_rows : B);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ------------------------------------------------------------------------------------------ tiny backward kernels
// out[z][i][j] = sum_b a[z*a_zs + b*a_bs + i] * c[b*K + j]          (outer products over the batch)
static __global__ void small_outer_kernel(const float *__restrict__ a, long a_zs, long a_bs, const float *__restrict__ c,
                                   float *__restrict__ out, int Z, int B, int N, int K)
{
    const size_t n = (size_t)Z * N * K;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int j = (int)(idx % K);
        const int i = (int)((idx / K) % N);
        const int z = (int)(idx / ((size_t)K * N));
        float s = 0.f;
        for (int b = 0; b < B; ++b) s = fmaf(a[(size_t)z * a_zs + (size_t)b * a_bs + i], c[(size_t)b * K + j], s);
        out[idx] = s;
    }
}

// out[b][j] (+)= sum_z sum_i W[z*w_zs + i*K + j] * a[z*a_zs + b*a_bs + i]   (transposed linears, summed over layers)
// grid (K/64, B, Z): lanes run along j (coalesced rows of W), the 4 waves split i, one fp32 atomic per
// (z, b, j) when Z > 1 (out must be zeroed by the caller then).
static __global__ __launch_bounds__(256) void small_linear_t_kernel(const float *__restrict__ W, long w_zs,
                                                                    const float *__restrict__ a, long a_zs, long a_bs,
                                                                    float *__restrict__ out, int Z, int B, int N, int K)
{
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = blockIdx.x * 64 + lane;
    const int b = blockIdx.y, z = blockIdx.z;
    const float *w = W + (size_t)z * w_zs + min(j, K - 1);
    const float *av = a + (size_t)z * a_zs + (size_t)b * a_bs;
    const int per = (N + 3) / 4;
    const int i0 = wave * per, i1 = min(N, i0 + per);
    float s = 0.f;
#pragma unroll 8
    for (int i = i0; i < i1; ++i) s = fmaf(w[(size_t)i * K], av[i], s);
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && j < K) {
        const float t = red[0][lane] + red[1][lane] + red[2][lane] + red[3][lane];
        if (Z > 1) atomicAdd(out + (size_t)b * K + j, t);
        else out[(size_t)b * K + j] = t;
    }
}

// da = dm * mish'(x),  mish'(x) = tanh(sp) + x (1 - tanh(sp)^2) sigmoid(x),  sp = softplus(x)
static __global__ void mish_bwd_kernel(const float *__restrict__ dm, const float *__restrict__ x, float *__restrict__ da, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    const float ts = tanhf(mg_softplus(v));
    const float sg = 1.f / (1.f + expf(-v));
    da[i] = dm[i] * (ts + v * (1.f - ts * ts) * sg);
}

// out[b,c,l] = alpha * in[b*in_bs + c*L + l] * (mask[b,c,l] > 0)
static __global__ void scale_mask_kernel(const float *__restrict__ in, long in_bs, const float *__restrict__ mask,
                                  float *__restrict__ out, float alpha, int CL, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t b = i / CL;
        const size_t r = i - b * CL;
        out[i] = mask[i] > 0.f ? alpha * in[b * in_bs + r] : 0.f;
    }
}

