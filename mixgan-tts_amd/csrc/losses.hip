// Adversarial / feature-matching / mel losses on the path (model/loss.py:12-30, 221-242, 255-259):
// sum reductions with one fp32 atomic per workgroup, and their gradients.  The upstream gradient
// g is read from device memory (a 0-dim tensor) so no host synchronisation is needed.
#include "common.h"

__device__ __forceinline__ float block_sum(float v, float *red)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// mode 0: sum (a - c)^2      mode 1: sum |a - b|
__global__ __launch_bounds__(256) void loss_sum_kernel(const float *__restrict__ a, const float *__restrict__ b, float c,
                                                       int mode, size_t n, float *__restrict__ out)
{
    __shared__ float red[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = mode == 0 ? a[i] - c : a[i] - b[i];
        s += mode == 0 ? d * d : fabsf(d);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(out, s);
}

// da[i] = g * coef * (mode 0: 2 (a - c);  mode 1: sign(a - b))
__global__ __launch_bounds__(256) void loss_grad_kernel(const float *__restrict__ a, const float *__restrict__ b, float c,
                                                        int mode, const float *__restrict__ g, float coef, size_t n,
                                                        float *__restrict__ da)
{
    const float k = g[0] * coef;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float d = mode == 0 ? a[i] - c : a[i] - b[i];
        da[i] = mode == 0 ? 2.f * k * d : (d > 0.f ? k : (d < 0.f ? -k : 0.f));
    }
}

extern "C" int mg_loss_sum(const float *a, const float *b, float c, int mode, size_t n, float *out, void *stream)
{
    if (!a || !out || (mode == 1 && !b)) return MG_ERR_ARG;
    if (mode < 0 || mode > 1) return MG_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out, 0, sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    if (n == 0) return MG_OK;
    const int blocks = (int)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    hipLaunchKernelGGL(loss_sum_kernel, dim3(blocks), dim3(256), 0, st, a, b, c, mode, n, out);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_loss_grad(const float *a, const float *b, float c, int mode, const float *g, float coef, size_t n,
                            float *da, void *stream)
{
    if (!a || !g || !da || (mode == 1 && !b)) return MG_ERR_ARG;
    if (mode < 0 || mode > 1) return MG_ERR_ARG;
    if (n == 0) return MG_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(loss_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, b, c, mode, g, coef, n, da);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// mel L1 of model/loss.py:229-242,255-259: rows (frames) whose pad flag is set are zeroed in both
// tensors; a row counts iff its (zero-filled) target has a non-zero entry.  One wave per row.
// out[0] += sum |p - t| * w,  out[1] += w * M;   if dpred: dpred = g / out_den * sign(p - t) * w
__global__ __launch_bounds__(256) void mel_l1_kernel(const float *__restrict__ pred, const float *__restrict__ targ,
                                                     const uint8_t *__restrict__ pad, int rows, int M,
                                                     float *__restrict__ out, const float *__restrict__ g,
                                                     const float *__restrict__ den, float *__restrict__ dpred)
{
    // one wave per row, grid-stride over rows; forward sums are combined per workgroup so that only
    // gridDim.x atomics land on the two result words (same-address atomics serialise)
    __shared__ float red[2][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc_l1 = 0.f, acc_n = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < rows; row += gridDim.x * 4) {
        const bool padded = pad && pad[row];
        const float *p = pred + (size_t)row * M, *t = targ + (size_t)row * M;
        float tabs = 0.f, l1 = 0.f;
        for (int m = lane; m < M; m += 64) {
            const float tv = padded ? 0.f : t[m], pv = padded ? 0.f : p[m];
            tabs += fabsf(tv);
            l1 += fabsf(pv - tv);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            tabs += __shfl_xor(tabs, o, 64);
            l1 += __shfl_xor(l1, o, 64);
        }
        const float w = tabs != 0.f ? 1.f : 0.f;
        if (!dpred) {
            acc_l1 += w * l1;
            acc_n += w * (float)M;
        } else {
            const float k = (padded || w == 0.f) ? 0.f : g[0] / den[0];
            for (int m = lane; m < M; m += 64) {
                const float d = p[m] - t[m];
                dpred[(size_t)row * M + m] = d > 0.f ? k : (d < 0.f ? -k : 0.f);
            }
        }
    }
    if (!dpred) {
        if (lane == 0) {
            red[0][wave] = acc_l1;
            red[1][wave] = acc_n;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(out, red[0][0] + red[0][1] + red[0][2] + red[0][3]);
            atomicAdd(out + 1, red[1][0] + red[1][1] + red[1][2] + red[1][3]);
        }
    }
}

extern "C" int mg_mel_l1_fwd(const float *pred, const float *targ, const uint8_t *pad, int rows, int M, float *out2,
                             void *stream)
{
    if (!pred || !targ || !out2) return MG_ERR_ARG;
    if (rows <= 0 || M <= 0) return MG_ERR_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out2, 0, 2 * sizeof(float), st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(mel_l1_kernel, dim3(min(mg_cdiv(rows, 4), 512)), dim3(256), 0, st, pred, targ, pad, rows, M, out2,
                       nullptr, nullptr, nullptr);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_mel_l1_bwd(const float *pred, const float *targ, const uint8_t *pad, int rows, int M, const float *g,
                             const float *den, float *dpred, void *stream)
{
    if (!pred || !targ || !g || !den || !dpred) return MG_ERR_ARG;
    if (rows <= 0 || M <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(mel_l1_kernel, dim3(min(mg_cdiv(rows, 4), 2048)), dim3(256), 0, (hipStream_t)stream, pred, targ, pad,
                       rows, M, nullptr, g, den, dpred);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// Weighted sum of up to MG_LOSS_MAX_TERMS mean-reduced terms in ONE launch (and ONE for its gradient).
// The adversarial + feature-matching part of the generator loss is ten such terms (model/loss.py:12-30,221-227);
// one launch pair per term plus the scalar arithmetic between them was ~110 launches of a few microseconds each
// per training step.  Per-block partial sums land in scratch; the last block to finish adds them in index order
// (fixed summation order), writes the total and the per-term means, and re-arms the ticket.
// ---------------------------------------------------------------------------------------------
struct MultiLossArgs {
    MgLossTerm t[MG_LOSS_MAX_TERMS];
    int first_block[MG_LOSS_MAX_TERMS + 1];
    int nterms;
    float *partial;      // [total blocks]
    unsigned *ticket;    // zero before the first launch; re-armed by the kernel
    float *out;          // [1 + MG_LOSS_GROUPS + nterms]: total, group subtotals, then each term's unweighted mean
    const float *g;      // backward: upstream gradient (device scalar)
};

__device__ __forceinline__ int ml_term_of(const MultiLossArgs &a, int block)
{
    int k = 0;
    while (k + 1 < a.nterms && block >= a.first_block[k + 1]) ++k;
    return k;
}

__global__ __launch_bounds__(256) void multi_loss_fwd_kernel(MultiLossArgs a)
{
    __shared__ float red[4];
    __shared__ unsigned last;
    const int k = ml_term_of(a, blockIdx.x);
    const MgLossTerm t = a.t[k];
    const int nb = a.first_block[k + 1] - a.first_block[k], lb = blockIdx.x - a.first_block[k];
    float s = 0.f;
    for (size_t i = (size_t)lb * 256 + threadIdx.x; i < t.n; i += (size_t)nb * 256) {
        const float d = t.mode == 0 ? t.a[i] - t.c : t.a[i] - t.b[i];
        s += t.mode == 0 ? d * d : fabsf(d);
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) {
        __hip_atomic_store(a.partial + blockIdx.x, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        last = atomicAdd(a.ticket, 1u) == gridDim.x - 1 ? 1u : 0u;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    if (threadIdx.x < 64) {
        // wave 0: lane j < nterms adds its term's partials in block order
        float total = 0.f;
        if ((int)threadIdx.x < a.nterms) {
            const int j = threadIdx.x;
            float sj = 0.f;
            for (int b = a.first_block[j]; b < a.first_block[j + 1]; ++b)
                sj += __hip_atomic_load(a.partial + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const float mean = a.t[j].n ? sj / (float)a.t[j].n : 0.f;
            a.out[1 + MG_LOSS_GROUPS + j] = mean;
            total = a.t[j].weight * mean;
        }
        // terms in index order: every lane adds them one by one (nterms <= 16); lane 0 writes
        float acc = 0.f, grp[MG_LOSS_GROUPS];
#pragma unroll
        for (int q = 0; q < MG_LOSS_GROUPS; ++q) grp[q] = 0.f;
        for (int j = 0; j < a.nterms; ++j) {
            const float w = __shfl(total, j, 64);
            acc += w;
#pragma unroll
            for (int q = 0; q < MG_LOSS_GROUPS; ++q)
                if (a.t[j].group == q) grp[q] += w;
        }
        if (threadIdx.x == 0) {
            a.out[0] = acc;
#pragma unroll
            for (int q = 0; q < MG_LOSS_GROUPS; ++q) a.out[1 + q] = grp[q];
            *a.ticket = 0u;
        }
    }
}

__global__ __launch_bounds__(256) void multi_loss_bwd_kernel(MultiLossArgs a)
{
    const int k = ml_term_of(a, blockIdx.x);
    const MgLossTerm t = a.t[k];
    if (!t.da) return;
    const int nb = a.first_block[k + 1] - a.first_block[k], lb = blockIdx.x - a.first_block[k];
    const float kk = a.g[0] * t.weight / (float)t.n;
    for (size_t i = (size_t)lb * 256 + threadIdx.x; i < t.n; i += (size_t)nb * 256) {
        const float d = t.mode == 0 ? t.a[i] - t.c : t.a[i] - t.b[i];
        t.da[i] = t.mode == 0 ? 2.f * kk * d : (d > 0.f ? kk : (d < 0.f ? -kk : 0.f));
    }
}

static int multi_loss_fill(MultiLossArgs &a, const MgLossTerm *terms, int nterms, int max_blocks_per_term)
{
    if (!terms || nterms < 1 || nterms > MG_LOSS_MAX_TERMS) return MG_ERR_ARG;
    int at = 0;
    for (int k = 0; k < nterms; ++k) {
        const MgLossTerm &t = terms[k];
        if (!t.a || (t.mode == 1 && !t.b) || t.mode < 0 || t.mode > 1 || t.group < 0 || t.group >= MG_LOSS_GROUPS)
            return MG_ERR_ARG;
        if (t.n == 0) return MG_ERR_SHAPE;
        a.t[k] = t;
        a.first_block[k] = at;
        const size_t want = (t.n + 2047) / 2048;   // ~8 elements per thread
        at += (int)(want < 1 ? 1 : (want > (size_t)max_blocks_per_term ? (size_t)max_blocks_per_term : want));
    }
    a.first_block[nterms] = at;
    a.nterms = nterms;
    return MG_OK;
}

extern "C" size_t mg_multi_loss_scratch_floats(void) { return (size_t)MG_LOSS_MAX_TERMS * 64 + 1; }

extern "C" int mg_multi_loss_fwd(const MgLossTerm *terms, int nterms, float *scratch, float *out, void *stream)
{
    if (!scratch || !out) return MG_ERR_ARG;
    MultiLossArgs a;
    MG_TRY(multi_loss_fill(a, terms, nterms, 64));
    a.partial = scratch;
    a.ticket = reinterpret_cast<unsigned *>(scratch + (size_t)MG_LOSS_MAX_TERMS * 64);
    a.out = out;
    a.g = nullptr;
    hipLaunchKernelGGL(multi_loss_fwd_kernel, dim3(a.first_block[nterms]), dim3(256), 0, (hipStream_t)stream, a);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_multi_loss_bwd(const MgLossTerm *terms, int nterms, const float *g, void *stream)
{
    if (!g) return MG_ERR_ARG;
    MultiLossArgs a;
    MG_TRY(multi_loss_fill(a, terms, nterms, 256));
    a.partial = nullptr;
    a.ticket = nullptr;
    a.out = nullptr;
    a.g = g;
    hipLaunchKernelGGL(multi_loss_bwd_kernel, dim3(a.first_block[nterms]), dim3(256), 0, (hipStream_t)stream, a);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
