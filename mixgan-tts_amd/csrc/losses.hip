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
