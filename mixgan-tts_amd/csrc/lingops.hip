// Device-side versions of the four host-loop index functions of the (otherwise out-of-scope)
// LinguisticEncoder -- SURVEY.md section 8 f1, the first "next" row: each of them walks the batch
// in Python with one .item() per phoneme / word and serialises the GPU in both generator forwards
// of a training step.
//   word_level_pooling  utils/tools.py:394-413            (sum / mean of phoneme rows per word)
//   LengthRegulator     model/linguistic_encoder.py:383-416 (repeat word rows by duration, pad / crop)
//   get_mapping_mask    model/linguistic_encoder.py:185-199 (frames-of-word x phonemes-of-word blocks)
//   get_rel_coef        model/linguistic_encoder.py:222-236 (position in segment / segment length)
// All are integer prefix sums + gathers: HBM-bound byte movement.  A workgroup rebuilds the (short)
// inclusive prefix sum of its batch row in LDS and binary-searches it; rows are [.., H]
// frame-major with H contiguous, so every copy is a coalesced row.
#include "common.h"

#define LG_MAXT 4096  // longest duration row held in LDS

// inclusive prefix sum of max(v[i], lo) over i < n into cum[] (cum[i] = sum_{j<=i}); returns the total
__device__ __forceinline__ long lg_scan(const int64_t *__restrict__ v, int n, long lo, long *cum)
{
    if (threadIdx.x == 0) {
        long s = 0;
        for (int i = 0; i < n; ++i) {
            long d = v[i];
            if (d < lo) d = lo;
            s += d;
            cum[i] = s;
        }
    }
    __syncthreads();
    return n > 0 ? cum[n - 1] : 0;
}

// first i in [0, n) with cum[i] > f  (n if none)
__device__ __forceinline__ int lg_upper(const long *cum, int n, long f)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cum[mid] > f) hi = mid;
        else lo = mid + 1;
    }
    return lo;
}

// ------------------------------------------------------------------------------------ LengthRegulator
__global__ __launch_bounds__(256) void length_regulate_kernel(const float *__restrict__ x, const int64_t *__restrict__ dur,
                                                              float *__restrict__ out, int64_t *__restrict__ mel_len,
                                                              int Tw, int H, int Lmax)
{
    __shared__ long cum[LG_MAXT];
    const int b = blockIdx.y;
    const long total = lg_scan(dur + (size_t)b * Tw, Tw, 0, cum);
    if (blockIdx.x == 0 && threadIdx.x == 0) mel_len[b] = total;
    const int f0 = blockIdx.x * 16;
    for (int f = f0; f < min(f0 + 16, Lmax); ++f) {
        float *o = out + ((size_t)b * Lmax + f) * H;
        if (f < total) {
            const int w = lg_upper(cum, Tw, f);
            const float *src = x + ((size_t)b * Tw + w) * H;
            for (int h = threadIdx.x; h < H; h += 256) o[h] = src[h];
        } else {
            for (int h = threadIdx.x; h < H; h += 256) o[h] = 0.f;
        }
    }
}

__global__ __launch_bounds__(256) void length_regulate_bwd_kernel(const float *__restrict__ dout,
                                                                  const int64_t *__restrict__ dur,
                                                                  float *__restrict__ dx, int Tw, int H, int Lmax)
{
    __shared__ long cum[LG_MAXT];
    const int b = blockIdx.y, w = blockIdx.x;
    lg_scan(dur + (size_t)b * Tw, Tw, 0, cum);
    const long s = w ? cum[w - 1] : 0;
    const long e = min(cum[w], (long)Lmax);
    for (int h = threadIdx.x; h < H; h += 256) {
        float acc = 0.f;
        for (long f = s; f < e; ++f) acc += dout[((size_t)b * Lmax + f) * H + h];
        dx[((size_t)b * Tw + w) * H + h] = acc;
    }
}

extern "C" int mg_length_regulate_fwd(const float *x, const int64_t *dur, float *out, int64_t *mel_len, int B, int Tw,
                                      int H, int Lmax, void *stream)
{
    if (!x || !dur || !out || !mel_len) return MG_ERR_ARG;
    if (B <= 0 || Tw <= 0 || Tw > LG_MAXT || H <= 0 || Lmax <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(length_regulate_kernel, dim3(mg_cdiv(Lmax, 16), B), dim3(256), 0, (hipStream_t)stream, x, dur, out,
                       mel_len, Tw, H, Lmax);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_length_regulate_bwd(const float *dout, const int64_t *dur, float *dx, int B, int Tw, int H, int Lmax,
                                      void *stream)
{
    if (!dout || !dur || !dx) return MG_ERR_ARG;
    if (B <= 0 || Tw <= 0 || Tw > LG_MAXT || H <= 0 || Lmax <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(length_regulate_bwd_kernel, dim3(Tw, B), dim3(256), 0, (hipStream_t)stream, dout, dur, dx, Tw, H,
                       Lmax);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ------------------------------------------------------------------------------------ word_level_pooling
// out[b, w, :] = sum (or mean) over the phonemes of word w; rows w >= src_w_len[b] are zero.
// backward (dsrc != null): dsrc[b, p, :] = dout[b, word(p), :] (/ n), phonemes past the last word get 0.
__global__ __launch_bounds__(256) void word_pool_kernel(const float *__restrict__ src, const int64_t *__restrict__ wb,
                                                        const int64_t *__restrict__ src_w_len, float *__restrict__ out,
                                                        const float *__restrict__ dout, float *__restrict__ dsrc, int Tp,
                                                        int Tw, int H, int Wout, int mean)
{
    __shared__ long cum[LG_MAXT];
    const int b = blockIdx.y, w = blockIdx.x;
    const int wl = (int)min((long)src_w_len[b], (long)Tw);
    const long total = lg_scan(wb + (size_t)b * Tw, wl, 0, cum);
    const bool live = w < wl;
    const long s = live ? (w ? cum[w - 1] : 0) : 0;
    const long e = live ? min(cum[w], (long)Tp) : 0;
    const float n = (float)(live ? (cum[w] - s) : 0);
    if (!dsrc) {
        for (int h = threadIdx.x; h < H; h += 256) {
            float acc = 0.f;
            for (long p = s; p < e; ++p) acc += src[((size_t)b * Tp + p) * H + h];  // in order, like torch.sum
            if (mean && live) acc = acc / n;                                          // n == 0 -> NaN, as the reference
            out[((size_t)b * Wout + w) * H + h] = acc;
        }
    } else {
        for (int h = threadIdx.x; h < H; h += 256) {
            float g = live ? dout[((size_t)b * Wout + w) * H + h] : 0.f;
            if (mean && live) g = g / n;
            for (long p = s; p < e; ++p) dsrc[((size_t)b * Tp + p) * H + h] = g;
        }
        // phonemes that belong to no word: cleared by the block of the last word slot
        if (w == Wout - 1)
            for (long p = min(total, (long)Tp); p < Tp; ++p)
                for (int h = threadIdx.x; h < H; h += 256) dsrc[((size_t)b * Tp + p) * H + h] = 0.f;
    }
}

extern "C" int mg_word_pool_fwd(const float *src, const int64_t *wb, const int64_t *src_w_len, float *out, int B, int Tp,
                                int Tw, int H, int Wout, int mean, void *stream)
{
    if (!src || !wb || !src_w_len || !out) return MG_ERR_ARG;
    if (B <= 0 || Tp <= 0 || Tw <= 0 || Tw > LG_MAXT || H <= 0 || Wout <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(word_pool_kernel, dim3(Wout, B), dim3(256), 0, (hipStream_t)stream, src, wb, src_w_len, out, nullptr,
                       nullptr, Tp, Tw, H, Wout, mean);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_word_pool_bwd(const float *dout, const int64_t *wb, const int64_t *src_w_len, float *dsrc, int B, int Tp,
                                int Tw, int H, int Wout, int mean, void *stream)
{
    if (!dout || !wb || !src_w_len || !dsrc) return MG_ERR_ARG;
    if (B <= 0 || Tp <= 0 || Tw <= 0 || Tw > LG_MAXT || H <= 0 || Wout <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(word_pool_kernel, dim3(Wout, B), dim3(256), 0, (hipStream_t)stream, nullptr, wb, src_w_len, nullptr,
                       dout, dsrc, Tp, Tw, H, Wout, mean);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ------------------------------------------------------------------------------------ get_mapping_mask
// out[b, q, k] = 1 iff frame q and phoneme k belong to the same word i < src_w_len[b]
__global__ __launch_bounds__(256) void mapping_mask_kernel(const int64_t *__restrict__ dur_w, const int64_t *__restrict__ wb,
                                                           const int64_t *__restrict__ src_w_len, uint8_t *__restrict__ out,
                                                           int Tw, int Lq, int Lkv)
{
    __shared__ long cw[LG_MAXT / 2], cp[LG_MAXT / 2];
    const int b = blockIdx.y;
    const int l = (int)min((long)src_w_len[b], (long)Tw);
    const long tq = lg_scan(dur_w + (size_t)b * Tw, l, -(1L << 40), cw);  // cumsum of the raw values, as torch.cumsum
    lg_scan(wb + (size_t)b * Tw, l, -(1L << 40), cp);
    const int q0 = blockIdx.x * 8;
    for (int q = q0; q < min(q0 + 8, Lq); ++q) {
        long ks = 0, ke = 0;
        if (q < tq) {
            const int i = lg_upper(cw, l, q);
            ks = i ? cp[i - 1] : 0;
            ke = cp[i];
        }
        uint8_t *o = out + ((size_t)b * Lq + q) * Lkv;
        for (int k = threadIdx.x; k < Lkv; k += 256) o[k] = (k >= ks && k < ke) ? 1 : 0;
    }
}

extern "C" int mg_mapping_mask(const int64_t *dur_w, const int64_t *wb, const int64_t *src_w_len, uint8_t *out, int B,
                               int Tw, int Lq, int Lkv, void *stream)
{
    if (!dur_w || !wb || !src_w_len || !out) return MG_ERR_ARG;
    if (B <= 0 || Tw <= 0 || Tw > LG_MAXT / 2 || Lq <= 0 || Lkv <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(mapping_mask_kernel, dim3(mg_cdiv(Lq, 8), B), dim3(256), 0, (hipStream_t)stream, dur_w, wb, src_w_len,
                       out, Tw, Lq, Lkv);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ------------------------------------------------------------------------------------ get_rel_coef
// out[b, j] = idx / seg with idx = position of j inside its segment, seg = that segment's length,
// seg forced to 1 where mask[b, j] == 0; positions past the last segment have idx = seg = 0.
__global__ __launch_bounds__(256) void rel_coef_kernel(const int64_t *__restrict__ dur, const int64_t *__restrict__ dur_len,
                                                       const uint8_t *__restrict__ mask, float *__restrict__ out, int T,
                                                       int Lout)
{
    __shared__ long cum[LG_MAXT];
    const int b = blockIdx.y;
    const int n = (int)min((long)dur_len[b], (long)T);
    const long total = lg_scan(dur + (size_t)b * T, n, 0, cum);
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= Lout) return;
    long idx = 0, seg = 0;
    if (j < total) {
        const int i = lg_upper(cum, n, j);
        const long s = i ? cum[i - 1] : 0;
        idx = j - s;
        seg = cum[i] - s;
    }
    if (!mask[(size_t)b * Lout + j]) seg = 1;
    out[(size_t)b * Lout + j] = (float)idx / (float)seg;
}

extern "C" int mg_rel_coef(const int64_t *dur, const int64_t *dur_len, const uint8_t *mask, float *out, int B, int T,
                           int Lout, void *stream)
{
    if (!dur || !dur_len || !mask || !out) return MG_ERR_ARG;
    if (B <= 0 || T <= 0 || T > LG_MAXT || Lout <= 0) return MG_ERR_SHAPE;
    hipLaunchKernelGGL(rel_coef_kernel, dim3(mg_cdiv(Lout, 256), B), dim3(256), 0, (hipStream_t)stream, dur, dur_len, mask,
                       out, T, Lout);
    MG_LAUNCH_CHECK();
    return MG_OK;
}
