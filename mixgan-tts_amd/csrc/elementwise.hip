// HBM-bound diffusion algebra: diffuse_fn/q_sample, q_posterior_sample, (de)norm + transposes.
// One pass over each tensor, 16-byte accesses where the row pitch allows, LDS-tiled transposes so
// that both the [B,L,M] side and the [B,M,L] side are read/written in full 256-byte segments.
#include "common.h"

#define TL 64  // frames per transpose tile

// norm_spec, exactly the reference's op order (model/diffusion.py:228-229)
__device__ __forceinline__ float norm1(float x, float mn, float mx) { return (x - mn) / (mx - mn) * 2.f - 1.f; }
// denorm_spec (model/diffusion.py:231-232)
__device__ __forceinline__ float denorm1(float x, float mn, float mx) { return (x + 1.f) / 2.f * (mx - mn) + mn; }

// ---------------------------------------------------------------------------------------------
// mel [B,L,M] -> x_t [B,M,L]
// ---------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void diffuse_kernel(const float *__restrict__ mel, const int64_t *__restrict__ t,
                                                      const float *__restrict__ noise,
                                                      const uint8_t *__restrict__ keep,
                                                      const float *__restrict__ spec_min,
                                                      const float *__restrict__ spec_max,
                                                      const float *__restrict__ sqrt_ac,
                                                      const float *__restrict__ sqrt_1mac, float *__restrict__ out,
                                                      int L, int M, int T)
{
    extern __shared__ float tile[];  // [TL][M+1]
    const int b = blockIdx.y;
    const int l0 = blockIdx.x * TL;
    const int nl = min(TL, L - l0);
    const int P = M + 1;
    constexpr int W = VEC ? 4 : 1;
    const float *src = mel + ((size_t)b * L + l0) * M;
    for (int idx = threadIdx.x; idx < nl * M / W; idx += 256) {
        const int e = idx * W;
        const int l = e / M, m = e - l * M;
        if (VEC) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(src + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) tile[l * P + m + j] = norm1(q[j], spec_min[m + j], spec_max[m + j]);
        } else {
            tile[l * P + m] = norm1(src[e], spec_min[m], spec_max[m]);
        }
    }
    __syncthreads();
    long tb = t[b];
    const bool clean = tb < 0;  // model/diffusion.py:180-184: x_{-1} is the ground-truth mel
    if (tb < 0) tb = 0;
    if (tb >= T) tb = T - 1;
    const float ca = sqrt_ac[tb], cb = sqrt_1mac[tb];
    for (int idx = threadIdx.x; idx < M * TL / W; idx += 256) {
        const int m = idx / (TL / W), l = (idx - m * (TL / W)) * W;
        if (l >= nl) continue;
        const size_t o = ((size_t)b * M + m) * L + l0 + l;
        float nz[W], v[W];
        if (VEC) {
            const f32x4 q = *reinterpret_cast<const f32x4 *>(noise + o);
#pragma unroll
            for (int j = 0; j < W; ++j) nz[j] = q[j];
        } else {
            nz[0] = clean ? 0.f : noise[o];
        }
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float x0 = tile[(l + j) * P + m];
            v[j] = clean ? x0 : ca * x0 + cb * nz[j];
            if (keep) v[j] *= (float)keep[(size_t)b * L + l0 + l + j];
        }
        if (VEC) {
            f32x4 r = {v[0], v[W > 1 ? 1 : 0], v[W > 2 ? 2 : 0], v[W > 3 ? 3 : 0]};
            *reinterpret_cast<f32x4 *>(out + o) = r;
        } else {
            out[o] = v[0];
        }
    }
}

extern "C" int mg_diffuse_fwd(const float *mel, const int64_t *t, const float *noise, const uint8_t *keep,
                              const float *spec_min, const float *spec_max, const float *sqrt_ac,
                              const float *sqrt_1mac, float *out, int B, int L, int M, int T, void *stream)
{
    if (!mel || !t || !noise || !spec_min || !spec_max || !sqrt_ac || !sqrt_1mac || !out) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || M <= 0 || M > 1024 || T <= 0) return MG_ERR_SHAPE;
    dim3 grid(mg_cdiv(L, TL), B);
    const bool vec = M % 4 == 0 && L % 4 == 0 && ((((uintptr_t)mel | (uintptr_t)noise | (uintptr_t)out) & 15) == 0);
    if (vec)
        hipLaunchKernelGGL(diffuse_kernel<true>, grid, dim3(256), (size_t)TL * (M + 1) * sizeof(float), (hipStream_t)stream,
                           mel, t, noise, keep, spec_min, spec_max, sqrt_ac, sqrt_1mac, out, L, M, T);
    else
        hipLaunchKernelGGL(diffuse_kernel<false>, grid, dim3(256), (size_t)TL * (M + 1) * sizeof(float), (hipStream_t)stream,
                           mel, t, noise, keep, spec_min, spec_max, sqrt_ac, sqrt_1mac, out, L, M, T);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// posterior sample, all tensors [B,M,L]
// ---------------------------------------------------------------------------------------------
template <int V>
__global__ __launch_bounds__(256) void posterior_kernel(const float *__restrict__ x0, const float *__restrict__ xt,
                                                        const int64_t *__restrict__ t,
                                                        const float *__restrict__ noise,
                                                        const uint8_t *__restrict__ keep,
                                                        const float *__restrict__ coef1,
                                                        const float *__restrict__ coef2,
                                                        const float *__restrict__ logvar, float *__restrict__ out,
                                                        float *__restrict__ x0c_out, int clip, int L, int ML, int T)
{
    // grid.y = batch; grid.x strides over M*L/V vectors of this batch element
    const int b = blockIdx.y;
    long tb = t[b];
    if (tb < 0) tb = 0;
    if (tb >= T) tb = T - 1;
    const float c1 = coef1[tb], c2 = coef2[tb];
    const float sg = tb == 0 ? 0.f : expf(0.5f * logvar[tb]);
    const size_t base = (size_t)b * ML;
    const int nvec = ML / V;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nvec; i += gridDim.x * 256) {
        const int e0 = i * V;
        float a[V], x[V], n[V], k[V];
        if (V == 4) {
            const f32x4 av = *reinterpret_cast<const f32x4 *>(x0 + base + e0);
            const f32x4 xv = *reinterpret_cast<const f32x4 *>(xt + base + e0);
            const f32x4 nv = *reinterpret_cast<const f32x4 *>(noise + base + e0);
#pragma unroll
            for (int j = 0; j < V; ++j) {
                a[j] = av[j];
                x[j] = xv[j];
                n[j] = nv[j];
            }
        } else {
            a[0] = x0[base + e0];
            x[0] = xt[base + e0];
            n[0] = noise[base + e0];
        }
        const int l = e0 % L;  // V divides L on the vector path, so the V elements share a row
#pragma unroll
        for (int j = 0; j < V; ++j) k[j] = keep ? (float)keep[(size_t)b * L + l + j] : 1.f;
        float o[V], ac[V];
#pragma unroll
        for (int j = 0; j < V; ++j) {
            float s = a[j] * k[j];
            if (clip) s = fminf(fmaxf(s, -1.f), 1.f);
            ac[j] = s;
            o[j] = (c1 * s + c2 * x[j] + sg * n[j]) * k[j];
        }
        if (V == 4) {
            f32x4 ov = {o[0], o[1], o[2], o[3]};
            *reinterpret_cast<f32x4 *>(out + base + e0) = ov;
            if (x0c_out) {
                f32x4 cv = {ac[0], ac[1], ac[2], ac[3]};
                *reinterpret_cast<f32x4 *>(x0c_out + base + e0) = cv;
            }
        } else {
            out[base + e0] = o[0];
            if (x0c_out) x0c_out[base + e0] = ac[0];
        }
    }
}

extern "C" int mg_posterior_sample_fwd(const float *x0, const float *x_t, const int64_t *t, const float *noise,
                                       const uint8_t *keep, const float *coef1, const float *coef2,
                                       const float *logvar, float *out, float *x0_clamped_out, int clip, int B, int L,
                                       int M, int T, void *stream)
{
    if (!x0 || !x_t || !t || !noise || !coef1 || !coef2 || !logvar || !out) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || M <= 0 || T <= 0) return MG_ERR_SHAPE;
    const int ML = M * L;
    const bool vec = (L % 4 == 0) && ((((uintptr_t)x0 | (uintptr_t)x_t | (uintptr_t)noise | (uintptr_t)out |
                                        (uintptr_t)x0_clamped_out) & 15) == 0);
    const int V = vec ? 4 : 1;
    dim3 grid(min(mg_cdiv(ML / V, 256), 512), B);
    if (vec)
        hipLaunchKernelGGL(posterior_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, x0, x_t, t, noise, keep, coef1,
                           coef2, logvar, out, x0_clamped_out, clip, L, ML, T);
    else
        hipLaunchKernelGGL(posterior_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x0, x_t, t, noise, keep, coef1,
                           coef2, logvar, out, x0_clamped_out, clip, L, ML, T);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// [B,M,L] <-> [B,L,M] with optional norm/denorm and keep mask
// ---------------------------------------------------------------------------------------------
// VEC: 16-byte global accesses on both sides (M % 4 == 0, L % 4 == 0, 16-byte aligned bases and batch stride); the LDS
// tile keeps its odd pitch, so a vector is 4 scalar LDS accesses.  Same arithmetic as the scalar path.
template <bool VEC, int TLV>
__global__ __launch_bounds__(256) void transpose_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                        const float *__restrict__ spec_min,
                                                        const float *__restrict__ spec_max,
                                                        const uint8_t *__restrict__ keep, int to_blm, int mode, int L,
                                                        int M, long bml_bs)
{
    // bml_bs: batch stride (floats) of the [B,M,L] side, so it may be a channel slice of a wider tensor
    extern __shared__ float tile[];  // [TLV][M+1]
    const int b = blockIdx.y;
    const int l0 = blockIdx.x * TLV;
    const int nl = min(TLV, L - l0);
    const int P = M + 1;
    constexpr int W = VEC ? 4 : 1;
    if (to_blm) {
        // read [M][nl] rows of the BML tensor
        for (int idx = threadIdx.x; idx < M * TLV / W; idx += 256) {
            const int m = idx / (TLV / W), l = (idx - m * (TLV / W)) * W;
            if (l >= nl) continue;
            const float *p = in + (size_t)b * bml_bs + (size_t)m * L + l0 + l;
            if (VEC) {
                const f32x4 v = *reinterpret_cast<const f32x4 *>(p);
#pragma unroll
                for (int j = 0; j < 4; ++j) tile[(l + j) * P + m] = v[j];
            } else {
                tile[l * P + m] = *p;
            }
        }
        __syncthreads();
        float *dst = out + ((size_t)b * L + l0) * M;
        for (int idx = threadIdx.x; idx < nl * M / W; idx += 256) {
            const int e = idx * W;
            const int l = e / M, m = e - l * M;
            const float kp = keep ? (float)keep[(size_t)b * L + l0 + l] : 1.f;
            float v[W];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                v[j] = tile[l * P + m + j];
                if (mode == 2) v[j] = denorm1(v[j], spec_min[m + j], spec_max[m + j]);
                if (keep) v[j] *= kp;
            }
            if (VEC) {
                f32x4 o = {v[0], v[W > 1 ? 1 : 0], v[W > 2 ? 2 : 0], v[W > 3 ? 3 : 0]};
                *reinterpret_cast<f32x4 *>(dst + e) = o;
            } else {
                dst[e] = v[0];
            }
        }
    } else {
        const float *src = in + ((size_t)b * L + l0) * M;
        for (int idx = threadIdx.x; idx < nl * M / W; idx += 256) {
            const int e = idx * W;
            const int l = e / M, m = e - l * M;
            float v[W];
            if (VEC) {
                const f32x4 q = *reinterpret_cast<const f32x4 *>(src + e);
#pragma unroll
                for (int j = 0; j < W; ++j) v[j] = q[j];
            } else {
                v[0] = src[e];
            }
#pragma unroll
            for (int j = 0; j < W; ++j) {
                if (mode == 1) v[j] = norm1(v[j], spec_min[m + j], spec_max[m + j]);
                tile[l * P + m + j] = v[j];
            }
        }
        __syncthreads();
        for (int idx = threadIdx.x; idx < M * TLV / W; idx += 256) {
            const int m = idx / (TLV / W), l = (idx - m * (TLV / W)) * W;
            if (l >= nl) continue;
            float v[W];
#pragma unroll
            for (int j = 0; j < W; ++j) {
                v[j] = tile[(l + j) * P + m];
                if (keep) v[j] *= (float)keep[(size_t)b * L + l0 + l + j];
            }
            float *p = out + (size_t)b * bml_bs + (size_t)m * L + l0 + l;
            if (VEC) {
                f32x4 o = {v[0], v[W > 1 ? 1 : 0], v[W > 2 ? 2 : 0], v[W > 3 ? 3 : 0]};
                *reinterpret_cast<f32x4 *>(p) = o;
            } else {
                *p = v[0];
            }
        }
    }
}

extern "C" int mg_transpose_bml_strided(const float *in, float *out, const float *spec_min, const float *spec_max,
                                        const uint8_t *keep, int to_blm, int mode, int B, int L, int M, long bml_bs,
                                        void *stream);

extern "C" int mg_transpose_bml(const float *in, float *out, const float *spec_min, const float *spec_max,
                                const uint8_t *keep, int to_blm, int mode, int B, int L, int M, void *stream)
{
    return mg_transpose_bml_strided(in, out, spec_min, spec_max, keep, to_blm, mode, B, L, M, 0, stream);
}

extern "C" int mg_transpose_bml_strided(const float *in, float *out, const float *spec_min, const float *spec_max,
                                        const uint8_t *keep, int to_blm, int mode, int B, int L, int M, long bml_bs,
                                        void *stream)
{
    if (!in || !out) return MG_ERR_ARG;
    if (mode < 0 || mode > 2 || (mode != 0 && (!spec_min || !spec_max))) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || M <= 0 || M > 1024) return MG_ERR_SHAPE;
    dim3 grid(mg_cdiv(L, TL), B);
    const long bs = bml_bs ? bml_bs : (long)M * L;
    const bool vec = M % 4 == 0 && L % 4 == 0 && bs % 4 == 0 && ((((uintptr_t)in | (uintptr_t)out) & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    // measured (B=128, L=4000, M=80): [B,L,M] -> [B,M,L] goes from 4.1 to 5.6 TB/s with 16-byte accesses; the opposite
    // direction (strided 256-byte row reads) is slightly SLOWER vectorised (2.3 vs 2.6 TB/s) and does not improve with
    // 128-frame tiles either, so it keeps the scalar path
    const size_t lds = (size_t)TL * (M + 1) * sizeof(float);
    if (vec && !to_blm)
        hipLaunchKernelGGL((transpose_kernel<true, TL>), grid, dim3(256), lds, st, in, out, spec_min, spec_max, keep, to_blm, mode, L, M, bs);
    else
        hipLaunchKernelGGL((transpose_kernel<false, TL>), grid, dim3(256), lds, st, in, out, spec_min, spec_max, keep, to_blm, mode, L, M, bs);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// norm_spec / denorm_spec on [..., M] tensors and their gradients
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void spec_affine_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          const float *__restrict__ spec_min,
                                                          const float *__restrict__ spec_max, int mode, size_t n, int M)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i % M);
        const float mn = spec_min[m], mx = spec_max[m], v = in[i];
        float r;
        if (mode == 1) r = norm1(v, mn, mx);
        else if (mode == 2) r = denorm1(v, mn, mx);
        else if (mode == 3) r = v / (mx - mn) * 2.f;   // d norm_spec / dx
        else r = v / 2.f * (mx - mn);                  // d denorm_spec / dx
        out[i] = r;
    }
}

extern "C" int mg_spec_affine(const float *in, float *out, const float *spec_min, const float *spec_max, int mode,
                              size_t n, int M, void *stream)
{
    if (!in || !out || !spec_min || !spec_max) return MG_ERR_ARG;
    if (mode < 1 || mode > 4) return MG_ERR_ARG;
    if (M <= 0) return MG_ERR_SHAPE;
    if (n == 0) return MG_OK;
    const int blocks = (int)((n + 255) / 256 < 2048 ? (n + 255) / 256 : 2048);
    hipLaunchKernelGGL(spec_affine_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, in, out, spec_min, spec_max,
                       mode, n, M);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// gradient of diffuse_trace (model/diffusion.py:167-175) w.r.t. x_start [B, L, M]:
//   trace[0]   = clamp(norm(x), -1, 1) * keep            -> slope_m * keep * 1[|norm(x)| <= 1]
//   trace[t+1] = (sqrt_ac[t] norm(x) + sqrt(1-ac[t]) eps) * keep -> slope_m * keep * sqrt_ac[t]
// g: [T+1, B, L, M] stacked output gradients; slope_m = 2 / (spec_max[m] - spec_min[m]).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void diffuse_trace_bwd_kernel(const float *__restrict__ g, const float *__restrict__ x,
                                                                const float *__restrict__ spec_min,
                                                                const float *__restrict__ spec_max,
                                                                const uint8_t *__restrict__ keep,
                                                                const float *__restrict__ sqrt_ac, float *__restrict__ dx,
                                                                int T, size_t n, int M)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const int m = (int)(i % M);
        const float mn = spec_min[m], mx = spec_max[m];
        const float nv = norm1(x[i], mn, mx);
        float a = fabsf(nv) <= 1.f ? g[i] : 0.f;
        for (int t = 0; t < T; ++t) a = fmaf(sqrt_ac[t], g[(size_t)(t + 1) * n + i], a);
        dx[i] = (keep && !keep[i / M]) ? 0.f : a * 2.f / (mx - mn);
    }
}

extern "C" int mg_diffuse_trace_bwd(const float *g, const float *x_start, const float *spec_min, const float *spec_max,
                                    const uint8_t *keep, const float *sqrt_alphas_cumprod, float *d_x, int T, int B, int L,
                                    int M, void *stream)
{
    if (!g || !x_start || !spec_min || !spec_max || !sqrt_alphas_cumprod || !d_x) return MG_ERR_ARG;
    if (T < 0 || B <= 0 || L <= 0 || M <= 0) return MG_ERR_SHAPE;
    const size_t n = (size_t)B * L * M;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(diffuse_trace_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, x_start, spec_min,
                       spec_max, keep, sqrt_alphas_cumprod, d_x, T, n, M);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
// gradient of mg_posterior_sample_fwd w.r.t. x0 (the only differentiable input on the path)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void posterior_bwd_kernel(const float *__restrict__ x0,
                                                            const int64_t *__restrict__ t,
                                                            const uint8_t *__restrict__ keep,
                                                            const float *__restrict__ coef1,
                                                            const float *__restrict__ g_x0c,
                                                            const float *__restrict__ g_xpp, float *__restrict__ gx0,
                                                            int clip, int L, int ML, int T)
{
    const int b = blockIdx.y;
    long tb = t[b];
    if (tb < 0) tb = 0;
    if (tb >= T) tb = T - 1;
    const float c1 = coef1[tb];
    const size_t base = (size_t)b * ML;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ML; i += gridDim.x * 256) {
        const float k = keep ? (float)keep[(size_t)b * L + (i % L)] : 1.f;
        const float s = x0[base + i] * k;
        const bool pass = !clip || (s >= -1.f && s <= 1.f);  // torch.clamp passes the gradient on [min, max]
        float g = g_x0c ? g_x0c[base + i] : 0.f;
        if (g_xpp) g += c1 * k * g_xpp[base + i];
        gx0[base + i] = pass ? g * k : 0.f;
    }
}

extern "C" int mg_posterior_sample_bwd(const float *x0, const int64_t *t, const uint8_t *keep, const float *coef1,
                                       const float *g_x0c, const float *g_xpp, float *g_x0, int clip, int B, int L,
                                       int M, int T, void *stream)
{
    if (!x0 || !t || !coef1 || !g_x0 || (!g_x0c && !g_xpp)) return MG_ERR_ARG;
    if (B <= 0 || L <= 0 || M <= 0 || T <= 0) return MG_ERR_SHAPE;
    dim3 grid(min(mg_cdiv(M * L, 256), 512), B);
    hipLaunchKernelGGL(posterior_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, x0, t, keep, coef1, g_x0c, g_xpp,
                       g_x0, clip, L, M * L, T);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int mg_version(void) { return MG_VERSION; }

extern "C" const char *mg_error_string(int code)
{
    switch (code) {
    case MG_OK: return "ok";
    case MG_ERR_ARG: return "invalid argument (null pointer or bad enum)";
    case MG_ERR_SHAPE: return "unsupported shape";
    case MG_ERR_WORKSPACE: return "workspace too small";
    default: break;
    }
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "unknown error";
}
