// Shared device/host helpers for libmixgan_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "../../include/mixgan_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define MG_LAUNCH_CHECK()                          \
    do {                                           \
        hipError_t e__ = hipGetLastError();        \
        if (e__ != hipSuccess) return (int)e__;    \
    } while (0)

#define MG_TRY(expr)                \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != MG_OK) return rc__; \
    } while (0)

static inline int mg_cdiv(int a, int b) { return (a + b - 1) / b; }
static inline size_t mg_align_up(size_t a, size_t b) { return (a + b - 1) / b * b; }

// v_rcp_f32 (1 ulp) instead of the IEEE division sequence (~10 VALU instructions): the gate epilogue evaluates 64
// of these per lane per layer between two barriers, where nothing overlaps them.
__device__ __forceinline__ float mg_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
// tanh via one exp: 1 - 2/(exp(2x)+1); exact limits at +-inf, abs err ~1e-7.
__device__ __forceinline__ float mg_tanh(float x) { return 1.0f - 2.0f * __builtin_amdgcn_rcpf(__expf(2.0f * x) + 1.0f); }

template <int ACT>
__device__ __forceinline__ float mg_act(float v)
{
    if (ACT == MG_ACT_RELU) return v > 0.f ? v : 0.f;
    if (ACT == MG_ACT_LRELU02) return v > 0.f ? v : 0.2f * v;
    if (ACT == MG_ACT_TANH) return tanhf(v);
    return v;
}
