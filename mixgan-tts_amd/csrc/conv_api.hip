// Weight packing + the generic mg_conv1d_fwd entry point.
#include "conv_mfma.h"

// ---------------------------------------------------------------------------------------------
// pack: [Co, Ci, K] fp32 -> Wp[mb][q][lane][e]  (layout: conv_mfma.h header)
// ---------------------------------------------------------------------------------------------
__global__ void pack_conv_kernel(const float *__restrict__ w, float *__restrict__ wp, int Co, int Ci, int K, int CK,
                                 int CiP, int MB, int mode)
{
    const int Q = CiP * K / 8;
    const size_t total = (size_t)MB * Q * 256;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int e = (int)(idx & 3);
        const int lane = (int)((idx >> 2) & 63);
        const size_t gq = idx >> 8;
        const int q = (int)(gq % Q);
        const int mb = (int)(gq / Q);
        const int qc = K * (CK / 8);
        const int chunk = q / qc;
        const int rem = q - chunk * qc;
        const int tap = rem / (CK / 8);
        const int g = rem - tap * (CK / 8);
        const int ci = chunk * CK + g * 8 + 2 * e + (lane >> 5);
        const int r = lane & 31;
        float v = 0.f;
        if (mode == MG_PACK_PLAIN) {
            const int row = mb * 32 + r;
            if (row < Co && ci < Ci) v = w[((size_t)row * Ci + ci) * K + tap];
        } else if (mode == MG_PACK_GATE) {
            const int half = mb & 1, rr = (mb >> 1) * 32 + r;
            if (rr < Co / 2 && ci < Ci) v = w[((size_t)(half * (Co / 2) + rr) * Ci + ci) * K + tap];
        } else {  // MG_PACK_DGRAD: rows = source Ci, reduction = source Co, taps flipped
            const int row = mb * 32 + r;
            if (row < Ci && ci < Co) v = w[((size_t)ci * Ci + row) * K + (K - 1 - tap)];
        }
        wp[idx] = v;
    }
}

static int pack_dims(int Co, int Ci, int K, int mode, int *Mrows, int *Kin, int *MB)
{
    if (Co <= 0 || Ci <= 0 || !(K == 1 || K == 3 || K == 5 || K == 9)) return MG_ERR_SHAPE;
    if (mode == MG_PACK_PLAIN) {
        *Mrows = Co;
        *Kin = Ci;
        *MB = mg_conv_mblocks(Co);
    } else if (mode == MG_PACK_GATE) {
        if (Co % 2) return MG_ERR_SHAPE;
        *Mrows = Co;
        *Kin = Ci;
        *MB = mg_round_up(2 * mg_cdiv(Co / 2, 32), 4);
    } else if (mode == MG_PACK_DGRAD) {
        *Mrows = Ci;
        *Kin = Co;
        *MB = mg_conv_mblocks(Ci);
    } else
        return MG_ERR_ARG;
    return MG_OK;
}

extern "C" size_t mg_conv_packed_floats(int Co, int Ci, int K, int mode)
{
    int Mrows, Kin, MB;
    if (pack_dims(Co, Ci, K, mode, &Mrows, &Kin, &MB) != MG_OK) return 0;
    const int CiP = mg_round_up(Kin, mg_conv_ck(K));
    return (size_t)MB * (CiP * K / 8) * 256;
}

extern "C" int mg_conv_pack(const float *w, float *packed, int Co, int Ci, int K, int mode, void *stream)
{
    if (!w || !packed) return MG_ERR_ARG;
    int Mrows, Kin, MB;
    MG_TRY(pack_dims(Co, Ci, K, mode, &Mrows, &Kin, &MB));
    const int CK = mg_conv_ck(K);
    const int CiP = mg_round_up(Kin, CK);
    const size_t total = (size_t)MB * (CiP * K / 8) * 256;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(pack_conv_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, w, packed, Co, Ci, K, CK, CiP,
                       MB, mode);
    MG_LAUNCH_CHECK();
    return MG_OK;
}

extern "C" int mg_conv1d_fwd(const float *in, const float *in_vec, const float *packed, const float *bias,
                             const float *add, float *out, int B, int Ci, int Lin, int Co, int Lout, int K, int stride,
                             int pad, int act, float alpha, int accumulate, void *stream)
{
    if (!in || !packed || !out) return MG_ERR_ARG;
    if (act < 0 || act > MG_ACT_TANH) return MG_ERR_ARG;
    if (B <= 0 || Ci <= 0 || Co <= 0 || Lin <= 0 || Lout <= 0 || pad < 0) return MG_ERR_SHAPE;
    if ((Lin + 2 * pad - K) / stride + 1 < Lout) return MG_ERR_SHAPE;  // would read past the padded input
    ConvShape s{B, Ci, Lin, Lout, K, stride, pad, Co, 0, 0};
    EpiBiasAct::Params ep{out, bias, add, alpha, Co, act, accumulate};
    return conv_launch<EpiBiasAct>(s, in, in_vec, packed, ep, (hipStream_t)stream);
}
