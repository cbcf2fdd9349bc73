// Shared pieces of the two convolution kernels (fp32 MFMA: tcs_conv.hip, fp16-split MFMA: tcs_conv_f16.hip).
#pragma once
#include "tcs_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
    const float* src[TCS_MAX_SRC];
    int src_ch[TCS_MAX_SRC];
    int src_end[TCS_MAX_SRC];
    const float* w;
    const float* bias;
    int B, H, W, Cin, Cout, CoutPad;
    int act;
    float post_scale;
    const float* add1;
    const float* add2;
    const float* h;
    const float* z;
    int keep_z, hidden;
    float* out;
    int out_ctot, out_coff;
    float* out2;
    int npx, npatch, nct;
    float w_unscale;        // fp16-split kernel: 2^-s undoing the weight pre-scale (1 for the fp32 kernel)
};

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ float apply_act(float v, int act) {
    switch (act) {
        case TCS_ACT_RELU: return fmaxf(v, 0.f);
        case TCS_ACT_SIGMOID: return sigmoidf_(v);
        case TCS_ACT_TANH: return tanhf(v);
        case TCS_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        default: return v;
    }
}

static inline int cout_tile(int Cout) { return Cout > 64 ? 128 : (Cout > 32 ? 64 : 32); }
static inline int round_up(int a, int m) { return (a + m - 1) / m * m; }


// One output element: bias, addends, activation / GRU gate arithmetic (update.py:81-85, 30-34, 62-66), store.
template <int EPI>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, int b, int co, size_t pix, size_t HW, float acc) {
    float v = acc + (a.bias ? a.bias[co] : 0.f);
    if (EPI == TCS_EPI_LINEAR) {
        if (a.add1) v += a.add1[((size_t)b * a.Cout + co) * HW + pix];
        a.out[((size_t)b * a.out_ctot + a.out_coff + co) * HW + pix] = apply_act(v, a.act) * a.post_scale;
    } else if (EPI == TCS_EPI_GRU_ZR) {
        if (co < a.hidden) {
            const size_t o = ((size_t)b * a.hidden + co) * HW + pix;
            if (a.add1) v += a.add1[o];
            a.out[o] = sigmoidf_(v);
        } else {
            const size_t o = ((size_t)b * a.hidden + (co - a.hidden)) * HW + pix;
            if (a.add2) v += a.add2[o];
            a.out2[o] = sigmoidf_(v) * a.h[o];
        }
    } else {
        const size_t o = ((size_t)b * a.hidden + co) * HW + pix;
        if (a.add1) v += a.add1[o];
        const float q = tanhf(v), zz = a.z[o], hh = a.h[o];
        a.out[o] = a.keep_z ? zz * hh + (1.f - zz) * q : (1.f - zz) * hh + zz * q;
    }
}

int tcs_conv_f16x3_launch(ConvArgs& a, int ksize, int epilogue, hipStream_t s);   // tcs_conv_f16.hip
