// Shared pieces of the two convolution kernels (fp32 MFMA: tcs_conv.hip, fp16-split MFMA: tcs_conv_f16.hip).
#pragma once
#include "tcs_common.h"
#include "tcs_s16.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs {
    const float* src[TCS_MAX_SRC];
    int src_ch[TCS_MAX_SRC];
    int src_end[TCS_MAX_SRC];
    const float* w;
    const float* bias;
    int B, H, W, Cin, Cout, CoutPad;   // H, W: OUTPUT grid of the kernel (= input grid for stride 1)
    int Hin, Win;                      // input grid (stride-2 convolutions: H = (Hin-1)/2+1)
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
    int src_align8;         // every source boundary is a multiple of 8 channels (one source select per 8-channel group)
    int w_bytes;            // size of the packed weight buffer (buffer-load bound)
    float w_unscale;        // fp16-split kernel: 2^-s undoing the weight pre-scale (1 for the fp32 kernel)
    _Float16* out16;        // LINEAR: optional S16 copy of the output (tcs_s16.h), written at group offset out16_goff
    int out16_groups, out16_goff;
    int in_transform;       // k_conv7x7<3>: 1 = samples are read as 2 * (x / 255) - 1 (tc_stereo.py:101-102)
    const float* src_b2;    // k_conv7x7<3>: batch elements >= b_split come from this tensor
    int b_split;
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


// Pointers selected among kernel arguments lose their address space (hipcc then emits flat_load + conservative
// s_waitcnt vmcnt(0) between loads, which serialised the staging loads at full memory latency).  Casting back to the
// global address space gives global_load_dword with an SGPR base.
typedef const __attribute__((address_space(1))) float* gptr_t;

// base (SGPR pair) + 32-bit unsigned byte offset (VGPR): selects the `global_load v, v_off, s[base]` addressing form,
// which needs no per-load 64-bit VGPR address (those temporaries made hipcc emit s_waitcnt vmcnt(0) between loads).
typedef const __attribute__((address_space(1))) char* gbytes_t;
__device__ __forceinline__ float gload_f32(gptr_t base, unsigned byte_off) {
    return *reinterpret_cast<gptr_t>(reinterpret_cast<gbytes_t>(base) + byte_off);
}

// Source tensor of concatenated channel g (branch-free selects; g must be < Cin).
__device__ __forceinline__ gptr_t conv_src_ptr(const ConvArgs& a, int b, int g, size_t HW) {
    const float* sp = a.src[0];
    int cb = 0, cs = a.src_ch[0];
    const bool p1 = g >= a.src_end[0], p2 = g >= a.src_end[1], p3 = g >= a.src_end[2];
    sp = p1 ? a.src[1] : sp; cb = p1 ? a.src_end[0] : cb; cs = p1 ? a.src_ch[1] : cs;
    sp = p2 ? a.src[2] : sp; cb = p2 ? a.src_end[1] : cb; cs = p2 ? a.src_ch[2] : cs;
    sp = p3 ? a.src[3] : sp; cb = p3 ? a.src_end[2] : cb; cs = p3 ? a.src_ch[3] : cs;
    return (gptr_t)(sp + ((size_t)b * cs + (g - cb)) * HW);
}

// Epilogue of one 32(cout) x 32(pixel) accumulator tile: lane = pixel, the 16 registers walk output channels
// co0 + (reg&3) + 8*(reg>>2).  Bias, addends, activation / GRU gate arithmetic (update.py:81-85, 30-34, 62-66).
// Every load is unconditional on a clamped (always valid) index and the loads of the 16 registers are issued
// together; only the stores are predicated.  (Per-element "load or zero" branches made hipcc wait for each
// load in turn: 32 serial L2 round trips per lane.)
template <int EPI>
__device__ __forceinline__ void conv_epilogue_tile(const ConvArgs& a, int b, int co0, size_t pix, size_t HW, const f32x16& acc,
                                                   float scale) {
    int cc[16];
    bool ok[16];
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + (r & 3) + 8 * (r >> 2);
        ok[r] = co < a.Cout;
        cc[r] = min(co, a.Cout - 1);
        v[r] = acc[r] * scale;
    }
    if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += a.bias[cc[r]];
    }
    if (EPI == TCS_EPI_LINEAR) {
        const bool late = a.act == TCS_ACT_RELU_ADD_RELU;          // relu(relu(v) + addend)
        if (a.add1) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = a.add1[((size_t)b * a.Cout + cc[r]) * HW + pix];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (late ? fmaxf(v[r], 0.f) : v[r]) + t[r];
        }
        const int act = late ? TCS_ACT_RELU : a.act;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = apply_act(v[r], act) * a.post_scale;
        if (a.out) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (ok[r]) a.out[((size_t)b * a.out_ctot + a.out_coff + cc[r]) * HW + pix] = v[r];
        }
        if (a.out16) {
            // the 4 registers 4q..4q+3 are 4 consecutive channels of group (co0 >> 3) + q: one 8-byte store per {hi, lo}
            const int Hp = a.H + 2, Wp = a.W + 2, py = (int)(pix / a.W), px = (int)(pix - (size_t)py * a.W);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c_first = co0 + 8 * q;
                if (c_first < a.Cout) {
                    _Float16* o = a.out16 + s16_unit(b, a.out16_groups, a.out16_goff + (c_first >> 3), 0, Hp, Wp, py, px) + (co0 & 4);
                    s16_store4(o, (size_t)Hp * Wp * 8, &v[4 * q], a.Cout - c_first);
                }
            }
        }
    } else if (EPI == TCS_EPI_GRU_ZR) {
        // channel < hidden: z = sigmoid(. + cz) -> out ; else r = sigmoid(. + cr), out2 = r * h
        size_t o[16];
        bool isz[16];
        float ad[16], hh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            isz[r] = cc[r] < a.hidden;
            o[r] = ((size_t)b * a.hidden + (isz[r] ? cc[r] : cc[r] - a.hidden)) * HW + pix;
        }
        const bool any_add = a.add1 != nullptr || a.add2 != nullptr;
        if (any_add) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* p = isz[r] ? a.add1 : a.add2;
                ad[r] = p ? p[o[r]] : 0.f;
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) hh[r] = a.h[o[r]];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float g = sigmoidf_(v[r] + (any_add ? ad[r] : 0.f));
            if (ok[r]) {
                if (isz[r]) a.out[o[r]] = g;
                else a.out2[o[r]] = g * hh[r];
            }
        }
    } else if (EPI == TCS_EPI_DECONV2X) {
        // ConvTranspose2d(4, stride 2, pad 1) computed as a 3x3 convolution with 4*C output channels (one group per
        // output parity), pixel-shuffled on the way out: channel parity*C + c of pixel (i,j) -> out[c][2i+py][2j+px]
        const int C = a.hidden, Wo = 2 * a.W;
        const int i = (int)(pix / a.W), j = (int)(pix - (size_t)i * a.W);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int par = cc[r] / C, c = cc[r] - par * C;
            if (ok[r]) a.out[(((size_t)b * C + c) * (2 * a.H) + 2 * i + (par >> 1)) * Wo + 2 * j + (par & 1)] = v[r];
        }
    } else {
        size_t o[16];
        float ad[16], zz[16], hh[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] = ((size_t)b * a.hidden + cc[r]) * HW + pix;
        if (a.add1) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ad[r] = a.add1[o[r]];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { zz[r] = a.z[o[r]]; hh[r] = a.h[o[r]]; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float q = tanhf(v[r] + (a.add1 ? ad[r] : 0.f));
            if (ok[r]) a.out[o[r]] = a.keep_z ? zz[r] * hh[r] + (1.f - zz[r]) * q : (1.f - zz[r]) * hh[r] + zz[r] * q;
        }
    }
}

int tcs_conv_f16x3_launch(ConvArgs& a, int ksize, int epilogue, int stride, hipStream_t s);   // tcs_conv_f16.hip
