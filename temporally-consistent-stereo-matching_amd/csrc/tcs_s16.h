// S16 ("pre-split") activation tensors: element access shared by the kernels that produce or consume them.
// Layout (include/tcs_mi355.h): _Float16 [B][G][2 = hi|lo][H+2][W+2][8]; a 16-byte unit = 8 consecutive channels of a pixel.
#pragma once
#include "tcs_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// Fixed-point InstanceNorm accumulators of the fused transposed convolutions (tcs_conv_s16.hip: s16_deconv_sums, tcs_s16_ops.hip:
// k_in_apply_sums_s16): S16_IN_STRIDE 64-bit words per (b, channel), words 0 and 1 = sum x * 2^20, sum x^2 * 2^16.
#ifndef S16_IN_STRIDE
#define S16_IN_STRIDE 2
#endif

// Domain guard of the split: |x| <= 65504.  A value outside it is clamped (a NaN becomes 65504) and leaves a mark in a
// device-side flag word that the host can read once per frame (tcs_s16_flags): bit 0 = a finite value was saturated,
// bit 1 = a non-finite value was seen.  One flag word per translation unit that includes this header (no relocatable
// device code): tcs_s16_flags ORs them.  The test costs two compares per pair; the atomic runs only on a violation.
#ifndef TCS_S16_FLAG_VAR
#define TCS_S16_FLAG_VAR tcs_s16_flag_word
#endif
static __device__ unsigned int TCS_S16_FLAG_VAR;

__device__ __forceinline__ void s16_guard(float x0, float x1) {
    const bool bad = !(fabsf(x0) <= 65504.f) || !(fabsf(x1) <= 65504.f);
    if (__builtin_expect(bad, 0)) {
        const bool nonfinite = !(fabsf(x0) <= 3.402823466e38f) || !(fabsf(x1) <= 3.402823466e38f);
        atomicOr(&TCS_S16_FLAG_VAR, nonfinite ? 2u : 1u);
    }
}

// x = hi + lo (+ <= 2^-22 |x|); |x| saturates at 65504 (fp16 range); two values per v_cvt_pk_f16_f32
__device__ __forceinline__ void s16_split2(float x0, float x1, half2_t& hi, half2_t& lo) {
    s16_guard(x0, x1);
    float2_t x;
    x[0] = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f);
    x[1] = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
    hi = __builtin_convertvector(x, half2_t);
    const float2_t back = __builtin_convertvector(hi, float2_t);
    lo = __builtin_convertvector(x - back, half2_t);
}

__device__ __forceinline__ void split4(const float* v, half4& hi, half4& lo) {
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        half2_t h2, l2;
        s16_split2(v[j], v[j + 1], h2, l2);
        hi[j] = h2[0]; hi[j + 1] = h2[1]; lo[j] = l2[0]; lo[j + 1] = l2[1];
    }
}

__device__ __forceinline__ void split8(const float* v, half8& hi, half8& lo) {
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        half2_t h2, l2;
        s16_split2(v[j], v[j + 1], h2, l2);
        hi[j] = h2[0]; hi[j + 1] = h2[1]; lo[j] = l2[0]; lo[j + 1] = l2[1];
    }
}

// offset (in halves) of unit (b, g, hl, y, x) of an S16 tensor with G groups; (y, x) are interior coordinates, Hp = H + 2, Wp = W + 2
__device__ __forceinline__ size_t s16_unit(int b, int G, int g, int hl, int Hp, int Wp, int y, int x) {
    return ((((size_t)b * G + g) * 2 + hl) * Hp + (y + 1)) * (size_t)Wp * 8 + (size_t)(x + 1) * 8;
}

// the 8 channels of one unit as fp32 (hi + lo)
__device__ __forceinline__ void s16_load8(const _Float16* unit_hi, size_t plane_halves, float* v) {
    const half8 hi = *reinterpret_cast<const half8*>(unit_hi);
    const half8 lo = *reinterpret_cast<const half8*>(unit_hi + plane_halves);
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (float)hi[j] + (float)lo[j];
}

__device__ __forceinline__ void s16_store8(_Float16* unit_hi, size_t plane_halves, const float* v) {
    half8 hi, lo;
    split8(v, hi, lo);
    *reinterpret_cast<half8*>(unit_hi) = hi;
    *reinterpret_cast<half8*>(unit_hi + plane_halves) = lo;
}

// 4 consecutive channels (slot sub4 in {0,4} of a unit) of one pixel: the shape a 32x32 accumulator tile hands to a lane.
// `nvalid` < 4: only the first nvalid channels are stored (a convolution whose Cout is not a multiple of 4 must not
// touch the channels after its last one: they may belong to another producer, e.g. channel 127 of the motion features).
__device__ __forceinline__ void s16_store4(_Float16* o, size_t plane_halves, const float* v, int nvalid) {
    half4 hi, lo;
    split4(v, hi, lo);
    if (nvalid >= 4) {
        *reinterpret_cast<half4*>(o) = hi;
        *reinterpret_cast<half4*>(o + plane_halves) = lo;
    } else {
#pragma unroll
        for (int j = 0; j < 3; ++j)
            if (j < nvalid) { o[j] = hi[j]; o[plane_halves + j] = lo[j]; }
    }
}
