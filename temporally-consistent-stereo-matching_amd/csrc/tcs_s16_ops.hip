// Memory-bound glue of the refinement loop on S16 ("pre-split") activations (include/tcs_mi355.h, tcs_s16.h):
// pool2x / interp of core/update.py:114-124, the InstanceNorm of the U-Net up-blocks (core/utils/basic_layers.py:28-35),
// the candidate stencil of DispRefine (core/update.py:259-289) and the blend kernel's hand-off of the next iteration's
// flow input (core/tc_stereo.py:180, core/update.py:126).  One thread per 16-byte unit (8 channels of one pixel): loads and
// stores are 16 B per lane, consecutive lanes = consecutive pixels; all arithmetic in fp32 on hi + lo.
// The zero border of an S16 tensor doubles as the zero padding of the 3x3 pooling window: no bounds tests.
#include "tcs_s16.h"

// update.py:114-115: avg_pool2d(3, stride 2, padding 1), divisor always 9.  Input taps 2yo-1 .. 2yo+1 are padded rows
// 2yo .. 2yo+2 <= H+1 of the source: always inside its zero border.
__global__ __launch_bounds__(256) void k_avgpool3s2_s16(const _Float16* __restrict__ x, int G, int H, int W, int Ho, int Wo,
                                                         _Float16* __restrict__ out, int Go) {
    const int bg = blockIdx.y, b = bg / G, g = bg - b * G;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int yo = p / Wo, xo = p - yo * Wo;
    const int Hp = H + 2, Wp = W + 2;
    const size_t plane = (size_t)Hp * Wp * 8;
    const _Float16* s = x + (((size_t)b * G + g) * 2) * plane;
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < 3; ++v) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            float t[8];
            s16_load8(s + ((size_t)(2 * yo + v) * Wp + (2 * xo + u)) * 8, plane, t);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] += t[j];
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = acc[j] / 9.f;
    s16_store8(out + s16_unit(b, Go, g, 0, Ho + 2, Wo + 2, yo, xo), (size_t)(Ho + 2) * (Wo + 2) * 8, acc);
}

// update.py:122-124: F.interpolate(bilinear, align_corners=True)
__global__ __launch_bounds__(256) void k_resize_bilinear_s16(const _Float16* __restrict__ x, int G, int H, int W, int Ho, int Wo,
                                                              _Float16* __restrict__ out, int Go) {
    const int bg = blockIdx.y, b = bg / G, g = bg - b * G;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int yo = p / Wo, xo = p - yo * Wo;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = sy * (float)yo, fx = sx * (float)xo;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const int Hp = H + 2, Wp = W + 2;
    const size_t plane = (size_t)Hp * Wp * 8;
    const _Float16* s = x + (((size_t)b * G + g) * 2) * plane;
    float a00[8], a01[8], a10[8], a11[8], r[8];
    s16_load8(s + ((size_t)(y0 + 1) * Wp + x0 + 1) * 8, plane, a00);
    s16_load8(s + ((size_t)(y0 + 1) * Wp + x1 + 1) * 8, plane, a01);
    s16_load8(s + ((size_t)(y1 + 1) * Wp + x0 + 1) * 8, plane, a10);
    s16_load8(s + ((size_t)(y1 + 1) * Wp + x1 + 1) * 8, plane, a11);
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (1.f - ly) * ((1.f - lx) * a00[j] + lx * a01[j]) + ly * ((1.f - lx) * a10[j] + lx * a11[j]);
    s16_store8(out + s16_unit(b, Go, g, 0, Ho + 2, Wo + 2, yo, xo), (size_t)(Ho + 2) * (Wo + 2) * 8, r);
}

// ---------------------------------------------------------------------------------------------------------------------
// InstanceNorm2d (affine=False, biased variance) + activation + optional S16 addend on an S16 tensor, two launches:
//   k_in_stats_s16: grid (slices, B*G).  A block reduces its slice of the plane for the 8 channels of its group with the
//       plane slice held in registers: sum -> slice mean -> sum of squares AROUND that mean (the reference's two-pass
//       variance, not E[x^2] - mean^2), and writes (mean_i, M2_i) per channel.
//   k_in_apply_s16: every block first merges the partials of its (b, group) with Chan's formula
//       (M2 = sum M2_i + n_i (mean_i - mean)^2), then normalises its own slice.
// A per-(b, channel) block like the fp32 kernel would leave a 64-channel tensor with 8 workgroups on 256 CUs.
// ---------------------------------------------------------------------------------------------------------------------
#define INS_T 256
#define INS_UPT 10                 // units per thread: a slice is up to 2560 pixels

__device__ __forceinline__ void block_sum8(float* v, float* red /* [INS_T/64][8] */) {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = wave_sum(v[j]);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) red[(threadIdx.x >> 6) * 8 + j] = v[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < INS_T / 64; ++w) s += red[w * 8 + j];
        v[j] = s;
    }
}

__global__ __launch_bounds__(INS_T) void k_in_stats_s16(const _Float16* __restrict__ x, int G, int H, int W, int slice_px,
                                                         float* __restrict__ partial /* [B*G][slices][16] */) {
    __shared__ float red[(INS_T / 64) * 8];
    const int bg = blockIdx.y, sl = blockIdx.x, nsl = gridDim.x;
    const int HW = H * W, Wp = W + 2;
    const size_t plane = (size_t)(H + 2) * Wp * 8;
    const _Float16* s = x + ((size_t)bg * 2) * plane;
    const int p_lo = sl * slice_px, p_hi = min(HW, p_lo + slice_px), n = p_hi - p_lo;
    float v[INS_UPT][8], sum[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < INS_UPT; ++k) {
        const int p = p_lo + threadIdx.x + INS_T * k;
        const bool ok = p < p_hi;
        const int pc = ok ? p : p_lo;
        const int y = pc / W, xx = pc - y * W;
        s16_load8(s + ((size_t)(y + 1) * Wp + xx + 1) * 8, plane, v[k]);
#pragma unroll
        for (int j = 0; j < 8; ++j) { v[k][j] = ok ? v[k][j] : 0.f; sum[j] += v[k][j]; }
    }
    block_sum8(sum, red);
    float mean[8], m2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 8; ++j) mean[j] = sum[j] / (float)n;
#pragma unroll
    for (int k = 0; k < INS_UPT; ++k) {
        const bool ok = p_lo + threadIdx.x + INS_T * k < p_hi;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = ok ? v[k][j] - mean[j] : 0.f; m2[j] = fmaf(d, d, m2[j]); }
    }
    block_sum8(m2, red);
    if (threadIdx.x < 8) {
        float* o = partial + ((size_t)bg * nsl + sl) * 16;
        o[threadIdx.x] = mean[threadIdx.x];
        o[8 + threadIdx.x] = m2[threadIdx.x];
    }
}

__device__ __forceinline__ float s16_act(float v, int act) {
    switch (act) {
        case TCS_ACT_RELU: return fmaxf(v, 0.f);
        case TCS_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        default: return v;
    }
}

__global__ __launch_bounds__(INS_T) void k_in_apply_s16(const _Float16* __restrict__ x, int G, int H, int W, int slice_px,
                                                         const float* __restrict__ partial, float eps, int act,
                                                         const _Float16* __restrict__ addend, int Ga, _Float16* __restrict__ out, int Go) {
    const int bg = blockIdx.y, sl = blockIdx.x, nsl = gridDim.x, b = bg / G, g = bg - b * G;
    const int HW = H * W, Wp = W + 2;
    const size_t plane = (size_t)(H + 2) * Wp * 8;
    // merge the slices' (n_i, mean_i, M2_i): every thread does it for all 8 channels (<= 8 slices x 16 floats, L2 hits)
    float mean[8], rstd[8];
    {
        float tot[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nsl; ++i) {
            const float ni = (float)(min(HW, (i + 1) * slice_px) - i * slice_px);
            const float* pi = partial + ((size_t)bg * nsl + i) * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) tot[j] += ni * pi[j];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) mean[j] = tot[j] / (float)HW;
        float m2[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < nsl; ++i) {
            const float ni = (float)(min(HW, (i + 1) * slice_px) - i * slice_px);
            const float* pi = partial + ((size_t)bg * nsl + i) * 16;
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float d = pi[j] - mean[j]; m2[j] += pi[8 + j] + ni * d * d; }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) rstd[j] = 1.0f / sqrtf(m2[j] / (float)HW + eps);
    }
    const _Float16* s = x + ((size_t)bg * 2) * plane;
    const _Float16* ad = addend ? addend + (((size_t)b * Ga + g) * 2) * plane : nullptr;
    _Float16* o = out + (((size_t)b * Go + g) * 2) * plane;
    const int p_lo = sl * slice_px, p_hi = min(HW, p_lo + slice_px);
    for (int p = p_lo + threadIdx.x; p < p_hi; p += INS_T) {
        const int y = p / W, xx = p - y * W;
        const size_t u = ((size_t)(y + 1) * Wp + xx + 1) * 8;
        float v[8], t[8];
        s16_load8(s + u, plane, v);
        if (ad) s16_load8(ad + u, plane, t);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s16_act((v[j] - mean[j]) * rstd[j], act) + (ad ? t[j] : 0.f);
        s16_store8(o + u, plane, v);
    }
}

// update.py:259-289 with the stem's input laid out for tcs_conv2d_s16: out16 = S16 [B][4 groups][...] holding the 27
// channels cat(candidates(9), |g_c - g_n| x (9), |g_c - g_n| y (9)) (+5 zero channels), cand9 = the 9 candidates as
// fp32 NCHW for the blend kernel.
__global__ __launch_bounds__(256) void k_propagate_s16(const float* __restrict__ grad, const float* __restrict__ disp, int H, int W,
                                                        float* __restrict__ cand9, _Float16* __restrict__ out16, int Go) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* d = disp + (size_t)b * HW;
    const float* gx = grad + (size_t)b * 2 * HW;
    const float* gy = gx + HW;
    const float gcx = gx[p], gcy = gy[p];
    float f[32];
#pragma unroll
    for (int k = 27; k < 32; ++k) f[k] = 0.f;
#pragma unroll
    for (int v = 0; v < 3; ++v) {
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int yy = y + v - 1, xx = x + u - 1;
            const bool in = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int q = min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1);
            const float dn = d[q];                              // replicate pad
            const float gnx = in ? gx[q] : 0.f;                 // zero pad
            const float gny = in ? gy[q] : 0.f;
            const int k = 3 * v + u;
            f[k] = dn + gnx * (float)(1 - u) + gny * (float)(1 - v);
            f[9 + k] = fabsf(gcx - gnx);
            f[18 + k] = fabsf(gcy - gny);
        }
    }
    if (cand9) {
#pragma unroll
        for (int k = 0; k < 9; ++k) cand9[((size_t)b * 9 + k) * HW + p] = f[k];
    }
    const size_t plane = (size_t)(H + 2) * (W + 2) * 8;
#pragma unroll
    for (int g = 0; g < 4; ++g) s16_store8(out16 + s16_unit(b, Go, g, 0, H + 2, W + 2, y, x), plane, f + 8 * g);
}

// one channel of an S16 tensor from a fp32 [B,1,H,W] tensor (the flow channel 127 of the motion features at frame start)
__global__ __launch_bounds__(256) void k_s16_set_channel(const float* __restrict__ x, int H, int W, _Float16* __restrict__ out, int Go, int ch) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, xx = p - y * W;
    half2_t hi, lo;
    s16_split2(x[(size_t)b * HW + p], 0.f, hi, lo);
    _Float16* o = out + s16_unit(b, Go, ch >> 3, 0, H + 2, W + 2, y, xx) + (ch & 7);
    o[0] = hi[0];
    o[(size_t)(H + 2) * (W + 2) * 8] = lo[0];
}

extern "C" {

int tcs_avgpool3s2_s16(const void* x, int B, int groups, int H, int W, void* out, int out_groups, tcs_stream_t stream) {
    if (!x || !out || B <= 0 || groups <= 0 || out_groups < groups || H <= 0 || W <= 0) return TCS_EINVAL;
    if ((long long)B * groups > 65535) return TCS_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(k_avgpool3s2_s16, dim3(tcs_cdiv((long long)Ho * Wo, 256), B * groups), dim3(256), 0, tcs_stream(stream),
                       reinterpret_cast<const _Float16*>(x), groups, H, W, Ho, Wo, reinterpret_cast<_Float16*>(out), out_groups);
    return tcs_launch_status();
}

int tcs_resize_bilinear_s16(const void* x, int B, int groups, int H, int W, int Ho, int Wo, void* out, int out_groups, tcs_stream_t stream) {
    if (!x || !out || B <= 0 || groups <= 0 || out_groups < groups || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return TCS_EINVAL;
    if ((long long)B * groups > 65535) return TCS_EUNSUPPORTED;
    hipLaunchKernelGGL(k_resize_bilinear_s16, dim3(tcs_cdiv((long long)Ho * Wo, 256), B * groups), dim3(256), 0, tcs_stream(stream),
                       reinterpret_cast<const _Float16*>(x), groups, H, W, Ho, Wo, reinterpret_cast<_Float16*>(out), out_groups);
    return tcs_launch_status();
}

size_t tcs_instance_norm_s16_workspace_bytes(int B, int groups, int H, int W) {
    if (B <= 0 || groups <= 0 || H <= 0 || W <= 0) return 0;
    const int nsl = tcs_cdiv((long long)H * W, INS_T * INS_UPT);
    return (size_t)B * groups * nsl * 16 * sizeof(float);
}

int tcs_instance_norm_s16(const void* x, int B, int groups, int H, int W, float eps, int act, const void* addend, int addend_groups,
                          void* out, int out_groups, void* workspace, tcs_stream_t stream) {
    if (!x || !out || !workspace || B <= 0 || groups <= 0 || out_groups < groups || H <= 0 || W <= 0 || eps < 0.f) return TCS_EINVAL;
    if (addend && addend_groups < groups) return TCS_EINVAL;
    if (act != TCS_ACT_NONE && act != TCS_ACT_RELU && act != TCS_ACT_LEAKY) return TCS_EUNSUPPORTED;
    if ((long long)B * groups > 65535) return TCS_EUNSUPPORTED;
    if (out_groups != groups || (addend && addend_groups != groups)) return TCS_EUNSUPPORTED;   // same-shape tensors only
    const int HW = H * W, nsl = tcs_cdiv(HW, INS_T * INS_UPT), slice = tcs_cdiv(HW, nsl);
    hipStream_t s = tcs_stream(stream);
    hipLaunchKernelGGL(k_in_stats_s16, dim3(nsl, B * groups), dim3(INS_T), 0, s, reinterpret_cast<const _Float16*>(x), groups, H, W, slice,
                       reinterpret_cast<float*>(workspace));
    hipLaunchKernelGGL(k_in_apply_s16, dim3(nsl, B * groups), dim3(INS_T), 0, s, reinterpret_cast<const _Float16*>(x), groups, H, W, slice,
                       reinterpret_cast<const float*>(workspace), eps, act, reinterpret_cast<const _Float16*>(addend), addend_groups,
                       reinterpret_cast<_Float16*>(out), out_groups);
    return tcs_launch_status();
}

int tcs_propagate_disparity_s16(const float* grad, const float* disp, int B, int H, int W, float* cand9, void* out16, int out_groups,
                                tcs_stream_t stream) {
    if (!grad || !disp || !out16 || out_groups < 4 || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_propagate_s16, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), grad, disp, H, W, cand9,
                       reinterpret_cast<_Float16*>(out16), out_groups);
    return tcs_launch_status();
}

int tcs_s16_set_channel(const float* x, int B, int H, int W, void* s16, int groups_total, int channel, tcs_stream_t stream) {
    if (!x || !s16 || B <= 0 || B > 65535 || H <= 0 || W <= 0 || channel < 0 || channel >= groups_total * 8) return TCS_EINVAL;
    hipLaunchKernelGGL(k_s16_set_channel, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), x, H, W,
                       reinterpret_cast<_Float16*>(s16), groups_total, channel);
    return tcs_launch_status();
}

}  // extern "C"
