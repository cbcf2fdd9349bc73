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
//   k_in_apply_s16: one thread per unit; every thread merges the partials of its (b, group) with Chan's pairwise formula
//       (M2 = M2_a + M2_b + d^2 n_a n_b / n), then normalises its unit.
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
        case TCS_ACT_RELU_ADD_RELU: return fmaxf(v, 0.f);          // the outer ReLU is applied after the addend (k_in_apply_s16)
        case TCS_ACT_LEAKY: return v > 0.f ? v : 0.01f * v;
        default: return v;
    }
}

__global__ __launch_bounds__(INS_T) void k_in_apply_s16(const _Float16* __restrict__ x, int G, int H, int W, int slice_px, int nsl,
                                                         const float* __restrict__ partial, float eps, int act,
                                                         const _Float16* __restrict__ addend, int Ga, _Float16* __restrict__ out, int Go) {
    // one thread per unit (a 64-channel 1/4-scale tensor is 600 workgroups; a loop of 10 units per thread over 64 workgroups
    // was 10 dependent memory round trips long: 25 us)
    const int bg = blockIdx.y, b = bg / G, g = bg - b * G;
    const int HW = H * W, Wp = W + 2;
    const size_t plane = (size_t)(H + 2) * Wp * 8;
    const int p = blockIdx.x * INS_T + threadIdx.x;
    const bool ok = p < HW;
    const int pc = ok ? p : HW - 1;
    const int y = pc / W, xx = pc - y * W;
    const size_t u = ((size_t)(y + 1) * Wp + xx + 1) * 8;
    float v[8], t[8];
    s16_load8(x + ((size_t)bg * 2) * plane + u, plane, v);
    if (addend) s16_load8(addend + (((size_t)b * Ga + g) * 2) * plane + u, plane, t);
    // merge the slices' (n_i, mean_i, M2_i) pairwise (Chan et al.): the loads do not depend on the running sums, so they
    // are all in flight together; every lane reads the same addresses (one L2 transaction per wave-load)
    float n = 0.f, mean[8], m2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { mean[j] = 0.f; m2[j] = 0.f; }
#pragma unroll 4
    for (int i = 0; i < nsl; ++i) {
        const float4* pi = reinterpret_cast<const float4*>(partial + ((size_t)bg * nsl + i) * 16);
        const float4 a0 = pi[0], a1 = pi[1], q0 = pi[2], q1 = pi[3];
        const float pm[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w}, p2[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        const float ni = (float)(min(HW, (i + 1) * slice_px) - i * slice_px), tot = n + ni, f = ni / tot;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float d = pm[j] - mean[j];
            mean[j] += d * f;
            m2[j] += p2[j] + d * d * n * f;
        }
        n = tot;
    }
    if (!ok) return;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float rstd = 1.0f / sqrtf(m2[j] / (float)HW + eps);
        v[j] = s16_act((v[j] - mean[j]) * rstd, act) + (addend ? t[j] : 0.f);
        if (act == TCS_ACT_RELU_ADD_RELU) v[j] = fmaxf(v[j], 0.f);       // relu(relu(norm(x)) + skip), extractor.py:44-58
    }
    s16_store8(out + (((size_t)b * Go + g) * 2) * plane + u, plane, v);
}

// the apply half for statistics that the producing transposed convolution accumulated (tcs_conv_s16.hip: s16_deconv_sums):
// sums = [B][C][2] 64-bit fixed point (sum x * 2^20, sum x^2 * 2^16) over the H*W pixels; one thread per unit, no merge loop
__global__ __launch_bounds__(INS_T) void k_in_apply_sums_s16(const _Float16* __restrict__ x, int Gt, int G, int H, int W, int C,
                                                              const long long* __restrict__ sums, float eps, int act,
                                                              const _Float16* __restrict__ addend, int Ga, _Float16* __restrict__ out, int Go) {
    const int b = blockIdx.y / G, g = blockIdx.y - b * G, bg = b * Gt + g;      // G = C/8 real groups of a tensor with Gt groups
    const int HW = H * W, Wp = W + 2;
    const size_t plane = (size_t)(H + 2) * Wp * 8;
    const int p = blockIdx.x * INS_T + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, xx = p - y * W;
    const size_t u = ((size_t)(y + 1) * Wp + xx + 1) * 8;
    float v[8], t[8];
    s16_load8(x + ((size_t)bg * 2) * plane + u, plane, v);
    if (addend) s16_load8(addend + (((size_t)b * Ga + g) * 2) * plane + u, plane, t);
    const long long* sp = sums + ((size_t)b * C + min(g * 8, C - 8)) * S16_IN_STRIDE;      // wave-uniform: 16 scalar-cache loads
    const double inv_n = 1.0 / (double)HW;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        // mean and E[x^2] from the integer sums in double (two conversions and three multiplications per channel: exact to 2^-53, so
        // that E[x^2] - mean^2 loses nothing the sums still had)
        const double mu = (double)sp[S16_IN_STRIDE * j] * (inv_n / 1048576.0), ex2 = (double)sp[S16_IN_STRIDE * j + 1] * (inv_n / 65536.0);
        const float mean = (float)mu, rstd = 1.0f / sqrtf(fmaxf((float)(ex2 - mu * mu), 0.f) + eps);
        const bool real = g * 8 + j < C;                   // padding channels of the last group stay zero
        v[j] = real ? s16_act((v[j] - mean) * rstd, act) + (addend ? t[j] : 0.f) : 0.f;
        if (act == TCS_ACT_RELU_ADD_RELU) v[j] = fmaxf(v[j], 0.f);
    }
    s16_store8(out + (((size_t)b * Go + g) * 2) * plane + u, plane, v);
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


// k_propagate_s16 with the refined gradient taken from tap partials (tcs_stencil.hip, "Tap partials"): residual_head[2] of the
// gradient predictor never runs as a launch.  grad[o] = (g5[o] + bias[o] + taps[o]) * post_scale (core/update.py:213) is assembled for a
// 16x16 pixel tile and its 1-pixel halo in LDS (clamped positions; the zero padding of update.py:277 is applied at use), written out
// as the fp32 gradient, and the candidate stencil of update.py:259-289 runs from LDS.
#define PT_T 16                     // tile: PT_T x PT_TY pixels; with the halo 18 x 10 = 180 elements, one per thread
#define PT_TY 8
#define PT_S (PT_T + 2)
template <int NT>
__device__ __forceinline__ float taps_at_s16ops(const float* __restrict__ tp, int ntile, int nplanes, int o9, int gy, int gx, int H, int W) {
    const size_t HW = (size_t)H * W;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int qy = gy + t / 3 - 1, qx = gx + t % 3 - 1;
        const bool in = qy >= 0 && qy < H && qx >= 0 && qx < W;
        const int q = min(max(qy, 0), H - 1) * W + min(max(qx, 0), W - 1);
        float s = 0.f;
        if (NT > 0) {                                      // all NT loads of a tap in flight together (tcs_stencil.hip: taps_at)
            float ld[NT > 0 ? NT : 1];
#pragma unroll
            for (int k = 0; k < NT; ++k) ld[k] = tp[((size_t)k * nplanes + o9 + t) * HW + q];
#pragma unroll
            for (int k = 0; k < NT; ++k) s += ld[k];
        } else {
            for (int k = 0; k < ntile; ++k) s += tp[((size_t)k * nplanes + o9 + t) * HW + q];
        }
        acc += in ? s : 0.f;
    }
    return acc;
}

template <int NT>
__global__ __launch_bounds__(256) void k_taps_propagate_s16(const float* __restrict__ taps, int ntile, const float* __restrict__ bias,
                                                             const float* __restrict__ g5, float post_scale, const float* __restrict__ disp,
                                                             int H, int W, float* __restrict__ grad_out, float* __restrict__ cand9,
                                                             _Float16* __restrict__ out16, int Go) {
    __shared__ float sg[2][PT_S * (PT_TY + 2)];
    const int b = blockIdx.z, HW = H * W;
    const int tx0 = blockIdx.x * PT_T, ty0 = blockIdx.y * PT_TY;
    const float* tp = taps + (size_t)b * ntile * 18 * HW;
    for (int e = threadIdx.x; e < PT_S * (PT_TY + 2); e += 256) {
        const int ey = e / PT_S, ex = e - ey * PT_S;
        const int gy = min(max(ty0 + ey - 1, 0), H - 1), gx = min(max(tx0 + ex - 1, 0), W - 1), q = gy * W + gx;
        const bool own = ey >= 1 && ey <= PT_TY && ex >= 1 && ex <= PT_T && ty0 + ey - 1 < H && tx0 + ex - 1 < W;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
            // same order of additions as k_taps_sum (tcs_stencil.hip): bit-equal to the materialised gradient
            const float v = ((taps_at_s16ops<NT>(tp, ntile, 18, o * 9, gy, gx, H, W) + (bias ? bias[o] : 0.f)) + g5[((size_t)b * 2 + o) * HW + q]) * post_scale;
            sg[o][e] = v;
            if (own && grad_out) grad_out[((size_t)b * 2 + o) * HW + q] = v;
        }
    }
    __syncthreads();
    const int ly = threadIdx.x / PT_T, lx = threadIdx.x - ly * PT_T;
    const int y = ty0 + ly, x = tx0 + lx;
    if (ly >= PT_TY || y >= H || x >= W) return;
    const int p = y * W + x;
    const float* d = disp + (size_t)b * HW;
    const float gcx = sg[0][(ly + 1) * PT_S + lx + 1], gcy = sg[1][(ly + 1) * PT_S + lx + 1];
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
            const float dn = d[q];                                                   // replicate pad
            const float gnx = in ? sg[0][(ly + v) * PT_S + lx + u] : 0.f;            // zero pad
            const float gny = in ? sg[1][(ly + v) * PT_S + lx + u] : 0.f;
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
    if (act != TCS_ACT_NONE && act != TCS_ACT_RELU && act != TCS_ACT_LEAKY && act != TCS_ACT_RELU_ADD_RELU) return TCS_EUNSUPPORTED;
    if (act == TCS_ACT_RELU_ADD_RELU && !addend) return TCS_EINVAL;
    if ((long long)B * groups > 65535) return TCS_EUNSUPPORTED;
    if (out_groups != groups || (addend && addend_groups != groups)) return TCS_EUNSUPPORTED;   // same-shape tensors only
    const int HW = H * W, nsl = tcs_cdiv(HW, INS_T * INS_UPT), slice = tcs_cdiv(HW, nsl);

    hipStream_t s = tcs_stream(stream);
    hipLaunchKernelGGL(k_in_stats_s16, dim3(nsl, B * groups), dim3(INS_T), 0, s, reinterpret_cast<const _Float16*>(x), groups, H, W, slice,
                       reinterpret_cast<float*>(workspace));
    hipLaunchKernelGGL(k_in_apply_s16, dim3(tcs_cdiv(HW, INS_T), B * groups), dim3(INS_T), 0, s, reinterpret_cast<const _Float16*>(x), groups, H, W,
                       slice, nsl, reinterpret_cast<const float*>(workspace), eps, act, reinterpret_cast<const _Float16*>(addend), addend_groups,
                       reinterpret_cast<_Float16*>(out), out_groups);
    return tcs_launch_status();
}

int tcs_instance_norm_apply_s16(const void* x, int B, int groups, int H, int W, int act, const void* addend, int addend_groups,
                                void* out, int out_groups, const void* in_stats, int C, float eps, tcs_stream_t stream) {
    if (!x || !out || !in_stats || B <= 0 || groups <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 32 != 0 || groups * 8 < C || !(eps >= 0.f)) return TCS_EINVAL;
    if ((long long)B * (C / 8) > 65535 || (long long)H * W > (1 << 20)) return TCS_EUNSUPPORTED;
    if (act != TCS_ACT_NONE && act != TCS_ACT_RELU && act != TCS_ACT_LEAKY && act != TCS_ACT_RELU_ADD_RELU) return TCS_EUNSUPPORTED;
    if (act == TCS_ACT_RELU_ADD_RELU && !addend) return TCS_EINVAL;
    if (out_groups != groups || (addend && addend_groups != groups)) return TCS_EUNSUPPORTED;
    const int G = C / 8;                                   // real groups (C % 32 == 0: no partial group)
    hipLaunchKernelGGL(k_in_apply_sums_s16, dim3(tcs_cdiv((long long)H * W, INS_T), B * G), dim3(INS_T), 0, tcs_stream(stream),
                       reinterpret_cast<const _Float16*>(x), groups, G, H, W, C, reinterpret_cast<const long long*>(in_stats), eps, act,
                       reinterpret_cast<const _Float16*>(addend), addend_groups, reinterpret_cast<_Float16*>(out), out_groups);
    return tcs_launch_status();
}

int tcs_propagate_disparity_s16(const float* grad, const float* disp, int B, int H, int W, float* cand9, void* out16, int out_groups,
                                tcs_stream_t stream) {
    if (!grad || !disp || !out16 || out_groups < 4 || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_propagate_s16, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), grad, disp, H, W, cand9,
                       reinterpret_cast<_Float16*>(out16), out_groups);
    return tcs_launch_status();
}

int tcs_taps_propagate_s16(const float* taps, int ntile, const float* bias2, const float* g5, float post_scale, const float* disp, int B, int H,
                           int W, float* grad_out, float* cand9, void* out16, int out_groups, tcs_stream_t stream) {
    if (!taps || !g5 || !disp || !out16 || ntile <= 0 || out_groups < 4 || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    auto kern = ntile == 4 ? k_taps_propagate_s16<4> : (ntile == 8 ? k_taps_propagate_s16<8> : k_taps_propagate_s16<0>);
    hipLaunchKernelGGL(kern, dim3(tcs_cdiv(W, PT_T), tcs_cdiv(H, PT_TY), B), dim3(256), 0, tcs_stream(stream), taps, ntile, bias2,
                       g5, post_scale, disp, H, W, grad_out, cand9, reinterpret_cast<_Float16*>(out16), out_groups);
    return tcs_launch_status();
}

int tcs_s16_set_channel(const float* x, int B, int H, int W, void* s16, int groups_total, int channel, tcs_stream_t stream) {
    if (!x || !s16 || B <= 0 || B > 65535 || H <= 0 || W <= 0 || channel < 0 || channel >= groups_total * 8) return TCS_EINVAL;
    hipLaunchKernelGGL(k_s16_set_channel, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), x, H, W,
                       reinterpret_cast<_Float16*>(s16), groups_total, channel);
    return tcs_launch_status();
}

}  // extern "C"

// =====================================================================================================================
// HiddenstateUpdater (core/update.py:57-68) as ONE launch.
//
//   x  = W2 . LeakyReLU(w1 * delta + b1) + b2                       1 -> 64 -> 64     (1x1)
//   z, r = sigmoid(Wzr . [h, x] + bzr)                              192 -> 256        (1x1)
//   q  = tanh(Wq . [r*h, x] + bq)                                   192 -> 128        (1x1)
//   h' = z*h + (1-z)*q                                              in place on the S16 hidden state
//
// Every layer is pixelwise, so a wave keeps its 32 pixels from the first layer to the last: an accumulator tile
// (32 channels x 32 pixels, pixel on the lane) becomes the B operand of the next layer's MFMAs after bias + activation +
// fp16 split, with no LDS round trip and no lane movement (registers 8s..8s+7 of a tile are k-step s; the weights of
// such K ranges are packed in the matching permuted channel order by tcs_pack_weight_frags).  h is fetched ONCE, in that same
// accumulator order (8-byte half-units of the S16 tensor), and serves as B operand of the gates, as the factor of r*h and in the
// final blend; z stays in registers.
//
// Weights (304 KB of A fragments per wave-pass) stream through LDS: the four waves of a workgroup share every fragment, and
// one k-step of a four-tile pass is 8 KiB = 8 LDS-DMA instructions (global_load_lds_dwordx4, two per wave).  The 40 stages of the
// four products (W2: 4 x 4 KiB; z, r, q: 12 x 8 KiB each) form ONE software pipeline over HU_D stage buffers, running across the products'
// boundaries, so the weights of the next product arrive during the activation arithmetic of the current one:
//     wait own pieces of stage g+1 (counted vmcnt) -> s_barrier -> refill the buffer of stage g-1 with stage g+HU_D-1 -> 2*NT
//     ds_read_b128 of stage g+1 (into the other register set) -> wait for stage g's fragments (counted lgkmcnt) -> 3*NT MFMAs of stage g.
// Round 2's version read the fragments per wave from L2 (310 KB per wave, 186 MB per launch through the L1s): 30 us alone,
// 41-47 us inside the loop; deeper register prefetch changed nothing (the L1 fill rate, not latency, was the bound).
// =====================================================================================================================
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct HuArgs {
    _Float16* h;                 // S16 [B][h_groups][2][H+2][W+2][8], 128 channels, updated in place
    int h_groups;
    const float* delta;          // [B,1,H,W]
    const float* w1; const float* b1;            // convs[0]: [64] each
    const uint4* W2; const float* b2;            // fragment-packed 64x64, natural K (layer 1 is produced in natural order)
    const uint4* Wzr; const float* bzr;          // 256 x 192, accumulator-order K
    const uint4* Wq; const float* bq;            // 128 x 192, accumulator-order K
    float us2, uszr, usq;                        // 2^-scale of the packed weights
    int B, H, W;
};

#define HU_MMA3(ACC, AHI, ALO, BHI, BLO)                                          \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(ALO, BHI, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AHI, BLO, ACC, 0, 0, 0);        \
    ACC = __builtin_amdgcn_mfma_f32_32x32x16_f16(AHI, BHI, ACC, 0, 0, 0);

// accumulator registers 8*sub .. 8*sub+7 of a tile (already bias-added / activated, fp32) -> B fragment (hi, lo) of a k-step
__device__ __forceinline__ void hu_frag_from_acc(const float* v8, half8& hi, half8& lo) { split8(v8, hi, lo); }

__device__ __forceinline__ const char* uniform_ptr_ops(const char* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

// sigmoid / tanh on the hardware exp2 and reciprocal with one correction step each: exp(x) = 2^(x*log2e) with the rounding
// error of the product folded back in (e * (1 + lo*ln2)), 1/d by v_rcp_f32 + one Newton step: <= 1-2 ulp, like expf and an IEEE
// division, at 9 instead of ~45 instructions per value.  The K loops of this kernel are 12 k-steps long, so the library
// versions (192 values per lane) would cost as much issue time as all of its MFMAs.
__device__ __forceinline__ float hu_exp(float x) {
    const float hi = x * 1.4426950408889634f;
    const float lo = fmaf(x, 1.4426950408889634f, -hi) + x * 1.9259629911266175e-8f;      // product rounding + log2(e) tail
    const float e = __builtin_amdgcn_exp2f(hi);
    return fmaf(e, lo * 0.6931471805599453f, e);
}
__device__ __forceinline__ float hu_rcp(float d) {
    const float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
// The argument is clamped to +-30 first: beyond |v| ~ 88.7 exp2 overflows to Inf, the correction term turns Inf into NaN (Inf - Inf) and
// v_rcp's Newton step does the same (-Inf * 0) — where torch.sigmoid (core/update.py:62-63) just saturates.  sigmoid(-30) = 9.4e-14:
// the clamp changes no result by more than that.  Found on BASELINE configs[1]: frame 9, iteration 30 of one free run, a 6-px jump
// of the refined disparity -> gate pre-activations beyond -88.7 -> NaN stored as 65504 into net08 (tools/flag_bisect.py).
__device__ __forceinline__ float hu_sigmoid(float v) { return hu_rcp(1.0f + hu_exp(-__builtin_amdgcn_fmed3f(v, -30.f, 30.f))); }
__device__ __forceinline__ float hu_tanh(float v) {
    const float e = hu_exp(2.0f * fminf(fmaxf(v, -15.f), 15.f));
    return 1.0f - 2.0f * hu_rcp(e + 1.0f);
}

#define HU_WAVES 4
#define HU_NBIAS (64 + 64 + 64 + 256 + 128)           // w1, b1, b2, bzr, bq
#define HU_D 6                                        // stage buffers: one being multiplied, one being fetched into registers, HU_D - 2 in flight
#define HU_STAGE_BYTES 8192
#define HU_RING_OFF 4096                              // byte offset of the stage ring behind the bias table
#define HU_NSTAGE 40                                  // W2: 0-3 (4 KiB), z: 4-15, r: 16-27, q: 28-39 (8 KiB each)
#define HU_PIECES(G) ((G) < 4 ? 1 : ((G) < HU_NSTAGE ? 2 : 0))      // LDS-DMA instructions per wave and stage

// LDS-DMA of one 1 KiB piece (as in tcs_conv_s16.hip): 64 lanes x 16 B from (wave-uniform base + per-lane byte offset) to LDS byte
// address DST (wave-uniform, in M0) + lane * 16
#define HU_DMA(VOFF, DST, BASE)                                                                                       \
    {                                                                                                                 \
        unsigned keep_;                                                                                               \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"  \
                     : "=&s"(keep_) : "v"(VOFF), "s"(DST), "s"(BASE) : "memory");                                    \
    }
// stage G (a compile-time constant after unrolling) -> its 4 or 8 KiB of packed fragments; wave w copies piece w (and w + 4)
#define HU_ISSUE(G)                                                                                                   \
    {                                                                                                                 \
        if ((G) < HU_NSTAGE) {                                                                                        \
            const char* src_ = (G) < 4 ? wb2 + (size_t)(G) * 4096                                                     \
                             : ((G) < 16 ? wbzr + (size_t)((G) - 4) * 16384                                           \
                             : ((G) < 28 ? wbzr + (size_t)((G) - 16) * 16384 + 8192 : wbq + (size_t)((G) - 28) * 8192)); \
            const char* base_ = uniform_ptr_ops(src_);                                                                \
            const unsigned dst_ = lds_ring + (unsigned)((G) % HU_D) * HU_STAGE_BYTES + (unsigned)wave * 1024u;        \
            asm volatile("s_nop 4" ::: "memory");                                                                     \
            HU_DMA(voff0, dst_, base_)                                                                                \
            if ((G) >= 4) HU_DMA(voff1, dst_ + 4096u, base_)                                                          \
        }                                                                                                             \
    }
#define HU_WAITV_N(N)                                                                                                 \
    {                                                                                                                 \
        if ((N) >= 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");                                                \
        else if ((N) == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");                                           \
        else if ((N) == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");                                           \
        else if ((N) == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");                                           \
        else if ((N) == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");                                           \
        else if ((N) == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");                                           \
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                         \
    }
#define HU_WAITL_N(N)                                                                                                 \
    {                                                                                                                 \
        if ((N) >= 8) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");                                              \
        else if ((N) == 4) asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");                                         \
        else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                                            \
    }
#define HU_DSREAD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF) : "memory")
// stage G becomes readable: its own DMA pieces have landed (stages G+1 .. G+HU_D-3, issued earlier, may stay in flight), everyone's have
// (barrier: also "everyone has finished READING stage G-2", whose buffer the next DMA overwrites), the pipeline is topped up, and the NT
// fragment pairs of stage G start moving into registers
#define HU_BEGIN(G, NT, AH, AL)                                                                                       \
    {                                                                                                                 \
        HU_WAITV_N(HU_PIECES((G) + 1) + HU_PIECES((G) + 2) + HU_PIECES((G) + 3))                                      \
        __builtin_amdgcn_s_barrier();                                                                                 \
        HU_ISSUE((G) + HU_D - 2)                                                                                      \
        const unsigned ra_ = lds_ring + (unsigned)((G) % HU_D) * HU_STAGE_BYTES + voff_lane;                          \
        _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                                              \
            HU_DSREAD(AH[t], ra_, (t * 2 + 0) * 1024);                                                                \
            HU_DSREAD(AL[t], ra_, (t * 2 + 1) * 1024);                                                                \
        }                                                                                                             \
    }
// one product: NK k-steps starting at global stage G0, NT output tiles; BFRAG(s, bh, bl) yields the B operand of k-step s.  The fragments
// of k-step s+1 are on their way from LDS while the 3*NT MFMAs of k-step s issue (two register sets; LDS returns data in order, so
// "all but the newest 2*NT reads" = this k-step's have arrived).
#define HU_GEMM(ACC, NT, NK, G0, BFRAG)                                                                               \
    {                                                                                                                 \
        half8 ah_[2][NT], al_[2][NT];                                                                                 \
        HU_BEGIN((G0), NT, ah_[0], al_[0])                                                                            \
        _Pragma("unroll") for (int s = 0; s < NK; ++s) {                                                              \
            if (s + 1 < NK) { HU_BEGIN((G0) + s + 1, NT, ah_[(s + 1) & 1], al_[(s + 1) & 1]) }                        \
            half8 bh_, bl_;                                                                                           \
            BFRAG(s, bh_, bl_)                                                                                        \
            HU_WAITL_N(s + 1 < NK ? 2 * NT : 0)                                                                       \
            _Pragma("unroll") for (int t = 0; t < NT; ++t) { HU_MMA3(ACC[t], ah_[s & 1][t], al_[s & 1][t], bh_, bl_) } \
        }                                                                                                             \
    }

__global__ __launch_bounds__(64 * HU_WAVES) void k_hidden_update_s16(HuArgs a) {
    extern __shared__ __attribute__((aligned(16))) float hu_lds[];            // [HU_NBIAS] biases | (at HU_RING_OFF) HU_D stage buffers
    const int lane = threadIdx.x & 63, l31 = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int npx = (a.W + 31) / 32;
    const int b = blockIdx.y;
    const int y = (blockIdx.x / npx) * HU_WAVES + wave, x = (blockIdx.x % npx) * 32 + l31;
    const bool active = y < a.H && x < a.W;
    const int yc = min(y, a.H - 1), xc = min(x, a.W - 1);                       // clamped: inactive lanes compute on a valid pixel
    const int Hp = a.H + 2, Wp = a.W + 2;
    const size_t plane = (size_t)Hp * Wp * 8;
    const unsigned lds_ring = __builtin_amdgcn_groupstaticsize() + HU_RING_OFF;
    const unsigned voff_lane = (unsigned)lane * 16u;
    const unsigned voff0 = (unsigned)wave * 1024u + voff_lane, voff1 = voff0 + 4096u;
    const char* wb2 = reinterpret_cast<const char*>(a.W2);
    const char* wbzr = reinterpret_cast<const char*>(a.Wzr);
    const char* wbq = reinterpret_cast<const char*>(a.Wq);

    // ---- the weight pipeline starts before anything else: stages 0 .. HU_D-3 (HU_BEGIN(g) adds stage g + HU_D - 2) ---------------------
#pragma unroll
    for (int g = 0; g < HU_D - 2; ++g) HU_ISSUE(g)

    // biases (and the single-channel first layer) through LDS: per-lane global loads of them inside the activation code
    // serialised on ~90 separate memory round trips
    for (int i = threadIdx.x; i < HU_NBIAS; i += 64 * HU_WAVES)
        hu_lds[i] = i < 64 ? a.w1[i] : (i < 128 ? a.b1[i - 64] : (i < 192 ? a.b2[i - 128] : (i < 448 ? a.bzr[i - 192] : a.bq[i - 448])));
    const float* s_w1 = hu_lds, *s_b1 = hu_lds + 64, *s_b2 = hu_lds + 128, *s_bzr = hu_lds + 192, *s_bq = hu_lds + 448;

    // h in accumulator order, as B fragments: k-step s (16 channels) = tile s>>1, registers 8(s&1) .. 8(s&1)+7 = groups
    // 4(s>>1) + 2(s&1) + {0, 1}, slot 4*hh — two 8-byte half-units per plane
    half8 hfh[8], hfl[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const _Float16* u = a.h + s16_unit(b, a.h_groups, 4 * (s >> 1) + 2 * (s & 1) + q, 0, Hp, Wp, yc, xc) + 4 * hh;
            const half4 hi = *reinterpret_cast<const half4*>(u);
            const half4 lo = *reinterpret_cast<const half4*>(u + plane);
#pragma unroll
            for (int j = 0; j < 4; ++j) { hfh[s][4 * q + j] = hi[j]; hfl[s][4 * q + j] = lo[j]; }
        }
    }
    // ---- layer 1 (VALU) straight into B-operand form: channel 16s + 8hh + j ------------------------------------------
    const float d = a.delta[((size_t)b * a.H + yc) * a.W + xc];
    __syncthreads();
    half8 x1h[4], x1l[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = 16 * s + 8 * hh + j;
            const float t = fmaf(s_w1[c], d, s_b1[c]);
            v[j] = t > 0.f ? t : 0.01f * t;
        }
        split8(v, x1h[s], x1l[s]);
    }
    // every compiler-visible global load above has been consumed (the compiler waited for it: its waits also cover the older DMA pieces);
    // from here to the final stores the only vector-memory traffic is the weight pipeline, waited for by count
    // ---- layer 2: x = W2 . x1 + b2 -> fragments of 4 k-steps --------------------------------------------------------------
    half8 xh[4], xl[4];
    {
        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#define HU_B_X1(S, BH, BL) { BH = x1h[(S) < 4 ? (S) : 0]; BL = x1l[(S) < 4 ? (S) : 0]; }
        HU_GEMM(acc, 2, 4, 0, HU_B_X1)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) v[i] = acc[t][i] * a.us2 + s_b2[32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh];
            hu_frag_from_acc(v, xh[2 * t], xl[2 * t]);
            hu_frag_from_acc(v + 8, xh[2 * t + 1], xl[2 * t + 1]);
        }
    }
#define HU_B_HX(S, BH, BL) { if ((S) < 8) { BH = hfh[(S) < 8 ? (S) : 0]; BL = hfl[(S) < 8 ? (S) : 0]; } else { BH = xh[(S) >= 8 ? (S) - 8 : 0]; BL = xl[(S) >= 8 ? (S) - 8 : 0]; } }
    // ---- z = sigmoid(Wzr[0:128] . [h, x] + bz): tiles 0-3 of the packed z|r matrix; stays in registers ------------------------
    f32x16 zz[4];
    {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        HU_GEMM(acc, 4, 12, 4, HU_B_HX)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) zz[t][i] = hu_sigmoid(acc[t][i] * a.uszr + s_bzr[32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh]);
    }
    // ---- r = sigmoid(Wzr[128:256] . [h, x] + br): tiles 4-7; r*h -> fragments of the q layer's first 8 k-steps -------------------
    half8 rhh[8], rhl[8];
    {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
        HU_GEMM(acc, 4, 12, 16, HU_B_HX)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pre = acc[t][i] * a.uszr + s_bzr[128 + 32 * t + (i & 3) + 8 * (i >> 2) + 4 * hh];
                const float hv = (float)hfh[2 * t + (i >> 3)][i & 7] + (float)hfl[2 * t + (i >> 3)][i & 7];
                v[i] = hv * hu_sigmoid(pre);
            }
            hu_frag_from_acc(v, rhh[2 * t], rhl[2 * t]);
            hu_frag_from_acc(v + 8, rhh[2 * t + 1], rhl[2 * t + 1]);
        }
    }
    // ---- q = tanh(Wq . [r*h, x] + bq); h' = z*h + (1-z)*q ----------------------------------------------------------------
    {
        f32x16 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
#define HU_B_RHX(S, BH, BL) { if ((S) < 8) { BH = rhh[(S) < 8 ? (S) : 0]; BL = rhl[(S) < 8 ? (S) : 0]; } else { BH = xh[(S) >= 8 ? (S) - 8 : 0]; BL = xl[(S) >= 8 ? (S) - 8 : 0]; } }
        HU_GEMM(acc, 4, 12, 28, HU_B_RHX)
        if (!active) return;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                _Float16* hp = a.h + s16_unit(b, a.h_groups, 4 * t + q, 0, Hp, Wp, y, x) + 4 * hh;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = 4 * q + j;
                    const float qv = hu_tanh(acc[t][i] * a.usq + s_bq[32 * t + 8 * q + 4 * hh + j]);
                    const float hv = (float)hfh[2 * t + (i >> 3)][i & 7] + (float)hfl[2 * t + (i >> 3)][i & 7];
                    v[j] = zz[t][i] * hv + (1.f - zz[t][i]) * qv;
                }
                s16_store4(hp, plane, v, 4);
            }
        }
    }
}

// [Cout][Cin] fp32 (1x1 weights) * 2^scale -> A fragments of v_mfma_f32_32x32x16_f16, (hi, lo) split:
// unit index ((kstep * ntiles + tile) * 2 + part) * 64 + lane, lane (m = l & 31, hh = l >> 5), element j = channel
//   kstep*16 + 8*hh + j                    for k-steps < nat_ksteps (B operand in S16 / natural order), else
//   kstep*16 + 8*(j>>2) + 4*hh + (j&3)     (B operand = accumulator registers of a preceding layer).
__global__ __launch_bounds__(256) void k_pack_weight_frags(const float* __restrict__ w, int Cout, int Cin, int nk, int ntiles, int nat_ksteps,
                                                            float scale, uint4* __restrict__ packed) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= nk * ntiles * 128) return;
    const int lane = u & 63, part = (u >> 6) & 1, tile = (u >> 7) % ntiles, ks = (u >> 7) / ntiles;
    const int m = lane & 31, hh = lane >> 5, co = tile * 32 + m;
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = ks * 16 + (ks < nat_ksteps ? 8 * hh + j : 8 * (j >> 2) + 4 * hh + (j & 3));
        float xw = (co < Cout && ci < Cin) ? w[(size_t)co * Cin + ci] * scale : 0.f;
        xw = __builtin_amdgcn_fmed3f(xw, -65504.f, 65504.f);
        const _Float16 hi = (_Float16)xw, lo = (_Float16)(xw - (float)hi);
        v[j] = part ? lo : hi;
    }
    packed[u] = *reinterpret_cast<uint4*>(&v);
}

int tcs_s16_flag_take_conv_s16(unsigned int*);
int tcs_s16_flag_take_s16_ops(unsigned int*);
int tcs_s16_flag_take_conv(unsigned int*);
int tcs_s16_flag_take_conv_f16(unsigned int*);
int tcs_s16_flag_take_stencil(unsigned int*);

extern "C" {

int tcs_s16_flags_detail(unsigned int* per_unit /* [TCS_S16_FLAG_UNITS] */) {
    if (!per_unit) return TCS_EINVAL;
    if (hipDeviceSynchronize() != hipSuccess) return TCS_ELAUNCH;
    for (int i = 0; i < TCS_S16_FLAG_UNITS; ++i) per_unit[i] = 0;
    int rc = tcs_s16_flag_take_conv_s16(&per_unit[0]);
    if (rc == TCS_OK) rc = tcs_s16_flag_take_s16_ops(&per_unit[1]);
    if (rc == TCS_OK) rc = tcs_s16_flag_take_conv(&per_unit[2]);
    if (rc == TCS_OK) rc = tcs_s16_flag_take_conv_f16(&per_unit[3]);
    if (rc == TCS_OK) rc = tcs_s16_flag_take_stencil(&per_unit[4]);
    return rc;
}

int tcs_s16_flags(unsigned int* flags_out) {
    if (!flags_out) return TCS_EINVAL;
    unsigned int u[TCS_S16_FLAG_UNITS];
    const int rc = tcs_s16_flags_detail(u);
    unsigned int v = 0;
    for (int i = 0; i < TCS_S16_FLAG_UNITS; ++i) v |= u[i];
    *flags_out = v;
    return rc;
}

size_t tcs_weight_frags_bytes(int Cout, int Cin) {
    if (Cout <= 0 || Cin <= 0) return 0;
    return (size_t)((Cin + 15) / 16) * ((Cout + 31) / 32) * 128 * 16;
}

int tcs_pack_weight_frags(const float* w_oi, int Cout, int Cin, int natural_channels, int scale_log2, void* packed, tcs_stream_t stream) {
    if (!w_oi || !packed || Cout <= 0 || Cin <= 0 || natural_channels < 0 || natural_channels % 16 != 0 || natural_channels > Cin ||
        scale_log2 < -60 || scale_log2 > 60) return TCS_EINVAL;
    const int nk = (Cin + 15) / 16, ntiles = (Cout + 31) / 32, n = nk * ntiles * 128;
    hipLaunchKernelGGL(k_pack_weight_frags, dim3((n + 255) / 256), dim3(256), 0, tcs_stream(stream), w_oi, Cout, Cin, nk, ntiles,
                       natural_channels / 16, ldexpf(1.0f, scale_log2), reinterpret_cast<uint4*>(packed));
    return tcs_launch_status();
}

int tcs_hidden_update_s16(void* h, int h_groups, const float* delta, const float* w1, const float* b1, const void* W2, const float* b2,
                          float unscale2, const void* Wzr, const float* bzr, float unscale_zr, const void* Wq, const float* bq,
                          float unscale_q, int B, int H, int W, tcs_stream_t stream) {
    if (!h || !delta || !w1 || !b1 || !W2 || !b2 || !Wzr || !bzr || !Wq || !bq || h_groups < 16 || B <= 0 || B > 65535 || H <= 0 || W <= 0)
        return TCS_EINVAL;
    HuArgs a;
    a.h = reinterpret_cast<_Float16*>(h); a.h_groups = h_groups; a.delta = delta; a.w1 = w1; a.b1 = b1;
    a.W2 = reinterpret_cast<const uint4*>(W2); a.b2 = b2; a.Wzr = reinterpret_cast<const uint4*>(Wzr); a.bzr = bzr;
    a.Wq = reinterpret_cast<const uint4*>(Wq); a.bq = bq; a.us2 = unscale2; a.uszr = unscale_zr; a.usq = unscale_q;
    a.B = B; a.H = H; a.W = W;
    const size_t lds = (size_t)HU_RING_OFF + (size_t)HU_D * HU_STAGE_BYTES;
    (void)hipGetLastError();
    hipLaunchKernelGGL(k_hidden_update_s16, dim3(tcs_cdiv(W, 32) * tcs_cdiv(H, HU_WAVES), B), dim3(64 * HU_WAVES), lds, tcs_stream(stream), a);
    return tcs_launch_status();
}

}  // extern "C"

// this translation unit's S16 domain flag (tcs_s16.h): read-and-clear for tcs_s16_flags()
int tcs_s16_flag_take_s16_ops(unsigned int* out) {
    unsigned int v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(TCS_S16_FLAG_VAR), sizeof(v)) != hipSuccess) return TCS_ELAUNCH;
    if (v && hipMemcpyToSymbol(HIP_SYMBOL(TCS_S16_FLAG_VAR), &zero, sizeof(zero)) != hipSuccess) return TCS_ELAUNCH;
    *out |= v;
    return TCS_OK;
}
