// Small memory-bound stencils of the TC-Stereo refinement loop on gfx950.
// Replaces the grouped-conv "one-hot kernel" formulations of the reference
// (core/utils/geo_utils.py:73-132, core/update.py:259-300, core/tc_stereo.py:75-88) and the
// avg_pool2d / interpolate glue of core/update.py:114-124 with direct stencil kernels.
// One thread per output pixel; lanes run along x so every load/store is coalesced.
#include "tcs_s16.h"

// tc_stereo.py:188-189
__global__ __launch_bounds__(256) void k_flow_step(float* __restrict__ coords1, const float* __restrict__ delta, int W, int n,
                                                   float* __restrict__ disp_q) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float c = coords1[i] + delta[i];
    coords1[i] = c;
    disp_q[i] = (float)(i % W) - c;
}

// geo_utils.py:115-132: replicate pad, forward differences
__global__ __launch_bounds__(256) void k_grad_xy(const float* __restrict__ disp, int H, int W, float scale, float* __restrict__ grad) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* d = disp + (size_t)b * HW;
    const float c = d[p];
    const float r = d[y * W + min(x + 1, W - 1)];
    const float dn = d[min(y + 1, H - 1) * W + x];
    grad[((size_t)b * 2 + 0) * HW + p] = scale * (r - c);
    grad[((size_t)b * 2 + 1) * HW + p] = scale * (dn - c);
}

// geo_utils.py:73-101 (level=2): 16 neighbour vectors (dilation 1 then 2, clockwise from top-left),
// zero-padded disparity, cross product of vector k with vector (k+2) mod 16
__global__ __launch_bounds__(256) void k_grad_candidates(const float* __restrict__ disp, int H, int W, float* __restrict__ out) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* d = disp + (size_t)b * HW;
    const float c = d[p];
    const int dv[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
    const int du[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
    float vx[16], vy[16], vz[16];
#pragma unroll
    for (int s = 1; s <= 2; ++s) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int yy = y + s * dv[k], xx = x + s * du[k];
            const float nb = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? d[yy * W + xx] : 0.f;
            const int i = (s - 1) * 8 + k;
            vx[i] = (float)(s * du[k]);
            vy[i] = (float)(s * dv[k]);
            vz[i] = nb - c;
        }
    }
    float* o = out + (size_t)b * 32 * HW + p;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = (k + 2) & 15;
        const float nx = vy[k] * vz[r] - vz[k] * vy[r];
        const float ny = vz[k] * vx[r] - vx[k] * vz[r];
        const float nz = vx[k] * vy[r] - vy[k] * vx[r];
        o[(size_t)k * HW] = -nx / nz;
        o[(size_t)(16 + k) * HW] = -ny / nz;
    }
}

// k_flow_step + k_grad_xy + k_grad_candidates in one launch: the three run back to back on the same 19,200-pixel field once
// per iteration and each is a 4.6-us dispatch.  A neighbour's disparity is recomputed from coords1 + delta (same
// arithmetic as k_flow_step, so the three outputs are bit-identical to the separate kernels'); coords1 is NOT updated —
// the caller replaces it with the blend kernel's output (tc_stereo.py:188-189, 212-213).
__global__ __launch_bounds__(256) void k_flow_step_grads(const float* __restrict__ coords1, const float* __restrict__ delta, int H, int W,
                                                         float scale, float* __restrict__ disp_q, float* __restrict__ grad,
                                                         float* __restrict__ cands) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* c1 = coords1 + (size_t)b * HW;
    const float* dl = delta + (size_t)b * HW;
#define DISP_AT(YY, XX) ((float)(XX) - (c1[(YY) * W + (XX)] + dl[(YY) * W + (XX)]))
    const float c = DISP_AT(y, x);
    disp_q[(size_t)b * HW + p] = c;
    {
        const int xr = min(x + 1, W - 1), yd = min(y + 1, H - 1);
        const float r = DISP_AT(y, xr), dn = DISP_AT(yd, x);
        grad[((size_t)b * 2 + 0) * HW + p] = scale * (r - c);
        grad[((size_t)b * 2 + 1) * HW + p] = scale * (dn - c);
    }
    const int dv[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
    const int du[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
    float vx[16], vy[16], vz[16];
#pragma unroll
    for (int s = 1; s <= 2; ++s) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int yy = y + s * dv[k], xx = x + s * du[k];
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
            const float nb_raw = DISP_AT(yc, xc);
            const float nb = ok ? nb_raw : 0.f;
            const int i = (s - 1) * 8 + k;
            vx[i] = (float)(s * du[k]);
            vy[i] = (float)(s * dv[k]);
            vz[i] = nb - c;
        }
    }
#undef DISP_AT
    float* o = cands + (size_t)b * 32 * HW + p;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = (k + 2) & 15;
        const float nx = vy[k] * vz[r] - vz[k] * vy[r];
        const float ny = vz[k] * vx[r] - vx[k] * vz[r];
        const float nz = vx[k] * vy[r] - vy[k] * vx[r];
        o[(size_t)k * HW] = -nx / nz;
        o[(size_t)(16 + k) * HW] = -ny / nz;
    }
}

// update.py:259-289
__global__ __launch_bounds__(256) void k_propagate(const float* __restrict__ grad, const float* __restrict__ disp, int H, int W,
                                                   float* __restrict__ out) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* d = disp + (size_t)b * HW;
    const float* gx = grad + (size_t)b * 2 * HW;
    const float* gy = gx + HW;
    const float gcx = gx[p], gcy = gy[p];
    float* o = out + (size_t)b * 27 * HW + p;
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
            o[(size_t)k * HW] = dn + gnx * (float)(1 - u) + gny * (float)(1 - v);
            o[(size_t)(9 + k) * HW] = fabsf(gcx - gnx);
            o[(size_t)(18 + k) * HW] = fabsf(gcy - gny);
        }
    }
}

// update.py:298-300 and tc_stereo.py:198-202
__global__ __launch_bounds__(256) void k_softmax_blend(const float* __restrict__ logits, const float* __restrict__ cand, int cand_ctot,
                                                       const float* __restrict__ disp_q, int W, int HW, float* __restrict__ refined,
                                                       float* __restrict__ delta, float* __restrict__ coords1,
                                                       float* __restrict__ flow_x, float* __restrict__ flow_x_ch, long long flow_x_ch_bstride,
                                                       _Float16* __restrict__ fx16, int fx16_groups, int fx16_ch, int H) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const float* l = logits + (size_t)b * 9 * HW + p;
    const float* c = cand + (size_t)b * cand_ctot * HW + p;
    float w[9], m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { w[k] = l[(size_t)k * HW]; m = fmaxf(m, w[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { w[k] = expf(w[k] - m); s += w[k]; }
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) r += (w[k] / s) * c[(size_t)k * HW];
    const size_t o = (size_t)b * HW + p;
    refined[o] = r;
    if (delta) delta[o] = r - disp_q[o];
    const float xf = (float)(p % W);
    const float c1 = xf - r;
    if (coords1) coords1[o] = c1;
    // next iteration's motion-encoder input, coords1 - coords0 (tc_stereo.py:180): contiguous copy for the 7x7 stem and
    // a copy straight into channel 127 of the motion feature buffer (update.py:126)
    if (flow_x) flow_x[o] = c1 - xf;
    if (flow_x_ch) flow_x_ch[(size_t)b * flow_x_ch_bstride + p] = c1 - xf;
    if (fx16) {          // ... or into channel fx16_ch of the S16 motion feature tensor (hi and lo halves, 2 bytes each)
        half2_t hi, lo;
        s16_split2(c1 - xf, 0.f, hi, lo);
        _Float16* o16 = fx16 + s16_unit(b, fx16_groups, fx16_ch >> 3, 0, H + 2, W + 2, p / W, p % W) + (fx16_ch & 7);
        o16[0] = hi[0];
        o16[(size_t)(H + 2) * (W + 2) * 8] = lo[0];
    }
}

// tc_stereo.py:75-88 (factor 4) on flow = -disp, clipped like the returned dict (tc_stereo.py:223-224)
__global__ __launch_bounds__(256) void k_convex_upsample(const float* __restrict__ disp, const float* __restrict__ mask, int H, int W,
                                                         int clip, float* __restrict__ up, float* __restrict__ flow_q) {
    const int b = blockIdx.y, HW = H * W;
    const int Wu = 4 * W, Hu = 4 * H;
    const int pu = blockIdx.x * 256 + threadIdx.x;
    if (pu >= Hu * Wu) return;
    const int yu = pu / Wu, xu = pu - yu * Wu;
    const int y = yu >> 2, i = yu & 3, x = xu >> 2, j = xu & 3;
    const float* d = disp + (size_t)b * HW;
    const float* mk = mask + ((size_t)b * 144 + i * 4 + j) * HW + y * W + x;
    float w[9], m = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) { w[k] = mk[(size_t)k * 16 * HW]; m = fmaxf(m, w[k]); }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) { w[k] = expf(w[k] - m); s += w[k]; }
    float r = 0.f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        const float f = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 4.f * (-d[yy * W + xx]) : 0.f;
        r += (w[k] / s) * f;
    }
    up[(size_t)b * Hu * Wu + pu] = clip ? fminf(r, 0.f) : r;
    if (flow_q && i == 0 && j == 0) {
        const float fq = -d[y * W + x];
        flow_q[(size_t)b * HW + y * W + x] = clip ? fminf(fq, 0.f) : fq;
    }
}

// update.py:114-115: 3x3 stride-2 pad-1 average, divisor always 9
__global__ __launch_bounds__(256) void k_avgpool3s2(const float* __restrict__ x, int H, int W, int Ho, int Wo, float* __restrict__ out) {
    const int bc = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int yo = p / Wo, xo = p - yo * Wo;
    const float* s = x + (size_t)bc * H * W;
    float acc = 0.f;
#pragma unroll
    for (int v = -1; v <= 1; ++v) {
#pragma unroll
        for (int u = -1; u <= 1; ++u) {
            const int yy = 2 * yo + v, xx = 2 * xo + u;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) acc += s[yy * W + xx];
        }
    }
    out[(size_t)bc * Ho * Wo + p] = acc / 9.f;
}


// =====================================================================================================================
// "Tap partials": a 3x3 convolution with ONE or TWO output channels that follows a wide convolution (FlowHead.conv2, 256 -> 1,
// core/update.py:13-17; DispGradPredictor.residual_head[2], 128 -> 2, core/update.py:196,213) never runs as a launch of its own.
//   out[o][p] = b[o] + sum_t sum_c w[o][c][t] * y[c][p + d(t)],   d(t) = (t/3 - 1, t%3 - 1), y = 0 outside the image
// The producer of y (tcs_conv2d_s16, tap_* fields) holds 32 channels of y per workgroup in registers and leaves
//   P[b][tile][o*9 + t][q] = sum_{c in tile} w[o][c][t] * y[c][q]
// per source pixel q — 9 (18) fp32 planes per 32-channel tile instead of the 256 (128) channels of y.  The consumer sums the planes
// in a fixed order (tile-major inside a tap: results do not depend on scheduling) at q = p + d(t) inside the image.  Three consumers:
// the plain sum (API paths), the flow-step / gradient stencils of an iteration, and DispRefine's candidate stencil.
// =====================================================================================================================
// NT = number of tiles when known at compile time (the 9 * NT loads are then all in flight together; with a run-time loop every load
// waited for the previous one: 72 serial round trips per element), 0 = run-time `ntile`
template <int NT>
__device__ __forceinline__ float taps_at(const float* __restrict__ tp /* [ntile][nout*9][HW] of this batch element */, int ntile, int nplanes,
                                         int o9, int gy, int gx, int H, int W) {
    const size_t HW = (size_t)H * W;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int qy = gy + t / 3 - 1, qx = gx + t % 3 - 1;
        const bool in = qy >= 0 && qy < H && qx >= 0 && qx < W;
        const int q = min(max(qy, 0), H - 1) * W + min(max(qx, 0), W - 1);
        float s = 0.f;
        if (NT > 0) {
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

// out[b][o][p] = (addend[b][o][p] + bias[o] + taps) * scale
template <int NT>
__global__ __launch_bounds__(256) void k_taps_sum(const float* __restrict__ taps, int ntile, int nout, const float* __restrict__ bias,
                                                  const float* __restrict__ addend, float scale, int H, int W, float* __restrict__ out) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float* tp = taps + (size_t)b * ntile * nout * 9 * HW;
    for (int o = 0; o < nout; ++o) {
        float v = taps_at<NT>(tp, ntile, nout * 9, o * 9, y, x, H, W) + (bias ? bias[o] : 0.f);
        if (addend) v += addend[((size_t)b * nout + o) * HW + p];
        out[((size_t)b * nout + o) * HW + p] = v * scale;
    }
}

// k_flow_step_grads with delta = FlowHead.conv2's output taken from tap partials (+ bias): the disparity of a 16x16 pixel tile and its
// 2-pixel halo is assembled once in LDS (positions outside the image hold the value of the clamped position: the replicate padding
// of geo_utils.py:117; the zero padding of geo_utils.py:91 is applied where the candidates are formed), then the three stencils
// run from LDS.  Also emits delta itself (nullable) for callers that want FlowHead's output as a tensor.
#define FT_T 16                     // tile: FT_T x FT_TY pixels; with the halo (FT_T + 4) x (FT_TY + 4) = 240 elements, one per thread
#define FT_TY 8
#define FT_S (FT_T + 4)
template <int NT>
__global__ __launch_bounds__(256) void k_flow_taps_step_grads(const float* __restrict__ coords1, const float* __restrict__ taps, int ntile,
                                                              const float* __restrict__ bias, int H, int W, float scale,
                                                              float* __restrict__ disp_q, float* __restrict__ grad, float* __restrict__ cands,
                                                              float* __restrict__ delta_out) {
    __shared__ float sd[FT_S * (FT_TY + 4)];
    const int b = blockIdx.z, HW = H * W;
    const int tx0 = blockIdx.x * FT_T, ty0 = blockIdx.y * FT_TY;
    const float* c1 = coords1 + (size_t)b * HW;
    const float* tp = taps + (size_t)b * ntile * 9 * HW;
    const float bs = bias ? bias[0] : 0.f;
    for (int e = threadIdx.x; e < FT_S * (FT_TY + 4); e += 256) {
        const int ey = e / FT_S, ex = e - ey * FT_S;
        const int gy = min(max(ty0 + ey - 2, 0), H - 1), gx = min(max(tx0 + ex - 2, 0), W - 1);
        const float dl = taps_at<NT>(tp, ntile, 9, 0, gy, gx, H, W) + bs;
        sd[e] = (float)gx - (c1[gy * W + gx] + dl);
        if (delta_out && ey >= 2 && ey < FT_TY + 2 && ex >= 2 && ex < FT_T + 2 && ty0 + ey - 2 < H && tx0 + ex - 2 < W)
            delta_out[(size_t)b * HW + gy * W + gx] = dl;
    }
    __syncthreads();
    const int ly = threadIdx.x / FT_T, lx = threadIdx.x - ly * FT_T;
    const int y = ty0 + ly, x = tx0 + lx;
    if (ly >= FT_TY || y >= H || x >= W) return;
    const int p = y * W + x;
#define FT_D(DY, DX) sd[(ly + 2 + (DY)) * FT_S + (lx + 2 + (DX))]
    const float c = FT_D(0, 0);
    disp_q[(size_t)b * HW + p] = c;
    grad[((size_t)b * 2 + 0) * HW + p] = scale * (FT_D(0, 1) - c);
    grad[((size_t)b * 2 + 1) * HW + p] = scale * (FT_D(1, 0) - c);
    const int dv[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
    const int du[8] = {-1, 0, 1, 1, 1, 0, -1, -1};
    float vx[16], vy[16], vz[16];
#pragma unroll
    for (int s = 1; s <= 2; ++s) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int yy = y + s * dv[k], xx = x + s * du[k];
            const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
            const float nb = ok ? FT_D(s * dv[k], s * du[k]) : 0.f;
            const int i = (s - 1) * 8 + k;
            vx[i] = (float)(s * du[k]);
            vy[i] = (float)(s * dv[k]);
            vz[i] = nb - c;
        }
    }
#undef FT_D
    float* o = cands + (size_t)b * 32 * HW + p;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int r = (k + 2) & 15;
        const float nx = vy[k] * vz[r] - vz[k] * vy[r];
        const float ny = vz[k] * vx[r] - vx[k] * vz[r];
        const float nz = vx[k] * vy[r] - vy[k] * vx[r];
        o[(size_t)k * HW] = -nx / nz;
        o[(size_t)(16 + k) * HW] = -ny / nz;
    }
}

// [nout][C][3][3] fp32 * scale -> A fragments of v_mfma_f32_32x32x16_f16, (hi, lo) split, for the product
// P[o*9 + t][pixel] = sum_c w[o][c][t] * y[c][pixel] taken 32 channels (two k-steps) at a time from a producer's accumulator registers:
// unit ((kstep * 2 + part) * 64 + lane), lane (m = l & 31: row o*9 + t, rows >= 9*nout zero; hh = l >> 5), element j = channel
// kstep*16 + 8*(j>>2) + 4*hh + (j&3) — the order in which accumulator registers 8s .. 8s+7 of a 32-channel tile hold channels (the
// accumulator-order k-steps of k_pack_weight_frags, tcs_s16_ops.hip).
typedef _Float16 tap_half8 __attribute__((ext_vector_type(8)));
__global__ __launch_bounds__(256) void k_pack_tap_weights(const float* __restrict__ w, int nout, int C, int nk, float scale, uint4* __restrict__ packed) {
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= nk * 128) return;
    const int lane = u & 63, part = (u >> 6) & 1, ks = u >> 7;
    const int m = lane & 31, hh = lane >> 5, o = m / 9, t = m - o * 9;
    tap_half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = ks * 16 + 8 * (j >> 2) + 4 * hh + (j & 3);
        float xw = (m < 9 * nout && c < C) ? w[((size_t)o * C + c) * 9 + t] * scale : 0.f;
        xw = __builtin_amdgcn_fmed3f(xw, -65504.f, 65504.f);
        const _Float16 hi = (_Float16)xw, lo = (_Float16)(xw - (float)hi);
        v[j] = part ? lo : hi;
    }
    packed[u] = *reinterpret_cast<uint4*>(&v);
}

extern "C" {

size_t tcs_tap_weights_floats(int nout, int C) {
    if (nout < 1 || nout > 2 || C <= 0) return 0;
    return (size_t)((C + 31) / 32) * 2 * 128 * 4;          // two 16-channel k-steps per tile x (hi, lo) x 64 lanes x 16 bytes
}

int tcs_pack_tap_weights(const float* w_oihw, int nout, int C, int scale_log2, float* packed, tcs_stream_t stream) {
    if (!w_oihw || !packed || nout < 1 || nout > 2 || C <= 0 || scale_log2 < -60 || scale_log2 > 60) return TCS_EINVAL;
    const int nk = ((C + 31) / 32) * 2, n = nk * 128;
    hipLaunchKernelGGL(k_pack_tap_weights, dim3((n + 255) / 256), dim3(256), 0, tcs_stream(stream), w_oihw, nout, C, nk, ldexpf(1.0f, scale_log2),
                       reinterpret_cast<uint4*>(packed));
    return tcs_launch_status();
}

int tcs_taps_sum(const float* taps, int ntile, int nout, const float* bias, const float* addend, float scale, int B, int H, int W, float* out,
                 tcs_stream_t stream) {
    if (!taps || !out || ntile <= 0 || nout < 1 || nout > 2 || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    auto kern = ntile == 8 ? k_taps_sum<8> : (ntile == 4 ? k_taps_sum<4> : k_taps_sum<0>);
    hipLaunchKernelGGL(kern, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), taps, ntile, nout, bias, addend,
                       scale, H, W, out);
    return tcs_launch_status();
}

int tcs_flow_taps_step_grads(const float* coords1, const float* taps, int ntile, const float* bias, int B, int H, int W, float scale,
                             float* disp_q, float* grad, float* cands, float* delta_out, tcs_stream_t stream) {
    if (!coords1 || !taps || !disp_q || !grad || !cands || ntile <= 0 || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    auto kern = ntile == 8 ? k_flow_taps_step_grads<8> : (ntile == 4 ? k_flow_taps_step_grads<4> : k_flow_taps_step_grads<0>);
    hipLaunchKernelGGL(kern, dim3(tcs_cdiv(W, FT_T), tcs_cdiv(H, FT_TY), B), dim3(256), 0, tcs_stream(stream), coords1, taps,
                       ntile, bias, H, W, scale, disp_q, grad, cands, delta_out);
    return tcs_launch_status();
}


int tcs_flow_step(float* coords1, const float* delta, int B, int H, int W, float* disp_q, tcs_stream_t stream) {
    if (!coords1 || !delta || !disp_q || B <= 0 || H <= 0 || W <= 0) return TCS_EINVAL;
    const int n = B * H * W;
    hipLaunchKernelGGL(k_flow_step, dim3(tcs_cdiv(n, 256)), dim3(256), 0, tcs_stream(stream), coords1, delta, W, n, disp_q);
    return tcs_launch_status();
}

int tcs_flow_step_grads(const float* coords1, const float* delta, int B, int H, int W, float scale, float* disp_q, float* grad,
                        float* cands, tcs_stream_t stream) {
    if (!coords1 || !delta || !disp_q || !grad || !cands || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_flow_step_grads, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), coords1, delta, H, W,
                       scale, disp_q, grad, cands);
    return tcs_launch_status();
}

int tcs_disp_gradient_xy(const float* disp, int B, int H, int W, float scale, float* grad, tcs_stream_t stream) {
    if (!disp || !grad || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_grad_xy, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), disp, H, W, scale, grad);
    return tcs_launch_status();
}

int tcs_grad_candidates(const float* disp, int B, int H, int W, float* cands, tcs_stream_t stream) {
    if (!disp || !cands || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_grad_candidates, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), disp, H, W, cands);
    return tcs_launch_status();
}

int tcs_propagate_disparity(const float* grad, const float* disp, int B, int H, int W, float* out27, tcs_stream_t stream) {
    if (!grad || !disp || !out27 || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_propagate, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream), grad, disp, H, W, out27);
    return tcs_launch_status();
}

int tcs_softmax_blend(const float* logits9, const float* cand, int cand_ctot, const float* disp_q,
                      int B, int H, int W, float* refined, float* delta_disp, float* coords1, float* flow_x, float* flow_x_ch,
                      long long flow_x_ch_bstride, tcs_stream_t stream) {
    if (!logits9 || !cand || !refined || cand_ctot < 9 || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    if (delta_disp && !disp_q) return TCS_EINVAL;
    if (flow_x_ch && flow_x_ch_bstride < (long long)H * W) return TCS_EINVAL;
    hipLaunchKernelGGL(k_softmax_blend, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream),
                       logits9, cand, cand_ctot, disp_q, W, H * W, refined, delta_disp, coords1, flow_x, flow_x_ch, flow_x_ch_bstride,
                       (_Float16*)nullptr, 0, 0, H);
    return tcs_launch_status();
}

int tcs_softmax_blend_s16(const float* logits9, const float* cand, int cand_ctot, const float* disp_q,
                          int B, int H, int W, float* refined, float* delta_disp, float* coords1, float* flow_x,
                          void* flow_x_s16, int flow_x_s16_groups, int flow_x_s16_channel, tcs_stream_t stream) {
    if (!logits9 || !cand || !refined || cand_ctot < 9 || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    if (delta_disp && !disp_q) return TCS_EINVAL;
    if (flow_x_s16 && (flow_x_s16_channel < 0 || flow_x_s16_channel >= 8 * flow_x_s16_groups)) return TCS_EINVAL;
    hipLaunchKernelGGL(k_softmax_blend, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream),
                       logits9, cand, cand_ctot, disp_q, W, H * W, refined, delta_disp, coords1, flow_x, (float*)nullptr, 0LL,
                       reinterpret_cast<_Float16*>(flow_x_s16), flow_x_s16_groups, flow_x_s16_channel, H);
    return tcs_launch_status();
}

int tcs_convex_upsample(const float* disp, const float* mask, int B, int H, int W, int clip, float* flow_up, float* flow_q,
                        tcs_stream_t stream) {
    if (!disp || !mask || !flow_up || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_convex_upsample, dim3(tcs_cdiv((long long)16 * H * W, 256), B), dim3(256), 0, tcs_stream(stream),
                       disp, mask, H, W, clip, flow_up, flow_q);
    return tcs_launch_status();
}

int tcs_avgpool3s2(const float* x, int B, int C, int H, int W, float* out, tcs_stream_t stream) {
    if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0) return TCS_EINVAL;
    if ((long long)B * C > 65535) return TCS_EUNSUPPORTED;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    hipLaunchKernelGGL(k_avgpool3s2, dim3(tcs_cdiv((long long)Ho * Wo, 256), B * C), dim3(256), 0, tcs_stream(stream), x, H, W, Ho, Wo, out);
    return tcs_launch_status();
}

int tcs_abi_version(void) { return 7; }

const char* tcs_error_string(int code) {
    switch (code) {
        case TCS_OK: return "ok";
        case TCS_EINVAL: return "invalid argument";
        case TCS_ELAUNCH: return "HIP launch failed";
        case TCS_EUNSUPPORTED: return "unsupported shape";
        default: return "unknown error";
    }
}

}  // extern "C"

// this translation unit's S16 domain flag (tcs_s16.h): read-and-clear for tcs_s16_flags()
int tcs_s16_flag_take_stencil(unsigned int* out) {
    unsigned int v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(TCS_S16_FLAG_VAR), sizeof(v)) != hipSuccess) return TCS_ELAUNCH;
    if (v && hipMemcpyToSymbol(HIP_SYMBOL(TCS_S16_FLAG_VAR), &zero, sizeof(zero)) != hipSuccess) return TCS_ELAUNCH;
    *out |= v;
    return TCS_OK;
}
