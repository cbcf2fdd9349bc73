// Temporal warp for TC-Stereo on gfx950: pose-based forward warp of the previous frame's disparity
// and features (softmax splatting), backward grid, hidden-state sampling.
// Replaces core/utils/geo_utils.py (warp, get_backward_grid) and the CuPy kernel softsplat_out
// (core/utils/splatting/softsplat.py:285-335) of the reference.
//
// Splat design: one thread per SOURCE pixel and channel group.  Lanes are horizontally adjacent
// source pixels, which for a rigid camera motion land on horizontally adjacent targets, so each
// atomic wave-instruction covers ~256 contiguous bytes of one NCHW channel plane — the shape the
// memory-side f32 atomics of MI355X run fastest on.  Geometry (flow, weights, exp(metric)) is
// computed once per thread, not once per (channel, pixel) as the reference kernel does.
#include "tcs_common.h"

struct Cam {
    float K[9], Ki[9], T[12], bf;
};

__device__ __forceinline__ Cam load_cam(const float* __restrict__ T_rel, const float* __restrict__ K,
                                        const float* __restrict__ K_inv, const float* __restrict__ baseline, int b) {
    Cam c;
#pragma unroll
    for (int i = 0; i < 9; ++i) { c.K[i] = K[b * 9 + i]; c.Ki[i] = K_inv[b * 9 + i]; }
#pragma unroll
    for (int i = 0; i < 12; ++i) c.T[i] = T_rel[b * 16 + i];
    c.bf = baseline[b] * c.K[0];
    return c;
}

__device__ __forceinline__ float fix_nonfinite(float v) { return isfinite(v) ? v : -1.0f; }

// disparity at (x,y) -> 3-D point -> rigid transform -> (new depth, projected pixel)
__device__ __forceinline__ void reproject(const Cam& c, float disp, float x, float y, float& zp, float& u, float& v) {
    const float depth = c.bf / fmaxf(disp, 0.001f);                       // disp2depth, geo_utils.py:16
    const float rx = c.Ki[0] * x + c.Ki[1] * y + c.Ki[2];                 // pixel2point, geo_utils.py:41
    const float ry = c.Ki[3] * x + c.Ki[4] * y + c.Ki[5];
    const float rz = c.Ki[6] * x + c.Ki[7] * y + c.Ki[8];
    const float px = depth * rx, py = depth * ry, pz = depth * rz;
    const float qx = c.T[0] * px + c.T[1] * py + c.T[2] * pz + c.T[3];    // relative_transform, geo_utils.py:144
    const float qy = c.T[4] * px + c.T[5] * py + c.T[6] * pz + c.T[7];
    const float qz = c.T[8] * px + c.T[9] * py + c.T[10] * pz + c.T[11];
    zp = qz;
    u = fix_nonfinite((c.K[0] * qx + c.K[1] * qy + c.K[2] * qz) / qz);    // point2pixel, geo_utils.py:55-56
    v = fix_nonfinite((c.K[3] * qx + c.K[4] * qy + c.K[5] * qz) / qz);
}

// ------------------------------------------------------------------------------------------------
// forward geometry: new disparity, validity, forward flow, per-block partial sums of the disparity
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_warp_geometry(const float* __restrict__ disp, const float* __restrict__ T_rel,
                                                       const float* __restrict__ K, const float* __restrict__ K_inv,
                                                       const float* __restrict__ baseline, int H, int W,
                                                       float* __restrict__ cur_disp, float* __restrict__ valid,
                                                       float* __restrict__ flow, float* __restrict__ partial) {
    __shared__ float red[4];
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    float cd = 0.f;
    if (p < HW) {
        const Cam c = load_cam(T_rel, K, K_inv, baseline, b);
        const int y = p / W, x = p - y * W;
        float zp, u, v;
        reproject(c, disp[(size_t)b * HW + p], (float)x, (float)y, zp, u, v);
        cd = fix_nonfinite(c.bf / zp);                                     // depth2disp, geo_utils.py:27-28
        cur_disp[(size_t)b * HW + p] = cd;
        valid[(size_t)b * HW + p] = (cd > 0.f && cd < (float)W) ? 1.f : 0.f;   // geo_utils.py:186
        flow[((size_t)b * 2 + 0) * HW + p] = u - (float)x;                 // geo_utils.py:192
        flow[((size_t)b * 2 + 1) * HW + p] = v - (float)y;
    }
    const float s = wave_sum(cd);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// fixed-order reduction of the partial sums -> mean over ALL pixels of the batch (geo_utils.py:193)
__global__ __launch_bounds__(256) void k_mean(const float* __restrict__ partial, int n, float inv_count, float* __restrict__ mean) {
    __shared__ float red[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *mean = ((red[0] + red[1]) + (red[2] + red[3])) * inv_count;
}

__global__ __launch_bounds__(256) void k_metric(const float* __restrict__ cur_disp, const float* __restrict__ mean, int n,
                                                float* __restrict__ metric) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) metric[i] = fminf(fmaxf(cur_disp[i] - *mean, -50.f), 50.f);
}

// ------------------------------------------------------------------------------------------------
// splat.  MODE 0: plain summation splat of `in` (softsplat_func.forward, softsplat.py:285-335).
//         MODE 1: warp(): channels [0,C) are prev_fmap, channel C the new disparity, channel C+1 the
//                 normaliser; every value is scaled by valid*exp(metric) (softsplat.py:236,250).
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_splat(const float* __restrict__ in, const float* __restrict__ flow,
                                               const float* __restrict__ cur_disp, const float* __restrict__ valid,
                                               const float* __restrict__ mean, int C, int H, int W, int ch_per_group,
                                               float* __restrict__ out) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const int y = p / W, x = p - y * W;
    const float fx = (float)x + flow[((size_t)b * 2 + 0) * HW + p];
    const float fy = (float)y + flow[((size_t)b * 2 + 1) * HW + p];
    if (!isfinite(fx) || !isfinite(fy)) return;                            // softsplat.py:301-302
    float scale = 1.f, cd = 0.f;
    if (MODE == 1) {
        if (valid[(size_t)b * HW + p] == 0.f) return;                      // contributes exact zeros
        cd = cur_disp[(size_t)b * HW + p];
        scale = expf(fminf(fmaxf(cd - *mean, -50.f), 50.f));               // geo_utils.py:193, softsplat.py:250
    }
    const float x0f = floorf(fx), y0f = floorf(fy);
    if (!(x0f >= -2.f && x0f <= (float)W && y0f >= -2.f && y0f <= (float)H)) return;   // every corner out of frame
    const int x0 = (int)x0f, y0 = (int)y0f;
    // corner weights exactly as softsplat.py:315-318
    const float wnw = ((float)(x0 + 1) - fx) * ((float)(y0 + 1) - fy);
    const float wne = (fx - (float)x0) * ((float)(y0 + 1) - fy);
    const float wsw = ((float)(x0 + 1) - fx) * (fy - (float)y0);
    const float wse = (fx - (float)x0) * (fy - (float)y0);
    const bool xl = x0 >= 0 && x0 < W, xr = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yt = y0 >= 0 && y0 < H, yb = y0 + 1 >= 0 && y0 + 1 < H;
    const int tnw = y0 * W + x0;
    const int Ctot = (MODE == 1) ? C + 2 : C;
    const int c_lo = blockIdx.z * ch_per_group, c_hi = min(Ctot, c_lo + ch_per_group);
    float* ob = out + (size_t)b * Ctot * HW;
    for (int c = c_lo; c < c_hi; ++c) {
        float v;
        if (MODE == 1) v = (c < C) ? in[((size_t)b * C + c) * HW + p] * scale : (c == C ? cd * scale : scale);
        else v = in[((size_t)b * C + c) * HW + p];
        float* o = ob + (size_t)c * HW + tnw;
        if (xl && yt) unsafeAtomicAdd(o, v * wnw);
        if (xr && yt) unsafeAtomicAdd(o + 1, v * wne);
        if (xl && yb) unsafeAtomicAdd(o + W, v * wsw);
        if (xr && yb) unsafeAtomicAdd(o + W + 1, v * wse);
    }
}

// explicit zero fill of the splat accumulator.  (A captured hipMemsetAsync node did not re-clear the
// 20 MB accumulator on HIP-graph replays on ROCm 7.2: the second replay of a frame summed into the previous
// frame's values.  A kernel node has no such ambiguity.)
__global__ __launch_bounds__(256) void k_zero_fill(float4* __restrict__ p, size_t n4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n4) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// normalise (softsplat.py:257-270, 'clipeps'), split, and the temporal cost (tc_stereo.py:139-140)
// One block = 64 consecutive pixels x 4 channel quarters (one wave each: coalesced rows of 64 floats per channel); the
// three sums of the cosine are combined across the quarters through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(256) void k_warp_finish(const float* __restrict__ acc, const float* __restrict__ cur_fmap,
                                                     int C, int HW, float* __restrict__ out_disp, float* __restrict__ out_fmap,
                                                     float* __restrict__ out_mask, float* __restrict__ out_cost) {
    __shared__ float s_part[3][4][64];
    const int b = blockIdx.y, lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p_raw = blockIdx.x * 64 + lane;
    const bool active = p_raw < HW;
    const int p = active ? p_raw : HW - 1;
    const float* a = acc + (size_t)b * (C + 2) * HW + p;
    const float norm = a[(size_t)(C + 1) * HW];
    const float mask = (norm != 0.f) ? 1.f : 0.f;
    const float den = fmaxf(norm, 1e-7f);
    if (q == 0 && active) {
        out_disp[(size_t)b * HW + p] = a[(size_t)C * HW] / den;
        out_mask[(size_t)b * HW + p] = mask;
    }
    const int cq = (C + 3) / 4, c_lo = q * cq, c_hi = min(C, c_lo + cq);
    float dot = 0.f, n1 = 0.f, nw = 0.f;
    for (int c = c_lo; c < c_hi; ++c) {
        const float fw = a[(size_t)c * HW] / den;
        if (out_fmap && active) out_fmap[((size_t)b * C + c) * HW + p] = fw;
        if (out_cost) {
            const float f1 = cur_fmap[((size_t)b * C + c) * HW + p];
            dot = fmaf(f1, fw, dot);
            n1 = fmaf(f1, f1, n1);
            nw = fmaf(fw, fw, nw);
        }
    }
    if (!out_cost) return;
    s_part[0][q][lane] = dot; s_part[1][q][lane] = n1; s_part[2][q][lane] = nw;
    __syncthreads();
    if (q == 0 && active) {
        float t[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) t[k] = ((s_part[k][0][lane] + s_part[k][1][lane]) + s_part[k][2][lane]) + s_part[k][3][lane];
        out_cost[(size_t)b * HW + p] = t[0] / (fmaxf(sqrtf(t[1]), 1e-12f) * fmaxf(sqrtf(t[2]), 1e-12f)) * mask;
    }
}

// ------------------------------------------------------------------------------------------------
// backward grid (geo_utils.py:201-236)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_backward_grid(const float* __restrict__ disp, const float* __restrict__ T_rel,
                                                       const float* __restrict__ K, const float* __restrict__ K_inv,
                                                       const float* __restrict__ baseline, int H, int W, float* __restrict__ grid) {
    const int b = blockIdx.y, HW = H * W;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HW) return;
    const Cam c = load_cam(T_rel, K, K_inv, baseline, b);
    const int y = p / W, x = p - y * W;
    float zp, u, v;
    reproject(c, fmaxf(disp[(size_t)b * HW + p], 0.01f), (float)x, (float)y, zp, u, v);
    const bool ok = zp > 0.f;
    grid[((size_t)b * 2 + 0) * HW + p] = ok ? u : -1.f;
    grid[((size_t)b * 2 + 1) * HW + p] = ok ? v : -1.f;
}

// bilinear_sampler (utils.py:82-97): zeros padding, align_corners=True, pixel coordinates
__global__ __launch_bounds__(256) void k_bilinear_sample(const float* __restrict__ img, const float* __restrict__ grid,
                                                         int C, int Hi, int Wi, int HWo, int ch_per_group,
                                                         float* __restrict__ out) {
    const int b = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= HWo) return;
    float gx = grid[((size_t)b * 2 + 0) * HWo + p], gy = grid[((size_t)b * 2 + 1) * HWo + p];
    const bool fin = isfinite(gx) && isfinite(gy);
    gx = fminf(fmaxf(fin ? gx : -8.f, -8.f), (float)Wi + 8.f);
    gy = fminf(fmaxf(fin ? gy : -8.f, -8.f), (float)Hi + 8.f);
    const float x0f = floorf(gx), y0f = floorf(gy);
    const float ax = gx - x0f, ay = gy - y0f;
    const int x0 = (int)x0f, y0 = (int)y0f;
    const bool xl = x0 >= 0 && x0 < Wi, xr = x0 + 1 >= 0 && x0 + 1 < Wi;
    const bool yt = y0 >= 0 && y0 < Hi, yb = y0 + 1 >= 0 && y0 + 1 < Hi;
    const int o00 = y0 * Wi + x0;
    const int c_lo = blockIdx.z * ch_per_group, c_hi = min(C, c_lo + ch_per_group);
    const size_t HWi = (size_t)Hi * Wi;
    for (int c = c_lo; c < c_hi; ++c) {
        const float* s = img + ((size_t)b * C + c) * HWi + o00;
        const float v00 = (xl && yt) ? s[0] : 0.f;
        const float v01 = (xr && yt) ? s[1] : 0.f;
        const float v10 = (xl && yb) ? s[Wi] : 0.f;
        const float v11 = (xr && yb) ? s[Wi + 1] : 0.f;
        const float top = (1.f - ax) * v00 + ax * v01;
        const float bot = (1.f - ax) * v10 + ax * v11;
        out[((size_t)b * C + c) * HWo + p] = (1.f - ay) * top + ay * bot;
    }
}

// bilinear resize, align_corners=True (F.interpolate), times `scale`
__global__ __launch_bounds__(256) void k_resize_bilinear(const float* __restrict__ x, int C, int H, int W, int Ho, int Wo,
                                                         float scale, float* __restrict__ out) {
    const int bc = blockIdx.y;
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= Ho * Wo) return;
    const int yo = p / Wo, xo = p - yo * Wo;
    const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = sy * (float)yo, fx = sx * (float)xo;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < H - 1), x1 = x0 + (x0 < W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0;
    const float* s = x + (size_t)bc * H * W;
    const float v = (1.f - ly) * ((1.f - lx) * s[y0 * W + x0] + lx * s[y0 * W + x1]) +
                    ly * ((1.f - lx) * s[y1 * W + x0] + lx * s[y1 * W + x1]);
    out[(size_t)bc * Ho * Wo + p] = scale * v;
}

// ------------------------------------------------------------------------------------------------
// camera algebra of tc_stereo.py:121-127,159 on the device (one thread per batch element): keeps the
// frame free of host round trips (torch.linalg.inv synchronises) so it can be captured in a HIP graph.
// ------------------------------------------------------------------------------------------------
template <int N>
__device__ void invert_small(const float* __restrict__ m, float* __restrict__ inv) {
    float a[N][2 * N];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) { a[i][j] = m[i * N + j]; a[i][N + j] = (i == j) ? 1.f : 0.f; }
    for (int c = 0; c < N; ++c) {                      // Gauss-Jordan, partial pivoting
        int p = c;
        for (int r = c + 1; r < N; ++r) if (fabsf(a[r][c]) > fabsf(a[p][c])) p = r;
        if (p != c) for (int j = 0; j < 2 * N; ++j) { const float t = a[c][j]; a[c][j] = a[p][j]; a[p][j] = t; }
        const float piv = 1.0f / a[c][c];
        for (int j = 0; j < 2 * N; ++j) a[c][j] *= piv;
        for (int r = 0; r < N; ++r) {
            if (r == c) continue;
            const float f = a[r][c];
            for (int j = 0; j < 2 * N; ++j) a[r][j] = fmaf(-f, a[c][j], a[r][j]);
        }
    }
    for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) inv[i * N + j] = a[i][N + j];
}

__device__ void matmul4(const float* a, const float* b, float* o) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            float s = 0.f;
            for (int k = 0; k < 4; ++k) s = fmaf(a[i * 4 + k], b[k * 4 + j], s);
            o[i * 4 + j] = s;
        }
}

__global__ void k_pose_prepare(const float* __restrict__ K, const float* __restrict__ T, const float* __restrict__ Tp, float scale,
                               int B, float* __restrict__ Ks, float* __restrict__ Ksi, float* __restrict__ Trel,
                               float* __restrict__ Tback) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    float ks[9], inv3[9];
    for (int i = 0; i < 9; ++i) ks[i] = K[b * 9 + i] * (i < 6 ? scale : 1.f);     // rows 0,1 scaled (tc_stereo.py:122)
    invert_small<3>(ks, inv3);
    for (int i = 0; i < 9; ++i) { Ks[b * 9 + i] = ks[i]; Ksi[b * 9 + i] = inv3[i]; }
    if (T && Tp) {
        float inv4[16], o[16];
        invert_small<4>(Tp + b * 16, inv4);
        matmul4(T + b * 16, inv4, o);                                             // geo_utils.py:148-155
        for (int i = 0; i < 16; ++i) Trel[b * 16 + i] = o[i];
        invert_small<4>(T + b * 16, inv4);
        matmul4(Tp + b * 16, inv4, o);                                            // tc_stereo.py:159
        for (int i = 0; i < 16; ++i) Tback[b * 16 + i] = o[i];
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
static size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

struct WarpWs {
    float *cur_disp, *valid, *flow, *partial, *mean, *acc;
    size_t acc_bytes;
};

static WarpWs carve(void* ws, int B, int C, int H, int W) {
    const size_t HW = (size_t)H * W;
    char* p = reinterpret_cast<char*>(ws);
    WarpWs w;
    w.cur_disp = reinterpret_cast<float*>(p); p += al256(B * HW * 4);
    w.valid = reinterpret_cast<float*>(p);    p += al256(B * HW * 4);
    w.flow = reinterpret_cast<float*>(p);     p += al256(2 * B * HW * 4);
    w.partial = reinterpret_cast<float*>(p);  p += al256((size_t)B * tcs_cdiv(HW, 256) * 4);
    w.mean = reinterpret_cast<float*>(p);     p += 256;
    w.acc = reinterpret_cast<float*>(p);
    w.acc_bytes = (size_t)B * (C + 2) * HW * 4;
    return w;
}

extern "C" {

size_t tcs_warp_workspace_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C < 0 || H <= 0 || W <= 0) return 0;
    const size_t HW = (size_t)H * W;
    return al256(B * HW * 4) * 2 + al256(2 * B * HW * 4) + al256((size_t)B * tcs_cdiv(HW, 256) * 4) + 256 +
           al256((size_t)B * (C + 2) * HW * 4);
}

static int geometry(const float* prev_disp, const float* T_rel, const float* K, const float* K_inv, const float* baseline,
                    int B, int H, int W, const WarpWs& w, hipStream_t s) {
    const int nb = tcs_cdiv((long long)H * W, 256);
    hipLaunchKernelGGL(k_warp_geometry, dim3(nb, B), dim3(256), 0, s, prev_disp, T_rel, K, K_inv, baseline, H, W,
                       w.cur_disp, w.valid, w.flow, w.partial);
    hipLaunchKernelGGL(k_mean, dim3(1), dim3(256), 0, s, w.partial, nb * B, 1.0f / (float)((long long)B * H * W), w.mean);
    return tcs_launch_status();
}

int tcs_warp_geometry(const float* prev_disp, const float* T_rel, const float* K, const float* K_inv,
                      const float* baseline, int B, int H, int W,
                      float* cur_disp, float* valid, float* flow, float* metric, void* workspace, tcs_stream_t stream) {
    if (!prev_disp || !T_rel || !K || !K_inv || !baseline || !cur_disp || !valid || !flow || !metric || !workspace)
        return TCS_EINVAL;
    if (B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipStream_t s = tcs_stream(stream);
    WarpWs w = carve(workspace, B, 0, H, W);
    w.cur_disp = cur_disp; w.valid = valid; w.flow = flow;
    int rc = geometry(prev_disp, T_rel, K, K_inv, baseline, B, H, W, w, s);
    if (rc) return rc;
    const int n = B * H * W;
    hipLaunchKernelGGL(k_metric, dim3(tcs_cdiv(n, 256)), dim3(256), 0, s, cur_disp, w.mean, n, metric);
    return tcs_launch_status();
}

int tcs_warp_forward(const float* prev_disp, const float* prev_fmap, const float* T_rel, const float* K,
                     const float* K_inv, const float* baseline, int B, int C, int H, int W,
                     float* out_disp, float* out_fmap, float* out_mask, const float* cur_fmap, float* out_cost,
                     void* workspace, tcs_stream_t stream) {
    if (!prev_disp || !prev_fmap || !T_rel || !K || !K_inv || !baseline || !out_disp || !out_mask || !workspace)
        return TCS_EINVAL;
    if ((cur_fmap == nullptr) != (out_cost == nullptr)) return TCS_EINVAL;
    if (B <= 0 || B > 65535 || C <= 0 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipStream_t s = tcs_stream(stream);
    WarpWs w = carve(workspace, B, C, H, W);
    int rc = geometry(prev_disp, T_rel, K, K_inv, baseline, B, H, W, w, s);
    if (rc) return rc;
    {
        const size_t n4 = (w.acc_bytes + 15) / 16;          // the carve is 256-byte aligned and padded
        hipLaunchKernelGGL(k_zero_fill, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, reinterpret_cast<float4*>(w.acc), n4);
    }
    const int nb = tcs_cdiv((long long)H * W, 256);
    const int cpg = 16, groups = tcs_cdiv(C + 2, cpg);
    hipLaunchKernelGGL(k_splat<1>, dim3(nb, B, groups), dim3(256), 0, s, prev_fmap, w.flow, w.cur_disp, w.valid, w.mean,
                       C, H, W, cpg, w.acc);
    hipLaunchKernelGGL(k_warp_finish, dim3(tcs_cdiv((long long)H * W, 64), B), dim3(256), 0, s, w.acc, cur_fmap, C, H * W, out_disp, out_fmap,
                       out_mask, out_cost);
    return tcs_launch_status();
}

int tcs_softsplat_sum(const float* in, const float* flow, int B, int C, int H, int W, float* out, tcs_stream_t stream) {
    if (!in || !flow || !out || B <= 0 || B > 65535 || C <= 0 || H <= 0 || W <= 0) return TCS_EINVAL;
    const int nb = tcs_cdiv((long long)H * W, 256);
    const int cpg = 16, groups = tcs_cdiv(C, cpg);
    hipLaunchKernelGGL(k_splat<0>, dim3(nb, B, groups), dim3(256), 0, tcs_stream(stream), in, flow,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, C, H, W, cpg, out);
    return tcs_launch_status();
}

int tcs_backward_grid(const float* disp, const float* T_rel, const float* K, const float* K_inv,
                      const float* baseline, int B, int H, int W, float* grid, tcs_stream_t stream) {
    if (!disp || !T_rel || !K || !K_inv || !baseline || !grid || B <= 0 || B > 65535 || H <= 0 || W <= 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_backward_grid, dim3(tcs_cdiv((long long)H * W, 256), B), dim3(256), 0, tcs_stream(stream),
                       disp, T_rel, K, K_inv, baseline, H, W, grid);
    return tcs_launch_status();
}

int tcs_bilinear_sample(const float* img, const float* grid, int B, int C, int Hi, int Wi, int Ho, int Wo,
                        float* out, tcs_stream_t stream) {
    if (!img || !grid || !out || B <= 0 || B > 65535 || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return TCS_EINVAL;
    const int cpg = 16;
    hipLaunchKernelGGL(k_bilinear_sample, dim3(tcs_cdiv((long long)Ho * Wo, 256), B, tcs_cdiv(C, cpg)), dim3(256), 0,
                       tcs_stream(stream), img, grid, C, Hi, Wi, Ho * Wo, cpg, out);
    return tcs_launch_status();
}

int tcs_grid_halve(const float* grid, int B, int H, int W, float* out, tcs_stream_t stream) {
    if (!grid || !out || B <= 0 || H < 2 || W < 2) return TCS_EINVAL;
    const int Ho = H / 2, Wo = W / 2;
    hipLaunchKernelGGL(k_resize_bilinear, dim3(tcs_cdiv((long long)Ho * Wo, 256), B * 2), dim3(256), 0, tcs_stream(stream),
                       grid, 2, H, W, Ho, Wo, 0.5f, out);
    return tcs_launch_status();
}

int tcs_pose_prepare(const float* K, const float* T, const float* T_prev, float scale, int B,
                     float* K_scaled, float* K_scaled_inv, float* T_rel, float* T_back, tcs_stream_t stream) {
    if (!K || !K_scaled || !K_scaled_inv || B <= 0) return TCS_EINVAL;
    if ((T == nullptr) != (T_prev == nullptr)) return TCS_EINVAL;
    if (T && (!T_rel || !T_back)) return TCS_EINVAL;
    hipLaunchKernelGGL(k_pose_prepare, dim3(tcs_cdiv(B, 64)), dim3(64), 0, tcs_stream(stream), K, T, T_prev, scale, B,
                       K_scaled, K_scaled_inv, T_rel, T_back);
    return tcs_launch_status();
}

int tcs_resize_bilinear(const float* x, int B, int C, int H, int W, int Ho, int Wo, float* out, tcs_stream_t stream) {
    if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0) return TCS_EINVAL;
    if ((long long)B * C > 65535) return TCS_EUNSUPPORTED;
    hipLaunchKernelGGL(k_resize_bilinear, dim3(tcs_cdiv((long long)Ho * Wo, 256), B * C), dim3(256), 0, tcs_stream(stream),
                       x, C, H, W, Ho, Wo, 1.0f, out);
    return tcs_launch_status();
}

}  // extern "C"
