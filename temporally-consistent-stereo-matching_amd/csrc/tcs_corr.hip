// Correlation volume for TC-Stereo on gfx950: build (fp32 MFMA), skewed pyramid, first-frame
// argmax, and the per-iteration lookup.  Replaces core/corr.py (CorrBlock1D) of the reference.
//
// Data layout in HBM (see include/tcs_mi355.h): level i is stored "skewed",
//   P_i[b][h][d][w1] = L_i[b][h][w1][j],  d = ((w1 >> i) - j) mod W_i,  W_i = W >> i,
// so that the lookup's taps for horizontally adjacent pixels are contiguous along w1.
#include "tcs_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ------------------------------------------------------------------------------------------------
// 1. inverse L2 norms over channels (F.normalize, corr.py:58-59; eps 1e-12)
// ------------------------------------------------------------------------------------------------
// One block = 64 consecutive pixels x 4 channel quarters (a wave each); the quarters' sums are added in a fixed order.
__global__ __launch_bounds__(256) void k_inv_norm(const float* __restrict__ f1, const float* __restrict__ f2,
                                                  int C, int HW, float* __restrict__ rn) {
    __shared__ float s_part[4][64];
    const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int p_raw = blockIdx.x * 64 + lane;
    const int b = blockIdx.y, m = blockIdx.z;
    const int p = min(p_raw, HW - 1);
    const float* f = (m == 0 ? f1 : f2) + (size_t)b * C * HW + p;
    const int cq = (C + 3) / 4, c_hi = min(C, (q + 1) * cq);
    float s = 0.f;
    for (int c = q * cq; c < c_hi; ++c) {
        const float v = f[(size_t)c * HW];
        s = fmaf(v, v, s);
    }
    s_part[q][lane] = s;
    __syncthreads();
    if (q == 0 && p_raw < HW) {
        const float t = ((s_part[0][lane] + s_part[1][lane]) + s_part[2][lane]) + s_part[3][lane];
        rn[((size_t)m * gridDim.y + b) * HW + p] = 1.0f / fmaxf(sqrtf(t), 1e-12f);
    }
}

// ------------------------------------------------------------------------------------------------
// 2. V[b,h,w1,w2] = sum_c f1[b,c,h,w1] f2[b,c,h,w2] * rn1 * rn2   (einsum of corr.py:60)
//    One block = 64(w1) x 64(w2) of one image row, 4 waves as 2x2 tiles of 32x32,
//    v_mfma_f32_32x32x2_f32 (exact fp32).  Operands are read NCHW: w contiguous -> coalesced.
// ------------------------------------------------------------------------------------------------
#define CG_KC 32
__global__ __launch_bounds__(256) void k_corr_gemm(const float* __restrict__ f1, const float* __restrict__ f2,
                                                   const float* __restrict__ rn, int B, int C, int H, int W,
                                                   float* __restrict__ vol) {
    __shared__ float sa[CG_KC][64];
    __shared__ float sb[CG_KC][64];
    const int w2_0 = blockIdx.x * 64, w1_0 = blockIdx.y * 64;
    const int bh = blockIdx.z, b = bh / H, h = bh % H;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, l31 = lane & 31, half = lane >> 5;
    const size_t HW = (size_t)H * W;
    const float* p1 = f1 + (size_t)b * C * HW + (size_t)h * W;
    const float* p2 = f2 + (size_t)b * C * HW + (size_t)h * W;

    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;

    const int col = tid & 63, krow = tid >> 6;
    for (int c0 = 0; c0 < C; c0 += CG_KC) {
        __syncthreads();
#pragma unroll
        for (int r = 0; r < CG_KC / 4; ++r) {
            const int k = krow + 4 * r, c = c0 + k;
            const bool okc = c < C;
            sa[k][col] = (okc && w1_0 + col < W) ? p1[(size_t)c * HW + w1_0 + col] : 0.f;
            sb[k][col] = (okc && w2_0 + col < W) ? p2[(size_t)c * HW + w2_0 + col] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < CG_KC; kk += 2) {
            const float a = sa[kk + half][wr * 32 + l31];
            const float bb = sb[kk + half][wc * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb, acc, 0, 0, 0);
        }
    }
    const int w2 = w2_0 + wc * 32 + l31;
    if (w2 >= W) return;
    const float r2 = rn[((size_t)B + b) * HW + (size_t)h * W + w2];
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
        const int w1 = w1_0 + wr * 32 + (reg & 3) + 8 * (reg >> 2) + 4 * half;
        if (w1 < W) {
            const float r1 = rn[(size_t)b * HW + (size_t)h * W + w1];
            vol[((size_t)bh * W + w1) * W + w2] = acc[reg] * r1 * r2;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 3. finalize: natural level 0 -> pooled levels, skewed stores, masked cost volume, argmax.
//    One block = 32 rows (w1) of one (b,h); the full w2 extent of those rows lives in LDS.
// ------------------------------------------------------------------------------------------------
struct FinalizeArgs {
    const float* vol;          // natural level 0 [B,H,W,W]
    float* pyr[4];             // skewed levels
    float* nat[4];             // natural levels 1..3 (index 0 unused), nullable
    float* cost;               // [B,W,H,W] nullable
    float* sdisp; float* scost; float* smask;   // nullable trio
    int B, H, W;
};

__device__ __forceinline__ int pos_mod(int a, int m) {
    int r = a % m;
    return r < 0 ? r + m : r;
}

__global__ __launch_bounds__(256) void k_corr_finalize(FinalizeArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int W = a.W, H = a.H;
    int Wl[4], Wp[4];
    float* L[4];
    {
        float* p = lds;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            Wl[i] = W >> i;
            Wp[i] = Wl[i] | 1;             // odd row pitch: column reads are bank-conflict free
            L[i] = p;
            p += 32 * Wp[i];
        }
    }
    const int w1_0 = blockIdx.x * 32;
    const int bh = blockIdx.y, b = bh / H, h = bh % H;
    const int tid = threadIdx.x;
    const int rows = min(32, W - w1_0);

    // level 0 rows -> LDS (coalesced along w2)
    for (int idx = tid; idx < 32 * W; idx += 256) {
        const int r = idx / W, j = idx - r * W;
        L[0][r * Wp[0] + j] = (r < rows) ? a.vol[((size_t)bh * W + w1_0 + r) * W + j] : 0.f;
    }
    __syncthreads();
    // pooled levels: avg_pool2d([1,2]) chained (corr.py:21-23); trailing odd element dropped
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        for (int idx = tid; idx < 32 * Wl[i]; idx += 256) {
            const int r = idx / Wl[i], j = idx - r * Wl[i];
            L[i][r * Wp[i] + j] = 0.5f * (L[i - 1][r * Wp[i - 1] + 2 * j] + L[i - 1][r * Wp[i - 1] + 2 * j + 1]);
        }
        __syncthreads();
    }
    // skewed stores: for a fixed d the 32 rows are 32 consecutive floats along w1
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float* dst = a.pyr[i] + (size_t)bh * Wl[i] * W;
        for (int idx = tid; idx < 32 * Wl[i]; idx += 256) {
            const int d = idx >> 5, r = idx & 31;
            if (r < rows) {
                const int w1 = w1_0 + r;
                const int j = pos_mod((w1 >> i) - d, Wl[i]);
                dst[(size_t)d * W + w1] = L[i][r * Wp[i] + j];
            }
        }
        if (i > 0 && a.nat[i]) {
            float* nd = a.nat[i] + ((size_t)bh * W + w1_0) * Wl[i];
            for (int idx = tid; idx < rows * Wl[i]; idx += 256) {
                const int r = idx / Wl[i], j = idx - r * Wl[i];
                nd[idx] = L[i][r * Wp[i] + j];
            }
        }
    }
    // masked cost volume [b][w2][h][w1] (corr.py:25-31)
    if (a.cost) {
        for (int idx = tid; idx < 32 * W; idx += 256) {
            const int w2 = idx >> 5, r = idx & 31;
            if (r < rows) {
                const int w1 = w1_0 + r;
                a.cost[(((size_t)b * W + w2) * H + h) * W + w1] = (w2 <= w1) ? L[0][r * Wp[0] + w2] : 0.f;
            }
        }
    }
    // argmax_disp (corr.py:67-79): winner, runner-up with the winner's +-1 neighbourhood zeroed
    if (a.sdisp) {
        const int lane = tid & 63, wave = tid >> 6;
        for (int r = wave * 8; r < wave * 8 + 8; ++r) {
            if (r >= rows) break;                       // wave-uniform
            const int w1 = w1_0 + r;
            const float* row = L[0] + r * Wp[0];
            float best = -INFINITY;
            int bi = 0x7fffffff;
            for (int w2 = lane; w2 < W; w2 += 64) {
                const float v = (w2 <= w1) ? row[w2] : 0.f;
                if (v > best) { best = v; bi = w2; }
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                const float ov = __shfl_xor(best, o, 64);
                const int oi = __shfl_xor(bi, o, 64);
                if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
            }
            float sub = -INFINITY;
            for (int w2 = lane; w2 < W; w2 += 64) {
                float v = (w2 <= w1) ? row[w2] : 0.f;
                if (w2 >= bi - 1 && w2 <= bi + 1) v = 0.f;
                sub = fmaxf(sub, v);
            }
            sub = wave_max(sub);
            if (lane == 0) {
                const float m = (best - sub > 0.3f) ? 1.f : 0.f;
                const size_t o = (size_t)bh * W + w1;
                a.sdisp[o] = (float)(w1 - bi) * m;
                a.scost[o] = best * m;
                a.smask[o] = m;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// 4. lookup (corr.py:33-52 + utils.py:82-97).  Block = 64 consecutive pixels x 4 levels (one wave
//    per level, so the level is wave-uniform).  Lane = pixel: stores are 256-B coalesced per
//    channel; loads hit the skewed rows (contiguous along w1 when disparity is smooth).
// ------------------------------------------------------------------------------------------------
struct LookupArgs {
    const float* pyr[4];
    const float* coords;
    float* out;
    unsigned long long* stamps;     // nullable: [gridDim.x][2] device wall-clock (100 MHz) at block start / end
    int B, H, W, radius;
};

// Plain scalar parameters (not a struct): hipcc can then preload the first 16 kernarg dwords into SGPRs at wave launch
// (-mllvm -amdgpu-kernarg-preload-count), which takes the argument fetch off this latency-bound kernel's critical path.
template <int R, int LPB = 4>
__global__ __launch_bounds__(64 * LPB) void k_corr_lookup(const float* __restrict__ pyr0, const float* __restrict__ pyr1,
                                                     const float* __restrict__ pyr2, const float* __restrict__ pyr3,
                                                     const float* __restrict__ coords_p, float* __restrict__ out_p,
                                                     int Bn, int Hn, int Wn, int radius_n, unsigned long long* stamps_p) {
    struct { const float* coords; float* out; unsigned long long* stamps; int B, H, W, radius; } a =
        {coords_p, out_p, stamps_p, Bn, Hn, Wn, radius_n};
    const int lane = threadIdx.x & 63;
    // LPB pyramid levels per workgroup (one wave each); the 4 / LPB workgroups of a 64-pixel group are neighbours in the
    // grid.  (A level-major order that keeps all levels of a group on one XCD measured equal on HIP events, 2.85 us, and
    // slightly slower on the in-kernel interval, 2.58 vs 2.39 us, with unchanged PMC traffic: not kept.)
    const unsigned grp = blockIdx.x / (4 / LPB);
    const int level = __builtin_amdgcn_readfirstlane((int)(blockIdx.x % (4 / LPB)) * LPB + (int)(threadIdx.x >> 6));
    const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();      // first instruction; stored at the end
    const int HW = a.H * a.W;
    const unsigned total = (unsigned)a.B * (unsigned)HW;                      // host checks B*H*W < 2^31
    const unsigned p_raw = grp * 64u + (unsigned)lane;
    // No early exit and no per-tap branches: an out-of-range lane works on the last pixel and only its stores are
    // masked, every tap load is unconditional on a clamped address and zero-selected afterwards.  The kernel is three
    // dependent memory round trips long (arguments -> coordinate -> taps -> stores); branches around the loads made hipcc
    // fetch the arguments in four separate waits and issue each tap under its own exec mask.
    const bool active = p_raw < total;
    const unsigned p = active ? p_raw : total - 1;
    const int b = (int)(p / (unsigned)HW);
    const int hw = (int)(p - (unsigned)b * (unsigned)HW);
    const int h = hw / a.W, w1 = hw - h * a.W;
    const int radius = (R > 0) ? R : a.radius;
    const int taps = 2 * radius + 1;

    const int Wl = a.W >> level;
    float x = a.coords[p] * (1.0f / (float)(1 << level));
    x = fminf(fmaxf(x, -1048576.f), 1048576.f);        // keeps the int conversion defined; NaN -> -2^20
    if (!(x == x)) x = -1048576.f;
    const float x0 = floorf(x);
    const float fr = x - x0;
    const int j0 = (int)x0 - radius;
    const int q = w1 >> level;
    // wave-uniform select (a local array indexed by `level` would live in scratch)
    const float* pyr_l = level == 0 ? pyr0 : (level == 1 ? pyr1 : (level == 2 ? pyr2 : pyr3));
    const float* base = pyr_l + ((size_t)(b * a.H + h) * Wl) * a.W + w1;

    float* o = a.out + ((size_t)b * 4 * taps + (size_t)level * taps) * HW + hw;
    if (R > 0) {
        float v[2 * (R > 0 ? R : 1) + 2];
#pragma unroll
        for (int t = 0; t <= 2 * R + 1; ++t) {
            const int j = j0 + t;
            const bool ok = j >= 0 && j < Wl;
            int d = q - j;
            d = d < 0 ? d + Wl : (d >= Wl ? d - Wl : d);   // ragged widths: q can equal Wl
            d = ok ? d : 0;
            const float val = base[(size_t)d * a.W];
            v[t] = ok ? val : 0.f;
        }
        if (active) {
#pragma unroll
            for (int t = 0; t < 2 * R + 1; ++t) o[(size_t)t * HW] = (1.f - fr) * v[t] + fr * v[t + 1];
        }
    } else {
        float prev;
        {
            const int j = j0;
            int d = q - j;
            d = d < 0 ? d + Wl : (d >= Wl ? d - Wl : d);
            prev = (j >= 0 && j < Wl) ? base[(size_t)d * a.W] : 0.f;
        }
        for (int t = 0; t < taps; ++t) {
            const int j = j0 + t + 1;
            int d = q - j;
            d = d < 0 ? d + Wl : (d >= Wl ? d - Wl : d);
            const float nxt = (j >= 0 && j < Wl) ? base[(size_t)d * a.W] : 0.f;
            if (active) o[(size_t)t * HW] = (1.f - fr) * prev + fr * nxt;
            prev = nxt;
        }
    }
    if (a.stamps && lane == 0) {
        __builtin_amdgcn_s_waitcnt(0);      // this wave's stores have been issued and acknowledged
        if (threadIdx.x == 0) a.stamps[2 * blockIdx.x] = t_start;            // one slot pair per workgroup
        atomicMax(&a.stamps[2 * blockIdx.x + 1], (unsigned long long)__builtin_amdgcn_s_memrealtime());
    }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
extern "C" {

size_t tcs_corr_level_bytes(int B, int H, int W, int level) {
    if (B <= 0 || H <= 0 || W <= 0 || level < 0 || level > 3) return 0;
    return (size_t)B * H * (size_t)(W >> level) * W * sizeof(float);
}

static size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

size_t tcs_corr_build_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return align256((size_t)B * H * W * W * sizeof(float)) + align256((size_t)2 * B * H * W * sizeof(float));
}

float* tcs_corr_ws_level0(void* workspace) { return reinterpret_cast<float*>(workspace); }

int tcs_corr_build(const float* fmap1, const float* fmap2, int B, int C, int H, int W,
                   float* pyr0, float* pyr1, float* pyr2, float* pyr3,
                   float* nat1, float* nat2, float* nat3, float* cost_volume,
                   float* sparse_disp, float* sparse_cost, float* sparse_mask,
                   void* workspace, tcs_stream_t stream) {
    if (!fmap1 || !fmap2 || !pyr0 || !pyr1 || !pyr2 || !pyr3 || !workspace) return TCS_EINVAL;
    if (B <= 0 || C <= 0 || H <= 0 || W < 8) return TCS_EINVAL;
    const int n_sparse = (sparse_disp != nullptr) + (sparse_cost != nullptr) + (sparse_mask != nullptr);
    if (n_sparse != 0 && n_sparse != 3) return TCS_EINVAL;
    if ((long long)B * H > 65535) return TCS_EUNSUPPORTED;
    // LDS of the finalize kernel: 32 rows x (W + W/2 + W/4 + W/8 + pads)
    size_t lds = 0;
    for (int i = 0; i < 4; ++i) lds += (size_t)32 * ((W >> i) | 1) * sizeof(float);
    if (lds > 160 * 1024) return TCS_EUNSUPPORTED;

    hipStream_t s = tcs_stream(stream);
    float* vol = reinterpret_cast<float*>(workspace);
    float* rn = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + align256((size_t)B * H * W * W * sizeof(float)));
    const int HW = H * W;
    hipLaunchKernelGGL(k_inv_norm, dim3(tcs_cdiv(HW, 64), B, 2), dim3(256), 0, s, fmap1, fmap2, C, HW, rn);
    hipLaunchKernelGGL(k_corr_gemm, dim3(tcs_cdiv(W, 64), tcs_cdiv(W, 64), B * H), dim3(256), 0, s,
                       fmap1, fmap2, rn, B, C, H, W, vol);
    FinalizeArgs fa;
    fa.vol = vol;
    fa.pyr[0] = pyr0; fa.pyr[1] = pyr1; fa.pyr[2] = pyr2; fa.pyr[3] = pyr3;
    fa.nat[0] = nullptr; fa.nat[1] = nat1; fa.nat[2] = nat2; fa.nat[3] = nat3;
    fa.cost = cost_volume;
    fa.sdisp = sparse_disp; fa.scost = sparse_cost; fa.smask = sparse_mask;
    fa.B = B; fa.H = H; fa.W = W;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_corr_finalize),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return TCS_ELAUNCH;
    }
    hipLaunchKernelGGL(k_corr_finalize, dim3(tcs_cdiv(W, 32), B * H), dim3(256), lds, s, fa);
    return tcs_launch_status();
}

// Pyramid levels per workgroup.  One sequence (300 pixel groups at 640x480) is latency-bound: 1,200 single-wave workgroups
// spread evenly over the 256 CUs and finish 12 % sooner than 300 four-wave ones (2.96 vs 3.36 us per launch); large grids
// (several sequences per launch) stream better with four levels per workgroup (7.0 vs 9.6 us at 4 sequences).
static int lookup_levels_per_block(long long groups) { return groups <= 512 ? 1 : 4; }

int tcs_corr_lookup_blocks(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const long long groups = tcs_cdiv((long long)B * H * W, 64);
    const long long g8 = (groups + 7) / 8 * 8;
    return (int)(g8 * (4 / lookup_levels_per_block(groups)));
}

int tcs_corr_lookup(const float* pyr0, const float* pyr1, const float* pyr2, const float* pyr3,
                    const float* coords, int B, int H, int W, int radius, float* out, unsigned long long* stamps,
                    tcs_stream_t stream) {
    if (!pyr0 || !pyr1 || !pyr2 || !pyr3 || !coords || !out) return TCS_EINVAL;
    if (B <= 0 || H <= 0 || W < 8 || radius < 0 || radius > 16) return TCS_EINVAL;
    if ((long long)B * H * W >= 2147483647LL) return TCS_EINVAL;               // the kernel indexes pixels in 32 bits
    LookupArgs a;
    a.pyr[0] = pyr0; a.pyr[1] = pyr1; a.pyr[2] = pyr2; a.pyr[3] = pyr3;
    a.coords = coords; a.out = out; a.stamps = stamps; a.B = B; a.H = H; a.W = W; a.radius = radius;
    const int blocks = tcs_cdiv((long long)B * H * W, 64);
    const int lpb = lookup_levels_per_block(blocks);
    const int g8 = (blocks + 7) / 8 * 8;
    if (radius == 4 && lpb == 1)
        hipLaunchKernelGGL((k_corr_lookup<4, 1>), dim3(g8 * 4), dim3(64), 0, tcs_stream(stream), pyr0, pyr1, pyr2, pyr3, coords, out, B,
                           H, W, radius, stamps);
    else if (radius == 4)
        hipLaunchKernelGGL(k_corr_lookup<4>, dim3(g8), dim3(256), 0, tcs_stream(stream), pyr0, pyr1, pyr2, pyr3, coords, out, B, H, W,
                           radius, stamps);
    else
        hipLaunchKernelGGL(k_corr_lookup<0>, dim3(g8), dim3(256), 0, tcs_stream(stream), pyr0, pyr1, pyr2, pyr3, coords, out, B, H, W,
                           radius, stamps);
    return tcs_launch_status();
}

}  // extern "C"
