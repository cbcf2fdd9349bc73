// Convolutions of the TC-Stereo update step on the gfx950 matrix cores.
//
// Implicit GEMM, D[cout][pixel] = sum_k W[cout][k] * X[k][pixel], k = (channel, tap), computed with
// v_mfma_f32_32x32x2_f32: fp32 operands, fp32 accumulation, bit-for-bit an fmaf chain, so results
// stay within summation-order noise of the reference's fp32 CPU convolutions.
//
//   - A operand (weights)  lane l: W[cout = l&31][k = l>>5]   -> ds_read_b32, 32 consecutive floats per half
//   - B operand (input)    lane l: X[k = l>>5][pixel = l&31]   -> ds_read_b32, 32 consecutive floats per half
//   - D: lane holds pixel (l&31); registers walk 16 output channels -> NCHW stores are 128-B rows.
//
// A block owns a 4-row x 32-column patch of one image and NT = 32*MT output channels; wave w owns
// patch row w.  The input halo tile and the weight slice of one channel chunk are staged through
// LDS; the 9 taps of a 3x3 filter re-read the same LDS tile at shifted offsets, so HBM/L2 sees each
// input element once per block.  torch.cat of the reference (update.py:79-80,84) is virtual: the
// staging loop walks up to four source tensors.  Bias, context addends, activations and the GRU
// gate arithmetic (update.py:81-85, 30-34, 62-66) run in the epilogue on the accumulators.
//
// Block -> (patch, cout tile): cout tile = blockIdx.x % n_tiles.  Blocks b and b+8 share an XCD
// (round-robin dispatch), so with 1, 2, 4 or 8 cout tiles every XCD's L2 holds a single weight slice.
#include "tcs_conv_common.h"

extern "C" size_t tcs_conv_packed_floats_f16x3(int Cout, int Cin, int ksize);

template <int KS, int MT, int KC, int EPI>
__device__ __forceinline__ void conv_mfma_body(const ConvArgs& a, const int bid, const int b) {
    constexpr int NT = 32 * MT, HALO = KS / 2, IH = 4 + KS - 1, IW = 32 + KS - 1, TAPS = KS * KS;
    constexpr int IN_CH = IH * IW;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* s_in = lds;                       // [KC][IH][IW]
    float* s_w = lds + ((KC * IN_CH + 3) & ~3);   // [KC][TAPS][NT], 16-B aligned

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = bid % a.nct, patch = bid / a.nct;
    const int y0 = (patch / a.npx) * 4, x0 = (patch % a.npx) * 32;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;

    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;

    // ---- staging plan, fixed for the whole K loop ----------------------------------------------
    // input: thread -> one halo position (row r, column cc) of channel lane `sub`; per chunk it only
    // adds the channel base to a precomputed pixel offset (no div/mod in the loop).
    constexpr int R = (256 / IN_CH) > 0 ? (256 / IN_CH) : 1;      // channel lanes (3x3: 1, 1x1: 2)
    constexpr int IN_PT = (KC + R - 1) / R;                       // input values per thread per chunk
    constexpr int W4 = KC * TAPS * NT / 4;                        // float4s of one weight chunk
    constexpr int W_PT = (W4 + 255) / 256;
    const int sub = tid / IN_CH, pos = tid - sub * IN_CH;
    const bool in_active = sub < R;
    const int sr = pos / IW, sc = pos - sr * IW;
    const int gy = y0 - HALO + sr, gx = x0 - HALO + sc;
    const bool in_ok = in_active && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const size_t pixoff = in_ok ? (size_t)gy * W + gx : 0;
    const float* wsrc = a.w + ct * NT;

    float in_reg[IN_PT];
    f32x4 w_reg[W_PT];

    // (macros, not lambdas: by-reference lambda captures kept these arrays in scratch memory)
#define TCS_LOAD_CHUNK(C0)                                                                              \
    {                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < IN_PT; ++j) {                                             \
            const int g = (C0) + sub + j * R;                                                           \
            /* unconditional load from a clamped (valid) address; zero-selected in TCS_STORE_CHUNK, so nothing waits \
               for the value before the MFMA phase */                                                   \
            in_reg[j] = conv_src_ptr(a, b, min(g, a.Cin - 1), HW)[pixoff];                              \
        }                                                                                               \
        _Pragma("unroll") for (int j = 0; j < W_PT; ++j) {                                              \
            const int idx = min(tid + 256 * j, W4 - 1);                                                 \
            const int row = idx / (NT / 4), q = idx % (NT / 4);                                         \
            w_reg[j] = *reinterpret_cast<const f32x4*>(wsrc + ((size_t)(C0) * TAPS + row) * a.CoutPad + q * 4); \
        }                                                                                               \
    }
#define TCS_STORE_CHUNK(C0)                                                                             \
    {                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < IN_PT; ++j)                                               \
            if (in_active && sub + j * R < KC)                                                          \
                s_in[(sub + j * R) * IN_CH + pos] = (in_ok && (C0) + sub + j * R < a.Cin) ? in_reg[j] : 0.f; \
        _Pragma("unroll") for (int j = 0; j < W_PT; ++j) {                                              \
            const int idx = tid + 256 * j;                                                              \
            if (idx < W4) *reinterpret_cast<f32x4*>(s_w + idx * 4) = w_reg[j];                         \
        }                                                                                               \
    }

    // ---- K loop: global loads of chunk i+1 are in flight while the matrix cores work on chunk i ----
    const int cin_loop = (a.Cin + KC - 1) / KC * KC;
    TCS_LOAD_CHUNK(0)
    TCS_STORE_CHUNK(0)
    __syncthreads();
    for (int c0 = 0; c0 < cin_loop; c0 += KC) {
        const bool has_next = c0 + KC < cin_loop;
        if (has_next) TCS_LOAD_CHUNK(c0 + KC)
#pragma unroll
        for (int t = 0; t < TAPS; ++t) {
            const int dy = t / KS, dx = t % KS;
            const float* pin = s_in + (wave + dy) * IW + dx + l31 + half * IN_CH;
            const float* pw = s_w + (half * TAPS + t) * NT + l31;
#pragma unroll
            for (int kk = 0; kk < KC; kk += 2) {
                const float bv = pin[kk * IN_CH];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const float av = pw[kk * TAPS * NT + m * 32];
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[m], 0, 0, 0);
                }
            }
        }
        if (has_next) {
            __syncthreads();            // every wave has finished reading this chunk
            TCS_STORE_CHUNK(c0 + KC)    // (waits here for the prefetched registers)
            __syncthreads();
        }
    }

#undef TCS_LOAD_CHUNK
#undef TCS_STORE_CHUNK

    // ---- epilogue ----
    const int px = x0 + l31, py = y0 + wave;
    if (px >= W || py >= H) return;
    const size_t pix = (size_t)py * W + px;
#pragma unroll
    for (int m = 0; m < MT; ++m) conv_epilogue_tile<EPI>(a, b, ct * NT + m * 32 + 4 * half, pix, HW, acc[m], 1.0f);
}

template <int KS, int MT, int KC, int EPI>
__global__ __launch_bounds__(256) void k_conv_mfma(ConvArgs a) {
    conv_mfma_body<KS, MT, KC, EPI>(a, blockIdx.x, blockIdx.y);
}

// Two independent convolutions of the same tile instance as ONE launch: blocks [0, n0) are problem 0's, the rest problem 1's
// (the first layers of the gradient predictor's two stems, core/update.py:200-205 — see k_conv_s16_pair in tcs_conv_s16.hip for
// what a grouped launch saves).  Bit-equal to the two separate launches.
template <int KS, int MT, int KC, int EPI>
__global__ __launch_bounds__(256) void k_conv_mfma_pair(ConvArgs a0, ConvArgs a1, int n0) {
    const int bid = blockIdx.x;
    if (bid < n0) conv_mfma_body<KS, MT, KC, EPI>(a0, bid, blockIdx.y);
    else conv_mfma_body<KS, MT, KC, EPI>(a1, bid - n0, blockIdx.y);
}

// epilogue of the single-input-channel kernels: 16 output channels co_lo .. co_lo+15 of one pixel -> fp32 NCHW and/or S16
__device__ __forceinline__ void cin1_store16(const ConvArgs& a, int b, int co_lo, int y, int x, int p, int HW, const float* acc) {
    float v[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        const int co = min(co_lo + c, a.Cout - 1);
        float t = acc[c] + (a.bias ? a.bias[co] : 0.f);
        if (a.add1) t += a.add1[((size_t)b * a.Cout + co) * HW + p];
        v[c] = co_lo + c < a.Cout ? apply_act(t, a.act) * a.post_scale : 0.f;
    }
    if (a.out) {
#pragma unroll
        for (int c = 0; c < 16; ++c)
            if (co_lo + c < a.Cout) a.out[((size_t)b * a.out_ctot + a.out_coff + co_lo + c) * HW + p] = v[c];
    }
    if (a.out16) {                                      // two 8-channel units (Cout is a multiple of 8 on this path)
        const int Hp = a.H + 2, Wp = a.W + 2;
#pragma unroll
        for (int g = 0; g < 2; ++g)
            if (co_lo + 8 * g < a.Cout)
                s16_store8(a.out16 + s16_unit(b, a.out16_groups, a.out16_goff + (co_lo >> 3) + g, 0, Hp, Wp, y, x), (size_t)Hp * Wp * 8, v + 8 * g);
    }
}

// Single-input-channel convolutions (BasicMotionEncoder.convf1 7x7, HiddenstateUpdater.convs.0 1x1):
// K is too small for the matrix cores to pay; one thread per pixel, 16 output channels per block.z,
// weights are wave-uniform (scalar loads).
template <int KS>
__global__ __launch_bounds__(256) void k_conv_cin1(ConvArgs a) {
    constexpr int HALO = KS / 2, TAPS = KS * KS;
    const int b = blockIdx.y, H = a.H, W = a.W;
    const int HW = H * W;
    const int p_raw = blockIdx.x * 256 + threadIdx.x;
    const bool active = p_raw < HW;
    const int p = active ? p_raw : HW - 1;
    const int y = p / W, x = p - y * W;
    const float* s = a.src[0] + (size_t)b * HW;
    // taps: unconditional loads on clamped coordinates, zero-selected afterwards (a "load or zero" branch per tap makes
    // hipcc wait for each load in turn)
    float in[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
        const int yy = y + t / KS - HALO, xx = x + t % KS - HALO;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        const float v = s[min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];
        in[t] = ok ? v : 0.f;
    }
    // 16 output channels per block.z; their TAPS x 16 weights go through LDS and are read back as broadcasts
    // (784 wave-uniform scalar loads for the 7x7 stem overflow the SGPR file and spill)
    const int co_lo = blockIdx.z * 16;
    __shared__ __attribute__((aligned(16))) float s_w[TAPS * 16];
    for (int i = threadIdx.x; i < TAPS * 16; i += 256) s_w[i] = a.w[(size_t)(i >> 4) * a.CoutPad + co_lo + (i & 15)];
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.f;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) {
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const f32x4 w4 = *reinterpret_cast<const f32x4*>(&s_w[t * 16 + c4 * 4]);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c4 * 4 + c] = fmaf(w4[c], in[t], acc[c4 * 4 + c]);
        }
        // one filter row at a time: left alone, the scheduler hoists all TAPS*4 LDS reads above the FMAs and spills
        if (t % KS == KS - 1) { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); }
    }
    if (!active) return;
    cin1_store16(a, b, co_lo, y, x, p, HW, acc);
}

// 7x7 stems: CIN = 1 (BasicMotionEncoder.convf1 on the flow, update.py:113) and CIN = 3 (the RGB stems of the feature /
// context encoders, extractor.py:205,270: the last layer that ran on MIOpen).  K = 49*CIN is small and the tile-halo ratio of
// a 7x7 window is poor for the 32-pixel MFMA tiles, so this is a plain fp32 FMA kernel (bit-for-bit an fmaf chain like the
// reference's CPU convolution up to summation order): 64 x 4 pixel tile per block, the tile (+3 halo) of every input channel and
// the block's 49*CIN x 16 weights live in LDS, the filter rows run as a rolled loop so that nothing spills (fully unrolled, the
// weight reads were hoisted and spilled to scratch).  16 output channels per blockIdx.z.
// 8 output channels co_lo .. co_lo+7 (one S16 unit) of one pixel -> fp32 NCHW and/or S16
__device__ __forceinline__ void cin1_store8(const ConvArgs& a, int b, int co_lo, int y, int x, int p, int HW, const float* acc) {
    float v[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int co = min(co_lo + c, a.Cout - 1);
        float t = acc[c] + (a.bias ? a.bias[co] : 0.f);
        if (a.add1) t += a.add1[((size_t)b * a.Cout + co) * HW + p];
        v[c] = co_lo + c < a.Cout ? apply_act(t, a.act) * a.post_scale : 0.f;
    }
    if (a.out) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            if (co_lo + c < a.Cout) a.out[((size_t)b * a.out_ctot + a.out_coff + co_lo + c) * HW + p] = v[c];
    }
    if (a.out16 && co_lo < a.Cout)
        s16_store8(a.out16 + s16_unit(b, a.out16_groups, a.out16_goff + (co_lo >> 3), 0, a.H + 2, a.W + 2, y, x), (size_t)(a.H + 2) * (a.W + 2) * 8, v);
}

template <int CIN>
__global__ __launch_bounds__(256) void k_conv7x7(ConvArgs a) {
    constexpr int KS = 7, HALO = 3, TW = 64, TH = 4, IW = TW + 2 * HALO, IH = TH + 2 * HALO, NC = 8;
    __shared__ float s_in[CIN * IH * IW];
    const int b = blockIdx.y, H = a.H, W = a.W, HW = H * W;
    const int ntx = (W + TW - 1) / TW;
    const int tx0 = (blockIdx.x % ntx) * TW, ty0 = (blockIdx.x / ntx) * TH;
    const int co_lo = blockIdx.z * NC;
    // RGB stem: the left | right batch concatenation and the image normalisation of tc_stereo.py:101-107 happen here
    const float* s = (CIN == 3 && a.src_b2 && b >= a.b_split) ? a.src_b2 + (size_t)(b - a.b_split) * CIN * HW : a.src[0] + (size_t)b * CIN * HW;
    const bool norm = CIN == 3 && a.in_transform == 1;
    for (int i = threadIdx.x; i < CIN * IH * IW; i += 256) {
        const int ci = i / (IH * IW), q = i - ci * (IH * IW);
        const int r = q / IW, c = q - r * IW;
        const int yy = ty0 - HALO + r, xx = tx0 - HALO + c;
        const bool ok = yy >= 0 && yy < H && xx >= 0 && xx < W;
        float v = s[(size_t)ci * HW + min(max(yy, 0), H - 1) * W + min(max(xx, 0), W - 1)];
        if (norm) v = 2.0f * (v / 255.0f) - 1.0f;           // the reference's own arithmetic (x2 is exact, so a fused multiply-add changes nothing)
        s_in[i] = ok ? v : 0.f;
    }
    __syncthreads();
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6;
    float acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) acc[c] = 0.f;
    // Weights: the block's NC output channels of one tap are NC contiguous floats of the packed weights at a block-uniform
    // address, so a filter row is 7 scalar loads (constant cache, 56 SGPRs) issued together.  They must NOT go through LDS:
    // wave-uniform `ds_read_b128` reads of an LDS copy returned wrong values for lanes 48-63 whenever the workgroup shared its
    // CU with another kernel's workgroups (tools/race_probe.py; tests/test_gpu_parity.py::test_stem7x7_beside_concurrent_kernels),
    // although the kernel was correct when it ran alone.
    const float* wb = a.w + co_lo;
#pragma unroll 1
    for (int ct = 0; ct < CIN * KS; ++ct) {                  // (input channel, filter row)
        const int ci = ct / KS, ty = ct - ci * KS;
        float w[KS][NC];
#pragma unroll
        for (int tx = 0; tx < KS; ++tx)
#pragma unroll
            for (int c = 0; c < NC; ++c) w[tx][c] = wb[(size_t)(ct * KS + tx) * a.CoutPad + c];
        const float* in_row = s_in + (ci * IH + ly + ty) * IW + lx;
#pragma unroll
        for (int tx = 0; tx < KS; ++tx) {
            const float v = in_row[tx];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fmaf(w[tx][c], v, acc[c]);
        }
    }
    const int x = tx0 + lx, y = ty0 + ly;
    if (x >= W || y >= H) return;
    cin1_store8(a, b, co_lo, y, x, y * W + x, HW, acc);
}

__global__ __launch_bounds__(256) void k_pack_weight(const float* __restrict__ w, int Cout, int Cin, int taps, int CinPad, int CoutPad,
                                                     float* __restrict__ packed) {
    const size_t n = (size_t)CinPad * taps * CoutPad;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int co = (int)(i % CoutPad);
    const int t = (int)((i / CoutPad) % taps);
    const int ci = (int)(i / ((size_t)CoutPad * taps));
    packed[i] = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + t] : 0.f;
}

// a launch that tcs_conv2d_group has asked to be planned instead of issued (tcs_conv_s16.hip: S16Plan)
struct ConvPlan {
    bool filled;
    int key;                        // KS * 10000 + MT * 1000 + KC * 10 + EPI
    ConvArgs args;
    int (*launch_alone)(const ConvArgs&, hipStream_t);
};
static thread_local ConvPlan* g_conv_plan = nullptr;

template <int KS, int MT, int KC, int EPI>
static int launch_mfma(const ConvArgs& a, hipStream_t s) {
    constexpr int IH = 4 + KS - 1, IW = 32 + KS - 1, TAPS = KS * KS, NT = 32 * MT;
    const size_t lds = ((size_t)((KC * IH * IW + 3) & ~3) + (size_t)KC * TAPS * NT) * sizeof(float);
    auto kern = k_conv_mfma<KS, MT, KC, EPI>;
    if (g_conv_plan && !g_conv_plan->filled) {
        g_conv_plan->filled = true;
        g_conv_plan->key = KS * 10000 + MT * 1000 + KC * 10 + EPI;
        g_conv_plan->args = a;
        g_conv_plan->launch_alone = &launch_mfma<KS, MT, KC, EPI>;
        return TCS_OK;
    }
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return TCS_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(a.npatch * a.nct, a.B), dim3(256), lds, s, a);
    return tcs_launch_status();
}

// KC per tile shape: the narrow tile gets a deeper chunk so that each barrier pair covers >= 72 MFMAs per wave.
template <int KS, int EPI>
static int launch_by_tile(ConvArgs& a, int nt_pack, hipStream_t s) {
    // Occupancy-driven tile choice: the widest cout tile (most input reuse) that still yields >= 2 blocks per CU;
    // small feature maps (1/8, 1/16 scale) fall back to 32-wide tiles so every SIMD gets a wave.
    int nt = nt_pack;
    while (nt > 32 && (long long)a.npatch * a.B * (a.CoutPad / nt) < 512) nt >>= 1;
    a.nct = a.CoutPad / nt;
    constexpr int KCW = (KS == 1) ? 32 : 8, KCN = (KS == 1) ? 32 : 16;
    switch (nt) {
        case 128: return launch_mfma<KS, 4, KCW, EPI>(a, s);
        case 64: return launch_mfma<KS, 2, KCW, EPI>(a, s);
        default: return launch_mfma<KS, 1, KCN, EPI>(a, s);
    }
}

extern "C" {

size_t tcs_conv_packed_floats(int Cout, int Cin, int ksize) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3 && ksize != 7)) return 0;
    return (size_t)round_up(Cin, 32) * ksize * ksize * round_up(Cout, cout_tile(Cout));
}

int tcs_pack_conv_weight(const float* w_oihw, int Cout, int Cin, int ksize, float* packed, tcs_stream_t stream) {
    const size_t n = tcs_conv_packed_floats(Cout, Cin, ksize);
    if (!w_oihw || !packed || n == 0) return TCS_EINVAL;
    hipLaunchKernelGGL(k_pack_weight, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tcs_stream(stream), w_oihw, Cout, Cin,
                       ksize * ksize, round_up(Cin, 32), round_up(Cout, cout_tile(Cout)), packed);
    return tcs_launch_status();
}

int tcs_conv2d(const tcs_conv_desc* d, tcs_stream_t stream) {
    if (!d || !d->weight || (!d->out && !d->out16)) return TCS_EINVAL;
    if (d->out16 && (d->epilogue != TCS_EPI_LINEAR || d->stride == 2 || d->out16_group_offset < 0 ||
                     d->out16_group_offset + (d->Cout + 7) / 8 > d->out16_groups || (d->Cin == 1 && d->Cout % 8 != 0)))
        return TCS_EINVAL;
    if (!d->out && d->epilogue != TCS_EPI_LINEAR) return TCS_EINVAL;
    if (d->n_src < 1 || d->n_src > TCS_MAX_SRC) return TCS_EINVAL;
    if (d->B <= 0 || d->B > 65535 || d->H <= 0 || d->W <= 0 || d->Cin <= 0 || d->Cout <= 0) return TCS_EINVAL;
    ConvArgs a;
    int tot = 0;
    for (int i = 0; i < TCS_MAX_SRC; ++i) {
        const bool used = i < d->n_src;
        if (used && (!d->src[i] || d->src_ch[i] <= 0)) return TCS_EINVAL;
        a.src[i] = used ? d->src[i] : d->src[0];
        a.src_ch[i] = used ? d->src_ch[i] : 1;
        tot += used ? d->src_ch[i] : 0;
        a.src_end[i] = used ? tot : 0x7fffffff;
    }
    if (tot != d->Cin) return TCS_EINVAL;
    const int nt = cout_tile(d->Cout);
    a.w = d->weight; a.bias = d->bias;
    const int stride = d->stride == 2 ? 2 : 1;
    a.Hin = d->H; a.Win = d->W;
    a.B = d->B; a.H = stride == 2 ? (d->H - 1) / 2 + 1 : d->H; a.W = stride == 2 ? (d->W - 1) / 2 + 1 : d->W; a.Cin = d->Cin; a.Cout = d->Cout; a.CoutPad = round_up(d->Cout, nt);
    if (d->act == TCS_ACT_RELU_ADD_RELU && (!d->addend || d->epilogue != TCS_EPI_LINEAR || d->Cin == 1)) return TCS_EINVAL;
    a.act = d->act; a.post_scale = d->post_scale;
    a.add1 = d->addend; a.add2 = d->addend2; a.h = d->h; a.z = d->z;
    a.keep_z = d->blend_keep_z; a.hidden = 0;
    a.out = d->out; a.out_ctot = d->out_ctot; a.out_coff = d->out_coff; a.out2 = d->out2;
    a.out16 = reinterpret_cast<_Float16*>(d->out16); a.out16_groups = d->out16_groups; a.out16_goff = d->out16_group_offset;
    a.in_transform = d->in_transform; a.src_b2 = d->src_batch2; a.b_split = d->batch_split;
    if (a.in_transform || a.src_b2) {                   // honoured by the 7x7 RGB stem only
        if (d->ksize != 7 || d->Cin != 3 || d->n_src != 1 || d->math != TCS_MATH_F32 || d->epilogue != TCS_EPI_LINEAR) return TCS_EUNSUPPORTED;
        if (a.in_transform != 0 && a.in_transform != 1) return TCS_EINVAL;
        if (a.src_b2 && (a.b_split <= 0 || a.b_split >= d->B)) return TCS_EINVAL;
    }
    a.npx = tcs_cdiv(d->W, 32);
    a.npatch = a.npx * tcs_cdiv(d->H, 4);
    a.nct = a.CoutPad / nt;
    a.w_unscale = 1.0f;
    a.w_bytes = 0;
    hipStream_t s = tcs_stream(stream);

    if (d->math == TCS_MATH_F16X3) {
        // fp16-split matrix-core path (tcs_conv_f16.hip); weights must come from tcs_pack_conv_weight_f16x3
        if (d->Cin == 1 || (d->ksize != 1 && d->ksize != 3)) return TCS_EUNSUPPORTED;
        if (d->epilogue == TCS_EPI_LINEAR && d->out && (d->out_ctot < d->out_coff + d->Cout || d->out_coff < 0)) return TCS_EINVAL;
        const bool gru = d->epilogue == TCS_EPI_GRU_ZR || d->epilogue == TCS_EPI_GRU_Q;
        if (gru && !d->h) return TCS_EINVAL;
        if (d->epilogue == TCS_EPI_GRU_ZR && (!d->out2 || (d->Cout & 1))) return TCS_EINVAL;
        if (d->epilogue == TCS_EPI_GRU_Q && !d->z) return TCS_EINVAL;
        if (d->epilogue == TCS_EPI_DECONV2X && (d->Cout % 4 != 0 || stride != 1)) return TCS_EINVAL;
        a.hidden = d->epilogue == TCS_EPI_GRU_ZR ? d->Cout / 2 : (d->epilogue == TCS_EPI_GRU_Q ? d->Cout : 0);
        if (d->epilogue == TCS_EPI_DECONV2X) a.hidden = d->Cout / 4;
        a.CoutPad = round_up(d->Cout, 32);
        a.w_unscale = d->weight_unscale;
        a.w_bytes = (int)(tcs_conv_packed_floats_f16x3(d->Cout, d->Cin, d->ksize) * sizeof(float));
        a.npx = tcs_cdiv(a.W, 32);
        return tcs_conv_f16x3_launch(a, d->ksize, d->epilogue, stride, s);
    }
    if (d->math != TCS_MATH_F32) return TCS_EINVAL;
    if (stride != 1 || d->epilogue == TCS_EPI_DECONV2X) return TCS_EUNSUPPORTED;      // fp32 kernel: stride-1 'same' only

    if (d->epilogue == TCS_EPI_LINEAR) {
        if (d->out && (d->out_ctot < d->out_coff + d->Cout || d->out_coff < 0)) return TCS_EINVAL;
        if (d->Cin == 1) {
            const dim3 grid(tcs_cdiv((long long)d->H * d->W, 256), d->B, tcs_cdiv(d->Cout, 16));
            if (d->ksize == 1) hipLaunchKernelGGL(k_conv_cin1<1>, grid, dim3(256), 0, s, a);
            else if (d->ksize == 3) hipLaunchKernelGGL(k_conv_cin1<3>, grid, dim3(256), 0, s, a);
            else if (d->ksize == 7) {
                const dim3 g7(tcs_cdiv(d->W, 64) * tcs_cdiv(d->H, 4), d->B, tcs_cdiv(d->Cout, 8));
                hipLaunchKernelGGL(k_conv7x7<1>, g7, dim3(256), 0, s, a);
            }
            else return TCS_EUNSUPPORTED;
            return tcs_launch_status();
        }
        if (d->ksize == 3) return launch_by_tile<3, TCS_EPI_LINEAR>(a, nt, s);
        if (d->ksize == 1) return launch_by_tile<1, TCS_EPI_LINEAR>(a, nt, s);
        if (d->ksize == 7 && d->Cin == 3 && d->n_src == 1 && d->act != TCS_ACT_RELU_ADD_RELU) {   // RGB stem
            const dim3 g7(tcs_cdiv(d->W, 64) * tcs_cdiv(d->H, 4), d->B, tcs_cdiv(d->Cout, 8));
            hipLaunchKernelGGL(k_conv7x7<3>, g7, dim3(256), 0, s, a);
            return tcs_launch_status();
        }
        return TCS_EUNSUPPORTED;
    }
    // GRU epilogues: hidden = Cout/2 (ZR) or Cout (Q); tiles must not straddle the z|r boundary
    if (!d->h) return TCS_EINVAL;
    if (d->epilogue == TCS_EPI_GRU_ZR) {
        if (!d->out2 || (d->Cout & 1)) return TCS_EINVAL;
        a.hidden = d->Cout / 2;
        if (d->ksize == 3) return launch_by_tile<3, TCS_EPI_GRU_ZR>(a, nt, s);
        if (d->ksize == 1) return launch_by_tile<1, TCS_EPI_GRU_ZR>(a, nt, s);
        return TCS_EUNSUPPORTED;
    }
    if (d->epilogue == TCS_EPI_GRU_Q) {
        if (!d->z) return TCS_EINVAL;
        a.hidden = d->Cout;
        if (d->ksize == 3) return launch_by_tile<3, TCS_EPI_GRU_Q>(a, nt, s);
        if (d->ksize == 1) return launch_by_tile<1, TCS_EPI_GRU_Q>(a, nt, s);
        return TCS_EUNSUPPORTED;
    }
    return TCS_EINVAL;
}

int tcs_conv2d_group(const tcs_conv_desc* const* descs, int n, tcs_stream_t stream) {
    if (!descs || n < 1 || n > 2) return TCS_EINVAL;
    if (n == 1) return tcs_conv2d(descs[0], stream);
    ConvPlan plan[2];
    for (int i = 0; i < 2; ++i) {
        plan[i].filled = false;
        g_conv_plan = &plan[i];
        const int rc = tcs_conv2d(descs[i], stream);      // kernels without a planner (fp16-split, 7x7, single-channel) launch right here
        g_conv_plan = nullptr;
        if (rc != TCS_OK) return rc;
    }
    hipStream_t s = tcs_stream(stream);
    constexpr int key = 3 * 10000 + 1 * 1000 + 16 * 10 + TCS_EPI_LINEAR;
    if (plan[0].filled && plan[1].filled && plan[0].key == key && plan[1].key == key && plan[0].args.B == plan[1].args.B) {
        constexpr size_t lds = ((size_t)((16 * 6 * 34 + 3) & ~3) + (size_t)16 * 9 * 32) * sizeof(float);
        const int n0 = plan[0].args.npatch * plan[0].args.nct, n1 = plan[1].args.npatch * plan[1].args.nct;
        hipLaunchKernelGGL((k_conv_mfma_pair<3, 1, 16, TCS_EPI_LINEAR>), dim3(n0 + n1, plan[0].args.B), dim3(256), lds, s, plan[0].args,
                           plan[1].args, n0);
        return tcs_launch_status();
    }
    for (int i = 0; i < 2; ++i) {
        if (!plan[i].filled) continue;                     // (already issued)
        const int r = plan[i].launch_alone(plan[i].args, s);
        if (r != TCS_OK) return r;
    }
    return TCS_OK;
}

}  // extern "C"

// this translation unit's S16 domain flag (tcs_s16.h): read-and-clear for tcs_s16_flags()
int tcs_s16_flag_take_conv(unsigned int* out) {
    unsigned int v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(TCS_S16_FLAG_VAR), sizeof(v)) != hipSuccess) return TCS_ELAUNCH;
    if (v && hipMemcpyToSymbol(HIP_SYMBOL(TCS_S16_FLAG_VAR), &zero, sizeof(zero)) != hipSuccess) return TCS_ELAUNCH;
    *out |= v;
    return TCS_OK;
}
