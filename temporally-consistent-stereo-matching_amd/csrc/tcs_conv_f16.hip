// fp16-split convolution for gfx950: fp32-equivalent accuracy on the 16x faster half-precision matrix pipe.
//
// Every fp32 operand x is split exactly into two halves, x = hi + lo + eps, hi = fp16(x),
// lo = fp16(x - hi), |eps| <= 2^-22 |x|.  A product then needs three v_mfma_f32_32x32x16_f16
//     w*x ~= w_hi*x_hi + w_hi*x_lo + w_lo*x_hi            (the dropped w_lo*x_lo term is ~2^-22)
// whose partial products are exact in the fp32 accumulator (11+11 significand bits).  Net error per
// product ~2^-21 relative — the same order as fp32 rounding — at 16/3 = 5.3x the fp32 MFMA rate.
// Weights are pre-scaled by a power of two per layer (exactly undone in the epilogue) so that w_lo
// stays in fp16's normal range; activations saturate at +-65504 (never reached on this path: hidden
// states are tanh/sigmoid bounded, disparities < 2^9).  Parity with the fp32 CPU reference is
// enforced by the same tests and tolerances as the fp32 MFMA kernel (tests/test_gpu_parity.py).
//
// Tiling is the fp32 kernel's (4-row x 32-column patch, 32*MT output channels, wave = patch row).
// K runs over 16-channel groups per tap: one MFMA K-step = 16 channels of one filter tap.
// LDS images are laid out so that every operand fetch is ONE conflict-free ds_read_b128:
//   input   [kstep][h][position][8 halves]          lane (r,h) reads position base+r of half-group h
//   weights [kstep][tap][m][hi|lo][h*32 + r][8]     lane l reads slot l of a 1 KiB tile
// The weight image is produced once at pack time (tcs_pack_conv_weight_f16x3) in exactly this order,
// so staging weights is a straight 16-byte-per-lane copy; activations are split on the fly.
#include "tcs_conv_common.h"

#ifdef TCS_CONV_STAMPS
// Diagnostic build only (lib/libtcs_mi355_stamps.so): per-phase shader-clock sums of the K loop, per wave.
__device__ unsigned long long tcs_conv_stamps[8 * 16384];
#define TCS_STAMP(T) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(T) :: "memory"); }
#endif

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));

typedef float float2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));      // native vector: stays in registers (SROA)

__device__ __forceinline__ void split_f16(float x, _Float16& hi, _Float16& lo) {
    x = __builtin_amdgcn_fmed3f(x, -65504.f, 65504.f);
    hi = (_Float16)x;
    lo = (_Float16)(x - (float)hi);
}

// two values at a time: v_cvt_pk_f16_f32 (round-to-nearest-even, packed) halves the conversion work
__device__ __forceinline__ void split_f16x2(float x0, float x1, half2_t& hi, half2_t& lo) {
    float2_t v;
    v[0] = __builtin_amdgcn_fmed3f(x0, -65504.f, 65504.f);
    v[1] = __builtin_amdgcn_fmed3f(x1, -65504.f, 65504.f);
    hi = __builtin_convertvector(v, half2_t);
    const float2_t back = __builtin_convertvector(hi, float2_t);
    lo = __builtin_convertvector(v - back, half2_t);
}

// MT = 32-wide output-channel tiles per wave, MP = patch rows per wave (block patch = 4*MP rows x 32 columns),
// KSTEPS = 16-channel MFMA K-steps per LDS chunk; the block has ROWS waves and covers 32*MT output channels; wave `row`
// owns patch row(s) `row`.
// A wave issues one instruction every ~4-5 cycles, and a chunk costs ~700 non-MFMA instructions per block (loads,
// fp16 split, LDS traffic) against 27*MT*MP MFMAs per wave: with 4 waves the block is issue-bound at 4-5x the MFMA time
// (measured with in-kernel stamps, tools/conv_phases.py).
// ROWS = waves along the patch rows (patch = ROWS*MP rows): 5 instead of 4 turns the 600- and 300-workgroup grids of the
// 1/4-scale layers (2.3 and 1.2 workgroups per CU: some CUs carry one more than the others for the whole kernel) into
// 480 and 240 (at most 2 / 1 per CU).
template <int KS, int MT, int MP, int KSTEPS, int EPI, int STRIDE = 1, int ROWS = 4>
__global__ __launch_bounds__(64 * ROWS) void k_conv_f16x3(ConvArgs a) {
    constexpr int NTHREADS = 64 * ROWS, MTB = MT;                         // MTB: cout tiles per block
    constexpr int HALO = KS / 2, PR = ROWS * MP, IH = STRIDE * PR + KS - STRIDE, IW = STRIDE * 32 + KS - STRIDE, TAPS = KS * KS,
                  IN_CH = IH * IW;
    constexpr int NT = 32 * MTB, KC = 16 * KSTEPS, NG = 2 * KSTEPS;           // NG: 8-channel groups per chunk
    constexpr int IN_BYTES = NG * IN_CH * 16;                                  // one of {hi, lo}
    constexpr int W_UNITS = KSTEPS * TAPS * MTB * 2 * 64;                      // 16-byte units per chunk
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];
    unsigned char* s_in_hi = lds8;
    unsigned char* s_in_lo = lds8 + IN_BYTES;
    unsigned char* s_w = lds8 + 2 * IN_BYTES;

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % ROWS, wave_g = wave_all / ROWS;               // patch row, cout group
    const int bid = blockIdx.x;
    const int ct = bid % a.nct, patch = bid / a.nct;
    const int b = blockIdx.y;
    const int y0 = (patch / a.npx) * PR, x0 = (patch % a.npx) * 32;
    const int H = a.H, W = a.W;                     // output grid
    const size_t HW = (size_t)H * W;                // output plane (epilogue)
    const size_t HWi = (size_t)a.Hin * a.Win;       // input plane (staging)

    f32x16 acc[MT][MP];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int p = 0; p < MP; ++p)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[m][p][i] = 0.f;

    // ---- staging plan -------------------------------------------------------------------------------
    // A thread owns PPT halo positions and, for each, GPT of the chunk's NG 8-channel groups.  The group index
    // is uniform across a wave (across the block when PARTS == 1), so the per-chunk base pointer of a group is
    // computed on the scalar unit and every load is `global_load_dword v, v_pixel_offset, s[base]`.
    constexpr int PARTS = (NTHREADS / ((IN_CH <= 128) ? 128 : 256)) > 0 ? (NTHREADS / ((IN_CH <= 128) ? 128 : 256)) : 1;
    constexpr int TPP = NTHREADS / PARTS;                     // threads per part
    constexpr int PPT = (IN_CH + TPP - 1) / TPP;              // positions per thread
    constexpr int GPT = NG / PARTS;                           // groups per thread
    static_assert(NG % PARTS == 0, "groups must split evenly");
    const int part = PARTS == 1 ? 0 : __builtin_amdgcn_readfirstlane(tid / TPP);
    const int tpos = tid - part * TPP;
    int s_pix[PPT];                                           // pixel offset, -1 = zero padding, -2 = no slot
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int pos = tpos + TPP * k;
        const int sr = pos / IW, sc = pos - sr * IW;
        const int gy = STRIDE * y0 - HALO + sr, gx = STRIDE * x0 - HALO + sc;
        s_pix[k] = pos >= IN_CH ? -2 : ((gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) ? gy * a.Win + gx : -1);
    }
    constexpr int W_PT = (W_UNITS + NTHREADS - 1) / NTHREADS;
    const int nct32 = a.CoutPad / 32;
    const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.w);
    int w_off[W_PT];                                          // per-thread part of the weight unit index
#pragma unroll
    for (int j = 0; j < W_PT; ++j) {
        const int u = min(tid + NTHREADS * j, W_UNITS - 1);
        const int piece = u / (MTB * 128), within = u - piece * (MTB * 128);   // piece = (kstep, tap)
        w_off[j] = piece * nct32 * 128 + within;
    }
    const size_t w_chunk_units = (size_t)TAPS * nct32 * 128;                   // units per 16-channel k-step
    const size_t w_ct = (size_t)ct * MTB * 128;

    float in_regA[PPT * GPT * 8];
    u32x4 w_regA[W_PT];

    // Per-group scalar base pointers of the NEXT chunk to be loaded.  They are computed one phase ahead of their use
    // (right after the previous TCS_LOAD_CHUNK), so the scalar kernarg loads behind conv_src_ptr() land during the MFMA
    // phase instead of stalling the load issue (measured: 2.1k of 5.7k cycles per chunk before this).
    gptr_t nbase[GPT];
#define TCS_GROUP_BASES(C0)                                                                                 \
    _Pragma("unroll") for (int gi = 0; gi < GPT; ++gi)                                                      \
        nbase[gi] = conv_src_ptr(a, b, min((C0) + (part * GPT + gi) * 8, max(a.Cin - 8, 0)), HWi);
#define TCS_LOAD_CHUNK(in_reg, w_reg, C0)                                                                   \
    {                                                                                                       \
        /* weights first: their address registers alias last chunk's weight registers, and hipcc guards that  \
           reuse with counted vmcnt waits; issued first, those waits find nothing outstanding */              \
        const u32x4* wchunk = wsrc + (size_t)((C0) / 16) * w_chunk_units + w_ct;      /* scalar */          \
        _Pragma("unroll") for (int j = 0; j < W_PT; ++j)                                                    \
            w_reg[j] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(                   \
                (gbytes_t)(const char*)wchunk + (unsigned)w_off[j] * 16u);                                \
        _Pragma("unroll") for (int gi = 0; gi < GPT; ++gi) {                                                \
            const int g0 = (C0) + (part * GPT + gi) * 8;                  /* wave-uniform */                \
            if (a.src_align8) {                                                                             \
                /* one scalar base pointer per group; loads are unconditional on clamped (valid) addresses and \
                   zero-selected in TCS_STORE_CHUNK, so nothing waits for them before the MFMA phase */      \
                gptr_t bj = nbase[gi];                                    /* scalar; +HWi per channel */    \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                             \
                    _Pragma("unroll") for (int k = 0; k < PPT; ++k)                                         \
                        in_reg[(k * GPT + gi) * 8 + j] = bj[max(s_pix[k], 0)];                              \
                    bj += HWi;                                                                              \
                }                                                                                           \
            } else {                                                                                        \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                             \
                    gptr_t base = conv_src_ptr(a, b, min(g0 + j, a.Cin - 1), HWi);                          \
                    _Pragma("unroll") for (int k = 0; k < PPT; ++k)                                         \
                        in_reg[(k * GPT + gi) * 8 + j] = base[max(s_pix[k], 0)];                            \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
    }
#define TCS_STORE_CHUNK(in_reg, w_reg, C0)                                                                                 \
    {                                                                                                       \
        _Pragma("unroll") for (int k = 0; k < PPT; ++k) {                                                   \
            if (s_pix[k] > -2) {                                                                            \
                const bool pix_ok = s_pix[k] >= 0;                                                          \
                _Pragma("unroll") for (int gi = 0; gi < GPT; ++gi) {                                        \
                    const int grp = part * GPT + gi;                                                        \
                    const int gq = (C0) + grp * 8;                                                          \
                    half8 hi8, lo8;                                                                         \
                    _Pragma("unroll") for (int j = 0; j < 8; j += 2) {                                      \
                        half2_t h2, l2;                                                                     \
                        const float x0_ = (pix_ok && gq + j < a.Cin) ? in_reg[(k * GPT + gi) * 8 + j] : 0.f;        \
                        const float x1_ = (pix_ok && gq + j + 1 < a.Cin) ? in_reg[(k * GPT + gi) * 8 + j + 1] : 0.f; \
                        split_f16x2(x0_, x1_, h2, l2);                                                      \
                        hi8[j] = h2[0]; hi8[j + 1] = h2[1]; lo8[j] = l2[0]; lo8[j + 1] = l2[1];             \
                    }                                                                                       \
                    const size_t unit = (size_t)grp * IN_CH + tpos + TPP * k;                               \
                    *reinterpret_cast<half8*>(s_in_hi + unit * 16) = hi8;                                   \
                    *reinterpret_cast<half8*>(s_in_lo + unit * 16) = lo8;                                   \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < W_PT; ++j) {                                                  \
            const int u = tid + NTHREADS * j;                                                               \
            if (u < W_UNITS) *reinterpret_cast<u32x4*>(s_w + (size_t)u * 16) = w_reg[j];                    \
        }                                                                                                   \
    }

    // operand fetch of one (k-step, tap): MP input fragments + MT weight fragments, hi and lo, one ds_read_b128 each
    struct Frag { half8 b_hi[MP], b_lo[MP], a_hi[MT], a_lo[MT]; };
#define TCS_FETCH(F, KSI, T)                                                                                \
    {                                                                                                       \
        const int dy_ = (T) / KS, dx_ = (T) % KS;                                                           \
        _Pragma("unroll") for (int p = 0; p < MP; ++p) {                                                    \
            const size_t boff = ((size_t)(2 * (KSI) + half) * IN_CH + (STRIDE * (wave * MP + p) + dy_) * IW + dx_ + STRIDE * l31) * 16; \
            F.b_hi[p] = *reinterpret_cast<const half8*>(s_in_hi + boff);                                    \
            F.b_lo[p] = *reinterpret_cast<const half8*>(s_in_lo + boff);                                    \
        }                                                                                                   \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                    \
            const unsigned char* wt = s_w + ((size_t)(((KSI) * TAPS + (T)) * MTB + wave_g * MT + m) * 128 + lane) * 16; \
            F.a_hi[m] = *reinterpret_cast<const half8*>(wt);                                                \
            F.a_lo[m] = *reinterpret_cast<const half8*>(wt + 1024);                                         \
        }                                                                                                   \
    }
#define TCS_MMA(F)                                                                                          \
    _Pragma("unroll") for (int m = 0; m < MT; ++m)                                                          \
        _Pragma("unroll") for (int p = 0; p < MP; ++p) {                                                    \
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_lo[m], F.b_hi[p], acc[m][p], 0, 0, 0);   \
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_hi[m], F.b_lo[p], acc[m][p], 0, 0, 0);   \
            acc[m][p] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_hi[m], F.b_hi[p], acc[m][p], 0, 0, 0);   \
        }

    // Hand-pipelined operand fetch: inline-asm ds_read_b128 with counted s_waitcnt, so that the R = 2*MP + 2*MT reads of
    // step i+1 are in flight during the 3*MT*MP MFMAs of step i (hipcc puts compiler-visible ds_reads right in front of
    // their consumer with lgkmcnt(0)).  Measured: MFMA phase 1,450 -> 925 cycles per chunk for one block per CU, and the
    // narrow tiles (MT = 1) gain 5-12 %; the MT = 2 tiles are 5-8 % faster with the compiler's own schedule, so the
    // choice follows MT.  This kernel has no static __shared__, so the dynamic segment starts at
    // __builtin_amdgcn_groupstaticsize().
    const unsigned lds_base = __builtin_amdgcn_groupstaticsize();
    const unsigned addr_b = lds_base + (unsigned)((half * IN_CH + wave * MP * STRIDE * IW + STRIDE * l31) * 16);
    const unsigned addr_a = lds_base + (unsigned)(2 * IN_BYTES + (wave_g * MT * 128 + lane) * 16);
    // the immediate offset field is 16 bits; the (compile-time) part above 32 KB goes into the address register
#define TCS_DSREAD(DST, ADDR, OFF) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"((ADDR) + (unsigned)((OFF) & ~0x7fff)), "i"((OFF) & 0x7fff) : "memory")
#define TCS_FETCH_ASM(F, KSI, T)                                                                            \
    {                                                                                                       \
        _Pragma("unroll") for (int p = 0; p < MP; ++p) {                                                    \
            TCS_DSREAD(F.b_hi[p], addr_b, ((2 * (KSI)) * IN_CH + (STRIDE * p + (T) / KS) * IW + (T) % KS) * 16);            \
            TCS_DSREAD(F.b_lo[p], addr_b, ((2 * (KSI)) * IN_CH + (STRIDE * p + (T) / KS) * IW + (T) % KS) * 16 + IN_BYTES); \
        }                                                                                                   \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                    \
            TCS_DSREAD(F.a_hi[m], addr_a, (((KSI) * TAPS + (T)) * MTB + m) * 2048);                         \
            TCS_DSREAD(F.a_lo[m], addr_a, (((KSI) * TAPS + (T)) * MTB + m) * 2048 + 1024);                  \
        }                                                                                                   \
    }
#define TCS_WAIT_LGKM(N) { asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
#define TCS_COMPUTE_ASM()                                                                                       \
    {                                                                                                       \
        constexpr int NSTEP = KSTEPS * TAPS, R = 2 * MP + 2 * MT;                                           \
        static_assert(R <= 15, "lgkmcnt field");                                                            \
        Frag f0, f1;                                                                                        \
        TCS_FETCH_ASM(f0, 0, 0)                                                                             \
        _Pragma("unroll") for (int i = 0; i < NSTEP; i += 2) {                                              \
            if (i + 1 < NSTEP) { TCS_FETCH_ASM(f1, (i + 1) / TAPS, (i + 1) % TAPS) TCS_WAIT_LGKM(R) } else TCS_WAIT_LGKM(0) \
            TCS_MMA(f0)                                                                                     \
            if (i + 1 < NSTEP) {                                                                            \
                if (i + 2 < NSTEP) { TCS_FETCH_ASM(f0, (i + 2) / TAPS, (i + 2) % TAPS) TCS_WAIT_LGKM(R) } else TCS_WAIT_LGKM(0) \
                TCS_MMA(f1)                                                                                 \
            }                                                                                               \
        }                                                                                                   \
    }
    // compute on the chunk that is in LDS; software pipeline over the KSTEPS*TAPS steps: the ds_reads of step i+1 are
    // in flight during the MFMAs of step i
#define TCS_COMPUTE_C()                                                                                       \
    {                                                                                                       \
        constexpr int NSTEP = KSTEPS * TAPS;                                                                \
        Frag f0, f1;                                                                                        \
        TCS_FETCH(f0, 0, 0)                                                                                 \
        _Pragma("unroll") for (int i = 0; i < NSTEP; i += 2) {                                              \
            if (i + 1 < NSTEP) TCS_FETCH(f1, (i + 1) / TAPS, (i + 1) % TAPS)                                \
            TCS_MMA(f0)                                                                                     \
            if (i + 1 < NSTEP) {                                                                            \
                if (i + 2 < NSTEP) TCS_FETCH(f0, (i + 2) / TAPS, (i + 2) % TAPS)                            \
                TCS_MMA(f1)                                                                                 \
            }                                                                                               \
        }                                                                                                   \
    }

#define TCS_COMPUTE() { if constexpr (MT == 1) { TCS_COMPUTE_ASM() } else { TCS_COMPUTE_C() } }

    const int nchunks = (a.Cin + KC - 1) / KC;
    TCS_GROUP_BASES(0)
    TCS_LOAD_CHUNK(in_regA, w_regA, 0)
    TCS_GROUP_BASES(KC)
    TCS_STORE_CHUNK(in_regA, w_regA, 0)
    __syncthreads();
    {
        // global loads of chunk i+1 are in flight during the MFMAs of chunk i
#ifdef TCS_CONV_STAMPS
        unsigned long long ts0, ts1, ts2, ts3, ts4, ts5, acc_t[5] = {0, 0, 0, 0, 0};
#endif
        for (int i = 0; i < nchunks; ++i) {
            const bool has_next = i + 1 < nchunks;
#ifdef TCS_CONV_STAMPS
            TCS_STAMP(ts0)
#endif
            // TCS_ABLATE_* are diagnostic builds (tools/conv_ablate.sh): results are wrong, only the timing is of interest
#ifndef TCS_ABLATE_LOAD
            if (has_next) {
                TCS_LOAD_CHUNK(in_regA, w_regA, (i + 1) * KC)
                TCS_GROUP_BASES((i + 2) * KC)
            }
#endif
#ifdef TCS_CONV_STAMPS
            TCS_STAMP(ts1)
#endif
#ifndef TCS_ABLATE_MMA
            TCS_COMPUTE()
#endif
#ifdef TCS_CONV_STAMPS
            TCS_STAMP(ts2)
#endif
            if (has_next) {
                __syncthreads();
#ifdef TCS_CONV_STAMPS
                TCS_STAMP(ts3)
#endif
#ifndef TCS_ABLATE_STORE
                TCS_STORE_CHUNK(in_regA, w_regA, (i + 1) * KC)
#endif
#ifdef TCS_CONV_STAMPS
                TCS_STAMP(ts4)
#endif
                __syncthreads();
#ifdef TCS_CONV_STAMPS
                TCS_STAMP(ts5)
                acc_t[0] += ts1 - ts0; acc_t[1] += ts2 - ts1; acc_t[2] += ts3 - ts2; acc_t[3] += ts4 - ts3; acc_t[4] += ts5 - ts4;
#endif
            }
        }
#ifdef TCS_CONV_STAMPS
        if (lane == 0) {
            const size_t w_ = ((size_t)blockIdx.x * ROWS + wave_all) % 16384;
            for (int q = 0; q < 5; ++q) tcs_conv_stamps[w_ * 8 + q] = acc_t[q];
            tcs_conv_stamps[w_ * 8 + 5] = nchunks;
        }
#endif
    }
#undef TCS_COMPUTE
#undef TCS_COMPUTE_ASM
#undef TCS_COMPUTE_C
#undef TCS_FETCH_ASM
#undef TCS_DSREAD
#undef TCS_WAIT_LGKM
#undef TCS_LOAD_CHUNK
#undef TCS_GROUP_BASES
#undef TCS_STORE_CHUNK
#undef TCS_FETCH
#undef TCS_MMA

    const int px = x0 + l31;
    if (px >= W) return;
#pragma unroll
    for (int p = 0; p < MP; ++p) {
        const int py = y0 + wave * MP + p;
        if (py >= H) continue;
        const size_t pix = (size_t)py * W + px;
#pragma unroll
        for (int m = 0; m < MT; ++m)
            conv_epilogue_tile<EPI>(a, b, ct * NT + (wave_g * MT + m) * 32 + 4 * half, pix, HW, acc[m][p], a.w_unscale);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Wave-specialised variant: 4 consumer waves (MFMA only) + NP producer waves (global loads, fp16 split, LDS stores) per
// block, two LDS buffers, ONE barrier per chunk.  In k_conv_f16x3 every wave does all of a chunk's work in sequence
// (issue loads -> MFMAs -> wait loads -> split + store -> barriers), ~3,000 cycles per chunk of which 864*MT are MFMA
// (in-kernel stamps and phase-ablated builds, tools/conv_phases.py / tools/conv_ablate.sh): co-resident blocks run those
// phases in lock step, so the phases add instead of overlapping.  Here the staging of chunk i+1 runs on other waves of
// the same SIMDs while chunk i is multiplied, and a chunk costs max(MFMA, staging) instead of their sum.
// Same tiling, LDS images and packed weights as k_conv_f16x3<KS, MT, 1, KSTEPS, EPI>; stride 1 only.
template <int KS, int MT, int KSTEPS, int EPI, int NP>
__global__ __launch_bounds__(256 + 64 * NP) void k_conv_f16x3_ws(ConvArgs a) {
    constexpr int HALO = KS / 2, IH = 4 + KS - 1, IW = 32 + KS - 1, TAPS = KS * KS, IN_CH = IH * IW;
    constexpr int NT = 32 * MT, KC = 16 * KSTEPS, NG = 2 * KSTEPS;
    constexpr int IN_BYTES = NG * IN_CH * 16;                                  // one of {hi, lo}
    constexpr int W_UNITS = KSTEPS * TAPS * MT * 2 * 64;                       // 16-byte units per chunk
    constexpr int BUF_BYTES = 2 * IN_BYTES + W_UNITS * 16;
    constexpr int NPT = 64 * NP;                                               // producer threads
    extern __shared__ __attribute__((aligned(16))) unsigned char lds8[];

    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int ct = bid % a.nct, patch = bid / a.nct;
    const int b = blockIdx.y;
    const int y0 = (patch / a.npx) * 4, x0 = (patch % a.npx) * 32;
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W;
    const size_t HWi = (size_t)a.Hin * a.Win;
    const int nchunks = (a.Cin + KC - 1) / KC;

    if (wave_all >= 4) {
        // ---------------- producers ----------------
        const int ptid = tid - 256;
        constexpr int PARTS = NPT / ((IN_CH <= 128) ? 128 : 256);
        static_assert(PARTS >= 1 && NG % PARTS == 0, "producer threads must split the channel groups evenly");
        constexpr int TPP = NPT / PARTS, PPT = (IN_CH + TPP - 1) / TPP, GPT = NG / PARTS;
        const int part = PARTS == 1 ? 0 : __builtin_amdgcn_readfirstlane(ptid / TPP);
        const int tpos = ptid - part * TPP;
        int s_pix[PPT];                                           // pixel offset, -1 = zero padding, -2 = no slot
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int pos = tpos + TPP * k;
            const int sr = pos / IW, sc = pos - sr * IW;
            const int gy = y0 - HALO + sr, gx = x0 - HALO + sc;
            s_pix[k] = pos >= IN_CH ? -2 : ((gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win) ? gy * a.Win + gx : -1);
        }
        constexpr int W_PT = (W_UNITS + NPT - 1) / NPT;
        const int nct32 = a.CoutPad / 32;
        const u32x4* wsrc = reinterpret_cast<const u32x4*>(a.w);
        int w_off[W_PT];
#pragma unroll
        for (int j = 0; j < W_PT; ++j) {
            const int u = min(ptid + NPT * j, W_UNITS - 1);
            const int piece = u / (MT * 128), within = u - piece * (MT * 128);
            w_off[j] = piece * nct32 * 128 + within;
        }
        const size_t w_chunk_units = (size_t)TAPS * nct32 * 128;
        const size_t w_ct = (size_t)ct * MT * 128;
        float in_reg[PPT * GPT * 8];
        u32x4 w_reg[W_PT];
        gptr_t nbase[GPT];
#define WS_GROUP_BASES(C0)                                                                                  \
    _Pragma("unroll") for (int gi = 0; gi < GPT; ++gi)                                                      \
        nbase[gi] = conv_src_ptr(a, b, min((C0) + (part * GPT + gi) * 8, max(a.Cin - 8, 0)), HWi);
#define WS_LOAD(C0)                                                                                         \
    {                                                                                                       \
        const u32x4* wchunk = wsrc + (size_t)((C0) / 16) * w_chunk_units + w_ct;                            \
        _Pragma("unroll") for (int j = 0; j < W_PT; ++j)                                                    \
            w_reg[j] = *reinterpret_cast<const __attribute__((address_space(1))) u32x4*>(                   \
                (gbytes_t)(const char*)wchunk + (unsigned)w_off[j] * 16u);                                  \
        _Pragma("unroll") for (int gi = 0; gi < GPT; ++gi) {                                                \
            const int g0 = (C0) + (part * GPT + gi) * 8;                                                    \
            if (a.src_align8) {                                                                             \
                gptr_t bj = nbase[gi];                                                                      \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                             \
                    _Pragma("unroll") for (int k = 0; k < PPT; ++k)                                         \
                        in_reg[(k * GPT + gi) * 8 + j] = bj[max(s_pix[k], 0)];                              \
                    bj += HWi;                                                                              \
                }                                                                                           \
            } else {                                                                                        \
                _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                             \
                    gptr_t base = conv_src_ptr(a, b, min(g0 + j, a.Cin - 1), HWi);                          \
                    _Pragma("unroll") for (int k = 0; k < PPT; ++k)                                         \
                        in_reg[(k * GPT + gi) * 8 + j] = base[max(s_pix[k], 0)];                            \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
    }
#define WS_STORE(BUF, C0)                                                                                   \
    {                                                                                                       \
        unsigned char* s_in_hi = lds8 + (BUF) * BUF_BYTES;                                                  \
        unsigned char* s_in_lo = s_in_hi + IN_BYTES;                                                        \
        unsigned char* s_w = s_in_hi + 2 * IN_BYTES;                                                        \
        _Pragma("unroll") for (int k = 0; k < PPT; ++k) {                                                   \
            if (s_pix[k] > -2) {                                                                            \
                const bool pix_ok = s_pix[k] >= 0;                                                          \
                _Pragma("unroll") for (int gi = 0; gi < GPT; ++gi) {                                        \
                    const int grp = part * GPT + gi;                                                        \
                    const int gq = (C0) + grp * 8;                                                          \
                    half8 hi8, lo8;                                                                         \
                    _Pragma("unroll") for (int j = 0; j < 8; j += 2) {                                      \
                        half2_t h2, l2;                                                                     \
                        const float x0_ = (pix_ok && gq + j < a.Cin) ? in_reg[(k * GPT + gi) * 8 + j] : 0.f;        \
                        const float x1_ = (pix_ok && gq + j + 1 < a.Cin) ? in_reg[(k * GPT + gi) * 8 + j + 1] : 0.f; \
                        split_f16x2(x0_, x1_, h2, l2);                                                      \
                        hi8[j] = h2[0]; hi8[j + 1] = h2[1]; lo8[j] = l2[0]; lo8[j + 1] = l2[1];             \
                    }                                                                                       \
                    const size_t unit = (size_t)grp * IN_CH + tpos + TPP * k;                               \
                    *reinterpret_cast<half8*>(s_in_hi + unit * 16) = hi8;                                   \
                    *reinterpret_cast<half8*>(s_in_lo + unit * 16) = lo8;                                   \
                }                                                                                           \
            }                                                                                               \
        }                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < W_PT; ++j) {                                                  \
            const int u = ptid + NPT * j;                                                                   \
            if (u < W_UNITS) *reinterpret_cast<u32x4*>(s_w + (size_t)u * 16) = w_reg[j];                    \
        }                                                                                                   \
    }
        WS_GROUP_BASES(0)
        WS_LOAD(0)
        WS_GROUP_BASES(KC)
        WS_STORE(0, 0)
        if (nchunks > 1) {
            WS_LOAD(KC)
            WS_GROUP_BASES(2 * KC)
        }
        __syncthreads();                                          // chunk 0 is in buffer 0
        for (int i = 0; i < nchunks; ++i) {
            if (i + 1 < nchunks) {
                WS_STORE((i + 1) & 1, (i + 1) * KC)               // buffer (i+1)&1 was last read in iteration i-1
                if (i + 2 < nchunks) {
                    WS_LOAD((i + 2) * KC)
                    WS_GROUP_BASES((i + 3) * KC)
                }
            }
            __syncthreads();
        }
#undef WS_GROUP_BASES
#undef WS_LOAD
#undef WS_STORE
        return;
    }

    // ---------------- consumers: wave = patch row ----------------
    const int wave = wave_all;
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
    // operand fetch: inline-asm ds_read_b128 with counted waits (see k_conv_f16x3); no static __shared__ in this kernel
    const unsigned lds_base = __builtin_amdgcn_groupstaticsize();
    const unsigned addr_b0 = lds_base + (unsigned)((half * IN_CH + wave * IW + l31) * 16);
    const unsigned addr_a0 = lds_base + (unsigned)(2 * IN_BYTES + lane * 16);
    struct Frag { half8 b_hi, b_lo, a_hi[MT], a_lo[MT]; };
#define WS_DSREAD(DST, ADDR, OFF) \
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"((ADDR) + (unsigned)((OFF) & ~0x7fff)), "i"((OFF) & 0x7fff) : "memory")
#define WS_FETCH(F, KSI, T)                                                                                 \
    {                                                                                                       \
        WS_DSREAD(F.b_hi, addr_b, ((2 * (KSI)) * IN_CH + ((T) / KS) * IW + (T) % KS) * 16);                 \
        WS_DSREAD(F.b_lo, addr_b, ((2 * (KSI)) * IN_CH + ((T) / KS) * IW + (T) % KS) * 16 + IN_BYTES);      \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                    \
            WS_DSREAD(F.a_hi[m], addr_a, (((KSI) * TAPS + (T)) * MT + m) * 2048);                           \
            WS_DSREAD(F.a_lo[m], addr_a, (((KSI) * TAPS + (T)) * MT + m) * 2048 + 1024);                    \
        }                                                                                                   \
    }
#define WS_WAIT(N) { asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
#define WS_MMA(F)                                                                                           \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                        \
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_lo[m], F.b_hi, acc[m], 0, 0, 0);                \
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_hi[m], F.b_lo, acc[m], 0, 0, 0);                \
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_hi[m], F.b_hi, acc[m], 0, 0, 0);                \
    }
    __syncthreads();                                              // chunk 0 is in buffer 0
    for (int i = 0; i < nchunks; ++i) {
        const unsigned boff = (i & 1) ? (unsigned)BUF_BYTES : 0u;
        const unsigned addr_b = addr_b0 + boff, addr_a = addr_a0 + boff;
        constexpr int NSTEP = KSTEPS * TAPS, R = 2 + 2 * MT;
        Frag f0, f1;
        WS_FETCH(f0, 0, 0)
#pragma unroll
        for (int st = 0; st < NSTEP; st += 2) {
            if (st + 1 < NSTEP) { WS_FETCH(f1, (st + 1) / TAPS, (st + 1) % TAPS) WS_WAIT(R) } else WS_WAIT(0)
            WS_MMA(f0)
            if (st + 1 < NSTEP) {
                if (st + 2 < NSTEP) { WS_FETCH(f0, (st + 2) / TAPS, (st + 2) % TAPS) WS_WAIT(R) } else WS_WAIT(0)
                WS_MMA(f1)
            }
        }
        __syncthreads();
    }
#undef WS_DSREAD
#undef WS_FETCH
#undef WS_WAIT
#undef WS_MMA
    const int px = x0 + l31, py = y0 + wave;
    if (px >= W || py >= H) return;
    const size_t pix = (size_t)py * W + px;
#pragma unroll
    for (int m = 0; m < MT; ++m) conv_epilogue_tile<EPI>(a, b, ct * NT + m * 32 + 4 * half, pix, HW, acc[m], a.w_unscale);
}

// OIHW fp32 weights -> the LDS image order, split into (hi, lo) halves after scaling by 2^scale_log2.
// unit (16 B = 8 halves) index: ((((kchunk16 * TAPS + t) * nct32 + ct32) * 2 + part) * 64 + h*32 + r)
__global__ __launch_bounds__(256) void k_pack_weight_f16x3(const float* __restrict__ w, int Cout, int Cin, int taps, int nchunk16,
                                                           int nct32, float scale, uint4* __restrict__ packed) {
    const size_t n = (size_t)nchunk16 * taps * nct32 * 128;
    const size_t u = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= n) return;
    const int slot = (int)(u & 63), part = (int)((u >> 6) & 1);
    const size_t rest = u >> 7;
    const int ct32 = (int)(rest % nct32);
    const int t = (int)((rest / nct32) % taps);
    const int kc = (int)(rest / ((size_t)nct32 * taps));
    const int h = slot >> 5, r = slot & 31;
    const int co = ct32 * 32 + r;
    half8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int ci = kc * 16 + 8 * h + j;
        const float x = (co < Cout && ci < Cin) ? w[((size_t)co * Cin + ci) * taps + t] * scale : 0.f;
        _Float16 hi, lo;
        split_f16(x, hi, lo);
        v[j] = part ? lo : hi;
    }
    packed[u] = *reinterpret_cast<uint4*>(&v);
}

template <int KS, int MT, int MP, int KSTEPS, int EPI, int STRIDE = 1, int ROWS = 4>
static int launch_f16(ConvArgs& a, hipStream_t s) {
    constexpr int IH = STRIDE * ROWS * MP + KS - STRIDE, IW = STRIDE * 32 + KS - STRIDE, TAPS = KS * KS;
    const size_t lds = (size_t)2 * (2 * KSTEPS) * IH * IW * 16 + (size_t)KSTEPS * TAPS * MT * 2 * 1024;
    auto kern = k_conv_f16x3<KS, MT, MP, KSTEPS, EPI, STRIDE, ROWS>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return TCS_ELAUNCH;
    }
    a.npatch = a.npx * tcs_cdiv(a.H, ROWS * MP);
    a.nct = (a.CoutPad / 32) / MT;
    hipLaunchKernelGGL(kern, dim3(a.npatch * a.nct, a.B), dim3(64 * ROWS), lds, s, a);
    return tcs_launch_status();
}

template <int KS, int MT, int KSTEPS, int EPI, int NP>
static int launch_f16_ws(ConvArgs& a, hipStream_t s) {
    constexpr int IH = 4 + KS - 1, IW = 32 + KS - 1, TAPS = KS * KS;
    const size_t lds = 2 * ((size_t)2 * (2 * KSTEPS) * IH * IW * 16 + (size_t)KSTEPS * TAPS * MT * 2 * 1024);
    auto kern = k_conv_f16x3_ws<KS, MT, KSTEPS, EPI, NP>;
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return TCS_ELAUNCH;
    }
    a.npatch = a.npx * tcs_cdiv(a.H, 4);
    a.nct = (a.CoutPad / 32) / MT;
    hipLaunchKernelGGL(kern, dim3(a.npatch * a.nct, a.B), dim3(256 + 64 * NP), lds, s, a);
    return tcs_launch_status();
}

template <int KS, int EPI>
static int launch_f16_tile(ConvArgs& a, hipStream_t s) {
    // Tile choice by grid size only (no environment knobs: the library keeps no mutable global state).
    const int nct32 = a.CoutPad / 32;
    const long long px_tiles = (long long)a.npx * tcs_cdiv(a.H, 4) * a.B;           // 128-pixel patches
    // widest tile (most register reuse of LDS operands) that still leaves >= 2 blocks per CU
    int mt = (nct32 % 2 == 0 && px_tiles * (nct32 / 2) >= 512) ? 2 : 1;
    if (KS == 3) {
        // 5-row patches when they balance the grid better: efficiency = workgroups / (256 * rounds of one-per-CU).
        // Measured (tools/bench_conv.py): 32-channel tiles gain 4-8 % (gru08.q 83 -> 77 us, 128->128 32.3 -> 30.9 us), the
        // 64-channel tiles lose 20 % (gru08.zr 135 -> 163 us), so this is limited to MT = 1.
        const long long wg4 = (long long)a.npx * tcs_cdiv(a.H, 4) * a.B * (nct32 / mt), wg5 = (long long)a.npx * tcs_cdiv(a.H, 5) * a.B * (nct32 / mt);
        const double e4 = (double)wg4 / (256.0 * ((wg4 + 255) / 256)), e5 = (double)wg5 / (256.0 * ((wg5 + 255) / 256));
        if (mt == 1 && e5 > e4 + 0.05 && wg4 > 400 && wg4 <= 768) return launch_f16<3, 1, 1, 1, EPI, 1, 5>(a, s);
        // Wave-specialised kernel (k_conv_f16x3_ws) for the layers that cannot fill the chip: with <= 1-2 blocks per CU the
        // per-chunk latency chain of the plain kernel is exposed, and overlapping staging with the MFMAs gains 10-18 %
        // (tools/bench_conv.py: 128->128 at 1/32 scale 15.9 -> 13.5 us, gru16.zr 57.6 -> 50.9 us); on full grids it ties or loses.
        const long long blocks_mt1 = px_tiles * nct32;
        if (blocks_mt1 <= 200) return launch_f16_ws<3, 1, 1, EPI, 8>(a, s);
        if (blocks_mt1 <= 400) return launch_f16_ws<3, 1, 1, EPI, 4>(a, s);
        return mt == 2 ? launch_f16<3, 2, 1, 1, EPI>(a, s) : launch_f16<3, 1, 1, 1, EPI>(a, s);
    }
    return mt == 2 ? launch_f16<1, 2, 1, 4, EPI>(a, s) : launch_f16<1, 1, 1, 4, EPI>(a, s);
}

template <int EPI>
static int launch_f16_ks(ConvArgs& a, int ksize, hipStream_t s) {
    if (ksize == 3) return launch_f16_tile<3, EPI>(a, s);
    if (ksize == 1) return launch_f16_tile<1, EPI>(a, s);
    return TCS_EUNSUPPORTED;
}

int tcs_conv_f16x3_launch(ConvArgs& a, int ksize, int epilogue, int stride, hipStream_t s) {
    a.src_align8 = (a.Cin % 8 == 0) ? 1 : 0;
    for (int i = 0; i < TCS_MAX_SRC; ++i)
        if (a.src_end[i] != 0x7fffffff && (a.src_end[i] % 8) != 0) a.src_align8 = 0;
    if (stride == 2) {                       // 3x3 stride-2 pad-1 (conv_4_8 / conv_8_16 of the U-Nets), linear epilogue
        if (ksize != 3 || epilogue != TCS_EPI_LINEAR) return TCS_EUNSUPPORTED;
        return launch_f16<3, 1, 1, 1, TCS_EPI_LINEAR, 2>(a, s);
    }
    if (epilogue == TCS_EPI_DECONV2X) {
        if (ksize != 3) return TCS_EUNSUPPORTED;
        return launch_f16<3, 1, 1, 1, TCS_EPI_DECONV2X, 1>(a, s);
    }
    switch (epilogue) {
        case TCS_EPI_LINEAR: return launch_f16_ks<TCS_EPI_LINEAR>(a, ksize, s);
        case TCS_EPI_GRU_ZR: return launch_f16_ks<TCS_EPI_GRU_ZR>(a, ksize, s);
        case TCS_EPI_GRU_Q: return launch_f16_ks<TCS_EPI_GRU_Q>(a, ksize, s);
        default: return TCS_EINVAL;
    }
}

extern "C" size_t tcs_conv_packed_floats_f16x3(int Cout, int Cin, int ksize);
extern "C" int tcs_pack_conv_weight_f16x3(const float* w_oihw, int Cout, int Cin, int ksize, int scale_log2, float* packed,
                                          tcs_stream_t stream);

// ConvTranspose2d(4,2,1) weights [Cin][Cout][4][4] -> equivalent 3x3 conv weights [4*Cout][Cin][3][3]:
// output parity (py,px), tap offset (dy,dx) in {-1,0,1}: ky = py ? (dy == 1 ? 0 : 2) : (dy == 0 ? 1 : 3), valid when
// (py == 0 and dy in {0,-1}) or (py == 1 and dy in {+1,0}); same along x.
__global__ __launch_bounds__(256) void k_deconv_to_conv(const float* __restrict__ wt, int Cin, int Cout, float* __restrict__ w3) {
    const size_t n = (size_t)4 * Cout * Cin * 9;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int tap = (int)(i % 9), ci = (int)((i / 9) % Cin);
    const int oc = (int)(i / ((size_t)9 * Cin));
    const int par = oc / Cout, co = oc - par * Cout, py = par >> 1, px = par & 1;
    const int dy = tap / 3 - 1, dx = tap % 3 - 1;
    int ky = -1, kx = -1;
    if (py == 0) ky = dy == 0 ? 1 : (dy == -1 ? 3 : -1); else ky = dy == 1 ? 0 : (dy == 0 ? 2 : -1);
    if (px == 0) kx = dx == 0 ? 1 : (dx == -1 ? 3 : -1); else kx = dx == 1 ? 0 : (dx == 0 ? 2 : -1);
    w3[i] = (ky >= 0 && kx >= 0) ? wt[(((size_t)ci * Cout + co) * 4 + ky) * 4 + kx] : 0.f;
}

// InstanceNorm2d (affine=False): one 1024-thread block per (b, c) plane.  The plane is read ONCE into registers
// (up to 32 values per thread = 32 768 pixels; larger planes fall back to re-reading), mean and variance are
// two block reductions (variance around the mean, like the reference's two-pass formula), then the
// normalised, activated (+ addend) values are written.
#define IN_T 1024
#define IN_VPT 32
__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < IN_T / 64; ++i) s += red[i];
    return s;
}

__global__ __launch_bounds__(IN_T) void k_instance_norm(const float* __restrict__ x, int HW, float eps, int act,
                                                         const float* __restrict__ addend, float* __restrict__ out) {
    __shared__ float red[IN_T / 64];
    const float* p = x + (size_t)blockIdx.x * HW;
    const float* ad = addend ? addend + (size_t)blockIdx.x * HW : nullptr;
    float* o = out + (size_t)blockIdx.x * HW;
    const bool cached = HW <= IN_T * IN_VPT;
    float v[IN_VPT];
    float s = 0.f;
    if (cached) {
#pragma unroll
        for (int k = 0; k < IN_VPT; ++k) {
            const int i = threadIdx.x + IN_T * k;
            v[k] = i < HW ? p[i] : 0.f;
            s += v[k];
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += IN_T) s += p[i];
    }
    const float mean = block_sum_1024(s, red) / (float)HW;
    float q = 0.f;
    if (cached) {
#pragma unroll
        for (int k = 0; k < IN_VPT; ++k) {
            const int i = threadIdx.x + IN_T * k;
            const float d = i < HW ? v[k] - mean : 0.f;
            q = fmaf(d, d, q);
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += IN_T) { const float d = p[i] - mean; q = fmaf(d, d, q); }
    }
    const float rstd = 1.0f / sqrtf(block_sum_1024(q, red) / (float)HW + eps);
    if (cached) {
#pragma unroll
        for (int k = 0; k < IN_VPT; ++k) {
            const int i = threadIdx.x + IN_T * k;
            if (i < HW) {
                const float y = apply_act((v[k] - mean) * rstd, act);
                o[i] = ad ? y + ad[i] : y;
            }
        }
    } else {
        for (int i = threadIdx.x; i < HW; i += IN_T) {
            const float y = apply_act((p[i] - mean) * rstd, act);
            o[i] = ad ? y + ad[i] : y;
        }
    }
}

// 3x3 convolution with ONE output channel (FlowHead.conv2 256->1, update.py:13): a matrix-core tile would waste 31
// of 32 output columns, so this is a plain reduction.  A wave owns 62 consecutive output pixels of one row and loads 64
// columns (one halo column on each side): the left / right taps come from the neighbouring lanes (DPP row shifts), so a
// channel costs 3 loads per lane instead of 9 — the kernel was L1-bandwidth bound (177 MB through L1 for a 19.7 MB
// input).  The 8 waves of a block split the input channels and combine through LDS; weights are wave-uniform scalar
// loads; the channel loop is unrolled by 8 (24 independent loads in flight per lane).
#define C1_WAVES 8
#define C1_PX 62
__global__ __launch_bounds__(64 * C1_WAVES) void k_conv3x3_cout1(const float* __restrict__ x, const float* __restrict__ w /*[Cin][9]*/,
                                                                 const float* __restrict__ bias, int Cin, int H, int W,
                                                                 float* __restrict__ out) {
    __shared__ float part[C1_WAVES][64];
    const int b = blockIdx.z, HW = H * W;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int y = blockIdx.y;
    const int col = blockIdx.x * C1_PX + lane - 1;             // input column of this lane (lane 0 / 63: halo)
    const bool col_ok = col >= 0 && col < W;
    const int cc = min(max(col, 0), W - 1);
    int off[3];
    bool row_ok[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = y + r - 1;
        row_ok[r] = yy >= 0 && yy < H;
        off[r] = min(max(yy, 0), H - 1) * W + cc;
    }
    const int cpw = (Cin + C1_WAVES - 1) / C1_WAVES;
    const int c_lo = wave * cpw, c_hi = min(Cin, c_lo + cpw);
    float acc = 0.f;
#define C1_ONE_CHANNEL(V0, V1, V2, WP)                                                                       \
    {                                                                                                        \
        const float v_[3] = {V0, V1, V2};                                                                    \
        _Pragma("unroll") for (int r = 0; r < 3; ++r) {                                                      \
            const float m_ = (row_ok[r] && col_ok) ? v_[r] : 0.f;         /* zero padding */                 \
            const float l_ = __shfl_up(m_, 1), rr_ = __shfl_down(m_, 1);  /* columns col-1, col+1 */         \
            acc = fmaf((WP)[3 * r], l_, acc);                                                                \
            acc = fmaf((WP)[3 * r + 1], m_, acc);                                                            \
            acc = fmaf((WP)[3 * r + 2], rr_, acc);                                                           \
        }                                                                                                    \
    }
    int c = c_lo;
    for (; c + 8 <= c_hi; c += 8) {
        float v[8][3];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float* xp = x + ((size_t)b * Cin + c + k) * HW;
#pragma unroll
            for (int r = 0; r < 3; ++r) v[k][r] = xp[off[r]];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) C1_ONE_CHANNEL(v[k][0], v[k][1], v[k][2], w + (size_t)(c + k) * 9)
    }
    for (; c < c_hi; ++c) {
        const float* xp = x + ((size_t)b * Cin + c) * HW;
        C1_ONE_CHANNEL(xp[off[0]], xp[off[1]], xp[off[2]], w + (size_t)c * 9)
    }
#undef C1_ONE_CHANNEL
    part[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && lane >= 1 && lane <= C1_PX && col < W) {
        float r = bias ? bias[0] : 0.f;
#pragma unroll
        for (int k = 0; k < C1_WAVES; ++k) r += part[k][lane];
        out[(size_t)b * HW + (size_t)y * W + col] = r;
    }
}

#ifdef TCS_CONV_STAMPS
extern "C" int tcs_debug_read_conv_stamps(unsigned long long* host_out, int n_waves) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(tcs_conv_stamps), (size_t)n_waves * 8 * sizeof(unsigned long long)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" {

size_t tcs_deconv_packed_floats_f16x3(int Cin, int Cout) { return tcs_conv_packed_floats_f16x3(4 * Cout, Cin, 3); }

int tcs_pack_deconv4x4s2_f16x3(const float* w_iohw, int Cin, int Cout, int scale_log2, float* packed, float* scratch_oihw,
                               tcs_stream_t stream) {
    if (!w_iohw || !packed || !scratch_oihw || Cin <= 0 || Cout <= 0) return TCS_EINVAL;
    const size_t n = (size_t)4 * Cout * Cin * 9;
    hipLaunchKernelGGL(k_deconv_to_conv, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tcs_stream(stream), w_iohw, Cin, Cout,
                       scratch_oihw);
    return tcs_pack_conv_weight_f16x3(scratch_oihw, 4 * Cout, Cin, 3, scale_log2, packed, stream);
}

int tcs_instance_norm(const float* x, int B, int C, int H, int W, float eps, int act, const float* addend, float* out,
                      tcs_stream_t stream) {
    if (!x || !out || B <= 0 || C <= 0 || H <= 0 || W <= 0 || eps < 0.f) return TCS_EINVAL;
    hipLaunchKernelGGL(k_instance_norm, dim3((unsigned)((size_t)B * C)), dim3(IN_T), 0, tcs_stream(stream), x, H * W, eps, act,
                       addend, out);
    return tcs_launch_status();
}

int tcs_conv3x3_cout1(const float* x, const float* w_oihw, const float* bias, int B, int Cin, int H, int W, float* out,
                      tcs_stream_t stream) {
    if (!x || !w_oihw || !out || B <= 0 || B > 65535 || Cin <= 0 || H <= 0 || W <= 0) return TCS_EINVAL;
    if (H > 65535) return TCS_EINVAL;
    hipLaunchKernelGGL(k_conv3x3_cout1, dim3(tcs_cdiv(W, C1_PX), H, B), dim3(64 * C1_WAVES), 0, tcs_stream(stream), x, w_oihw, bias,
                       Cin, H, W, out);
    return tcs_launch_status();
}

size_t tcs_conv_packed_floats_f16x3(int Cout, int Cin, int ksize) {
    if (Cout <= 0 || Cin <= 0 || (ksize != 1 && ksize != 3)) return 0;
    const size_t nchunk16 = (size_t)((Cin + 63) / 64) * 4, nct32 = (size_t)(Cout + 31) / 32;
    return nchunk16 * ksize * ksize * nct32 * 128 * 4;        // 16-byte units * 4 floats
}

int tcs_pack_conv_weight_f16x3(const float* w_oihw, int Cout, int Cin, int ksize, int scale_log2, float* packed,
                               tcs_stream_t stream) {
    const size_t nfl = tcs_conv_packed_floats_f16x3(Cout, Cin, ksize);
    if (!w_oihw || !packed || nfl == 0 || scale_log2 < -60 || scale_log2 > 60) return TCS_EINVAL;
    const int nchunk16 = ((Cin + 63) / 64) * 4, nct32 = (Cout + 31) / 32;
    const size_t n = nfl / 4;
    hipLaunchKernelGGL(k_pack_weight_f16x3, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tcs_stream(stream), w_oihw, Cout, Cin,
                       ksize * ksize, nchunk16, nct32, ldexpf(1.0f, scale_log2), reinterpret_cast<uint4*>(packed));
    return tcs_launch_status();
}

}  // extern "C"

// this translation unit's S16 domain flag (tcs_s16.h): read-and-clear for tcs_s16_flags()
int tcs_s16_flag_take_conv_f16(unsigned int* out) {
    unsigned int v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(TCS_S16_FLAG_VAR), sizeof(v)) != hipSuccess) return TCS_ELAUNCH;
    if (v && hipMemcpyToSymbol(HIP_SYMBOL(TCS_S16_FLAG_VAR), &zero, sizeof(zero)) != hipSuccess) return TCS_ELAUNCH;
    *out |= v;
    return TCS_OK;
}
