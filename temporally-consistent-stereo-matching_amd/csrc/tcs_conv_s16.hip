// Convolutions on PRE-SPLIT activations ("S16" tensors) for gfx950: the refinement loop's conv kernel.
//
// The fp16-split contraction of tcs_conv_f16.hip (x = hi + lo, three v_mfma_f32_32x32x16_f16 per product, fp32
// accumulate, error ~2^-21) spent most of its issue slots turning fp32 NCHW activations into LDS operand images: each
// element was loaded as a dword, clamped, converted twice and stored to LDS ~6 times (4 cout tiles x 1.6 halo overlap) —
// 8.5 non-MFMA instructions per MFMA (profiles/r01_conv_gru08zr_pmc.txt).  Here the PRODUCER splits once, in its
// epilogue, and activations live in HBM already in operand form:
//
//   S16 tensor of a logical [B][C][H][W] fp32 tensor:   _Float16 [B][G][2][H+2][W+2][8]
//       G = ceil(C/8) rounded up to even, [2] = {hi, lo} planes, a 16-byte unit = 8 consecutive channels of one pixel,
//       one-pixel ZERO border (written once at allocation, never by a producer), zero padding channels.
//       Same 4 bytes per element as fp32.
//
// A unit is exactly one lane's B-operand fragment of v_mfma_f32_32x32x16_f16 (k = 8 channels of lane-half h), so staging
// is a straight copy: LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave-instruction, no VGPR round trip, no VALU).  The
// zero border replaces every bounds test of a 3x3 'same' convolution; tiles that overhang the right / bottom edge read
// clamped (valid, finite) positions and never store those pixels (MFMA columns are independent).
//
// Tiling: block = ROWS waves, wave = one output row of 32 pixels x 32*MT output channels (D: pixel on the lane).
// K loop over stages of KSTEPS 16-channel k-steps x all taps; NSTAGE LDS buffers, all filled by the prologue:
//     wait own DMA pieces of stage s (counted vmcnt) -> s_barrier -> refill the buffer of stage s-1 with stage s+NSTAGE-1
//     -> MFMAs of stage s
// i.e. ONE barrier per stage and NSTAGE-1 stages of prefetch in flight across it.  Operand fetches are inline-asm
// ds_read_b128 with counted lgkmcnt (hipcc would put vmcnt(0) in front of every LDS read that may alias a DMA).
// Weights: the f16x3 packed image of tcs_pack_conv_weight_f16x3, unchanged.
//
// Epilogues (bias, fp32 addends, activation, GRU gate arithmetic, pixel shuffle of the transposed convs) run on the
// accumulators and write S16 (8-byte stores: 4 channels of one pixel, hi or lo) and/or fp32 NCHW.
#include "tcs_conv_common.h"
#include "tcs_s16.h"

typedef float float4_t __attribute__((ext_vector_type(4)));

struct S16Args {
    const _Float16* src[TCS_MAX_SRC];   // S16 sources (virtual concat along channels)
    int src_groups[TCS_MAX_SRC];        // groups allocated per source tensor (even)
    int src_kend[TCS_MAX_SRC];          // cumulative k-steps after each source (unused sources: INT_MAX)
    const float* w;                     // f16x3 packed weights
    const float* bias;
    int B, H, W;                        // output grid
    int Hin, Win;                       // input grid (all sources)
    int nk;                             // total 16-channel k-steps
    int Cout, nct32;                    // output channels, packed 32-channel tiles
    int act;
    float post_scale, w_unscale;
    const float* add1;                  // fp32 NCHW addends ([B][Cout or hidden][H][W])
    const float* add2;
    int add_ctot;                       // channels per batch element of the addend tensors (>= Cout / hidden): lets add1 / add2
                                        // be channel slices of one wider tensor (the partial sums of a K-split layer)
    const _Float16* add16;              // LINEAR: S16 addend (residual skip), add16_groups groups
    int add16_groups;
    const _Float16* h;                  // GRU: hidden state, S16 [B][h_groups][2][H+2][W+2][8]
    int h_groups;
    const float* z;                     // GRU_Q: update gate, fp32 NCHW
    int keep_z, hidden;
    _Float16* out16;                    // S16 output (nullable) written at group offset out16_goff
    int out16_groups, out16_goff;
    float* out32;                       // fp32 NCHW output (nullable): LINEAR out / GRU_ZR z
    int out_ctot, out_coff;
    int npx, nct;
    int npatch, csplit;                 // block -> (patch, cout tile) mapping, see s16_block_tile()
    // TCS_EPI_BLEND9: DispRefine's softmax blend on the nine outputs (tcs_mi355.h)
    const float* bl_cand; int bl_cand_ctot; const float* bl_disp;
    float* bl_refined; float* bl_delta; float* bl_coords1; float* bl_flow; _Float16* bl_f16; int bl_f16_groups, bl_f16_ch;
    _Float16* out16b; int out16b_groups, out16_split;   // LINEAR: channels >= out16_split go to this second S16 tensor (tcs_mi355.h)
    void* in_ws;                        // DECONV2X: fixed-point (sum, sum of squares) accumulators of the output, see s16_deconv_sums() (nullable)
    const float* tap_w; float* tap_out; int tap_nout, tap_ntile;   // LINEAR: tap partials of a following 3x3 conv to 1-2 channels (tcs_stencil.hip)
    float tap_unscale;
    const float* warm_pyr0; const float* warm_pyr1; const float* warm_pyr2; const float* warm_pyr3; int warm_radius;   // BLEND9: see tcs_mi355.h
    int ablate;                         // diagnostic builds only (-DTCS_S16_ABLATE, tools/conv_s16_ablate.py): bit 0 skip the
                                        // input DMA, bit 1 skip the weight DMA, bit 2 skip operand reads + MFMAs (timing only)
};

// ---------------------------------------------------------------------------------------------------------------------
// epilogue of one 32(cout) x 32(pixel) accumulator tile; lane = pixel, register r -> channel co0 + (r&3) + 8*(r>>2)
// (co0 already includes the lane half's +4).  S16 stores: the 4 registers 4q..4q+3 are 4 consecutive channels of group
// (co0>>3)+q -> one 8-byte store per {hi, lo}.
// ---------------------------------------------------------------------------------------------------------------------
// `st`: this lane's pixel is real (its stores happen).  False only in LINEAR launches with tap partials, where EVERY lane runs the
// epilogue on a clamped (valid) pixel because the fold is an MFMA: the matrix instruction takes its weight rows from all 64 lanes.
// TP (LINEAR only): the launch writes tap partials.  A template parameter, not a run-time branch on a.tap_out: with the fold compiled into
// every LINEAR kernel the register allocation of ALL of them was the fold's (220 VGPRs against 116: two waves per SIMD instead of four).
template <int EPI, bool TP = false>
__device__ __forceinline__ void s16_epilogue_tile(const S16Args& a, int b, int co0, int py, int px, const f32x16& acc, bool st = true) {
    const int H = a.H, W = a.W;
    const size_t HW = (size_t)H * W, pix = (size_t)py * W + px;
#ifdef TCS_S16_PROBE_SLIM
    // diagnostic build (tools/conv_s16_probe.sh): a one-store epilogue, so that the register allocation — and with it the number of waves
    // per SIMD — is the K loop's own.  Results are wrong by construction; only the timing of LINEAR launches means anything.
    if (EPI == TCS_EPI_LINEAR && a.out32) {
        float t = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[r];
        if (st) a.out32[((size_t)b * a.out_ctot + a.out_coff + min(co0, a.Cout - 1)) * HW + pix] = t;
        return;
    }
#endif
    const int Hp = H + 2, Wp = W + 2;
    // tap partials: this tile's four weight fragments (k-steps 2*tile, 2*tile + 1; hi, lo) are requested first, used last
    const bool taps_here = TP && EPI == TCS_EPI_LINEAR && (co0 >> 5) < a.tap_ntile;
    uint4 tw0h = {0, 0, 0, 0}, tw0l = {0, 0, 0, 0}, tw1h = {0, 0, 0, 0}, tw1l = {0, 0, 0, 0};
    if (taps_here) {
        const uint4* twp = reinterpret_cast<const uint4*>(a.tap_w) + (size_t)(co0 >> 5) * 256 + (threadIdx.x & 63);
        tw0h = twp[0]; tw0l = twp[64]; tw1h = twp[128]; tw1l = twp[192];
    }
    int cc[16];
    bool ok[16];
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int co = co0 + (r & 3) + 8 * (r >> 2);
        ok[r] = co < a.Cout;
        cc[r] = min(co, a.Cout - 1);
        v[r] = acc[r] * a.w_unscale;
    }
    if (a.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += a.bias[cc[r]];
    }
    const int sub4 = co0 & 4;                        // this lane half's 4-channel slot inside a group
    if (EPI == TCS_EPI_LINEAR) {
        const bool late = a.act == TCS_ACT_RELU_ADD_RELU;          // relu(relu(v) + addend): the tail of a residual block
        if (a.add1) {
            float t[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) t[r] = a.add1[((size_t)b * a.add_ctot + cc[r]) * HW + pix];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = (late ? fmaxf(v[r], 0.f) : v[r]) + t[r];
        }
        if (a.add16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int gq = min((co0 >> 3) + q, a.add16_groups - 1);
                const _Float16* ap = a.add16 + s16_unit(b, a.add16_groups, gq, 0, Hp, Wp, py, px) + sub4;
                const half4 hi = *reinterpret_cast<const half4*>(ap);
                const half4 lo = *reinterpret_cast<const half4*>(ap + (size_t)Hp * Wp * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[4 * q + j] = (late ? fmaxf(v[4 * q + j], 0.f) : v[4 * q + j]) + (float)hi[j] + (float)lo[j];
            }
        }
        const int act = late ? TCS_ACT_RELU : a.act;
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = ok[r] ? apply_act(v[r], act) * a.post_scale : 0.f;
        if (a.out32) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (ok[r] && st) a.out32[((size_t)b * a.out_ctot + a.out_coff + cc[r]) * HW + pix] = v[r];
        }
        if (taps_here) {
            // tap partials (tcs_stencil.hip): P[tile][o*9 + t][pixel] = sum over this tile's 32 channels of w2[o][c][t] * v[c] — a 32 (rows
            // o*9 + t, <= 18 real) x 32 (channels) x 32 (pixels) product: the activated accumulator registers 8s .. 8s+7 ARE the B fragment
            // of k-step s (the register-chain of k_hidden_update_s16; weights packed in that channel order by tcs_pack_tap_weights), so the
            // fold costs 6 MFMAs and one round trip for the 4 weight fragments (fetched at the top of the epilogue).  A first version
            // with 9 x (4 weight loads -> 16 FMAs -> cross-half shuffle -> store) serialised 9 memory round trips per wave: +1.2 ms per frame.
            half8 bh0, bl0, bh1, bl1;
            split8(v, bh0, bl0);
            split8(v + 8, bh1, bl1);
            f32x16 pa;
#pragma unroll
            for (int i = 0; i < 16; ++i) pa[i] = 0.f;
            const half8 a0h = *reinterpret_cast<const half8*>(&tw0h), a0l = *reinterpret_cast<const half8*>(&tw0l);
            const half8 a1h = *reinterpret_cast<const half8*>(&tw1h), a1l = *reinterpret_cast<const half8*>(&tw1l);
            pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0l, bh0, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bl0, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(a0h, bh0, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1l, bh1, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bl1, pa, 0, 0, 0);
            pa = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1h, bh1, pa, 0, 0, 0);
            const int np = a.tap_nout * 9;
            float* tp = a.tap_out + ((size_t)(b * a.tap_ntile + (co0 >> 5)) * np) * HW + pix;
#pragma unroll
            for (int r = 0; r < 16; ++r) {                 // row of register r: (co0 & 4) + (r&3) + 8*(r>>2)
                const int ot = (co0 & 4) + (r & 3) + 8 * (r >> 2);
                if (ot < np && st) tp[(size_t)ot * HW] = pa[r] * a.tap_unscale;
            }
        }
        if (st && (a.out16 || (a.out16b && co0 >= a.out16_split))) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int g = (co0 >> 3) + q, c_first = co0 + 8 * q;  // 4 consecutive channels c_first .. c_first + 3
                if (c_first < a.Cout) {
                    // two layers in one launch: a tile (32 channels) lies wholly on one side of out16_split (a multiple of 32)
                    const bool second = a.out16b != nullptr && c_first >= a.out16_split;
                    if (!second && !a.out16) continue;              // first range consumed as tap partials only
                    _Float16* o = second ? a.out16b + s16_unit(b, a.out16b_groups, g - (a.out16_split >> 3), 0, Hp, Wp, py, px) + sub4
                                         : a.out16 + s16_unit(b, a.out16_groups, a.out16_goff + g, 0, Hp, Wp, py, px) + sub4;
                    s16_store4(o, (size_t)Hp * Wp * 8, &v[4 * q], a.Cout - c_first);
                }
            }
        }
    } else if (EPI == TCS_EPI_BLEND9) {
        // Cout = 9, one tile: lane (pixel, half 0) holds logits 0-3 in v[0..3] and logit 8 in v[4], its partner lane + 32 (same pixel)
        // holds logits 4-7 in v[0..3].  Both halves are active together (same pixel), so the exchange below is well defined.
        float lg[9];
#pragma unroll
        for (int j = 0; j < 4; ++j) { lg[j] = v[j]; lg[4 + j] = __shfl_xor(v[j], 32); }
        lg[8] = v[4];
        if ((co0 & 4) == 0) {                           // the lower half-wave finishes the pixel
            float m = -INFINITY, sum = 0.f, r = 0.f;
#pragma unroll
            for (int k = 0; k < 9; ++k) m = fmaxf(m, lg[k]);
#pragma unroll
            for (int k = 0; k < 9; ++k) { lg[k] = expf(lg[k] - m); sum += lg[k]; }
            const float* c = a.bl_cand + (size_t)b * a.bl_cand_ctot * HW + pix;
#pragma unroll
            for (int k = 0; k < 9; ++k) r += (lg[k] / sum) * c[(size_t)k * HW];
            const size_t o = (size_t)b * HW + pix;
            a.bl_refined[o] = r;
            if (a.bl_delta) a.bl_delta[o] = r - a.bl_disp[o];
            const float xf = (float)px, c1 = xf - r;
            if (a.bl_coords1) a.bl_coords1[o] = c1;
            if (a.bl_flow) a.bl_flow[o] = c1 - xf;
            if (a.bl_f16) {
                half2_t hi, lo;
                s16_split2(c1 - xf, 0.f, hi, lo);
                _Float16* o16 = a.bl_f16 + s16_unit(b, a.bl_f16_groups, a.bl_f16_ch >> 3, 0, Hp, Wp, py, px) + (a.bl_f16_ch & 7);
                o16[0] = hi[0];
                o16[(size_t)Hp * Wp * 8] = lo[0];
            }
            if (a.warm_pyr0) {
                // touch the rows of the skewed pyramid the next corr lookup reads for this pixel (k_corr_lookup's addressing, tcs_corr.hip).
                // The loads' destination is ONE register that stays allocated ("+v") until they have all come back: a load that returns
                // into a register the compiler has meanwhile given to an address of a later load faults (it did, once).
                float sink = 0.f;
#pragma unroll
                for (int level = 0; level < 4; ++level) {
                    const float* pl = level == 0 ? a.warm_pyr0 : (level == 1 ? a.warm_pyr1 : (level == 2 ? a.warm_pyr2 : a.warm_pyr3));
                    const int Wl = W >> level;
                    float x = c1 * (1.0f / (float)(1 << level));
                    x = fminf(fmaxf(x, -1048576.f), 1048576.f);
                    if (!(x == x)) x = -1048576.f;
                    const int j0 = (int)floorf(x) - a.warm_radius, q = px >> level;
                    const float* base = pl + ((size_t)(b * H + py) * Wl) * W + px;
                    for (int t = 0; t <= 2 * a.warm_radius + 1; ++t) {
                        const int j = j0 + t;
                        int d = q - j;
                        d = d < 0 ? d + Wl : (d >= Wl ? d - Wl : d);
                        d = (j >= 0 && j < Wl && d >= 0 && d < Wl) ? d : 0;
                        asm volatile("global_load_dword %0, %1, off" : "+v"(sink) : "v"(base + (size_t)d * W) : "memory");
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(sink) : : "memory");
            }
        }
    } else if (EPI == TCS_EPI_DECONV2X) {
        // ConvTranspose2d(4, 2, 1) as a 3x3 conv with 4*C outputs (one group of C per output parity), pixel-shuffled:
        // channel par*C + c of pixel (i,j) -> out[c][2i + (par>>1)][2j + (par&1)]; C is a multiple of 8
        const int C = a.hidden, Ho = 2 * H, Wo = 2 * W;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int cbase = co0 + 8 * q;                            // 4 consecutive channels, same parity group
            if (cbase < a.Cout) {
                const int par = cbase / C, c = cbase - par * C;
                float t[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) t[j] = apply_act(v[4 * q + j], a.act);
                half4 hi, lo;
                split4(t, hi, lo);
                _Float16* o = a.out16 + s16_unit(b, a.out16_groups, a.out16_goff + (c >> 3), 0, Ho + 2, Wo + 2, 2 * py + (par >> 1),
                                                  2 * px + (par & 1)) + (c & 4);
                *reinterpret_cast<half4*>(o) = hi;
                *reinterpret_cast<half4*>(o + (size_t)(Ho + 2) * (Wo + 2) * 8) = lo;
            }
        }
    } else {
        // GRU epilogues.  ZR: tile channels < hidden are z (-> fp32 out32), the others r (-> r*h, S16 out16).
        // Q: q = tanh(. + cq), h' = blend(z, h, q) -> S16 out16 (may alias a.h: same positions read then written).
        const bool is_r = (EPI == TCS_EPI_GRU_ZR) && (co0 >= a.hidden);           // tile-uniform (hidden % 32 == 0)
        const int chb = is_r ? co0 - a.hidden : co0;                               // channel within [0, hidden)
        float hh[16], ad[16], zz[16];
        size_t o32[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) o32[r] = ((size_t)b * a.hidden + chb + (r & 3) + 8 * (r >> 2)) * HW + pix;
        const float* addp = (EPI == TCS_EPI_GRU_ZR && is_r) ? a.add2 : a.add1;
        if (addp) {
            const size_t shift = (size_t)b * (a.add_ctot - a.hidden) * HW;         // addend batch stride may exceed `hidden`
#pragma unroll
            for (int r = 0; r < 16; ++r) ad[r] = addp[o32[r] + shift];
        }
        const bool need_h = (EPI == TCS_EPI_GRU_Q) || is_r;
        if (need_h) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const _Float16* hp = a.h + s16_unit(b, a.h_groups, (chb >> 3) + q, 0, Hp, Wp, py, px) + sub4;
                const half4 hi = *reinterpret_cast<const half4*>(hp);
                const half4 lo = *reinterpret_cast<const half4*>(hp + (size_t)Hp * Wp * 8);
#pragma unroll
                for (int j = 0; j < 4; ++j) hh[4 * q + j] = (float)hi[j] + (float)lo[j];
            }
        }
        if (EPI == TCS_EPI_GRU_Q) {
#pragma unroll
            for (int r = 0; r < 16; ++r) zz[r] = a.z[o32[r]];
        }
        float res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float pre = v[r] + (addp ? ad[r] : 0.f);
            if (EPI == TCS_EPI_GRU_ZR) {
                const float g = sigmoidf_(pre);
                res[r] = is_r ? g * hh[r] : g;
            } else {
                const float qv = tanhf(pre);
                res[r] = a.keep_z ? zz[r] * hh[r] + (1.f - zz[r]) * qv : (1.f - zz[r]) * hh[r] + zz[r] * qv;
            }
        }
        if (EPI == TCS_EPI_GRU_ZR && !is_r) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a.out32[o32[r]] = res[r];
        } else {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                half4 hi, lo;
                split4(&res[4 * q], hi, lo);
                _Float16* o = a.out16 + s16_unit(b, a.out16_groups, a.out16_goff + (chb >> 3) + q, 0, Hp, Wp, py, px) + sub4;
                *reinterpret_cast<half4*>(o) = hi;
                *reinterpret_cast<half4*>(o + (size_t)Hp * Wp * 8) = lo;
            }
            if (EPI == TCS_EPI_GRU_Q && a.out32) {
#pragma unroll
                for (int r = 0; r < 16; ++r) a.out32[o32[r]] = res[r];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// InstanceNorm statistics of a transposed convolution's output, accumulated by the convolution itself (TCS_EPI_DECONV2X with
// tcs_conv_s16_desc.in_stats): the up-blocks of both U-Nets are ConvTranspose2d -> InstanceNorm2d -> LeakyReLU
// (core/utils/basic_layers.py:28-35,57), and the statistics pass was a launch of its own on the iteration's serial chain.
// Every workgroup reduces its tile — ROWS x 32 pixels of the input grid x 32 output channels of ONE output parity — to
// (sum x, sum x^2) per channel in a fixed order (butterfly inside a wave, waves in row order), converts them to FIXED POINT
// (2^20 and 2^16 units) and adds them to the 64-bit accumulators [B][C][2] with fire-and-forget atomics: integer addition
// commutes, so the result does not depend on the order the workgroups arrive in (bit-reproducible), nobody waits and nobody
// is last.  The values are the STORED ones (hi + lo of the S16 split), which is what the apply kernel normalises.  The
// caller zero-fills the accumulators before the launch (core/update.py keeps one set per iteration and clears them all with
// one memset per frame).  (Three earlier versions kept (mean, M2) slots per workgroup and had the last workgroup — a ticket
// counter — merge them: device-coherent stores, their acknowledgement, the ticket and the merge were ~7 us of serial
// latency at the tail of a 15-us launch, as much as the statistics launch they replaced)
// Resolution: |x| <= 65504 and <= 2^20 pixels per channel keep both sums inside 63 bits; a partial sum is rounded to 1e-6
// (1.5e-5 for the squares), i.e. <= 3e-8 / 4e-7 per pixel of a 120 x 160 grid — below the fp32 rounding of the sums themselves.
// ---------------------------------------------------------------------------------------------------------------------
#define S16_IN_SCALE_SUM 1048576.f
#define S16_IN_SCALE_SQ 65536.f
template <int MT, int ROWS>
__device__ __forceinline__ void s16_deconv_sums(const S16Args& a, int b, int ct, int py, int px, int wave, int lane,
                                                const f32x16* acc, float* lds /* >= ROWS * 64 floats, free to use */) {
    const int C = a.hidden, l31 = lane & 31, half = lane >> 5;
    const bool valid = px < a.W && py < a.H;
    unsigned long long* sums = reinterpret_cast<unsigned long long*>(a.in_ws) + (size_t)b * C * S16_IN_STRIDE;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        float s1[16], s2[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            half2_t hi, lo;
            const float t0 = apply_act(acc[m][r] * a.w_unscale, a.act), t1 = apply_act(acc[m][r + 1] * a.w_unscale, a.act);
            float2_t x;
            x[0] = __builtin_amdgcn_fmed3f(t0, -65504.f, 65504.f);
            x[1] = __builtin_amdgcn_fmed3f(t1, -65504.f, 65504.f);
            hi = __builtin_convertvector(x, half2_t);
            const float2_t back = __builtin_convertvector(hi, float2_t);
            lo = __builtin_convertvector(x - back, half2_t);
            const float v0 = valid ? (float)hi[0] + (float)lo[0] : 0.f, v1 = valid ? (float)hi[1] + (float)lo[1] : 0.f;
            s1[r] = v0; s1[r + 1] = v1;
            s2[r] = v0 * v0; s2[r + 1] = v1 * v1;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) {             // over the 32 pixels of this lane half
                s1[r] += __shfl_xor(s1[r], o, 64);
                s2[r] += __shfl_xor(s2[r], o, 64);
            }
        }
        // wave partials -> LDS [wave][half][16 regs][2]; lanes 0 and 32 hold their half's 16 channels
        __syncthreads();                                   // the stage buffers are free once every wave has left the K loop
        if (l31 == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                lds[((wave * 2 + half) * 16 + r) * 2 + 0] = s1[r];
                lds[((wave * 2 + half) * 16 + r) * 2 + 1] = s2[r];
            }
        }
        __syncthreads();
        if (wave == 0 && lane < 32) {                      // lane = (half, r): the ROWS waves in row order
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int w = 0; w < ROWS; ++w) {
                t1 += lds[((w * 2 + (lane >> 4)) * 16 + (lane & 15)) * 2];
                t2 += lds[((w * 2 + (lane >> 4)) * 16 + (lane & 15)) * 2 + 1];
            }
            // channel of (half, r) inside the tile: 4*half + (r&3) + 8*(r>>2); the tile's 32 output channels of the 4*C lie in one parity
            const int hh = lane >> 4, r = lane & 15, cout = (ct * MT + m) * 32 + 4 * hh + (r & 3) + 8 * (r >> 2);
            if (cout < a.Cout) {
                const int c = cout % C;
                const long long q1 = __float2ll_rn(t1 * S16_IN_SCALE_SUM), q2 = __float2ll_rn(t2 * S16_IN_SCALE_SQ);
                __hip_atomic_fetch_add(sums + c * S16_IN_STRIDE + 0, (unsigned long long)q1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_fetch_add(sums + c * S16_IN_STRIDE + 1, (unsigned long long)q2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// the kernel
// ---------------------------------------------------------------------------------------------------------------------
// LDS-DMA of one 1 KiB piece: 64 lanes x 16 B from (wave-uniform base + per-lane byte offset) to LDS byte address `dst`
// (wave-uniform) + lane * 16.  M0 is saved/restored inside the statement (it is compiler-reserved).
#define S16_DMA(VOFF, DST, BASE)                                                                                      \
    {                                                                                                                 \
        unsigned keep_;                                                                                               \
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"  \
                     : "=&s"(keep_) : "v"(VOFF), "s"(DST), "s"(BASE) : "memory");                                    \
    }

// RS = 1 ("row split", 3x3 stride 1): a stage holds ONE filter row (3 taps) of KSTEPS k-steps and the ROWS x 34 input rows it
// reads, i.e. the 3x3 convolution runs as three 1x3 convolutions on vertically shifted inputs.  Activations are fetched
// 2x instead of 1.5x (L2-resident), but a stage shrinks from 49 to 21 KiB (MT = 2), so that two or three workgroups fit a CU
// with the 64-channel tile: the 32-channel tile (4 LDS reads per 3 MFMAs) is LDS-bandwidth bound, and one workgroup per CU
// cannot hide its own DMA latency and barriers.
// Block -> (pixel patch, cout tile).  The hardware deals workgroups round-robin over the 8 XCDs (blocks b and b+8 share an
// XCD and its 4 MiB L2; placement is a speed matter only).  A pixel patch's activations are read by every cout tile and a
// cout tile's weights by every patch; only one of the two streams can be made L2-resident per XCD:
//   csplit = 0: cout tile = b % nct — with nct in {1,2,4,8} each XCD sees ONE weight slice but streams ALL activations
//               (each activation byte then leaves the Infinity Cache 8 times).
//   csplit = S >= 1: the cout tiles are cut into S groups; the blocks one XCD receives are a contiguous run of the order
//               (group, patch, tile in group): an XCD works through a run of patches doing ALL tiles of the group for each,
//               so an activation tile is fetched from the Infinity Cache once per group and hits L2 for the other tiles.
__device__ __forceinline__ void s16_block_tile(const S16Args& a, int bid, int nblocks, int& patch, int& ct) {
    if (a.csplit <= 0) { ct = bid % a.nct; patch = bid / a.nct; return; }
    const int q = nblocks >> 3, r = nblocks & 7, xcd = bid & 7;                 // bijective XCD-contiguous renumbering
    const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    const int tg = a.nct / a.csplit;                                            // tiles per group
    const int per_group = a.npatch * tg;
    const int grp = v / per_group, w = v - grp * per_group;
    patch = w / tg;
    ct = grp * tg + (w - patch * tg);
}

// a wave-uniform pointer the compiler may have parked in VGPRs (SGPR pressure): back to an SGPR pair for the "s" operand
__device__ __forceinline__ const char* uniform_ptr(const char* p) {
    const unsigned long long u = reinterpret_cast<unsigned long long>(p);
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
    return reinterpret_cast<const char*>(((unsigned long long)hi << 32) | lo);
}

#ifdef TCS_S16_ABLATE
#define S16_ABL_DMA(IS_INPUT) (!(a.ablate & ((IS_INPUT) ? 1 : 2)))
#define S16_ABL_COMPUTE (!(a.ablate & 4))
// per-workgroup phase stamps of the diagnostic build (100 MHz device clock): [start, first stage landed, K loop done, end]
__device__ unsigned long long tcs_s16_stamps[4 * 8192];
#define S16_STAMP(I) { if (threadIdx.x == 0 && blockIdx.y == 0 && blockIdx.x < 8192) tcs_s16_stamps[4 * blockIdx.x + (I)] = __builtin_amdgcn_s_memrealtime(); }
#else
#define S16_STAMP(I) {}
#define S16_ABL_DMA(IS_INPUT) true
#define S16_ABL_COMPUTE true
#endif

// RPW ("rows per wave") = 2: a wave owns TWO output rows, i.e. every weight fragment it fetches feeds two activation fragments (6 operand
// fragments per 6 MFMAs instead of 8).  The block is ROWS / RPW waves; LDS per block is unchanged.  A tile OPTION (cfg 12xxxx), not the
// heuristic's choice: it only beat the one-row tile while that one was starved of occupancy (see s16_epilogue_tile's TP); with a
// one-store epilogue (-DTCS_S16_PROBE_SLIM) every tile shape times within 5 % — DESIGN.md section 4.
//
// Occupancy target (second __launch_bounds__ argument of hipcc = minimum waves per SIMD).  The K loop of a one-row 32-channel tile needs
// 59 VGPRs; what the kernel is allocated is its epilogue's appetite, and the allocator lands a few registers above a step of the
// occupancy table on the main instances: 98 on LINEAR (four waves per SIMD where 96 give five), 136 / 132 on GRU_Q / tap partials
// (three where 128 give four).  The targets below cost no spill.  GRU_ZR (100) is left alone: 96 costs two spilled registers and the
// shipped gru08.zr tile (8 waves per workgroup, two workgroups per CU either way) gains nothing from a fifth wave.
constexpr int s16_min_waves(int MT, int EPI, int RPW, bool TP) {
    // (two-row tiles: the K loop needs ~100 registers, prologue and epilogue take 182-212; forcing 128 makes the epilogue spill 54-108
    // registers to scratch and costs 50 % of the kernel: 128 -> 128 43.5 against 29.3 us)
#ifdef TCS_S16_PROBE_SLIM
    if (EPI == TCS_EPI_LINEAR && !TP) return RPW == 2 ? (MT == 2 ? 3 : 4) : (MT == 2 ? 4 : 8);
#endif
    if (RPW != 1 || MT != 1) return 1;
    if (TP || EPI == TCS_EPI_GRU_Q) return 4;
    if (EPI == TCS_EPI_LINEAR) return 5;
    return 1;
}

// The kernel body as a device function of (arguments, block index, blocks of this problem, batch element): k_conv_s16 runs one
// problem per launch, k_conv_s16_pair two independent problems in ONE launch (block-index ranges), see below.
template <int KS, int MT, int ROWS, int KSTEPS, int NSTAGE, int STRIDE, int EPI, int RS = 0, int RPW = 1, bool TP = false>
__device__ __forceinline__ void s16_conv_body(const S16Args& a, const int bid_, const int nblocks_, const int b) {
    static_assert(!TP || (EPI == TCS_EPI_LINEAR && KS == 3 && STRIDE == 1), "tap partials: 3x3 stride-1 LINEAR launches");
    static_assert(!RS || (KS == 3 && STRIDE == 1), "row split is for 3x3 stride-1 convolutions");
    static_assert(ROWS % RPW == 0 && (RPW == 1 || (!RS && EPI != TCS_EPI_DECONV2X && EPI != TCS_EPI_BLEND9)), "rows per wave");
    constexpr int NW = ROWS / RPW;                                  // waves per block
    constexpr int HALO = KS / 2, TAPS = KS * KS, TS = RS ? KS : TAPS;                       // TS: taps per stage
    // A strided 1x1 convolution (the projection shortcut of a down-sampling residual block) reads every other pixel: the DMA gathers
    // exactly those (per-lane source offsets are free), so the LDS tile is ROWS x 32 and dense.  Staging the full 2*ROWS-1 x 63
    // window (3.4x the bytes, 56 KiB per stage: one workgroup per CU) made the 64 -> 96 shortcut at half resolution cost 136 us.
    constexpr bool GATHER = KS == 1 && STRIDE == 2;
    constexpr int LSTR = GATHER ? 1 : STRIDE;                                               // pixel stride inside the LDS tile
    constexpr int IH = RS ? ROWS : (GATHER ? ROWS : STRIDE * ROWS + KS - STRIDE), IW = GATHER ? 32 : STRIDE * 32 + KS - STRIDE, IN_CH = IH * IW;
    constexpr int IN_UNITS = KSTEPS * 4 * IN_CH;                    // sub-tiles [kstep][lane half][hi|lo][IH][IW]
    constexpr int NPI = (IN_UNITS + 63) / 64, NPW = KSTEPS * TS * MT * 2, NP = NPI + NPW;
    constexpr int PPW = (NP + NW - 1) / NW;                         // DMA pieces per wave per stage
    constexpr int STAGE_BYTES = NP * 1024, W_OFF = NPI * 1024;
    constexpr int NSTEP = KSTEPS * TS, R = 2 * RPW + 2 * MT;        // operand reads per (k-step, tap)
    static_assert(R <= 15, "lgkmcnt field");
    static_assert(STAGE_BYTES <= 65536, "ds_read immediate offsets are 16 bits");

    S16_STAMP(0)
    const int tid = threadIdx.x, lane = tid & 63, l31 = lane & 31, half = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int ct, patch;
    s16_block_tile(a, bid_, nblocks_, patch, ct);
    const int y0 = (patch / a.npx) * ROWS, x0 = (patch % a.npx) * 32;
    const int Hp = a.Hin + 2, Wp = a.Win + 2;
    const unsigned plane = (unsigned)(Hp * Wp);                     // units per {hi|lo} plane
    const unsigned lds_base = __builtin_amdgcn_groupstaticsize();   // no static __shared__ in this kernel

    // ---- DMA plan: piece p = wave + j*ROWS of every stage; per-lane source offset fixed over the K loop ------------------
    unsigned voff[PPW];
#pragma unroll
    for (int j = 0; j < PPW; ++j) {
        const int p = min(wave + j * NW, NP - 1);                   // surplus slots repeat the last piece (same data, same place)
        if (p < NPI) {
            const int u = min(p * 64 + lane, IN_UNITS - 1);
            const int sub = u / IN_CH, pos = u - sub * IN_CH;
            const int row = pos / IW, col = pos - row * IW;
            // padded coordinates; RS adds the filter row (0..KS-1) through the stage's base pointer
            const int gy = GATHER ? min(STRIDE * (y0 + row) + 1, Hp - 1) : min(STRIDE * y0 + row - HALO + 1, Hp - 1 - (RS ? KS - 1 : 0));
            const int gx = GATHER ? min(STRIDE * (x0 + col) + 1, Wp - 1) : min(STRIDE * x0 + col - HALO + 1, Wp - 1);
            voff[j] = ((unsigned)sub * plane + (unsigned)(gy * Wp + gx)) * 16u;
        } else {
            const int uw = (p - NPI) * 64 + lane;
            const int slot = uw & 63, hl = (uw >> 6) & 1, m = (uw >> 7) % MT, kt = uw / (128 * MT);
            const int gt = (kt / TS) * TAPS + kt % TS;              // global tap index relative to the stage's first one
            voff[j] = ((unsigned)(gt * a.nct32 + m) * 128u + hl * 64u + slot) * 16u;
        }
    }
    // sources (batch-adjusted) and k-step boundaries in scalar registers before the loop: no s_load inside it
    const size_t plane_bytes = (size_t)plane * 16;
    const char* sb0 = reinterpret_cast<const char*>(a.src[0]) + (size_t)b * a.src_groups[0] * 2 * plane_bytes;
    const char* sb1 = reinterpret_cast<const char*>(a.src[1]) + (size_t)b * a.src_groups[1] * 2 * plane_bytes;
    const char* sb2 = reinterpret_cast<const char*>(a.src[2]) + (size_t)b * a.src_groups[2] * 2 * plane_bytes;
    const char* sb3 = reinterpret_cast<const char*>(a.src[3]) + (size_t)b * a.src_groups[3] * 2 * plane_bytes;
    const int ke0 = a.src_kend[0], ke1 = a.src_kend[1], ke2 = a.src_kend[2];
    const char* wbase = reinterpret_cast<const char*>(a.w) + (size_t)ct * MT * 2048;
    const size_t w_kstep_bytes = (size_t)TAPS * a.nct32 * 2048;
    const int nstage_total = (a.nk / KSTEPS) * (RS ? KS : 1);
    const size_t row_bytes = (size_t)Wp * 16, w_row_bytes = (size_t)KS * a.nct32 * 2048;

#define S16_ISSUE(BUF, SIDX)                                                                                          \
    {                                                                                                                 \
        const int k0_ = (RS ? (SIDX) / KS : (SIDX)) * KSTEPS, dy_ = RS ? (SIDX) % KS : 0;                             \
        const bool p1_ = k0_ >= ke0, p2_ = k0_ >= ke1, p3_ = k0_ >= ke2;                                              \
        const char* sp_ = p3_ ? sb3 : (p2_ ? sb2 : (p1_ ? sb1 : sb0));                                                \
        const int ks_ = p3_ ? ke2 : (p2_ ? ke1 : (p1_ ? ke0 : 0));                                                    \
        const char* in_ptr_ = uniform_ptr(sp_ + (size_t)(k0_ - ks_) * 4 * plane_bytes + dy_ * row_bytes);             \
        const char* w_ptr_ = uniform_ptr(wbase + (size_t)k0_ * w_kstep_bytes + dy_ * w_row_bytes);                    \
        const unsigned dst0_ = lds_base + (unsigned)(BUF) * STAGE_BYTES;                                              \
        asm volatile("s_nop 4" ::: "memory");       /* v_readfirstlane-written SGPRs -> VMEM base: wait states */        \
        _Pragma("unroll") for (int j = 0; j < PPW; ++j) {                                                             \
            const int p_ = min(wave + j * NW, NP - 1);                                                                \
            const char* base_ = p_ < NPI ? in_ptr_ : w_ptr_;                                                          \
            if (S16_ABL_DMA(p_ < NPI)) S16_DMA(voff[j], dst0_ + (unsigned)p_ * 1024u, base_)                          \
        }                                                                                                             \
    }

    f32x16 acc[MT * RPW];                                           // [m * RPW + row]
#pragma unroll
    for (int m = 0; m < MT * RPW; ++m)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;

    // operand fetch addresses inside a stage buffer
    const unsigned addr_b0 = lds_base + (unsigned)((half * 2 * IN_CH + LSTR * wave * RPW * IW + LSTR * l31) * 16);
    const unsigned addr_a0 = lds_base + (unsigned)(W_OFF + lane * 16);
    struct Frag { half8 b_hi[RPW], b_lo[RPW], a_hi[MT], a_lo[MT]; };
#define S16_DSREAD(DST, ADDR, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(DST) : "v"(ADDR), "i"(OFF) : "memory")
#define S16_FETCH(F, STEP)                                                                                            \
    {                                                                                                                 \
        const int kk_ = (STEP) / TS, t_ = (STEP) % TS, dy_ = RS ? 0 : t_ / KS, dx_ = t_ % KS;  /* constants after unrolling */ \
        _Pragma("unroll") for (int j = 0; j < RPW; ++j) {                                                             \
            S16_DSREAD(F.b_hi[j], addr_b, ((kk_ * 4 + 0) * IN_CH + (dy_ + LSTR * j) * IW + dx_) * 16);                \
            S16_DSREAD(F.b_lo[j], addr_b, ((kk_ * 4 + 1) * IN_CH + (dy_ + LSTR * j) * IW + dx_) * 16);                \
        }                                                                                                             \
        _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                              \
            S16_DSREAD(F.a_hi[m], addr_a, (((kk_ * TS + t_) * MT + m) * 2 + 0) * 1024);                               \
            S16_DSREAD(F.a_lo[m], addr_a, (((kk_ * TS + t_) * MT + m) * 2 + 1) * 1024);                               \
        }                                                                                                             \
    }
#define S16_WAIT_LGKM(N) { asm volatile("s_waitcnt lgkmcnt(%0)" :: "i"(N) : "memory"); __builtin_amdgcn_sched_barrier(0); }
#define S16_MMA(F)                                                                                                    \
    _Pragma("unroll") for (int m = 0; m < MT; ++m) {                                                                  \
        _Pragma("unroll") for (int j = 0; j < RPW; ++j) {                                                             \
            acc[m * RPW + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_lo[m], F.b_hi[j], acc[m * RPW + j], 0, 0, 0); \
            acc[m * RPW + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_hi[m], F.b_lo[j], acc[m * RPW + j], 0, 0, 0); \
            acc[m * RPW + j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(F.a_hi[m], F.b_hi[j], acc[m * RPW + j], 0, 0, 0); \
        }                                                                                                             \
    }

    // ---- prologue: every buffer is free, so NSTAGE stages go in flight at once ------------------------------------------
#pragma unroll
    for (int s = 0; s < NSTAGE; ++s)
        if (s < nstage_total) S16_ISSUE(s, s)

    int buf = 0;                                                    // buffer of the stage being multiplied
    for (int s = 0; s < nstage_total; ++s) {
        // own pieces of stage s have landed; the `newer` stages issued after it (PPW pieces each) may stay in flight:
        // stages 1 .. NSTAGE-1 of the prologue at s = 0, afterwards what iteration s-1 issued (stage s + NSTAGE - 2)
        const int newer = s == 0 ? min(nstage_total - 1, NSTAGE - 1) : min(nstage_total - 1 - s, NSTAGE - 2);
        static_assert((NSTAGE - 1) * PPW <= 63, "vmcnt field");
        if (NSTAGE >= 4 && newer == 3) { asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NSTAGE >= 4 ? 3 : 0) * PPW) : "memory"); }
        else if (NSTAGE >= 3 && newer == 2) { asm volatile("s_waitcnt vmcnt(%0)" :: "i"((NSTAGE >= 3 ? 2 : 0) * PPW) : "memory"); }
        else if (newer == 1) { asm volatile("s_waitcnt vmcnt(%0)" :: "i"(PPW) : "memory"); }
        else { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        __builtin_amdgcn_s_barrier();                               // everyone's pieces landed; everyone is done with stage s-1
        if (s == 0) { S16_STAMP(1) }
        if (NSTAGE >= 2 && s >= 1 && s + NSTAGE - 1 < nstage_total) {   // refill the buffer that stage s-1 was multiplied from
            const int nb = buf == 0 ? NSTAGE - 1 : buf - 1;
            S16_ISSUE(nb, s + NSTAGE - 1)
        }
        if (S16_ABL_COMPUTE) {
            const unsigned addr_b = addr_b0 + (unsigned)buf * STAGE_BYTES, addr_a = addr_a0 + (unsigned)buf * STAGE_BYTES;
            Frag f0, f1;
            S16_FETCH(f0, 0)
#pragma unroll
            for (int i = 0; i < NSTEP; i += 2) {
                if (i + 1 < NSTEP) { S16_FETCH(f1, (i + 1 < NSTEP ? i + 1 : 0)) S16_WAIT_LGKM(R) } else S16_WAIT_LGKM(0)
                S16_MMA(f0)
                if (i + 1 < NSTEP) {
                    if (i + 2 < NSTEP) { S16_FETCH(f0, (i + 2 < NSTEP ? i + 2 : 0)) S16_WAIT_LGKM(R) } else S16_WAIT_LGKM(0)
                    S16_MMA(f1)
                }
            }
        }
        if (NSTAGE == 1 && s + 1 < nstage_total) {
            // single buffer (small LDS footprint: up to five workgroups per CU hide each other's fills): everyone is done
            // reading before the next stage overwrites it
            __builtin_amdgcn_s_barrier();
            S16_ISSUE(0, s + 1)
        }
        buf = buf + 1 == NSTAGE ? 0 : buf + 1;
    }
#undef S16_ISSUE
#undef S16_FETCH
#undef S16_MMA
    S16_STAMP(2)

    const int px = x0 + l31, py = y0 + wave * RPW;
    if constexpr (EPI == TCS_EPI_DECONV2X) {
        // InstanceNorm sums FIRST: their atomics travel to the memory side (~2 us) while the tile is stored below.  (After the stores the
        // same code made the launch 6-9 us longer — as much as the statistics launch it replaces.)
        if (a.in_ws) {
            extern __shared__ __attribute__((aligned(16))) float s16_dyn_lds[];
            s16_deconv_sums<MT, ROWS>(a, b, ct, py, px, wave, lane, acc, s16_dyn_lds);
        }
    }
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        if (TP) {
            // tap partials: all 64 lanes run the epilogue (on clamped pixels; only real pixels store) — see s16_epilogue_tile
            const bool st = px < a.W && py + j < a.H;
#pragma unroll
            for (int m = 0; m < MT; ++m)
                s16_epilogue_tile<EPI, true>(a, b, (ct * MT + m) * 32 + 4 * half, min(py + j, a.H - 1), min(px, a.W - 1), acc[m * RPW + j], st);
        } else if (px < a.W && py + j < a.H) {
#pragma unroll
            for (int m = 0; m < MT; ++m) s16_epilogue_tile<EPI>(a, b, (ct * MT + m) * 32 + 4 * half, py + j, px, acc[m * RPW + j]);
        }
    }
#ifdef TCS_S16_ABLATE
    __builtin_amdgcn_s_waitcnt(0);                                  // stores acknowledged
    S16_STAMP(3)
#endif
}

template <int KS, int MT, int ROWS, int KSTEPS, int NSTAGE, int STRIDE, int EPI, int RS = 0, int RPW = 1, bool TP = false>
__global__ __launch_bounds__(64 * ROWS / RPW, s16_min_waves(MT, EPI, RPW, TP)) void k_conv_s16(S16Args a) {
    s16_conv_body<KS, MT, ROWS, KSTEPS, NSTAGE, STRIDE, EPI, RS, RPW, TP>(a, blockIdx.x, gridDim.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------------------
// Two independent convolutions as ONE launch ("grouped launch").  Inside a refinement iteration pairs of layers that do not
// depend on each other (the two 3x3 layers of the motion encoder's halves, the second layers of the gradient predictor's
// stems, DispRefine's context branch beside its candidate stem) used to run as parallel branches of the captured graph — and
// a dependency between two branches costs ~6 us on the branch that stays in its queue and ~11 us on the other one, against
// ~1.5 us between two launches of one queue (profiles/r04_iteration_timeline.txt).  Here both problems share a grid: blocks
// [0, n0) run problem 0's tile code, blocks [n0pad, gridDim.x) problem 1's (n0pad = n0 rounded up to a multiple of 8, so that
// problem 1's XCD-contiguous block renumbering still sees block % 8 = its XCD; the <= 7 blocks in between exit at once).  The
// two problems may be different template instances: the kernel is launched with the larger block, LDS and register budget;
// surplus waves of the smaller instance exit before its first barrier (a barrier counts the waves that are still alive).
// Same arithmetic, same order per output as two launches: results are bit-equal (tests/test_gpu_s16.py).
// ---------------------------------------------------------------------------------------------------------------------
template <int KS0, int MT0, int ROWS0, int KST0, int NST0, int EPI0, int KS1, int MT1, int ROWS1, int KST1, int NST1, int EPI1>
__global__ __launch_bounds__(64 * (ROWS0 > ROWS1 ? ROWS0 : ROWS1),
                             (s16_min_waves(MT0, EPI0, 1, false) < s16_min_waves(MT1, EPI1, 1, false) ? s16_min_waves(MT0, EPI0, 1, false)
                                                                                                      : s16_min_waves(MT1, EPI1, 1, false)))
void k_conv_s16_pair(S16Args a0, S16Args a1, int n0, int n0pad) {
    const int bid = blockIdx.x;
    if (bid < n0pad) {
        if (bid >= n0 || (int)threadIdx.x >= 64 * ROWS0) return;
        s16_conv_body<KS0, MT0, ROWS0, KST0, NST0, 1, EPI0>(a0, bid, n0, blockIdx.y);
    } else {
        if ((int)threadIdx.x >= 64 * ROWS1) return;
        s16_conv_body<KS1, MT1, ROWS1, KST1, NST1, 1, EPI1>(a1, bid - n0pad, (int)gridDim.x - n0pad, blockIdx.y);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 NCHW <-> S16
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_s16_from_f32(const float* __restrict__ x, int C, int H, int W, int G, int goff, int Gtot,
                                                       _Float16* __restrict__ out) {
    // thread = one unit (b, g, y, x): 8 channels of one pixel; channels >= C are written as zero
    const int HW = H * W;
    const size_t n = (size_t)G * HW;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const int g = (int)(i / HW), p = (int)(i - (size_t)g * HW);
    const int y = p / W, xx = p - y * W;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = g * 8 + j;
        v[j] = c < C ? x[((size_t)b * C + c) * HW + p] : 0.f;
    }
    half4 h0, l0, h1, l1;
    split4(v, h0, l0);
    split4(v + 4, h1, l1);
    half8 hi, lo;
#pragma unroll
    for (int j = 0; j < 4; ++j) { hi[j] = h0[j]; hi[j + 4] = h1[j]; lo[j] = l0[j]; lo[j + 4] = l1[j]; }
    _Float16* o = out + s16_unit(b, Gtot, goff + g, 0, H + 2, W + 2, y, xx);
    *reinterpret_cast<half8*>(o) = hi;
    *reinterpret_cast<half8*>(o + (size_t)(H + 2) * (W + 2) * 8) = lo;
}

__global__ __launch_bounds__(256) void k_s16_to_f32(const _Float16* __restrict__ s, int C, int H, int W, int Gtot, int goff,
                                                     float* __restrict__ out) {
    const int HW = H * W, G = (C + 7) / 8;
    const size_t n = (size_t)G * HW;
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int b = blockIdx.y;
    if (i >= n) return;
    const int g = (int)(i / HW), p = (int)(i - (size_t)g * HW);
    const int y = p / W, xx = p - y * W;
    const _Float16* u = s + s16_unit(b, Gtot, goff + g, 0, H + 2, W + 2, y, xx);
    const half8 hi = *reinterpret_cast<const half8*>(u);
    const half8 lo = *reinterpret_cast<const half8*>(u + (size_t)(H + 2) * (W + 2) * 8);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = g * 8 + j;
        if (c < C) out[((size_t)b * C + c) * HW + p] = (float)hi[j] + (float)lo[j];
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------------------------------
// A launch that tcs_conv2d_s16_group has asked to be PLANNED instead of issued: the chosen tile instance, its finished arguments and
// launch geometry, and a way to issue it alone after all (when no pair kernel exists for the two instances).
struct S16Plan {
    bool filled;
    long long key;                 // s16_key() of the instance
    S16Args args;
    int nblocks, threads, B;
    size_t lds;
    int (*launch_alone)(S16Args&, hipStream_t);
};
static thread_local S16Plan* g_s16_plan = nullptr;

constexpr long long s16_key(int KS, int MT, int ROWS, int KSTEPS, int NSTAGE, int STRIDE, int EPI, int RS, int RPW, bool TP) {
    return ((((((((((long long)KS * 10 + MT) * 10 + ROWS) * 10 + KSTEPS) * 10 + NSTAGE) * 10 + STRIDE) * 10 + EPI) * 10 + RS) * 10 + RPW) * 2 + (TP ? 1 : 0));
}

template <int KS, int MT, int ROWS, int KSTEPS, int NSTAGE, int STRIDE, int EPI, int RS = 0, int RPW = 1, bool TP = false>
static int launch_s16(S16Args& a, hipStream_t s) {
    constexpr bool GATHER = KS == 1 && STRIDE == 2;                 // (as in the kernel)
    constexpr int IH = RS ? ROWS : (GATHER ? ROWS : STRIDE * ROWS + KS - STRIDE), IW = GATHER ? 32 : STRIDE * 32 + KS - STRIDE, TS = RS ? KS : KS * KS;
    constexpr int NPI = (KSTEPS * 4 * IH * IW + 63) / 64, NP = NPI + KSTEPS * TS * MT * 2;
    constexpr size_t lds = (size_t)NSTAGE * NP * 1024;
    static_assert(lds <= 160 * 1024, "LDS budget");
    auto kern = k_conv_s16<KS, MT, ROWS, KSTEPS, NSTAGE, STRIDE, EPI, RS, RPW, TP>;
    a.npx = tcs_cdiv(a.W, 32);
    a.nct = a.nct32 / MT;
    a.npatch = a.npx * tcs_cdiv(a.H, ROWS);
    if (a.csplit > 0 && a.nct % a.csplit != 0) a.csplit = 1;
    if (g_s16_plan && !g_s16_plan->filled) {                        // tcs_conv2d_s16_group: plan, do not launch
        S16Plan& p = *g_s16_plan;
        p.filled = true;
        p.key = s16_key(KS, MT, ROWS, KSTEPS, NSTAGE, STRIDE, EPI, RS, RPW, TP);
        p.args = a;
        p.nblocks = a.npatch * a.nct; p.threads = 64 * ROWS / RPW; p.B = a.B; p.lds = lds;
        p.launch_alone = &launch_s16<KS, MT, ROWS, KSTEPS, NSTAGE, STRIDE, EPI, RS, RPW, TP>;
        return TCS_OK;
    }
    (void)hipGetLastError();                                        // a stale error of an earlier runtime call is not ours
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return TCS_ELAUNCH;
    }
    hipLaunchKernelGGL(kern, dim3(a.npatch * a.nct, a.B), dim3(64 * ROWS / RPW), lds, s, a);
    return tcs_launch_status();
}

template <int KS0, int MT0, int ROWS0, int KST0, int NST0, int EPI0, int KS1, int MT1, int ROWS1, int KST1, int NST1, int EPI1>
static int launch_s16_pair(const S16Plan& p0, const S16Plan& p1, hipStream_t s) {
    auto kern = k_conv_s16_pair<KS0, MT0, ROWS0, KST0, NST0, EPI0, KS1, MT1, ROWS1, KST1, NST1, EPI1>;
    const size_t lds = p0.lds > p1.lds ? p0.lds : p1.lds;
    (void)hipGetLastError();
    if (lds > 64 * 1024) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return TCS_ELAUNCH;
    }
    const int n0pad = (p0.nblocks + 7) & ~7;
    hipLaunchKernelGGL(kern, dim3(n0pad + p1.nblocks, p0.B), dim3(64 * (ROWS0 > ROWS1 ? ROWS0 : ROWS1)), lds, s, p0.args, p1.args, p0.nblocks, n0pad);
    return tcs_launch_status();
}

// tile configuration: cfg = CSPLIT*100000 + RS*10000 + MT*1000 + ROWS*100 + KSTEPS*10 + NSTAGE (0 = heuristic; CSPLIT: see
// s16_block_tile; RS = 1: row split, RS = 2: two rows per wave); unknown combinations -> EUNSUPPORTED
template <int KS, int STRIDE, int EPI>
static int launch_s16_cfg(S16Args& a, int cfg, hipStream_t s) {
#define S16_CASE(MT_, ROWS_, KST_, NST_) \
    case (MT_ * 1000 + ROWS_ * 100 + KST_ * 10 + NST_): return launch_s16<KS, MT_, ROWS_, KST_, NST_, STRIDE, EPI>(a, s);
#define S16_CASE_RS(MT_, ROWS_, KST_, NST_) \
    case (10000 + MT_ * 1000 + ROWS_ * 100 + KST_ * 10 + NST_): return launch_s16<KS, MT_, ROWS_, KST_, NST_, STRIDE, EPI, 1>(a, s);
#define S16_CASE_RPW2(MT_, ROWS_, KST_, NST_) \
    case (20000 + MT_ * 1000 + ROWS_ * 100 + KST_ * 10 + NST_): return launch_s16<KS, MT_, ROWS_, KST_, NST_, STRIDE, EPI, 0, 2>(a, s);
    if constexpr (KS == 3 && STRIDE == 1 && EPI == TCS_EPI_LINEAR) {
        if (a.tap_out) {                                // tap partials: their own instances (see s16_epilogue_tile), the tiles the loop uses
            switch (cfg) {
                case 1411: return launch_s16<KS, 1, 4, 1, 1, STRIDE, EPI, 0, 1, true>(a, s);
                case 1412: return launch_s16<KS, 1, 4, 1, 2, STRIDE, EPI, 0, 1, true>(a, s);
                case 1812: return launch_s16<KS, 1, 8, 1, 2, STRIDE, EPI, 0, 1, true>(a, s);
                case 21812: return launch_s16<KS, 1, 8, 1, 2, STRIDE, EPI, 0, 2, true>(a, s);
                default: return TCS_EUNSUPPORTED;
            }
        }
    } else if (a.tap_out) return TCS_EUNSUPPORTED;
    if constexpr (KS == 3 && STRIDE == 1 && (EPI == TCS_EPI_LINEAR || EPI == TCS_EPI_GRU_ZR || EPI == TCS_EPI_GRU_Q)) {
        switch (cfg) {
            S16_CASE_RPW2(1, 8, 1, 2) S16_CASE_RPW2(2, 8, 1, 2) S16_CASE_RPW2(1, 4, 1, 2) S16_CASE_RPW2(1, 4, 1, 1) S16_CASE_RPW2(2, 4, 1, 2)
            default: break;
        }
    }
    if constexpr (KS == 3 && STRIDE == 1) {
        switch (cfg) {
            S16_CASE(1, 4, 1, 1) S16_CASE(2, 4, 1, 1) S16_CASE(1, 8, 1, 1)
            S16_CASE(1, 4, 1, 2) S16_CASE(1, 4, 1, 3) S16_CASE(2, 4, 1, 2) S16_CASE(2, 4, 1, 3)
            S16_CASE(1, 5, 1, 2) S16_CASE(2, 5, 1, 2)
            S16_CASE(1, 8, 1, 2) S16_CASE(2, 8, 1, 2)
            S16_CASE_RS(2, 4, 1, 2) S16_CASE_RS(2, 4, 1, 3)
            default: return TCS_EUNSUPPORTED;
        }
    } else if constexpr (KS == 3) {
        switch (cfg) {
            S16_CASE(1, 4, 1, 2) S16_CASE(1, 4, 1, 1)
            default: return TCS_EUNSUPPORTED;
        }
    } else if constexpr (STRIDE == 2) {                 // 1x1 stride 2 (the projection shortcut of a down-sampling residual block)
        switch (cfg) {
            S16_CASE(1, 4, 1, 2) S16_CASE(1, 4, 2, 2)
            default: return TCS_EUNSUPPORTED;
        }
    } else {
        switch (cfg) {
            S16_CASE(1, 4, 1, 2) S16_CASE(2, 4, 1, 2)
            S16_CASE(1, 4, 2, 2) S16_CASE(1, 4, 2, 3) S16_CASE(2, 4, 2, 2) S16_CASE(2, 4, 2, 3)
            S16_CASE(1, 4, 4, 2) S16_CASE(2, 4, 4, 2)
            default: return TCS_EUNSUPPORTED;
        }
    }
#undef S16_CASE
#undef S16_CASE_RS
#undef S16_CASE_RPW2
}

// Tile choice by grid size, from the layer sweep of tools/bench_conv_s16.py on MI355X (gpurun_out/r2_s16_c.log):
//  * 3x3 on grids that give >= 180 workgroups with 8-row patches: see the branches below (gpurun_out/r2_s16_f.log);
//  * smaller grids (1/8 scale with 128 outputs, 1/16 scale, transposed convs): 4-row patches, 32-channel tiles, two stages;
//  * 1x1: two k-steps per stage; 64-channel tiles for Cout >= 256;
//  * CSPLIT = 1 (all cout tiles of a patch on one XCD, s16_block_tile): 0-3 % on 3x3, 20-30 % on the 1x1 layers.
static int s16_heuristic(const S16Args& a, int ksize, int stride, int kst1x1, int epilogue) {
    // strided 1x1 (projection shortcuts): one k-step per stage, all cout tiles of a patch on one XCD — 74.6 -> 56.5 us at 480x640 ->
    // 240x320 (two images), 31.6 -> 18.6 one level down (profiles/r03_conv_s16_stride2_sweep.txt)
    if (ksize == 1 && stride == 2) return 100000 + 1000 + 400 + 10 + 2;
    if (ksize == 1) {
        const int mt = (a.nct32 % 2 == 0 && a.nct32 >= 8) ? 2 : 1;
        return 100000 + mt * 1000 + 400 + kst1x1 * 10 + 2;
    }
    if (stride == 2) {
        // 3x3 stride 2: the stage is 37 + 18 KiB (a 9 x 65 input window per k-step).  Two stages = one workgroup per CU; on grids of more
        // than one workgroup per CU ONE stage (two per CU, hiding each other's fills) wins: feature extractor 480x640 -> 240x320
        // 180.6 -> 125.5 us, 240x320 -> 120x160 73.2 -> 48.7; the loop's small grids keep two stages.  All cout tiles of a patch on one
        // XCD (CSPLIT = 1) either way: 16.1 -> 13.8 us at 1/4 -> 1/8 scale (profiles/r03_conv_s16_stride2_sweep.txt)
        const long long blocks4 = (long long)tcs_cdiv(a.W, 32) * tcs_cdiv(a.H, 4) * a.B * a.nct32;
        return blocks4 > 256 ? 101411 : 101412;
    }
    const long long blocks8 = (long long)tcs_cdiv(a.W, 32) * tcs_cdiv(a.H, 8) * a.B * a.nct32;
    if (blocks8 >= 280) {
        // 1/4-scale grids: the long, wide layers (gru08.zr) keep 8-row patches with two stages; everything else runs as 4-row
        // patches with ONE 31 KiB stage, so that up to five workgroups per CU cover each other's fills (128->128: 25.4 -> 23.8 us,
        // 192->128: 33.4 -> 31.7 us, gru08.q 69.5 -> 58.7 us)
        if (a.nct32 >= 8 && a.nk >= 16) return 100000 + 1000 + 800 + 10 + 2;
        // (Two rows per wave — cfg 121812 — was the pick for 128..192 -> 96..128 channels and for the feature extractor's big grids while
        // every LINEAR instance carried the tap-partial fold's 220 registers (two waves per SIMD).  With the fold in its own instances
        // (95 registers, five waves) the one-row tile wins everywhere: 128 -> 128 25.9 against 29.8 us, 192 -> 128 33.8 / 38.7,
        // half-resolution 96 -> 96 89.5 / 98.0, full-resolution 64 -> 64 201 / 215 — profiles/r03_conv_s16_occupancy_sweep.txt.)
        return 100000 + 1000 + 400 + 10 + 1;
    }
    if (blocks8 >= 180) return 100000 + 1000 + 800 + 10 + 2;
    return 100000 + 1000 + 400 + 10 + 2;
}

extern "C" {

size_t tcs_deconv_in_stats_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || C % 32 != 0 || H <= 0 || W <= 0 || (long long)4 * H * W > (1 << 20)) return 0;    // (s16_deconv_sums: 63-bit sums)
    return (size_t)B * C * S16_IN_STRIDE * sizeof(long long);
}

size_t tcs_s16_bytes(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    const size_t G = (size_t)((C + 15) / 16) * 2;
    return (size_t)B * G * 2 * (H + 2) * (W + 2) * 16;
}

int tcs_s16_from_f32(const float* x, int B, int C, int H, int W, void* s16, int groups_total, int group_offset, tcs_stream_t stream) {
    if (!x || !s16 || B <= 0 || B > 65535 || C <= 0 || H <= 0 || W <= 0) return TCS_EINVAL;
    const int G = ((C + 15) / 16) * 2;
    if (group_offset < 0 || (group_offset & 1) || group_offset + G > groups_total) return TCS_EINVAL;
    const size_t n = (size_t)G * H * W;
    hipLaunchKernelGGL(k_s16_from_f32, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, tcs_stream(stream), x, C, H, W, G, group_offset,
                       groups_total, reinterpret_cast<_Float16*>(s16));
    return tcs_launch_status();
}

int tcs_s16_to_f32(const void* s16, int B, int C, int H, int W, int groups_total, int group_offset, float* out, tcs_stream_t stream) {
    if (!out || !s16 || B <= 0 || B > 65535 || C <= 0 || H <= 0 || W <= 0) return TCS_EINVAL;
    const int G = (C + 7) / 8;
    if (group_offset < 0 || group_offset + G > groups_total) return TCS_EINVAL;
    const size_t n = (size_t)G * H * W;
    hipLaunchKernelGGL(k_s16_to_f32, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, tcs_stream(stream),
                       reinterpret_cast<const _Float16*>(s16), C, H, W, groups_total, group_offset, out);
    return tcs_launch_status();
}

int tcs_conv2d_s16(const tcs_conv_s16_desc* d, tcs_stream_t stream) {
    if (!d || !d->weight || d->n_src < 1 || d->n_src > TCS_MAX_SRC) return TCS_EINVAL;
    if (d->B <= 0 || d->B > 65535 || d->H <= 0 || d->W <= 0 || d->Cout <= 0) return TCS_EINVAL;
    if (d->ksize != 1 && d->ksize != 3) return TCS_EUNSUPPORTED;
    const int stride = d->stride == 2 ? 2 : 1;
    if (stride == 2 && d->epilogue != TCS_EPI_LINEAR) return TCS_EUNSUPPORTED;
    S16Args a;
    int ktot = 0, cin = 0;
    for (int i = 0; i < TCS_MAX_SRC; ++i) {
        const bool used = i < d->n_src;
        if (used) {
            if (!d->src[i] || d->src_ch[i] <= 0 || d->src_groups[i] < ((d->src_ch[i] + 15) / 16) * 2 || (d->src_groups[i] & 1)) return TCS_EINVAL;
            // every source but the last must fill whole 16-channel k-steps, or the weights' K index would shift
            if (i + 1 < d->n_src && d->src_ch[i] % 16 != 0) return TCS_EINVAL;
            const int k = (d->src_ch[i] + 15) / 16;
            ktot += k;
            cin += d->src_ch[i];
        }
        a.src[i] = reinterpret_cast<const _Float16*>(used ? d->src[i] : d->src[0]);
        a.src_groups[i] = used ? d->src_groups[i] : d->src_groups[0];
        a.src_kend[i] = used ? ktot : 0x7fffffff;
    }
    if (cin != d->Cin) return TCS_EINVAL;
    a.w = d->weight; a.bias = d->bias;
    a.B = d->B; a.Hin = d->H; a.Win = d->W;
    a.H = stride == 2 ? (d->H - 1) / 2 + 1 : d->H;
    a.W = stride == 2 ? (d->W - 1) / 2 + 1 : d->W;
    a.Cout = d->Cout; a.nct32 = (d->Cout + 31) / 32;
    a.act = d->act; a.post_scale = d->post_scale; a.w_unscale = d->weight_unscale;
    a.add1 = d->addend; a.add2 = d->addend2;
    a.add16 = reinterpret_cast<const _Float16*>(d->addend16); a.add16_groups = d->addend16_groups;
    if (a.add16 && (d->epilogue != TCS_EPI_LINEAR || a.add16_groups < (d->Cout + 7) / 8)) return TCS_EINVAL;
    if (d->act == TCS_ACT_RELU_ADD_RELU && !a.add16 && !a.add1) return TCS_EINVAL;
    a.h = reinterpret_cast<const _Float16*>(d->h); a.h_groups = d->h_groups; a.z = d->z;
    a.keep_z = d->blend_keep_z; a.hidden = 0;
    a.out16 = reinterpret_cast<_Float16*>(d->out16); a.out16_groups = d->out16_groups; a.out16_goff = d->out16_group_offset;
    a.out32 = d->out32; a.out_ctot = d->out_ctot; a.out_coff = d->out_coff;
    a.out16b = reinterpret_cast<_Float16*>(d->out16b); a.out16b_groups = d->out16b_groups; a.out16_split = d->out16_split;
    a.tap_w = d->tap_weights; a.tap_out = d->tap_out; a.tap_nout = d->tap_nout; a.tap_ntile = d->tap_tiles; a.tap_unscale = d->tap_unscale;
    if (a.tap_out) {
        if (d->epilogue != TCS_EPI_LINEAR || stride != 1 || !a.tap_w || a.tap_nout < 1 || a.tap_nout > 2 || a.tap_ntile < 1 ||
            a.tap_ntile > (d->Cout + 31) / 32) return TCS_EINVAL;
    }
    if (a.out16b) {
        if (d->epilogue != TCS_EPI_LINEAR || (!a.out16 && !a.tap_out) || a.out16_split <= 0 || a.out16_split % 32 != 0 || a.out16_split >= d->Cout) return TCS_EINVAL;
        if (a.out16b_groups < (d->Cout - a.out16_split + 7) / 8) return TCS_EINVAL;
    }
    a.in_ws = d->in_stats;
    if (a.in_ws && (d->epilogue != TCS_EPI_DECONV2X || d->Cout % 128 != 0 || (long long)4 * d->H * d->W > (1 << 20))) return TCS_EINVAL;
    a.npx = 0; a.nct = 0; a.npatch = 0;
    a.csplit = (d->tile_cfg / 100000) % 10;            // 0 = cout tile fastest (one weight slice per XCD)
    a.ablate = d->tile_cfg / 1000000;                  // honoured by -DTCS_S16_ABLATE builds only
    a.bl_cand = nullptr; a.bl_cand_ctot = 0; a.bl_disp = nullptr; a.bl_refined = nullptr; a.bl_delta = nullptr; a.bl_coords1 = nullptr;
    a.bl_flow = nullptr; a.bl_f16 = nullptr; a.bl_f16_groups = 0; a.bl_f16_ch = 0;
    a.warm_pyr0 = a.warm_pyr1 = a.warm_pyr2 = a.warm_pyr3 = nullptr; a.warm_radius = 0;
    if (!a.out16 && !a.out32 && !a.tap_out && d->epilogue != TCS_EPI_BLEND9) return TCS_EINVAL;
    // the packed weight image pads K to a multiple of 64 channels (tcs_conv_packed_floats_f16x3): nk may not exceed it
    const int kpack = ((d->Cin + 63) / 64) * 4;
    int kst = 1;
    if (d->ksize == 1) {                               // two k-steps per stage when every source has an even number of them
        kst = 2;
        for (int i = 0; i < d->n_src; ++i)
            if (((d->src_ch[i] + 15) / 16) % 2 != 0) kst = 1;
    }
    int cfg = d->tile_cfg % 100000;
    if (cfg) kst = (cfg / 10) % 10;
    if (kst != 1 && kst != 2 && kst != 4) return TCS_EINVAL;
    for (int i = 0; i < d->n_src; ++i)
        if (((d->src_ch[i] + 15) / 16) % kst != 0) {
            // pad the LAST source's k-steps up to the stage size when its tensor and the packed weights hold that much
            if (i + 1 == d->n_src && d->src_groups[i] >= 2 * (((d->src_ch[i] + 15) / 16 + kst - 1) / kst * kst)) {
                const int k = (d->src_ch[i] + 15) / 16, kp = (k + kst - 1) / kst * kst;
                ktot += kp - k;
                a.src_kend[i] = ktot;
            } else return TCS_EINVAL;
        }
    if (ktot > kpack) return TCS_EINVAL;
    a.nk = ktot;
    const int out16_ch = d->epilogue == TCS_EPI_GRU_ZR ? d->Cout / 2 : (d->epilogue == TCS_EPI_DECONV2X ? d->Cout / 4 : (a.out16b ? a.out16_split : d->Cout));
    const int outG = a.out16 ? (out16_ch + 7) / 8 : 0;
    if (a.out16 && (a.out16_goff < 0 || a.out16_goff + outG > a.out16_groups)) return TCS_EINVAL;
    if (a.out32 && d->epilogue == TCS_EPI_LINEAR && (d->out_coff < 0 || d->out_coff + d->Cout > d->out_ctot)) return TCS_EINVAL;
    hipStream_t s = tcs_stream(stream);
    if (!cfg) {
        cfg = s16_heuristic(a, d->ksize, stride, kst, d->epilogue);
        a.csplit = cfg / 100000;
        cfg %= 100000;
    }
    if (a.nct32 % ((cfg / 1000) % 10) != 0) return TCS_EUNSUPPORTED;          // the cout tile must divide the packed tiles

    {   // addend channel stride: defaults to the epilogue's own channel count
        const int own = d->epilogue == TCS_EPI_GRU_ZR ? d->Cout / 2 : d->Cout;
        if (d->addend_ctot != 0 && d->addend_ctot < own) return TCS_EINVAL;
        a.add_ctot = d->addend_ctot ? d->addend_ctot : own;
    }
    switch (d->epilogue) {
        case TCS_EPI_LINEAR:
            if (stride == 2 && d->ksize == 1) return launch_s16_cfg<1, 2, TCS_EPI_LINEAR>(a, cfg, s);
            if (stride == 2) return launch_s16_cfg<3, 2, TCS_EPI_LINEAR>(a, cfg, s);
            return d->ksize == 3 ? launch_s16_cfg<3, 1, TCS_EPI_LINEAR>(a, cfg, s) : launch_s16_cfg<1, 1, TCS_EPI_LINEAR>(a, cfg, s);
        case TCS_EPI_BLEND9:
            if (d->ksize != 1 || stride != 1 || d->Cout != 9 || !d->blend_cand || d->blend_cand_ctot < 9 || !d->blend_refined) return TCS_EINVAL;
            if (d->blend_delta && !d->blend_disp) return TCS_EINVAL;
            if (d->blend_flow16 && (d->blend_flow16_channel < 0 || d->blend_flow16_channel >= 8 * d->blend_flow16_groups)) return TCS_EINVAL;
            a.bl_cand = d->blend_cand; a.bl_cand_ctot = d->blend_cand_ctot; a.bl_disp = d->blend_disp;
            a.bl_refined = d->blend_refined; a.bl_delta = d->blend_delta; a.bl_coords1 = d->blend_coords1; a.bl_flow = d->blend_flow_x;
            a.bl_f16 = reinterpret_cast<_Float16*>(d->blend_flow16); a.bl_f16_groups = d->blend_flow16_groups; a.bl_f16_ch = d->blend_flow16_channel;
            if (d->blend_warm_pyr[0]) {
                if (!d->blend_warm_pyr[1] || !d->blend_warm_pyr[2] || !d->blend_warm_pyr[3] || d->blend_warm_radius < 0 || d->blend_warm_radius > 16 ||
                    d->W < 8) return TCS_EINVAL;
                a.warm_pyr0 = d->blend_warm_pyr[0]; a.warm_pyr1 = d->blend_warm_pyr[1]; a.warm_pyr2 = d->blend_warm_pyr[2];
                a.warm_pyr3 = d->blend_warm_pyr[3]; a.warm_radius = d->blend_warm_radius;
            }
            return launch_s16_cfg<1, 1, TCS_EPI_BLEND9>(a, cfg, s);
        case TCS_EPI_DECONV2X:
            if (d->ksize != 3 || !a.out16 || d->Cout % 32 != 0) return TCS_EINVAL;
            a.hidden = d->Cout / 4;
            return launch_s16_cfg<3, 1, TCS_EPI_DECONV2X>(a, cfg, s);
        case TCS_EPI_GRU_ZR:
            if (!a.h || !a.out16 || !a.out32 || d->Cout % 64 != 0) return TCS_EINVAL;
            a.hidden = d->Cout / 2;
            if (a.h_groups < a.hidden / 8) return TCS_EINVAL;
            return d->ksize == 3 ? launch_s16_cfg<3, 1, TCS_EPI_GRU_ZR>(a, cfg, s) : launch_s16_cfg<1, 1, TCS_EPI_GRU_ZR>(a, cfg, s);
        case TCS_EPI_GRU_Q:
            if (!a.h || !a.z || !a.out16 || d->Cout % 32 != 0) return TCS_EINVAL;
            a.hidden = d->Cout;
            if (a.h_groups < a.hidden / 8) return TCS_EINVAL;
            return d->ksize == 3 ? launch_s16_cfg<3, 1, TCS_EPI_GRU_Q>(a, cfg, s) : launch_s16_cfg<1, 1, TCS_EPI_GRU_Q>(a, cfg, s);
        default: return TCS_EINVAL;
    }
}

// Pair kernels exist for the instance combinations the refinement loop groups (core/update.py); any other combination, and
// anything the planner cannot take (ablation builds), runs as two ordinary launches: the results are the same either way.
static int s16_launch_pair(const S16Plan& p0, const S16Plan& p1, hipStream_t s) {
    constexpr int L = TCS_EPI_LINEAR;
    constexpr long long k3412 = s16_key(3, 1, 4, 1, 2, 1, L, 0, 1, false), k3812 = s16_key(3, 1, 8, 1, 2, 1, L, 0, 1, false),
                        k1422 = s16_key(1, 1, 4, 2, 2, 1, L, 0, 1, false), k3411 = s16_key(3, 1, 4, 1, 1, 1, L, 0, 1, false);
    if (p0.B == p1.B) {
        if (p0.key == k3412 && p1.key == k3412) return launch_s16_pair<3, 1, 4, 1, 2, L, 3, 1, 4, 1, 2, L>(p0, p1, s);
        // 4-row single-stage 3x3 tile (31 KiB) beside the 1x1 tile (40 KiB): every workgroup of the launch is allocated the LARGER of
        // the two LDS sizes, and with the 8-row two-stage 3x3 tile (80 KiB) the 1x1 half ran two workgroups per CU instead of four
        if (p0.key == k3411 && p1.key == k1422) return launch_s16_pair<3, 1, 4, 1, 1, L, 1, 1, 4, 2, 2, L>(p0, p1, s);
        if (p0.key == k1422 && p1.key == k3411) return launch_s16_pair<3, 1, 4, 1, 1, L, 1, 1, 4, 2, 2, L>(p1, p0, s);
        if (p0.key == k3812 && p1.key == k1422) return launch_s16_pair<3, 1, 8, 1, 2, L, 1, 1, 4, 2, 2, L>(p0, p1, s);
        if (p0.key == k1422 && p1.key == k3812) return launch_s16_pair<3, 1, 8, 1, 2, L, 1, 1, 4, 2, 2, L>(p1, p0, s);
    }
    return 1;           // no pair kernel
}

int tcs_conv2d_s16_group(const tcs_conv_s16_desc* const* descs, int n, tcs_stream_t stream) {
    if (!descs || n < 1 || n > 2) return TCS_EINVAL;
    if (n == 1) return tcs_conv2d_s16(descs[0], stream);
    S16Plan plan[2];
    for (int i = 0; i < 2; ++i) {
        plan[i].filled = false;
        g_s16_plan = &plan[i];
        const int rc = tcs_conv2d_s16(descs[i], stream);
        g_s16_plan = nullptr;
        if (rc != TCS_OK) return rc;
        if (!plan[i].filled) return TCS_EINVAL;
    }
    hipStream_t s = tcs_stream(stream);
    const int rc = s16_launch_pair(plan[0], plan[1], s);
    if (rc <= 0) return rc;
    for (int i = 0; i < 2; ++i) {
        const int r = plan[i].launch_alone(plan[i].args, s);
        if (r != TCS_OK) return r;
    }
    return TCS_OK;
}

int tcs_conv2d_s16_group_fused(const tcs_conv_s16_desc* const* descs, int n) {
    // diagnostic: 1 when tcs_conv2d_s16_group would issue these descriptors as ONE launch (tests assert the loop's pairs are)
    if (!descs || n != 2) return 0;
    S16Plan plan[2];
    for (int i = 0; i < 2; ++i) {
        plan[i].filled = false;
        g_s16_plan = &plan[i];
        const int rc = tcs_conv2d_s16(descs[i], nullptr);
        g_s16_plan = nullptr;
        if (rc != TCS_OK || !plan[i].filled) return 0;
    }
    constexpr int L = TCS_EPI_LINEAR;
    const long long k3412 = s16_key(3, 1, 4, 1, 2, 1, L, 0, 1, false), k3812 = s16_key(3, 1, 8, 1, 2, 1, L, 0, 1, false),
                    k1422 = s16_key(1, 1, 4, 2, 2, 1, L, 0, 1, false), k3411 = s16_key(3, 1, 4, 1, 1, 1, L, 0, 1, false);
    if (plan[0].B != plan[1].B) return 0;
    const long long a = plan[0].key, b = plan[1].key;
    return (a == k3412 && b == k3412) || (a == k3812 && b == k1422) || (a == k1422 && b == k3812) || (a == k3411 && b == k1422) ||
           (a == k1422 && b == k3411);
}

}  // extern "C"

// this translation unit's S16 domain flag (tcs_s16.h): read-and-clear for tcs_s16_flags()
int tcs_s16_flag_take_conv_s16(unsigned int* out) {
    unsigned int v = 0, zero = 0;
    if (hipMemcpyFromSymbol(&v, HIP_SYMBOL(TCS_S16_FLAG_VAR), sizeof(v)) != hipSuccess) return TCS_ELAUNCH;
    if (v && hipMemcpyToSymbol(HIP_SYMBOL(TCS_S16_FLAG_VAR), &zero, sizeof(zero)) != hipSuccess) return TCS_ELAUNCH;
    *out |= v;
    return TCS_OK;
}
