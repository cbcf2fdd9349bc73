"""Tensor-level wrappers over the C ABI: validate, allocate outputs with torch, launch on the
current HIP stream.  One function per operator of the reference's hot path (include/tcs_mi355.h
cites the reference lines).  No arithmetic happens here."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import native as nv

ACT = {"none": 0, "relu": 1, "sigmoid": 2, "tanh": 3, "leaky": 4, "relu_add_relu": 5}
EPI_LINEAR, EPI_GRU_ZR, EPI_GRU_Q = 0, 1, 2


def _new(like: torch.Tensor, *shape) -> torch.Tensor:
    return torch.empty(shape, dtype=torch.float32, device=like.device)


def _dims4(t: torch.Tensor, name: str):
    if t.ndim != 4:
        raise ValueError(f"{name}: expected a 4-D NCHW tensor, got shape {tuple(t.shape)}")
    return tuple(int(s) for s in t.shape)


# ---------------------------------------------------------------------------------------------
# correlation
# ---------------------------------------------------------------------------------------------
@dataclass
class CorrPyramid:
    """Skewed 4-level pyramid in HBM (layout: include/tcs_mi355.h) plus what the build left behind."""
    levels: List[torch.Tensor]
    B: int
    H: int
    W: int
    workspace: torch.Tensor                       # holds the natural-layout level 0 [B,H,W,W]
    natural: Optional[List[torch.Tensor]] = None  # [B,H,W,W>>i], i = 0..3 (tests / API parity)
    cost_volume: Optional[torch.Tensor] = None    # [B,W,H,W]
    sparse: Optional[tuple] = None                # (disp, cost, mask), each [B,1,H,W]


def corr_build(fmap1: torch.Tensor, fmap2: torch.Tensor, argmax: bool = False, cost_volume: bool = False,
               natural: bool = False) -> CorrPyramid:
    B, Cc, H, W = _dims4(fmap1, "fmap1")
    if tuple(fmap2.shape) != (B, Cc, H, W):
        raise ValueError(f"fmap2 shape {tuple(fmap2.shape)} != fmap1 shape {tuple(fmap1.shape)}")
    L = nv.lib()
    ws_bytes = L.tcs_corr_build_workspace_bytes(B, H, W)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=fmap1.device)
    levels = [_new(fmap1, B, H, W >> i, W) for i in range(4)]
    nat = [None] * 4
    if natural:
        nat = [None] + [_new(fmap1, B, H, W, W >> i) for i in range(1, 4)]
    cost = _new(fmap1, B, W, H, W) if cost_volume else None
    sp = tuple(_new(fmap1, B, 1, H, W) for _ in range(3)) if argmax else (None, None, None)
    rc = L.tcs_corr_build(nv.ptr(fmap1, "fmap1"), nv.ptr(fmap2, "fmap2"), B, Cc, H, W,
                          *[nv.ptr(t) for t in levels], nv.ptr(nat[1]), nv.ptr(nat[2]), nv.ptr(nat[3]), nv.ptr(cost),
                          nv.ptr(sp[0]), nv.ptr(sp[1]), nv.ptr(sp[2]), nv.ptr(ws), nv.stream())
    nv.check(rc, "tcs_corr_build")
    out = CorrPyramid(levels, B, H, W, ws, cost_volume=cost, sparse=sp if argmax else None)
    if natural:
        out.natural = [ws[: B * H * W * W].view(B, H, W, W)] + nat[1:]
    return out


def corr_lookup(pyr: CorrPyramid, coords: torch.Tensor, radius: int = 4, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    if tuple(coords.shape) != (pyr.B, 1, pyr.H, pyr.W):
        raise ValueError(f"coords shape {tuple(coords.shape)} != {(pyr.B, 1, pyr.H, pyr.W)}")
    n_ch = 4 * (2 * radius + 1)
    if out is None:
        out = _new(coords, pyr.B, n_ch, pyr.H, pyr.W)
    elif tuple(out.shape) != (pyr.B, n_ch, pyr.H, pyr.W):
        raise ValueError("corr_lookup: bad `out` shape")
    stamps = LOOKUP_PROBE.next_slot(pyr) if LOOKUP_PROBE is not None else None
    rc = nv.lib().tcs_corr_lookup(*[nv.ptr(t) for t in pyr.levels], nv.ptr(coords, "coords"), pyr.B, pyr.H, pyr.W, radius,
                                  nv.ptr(out, "out"), stamps, nv.stream())
    nv.check(rc, "tcs_corr_lookup")
    return out


class LookupProbe:
    """Measurement hook for bench.py: hands each lookup launch its own slot of a device buffer in
    which the kernel's workgroups record the device wall clock at start/end (include/tcs_mi355.h,
    tcs_corr_lookup `stamps`).  Works under HIP-graph capture: the slot pointer is baked into the
    captured launch, so every replay refreshes the same `slots` launches."""

    def __init__(self, device, slots: int = 64):
        self.device, self.slots, self.buf, self.blocks, self.count, self.pixels = device, slots, None, 0, 0, 0

    def next_slot(self, pyr):
        blocks = nv.lib().tcs_corr_lookup_blocks(pyr.B, pyr.H, pyr.W)
        if self.buf is None or blocks != self.blocks:
            self.blocks, self.pixels = blocks, pyr.B * pyr.H * pyr.W
            self.buf = torch.zeros(self.slots, blocks, 2, dtype=torch.int64, device=self.device)
        slot = self.count % self.slots
        self.count += 1
        return self.buf[slot].data_ptr()

    def reset(self):
        """Enqueue a clear of the stamps (end stamps are accumulated with atomicMax)."""
        if self.buf is not None:
            self.buf.zero_()

    def durations_us(self, snapshot=None):
        """Per-launch durations of the slots written since the last reset (100 MHz clock -> 10 ns ticks)."""
        buf = (self.buf if snapshot is None else snapshot).cpu()
        start, end = buf[..., 0], buf[..., 1]
        used = (end > 0).any(dim=1)
        big = torch.iinfo(torch.int64).max
        s = torch.where(start > 0, start, torch.full_like(start, big)).min(dim=1).values
        e = end.max(dim=1).values
        return ((e - s)[used].double() * 0.01).tolist()


LOOKUP_PROBE: Optional[LookupProbe] = None


def pose_prepare(K, T=None, T_prev=None, scale: float = 0.25):
    """-> (K_scaled, K_scaled_inv, T_rel, T_back); the last two are None without poses (tcs_pose_prepare)."""
    B = int(K.shape[0])
    K = K.reshape(B, 3, 3).float().contiguous()
    ks, ksi = torch.empty_like(K), torch.empty_like(K)
    trel = tback = None
    if T is not None:
        T, T_prev = T.reshape(B, 4, 4).float().contiguous(), T_prev.reshape(B, 4, 4).float().contiguous()
        trel, tback = torch.empty_like(T), torch.empty_like(T)
    nv.check(nv.lib().tcs_pose_prepare(nv.ptr(K, "K"), nv.ptr(T, "T"), nv.ptr(T_prev, "previous_T"), float(scale), B,
                                       nv.ptr(ks), nv.ptr(ksi), nv.ptr(trel), nv.ptr(tback), nv.stream()), "tcs_pose_prepare")
    return ks, ksi, trel, tback


# ---------------------------------------------------------------------------------------------
# temporal warp
# ---------------------------------------------------------------------------------------------
def _cam(T_rel, K, K_inv, baseline, B):
    T_rel = T_rel.reshape(B, 4, 4).float().contiguous()
    K = K.reshape(B, 3, 3).float().contiguous()
    K_inv = K_inv.reshape(B, 3, 3).float().contiguous()
    baseline = baseline.reshape(-1).float().contiguous()
    if baseline.numel() != B:
        raise ValueError(f"baseline has {baseline.numel()} entries for batch {B}")
    return T_rel, K, K_inv, baseline


def warp_forward(prev_disp, prev_fmap, T_rel, K, K_inv, baseline, cur_fmap=None, want_fmap=True):
    """-> (disp [B,1,H,W], fmap [B,C,H,W] or None, mask [B,1,H,W], cost [B,1,H,W] or None)"""
    B, Cc, H, W = _dims4(prev_fmap, "prev_fmap")
    if tuple(prev_disp.shape) != (B, 1, H, W):
        raise ValueError("prev_disp must be [B,1,H,W] matching prev_fmap")
    T_rel, K, K_inv, baseline = _cam(T_rel, K, K_inv, baseline, B)
    L = nv.lib()
    ws = torch.empty(L.tcs_warp_workspace_bytes(B, Cc, H, W) // 4, dtype=torch.float32, device=prev_fmap.device)
    o_disp, o_mask = _new(prev_disp, B, 1, H, W), _new(prev_disp, B, 1, H, W)
    o_fmap = _new(prev_fmap, B, Cc, H, W) if want_fmap else None
    o_cost = _new(prev_disp, B, 1, H, W) if cur_fmap is not None else None
    rc = L.tcs_warp_forward(nv.ptr(prev_disp, "prev_disp"), nv.ptr(prev_fmap, "prev_fmap"), nv.ptr(T_rel), nv.ptr(K), nv.ptr(K_inv),
                            nv.ptr(baseline), B, Cc, H, W, nv.ptr(o_disp), nv.ptr(o_fmap), nv.ptr(o_mask),
                            nv.ptr(cur_fmap, "cur_fmap"), nv.ptr(o_cost), nv.ptr(ws), nv.stream())
    nv.check(rc, "tcs_warp_forward")
    return o_disp, o_fmap, o_mask, o_cost


def warp_geometry(prev_disp, T_rel, K, K_inv, baseline):
    B, _, H, W = _dims4(prev_disp, "prev_disp")
    T_rel, K, K_inv, baseline = _cam(T_rel, K, K_inv, baseline, B)
    L = nv.lib()
    ws = torch.empty(L.tcs_warp_workspace_bytes(B, 0, H, W) // 4, dtype=torch.float32, device=prev_disp.device)
    cd, va, me = (_new(prev_disp, B, 1, H, W) for _ in range(3))
    fl = _new(prev_disp, B, 2, H, W)
    rc = L.tcs_warp_geometry(nv.ptr(prev_disp, "prev_disp"), nv.ptr(T_rel), nv.ptr(K), nv.ptr(K_inv), nv.ptr(baseline), B, H, W,
                             nv.ptr(cd), nv.ptr(va), nv.ptr(fl), nv.ptr(me), nv.ptr(ws), nv.stream())
    nv.check(rc, "tcs_warp_geometry")
    return cd, va, fl, me


def softsplat_sum(inp, flow):
    B, Cc, H, W = _dims4(inp, "tenIn")
    if tuple(flow.shape) != (B, 2, H, W):
        raise ValueError("tenFlow must be [B,2,H,W]")
    out = torch.zeros_like(inp)
    nv.check(nv.lib().tcs_softsplat_sum(nv.ptr(inp, "tenIn"), nv.ptr(flow, "tenFlow"), B, Cc, H, W, nv.ptr(out), nv.stream()),
             "tcs_softsplat_sum")
    return out


def backward_grid(disp, T_rel, K, K_inv, baseline):
    B, _, H, W = _dims4(disp, "disp")
    T_rel, K, K_inv, baseline = _cam(T_rel, K, K_inv, baseline, B)
    grid = _new(disp, B, 2, H, W)
    nv.check(nv.lib().tcs_backward_grid(nv.ptr(disp, "disp"), nv.ptr(T_rel), nv.ptr(K), nv.ptr(K_inv), nv.ptr(baseline), B, H, W,
                                        nv.ptr(grid), nv.stream()), "tcs_backward_grid")
    return grid


def bilinear_sample(img, grid):
    B, Cc, Hi, Wi = _dims4(img, "img")
    Bg, two, Ho, Wo = _dims4(grid, "grid")
    if Bg != B or two != 2:
        raise ValueError("grid must be [B,2,Ho,Wo]")
    out = _new(img, B, Cc, Ho, Wo)
    nv.check(nv.lib().tcs_bilinear_sample(nv.ptr(img, "img"), nv.ptr(grid, "grid"), B, Cc, Hi, Wi, Ho, Wo, nv.ptr(out), nv.stream()),
             "tcs_bilinear_sample")
    return out


def grid_halve(grid):
    B, _, H, W = _dims4(grid, "grid")
    out = _new(grid, B, 2, H // 2, W // 2)
    nv.check(nv.lib().tcs_grid_halve(nv.ptr(grid, "grid"), B, H, W, nv.ptr(out), nv.stream()), "tcs_grid_halve")
    return out


# ---------------------------------------------------------------------------------------------
# stencils
# ---------------------------------------------------------------------------------------------
def flow_step(coords1, delta, disp_q=None):
    B, _, H, W = _dims4(coords1, "coords1")
    if disp_q is None:
        disp_q = torch.empty_like(coords1)
    nv.check(nv.lib().tcs_flow_step(nv.ptr(coords1, "coords1"), nv.ptr(delta, "delta"), B, H, W, nv.ptr(disp_q), nv.stream()),
             "tcs_flow_step")
    return disp_q


def flow_step_grads(coords1, delta, scale: float = 1.0):
    """(disp_q, scale * gradient_xy(disp_q), grad_candidates(disp_q)) with disp_q = x - (coords1 + delta), one launch."""
    B, _, H, W = _dims4(coords1, "coords1")
    if tuple(delta.shape) != (B, 1, H, W):
        raise ValueError("flow_step_grads: bad delta shape")
    disp_q, grad, cands = torch.empty_like(coords1), _new(coords1, B, 2, H, W), _new(coords1, B, 32, H, W)
    nv.check(nv.lib().tcs_flow_step_grads(nv.ptr(coords1, "coords1"), nv.ptr(delta, "delta"), B, H, W, float(scale), nv.ptr(disp_q),
                                          nv.ptr(grad), nv.ptr(cands), nv.stream()), "tcs_flow_step_grads")
    return disp_q, grad, cands


def disp_gradient_xy(disp, scale: float = 1.0, out=None):
    B, _, H, W = _dims4(disp, "disp")
    out = _new(disp, B, 2, H, W) if out is None else out
    nv.check(nv.lib().tcs_disp_gradient_xy(nv.ptr(disp, "disp"), B, H, W, float(scale), nv.ptr(out), nv.stream()), "tcs_disp_gradient_xy")
    return out


def grad_candidates(disp, out=None):
    B, _, H, W = _dims4(disp, "disp")
    out = _new(disp, B, 32, H, W) if out is None else out
    nv.check(nv.lib().tcs_grad_candidates(nv.ptr(disp, "disp"), B, H, W, nv.ptr(out), nv.stream()), "tcs_grad_candidates")
    return out


def propagate_disparity(grad, disp, out=None):
    B, _, H, W = _dims4(disp, "disp")
    out = _new(disp, B, 27, H, W) if out is None else out
    nv.check(nv.lib().tcs_propagate_disparity(nv.ptr(grad, "grad"), nv.ptr(disp, "disp"), B, H, W, nv.ptr(out), nv.stream()),
             "tcs_propagate_disparity")
    return out


def softmax_blend(logits9, cand, disp_q=None, want_delta=False, coords1=None, refined=None, flow_x=None, flow_x_channel=None):
    """`flow_x` [B,1,H,W] and `flow_x_channel` (a one-channel slice t[:, c:c+1] of a contiguous [B,C,H,W] tensor) receive
    coords1 - x, the next iteration's motion-encoder input."""
    B, _, H, W = _dims4(logits9, "logits")
    refined = _new(logits9, B, 1, H, W) if refined is None else refined
    delta = _new(logits9, B, 1, H, W) if want_delta else None
    ch_ptr, ch_stride = None, 0
    if flow_x_channel is not None:
        if tuple(flow_x_channel.shape) != (B, 1, H, W) or flow_x_channel.stride()[2:] != (W, 1) or flow_x_channel.dtype != torch.float32:
            raise ValueError("flow_x_channel must be a [B,1,H,W] float32 channel slice with dense rows")
        if not flow_x_channel.is_cuda:
            raise RuntimeError("flow_x_channel: CPU tensor (the hot path has no CPU fallback)")
        ch_ptr, ch_stride = flow_x_channel.data_ptr(), int(flow_x_channel.stride()[0]) if B > 1 else H * W
    nv.check(nv.lib().tcs_softmax_blend(nv.ptr(logits9, "logits"), nv.ptr(cand, "cand"), int(cand.shape[1]), nv.ptr(disp_q), B, H, W,
                                        nv.ptr(refined), nv.ptr(delta), nv.ptr(coords1), nv.ptr(flow_x), ch_ptr, ch_stride, nv.stream()),
             "tcs_softmax_blend")
    return refined, delta


def convex_upsample(disp, mask, clip=True):
    B, _, H, W = _dims4(disp, "disp")
    if tuple(mask.shape) != (B, 144, H, W):
        raise ValueError("mask must be [B,144,H,W] (factor 4)")
    up, fq = _new(disp, B, 1, 4 * H, 4 * W), _new(disp, B, 1, H, W)
    nv.check(nv.lib().tcs_convex_upsample(nv.ptr(disp, "disp"), nv.ptr(mask, "mask"), B, H, W, int(clip), nv.ptr(up), nv.ptr(fq), nv.stream()),
             "tcs_convex_upsample")
    return up, fq


def avgpool3s2(x, out=None):
    B, Cc, H, W = _dims4(x, "x")
    out = _new(x, B, Cc, (H - 1) // 2 + 1, (W - 1) // 2 + 1) if out is None else out
    nv.check(nv.lib().tcs_avgpool3s2(nv.ptr(x, "x"), B, Cc, H, W, nv.ptr(out), nv.stream()), "tcs_avgpool3s2")
    return out


def resize_bilinear(x, Ho: int, Wo: int, out=None):
    B, Cc, H, W = _dims4(x, "x")
    out = _new(x, B, Cc, Ho, Wo) if out is None else out
    nv.check(nv.lib().tcs_resize_bilinear(nv.ptr(x, "x"), B, Cc, H, W, Ho, Wo, nv.ptr(out), nv.stream()), "tcs_resize_bilinear")
    return out


# ---------------------------------------------------------------------------------------------
# convolutions on the matrix cores
# ---------------------------------------------------------------------------------------------
MATH_F32, MATH_F16X3 = 0, 1


@dataclass
class PackedConv:
    weight: torch.Tensor           # kernel layout (tcs_pack_conv_weight / tcs_pack_conv_weight_f16x3)
    bias: Optional[torch.Tensor]
    cout: int
    cin: int
    ksize: int
    math: int = MATH_F32
    unscale: float = 1.0


def pack_conv(weight: torch.Tensor, bias: Optional[torch.Tensor], math: str = "f32") -> PackedConv:
    """math='f32': fp32 MFMA kernel.  math='f16x3': fp16 hi/lo split kernel (fp32-equivalent accuracy);
    falls back to 'f32' for the shapes the split kernel does not cover (Cin == 1, 7x7)."""
    cout, cin, kh, kw = (int(s) for s in weight.shape)
    if kh != kw:
        raise ValueError("square kernels only")
    L = nv.lib()
    w = weight.detach().float().contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    if math == "f16x3" and cin > 1 and kh in (1, 3):
        n = L.tcs_conv_packed_floats_f16x3(cout, cin, kh)
        wmax = float(w.abs().max())
        s_log2 = 0 if wmax == 0.0 else int(12 - math_floor_log2(wmax))           # max|w| * 2^s in [2^12, 2^13)
        s_log2 = max(-40, min(40, s_log2))
        packed = torch.empty(n, dtype=torch.float32, device=w.device)
        nv.check(L.tcs_pack_conv_weight_f16x3(nv.ptr(w, "weight"), cout, cin, kh, s_log2, nv.ptr(packed), nv.stream()),
                 "tcs_pack_conv_weight_f16x3")
        return PackedConv(packed, b, cout, cin, kh, MATH_F16X3, 2.0 ** (-s_log2))
    if math not in ("f32", "f16x3"):
        raise ValueError(f"unknown math mode {math!r}")
    n = L.tcs_conv_packed_floats(cout, cin, kh)
    if n == 0:
        raise ValueError(f"unsupported convolution [{cout},{cin},{kh},{kw}]")
    packed = torch.empty(n, dtype=torch.float32, device=w.device)
    nv.check(L.tcs_pack_conv_weight(nv.ptr(w, "weight"), cout, cin, kh, nv.ptr(packed), nv.stream()), "tcs_pack_conv_weight")
    return PackedConv(packed, b, cout, cin, kh)


def pack_deconv4x4s2(weight: torch.Tensor) -> PackedConv:
    """nn.ConvTranspose2d(Cin, Cout, 4, stride=2, padding=1, bias=False) -> the fp16-split layout of the equivalent
    3x3 convolution with 4*Cout parity-grouped outputs (tcs_pack_deconv4x4s2_f16x3); use with `deconv4x4s2`."""
    cin, cout, kh, kw = (int(s) for s in weight.shape)
    if (kh, kw) != (4, 4):
        raise ValueError("4x4 transposed convolutions only")
    L = nv.lib()
    w = weight.detach().float().contiguous()
    wmax = float(w.abs().max())
    s_log2 = 0 if wmax == 0.0 else max(-40, min(40, int(12 - math_floor_log2(wmax))))
    packed = torch.empty(L.tcs_deconv_packed_floats_f16x3(cin, cout), dtype=torch.float32, device=w.device)
    scratch = torch.empty(4 * cout * cin * 9, dtype=torch.float32, device=w.device)
    nv.check(L.tcs_pack_deconv4x4s2_f16x3(nv.ptr(w, "weight"), cin, cout, s_log2, nv.ptr(packed), nv.ptr(scratch), nv.stream()),
             "tcs_pack_deconv4x4s2_f16x3")
    return PackedConv(packed, None, 4 * cout, cin, 3, MATH_F16X3, 2.0 ** (-s_log2))


def deconv4x4s2(pc: PackedConv, srcs: Sequence[torch.Tensor], out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[B,Cin,H,W] -> [B,Cout,2H,2W] (ConvTranspose2d k=4, s=2, p=1, no bias)."""
    d = _desc(pc, srcs)
    cout = pc.cout // 4
    if out is None:
        out = _new(srcs[0], d.B, cout, 2 * d.H, 2 * d.W)
    d.epilogue, d.act = 3, 0
    d.out, d.out_ctot, d.out_coff = nv.ptr(out, "out"), cout, 0
    nv.check(nv.lib().tcs_conv2d(C.byref(d), nv.stream()), "tcs_conv2d[deconv2x]")
    return out


def instance_norm(x: torch.Tensor, act: str = "none", addend: Optional[torch.Tensor] = None, eps: float = 1e-5,
                  out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """act(InstanceNorm2d(x)) + addend (affine=False, biased variance)."""
    B, Cc, H, W = _dims4(x, "x")
    if addend is not None and tuple(addend.shape) != (B, Cc, H, W):
        raise ValueError("instance_norm: bad addend shape")
    out = torch.empty_like(x) if out is None else out
    nv.check(nv.lib().tcs_instance_norm(nv.ptr(x, "x"), B, Cc, H, W, float(eps), ACT[act], nv.ptr(addend, "addend"), nv.ptr(out),
                                        nv.stream()), "tcs_instance_norm")
    return out


def math_floor_log2(x: float) -> int:
    import math as _m
    return int(_m.floor(_m.log2(x)))


def _desc(pc: PackedConv, srcs: Sequence[torch.Tensor]) -> nv.ConvDesc:
    if not 1 <= len(srcs) <= 4:
        raise ValueError("1..4 sources")
    B, _, H, W = _dims4(srcs[0], "src0")
    d = nv.ConvDesc()
    tot = 0
    for i, s in enumerate(srcs):
        bs, cs, hs, ws_ = _dims4(s, f"src{i}")
        if (bs, hs, ws_) != (B, H, W):
            raise ValueError(f"src{i} shape {tuple(s.shape)} does not match src0 {tuple(srcs[0].shape)}")
        d.src[i] = nv.ptr(s, f"src{i}")
        d.src_ch[i] = cs
        tot += cs
    if tot != pc.cin:
        raise ValueError(f"sources carry {tot} channels, convolution expects {pc.cin}")
    d.n_src = len(srcs)
    d.weight = nv.ptr(pc.weight)
    d.bias = nv.ptr(pc.bias)
    d.B, d.H, d.W, d.Cin, d.Cout, d.ksize = B, H, W, pc.cin, pc.cout, pc.ksize
    d.post_scale = 1.0
    d.math, d.weight_unscale = pc.math, pc.unscale
    return d


_GROUP = None            # descriptors of the `with grouped():` block being recorded (LINEAR tcs_conv2d launches)


class grouped:
    """`with ops.grouped(): conv2d(...); conv2d(...)` — two INDEPENDENT tcs_conv2d layers issued as one launch at the end of the block
    where the library has a grouped kernel for them (tcs_conv2d_group), otherwise one after the other; same results either way.
    The fp32-tensor counterpart of tcs_mi355.s16.grouped."""

    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __enter__(self):
        global _GROUP
        if self.enabled:
            if _GROUP is not None:
                raise RuntimeError("ops.grouped() does not nest")
            _GROUP = []
        return self

    def __exit__(self, et, ev, tb):
        global _GROUP
        if not self.enabled:
            return False
        descs, _GROUP = _GROUP, None
        if et is not None or not descs:
            return False
        for i in range(0, len(descs), 2):
            chunk = descs[i:i + 2]
            arr = (C.POINTER(nv.ConvDesc) * len(chunk))(*[C.pointer(d) for d, _ in chunk])
            nv.check(nv.lib().tcs_conv2d_group(arr, len(chunk), nv.stream()), "tcs_conv2d_group[" + " | ".join(n for _, n in chunk) + "]")
        return False


def _launch_conv(d, name: str):
    if _GROUP is not None:
        _GROUP.append((d, name))
    else:
        nv.check(nv.lib().tcs_conv2d(C.byref(d), nv.stream()), name)


def conv2d(pc: PackedConv, srcs: Sequence[torch.Tensor], act: str = "none", addend=None, post_scale: float = 1.0,
           out: Optional[torch.Tensor] = None, out_coff: int = 0, stride: int = 1, out16=None, out16_group_offset: int = 0,
           image_pair: Optional[torch.Tensor] = None, in_transform: int = 0):
    """`out16` (a tcs_mi355.s16.S16): write the result in pre-split form for tcs_conv2d_s16 consumers INSTEAD of fp32 NCHW
    (returns out16); stride 1 only.  7x7 RGB stem only: `image_pair` = the right images, appended to srcs[0] along the batch
    (torch.cat((image1, image2), 0) without the copy), `in_transform=1` = samples read as 2 * (x / 255) - 1 (tc_stereo.py:101-107)."""
    d = _desc(pc, srcs)
    if image_pair is not None:
        if len(srcs) != 1 or tuple(image_pair.shape[1:]) != tuple(srcs[0].shape[1:]):
            raise ValueError("conv2d: `image_pair` must match the single source's [C,H,W]")
        d.src_batch2, d.batch_split = nv.ptr(image_pair, "image_pair"), d.B
        d.B = d.B + int(image_pair.shape[0])
    d.in_transform = int(in_transform)
    if out16 is not None:
        if stride != 1 or (out16.B, out16.H, out16.W) != (d.B, d.H, d.W):
            raise ValueError("conv2d: bad `out16`")
        if addend is not None and tuple(addend.shape) != (d.B, pc.cout, d.H, d.W):
            raise ValueError("conv2d: bad addend shape")
        d.stride = 1
        d.epilogue, d.act, d.post_scale = EPI_LINEAR, ACT[act], float(post_scale)
        d.addend = nv.ptr(addend, "addend")
        d.out16, d.out16_groups, d.out16_group_offset = out16.ptr(), out16.G, int(out16_group_offset)
        _launch_conv(d, "tcs_conv2d[s16 out]")
        return out16
    if stride not in (1, 2):
        raise ValueError("stride 1 or 2")
    if stride == 2 and (pc.math != MATH_F16X3 or pc.ksize != 3):
        raise NotImplementedError("stride-2 convolutions run on the fp16-split kernel (3x3 only)")
    Ho, Wo = ((d.H - 1) // 2 + 1, (d.W - 1) // 2 + 1) if stride == 2 else (d.H, d.W)
    d.stride = stride
    if out is None:
        out = _new(srcs[0], d.B, pc.cout, Ho, Wo)
    if out.shape[0] != d.B or tuple(out.shape[2:]) != (Ho, Wo):
        raise ValueError("conv2d: bad `out` shape")
    if addend is not None and tuple(addend.shape) != (d.B, pc.cout, Ho, Wo):
        raise ValueError("conv2d: bad addend shape")
    d.epilogue, d.act, d.post_scale = EPI_LINEAR, ACT[act], float(post_scale)
    d.addend = nv.ptr(addend, "addend")
    d.out, d.out_ctot, d.out_coff = nv.ptr(out, "out"), int(out.shape[1]), int(out_coff)
    _launch_conv(d, "tcs_conv2d")
    return out


def conv3x3_cout1(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor]) -> torch.Tensor:
    """Conv2d(Cin, 1, 3, padding=1) on the un-packed weight (tcs_conv3x3_cout1)."""
    B, Cin, H, W = _dims4(x, "x")
    if tuple(weight.shape) != (1, Cin, 3, 3):
        raise ValueError(f"expected a [1,{Cin},3,3] weight, got {tuple(weight.shape)}")
    out = _new(x, B, 1, H, W)
    w = weight.detach().float().contiguous()
    b = None if bias is None else bias.detach().float().contiguous()
    nv.check(nv.lib().tcs_conv3x3_cout1(nv.ptr(x, "x"), nv.ptr(w, "weight"), nv.ptr(b, "bias"), B, Cin, H, W, nv.ptr(out), nv.stream()),
             "tcs_conv3x3_cout1")
    return out


def gru_gates(pc_zr: PackedConv, srcs, h, cz=None, cr=None, z_out=None, rh_out=None):
    """z = sigmoid(conv_zr[:hid] + cz), rh = sigmoid(conv_zr[hid:] + cr) * h   (update.py:81-83, 30-33)."""
    d = _desc(pc_zr, srcs)
    hid = pc_zr.cout // 2
    if tuple(h.shape) != (d.B, hid, d.H, d.W):
        raise ValueError("gru_gates: bad h shape")
    z_out = torch.empty_like(h) if z_out is None else z_out
    rh_out = torch.empty_like(h) if rh_out is None else rh_out
    d.epilogue = EPI_GRU_ZR
    d.addend, d.addend2, d.h = nv.ptr(cz, "cz"), nv.ptr(cr, "cr"), nv.ptr(h, "h")
    d.out, d.out2, d.out_ctot, d.out_coff = nv.ptr(z_out, "z"), nv.ptr(rh_out, "rh"), hid, 0
    nv.check(nv.lib().tcs_conv2d(C.byref(d), nv.stream()), "tcs_conv2d[gru_zr]")
    return z_out, rh_out


def gru_update(pc_q: PackedConv, srcs, h, z, cq=None, keep_z: bool = False, out=None):
    """q = tanh(conv_q + cq); h' = (1-z)h + zq (keep_z=False, update.py:85) or zh + (1-z)q (update.py:34,66)."""
    d = _desc(pc_q, srcs)
    if tuple(h.shape) != (d.B, pc_q.cout, d.H, d.W) or z.shape != h.shape:
        raise ValueError("gru_update: bad h/z shape")
    out = torch.empty_like(h) if out is None else out
    d.epilogue = EPI_GRU_Q
    d.addend, d.h, d.z, d.blend_keep_z = nv.ptr(cq, "cq"), nv.ptr(h, "h"), nv.ptr(z, "z"), int(keep_z)
    d.out, d.out_ctot, d.out_coff = nv.ptr(out, "out"), pc_q.cout, 0
    nv.check(nv.lib().tcs_conv2d(C.byref(d), nv.stream()), "tcs_conv2d[gru_q]")
    return out
