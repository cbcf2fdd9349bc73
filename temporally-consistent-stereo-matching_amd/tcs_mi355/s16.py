"""Pre-split activations ("S16" tensors) and the convolutions that consume them (include/tcs_mi355.h,
csrc/tcs_conv_s16.hip).

Inside the refinement loop every activation that feeds a convolution lives in the operand form of the
fp16-split contraction: `_Float16 [B][G][2][H+2][W+2][8]` — groups of 8 channels, {hi, lo} planes with
x = hi + lo, a one-pixel zero border.  The producer splits once (conv epilogue, stencil kernel); the consumer
copies 16-byte units straight into LDS.  PyTorch only owns the memory: buffers come from a per-model pool
(`S16Pool`), zero-filled once so that the border stays zero for the life of the model, and are reused every
iteration and frame (also what makes the frame capturable without memset nodes).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import native as nv
from .ops import ACT, EPI_GRU_Q, EPI_GRU_ZR, EPI_LINEAR, MATH_F16X3, PackedConv

EPI_DECONV2X = 3
EPI_BLEND9 = 4


def groups_for(C_: int) -> int:
    return (C_ + 15) // 16 * 2


# ---- grouped launches (tcs_conv2d_s16_group) -------------------------------------------------------------------------
_GROUP: Optional[list] = None        # descriptors of the `with grouped():` block being recorded


class grouped:
    """`with s16.grouped(): a = conv2d(...); b = conv2d(...)` — the (two) tcs_conv2d_s16 launches made inside the block are
    INDEPENDENT layers (neither reads what the other writes) and go out as one launch at the end of the block where the library
    has a pair kernel for their tile instances, otherwise one after the other; the tensors the calls return are valid after the
    block.  What this replaces is a fork / join of two graph branches (tcs_mi355/streams.py): the pair keeps the concurrency
    without the cross-queue dependencies.  `enabled=False` (A/B runs): launches happen at once, as without the block."""

    def __init__(self, enabled: bool = True, report: bool = False):
        self.enabled, self.report = enabled, report
        self.fused: List[bool] = []          # with `report`: per pair, whether the library issued it as one launch

    def __enter__(self):
        global _GROUP
        if self.enabled:
            if _GROUP is not None:
                raise RuntimeError("s16.grouped() does not nest")
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
            arr = (C.POINTER(nv.ConvS16Desc) * len(chunk))(*[C.pointer(d) for d, _ in chunk])
            if self.report:
                self.fused.append(len(chunk) == 2 and bool(nv.lib().tcs_conv2d_s16_group_fused(arr, 2)))
            nv.check(nv.lib().tcs_conv2d_s16_group(arr, len(chunk), nv.stream()), "tcs_conv2d_s16_group[" + " | ".join(n for _, n in chunk) + "]")
        return False


def _launch(d, name: str):
    """tcs_conv2d_s16 now, or at the end of the enclosing `grouped()` block."""
    if _GROUP is not None:
        _GROUP.append((d, name))
    else:
        nv.check(nv.lib().tcs_conv2d_s16(C.byref(d), nv.stream()), name)




@dataclass
class S16:
    data: torch.Tensor          # float16 [B, G, 2, H+2, W+2, 8]
    C: int                      # logical channels

    @property
    def B(self):
        return int(self.data.shape[0])

    @property
    def G(self):
        return int(self.data.shape[1])

    @property
    def H(self):
        return int(self.data.shape[3]) - 2

    @property
    def W(self):
        return int(self.data.shape[4]) - 2

    @property
    def device(self):
        return self.data.device

    def ptr(self):
        return nv.ptr(self.data, "s16", dtype=torch.float16)

    def float(self) -> torch.Tensor:
        """fp32 NCHW copy (hi + lo) — the loop's boundary and tests."""
        return from_s16(self)


def zeros(B: int, C_: int, H: int, W: int, device, groups: Optional[int] = None) -> S16:
    G = groups_for(C_) if groups is None else int(groups)
    return S16(torch.zeros(B, G, 2, H + 2, W + 2, 8, dtype=torch.float16, device=device), C_)


class S16Pool:
    """Named, zero-initialised S16 buffers that live as long as the model: `get(key, B, C, H, W)` returns the same
    buffer for the same key and shape.  Producers write interior pixels of real channel groups only, so borders and
    padding channels stay zero without any per-frame memset."""

    def __init__(self):
        self.buffers: Dict[tuple, S16] = {}

    def get(self, key, B, C_, H, W, device, groups=None) -> S16:
        G = groups_for(C_) if groups is None else int(groups)
        k = (key, B, C_, H, W, G, str(device))
        buf = self.buffers.get(k)
        if buf is None:
            buf = zeros(B, C_, H, W, device, G)
            self.buffers[k] = buf
        return buf

    def get32(self, key, shape, device, zero: bool = False) -> torch.Tensor:
        """A persistent fp32 scratch tensor (e.g. a GRU's update gate).  Like the S16 buffers it is never freed while the
        model lives: captured HIP graphs hold raw pointers to it, and a freed block could be handed to another tensor.
        `zero`: zero-filled at allocation (ticket counters of the fused InstanceNorm statistics)."""
        k = (key, tuple(int(v) for v in shape), "f32", str(device))
        buf = self.buffers.get(k)
        if buf is None:
            buf = self.buffers[k] = (torch.zeros if zero else torch.empty)(*k[1], dtype=torch.float32, device=device)
        return buf

    def get_i64(self, key, shape, device) -> torch.Tensor:
        """A persistent zero-initialised int64 tensor (the fixed-point InstanceNorm sums of the fused transposed convolutions)."""
        k = (key, tuple(int(v) for v in shape), "i64", str(device))
        buf = self.buffers.get(k)
        if buf is None:
            buf = self.buffers[k] = torch.zeros(*k[1], dtype=torch.int64, device=device)
        return buf

    def bytes(self) -> int:
        return sum((b.data if isinstance(b, S16) else b).numel() * (b.data if isinstance(b, S16) else b).element_size() for b in self.buffers.values())


def to_s16(x: torch.Tensor, out: Optional[S16] = None, group_offset: int = 0) -> S16:
    """fp32 [B,C,H,W] -> S16 (into groups [group_offset, ...) of `out` when given: a virtual torch.cat)."""
    if x.ndim != 4:
        raise ValueError("to_s16 expects [B,C,H,W]")
    B, C_, H, W = (int(v) for v in x.shape)
    if out is None:
        out = zeros(B, C_, H, W, x.device)
    elif (out.B, out.H, out.W) != (B, H, W):
        raise ValueError("to_s16: `out` has another shape")
    nv.check(nv.lib().tcs_s16_from_f32(nv.ptr(x, "x"), B, C_, H, W, out.ptr(), out.G, int(group_offset), nv.stream()), "tcs_s16_from_f32")
    return out


def from_s16(s: S16, C_: Optional[int] = None, group_offset: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    C_ = s.C if C_ is None else int(C_)
    if out is None:
        out = torch.empty(s.B, C_, s.H, s.W, dtype=torch.float32, device=s.device)
    nv.check(nv.lib().tcs_s16_to_f32(s.ptr(), s.B, C_, s.H, s.W, s.G, int(group_offset), nv.ptr(out, "out"), nv.stream()), "tcs_s16_to_f32")
    return out


def _desc(pc: PackedConv, srcs: Sequence[S16], stride: int = 1) -> nv.ConvS16Desc:
    if pc.math != MATH_F16X3:
        raise ValueError("S16 convolutions need fp16-split packed weights (pack_conv(..., 'f16x3'))")
    if not 1 <= len(srcs) <= 4:
        raise ValueError("1..4 sources")
    d = nv.ConvS16Desc()
    B, H, W = srcs[0].B, srcs[0].H, srcs[0].W
    tot = 0
    for i, s in enumerate(srcs):
        if (s.B, s.H, s.W) != (B, H, W):
            raise ValueError(f"src{i} grid {(s.B, s.H, s.W)} does not match src0 {(B, H, W)}")
        d.src[i], d.src_ch[i], d.src_groups[i] = s.ptr(), s.C, s.G
        tot += s.C
    if tot != pc.cin:
        raise ValueError(f"sources carry {tot} channels, convolution expects {pc.cin}")
    d.n_src = len(srcs)
    d.weight, d.bias = nv.ptr(pc.weight), nv.ptr(pc.bias)
    d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.stride = B, H, W, pc.cin, pc.cout, pc.ksize, stride
    d.post_scale, d.weight_unscale = 1.0, pc.unscale
    return d


def _slice_ptr(t: Optional[torch.Tensor], ctot: int, c: int, grid, name: str):
    """Device pointer of an fp32 addend: a contiguous [B,c,H,W] tensor (ctot == 0), or a channel slice `base[:, c0:c0+c]` of a
    contiguous [B,ctot,H,W] tensor — element (b, ch) then lives at ptr + (b*ctot + ch)*H*W, which is what the kernels index."""
    if t is None:
        return None
    B, H, W = grid
    if not ctot:
        return nv.ptr(t, name)
    if t.dtype != torch.float32 or tuple(t.shape) != (B, c, H, W) or t.stride() != (ctot * H * W, H * W, W, 1):
        raise ValueError(f"{name}: expected a channel slice of a contiguous fp32 [B,{ctot},H,W] tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: tcs_mi355 kernels need a HIP device tensor")
    return t.data_ptr()


def _out_grid(s: S16, stride: int):
    return ((s.H - 1) // 2 + 1, (s.W - 1) // 2 + 1) if stride == 2 else (s.H, s.W)


def conv2d(pc: PackedConv, srcs: Sequence[S16], act: str = "none", addend: Optional[torch.Tensor] = None, post_scale: float = 1.0,
           out16: Optional[S16] = None, out16_group_offset: int = 0, out32: Optional[torch.Tensor] = None, out_coff: int = 0,
           stride: int = 1, want32: bool = False, tile_cfg: int = 0, addend16: Optional[S16] = None, addend_ctot: int = 0,
           out16b: Optional[S16] = None, out16_split: int = 0, taps: Optional["Taps"] = None, tap_weights: Optional["FragWeights"] = None):
    """act(conv(cat(srcs)) + bias + addend) * post_scale -> S16 (`out16`, allocated when neither output is given)
    and/or fp32 NCHW (`out32`, or allocated when want32).  Returns (out16, out32).  `addend_ctot` > Cout: `addend` is a
    channel slice of a [B, addend_ctot, Ho, Wo] tensor (pass the sliced view).  `out16b`: output channels from `out16_split`
    (a multiple of 32) on go to this second S16 tensor — two layers over the same input as one launch.  `taps` + `tap_weights`
    (`pack_taps`): the launch also leaves the tap partials of a following 3x3 convolution to 1-2 channels for the first `taps.ntile`
    32-channel tiles (tcs_conv_s16_desc.tap_*); `out16` is then optional."""
    d = _desc(pc, srcs, stride)
    Ho, Wo = _out_grid(srcs[0], stride)
    if out32 is None and want32:
        out32 = torch.empty(d.B, pc.cout, Ho, Wo, dtype=torch.float32, device=srcs[0].device)
    if out16 is None and out32 is None and taps is None:
        out16 = zeros(d.B, pc.cout, Ho, Wo, srcs[0].device)
    if out16 is not None and (out16.B, out16.H, out16.W) != (d.B, Ho, Wo):
        raise ValueError("conv2d: bad `out16` grid")
    if out32 is not None and (out32.shape[0] != d.B or tuple(out32.shape[2:]) != (Ho, Wo)):
        raise ValueError("conv2d: bad `out32` shape")
    if addend is not None and not addend_ctot and tuple(addend.shape) != (d.B, pc.cout, Ho, Wo):
        raise ValueError("conv2d: bad addend shape")
    d.epilogue, d.act, d.post_scale = EPI_LINEAR, ACT[act], float(post_scale)
    d.addend, d.addend_ctot = _slice_ptr(addend, addend_ctot, pc.cout, (d.B, Ho, Wo), "addend"), int(addend_ctot)
    if addend16 is not None:
        if (addend16.B, addend16.H, addend16.W) != (d.B, Ho, Wo):
            raise ValueError("conv2d: bad addend16 grid")
        d.addend16, d.addend16_groups = addend16.ptr(), addend16.G
    if out16 is not None:
        d.out16, d.out16_groups, d.out16_group_offset = out16.ptr(), out16.G, int(out16_group_offset)
    if out16b is not None:
        if (out16 is None and taps is None) or (out16b.B, out16b.H, out16b.W) != (d.B, Ho, Wo):
            raise ValueError("conv2d: `out16b` needs `out16` (or `taps`) and the same grid")
        d.out16b, d.out16b_groups, d.out16_split = out16b.ptr(), out16b.G, int(out16_split)
    if taps is not None:
        if tap_weights is None or tuple(taps.data.shape) != (d.B, taps.ntile, 9 * taps.nout, Ho, Wo):
            raise ValueError("conv2d: `taps` needs `tap_weights` and a [B, ntile, 9*nout, H, W] buffer")
        if tap_weights.data.numel() < nv.lib().tcs_tap_weights_floats(taps.nout, 32 * taps.ntile):
            raise ValueError("conv2d: `tap_weights` too small")
        d.tap_weights, d.tap_out, d.tap_nout, d.tap_tiles = nv.ptr(tap_weights.data, "tap_weights"), nv.ptr(taps.data, "taps"), taps.nout, taps.ntile
        d.tap_unscale = float(tap_weights.unscale)
    if out32 is not None:
        d.out32, d.out_ctot, d.out_coff = nv.ptr(out32, "out32"), int(out32.shape[1]), int(out_coff)
    d.tile_cfg = int(tile_cfg)
    _launch(d, "tcs_conv2d_s16")
    return out16, out32


def deconv_in_stats_ok(B: int, cout: int, H: int, W: int) -> bool:
    """Can the transposed convolution compute the InstanceNorm statistics of its own output (tcs_conv_s16_desc.in_stats) at this
    size ([H, W] = its INPUT grid)?"""
    return cout % 32 == 0 and nv.lib().tcs_deconv_in_stats_bytes(int(B), int(cout), int(H), int(W)) > 0


def deconv_in_stats_workspace(B: int, cout: int, H: int, W: int, device) -> torch.Tensor:
    """Zero-filled accumulators for ONE `deconv4x4s2(..., in_stats=)` launch: int64 [B, cout, stride >= 2], [..., 0:2] = fixed-point
    (sum x, sum x^2) of its output.  [H, W] is the transposed convolution's INPUT grid.  The launch ADDS: zero the tensor again before it is reused."""
    n = nv.lib().tcs_deconv_in_stats_bytes(int(B), int(cout), int(H), int(W))
    if n == 0:
        raise ValueError("deconv_in_stats_workspace: needs cout % 32 == 0 and an output grid of at most 2^20 pixels")
    return torch.zeros(int(B), int(cout), n // (8 * int(B) * int(cout)), dtype=torch.int64, device=device)


def deconv4x4s2(pc: PackedConv, srcs: Sequence[S16], out16: Optional[S16] = None, act: str = "none", tile_cfg: int = 0,
                in_stats: Optional[torch.Tensor] = None, eps: float = 1e-5) -> S16:
    """ConvTranspose2d(k=4, s=2, p=1, no bias) (+ activation): S16 [B,Cin,H,W] -> S16 [B,Cout,2H,2W].  With `in_stats`
    (deconv_in_stats_workspace: int64 [B, Cout, 2], ZERO before the launch) the launch also accumulates the sums InstanceNorm2d needs of its
    output there, for `instance_norm_apply`."""
    d = _desc(pc, srcs)
    cout = pc.cout // 4
    if out16 is None:
        out16 = zeros(d.B, cout, 2 * d.H, 2 * d.W, srcs[0].device)
    if (out16.B, out16.H, out16.W) != (d.B, 2 * d.H, 2 * d.W):
        raise ValueError("deconv4x4s2: bad `out16` grid")
    d.epilogue, d.act = EPI_DECONV2X, ACT[act]
    d.out16, d.out16_groups, d.out16_group_offset = out16.ptr(), out16.G, 0
    if in_stats is not None:
        need = nv.lib().tcs_deconv_in_stats_bytes(d.B, cout, d.H, d.W)
        if need == 0 or in_stats.dtype != torch.int64 or in_stats.numel() * 8 < need:
            raise ValueError("deconv4x4s2: `in_stats` must be an int64 tensor of [B, Cout, 2] (deconv_in_stats_workspace)")
        d.in_stats, d.in_eps = nv.ptr(in_stats, "in_stats", torch.int64), float(eps)
    d.tile_cfg = int(tile_cfg)
    _launch(d, "tcs_conv2d_s16[deconv2x]")
    return out16


def instance_norm_apply(x: S16, in_stats: torch.Tensor, act: str = "none", addend: Optional[S16] = None, out: Optional[S16] = None,
                        eps: float = 1e-5) -> S16:
    """act((x - mean) * rstd) + addend with the sums the producing `deconv4x4s2(..., in_stats=)` accumulated; `out` may be `x`."""
    out = zeros(x.B, x.C, x.H, x.W, x.device, x.G) if out is None else out
    if addend is not None and (addend.B, addend.H, addend.W, addend.G) != (x.B, x.H, x.W, x.G):
        raise ValueError("instance_norm_apply: bad addend")
    if (out.B, out.H, out.W, out.G) != (x.B, x.H, x.W, x.G):
        raise ValueError("instance_norm_apply: bad `out`")
    nv.check(nv.lib().tcs_instance_norm_apply_s16(x.ptr(), x.B, x.G, x.H, x.W, ACT[act], None if addend is None else addend.ptr(),
                                                  0 if addend is None else addend.G, out.ptr(), out.G,
                                                  nv.ptr(in_stats, "in_stats", torch.int64), x.C, float(eps), nv.stream()),
             "tcs_instance_norm_apply_s16")
    return out


def gru_gates(pc_zr: PackedConv, srcs: Sequence[S16], h: S16, cz=None, cr=None, z_out: Optional[torch.Tensor] = None,
              rh_out: Optional[S16] = None, tile_cfg: int = 0, addend_ctot: int = 0):
    """z = sigmoid(conv_zr[:hid] + cz) (fp32), rh = sigmoid(conv_zr[hid:] + cr) * h (S16)   (update.py:81-83, 30-33).
    `addend_ctot`: cz / cr are channel slices (views) of a [B, addend_ctot, H, W] tensor of partial sums."""
    d = _desc(pc_zr, srcs)
    hid = pc_zr.cout // 2
    if (h.B, h.H, h.W) != (d.B, d.H, d.W) or h.C != hid:
        raise ValueError("gru_gates: bad h")
    z_out = torch.empty(d.B, hid, d.H, d.W, dtype=torch.float32, device=h.device) if z_out is None else z_out
    rh_out = zeros(d.B, hid, d.H, d.W, h.device) if rh_out is None else rh_out
    d.epilogue = EPI_GRU_ZR
    d.addend = _slice_ptr(cz, addend_ctot, hid, (d.B, d.H, d.W), "cz")
    d.addend2 = _slice_ptr(cr, addend_ctot, hid, (d.B, d.H, d.W), "cr")
    d.addend_ctot = int(addend_ctot)
    d.h, d.h_groups = h.ptr(), h.G
    d.out32, d.out_ctot, d.out_coff = nv.ptr(z_out, "z"), hid, 0
    d.out16, d.out16_groups, d.out16_group_offset = rh_out.ptr(), rh_out.G, 0
    d.tile_cfg = int(tile_cfg)
    _launch(d, "tcs_conv2d_s16[gru_zr]")
    return z_out, rh_out


def gru_update(pc_q: PackedConv, srcs: Sequence[S16], h: S16, z: torch.Tensor, cq=None, keep_z: bool = False,
               out: Optional[S16] = None, out32: Optional[torch.Tensor] = None, tile_cfg: int = 0, addend_ctot: int = 0) -> S16:
    """q = tanh(conv_q + cq); h' = (1-z)h + zq (keep_z=False, update.py:85) or zh + (1-z)q (update.py:34,66).
    `out` may be `h` itself (in-place state update: each element is read and written by the same lane)."""
    d = _desc(pc_q, srcs)
    if (h.B, h.H, h.W) != (d.B, d.H, d.W) or h.C != pc_q.cout or tuple(z.shape) != (d.B, pc_q.cout, d.H, d.W):
        raise ValueError("gru_update: bad h/z")
    out = zeros(d.B, pc_q.cout, d.H, d.W, h.device) if out is None else out
    d.epilogue = EPI_GRU_Q
    d.addend, d.addend_ctot = _slice_ptr(cq, addend_ctot, pc_q.cout, (d.B, d.H, d.W), "cq"), int(addend_ctot)
    d.z, d.blend_keep_z = nv.ptr(z, "z"), int(keep_z)
    d.h, d.h_groups = h.ptr(), h.G
    d.out16, d.out16_groups, d.out16_group_offset = out.ptr(), out.G, 0
    if out32 is not None:
        d.out32, d.out_ctot, d.out_coff = nv.ptr(out32, "out32"), pc_q.cout, 0
    d.tile_cfg = int(tile_cfg)
    _launch(d, "tcs_conv2d_s16[gru_q]")
    return out


# ---------------------------------------------------------------------------------------------
# glue on S16 tensors (csrc/tcs_s16_ops.hip)
# ---------------------------------------------------------------------------------------------
def avgpool3s2(x: S16, out: Optional[S16] = None) -> S16:
    """avg_pool2d(3, stride 2, padding 1) (update.py:114-115)."""
    Ho, Wo = (x.H - 1) // 2 + 1, (x.W - 1) // 2 + 1
    out = zeros(x.B, x.C, Ho, Wo, x.device, x.G) if out is None else out
    if (out.B, out.H, out.W) != (x.B, Ho, Wo) or out.G < x.G:
        raise ValueError("avgpool3s2: bad `out`")
    nv.check(nv.lib().tcs_avgpool3s2_s16(x.ptr(), x.B, x.G, x.H, x.W, out.ptr(), out.G, nv.stream()), "tcs_avgpool3s2_s16")
    return out


def resize_bilinear(x: S16, Ho: int, Wo: int, out: Optional[S16] = None) -> S16:
    """F.interpolate(bilinear, align_corners=True) (update.py:122-124)."""
    out = zeros(x.B, x.C, Ho, Wo, x.device, x.G) if out is None else out
    if (out.B, out.H, out.W) != (x.B, Ho, Wo) or out.G < x.G:
        raise ValueError("resize_bilinear: bad `out`")
    nv.check(nv.lib().tcs_resize_bilinear_s16(x.ptr(), x.B, x.G, x.H, x.W, int(Ho), int(Wo), out.ptr(), out.G, nv.stream()),
             "tcs_resize_bilinear_s16")
    return out


_IN_WS: Dict[tuple, torch.Tensor] = {}


def instance_norm(x: S16, act: str = "none", addend: Optional[S16] = None, eps: float = 1e-5, out: Optional[S16] = None) -> S16:
    """act(InstanceNorm2d(x)) + addend on S16 tensors (affine=False, biased variance); `out` may be `x`."""
    out = zeros(x.B, x.C, x.H, x.W, x.device, x.G) if out is None else out
    if addend is not None and (addend.B, addend.H, addend.W, addend.G) != (x.B, x.H, x.W, x.G):
        raise ValueError("instance_norm: bad addend")
    if (out.B, out.H, out.W, out.G) != (x.B, x.H, x.W, x.G):
        raise ValueError("instance_norm: bad `out`")
    L = nv.lib()
    # one workspace per INPUT buffer (pool buffers live as long as the model): two launch sequences that run side by side (the extract
    # stage of the next frame beside the loop of the current one) never share the partial statistics
    key = (x.data.data_ptr(), x.B, x.G, x.H, x.W, str(x.device))
    ws = _IN_WS.get(key)
    if ws is None:
        ws = _IN_WS[key] = torch.empty(L.tcs_instance_norm_s16_workspace_bytes(x.B, x.G, x.H, x.W) // 4, dtype=torch.float32, device=x.device)
    nv.check(L.tcs_instance_norm_s16(x.ptr(), x.B, x.G, x.H, x.W, float(eps), ACT[act], None if addend is None else addend.ptr(),
                                     0 if addend is None else addend.G, out.ptr(), out.G, nv.ptr(ws), nv.stream()), "tcs_instance_norm_s16")
    return out


def propagate_disparity(grad: torch.Tensor, disp: torch.Tensor, out16: Optional[S16] = None, cand9: Optional[torch.Tensor] = None):
    """DispRefine.propagate_disparity (update.py:259-289): -> (S16 with the 27 stem channels, fp32 [B,9,H,W] candidates)."""
    B, _, H, W = (int(v) for v in disp.shape)
    out16 = zeros(B, 27, H, W, disp.device) if out16 is None else out16
    cand9 = torch.empty(B, 9, H, W, dtype=torch.float32, device=disp.device) if cand9 is None else cand9
    nv.check(nv.lib().tcs_propagate_disparity_s16(nv.ptr(grad, "grad"), nv.ptr(disp, "disp"), B, H, W, nv.ptr(cand9), out16.ptr(), out16.G,
                                                  nv.stream()), "tcs_propagate_disparity_s16")
    return out16, cand9


# ---------------------------------------------------------------------------------------------
# tap partials: a 3x3 convolution to 1-2 channels folded into its producer (csrc/tcs_stencil.hip)
# ---------------------------------------------------------------------------------------------
@dataclass
class Taps:
    """What a `conv2d(..., taps=)` launch leaves for the 3x3 convolution that follows it: fp32 [B, ntile, 9*nout, H, W] partial sums per
    32-channel tile of its output, + that convolution's bias.  Consumers: `taps_sum`, `flow_taps_step_grads`, `taps_propagate`."""
    data: torch.Tensor
    ntile: int
    nout: int
    bias: Optional[torch.Tensor] = None


def pack_taps(weight: torch.Tensor) -> "FragWeights":
    """[nout, C, 3, 3] weights of the folded convolution -> fp16-split MFMA fragments in the producer's accumulator order
    (tcs_pack_tap_weights); `.unscale` undoes the power-of-two pre-scale."""
    nout, C_ = int(weight.shape[0]), int(weight.shape[1])
    if tuple(weight.shape[2:]) != (3, 3) or nout not in (1, 2):
        raise ValueError("pack_taps: a 3x3 convolution with 1 or 2 output channels")
    w = weight.detach().float().contiguous()
    import math
    wmax = float(w.abs().max())
    s_log2 = 0 if wmax == 0.0 else max(-40, min(40, int(12 - math.floor(math.log2(wmax)))))
    out = torch.empty(nv.lib().tcs_tap_weights_floats(nout, C_), dtype=torch.float32, device=w.device)
    nv.check(nv.lib().tcs_pack_tap_weights(nv.ptr(w, "weight"), nout, C_, s_log2, nv.ptr(out), nv.stream()), "tcs_pack_tap_weights")
    return FragWeights(out, torch.zeros(0, device=w.device), 2.0 ** (-s_log2))


def taps_sum(t: Taps, addend: Optional[torch.Tensor] = None, scale: float = 1.0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """(addend + bias + folded convolution) * scale -> fp32 [B, nout, H, W]."""
    B, _, _, H, W = (int(v) for v in t.data.shape)
    out = torch.empty(B, t.nout, H, W, dtype=torch.float32, device=t.data.device) if out is None else out
    nv.check(nv.lib().tcs_taps_sum(nv.ptr(t.data, "taps"), t.ntile, t.nout, nv.ptr(t.bias), nv.ptr(addend, "addend"), float(scale), B, H, W,
                                   nv.ptr(out, "out"), nv.stream()), "tcs_taps_sum")
    return out


def flow_taps_step_grads(coords1: torch.Tensor, t: Taps, scale: float = 1.0, want_delta: bool = False):
    """`ops.flow_step_grads` with delta = FlowHead's output taken from tap partials -> (disp_q, scale * gradient, candidates[, delta])."""
    B, _, H, W = (int(v) for v in coords1.shape)
    if t.nout != 1 or tuple(t.data.shape[3:]) != (H, W):
        raise ValueError("flow_taps_step_grads: single-output taps on the coords grid")
    dq = torch.empty(B, 1, H, W, dtype=torch.float32, device=coords1.device)
    g = torch.empty(B, 2, H, W, dtype=torch.float32, device=coords1.device)
    c = torch.empty(B, 32, H, W, dtype=torch.float32, device=coords1.device)
    dl = torch.empty_like(dq) if want_delta else None
    nv.check(nv.lib().tcs_flow_taps_step_grads(nv.ptr(coords1, "coords1"), nv.ptr(t.data, "taps"), t.ntile, nv.ptr(t.bias), B, H, W, float(scale),
                                               nv.ptr(dq), nv.ptr(g), nv.ptr(c), nv.ptr(dl), nv.stream()), "tcs_flow_taps_step_grads")
    return (dq, g, c, dl) if want_delta else (dq, g, c)


def taps_propagate(t: Taps, g5: torch.Tensor, post_scale: float, disp: torch.Tensor, out16: Optional[S16] = None,
                   cand9: Optional[torch.Tensor] = None, grad: Optional[torch.Tensor] = None):
    """`propagate_disparity` with the gradient (g5 + bias + folded residual convolution) * post_scale (update.py:213) taken from tap
    partials -> (S16 stem input, fp32 candidates, fp32 gradient [B,2,H,W])."""
    B, _, H, W = (int(v) for v in disp.shape)
    if t.nout != 2 or tuple(t.data.shape[3:]) != (H, W) or tuple(g5.shape) != (B, 2, H, W):
        raise ValueError("taps_propagate: two-output taps and a [B,2,H,W] gradient on the disparity grid")
    out16 = zeros(B, 27, H, W, disp.device) if out16 is None else out16
    cand9 = torch.empty(B, 9, H, W, dtype=torch.float32, device=disp.device) if cand9 is None else cand9
    grad = torch.empty(B, 2, H, W, dtype=torch.float32, device=disp.device) if grad is None else grad
    nv.check(nv.lib().tcs_taps_propagate_s16(nv.ptr(t.data, "taps"), t.ntile, nv.ptr(t.bias), nv.ptr(g5, "g5"), float(post_scale), nv.ptr(disp, "disp"),
                                             B, H, W, nv.ptr(grad), nv.ptr(cand9), out16.ptr(), out16.G, nv.stream()), "tcs_taps_propagate_s16")
    return out16, cand9, grad


def set_channel(x: torch.Tensor, out: S16, channel: int) -> S16:
    """out[:, channel] = x ([B,1,H,W] fp32)."""
    B, _, H, W = (int(v) for v in x.shape)
    nv.check(nv.lib().tcs_s16_set_channel(nv.ptr(x, "x"), B, H, W, out.ptr(), out.G, int(channel), nv.stream()), "tcs_s16_set_channel")
    return out


def softmax_blend(logits9, cand9, disp_q, coords1, flow_x, flow_x_s16: Optional[S16] = None, flow_x_channel: int = 0, refined=None,
                  delta=None):
    """DispRefine's 9-way softmax blend (update.py:298-300) + the coordinate bookkeeping of tc_stereo.py:198-202; the next
    iteration's flow input also goes into channel `flow_x_channel` of `flow_x_s16` (the motion features, update.py:126)."""
    B, _, H, W = (int(v) for v in logits9.shape)
    refined = torch.empty(B, 1, H, W, dtype=torch.float32, device=logits9.device) if refined is None else refined
    delta = torch.empty_like(refined) if delta is None else delta
    nv.check(nv.lib().tcs_softmax_blend_s16(nv.ptr(logits9, "logits"), nv.ptr(cand9, "cand"), int(cand9.shape[1]), nv.ptr(disp_q), B, H, W,
                                            nv.ptr(refined), nv.ptr(delta), nv.ptr(coords1), nv.ptr(flow_x),
                                            None if flow_x_s16 is None else flow_x_s16.ptr(), 0 if flow_x_s16 is None else flow_x_s16.G,
                                            int(flow_x_channel), nv.stream()), "tcs_softmax_blend_s16")
    return refined, delta


def conv1x1_blend(pc: PackedConv, srcs: Sequence[S16], cand9, disp_q, coords1, flow_x, flow_x_s16: Optional[S16] = None,
                  flow_x_channel: int = 0, refined=None, delta=None, tile_cfg: int = 0, warm_pyramid=None, warm_radius: int = 4):
    """w_head's last 1x1 convolution (-> 9 logits) with DispRefine's softmax blend as its epilogue (update.py:297-300 +
    tc_stereo.py:198-202): one launch instead of conv + `softmax_blend`, same arithmetic and outputs -> (refined, delta)."""
    if pc.cout != 9 or pc.ksize != 1:
        raise ValueError("conv1x1_blend: a 1x1 convolution with 9 outputs")
    d = _desc(pc, srcs)
    refined = torch.empty(d.B, 1, d.H, d.W, dtype=torch.float32, device=srcs[0].device) if refined is None else refined
    delta = torch.empty_like(refined) if delta is None else delta
    d.epilogue, d.act, d.post_scale = EPI_BLEND9, ACT["none"], 1.0
    d.blend_cand, d.blend_cand_ctot, d.blend_disp = nv.ptr(cand9, "cand"), int(cand9.shape[1]), nv.ptr(disp_q, "disp")
    d.blend_refined, d.blend_delta = nv.ptr(refined, "refined"), nv.ptr(delta, "delta")
    d.blend_coords1, d.blend_flow_x = nv.ptr(coords1, "coords1"), nv.ptr(flow_x, "flow_x")
    if flow_x_s16 is not None:
        d.blend_flow16, d.blend_flow16_groups, d.blend_flow16_channel = flow_x_s16.ptr(), flow_x_s16.G, int(flow_x_channel)
    if warm_pyramid is not None:         # ops.CorrPyramid of this frame: the epilogue touches the rows the next lookup will read
        if (warm_pyramid.B, warm_pyramid.H, warm_pyramid.W) != (d.B, d.H, d.W):
            raise ValueError("conv1x1_blend: `warm_pyramid` has another grid")
        for i in range(4):
            d.blend_warm_pyr[i] = nv.ptr(warm_pyramid.levels[i], "pyr")
        d.blend_warm_radius = int(warm_radius)
    d.tile_cfg = int(tile_cfg)
    _launch(d, "tcs_conv2d_s16[blend9]")
    return refined, delta


# ---------------------------------------------------------------------------------------------
# fused pixelwise chains
# ---------------------------------------------------------------------------------------------
@dataclass
class FragWeights:
    """A 1x1 weight matrix as MFMA A fragments (tcs_pack_weight_frags)."""
    data: torch.Tensor
    bias: torch.Tensor
    unscale: float


def pack_frags(weight: torch.Tensor, bias: Optional[torch.Tensor], natural_channels: int) -> FragWeights:
    cout, cin = int(weight.shape[0]), int(weight.shape[1])
    w = weight.detach().float().reshape(cout, cin).contiguous()
    wmax = float(w.abs().max())
    import math
    s_log2 = 0 if wmax == 0.0 else max(-40, min(40, int(12 - math.floor(math.log2(wmax)))))
    L = nv.lib()
    buf = torch.empty(L.tcs_weight_frags_bytes(cout, cin) // 4, dtype=torch.float32, device=w.device)
    nv.check(L.tcs_pack_weight_frags(nv.ptr(w, "weight"), cout, cin, int(natural_channels), s_log2, nv.ptr(buf), nv.stream()),
             "tcs_pack_weight_frags")
    b = torch.zeros(cout, dtype=torch.float32, device=w.device) if bias is None else bias.detach().float().contiguous()
    return FragWeights(buf, b, 2.0 ** (-s_log2))


def hidden_update(h: S16, delta: torch.Tensor, w1: torch.Tensor, b1: torch.Tensor, W2: FragWeights, Wzr: FragWeights, Wq: FragWeights) -> S16:
    """HiddenstateUpdater (update.py:57-68) in one launch, `h` (128 channels) updated in place."""
    if h.C != 128 or tuple(delta.shape) != (h.B, 1, h.H, h.W):
        raise ValueError("hidden_update: h must carry 128 channels and delta must be [B,1,H,W]")
    nv.check(nv.lib().tcs_hidden_update_s16(h.ptr(), h.G, nv.ptr(delta, "delta"), nv.ptr(w1), nv.ptr(b1), nv.ptr(W2.data), nv.ptr(W2.bias),
                                            W2.unscale, nv.ptr(Wzr.data), nv.ptr(Wzr.bias), Wzr.unscale, nv.ptr(Wq.data), nv.ptr(Wq.bias),
                                            Wq.unscale, h.B, h.H, h.W, nv.stream()), "tcs_hidden_update_s16")
    return h


def take_flags() -> int:
    """Read-and-clear the device-side domain flags of the S16 producers (synchronises): bit 0 = a finite activation beyond the
    fp16 range was clamped to +-65504, bit 1 = a NaN / Inf was met.  The harness calls it once per sequence."""
    v = C.c_uint(0)
    nv.check(nv.lib().tcs_s16_flags(C.byref(v)), "tcs_s16_flags")
    return int(v.value)
