"""Drop-in for the reference's core/update.py: the refinement-loop blocks on the HIP library.

Module and parameter names equal the reference's, so `load_state_dict(strict=True)` accepts its
checkpoints; `nn.Conv2d` objects are kept as parameter containers only.  Their forward passes run
through `tcs_conv2d` (implicit GEMM on the matrix cores with virtual concat and fused epilogues), so
a ConvGRU is two launches instead of ~15 ATen ops, and the 24 host-synchronising NaN asserts per
iteration of the reference (update.py:27-35,58-67,78-86,155-158) are gone.  Every layer — including
the stride-2 convs, the transposed-conv + InstanceNorm up-blocks of both U-Nets and the
DisparityCompletor — runs on the HIP library; nothing here calls MIOpen.

Two tensor formats.  The reference-facing `forward` of every module takes and returns fp32 NCHW tensors.  Inside
the refinement loop the activations stay in the convolutions' operand form ("S16", tcs_mi355/s16.py): the `run`
methods take and return S16 tensors out of a per-model buffer pool, and `forward` is a thin conversion wrapper
around `run`, so module-level parity tests exercise exactly the kernels the frame uses.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from core.utils.basic_layers import Conv2x_IN
from tcs_mi355 import ops, s16
from tcs_mi355.streams import fork_join


# ---------------------------------------------------------------------------------------------
# packed-weight cache: one device-side repack per conv, redone only if the parameter changes
# ---------------------------------------------------------------------------------------------
# Contraction arithmetic of tcs_conv2d: "f16x3" = fp16 hi/lo split on the half-precision matrix pipe with fp32
# accumulation (fp32-equivalent accuracy, ~5x the fp32 MFMA rate; the loop's S16 kernels exist in this form only),
# "f32" = fp32-in/fp32-out MFMA.  A layer pins "f32" with `conv._tcs_math` (the gradient-candidate stem does: its inputs are
# unbounded).  There is no global switch: a frame whose loop is fp16-split while its head is not would be neither.
CONV_MATH = "f16x3"


def packed(conv: nn.Conv2d) -> ops.PackedConv:
    w, b = conv.weight, conv.bias
    math = getattr(conv, "_tcs_math", None) or CONV_MATH
    if conv.stride == (2, 2):
        math = "f16x3"          # stride-2 convolutions exist on the fp16-split kernel only
    key = (w.data_ptr(), w._version, w.device, None if b is None else (b.data_ptr(), b._version), math)
    hit = getattr(conv, "_tcs_packed", None)
    if hit is None or hit[0] != key:
        if conv.stride not in ((1, 1), (2, 2)) or conv.dilation != (1, 1) or conv.groups != 1 or \
                conv.padding != (conv.kernel_size[0] // 2,) * 2:
            raise NotImplementedError(f"tcs_conv2d covers stride-1/2 'same'-padded convolutions, got {conv}")
        hit = (key, ops.pack_conv(w, b, math))
        conv._tcs_packed = hit
    return hit[1]


def packed16(conv: nn.Conv2d) -> ops.PackedConv:
    """fp16-split packing also for a layer pinned to fp32 MFMA on its fp32-tensor path: the S16 kernels contract fp16 halves by construction."""
    if getattr(conv, "_tcs_math", None) == "f32":
        w, b = conv.weight, conv.bias
        key = (w.data_ptr(), w._version, w.device, None if b is None else (b.data_ptr(), b._version))
        hit = getattr(conv, "_tcs_packed16", None)
        if hit is None or hit[0] != key:
            hit = (key, ops.pack_conv(w, b, "f16x3"))
            conv._tcs_packed16 = hit
        return hit[1]
    return packed(conv)


def packed16_part(conv: nn.Conv2d, cin_slices, with_bias: bool = True) -> ops.PackedConv:
    """fp16-split packing of `conv.weight[:, cat(cin_slices)]` (with or without the bias): one K-slice of a layer.  A
    convolution is linear in its input channels, so a layer over cat(a, b) whose inputs are ready at different times runs as
    conv_a(a) + bias -> fp32 partial sum, then conv_b(b) with that sum as the epilogue's addend (tcs_mi355.h, addend_ctot)."""
    w, b = conv.weight, conv.bias
    key = (w.data_ptr(), w._version, w.device, None if b is None else (b.data_ptr(), b._version))
    cache = conv.__dict__.setdefault("_tcs_parts", {})
    spec = (tuple(tuple(int(v) for v in sl) for sl in cin_slices), bool(with_bias))
    hit = cache.get(spec)
    if hit is None or hit[0] != key:
        ws = torch.cat([w.detach()[:, lo:hi] for lo, hi in spec[0]], 1).contiguous()
        hit = (key, ops.pack_conv(ws, b if with_bias else None, "f16x3"))
        cache[spec] = hit
    return hit[1]


def packed16_cat(convs) -> ops.PackedConv:
    """fp16-split packing of several layers that read the same input, concatenated along Cout (weights and biases): one
    launch, the outputs split again by channel range in the epilogue (tcs_conv_s16_desc.out16b)."""
    key = tuple((c.weight.data_ptr(), c.weight._version, None if c.bias is None else (c.bias.data_ptr(), c.bias._version)) for c in convs)
    host = convs[0]
    cache = host.__dict__.setdefault("_tcs_cat", {})
    ids = tuple(id(c) for c in convs)
    hit = cache.get(ids)
    if hit is None or hit[0] != key:
        w = torch.cat([c.weight.detach() for c in convs], 0).contiguous()
        b = torch.cat([(c.bias.detach() if c.bias is not None else torch.zeros(c.out_channels, device=c.weight.device)) for c in convs], 0).contiguous()
        hit = (key, ops.pack_conv(w, b, "f16x3"))
        cache[ids] = hit
    return hit[1]


def tap_weights(conv2: nn.Conv2d) -> torch.Tensor:
    w = conv2.weight
    key = (w.data_ptr(), w._version, w.device)
    hit = getattr(conv2, "_tcs_taps", None)
    if hit is None or hit[0] != key:
        hit = (key, s16.pack_taps(w))
        conv2._tcs_taps = hit
    return hit[1]


def tap_partials(pool, conv1: nn.Conv2d, conv2: nn.Conv2d, srcs, pc=None, out16b=None) -> s16.Taps:
    """relu(conv1(srcs)) with conv2 (3x3 to 1-2 channels) folded into its epilogue as tap partials.  `pc` / `out16b`: conv1 is the
    leading channel range of a wider joint launch (packed16_cat) whose remaining channels go to `out16b`."""
    a = srcs[0]
    ntile = (conv1.out_channels + 31) // 32
    taps = s16.Taps(pool.get32((id(conv2), "taps"), (a.B, ntile, 9 * conv2.out_channels, a.H, a.W), a.device), ntile, conv2.out_channels,
                    None if conv2.bias is None else conv2.bias.detach())
    s16.conv2d(packed16(conv1) if pc is None else pc, srcs, act="relu", taps=taps, tap_weights=tap_weights(conv2), out16b=out16b,
               out16_split=conv1.out_channels if out16b is not None else 0)
    return taps


def pool_of(module) -> s16.S16Pool:
    """The S16 buffer pool of the model a module belongs to (TCStereo shares one; a stand-alone module gets its own)."""
    p = getattr(module, "_s16pool", None)
    if p is None:
        p = s16.S16Pool()
        for m in module.modules():
            m._s16pool = p
    return p


# A/B switches for benchmarking sessions (tools/, gpurun logs): comma-separated tokens in TCS_MI355_X.  Never set in production.
_X = set(t for t in os.environ.get("TCS_MI355_X", "").split(",") if t)
# Independent layer pairs as ONE grouped launch (tcs_conv2d_s16_group / tcs_conv2d_group) instead of two branches of the captured graph;
# "nogroup" (A/B): two launches in a row.
GROUP = "nogroup" not in _X


def conv16(pool, conv, srcs, act="none", addend=None, post_scale=1.0, out=None, want32=False, tag="o", addend16=None, pc=None, tile_cfg=0):
    """A Conv2d on S16 sources -> S16 (a pool buffer owned by this conv, or `out`), or fp32 NCHW when want32.
    `pc`: a K-slice of the layer's weights (packed16_part) when part of its input was already accumulated into `addend`."""
    a = srcs[0]
    pc = packed16(conv) if pc is None else pc
    stride = conv.stride[0]
    Ho, Wo = ((a.H - 1) // 2 + 1, (a.W - 1) // 2 + 1) if stride == 2 else (a.H, a.W)
    if want32:
        return s16.conv2d(pc, srcs, act=act, addend=addend, post_scale=post_scale, want32=True, stride=stride, addend16=addend16)[1]
    if out is None:
        out = pool.get((id(conv), tag), a.B, conv.out_channels, Ho, Wo, a.device)
    return s16.conv2d(pc, srcs, act=act, addend=addend, post_scale=post_scale, out16=out, stride=stride, addend16=addend16, tile_cfg=tile_cfg)[0]


def conv32to16(pool, conv, x, act="none", tag="o", image_pair=None, in_transform=0):
    """A Conv2d fed by a fp32 NCHW tensor (correlation features, disparity stencils, single-channel maps) writing S16.
    `image_pair` / `in_transform`: the 7x7 RGB stem reading raw left | right images (ops.conv2d)."""
    B, _, H, W = (int(v) for v in x.shape)
    B += 0 if image_pair is None else int(image_pair.shape[0])
    out = pool.get((id(conv), tag), B, conv.out_channels, H, W, x.device)
    return ops.conv2d(packed(conv), [x.float().contiguous()], act=act, out16=out,
                      image_pair=None if image_pair is None else image_pair.float().contiguous(), in_transform=in_transform)


def to16(pool, x, key):
    B, C_, H, W = (int(v) for v in x.shape)
    return s16.to_s16(x.float().contiguous(), out=pool.get(key, B, C_, H, W, x.device))


IN_SUM_SLOTS = 64           # sets of InstanceNorm accumulators per fused up-block and frame (one per call; DispGradPredictor.begin_frame clears them)


def up_block16(pool, block: Conv2x_IN, x: s16.S16, rem: s16.S16, slot=None) -> s16.S16:
    """Conv2x_IN(deconv=True, concat=False) on S16 tensors: transposed conv -> InstanceNorm -> LeakyReLU -> + rem ->
    3x3 conv [-> InstanceNorm] -> LeakyReLU (basic_layers.py:38-77).  `slot` (0 .. IN_SUM_SLOTS-1): the transposed convolution accumulates
    the InstanceNorm sums of its own output (tcs_conv_s16_desc.in_stats: fixed-point integer atomics) into set `slot` of the block's
    accumulators and the statistics launch goes away; the CALLER clears the sets before they are used again (`up_block_sums(...).zero_()`
    once per frame: the loop uses one set per iteration)."""
    dc = block.conv1.conv
    y = pool.get((id(dc), "o"), x.B, dc.out_channels, 2 * x.H, 2 * x.W, x.device)
    stats = None
    if slot is not None and "noinfuse" not in _X and s16.deconv_in_stats_ok(x.B, dc.out_channels, x.H, x.W):
        stats = up_block_sums(pool, block, x.B, x.device)[int(slot)]
    s16.deconv4x4s2(packed_deconv(dc), [x], out16=y, in_stats=stats)
    norm = (lambda **kw: s16.instance_norm_apply(y, stats, **kw)) if stats is not None else (lambda **kw: s16.instance_norm(y, **kw))
    if (y.H, y.W) != (rem.H, rem.W):         # odd-sized skip: nearest resize as the reference does (rare; through fp32)
        y32 = F.interpolate(norm(act="leaky", out=y).float(), size=(rem.H, rem.W), mode="nearest") + rem.float()
        y = to16(pool, y32, (id(dc), "resized"))
    else:
        y = norm(act="leaky", addend=rem, out=y)
    act2 = "leaky" if block.conv2.relu else "none"
    if block.conv2.use_in:
        z = conv16(pool, block.conv2.conv, [y])
        return s16.instance_norm(z, act=act2, out=z)
    return conv16(pool, block.conv2.conv, [y], act=act2)


def up_block_sums(pool, block: Conv2x_IN, B: int, device) -> torch.Tensor:
    """int64 [IN_SUM_SLOTS, B, C, stride]: the fixed-point InstanceNorm accumulators of `up_block16(..., slot=)` for this block."""
    dc = block.conv1.conv
    stride = s16.nv.lib().tcs_deconv_in_stats_bytes(1, dc.out_channels, 8, 8) // (8 * dc.out_channels)       # 64-bit words per channel
    return pool.get_i64((id(dc), "in_sums"), (IN_SUM_SLOTS, int(B), dc.out_channels, stride), device)


def hip_conv(conv, srcs, act="none", **kw):
    return ops.conv2d(packed(conv), [s.float().contiguous() for s in srcs], act=act, stride=conv.stride[0], **kw)


def packed_deconv(deconv: nn.ConvTranspose2d) -> ops.PackedConv:
    w = deconv.weight
    key = (w.data_ptr(), w._version, w.device)
    hit = getattr(deconv, "_tcs_packed", None)
    if hit is None or hit[0] != key:
        if deconv.kernel_size != (4, 4) or deconv.stride != (2, 2) or deconv.padding != (1, 1) or deconv.bias is not None:
            raise NotImplementedError(f"tcs deconv covers ConvTranspose2d(4, stride 2, pad 1, bias=False), got {deconv}")
        hit = (key, ops.pack_deconv4x4s2(w))
        deconv._tcs_packed = hit
    return hit[1]


def hip_up_block(block: Conv2x_IN, x, rem):
    """Conv2x_IN(deconv=True, concat=False) (basic_layers.py:38-77) on the HIP library:
    transposed conv -> InstanceNorm -> LeakyReLU -> (+ rem) -> 3x3 conv [-> InstanceNorm] -> LeakyReLU."""
    y = ops.deconv4x4s2(packed_deconv(block.conv1.conv), [x.float().contiguous()])
    if y.shape != rem.shape:
        y = F.interpolate(ops.instance_norm(y, act="leaky"), size=rem.shape[-2:], mode="nearest") + rem
    else:
        y = ops.instance_norm(y, act="leaky", addend=rem.float().contiguous())
    if block.conv2.use_in:
        z = hip_conv(block.conv2.conv, [y])
        return ops.instance_norm(z, act="leaky" if block.conv2.relu else "none")
    return hip_conv(block.conv2.conv, [y], act="leaky" if block.conv2.relu else "none")


def hip_seq(seq: nn.Sequential, srcs, last_act="none"):
    """conv -> ReLU -> conv chains declared as nn.Sequential(conv, ReLU, conv[, ReLU|Sigmoid])."""
    x = list(srcs)
    mods = list(seq)
    i = 0
    while i < len(mods):
        conv = mods[i]
        act = "none"
        if i + 1 < len(mods) and not isinstance(mods[i + 1], nn.Conv2d):
            nxt = mods[i + 1]
            act = {nn.ReLU: "relu", nn.LeakyReLU: "leaky", nn.Sigmoid: "sigmoid"}[type(nxt)]
            i += 1
        x = [hip_conv(conv, x, act=act)]
        i += 1
    return x[0]


def _conv(cin, cout, k, stride=1):
    return nn.Conv2d(cin, cout, k, stride, k // 2)


def _two(cin, mid, cout, k=3, tail=None):
    layers = [_conv(cin, mid, k), nn.ReLU(inplace=True), _conv(mid, cout, k)]
    return nn.Sequential(*layers, *([tail] if tail is not None else []))


# ---------------------------------------------------------------------------------------------
# recurrent cells
# ---------------------------------------------------------------------------------------------
class _GateCell(nn.Module):
    """z,r = sigma(conv_zr[h,x] (+cz,cr)); q = tanh(conv_q[r*h,x] (+cq)); blend (update.py:26-36,77-87)."""
    keep_z = False          # False: h' = (1-z)h + zq ; True: h' = zh + (1-z)q

    def __init__(self, hidden_dim, input_dim, kernel_size):
        super().__init__()
        self.convzr = _conv(hidden_dim + input_dim, hidden_dim * 2, kernel_size)
        self.convq = _conv(hidden_dim + input_dim, hidden_dim, kernel_size)

    def _step(self, h, xs, cz=None, cr=None, cq=None):
        h = h.float().contiguous()
        xs = [x.float().contiguous() for x in xs]
        z, rh = ops.gru_gates(packed(self.convzr), [h, *xs], h, cz, cr)
        return ops.gru_update(packed(self.convq), [rh, *xs], h, z, cq, keep_z=self.keep_z)

    def step16(self, pool, h: s16.S16, xs, cz=None, cr=None, cq=None) -> s16.S16:
        """The same cell on S16 tensors, updating `h` IN PLACE (two launches; z stays fp32, r*h is an S16 temporary).  cz / cr / cq may
        be channel slices (views) of one [B, 3*hidden, H, W] tensor — the context convolution's output, no chunk copies."""
        z = pool.get32((id(self), "z"), (h.B, h.C, h.H, h.W), h.device)
        rh = pool.get((id(self), "rh"), h.B, h.C, h.H, h.W, h.device)
        ctot = lambda t: 0 if (t is None or t.is_contiguous()) else int(t.stride(0)) // (h.H * h.W)
        if ctot(cz) != ctot(cr):
            cz, cr = cz.contiguous(), cr.contiguous()
        s16.gru_gates(packed16(self.convzr), [h, *xs], h, cz, cr, z_out=z, rh_out=rh, addend_ctot=ctot(cz))
        return s16.gru_update(packed16(self.convq), [rh, *xs], h, z, cq, keep_z=self.keep_z, out=h, addend_ctot=ctot(cq))


class ConvGRU(_GateCell):
    def __init__(self, hidden_dim, input_dim, kernel_size=3):
        super().__init__(hidden_dim, input_dim, kernel_size)

    def forward(self, h, cz, cr, cq, *x_list):
        return self._step(h, x_list, cz.contiguous(), cr.contiguous(), cq.contiguous())


class Lightfuse(_GateCell):
    keep_z = True

    def __init__(self, hidden_dim, input_dim):
        super().__init__(hidden_dim, input_dim, 1)

    def forward(self, h, x):
        return self._step(h, [x])


class Hardfuse(nn.Module):          # unused by the model (update.py:39-45); kept for API parity
    def forward(self, h, x, mask):
        return mask * h + (1 - mask) * x


class HiddenstateUpdater(_GateCell):
    keep_z = True

    def __init__(self, hidden_dim):
        super().__init__(hidden_dim, 64, 1)
        self.convs = nn.Sequential(_conv(1, 64, 1), nn.LeakyReLU(inplace=True), _conv(64, 64, 1))

    def _frags(self):
        """The cell's four weight matrices in the fused kernel's layouts, re-packed when a parameter changes."""
        ps = [self.convs[0].weight, self.convs[0].bias, self.convs[2].weight, self.convs[2].bias, self.convzr.weight, self.convzr.bias,
              self.convq.weight, self.convq.bias]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        hit = getattr(self, "_tcs_frags", None)
        if hit is None or hit[0] != key:
            w1 = self.convs[0].weight.detach().float().reshape(64).contiguous()
            b1 = self.convs[0].bias.detach().float().contiguous()
            hit = (key, (w1, b1, s16.pack_frags(self.convs[2].weight, self.convs[2].bias, 64),
                         s16.pack_frags(self.convzr.weight, self.convzr.bias, 0), s16.pack_frags(self.convq.weight, self.convq.bias, 0)))
            self._tcs_frags = hit
        return hit[1]

    def run(self, pool, h: s16.S16, delta: torch.Tensor) -> s16.S16:
        """One launch (tcs_hidden_update_s16): every layer is pixelwise, so a wave carries its 32 pixels through all of them."""
        if h.C == 128 and self.convs[0].out_channels == 64 and "nohu" not in _X:
            return s16.hidden_update(h, delta, *self._frags())
        x = conv32to16(pool, self.convs[0], delta, act="leaky")              # generic widths: layer by layer
        x = conv16(pool, self.convs[2], [x])
        return self.step16(pool, h, [x])

    def forward(self, h, x):
        pool = pool_of(self)
        return self.run(pool, to16(pool, h, (id(self), "h_in")), x.float().contiguous()).float()


# ---------------------------------------------------------------------------------------------
# disparity-space update block
# ---------------------------------------------------------------------------------------------
class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256, output_dim=2):
        super().__init__()
        self.conv1 = _conv(input_dim, hidden_dim, 3)
        self.conv2 = _conv(hidden_dim, output_dim, 3)
        self.relu = nn.ReLU(inplace=True)

    def run(self, pool, x: s16.S16, lazy: bool = False):
        """-> fp32 [B,out,H,W], or with `lazy` the tap partials of conv2 (s16.Taps) for a consumer that finishes the sum itself.
        conv2 (256 -> 1 or 2 channels, 3x3) never runs as a launch: conv1's epilogue leaves 9 partial sums per 32-channel tile and
        output (csrc/tcs_stencil.hip, "Tap partials"), and the 256-channel intermediate is never stored."""
        if self.conv2.out_channels <= 2 and "notaps" not in _X:
            taps = tap_partials(pool, self.conv1, self.conv2, [x])
            return taps if lazy else s16.taps_sum(taps)
        y = conv16(pool, self.conv1, [x], act="relu")
        return conv16(pool, self.conv2, [y], want32=True)

    def forward(self, x):
        pool = pool_of(self)
        return self.run(pool, to16(pool, x, (id(self), "x_in")))


class BasicMotionEncoder(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        planes = args.corr_levels * (2 * args.corr_radius + 1)
        self.convc1 = _conv(planes, 64, 1)
        self.convc2 = _conv(64, 64, 3)
        self.convf1 = _conv(1, 64, 7)
        self.convf2 = _conv(64, 64, 3)
        self.conv = _conv(128, 127, 3)

    def forward(self, flow, corr, out=None):
        """cat(features(126 + 1), flow) [N,128,H,W] fp32 (update.py:103-111); `out`, when given, receives it."""
        flow = flow.float().contiguous()
        pool = pool_of(self)
        n, _, h, w = flow.shape
        motion = pool.get((id(self), "motion_api"), n, 128, h, w, flow.device)
        s16.set_channel(flow, motion, 127)
        res = self.run(pool, flow, corr.float().contiguous(), motion).float()
        if out is not None:
            out.copy_(res)
            return out
        return res

    def run(self, pool, flow: torch.Tensor, corr: torch.Tensor, motion: s16.S16) -> s16.S16:
        """`motion`: the S16 [N,128,H,W] motion feature buffer whose channel 127 ALREADY holds `flow`; channels 0..126 are
        written here (the 127-channel epilogue never touches channel 127: masked partial store)."""
        # convc2 | convf2 (two independent 3x3 64 -> 64 layers) as ONE grouped launch (tcs_conv2d_s16_group) instead of two graph branches
        c1 = conv32to16(pool, self.convc1, corr, act="relu")
        f1 = conv32to16(pool, self.convf1, flow, act="relu")
        with s16.grouped(enabled=GROUP):
            c = conv16(pool, self.convc2, [c1], act="relu")
            f = conv16(pool, self.convf2, [f1], act="relu")
        return conv16(pool, self.conv, [c, f], act="relu", out=motion)


def pool2x(x):
    return ops.avgpool3s2(x.float().contiguous())


def pool4x(x):
    return F.avg_pool2d(x, 5, stride=4, padding=1)     # unused by the model (update.py:118-119)


def interp(x, dest):
    return ops.resize_bilinear(x.float().contiguous(), int(dest.shape[2]), int(dest.shape[3]))


class BasicMultiUpdateBlock(nn.Module):
    def __init__(self, args, hidden_dims=[]):
        super().__init__()
        self.args = args
        self.encoder = BasicMotionEncoder(args)
        n = args.n_gru_layers
        self.gru08 = ConvGRU(hidden_dims[2], 128 + hidden_dims[1] * (n > 1))
        self.gru16 = ConvGRU(hidden_dims[1], hidden_dims[0] * (n == 3) + hidden_dims[2])
        self.gru32 = ConvGRU(hidden_dims[0], hidden_dims[1])
        self.flow_head = FlowHead(hidden_dims[2], hidden_dim=256, output_dim=1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_in", nonlinearity="relu")
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, net, inp, corr=None, flow=None, iter08=True, iter16=True, iter32=True, update=True, motion_out=None):
        """Coarse-to-fine GRU sweep (update.py:145-168) on fp32 tensors; returns the new `net` list (+ delta_flow)."""
        pool = pool_of(self)
        net16 = [to16(pool, t, (id(self), "net_in", i)) for i, t in enumerate(net)]
        motion = None
        if iter08:
            flow = flow.float().contiguous()
            motion = pool.get((id(self), "motion_api"), flow.shape[0], 128, flow.shape[2], flow.shape[3], flow.device)
            s16.set_channel(flow, motion, 127)
        inp = [[t.float().contiguous() for t in trio] for trio in inp]
        res = self.run(pool, net16, inp, None if corr is None else corr.float().contiguous(), flow, motion, iter08, iter16, iter32, update)
        out = [n.float() for n in net16]
        if motion_out is not None and motion is not None:
            motion_out.copy_(motion.float())
        return (out, res) if update else out

    def begin_frame(self):
        """Called once per frame before the loop: drops what `gru16_early` keeps across the iterations of ONE frame (cat(cz, cr)).
        (The cache used to be keyed on the identity of the context tensors — which is the same object frame after frame once the
        extract stage's outputs are static graph tensors.)"""
        self._czr16_src = None

    def run_gru32(self, pool, net, inp):
        """gru32 on pool2x(net16) (update.py:147-148), in place, and interp(net32 -> 1/8 grid) for gru16.  Needs only net16 /
        net32, so the frame loop launches it for iteration i+1 as soon as gru16 of iteration i is done (tc_stereo.py)."""
        p = s16.avgpool3s2(net[1], out=pool.get((id(self), "pool16"), net[2].B, net[1].C, net[2].H, net[2].W, net[2].device))
        self.gru32.step16(pool, net[2], [p], *inp[2])
        return s16.resize_bilinear(net[2], net[1].H, net[1].W, out=pool.get((id(self), "up32"), net[1].B, net[2].C, net[1].H, net[1].W, net[1].device))

    # ---- gru16 split by input availability (update.py:149-153 is linear in its input channels before the gate non-linearity) ----
    # cat(net16, pool2x(net08), interp(net32)): net16 and interp(net32) are final as soon as gru32 has run, i.e. during the previous
    # iteration's refinement, when the GPU has idle capacity; pool2x(net08) arrives only after the hidden-state update on the critical
    # chain.  The early share of conv_zr / conv_q goes into fp32 partial sums (tcs_conv_s16_desc.addend_ctot), the late share finishes.
    def gru16_early(self, pool, net, inp, up32):
        g, h = self.gru16, net[1]
        hid, c0 = h.C, h.C + net[0].C                  # c0: first input channel of interp(net32)
        cz, cr, cq = inp[1]
        p_zr = pool.get32((id(self), "p16zr"), (h.B, 2 * hid, h.H, h.W), h.device)
        p_q = pool.get32((id(self), "p16q"), (h.B, hid, h.H, h.W), h.device)
        sliced = cr.data_ptr() == cz.data_ptr() + hid * h.H * h.W * 4 and cq.data_ptr() == cr.data_ptr() + hid * h.H * h.W * 4 \
            and cz.stride(0) == cr.stride(0) == cq.stride(0) == 3 * hid * h.H * h.W
        if sliced:                               # cz | cr | cq are the three thirds of the context convolution's output: cat(cz, cr) is a view
            ct = int(cz.stride(0)) // (h.H * h.W)
            czr = torch.as_strided(cz, (h.B, 2 * hid, h.H, h.W), cz.stride())
            s16.conv2d(packed16_part(g.convzr, ((0, hid), (c0, c0 + up32.C))), [h, up32], addend=czr, addend_ctot=ct, out32=p_zr)
            s16.conv2d(packed16_part(g.convq, ((c0, c0 + up32.C),)), [up32], addend=cq, addend_ctot=ct, out32=p_q)
            return p_zr, p_q
        czr = pool.get32((id(self), "czr16"), (h.B, 2 * hid, h.H, h.W), h.device)
        if getattr(self, "_czr16_src", None) is not cz:                  # cat(cz, cr): once per frame (inp is per frame)
            torch.cat([cz, cr], 1, out=czr)
            self._czr16_src = cz
        s16.conv2d(packed16_part(g.convzr, ((0, hid), (c0, c0 + up32.C))), [h, up32], addend=czr, out32=p_zr)
        s16.conv2d(packed16_part(g.convq, ((c0, c0 + up32.C),)), [up32], addend=cq.contiguous(), out32=p_q)
        return p_zr, p_q

    def gru16_late(self, pool, net, partial):
        """pool2x(net08) -> the rest of gru16 -> interp(net16) for gru08."""
        p_zr, p_q = partial
        g, h = self.gru16, net[1]
        hid = h.C
        p = s16.avgpool3s2(net[0], out=pool.get((id(self), "pool08"), h.B, net[0].C, h.H, h.W, h.device))
        z = pool.get32((id(g), "z"), (h.B, hid, h.H, h.W), h.device)
        rh = pool.get((id(g), "rh"), h.B, hid, h.H, h.W, h.device)
        s16.gru_gates(packed16_part(g.convzr, ((hid, hid + p.C),), with_bias=False), [p], h, p_zr[:, :hid], p_zr[:, hid:], z_out=z,
                      rh_out=rh, addend_ctot=2 * hid)
        s16.gru_update(packed16_part(g.convq, ((0, hid + p.C),), with_bias=False), [rh, p], h, z, p_q, keep_z=g.keep_z, out=h)
        return s16.resize_bilinear(h, net[0].H, net[0].W, out=pool.get((id(self), "up16"), h.B, h.C, net[0].H, net[0].W, h.device))

    def run_coarse(self, pool, net, inp, iter16=True, iter32=True, want_up16=True, up32=None):
        """gru32 -> gru16 (update.py:147-153) on S16 states, in place; returns interp(net16 -> 1/4 grid) for gru08 (or None).
        Independent of the motion encoder, which only feeds gru08.  `up32`: gru32 of this iteration already ran (run_gru32)."""
        n = self.args.n_gru_layers
        if iter32 and up32 is None:
            up32 = self.run_gru32(pool, net, inp)
        if iter16:
            xs = [s16.avgpool3s2(net[0], out=pool.get((id(self), "pool08"), net[1].B, net[0].C, net[1].H, net[1].W, net[1].device))]
            if n > 2:
                if up32 is None:
                    up32 = s16.resize_bilinear(net[2], net[1].H, net[1].W,
                                               out=pool.get((id(self), "up32"), net[1].B, net[2].C, net[1].H, net[1].W, net[1].device))
                xs.append(up32)
            self.gru16.step16(pool, net[1], xs, *inp[1])
        if want_up16 and n > 1:
            return s16.resize_bilinear(net[1], net[0].H, net[0].W,
                                       out=pool.get((id(self), "up16"), net[0].B, net[1].C, net[0].H, net[0].W, net[0].device))
        return None

    def run_fine(self, pool, net, inp, motion_features, up16, update=True, lazy=False):
        """gru08 on (motion features, upsampled net16) and the flow head (update.py:154-168); returns delta_flow (fp32), or with
        `lazy` the flow head's tap partials (FlowHead.run)."""
        xs = [motion_features] + ([up16] if up16 is not None else [])
        self.gru08.step16(pool, net[0], xs, *inp[0])
        return self.flow_head.run(pool, net[0], lazy=lazy) if update else None

    def run(self, pool, net, inp, corr, flow, motion, iter08=True, iter16=True, iter32=True, update=True):
        """`net`: list of S16 hidden states, updated IN PLACE; `inp`: per scale (cz, cr, cq) fp32; returns delta_flow (fp32)
        when `update`."""
        if not iter08:
            self.run_coarse(pool, net, inp, iter16, iter32, want_up16=False)
            return None
        # the coarse GRUs first (the longer chain stays in the launch list, tcs_mi355/streams.py), the motion encoder beside them
        up16, m = fork_join([lambda: self.run_coarse(pool, net, inp, iter16, iter32),
                             lambda: self.encoder.run(pool, flow, corr, motion)], site="coarse")
        return self.run_fine(pool, net, inp, m, up16, update)


# ---------------------------------------------------------------------------------------------
# gradient-space refinement
# ---------------------------------------------------------------------------------------------
class DispGradPredictor(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.conv_grad_stem = _two(2, 32, 32)
        self.conv_grad_candidate_stem = _two(32, 64, 64)
        # its input (-n_x/n_z of neighbour cross products, geo_utils.py:99-100) is unbounded: keep fp32 operands
        self.conv_grad_candidate_stem[0]._tcs_math = "f32"
        # (2 -> 32: K = 18.  On the fp32-MFMA kernel too, so that it can share a grouped launch with the candidate stem's first layer)
        self.conv_grad_stem[0]._tcs_math = "f32"
        relu = lambda: nn.ReLU(inplace=True)
        self.conv_4_4 = nn.Sequential(_conv(160, 64, 3), relu())
        self.conv_4_8 = nn.Sequential(_conv(64, 96, 3, 2), relu())
        self.conv_8_8 = nn.Sequential(_conv(160, 96, 3), relu())
        self.conv_8_16 = nn.Sequential(_conv(96, 128, 3, 2), relu())
        self.conv_16_16 = nn.Sequential(_conv(192, 128, 3), relu())
        self.conv_16_8 = Conv2x_IN(128, 96, deconv=True, concat=False, keep_concat=False, IN=False)
        self.conv_8_4 = Conv2x_IN(96, 64, deconv=True, concat=False, keep_concat=False, IN=False)
        self.residual_head = _two(64, 128, 2)
        self.conv_out = nn.Sequential(_conv(64, 64, 3), relu())

    def forward(self, disp_grad, disp, clist, g5=None, cands=None):
        """fp32 API (update.py:198-214): -> (refined gradient [N,2,H,W], context [N,64,H,W]).  `g5`, when given, is
        5*disp_grad already produced by the gradient kernel; `cands` the gradient candidates of `disp`."""
        disp = disp.float().contiguous()
        if g5 is None:
            g5 = (5 * disp_grad).float().contiguous()            # update.py:199
        if cands is None:
            cands = ops.grad_candidates(disp)                    # [N,32,H,W] (update.py:202-204)
        pool = pool_of(self)
        c16 = [to16(pool, c, (id(self), "clist_in", i)) for i, c in enumerate(clist)]
        grad, ctx = self.run(pool, g5, cands, self.prepare(pool, c16))
        return grad, ctx.float()

    def prepare(self, pool, clist):
        """Once per frame.  conv_4_4 / conv_8_8 / conv_16_16 read cat(features, clist[i]) (update.py:205-209) and `clist` does
        not change over the iterations, so its share of each convolution (40 % / 40 % / 33 % of the input channels) is
        computed here — bias included — and enters the per-iteration launch as the epilogue's fp32 addend."""
        pre = []
        for i, conv in enumerate((self.conv_4_4[0], self.conv_8_8[0], self.conv_16_16[0])):
            c = clist[i]
            lo = conv.in_channels - c.C
            out = pool.get32((id(conv), "ctx_share"), (c.B, conv.out_channels, c.H, c.W), c.device)
            s16.conv2d(packed16_part(conv, ((lo, conv.in_channels),)), [c], out32=out)
            pre.append(out)
        return pre

    def begin_frame(self, pool, B: int, device):
        """Once per frame, before the loop: clears the InstanceNorm accumulators of the two up-blocks (`run(..., slot=itr)` uses one set
        per iteration; two memsets per frame instead of two statistics launches per iteration)."""
        for blk in (self.conv_16_8, self.conv_8_4):
            up_block_sums(pool, blk, B, device).zero_()
        self._slots_used = set()

    def run(self, pool, g5: torch.Tensor, cands: torch.Tensor, pre, lazy: bool = False, slot=None):
        """g5 = 5 * gradient [N,2,H,W] fp32, cands [N,32,H,W] fp32, pre: `prepare(pool, clist)` of the 3 S16 context tensors
        (64 ch at 1/4, 1/8, 1/16) -> (gradient fp32 [N,2,H,W], context S16 [N,64,H,W]).  `lazy`: the gradient as
        (s16.Taps of residual_head[2], g5, 0.2) for DispRefine.run, which finishes (5*grad + residual) / 5 in its candidate stencil.
        `slot`: see up_block16 (the frame loop passes its iteration index after `begin_frame`)."""
        # The accumulators only ADD: a set may be used once between two begin_frame() calls.  A slot that was already used (a second run
        # with the same index, a caller that never called begin_frame) or lies beyond IN_SUM_SLOTS falls back to the statistics launch.
        used = self.__dict__.setdefault("_slots_used", None)
        if slot is not None and (used is None or slot in used or not 0 <= int(slot) < IN_SUM_SLOTS):
            slot = None
        if slot is not None:
            used.add(slot)

        def feat(conv, srcs, share):
            n = sum(t.C for t in srcs)
            return conv16(pool, conv, srcs, act="relu", addend=share, pc=packed16_part(conv, ((0, n),), with_bias=False))

        gs, cs = self.conv_grad_stem, self.conv_grad_candidate_stem
        # the two stems (update.py:200-205) layer by layer as grouped launches: [2 -> 32 | 32 -> 64] on the fp32-MFMA kernel (the candidates are
        # unbounded; the gradient stem rides the same instance), then [32 -> 32 | 64 -> 64] on S16 — no fork, no join
        with ops.grouped(enabled=GROUP):
            g1 = conv32to16(pool, gs[0], g5, act="relu")
            c1 = conv32to16(pool, cs[0], cands, act="relu")
        with s16.grouped(enabled=GROUP):
            x4_grad = conv16(pool, gs[2], [g1])
            x4_cand = conv16(pool, cs[2], [c1])
        x4 = feat(self.conv_4_4[0], [x4_grad, x4_cand], pre[0])
        x8 = conv16(pool, self.conv_4_8[0], [x4], act="relu")                    # 3x3 stride 2
        x8 = feat(self.conv_8_8[0], [x8], pre[1])
        x16 = conv16(pool, self.conv_8_16[0], [x8], act="relu")                  # 3x3 stride 2
        x16 = feat(self.conv_16_16[0], [x16], pre[2])
        x8_up = up_block16(pool, self.conv_16_8, x16, x8, slot=slot)
        x4_up = up_block16(pool, self.conv_8_4, x8_up, x4, slot=slot)

        rh0, co0 = self.residual_head[0], self.conv_out[0]
        if "noheadfuse" not in _X and rh0.out_channels % 32 == 0:
            # residual_head[0] and conv_out[0] read the same x4_up (update.py:212-214): one 64 -> 128 + 64 launch whose epilogue
            # sends the two channel ranges their own ways — one launch and a fork / join less per iteration
            ctx = pool.get((id(co0), "o"), x4_up.B, co0.out_channels, x4_up.H, x4_up.W, x4_up.device)
            if "notaps" not in _X:
                # ... and residual_head[2] (128 -> 2) is folded into that epilogue as tap partials: the 128-channel hidden tensor is
                # never stored; (5*grad + residual) / 5 (update.py:213) is finished by the consumer
                taps = tap_partials(pool, rh0, self.residual_head[2], [x4_up], pc=packed16_cat([rh0, co0]), out16b=ctx)
                return ((taps, g5, 0.2) if lazy else s16.taps_sum(taps, addend=g5, scale=0.2)), ctx
            h = pool.get((id(rh0), "o"), x4_up.B, rh0.out_channels, x4_up.H, x4_up.W, x4_up.device)
            s16.conv2d(packed16_cat([rh0, co0]), [x4_up], act="relu", out16=h, out16b=ctx, out16_split=rh0.out_channels)
            # (5*grad + residual) / 5 (update.py:213) in the epilogue of the last conv: addend = 5*grad, scale = 1/5
            return conv16(pool, self.residual_head[2], [h], addend=g5, post_scale=0.2, want32=True), ctx

        def head():
            h = conv16(pool, rh0, [x4_up], act="relu")
            return conv16(pool, self.residual_head[2], [h], addend=g5, post_scale=0.2, want32=True)

        grad, ctx = fork_join([head, lambda: conv16(pool, co0, [x4_up], act="relu")], site="heads")
        return grad, ctx


class DispRefine(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.context_compress = _two(192, 96, 96)
        self.disp_f_stem = _two(27, 96, 96, k=1)
        self.conv_fuse = _two(192, 128, 128, tail=nn.ReLU(inplace=True))
        self.w_head = nn.Sequential(_conv(128, 128, 3), nn.ReLU(inplace=True), _conv(128, 9, 1))
        factor = 2 ** args.n_downsample
        self.mask = nn.Sequential(_conv(128, 256, 3), nn.ReLU(inplace=True), _conv(256, factor * factor * 9, 1))

    def propagate_disparity(self, disparity_grad, disparity_map):
        """9 gradient-extrapolated neighbour candidates + 18 gradient differences (update.py:259-289)."""
        buf = ops.propagate_disparity(disparity_grad.float().contiguous(), disparity_map.float().contiguous())
        return buf[:, :9], buf[:, 9:]

    def forward(self, disp_grads, disp, context_disp, context_grad, test_mode=False, fused_outputs=None):
        """fp32 API (update.py:291-305): -> (refined disparity, 0.25 * upsampling mask or None when test_mode).
        `fused_outputs`, when a dict, receives 'delta_disp' (= refined - disp), 'coords1' (= x - refined) and 'flow_x'."""
        pool = pool_of(self)
        cd = to16(pool, context_disp, (id(self), "ctxd_in"))
        cg = to16(pool, context_grad, (id(self), "ctxg_in"))
        refined, mask, extra = self.run(pool, disp_grads.float().contiguous(), disp.float().contiguous(), cd, cg, want_mask=not test_mode)
        if fused_outputs is not None:
            fused_outputs.update(extra)
        return refined, mask

    def run(self, pool, disp_grads: torch.Tensor, disp: torch.Tensor, context_disp: s16.S16, context_grad: s16.S16, want_mask=False,
            motion: s16.S16 = None, warm_pyramid=None, warm_radius: int = 4):
        """-> (refined fp32, mask fp32 or None, dict(delta_disp, coords1, flow_x)).  With `motion`, the next iteration's flow input
        (coords1 - x) also lands in channel 127 of that S16 buffer (tc_stereo.py:180, update.py:126)."""
        # the candidate stencil (with the residual head's last convolution finished from its tap partials) stays on the chain; behind it
        # the context branch (the longer one) continues the chain and the two 1x1 layers of the candidate stem run beside it
        f27 = pool.get((id(self), "f27"), disp.shape[0], 27, disp.shape[2], disp.shape[3], disp.device)
        if isinstance(disp_grads, tuple):        # (tap partials of residual_head[2], 5*grad, 1/5) from DispGradPredictor.run(lazy=True)
            f27, cand9, self._last_grad = s16.taps_propagate(disp_grads[0], disp_grads[1], disp_grads[2], disp, out16=f27)
        else:
            f27, cand9 = s16.propagate_disparity(disp_grads, disp, out16=f27)

        cc, ds = self.context_compress, self.disp_f_stem
        # context_compress (3x3, 192 -> 96 -> 96) beside disp_f_stem (1x1, 27 -> 96 -> 96), layer by layer as grouped launches
        # (the 3x3 halves on the 4-row single-stage tile: a grouped launch allocates the larger LDS size of its two instances)
        t3 = 101411 if context_disp.H * context_disp.W >= 10000 else 0
        with s16.grouped(enabled=GROUP):
            c = conv16(pool, cc[0], [context_disp, context_grad], act="relu", tile_cfg=t3)
            d = conv16(pool, ds[0], [f27], act="relu")
        with s16.grouped(enabled=GROUP):
            context = conv16(pool, cc[2], [c], tile_cfg=t3)
            disp_f = conv16(pool, ds[2], [d])
        fused = conv16(pool, self.conv_fuse[0], [disp_f, context], act="relu")
        fused = conv16(pool, self.conv_fuse[2], [fused], act="relu")
        w = conv16(pool, self.w_head[0], [fused], act="relu")
        coords1, flow_x = torch.empty_like(disp), torch.empty_like(disp)
        if "noblendfuse" not in _X:
            # the blend runs as the epilogue of w_head's 1x1 convolution: one launch less on the iteration's critical chain
            # `warm_pyramid` (the frame's correlation pyramid): the blend also touches the rows the next iteration's lookup reads
            refined, delta = s16.conv1x1_blend(packed16(self.w_head[2]), [w], cand9, disp, coords1, flow_x, flow_x_s16=motion,
                                               flow_x_channel=127, warm_pyramid=warm_pyramid, warm_radius=warm_radius)
        else:
            logits = conv16(pool, self.w_head[2], [w], want32=True)
            refined, delta = s16.softmax_blend(logits, cand9, disp, coords1, flow_x, flow_x_s16=motion, flow_x_channel=127)
        mask = None
        if want_mask:
            m = conv16(pool, self.mask[0], [fused], act="relu")
            mask = conv16(pool, self.mask[2], [m], post_scale=0.25, want32=True)
        return refined, mask, dict(delta_disp=delta, coords1=coords1, flow_x=flow_x)


# ---------------------------------------------------------------------------------------------
# temporal disparity completion (once per frame)
# ---------------------------------------------------------------------------------------------
class DisparityCompletor(nn.Module):
    def __init__(self):
        super().__init__()
        relu = lambda: nn.ReLU(inplace=True)
        mlp = lambda cin, mid, cout: nn.Sequential(_conv(cin, mid, 1), relu(), _conv(mid, cout, 1))
        cin_block = lambda cin, mid, cout, s=1: nn.Sequential(_conv(cin, mid, 3, s), nn.InstanceNorm2d(mid), relu(), _conv(mid, cout, 3))
        self.conv_disp_stem = mlp(1, 64, 64)
        self.conv_cost_stem = mlp(1, 32, 32)
        self.conv_mask_stem = mlp(1, 32, 32)
        self.conv_disp_fuse = mlp(128, 128, 64)
        self.conv_4_4 = cin_block(192, 192, 64)
        self.conv_4_8 = cin_block(64, 64, 64, 2)
        self.conv_8_8 = cin_block(192, 192, 64)
        self.conv_8_16 = cin_block(64, 64, 64, 2)
        self.conv_16_16 = cin_block(192, 192, 64)
        self.conv_16_8 = Conv2x_IN(64, 64, deconv=True, concat=False, keep_concat=False, IN=True)
        self.conv_8_4 = Conv2x_IN(64, 64, deconv=True, concat=False, keep_concat=False, IN=True)
        self.disp_head = _two(64, 128, 1)
        self.w_head = _two(64, 128, 1, tail=nn.Sigmoid())
        self.conv_out16_disp = cin_block(192, 192, 128)
        self.conv_out8_disp = cin_block(192, 192, 128)
        self.conv_out4_disp = cin_block(192, 192, 128)

    def _cin(self, seq, srcs, act="none"):
        """conv -> InstanceNorm -> ReLU -> conv blocks (update.py:325-367)."""
        return hip_conv(seq[3], [ops.instance_norm(hip_conv(seq[0], srcs), act="relu")], act=act)

    def _forward_hip(self, disp, cost, mask, ctx, tanh_nets=False):
        d = (disp / 10).float().contiguous()
        stems = [hip_seq(self.conv_disp_stem, [d]), hip_seq(self.conv_cost_stem, [cost.float().contiguous()]),
                 hip_seq(self.conv_mask_stem, [(mask - 0.5).float().contiguous()])]
        x4_disp = hip_seq(self.conv_disp_fuse, stems)
        x4 = self._cin(self.conv_4_4, [x4_disp, ctx[0]])
        x8 = self._cin(self.conv_8_8, [self._cin(self.conv_4_8, [x4]), ctx[1]])
        x16_out = self._cin(self.conv_16_16, [self._cin(self.conv_8_16, [x8]), ctx[2]])
        x8_out = hip_up_block(self.conv_16_8, x16_out, x8)
        x4_out = hip_up_block(self.conv_8_4, x8_out, x4)
        disp_mono = hip_seq(self.disp_head, [x4_out])
        w = hip_conv(self.w_head[2], [hip_conv(self.w_head[0], [x4_out], act="relu")], act="sigmoid")
        completed = (w * d + (1 - w) * disp_mono) * 10
        na = "tanh" if tanh_nets else "none"                # the caller's torch.tanh (tc_stereo.py:167) in the last convolutions' epilogues
        nets = [self._cin(self.conv_out4_disp, [x4_out, ctx[0]], na), self._cin(self.conv_out8_disp, [x8_out, ctx[1]], na),
                self._cin(self.conv_out16_disp, [x16_out, ctx[2]], na)]
        return completed, disp_mono * 10, w, nets

    def run16(self, pool, disp, cost, mask, ctx, tanh_nets=False):
        """The block on pre-split (S16) tensors — the loop's kernels (tcs_conv2d_s16, S16 InstanceNorm, the up-blocks of the gradient
        predictor) instead of the fp32-tensor ones: -> (completed fp32, 10 * disp_mono fp32, w fp32, three S16 hidden states).  `ctx`: the
        three fp32 context tensors (update.py:369-399)."""
        d = (disp / 10).float().contiguous()

        def stem(seq, x):               # 1x1 (1 -> C) -> ReLU -> 1x1 (C -> C)
            return conv16(pool, seq[2], [conv32to16(pool, seq[0], x.float().contiguous(), act="relu")])

        def cin(seq, srcs, act="none"):  # conv -> InstanceNorm -> ReLU -> conv (update.py:325-367)
            y = conv16(pool, seq[0], srcs)
            return conv16(pool, seq[3], [s16.instance_norm(y, act="relu", out=y)], act=act)

        c16 = [to16(pool, c, (id(self), "ctx", i)) for i, c in enumerate(ctx)]
        x4_disp = conv16(pool, self.conv_disp_fuse[2], [conv16(pool, self.conv_disp_fuse[0], [stem(self.conv_disp_stem, d),
                         stem(self.conv_cost_stem, cost), stem(self.conv_mask_stem, mask - 0.5)], act="relu")])
        x4 = cin(self.conv_4_4, [x4_disp, c16[0]])
        x8 = cin(self.conv_8_8, [cin(self.conv_4_8, [x4]), c16[1]])
        x16_out = cin(self.conv_16_16, [cin(self.conv_8_16, [x8]), c16[2]])
        x8_out = up_block16(pool, self.conv_16_8, x16_out, x8)
        x4_out = up_block16(pool, self.conv_8_4, x8_out, x4)
        disp_mono = conv16(pool, self.disp_head[2], [conv16(pool, self.disp_head[0], [x4_out], act="relu")], want32=True)
        w = conv16(pool, self.w_head[2], [conv16(pool, self.w_head[0], [x4_out], act="relu")], act="sigmoid", want32=True)
        completed = (w * d + (1 - w) * disp_mono) * 10
        na = "tanh" if tanh_nets else "none"                # the caller's torch.tanh (tc_stereo.py:167) in the last convolutions' epilogues
        nets = [cin(self.conv_out4_disp, [x4_out, c16[0]], na), cin(self.conv_out8_disp, [x8_out, c16[1]], na),
                cin(self.conv_out16_disp, [x16_out, c16[2]], na)]
        return completed, disp_mono * 10, w, nets

    def forward(self, disp, cost, mask, context_list, tanh_nets=False):
        """`tanh_nets` (not in the reference's signature): return tanh of the three new hidden states, which is what
        TCStereo.forward applies to them next (tc_stereo.py:167)."""
        if not disp.is_cuda:
            raise RuntimeError("DisparityCompletor: CPU tensor (the hot path has no CPU fallback)")
        if "dc32" in _X:                                    # A/B: round 2's fp32-tensor kernels
            return self._forward_hip(disp, cost, mask, [c.float().contiguous() for c in context_list], tanh_nets)
        completed, mono, w, nets = self.run16(pool_of(self), disp, cost, mask, [c.float().contiguous() for c in context_list], tanh_nets)
        return completed, mono, w, [n.float() for n in nets]
