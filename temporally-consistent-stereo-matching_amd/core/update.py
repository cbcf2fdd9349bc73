"""Drop-in for the reference's core/update.py: the refinement-loop blocks on the HIP library.

Module and parameter names equal the reference's, so `load_state_dict(strict=True)` accepts its
checkpoints; `nn.Conv2d` objects are kept as parameter containers only.  Their forward passes run
through `tcs_conv2d` (implicit GEMM on the matrix cores with virtual concat and fused epilogues), so
a ConvGRU is two launches instead of ~15 ATen ops, and the 24 host-synchronising NaN asserts per
iteration of the reference (update.py:27-35,58-67,78-86,155-158) are gone.  Every layer — including
the stride-2 convs, the transposed-conv + InstanceNorm up-blocks of both U-Nets and the
DisparityCompletor — runs on the HIP library; nothing here calls MIOpen.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from core.utils.basic_layers import Conv2x_IN
from tcs_mi355 import ops
from tcs_mi355.streams import fork_join


# ---------------------------------------------------------------------------------------------
# packed-weight cache: one device-side repack per conv, redone only if the parameter changes
# ---------------------------------------------------------------------------------------------
# Contraction arithmetic of tcs_conv2d: "f16x3" = fp16 hi/lo split on the half-precision matrix pipe with fp32
# accumulation (fp32-equivalent accuracy, ~5x the fp32 MFMA rate), "f32" = fp32-in/fp32-out MFMA.  A layer can pin
# its own mode with `conv._tcs_math`.
CONV_MATH = os.environ.get("TCS_MI355_MATH", "f16x3")


def packed(conv: nn.Conv2d) -> ops.PackedConv:
    w, b = conv.weight, conv.bias
    math = getattr(conv, "_tcs_math", None) or CONV_MATH
    if conv.stride == (2, 2):
        math = "f16x3"          # stride-2 convolutions exist on the fp16-split kernel only (also under TCS_MI355_MATH=f32)
    key = (w.data_ptr(), w._version, w.device, None if b is None else (b.data_ptr(), b._version), math)
    hit = getattr(conv, "_tcs_packed", None)
    if hit is None or hit[0] != key:
        if conv.stride not in ((1, 1), (2, 2)) or conv.dilation != (1, 1) or conv.groups != 1 or \
                conv.padding != (conv.kernel_size[0] // 2,) * 2:
            raise NotImplementedError(f"tcs_conv2d covers stride-1/2 'same'-padded convolutions, got {conv}")
        hit = (key, ops.pack_conv(w, b, math))
        conv._tcs_packed = hit
    return hit[1]


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

    def forward(self, h, x):
        return self._step(h, [hip_seq(self.convs, [x])])


# ---------------------------------------------------------------------------------------------
# disparity-space update block
# ---------------------------------------------------------------------------------------------
class FlowHead(nn.Module):
    def __init__(self, input_dim=128, hidden_dim=256, output_dim=2):
        super().__init__()
        self.conv1 = _conv(input_dim, hidden_dim, 3)
        self.conv2 = _conv(hidden_dim, output_dim, 3)
        self.relu = nn.ReLU(inplace=True)

    def forward(self, x):
        y = hip_conv(self.conv1, [x], act="relu")
        if self.conv2.out_channels == 1:
            return ops.conv3x3_cout1(y, self.conv2.weight, self.conv2.bias)      # 256 -> 1: reduction kernel, not a 32-wide MFMA tile
        return hip_conv(self.conv2, [y])


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
        """`out`, when given, is a [N,128,H,W] buffer whose channel 127 ALREADY holds `flow` (the blend kernel of the
        previous iteration writes it there): only channels 0..126 are produced here."""
        flow = flow.float().contiguous()
        cor, flo = fork_join([
            lambda: hip_conv(self.convc2, [hip_conv(self.convc1, [corr], act="relu")], act="relu"),
            lambda: hip_conv(self.convf2, [hip_conv(self.convf1, [flow], act="relu")], act="relu")], site="enc")
        n, _, h, w = flow.shape
        prefilled = out is not None
        if not prefilled:
            out = torch.empty(n, 128, h, w, dtype=torch.float32, device=flow.device)
        hip_conv(self.conv, [cor, flo], act="relu", out=out)     # channels 0..126 in place: no torch.cat
        if not prefilled:
            out[:, 127:128].copy_(flow)
        return out


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
        """Coarse-to-fine GRU sweep (update.py:145-168); `net` is updated in place like the reference.
        `motion_out`: see BasicMotionEncoder.forward(out=...)."""
        n = self.args.n_gru_layers

        def coarse():            # gru32 -> gru16: independent of the motion encoder (which only feeds gru08)
            if iter32:
                net[2] = self.gru32(net[2], *inp[2], pool2x(net[1]))
            if iter16:
                extra = (interp(net[2], net[1]),) if n > 2 else ()
                net[1] = self.gru16(net[1], *inp[1], pool2x(net[0]), *extra)
            return interp(net[1], net[0]) if (iter08 and n > 1) else None

        if iter08:
            # the encoder (which forks again) stays on the current stream: ROCm 7.2 segfaults in hipStreamEndCapture when
            # a side branch of a captured fork forks a second time
            motion, up16 = fork_join([lambda: self.encoder(flow, corr, out=motion_out), coarse], site="coarse")
            extra = (up16,) if n > 1 else ()
            net[0] = self.gru08(net[0], *inp[0], motion, *extra)
        else:
            coarse()
        if not update:
            return net
        return net, self.flow_head(net[0])


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

    def _up(self, block: Conv2x_IN, x, rem):
        return hip_up_block(block, x, rem)

    def forward(self, disp_grad, disp, clist, g5=None, cands=None):
        """`g5`, when given, is 5*disp_grad already produced by the gradient kernel (saves an elementwise launch)."""
        disp = disp.float().contiguous()
        if g5 is None:
            g5 = (5 * disp_grad).contiguous()                    # update.py:199
        if cands is None:
            cands = ops.grad_candidates(disp)                    # [N,32,H,W] (update.py:202-204)
        x4_grad, x4_cand = fork_join([lambda: hip_seq(self.conv_grad_stem, [g5]),
                                      lambda: hip_seq(self.conv_grad_candidate_stem, [cands])], site="stems")
        x4 = hip_seq(self.conv_4_4, [x4_grad, x4_cand, clist[0]])
        x8 = hip_seq(self.conv_4_8, [x4])                                       # 3x3 stride 2
        x8 = hip_seq(self.conv_8_8, [x8, clist[1]])
        x16 = hip_seq(self.conv_8_16, [x8])                                     # 3x3 stride 2
        x16 = hip_seq(self.conv_16_16, [x16, clist[2]])
        x8_up = self._up(self.conv_16_8, x16, x8)
        x4_up = self._up(self.conv_8_4, x8_up, x4)
        def head():
            # (5*grad + residual) / 5 (update.py:213) in the epilogue of the last conv: addend = 5*grad, scale = 1/5
            h = hip_conv(self.residual_head[0], [x4_up], act="relu")
            return hip_conv(self.residual_head[2], [h], addend=g5, post_scale=0.2)

        grad, ctx = fork_join([head, lambda: hip_seq(self.conv_out, [x4_up])], site="heads")
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

    def _prop(self, disparity_grad, disparity_map):
        return ops.propagate_disparity(disparity_grad.float().contiguous(), disparity_map.float().contiguous())

    def propagate_disparity(self, disparity_grad, disparity_map):
        """9 gradient-extrapolated neighbour candidates + 18 gradient differences (update.py:259-289)."""
        buf = self._prop(disparity_grad, disparity_map)
        return buf[:, :9], buf[:, 9:]

    def forward(self, disp_grads, disp, context_disp, context_grad, test_mode=False, fused_outputs=None):
        """`fused_outputs`, when a dict, receives 'delta_disp' (= refined - disp) and 'coords1' (= x - refined) straight
        from the blend kernel (tc_stereo.py:198-202), saving two elementwise launches."""
        disp = disp.float().contiguous()
        def cand_branch():
            f27 = self._prop(disp_grads, disp)                    # cat(candidates, matrix) laid out by the kernel
            return f27, hip_seq(self.disp_f_stem, [f27])

        context, (feats27, disp_f) = fork_join([lambda: hip_seq(self.context_compress, [context_disp, context_grad]), cand_branch], site="refine")
        fused = hip_seq(self.conv_fuse, [disp_f, context])
        logits = hip_seq(self.w_head, [fused])
        if fused_outputs is not None:
            coords1 = torch.empty_like(disp)
            flow_x = torch.empty_like(disp)
            refined, delta = ops.softmax_blend(logits, feats27, disp_q=disp, want_delta=True, coords1=coords1, flow_x=flow_x,
                                               flow_x_channel=fused_outputs.get("flow_x_channel"))
            fused_outputs.update(delta_disp=delta, coords1=coords1, flow_x=flow_x)
        else:
            refined, _ = ops.softmax_blend(logits, feats27)
        mask = None
        if not test_mode:
            mask = hip_conv(self.mask[2], [hip_conv(self.mask[0], [fused], act="relu")], post_scale=0.25)
        return refined, mask


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

    def _cin(self, seq, srcs):
        """conv -> InstanceNorm -> ReLU -> conv blocks (update.py:325-367)."""
        return hip_conv(seq[3], [ops.instance_norm(hip_conv(seq[0], srcs), act="relu")])

    def _forward_hip(self, disp, cost, mask, ctx):
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
        nets = [self._cin(self.conv_out4_disp, [x4_out, ctx[0]]), self._cin(self.conv_out8_disp, [x8_out, ctx[1]]),
                self._cin(self.conv_out16_disp, [x16_out, ctx[2]])]
        return completed, disp_mono * 10, w, nets

    def forward(self, disp, cost, mask, context_list):
        if not disp.is_cuda:
            raise RuntimeError("DisparityCompletor: CPU tensor (the hot path has no CPU fallback)")
        return self._forward_hip(disp, cost, mask, [c.float().contiguous() for c in context_list])
