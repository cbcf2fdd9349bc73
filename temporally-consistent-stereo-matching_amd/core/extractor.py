"""Feature / context extractors (reference: core/extractor.py): the same module tree, so that the reference's
checkpoints load with strict=True (same attribute names, same parameter shapes).  BASELINE.json's north_star leaves
the extractor on PyTorch-ROCm; the batch/group-norm variants do stay there, but the whole `none` / `instance` norm
configuration (7x7 RGB stem, 3x3 / 1x1 trunk, heads) runs on tcs_conv2d like the refinement loop: on these shapes the
fp16-split kernel is 2-2.5x faster than MIOpen's fp32 solvers (tools/bench_extractor_convs.py) and the ReLU / residual-add
tails are fused into its epilogue; no MIOpen kernel is left in the steady-state trace of the shipped configuration."""
import torch
import torch.nn as nn

from tcs_mi355 import ops, s16


def _norm(kind, ch, stem=False):
    if kind == "group":
        return nn.GroupNorm(num_groups=8 if stem else ch // 8, num_channels=ch)
    if kind == "batch":
        return nn.BatchNorm2d(ch)
    if kind == "instance":
        return nn.InstanceNorm2d(ch)
    if kind == "none":
        return nn.Sequential()
    raise ValueError(f"unknown norm {kind!r}")


def _init(module, fan_mode):
    for m in module.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode=fan_mode, nonlinearity="relu")
        elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)) and m.weight is not None:
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)


class ResidualBlock(nn.Module):
    """Two 3x3 convs with a (projected, when stride or width changes) identity (extractor.py:5-58)."""

    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm_kind = norm_fn
        self.norm1, self.norm2 = _norm(norm_fn, planes), _norm(norm_fn, planes)
        self.downsample = None
        if stride != 1 or in_planes != planes:
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)

    def forward(self, x):
        if x.is_cuda and self.norm_kind in ("none", "instance") and _hip_trunk():
            return self._forward_hip(x)
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(skip + y)

    def run16(self, pool, x: "s16.S16") -> "s16.S16":
        """The block on pre-split (S16) tensors (tcs_conv2d_s16; `none` / `instance` norm): conv -> [IN] -> ReLU -> conv -> [IN] -> ReLU,
        + skip, ReLU, with the tails in the convolution / InstanceNorm epilogues; the projection shortcut is a 1x1 (stride s) conv."""
        from core.update import conv16
        inorm = self.norm_kind == "instance"
        skip = x
        if self.downsample is not None:
            skip = conv16(pool, self.downsample[0], [x])
            if inorm:
                skip = s16.instance_norm(skip, out=skip)
        if not inorm:
            y = conv16(pool, self.conv1, [x], act="relu")
            return conv16(pool, self.conv2, [y], act="relu_add_relu", addend16=skip)
        y = conv16(pool, self.conv1, [x])
        y = s16.instance_norm(y, act="relu", out=y)
        z = conv16(pool, self.conv2, [y])
        return s16.instance_norm(z, act="relu_add_relu", addend=skip, out=z)

    def _forward_hip(self, x):
        """relu(skip + relu(norm2(conv2(relu(norm1(conv1(x))))))) (extractor.py:44-58) in 2-6 launches."""
        from core.update import hip_conv, packed
        x = x.float().contiguous()
        inorm = self.norm_kind == "instance"
        if self.downsample is None:
            skip = x
        else:
            dconv = self.downsample[0]                         # 1x1, stride 1 or 2: a strided 1x1 conv reads every other pixel
            xs = x if dconv.stride == (1, 1) else x[:, :, ::dconv.stride[0], ::dconv.stride[1]].contiguous()
            skip = ops.conv2d(packed(dconv), [xs])
            if inorm:
                skip = ops.instance_norm(skip)
        if not inorm:
            y = hip_conv(self.conv1, [x], act="relu")
            return hip_conv(self.conv2, [y], act="relu_add_relu", addend=skip)
        y = ops.instance_norm(hip_conv(self.conv1, [x]), act="relu")
        out = ops.instance_norm(hip_conv(self.conv2, [y]), act="relu", addend=skip)      # relu(IN(.)) + skip
        return torch.relu_(out)                                # no-op when skip >= 0, kept for exactness


def _hip_trunk() -> bool:
    """TCS_MI355_EXTRACTOR=torch keeps the whole extractor on PyTorch-ROCm (used by tests as the comparison leg)."""
    import os
    return os.environ.get("TCS_MI355_EXTRACTOR", "hip") != "torch"


def hip_stem(enc, x):
    """relu(norm1(conv1(x))): the 7x7 RGB stem (extractor.py:205-207,270-272).  Stride 1 with `none` / `instance` norm runs on
    tcs_conv2d's 7x7 kernel (ReLU fused) and k_instance_norm; other settings (stride-2 stem, batch / group norm) stay on PyTorch."""
    kind = enc.norm_fn
    if x.is_cuda and _hip_trunk() and enc.conv1.stride == (1, 1) and kind in ("none", "instance"):
        from core.update import packed
        x = x.float().contiguous()
        pc = packed(enc.conv1)              # 7x7: pack_conv keeps the fp32 layout
        if kind == "none":
            return ops.conv2d(pc, [x], act="relu")
        return ops.instance_norm(ops.conv2d(pc, [x]), act="relu")
    return enc.relu1(enc.norm1(enc.conv1(x)))


def _s16_ok(kind: str, hw: int) -> bool:
    """S16 trunk: `none` norm anywhere, `instance` norm on planes the S16 InstanceNorm kernels are sized for (<= 1/4 scale)."""
    return kind == "none" or (kind == "instance" and hw <= 40960)


def hip_head16(pool, f, x: "s16.S16", act: str = "none") -> torch.Tensor:
    """An output head on an S16 trunk tensor -> fp32 NCHW (what the correlation build / context convolutions consume); `act`: an
    activation the caller would apply next, in the final convolution's epilogue."""
    from core.update import conv16
    if isinstance(f, nn.Sequential):
        for m in f:
            if isinstance(m, nn.Conv2d):
                return conv16(pool, m, [x], act=act, want32=True)
            x = m.run16(pool, x)
        raise ValueError("head without a final convolution")
    return conv16(pool, f, [x], act=act, want32=True)


def hip_head(f, x):
    """An output head: Conv2d, or Sequential(ResidualBlock, Conv2d) (extractor.py:221-238); `x` may be an S16 trunk tensor."""
    if isinstance(x, s16.S16):
        from core.update import pool_of
        return hip_head16(pool_of(f), f, x)
    if not (x.is_cuda and _hip_trunk()):
        return f(x)
    from core.update import hip_conv
    if isinstance(f, nn.Sequential):
        for m in f:
            x = hip_conv(m, [x]) if isinstance(m, nn.Conv2d) else m(x)
        return x
    return hip_conv(f, [x])


def _stage(in_planes, dim, norm_fn, stride):
    return nn.Sequential(ResidualBlock(in_planes, dim, norm_fn, stride), ResidualBlock(dim, dim, norm_fn, 1))


class BasicEncoder(nn.Module):
    """Matching-feature encoder used when shared_backbone is off (extractor.py:119-192)."""

    def __init__(self, output_dim=128, norm_fn="batch", dropout=0.0, downsample=3):
        super().__init__()
        self.norm_fn, self.downsample = norm_fn, downsample
        self.norm1 = _norm(norm_fn, 64, stem=True)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=1 + (downsample > 2), padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = _stage(64, 64, norm_fn, 1)
        self.layer2 = _stage(64, 96, norm_fn, 1 + (downsample > 1))
        self.layer3 = _stage(96, 128, norm_fn, 1 + (downsample > 0))
        self.conv2 = nn.Conv2d(128, output_dim, 1)
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        _init(self, "fan_out")

    def forward(self, x, dual_inp=False):
        parts = None
        if isinstance(x, (tuple, list)):
            parts = x[0].shape[0]
            x = torch.cat(x, 0)
        x = hip_stem(self, x)
        x = hip_head(self.conv2, self.layer3(self.layer2(self.layer1(x))))
        if self.training and self.dropout is not None:
            x = self.dropout(x)
        return x.split(parts, 0) if parts is not None else x


class MultiBasicEncoder(nn.Module):
    """Context network with per-scale heads at 1/4, 1/8, 1/16 (the strides are fixed; the
    08/16/32 attribute names are historical — extractor.py:195-296)."""

    def __init__(self, output_dim=[128], norm_fn="batch", dropout=0.0, downsample=3):
        super().__init__()
        self.norm_fn, self.downsample = norm_fn, downsample
        self.norm1 = _norm(norm_fn, 64, stem=True)
        self.conv1 = nn.Conv2d(3, 64, 7, stride=1, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        self.layer1 = _stage(64, 64, norm_fn, 1)
        self.layer2 = _stage(64, 96, norm_fn, 2)
        self.layer3 = _stage(96, 128, norm_fn, 2)
        self.layer4 = _stage(128, 128, norm_fn, 2)
        self.layer5 = _stage(128, 128, norm_fn, 2)

        def head(ch, with_block):
            conv = nn.Conv2d(128, ch, 3, padding=1)
            return nn.Sequential(ResidualBlock(128, 128, norm_fn, 1), conv) if with_block else conv

        self.outputs08 = nn.ModuleList([head(d[2], True) for d in output_dim])
        self.outputs16 = nn.ModuleList([head(d[1], True) for d in output_dim])
        self.outputs32 = nn.ModuleList([head(d[0], False) for d in output_dim])
        self.dropout = nn.Dropout2d(p=dropout) if dropout > 0 else None
        _init(self, "fan_out")

    def can16(self, x) -> bool:
        """The all-S16 path: HIP device tensor, `none` norm, stride-1 stem."""
        from core.update import _X
        return bool(x.is_cuda and _hip_trunk() and self.norm_fn == "none" and self.conv1.stride == (1, 1) and "noext16" not in _X)

    def trunk16(self, x, right=None, raw_images=False) -> "s16.S16":
        """Stem + layer1..3 on pre-split tensors (`none` norm): the stem writes S16 and every residual block stays S16.
        `right`: a second image batch appended to `x` along the batch inside the stem (no torch.cat); `raw_images`: 0..255 inputs,
        normalised to [-1, 1] by the stem's input staging (tc_stereo.py:101-107)."""
        from core.update import conv32to16, pool_of
        pool = pool_of(self)
        x = conv32to16(pool, self.conv1, x.float().contiguous(), act="relu", image_pair=right, in_transform=int(raw_images))   # 7x7 stem, S16 epilogue
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                x = blk.run16(pool, x)
        return x

    def heads16(self, x: "s16.S16", dual_inp, num_layers, relu_context=False):
        """The per-scale heads on the S16 trunk; only their final convolutions produce fp32.  With `dual_inp` the trunk holds
        left and right images (batch-major) and the heads read the left half.  `relu_context`: the second head of each scale (the
        context features) comes out ReLU'd, which is what TCStereo.forward applies to it next (tc_stereo.py:148)."""
        from core.update import pool_of
        pool = pool_of(self)
        if dual_inp:
            x = s16.S16(x.data[: x.B // 2], x.C)
        acts = lambda heads: [("relu" if (relu_context and j == 1) else "none") for j in range(len(heads))]
        scales = [[hip_head16(pool, f, x, a) for f, a in zip(self.outputs08, acts(self.outputs08))]]
        if num_layers >= 2:
            y = x
            for blk in self.layer4:
                y = blk.run16(pool, y)
            scales.append([hip_head16(pool, f, y, a) for f, a in zip(self.outputs16, acts(self.outputs16))])
        if num_layers >= 3:
            z = y
            for blk in self.layer5:
                z = blk.run16(pool, z)
            scales.append([hip_head16(pool, f, z, a) for f, a in zip(self.outputs32, acts(self.outputs32))])
        return scales

    def forward(self, x, dual_inp=False, num_layers=3):
        if self.can16(x):
            trunk = self.trunk16(x)
            return (*self.heads16(trunk, dual_inp, num_layers), *((trunk,) if dual_inp else ()))
        x = hip_stem(self, x)
        x = self.layer3(self.layer2(self.layer1(x)))
        tail = ()
        if dual_inp:
            tail = (x,)
            x = x[: x.shape[0] // 2]
        scales = [[hip_head(f, x) for f in self.outputs08]]
        if num_layers >= 2:
            y = self.layer4(x)
            scales.append([hip_head(f, y) for f in self.outputs16])
        if num_layers >= 3:
            z = self.layer5(y)
            scales.append([hip_head(f, z) for f in self.outputs32])
        return (*scales, *tail)
