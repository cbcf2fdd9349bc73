"""Conv + InstanceNorm + LeakyReLU building blocks of the two U-Nets (reference:
core/utils/basic_layers.py).  Parameter containers: the hot path runs them through the HIP library
(core/update.py:hip_up_block — transposed conv as a 3x3 conv + pixel-shuffle epilogue, k_instance_norm); the
`forward` methods below are plain-PyTorch equivalents kept for API parity and are not called by TCStereo."""
import torch
import torch.nn as nn
import torch.nn.functional as F


class BasicConv_IN(nn.Module):
    """conv(bias=False) -> optional InstanceNorm -> optional LeakyReLU(0.01) (basic_layers.py:9-35)."""

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, IN=True, relu=True, **kwargs):
        super().__init__()
        if is_3d:
            raise NotImplementedError("3-D variants are not used by TC-Stereo")
        make = nn.ConvTranspose2d if deconv else nn.Conv2d
        self.conv = make(in_channels, out_channels, bias=False, **kwargs)
        self.IN = nn.InstanceNorm2d(out_channels)
        self.use_in, self.relu = IN, relu

    def forward(self, x):
        y = self.conv(x)
        y = self.IN(y) if self.use_in else y
        return F.leaky_relu(y, 0.01) if self.relu else y


class Conv2x_IN(nn.Module):
    """x2 up (4x4 transposed conv, stride 2) or down block with a skip connection
    (basic_layers.py:38-77).  `conv1` always normalises and activates; `conv2` obeys IN/relu."""

    def __init__(self, in_channels, out_channels, deconv=False, is_3d=False, concat=True, keep_concat=True, IN=True,
                 relu=True, keep_dispc=False):
        super().__init__()
        if is_3d:
            raise NotImplementedError("3-D variants are not used by TC-Stereo")
        self.concat = concat
        self.conv1 = BasicConv_IN(in_channels, out_channels, deconv, IN=True, relu=True,
                                  kernel_size=4 if deconv else 3, stride=2, padding=1)
        c2_in = out_channels * 2 if concat else out_channels
        c2_out = out_channels * 2 if (concat and keep_concat) else out_channels
        self.conv2 = BasicConv_IN(c2_in, c2_out, False, IN=IN, relu=relu, kernel_size=3, stride=1, padding=1)

    def forward(self, x, rem):
        x = self.conv1(x)
        if x.shape != rem.shape:
            x = F.interpolate(x, size=rem.shape[-2:], mode="nearest")
        return self.conv2(_merge(x, rem, self.concat))


def _merge(x, rem, concat):
    return torch.cat((x, rem), 1) if concat else x + rem
