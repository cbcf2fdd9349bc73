"""Drop-in for the hot-path part of the reference's core/utils/utils.py.

`bilinear_sampler` and `coords_grid` run on the HIP library; `InputPadder` is re-exported from the
harness.  The training-only helpers of the reference file (forward_interpolate, gauss_blur,
MedianPool2d) are out of scope.
"""
import torch

from tcs_mi355 import ops
from tcs_mi355.harness import InputPadder  # noqa: F401  (core/utils/utils.py:7-48)


def coords_grid(batch, ht, wd, device=None):
    """[batch,2,ht,wd] with channel 0 = x, channel 1 = y (core/utils/utils.py:100-103)."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device, dtype=torch.float32),
                            torch.arange(wd, device=device, dtype=torch.float32), indexing="ij")
    return torch.stack((xs, ys), 0).unsqueeze(0).repeat(batch, 1, 1, 1)


def bilinear_sampler(img, coords, mode="bilinear", mask=False, align_corners=True):
    """Pixel-coordinate bilinear sampling, zeros outside (core/utils/utils.py:82-97).
    img [N,C,H,W]; coords [N,Ho,Wo,2] (x,y) like the reference."""
    if mode != "bilinear" or not align_corners:
        raise NotImplementedError("only mode='bilinear', align_corners=True is on the hot path")
    grid = coords.permute(0, 3, 1, 2).contiguous().float()
    out = ops.bilinear_sample(img.float().contiguous(), grid)
    if mask:
        H, W = img.shape[-2:]
        x, y = coords[..., :1], coords[..., 1:]
        inside = (x > 0) & (y > 0) & (x < W - 1) & (y < H - 1)
        return out, inside.float()
    return out


def upflow8(flow, mode="bilinear"):
    """Fallback upsampling of the reference (core/utils/utils.py:106-108); unused when a convex mask exists."""
    n, c, h, w = flow.shape
    return 8 * ops.resize_bilinear(flow.float().contiguous(), 8 * h, 8 * w)
