"""Drop-in for the forward path of the reference's core/utils/splatting/softsplat.py.

The CuPy-JIT CUDA kernel `softsplat_out` (softsplat.py:285-335) is replaced by the HIP kernel behind
`tcs_softsplat_sum`; this wrapper keeps the reference's mode handling (softsplat.py:232-274).
The model itself does not go through here: `warp()` uses the fused `tcs_warp_forward`, which folds
the pre-scale, the splat and the normalisation into one launch sequence.  Backward kernels
(softsplat_ingrad / softsplat_flowgrad) are training-only and out of scope.
"""
import torch

from tcs_mi355 import ops


class softsplat_func:
    """Inference-only stand-in for the autograd Function: `.apply(tenIn, tenFlow)` -> summed splat."""

    @staticmethod
    def apply(tenIn, tenFlow):
        return ops.softsplat_sum(tenIn.float().contiguous(), tenFlow.float().contiguous())


def softsplat(tenIn, tenFlow, tenMetric, strMode, valid_mask=None):
    kind, _, eps_mode = strMode.partition("-")
    if kind not in ("sum", "avg", "linear", "soft"):
        raise ValueError(f"unknown splatting mode {strMode!r}")
    if (tenMetric is None) != (kind in ("sum", "avg")):
        raise ValueError(f"mode {strMode!r} {'forbids' if kind in ('sum', 'avg') else 'needs'} a metric")
    ones = tenIn.new_ones(tenIn.shape[0], 1, tenIn.shape[2], tenIn.shape[3])
    valid_mask = ones if valid_mask is None else valid_mask
    x = tenIn * valid_mask
    if kind == "avg":
        x = torch.cat([x, ones * valid_mask], 1)
    elif kind == "linear":
        x = torch.cat([x * tenMetric, tenMetric * valid_mask], 1)
    elif kind == "soft":
        e = tenMetric.exp()
        x = torch.cat([x * e, e * valid_mask], 1)
    out = softsplat_func.apply(x, tenFlow)
    mask = None
    if kind != "sum":
        norm = out[:, -1:]
        mask = (norm != 0).float()
        norm = norm.clip(1e-7, None) if eps_mode == "clipeps" else norm + 1e-7
        out = out[:, :-1] / norm
    return out, mask
