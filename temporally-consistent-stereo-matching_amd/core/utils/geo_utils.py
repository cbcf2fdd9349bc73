"""Drop-in for the reference's core/utils/geo_utils.py on the HIP library.

Every function keeps the reference's name, argument order and return convention; the chains of
elementwise torch ops + tiny matmuls + NaN asserts of the reference become single fused kernels
(csrc/tcs_warp.hip, csrc/tcs_stencil.hip).
"""
import torch

from tcs_mi355 import ops


def cal_relative_transformation(T1, T2):
    """T2 @ inv(T1): pose1 -> pose2 for world->camera matrices (geo_utils.py:148-155)."""
    return torch.matmul(T2, torch.linalg.inv(T1))


def warp(disp, fmap, relative_T, K, K_inv, baseline):
    """Forward-project the previous frame's disparity and features into the current frame
    (geo_utils.py:158-198).  Returns (disp [N,1,H,W], fmap [N,C,H,W], mask [N,1,H,W])."""
    d, f, m, _ = ops.warp_forward(disp.float().contiguous(), fmap.float().contiguous(), relative_T, K, K_inv, baseline)
    return d, f, m


def get_backward_grid(disp, relative_T, K, K_inv, baseline):
    """Previous-frame pixel coordinates of every current pixel (geo_utils.py:201-236): [N,2,H,W]."""
    return ops.backward_grid(disp.float().contiguous(), relative_T, K, K_inv, baseline)


def disp2disp_gradient_xy(disp):
    """Forward differences on the replicate-padded map (geo_utils.py:115-132) -> (grads [N,2,H,W], edge_mask)."""
    grads = ops.disp_gradient_xy(disp.float().contiguous())
    edge_mask = (grads[:, :1].abs() < 5) & (grads[:, 1:].abs() < 5)
    return grads, edge_mask


def disp2disp_grad_candidates(disp, level=2):
    """Plane-fit gradient candidates from neighbour cross products (geo_utils.py:73-101) -> [N,2,16,H,W]."""
    if level != 2:
        raise NotImplementedError("the model uses level=2 (update.py:202)")
    n, _, h, w = disp.shape
    return ops.grad_candidates(disp.float().contiguous()).view(n, 2, 16, h, w)
