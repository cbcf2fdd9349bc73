"""Evaluation harness around the hot path: the caller side of `TCStereo.forward`.

Counterpart of validate_tartanair / validate_temporal_things (evaluate_stereo.py:119-223,264-345):
pads each frame (and shifts K), threads the temporal state dict through the model, and accumulates
EPE / D1 / D3 with the reference's validity mask and mask-rate weighting.  No wandb, no datasets:
sequences come from `tcs_mi355.synth` (or from files, when present, via `tcs_mi355.formats`).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence as Seq

import numpy as np
import torch
import torch.nn.functional as F


class InputPadder:
    """Replicate-pad [.., H, W] tensors up to a multiple of `divis_by` and move the principal point
    of K by the left/top pad (core/utils/utils.py:7-48).  'sintel' mode splits the padding evenly,
    any other mode pads bottom only (and left/right evenly)."""

    def __init__(self, dims, mode: str = "sintel", divis_by: int = 8):
        self.ht, self.wd = int(dims[-2]), int(dims[-1])
        extra_h = -self.ht % divis_by
        extra_w = -self.wd % divis_by
        left = extra_w // 2
        if mode == "sintel":
            top = extra_h // 2
        else:
            top = 0
        self.left, self.right, self.top, self.bottom = left, extra_w - left, top, extra_h - top

    def _shift(self, K, sign):
        K = K.clone()
        K[..., 0, 2] += sign * self.left
        K[..., 1, 2] += sign * self.top
        return K

    def pad(self, *tensors, K=None):
        """Like the reference (core/utils/utils.py:19-28): the bare list of padded tensors when `K` is None
        (`image1, image2 = padder.pad(image1, image2)`, evaluate_stereo.py:239), `(list, shifted K)` otherwise."""
        for t in tensors:
            if t.ndim != 4:
                raise ValueError("InputPadder.pad expects [N,C,H,W] tensors")
        padded = [F.pad(t, [self.left, self.right, self.top, self.bottom], mode="replicate") for t in tensors]
        return padded if K is None else (padded, self._shift(K, +1))

    def unpad(self, x, K=None):
        if x.ndim != 4:
            raise ValueError("InputPadder.unpad expects a [N,C,H,W] tensor")
        h, w = x.shape[-2:]
        y = x[..., self.top:h - self.bottom, self.left:w - self.right]
        return y if K is None else (y, self._shift(K, -1))


@dataclass
class FrameStats:
    epe: float
    d1_weighted: float       # mean(epe>1 over valid) * mask_rate
    d3_weighted: float
    mask_rate: float


@dataclass
class SequenceStats:
    frames: List[FrameStats] = field(default_factory=list)
    domain_flags: int = 0        # tcs_s16_flags after the sequence: bit 0 = an activation was clamped at +-65504, bit 1 = NaN / Inf seen

    def vector(self) -> np.ndarray:
        """[sum_epe, sum_d1w, sum_d3w, sum_rate, n_frames]: what one rank contributes to the gather."""
        if not self.frames:
            return np.zeros(5, np.float64)
        a = np.array([[f.epe, f.d1_weighted, f.d3_weighted, f.mask_rate, 1.0] for f in self.frames], np.float64)
        return a.sum(0)


def frame_metrics(disp_pr: torch.Tensor, disp_gt: torch.Tensor, max_disp: float = 192.0) -> Optional[FrameStats]:
    """evaluate_stereo.py:201-213: per-pixel |pr-gt|, valid = |gt| < 192, outlier rates at 1 and 3 px,
    weighted by the fraction of valid pixels.  Returns None when no pixel is valid (frame skipped)."""
    if disp_pr.shape != disp_gt.shape:
        raise ValueError(f"shape mismatch {tuple(disp_pr.shape)} vs {tuple(disp_gt.shape)}")
    err = (disp_pr - disp_gt).abs().flatten()
    valid = disp_gt.abs().flatten() < max_disp
    if not bool(valid.any()):
        return None
    rate = float(valid.float().mean())
    e = err[valid]
    return FrameStats(float(e.mean()), float((e > 1.0).float().mean()) * rate, float((e > 3.0).float().mean()) * rate, rate)


def reduce_stats(vectors: Seq[np.ndarray]) -> Dict[str, float]:
    """evaluate_stereo.py:214-220: epe = mean over frames; d1 = 100*mean(out*rate)/mean(rate)."""
    tot = np.sum(np.stack(vectors, 0), 0)
    n = max(tot[4], 1.0)
    rate = max(tot[3] / n, 1e-12)
    return {"epe": tot[0] / n, "d1": 100.0 * (tot[1] / n) / rate, "d3": 100.0 * (tot[2] / n) / rate, "frames": int(tot[4])}


@torch.no_grad()
def run_sequence(forward: Callable, seq, iters: int, device, temporal: bool = True, divis_by: int = 32,
                 collect: Optional[list] = None) -> SequenceStats:
    """One video sequence through `forward(image1, image2, iters=, test_mode=True, params=)`.
    State is reset per sequence (evaluate_stereo.py:170-174) and carried frame to frame
    (evaluate_stereo.py:182-197).  `collect`, if given, receives each frame's unpadded prediction."""
    stats = SequenceStats()
    K_raw = torch.as_tensor(seq.K, dtype=torch.float32, device=device)[None]
    baseline = torch.tensor([seq.baseline], dtype=torch.float32, device=device)
    params: dict = {}
    flow_q = fmap1 = prev_T = nets = None
    for fr in seq.frames:
        im1 = torch.as_tensor(fr.image1, device=device)[None]
        im2 = torch.as_tensor(fr.image2, device=device)[None]
        gt = torch.as_tensor(fr.disp_gt, device=device)[None]
        T = torch.as_tensor(fr.T, device=device)[None]
        padder = InputPadder(im1.shape, divis_by=divis_by)
        (im1, im2), K = padder.pad(im1, im2, K=K_raw)
        params.update(K=K, T=T, previous_T=prev_T, last_disp=flow_q, last_net_list=nets, fmap1=fmap1, baseline=baseline)
        out = forward(im1, im2, iters=iters, test_mode=True, params=params if (flow_q is not None and temporal) else None)
        flow_q, nets, fmap1, prev_T = out["flow_q"], out["net_list"], out["fmap1"], T
        disp_pr = padder.unpad(-out["flow"])
        if collect is not None:
            collect.append(disp_pr)
        fs = frame_metrics(disp_pr, gt)
        if fs is not None:
            stats.frames.append(fs)
    if torch.device(device).type == "cuda":
        # the device-side replacement of the reference's per-iteration NaN asserts: one read per sequence
        from . import s16
        stats.domain_flags = s16.take_flags()
        if stats.domain_flags:
            import warnings
            warnings.warn(f"activations left the fp16-split domain during this sequence (flags {stats.domain_flags:#x}: "
                          f"bit 0 = clamped at 65504, bit 1 = NaN/Inf)")
    return stats


# ---------------------------------------------------------------------------------------------------
# real data, when the box has it (BASELINE configs[2]): TartanAir trajectory folder + reference checkpoint
# ---------------------------------------------------------------------------------------------------
def _read_rgb(path: str) -> np.ndarray:
    """PNG -> [3,H,W] float32 0..255 (what evaluate_stereo.py:150-157 hands the model: read_gen + permute + float)."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.asarray(im.convert("RGB"), dtype=np.uint8)
    return np.ascontiguousarray(a.transpose(2, 0, 1)).astype(np.float32)


def load_tartanair_sequence(root: str, max_frames: Optional[int] = None, start: int = 0):
    """One TartanAir trajectory folder (e.g. .../abandonedfactory/Easy/P000) -> `synth.Sequence`.

    Layout and pairing follow the reference's temporal loader (core/stereo_datasets.py:491-495): sorted
    `image_left/*_left.png`, `image_right/*_right.png`, `depth_left/*_left_depth.npy`, and `pose_left.txt` with one
    pose per frame; disparity = 80 / (depth + 1e-5) (frame_utils.py:163-167); intrinsics and baseline are TartanAir's
    constants (evaluate_stereo.py:138-142).  Returns None (with the reason in `load_tartanair_sequence.why`) when the
    folder is absent or incomplete: callers then skip BASELINE configs[2] with that logged reason."""
    import glob
    import os

    from . import formats, synth

    def miss(why):
        load_tartanair_sequence.why = why
        return None

    if not root or not os.path.isdir(root):
        return miss(f"{root!r} is not a directory")
    left = sorted(glob.glob(os.path.join(root, "image_left", "*_left.png")))
    right = sorted(glob.glob(os.path.join(root, "image_right", "*_right.png")))
    depth = sorted(glob.glob(os.path.join(root, "depth_left", "*_left_depth.npy")))
    pose_file = os.path.join(root, "pose_left.txt")
    if not left or len(left) != len(right) or len(left) != len(depth) or not os.path.exists(pose_file):
        return miss(f"{root}: {len(left)} left / {len(right)} right / {len(depth)} depth files, pose_left.txt "
                    f"{'present' if os.path.exists(pose_file) else 'missing'}")
    poses = formats.read_tartanair_extrinsic(pose_file)
    if len(poses) < len(left):
        return miss(f"{root}: {len(poses)} poses for {len(left)} frames")
    stop = len(left) if max_frames is None else min(len(left), start + max_frames)
    frames = []
    for i in range(start, stop):
        disp, _valid = formats.read_disp_tartanair(depth[i])
        frames.append(synth.Frame(_read_rgb(left[i]), _read_rgb(right[i]), np.asarray(disp, np.float32)[None],
                                  np.asarray(poses[i], np.float32)))
    load_tartanair_sequence.why = ""
    return synth.Sequence(frames, synth.TARTANAIR_K.astype(np.float32), synth.TARTANAIR_BASELINE)


load_tartanair_sequence.why = ""


def load_checkpoint(model, path: str) -> int:
    """The reference's `.pth` (evaluate_stereo.py:385-390: `checkpoint['model']`, saved from DataParallel/DDP so keys may
    carry a `module.` prefix) into `model` with strict=True.  Only a weights-only load is attempted: a file that needs
    unpickling of arbitrary objects is refused rather than executed.  Returns the number of tensors loaded."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    state = ckpt["model"] if isinstance(ckpt, dict) and "model" in ckpt else ckpt
    state = {(k[len("module."):] if k.startswith("module.") else k): v for k, v in state.items()}
    model.load_state_dict(state, strict=True)
    return len(state)
