"""Synthetic stereo sequences of the shapes BASELINE.json names.

A scene is a handful of textured planes seen by a rectified pinhole stereo rig that moves along a
smooth trajectory.  Left and right images are ray-cast analytically (so occlusions are right), the
texture is a band-limited sum of sinusoids in plane coordinates, images are uint8-quantised and
returned as float 0..255 like the reference loader does (evaluate_stereo.py:156-157).  Ground-truth
disparity is fx*baseline/Z of the left view and is bounded by `max_disp` ("D" of the metric — the
reference has no D parameter; 192 only appears as the GT validity bound, evaluate_stereo.py:205).
Poses are world->camera 4x4 matrices, the convention of read_tartanair_extrinsic
(core/utils/frame_utils.py:231-259) as consumed by cal_relative_transformation (geo_utils.py:148-155).

Everything is numpy with a Philox generator: the same sequence on every machine.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List

import numpy as np

TARTANAIR_K = np.array([[320.0, 0.0, 320.0], [0.0, 320.0, 240.0], [0.0, 0.0, 1.0]])   # stereo_datasets.py:521-523
TARTANAIR_BASELINE = 0.25                                                               # stereo_datasets.py:524
KITTI_K = np.array([[721.5377, 0.0, 609.5593], [0.0, 721.5377, 172.854], [0.0, 0.0, 1.0]])
KITTI_BASELINE = 0.54                                                                   # stereo_datasets.py:632


@dataclass
class Frame:
    image1: np.ndarray      # [3,H,W] float32, 0..255 (left)
    image2: np.ndarray      # [3,H,W] float32, 0..255 (right)
    disp_gt: np.ndarray     # [1,H,W] float32, positive disparity of the left view
    T: np.ndarray           # [4,4] float32 world->camera


@dataclass
class Sequence:
    frames: List[Frame]
    K: np.ndarray           # [3,3] float32 (full resolution)
    baseline: float


class _Plane:
    def __init__(self, gen, point, normal, n_waves=8, extent=None):
        self.extent = extent                      # half-size in plane coordinates, None = infinite
        self.p0 = np.asarray(point, np.float64)
        n = np.asarray(normal, np.float64)
        self.n = n / np.linalg.norm(n)
        a = np.cross(self.n, [0.0, 1.0, 0.0])
        if np.linalg.norm(a) < 1e-3:
            a = np.cross(self.n, [1.0, 0.0, 0.0])
        self.u = a / np.linalg.norm(a)
        self.v = np.cross(self.n, self.u)
        # band-limited texture: random directions, wavelengths 0.1 .. 1.6 m, per colour channel
        self.freq = gen.uniform(0.6, 10.0, size=(n_waves, 1)) * _unit2(gen, n_waves)
        self.phase = gen.uniform(0, 2 * np.pi, size=(3, n_waves))
        self.amp = gen.uniform(0.4, 1.0, size=(3, n_waves))
        self.amp /= self.amp.sum(1, keepdims=True)
        self.base = gen.uniform(90, 165, size=3)

    def intersect(self, origin, dirs):
        """Ray parameter t (>0 in front) for rays origin + t*dirs, dirs [...,3]."""
        denom = dirs @ self.n
        num = (self.p0 - origin) @ self.n
        with np.errstate(divide="ignore", invalid="ignore"):
            t = num / denom
        t = np.where(np.abs(denom) < 1e-9, np.inf, t)
        if self.extent is not None:
            ok = np.isfinite(t) & (t > 0)
            rel = origin + dirs * np.where(ok, t, 0.0)[..., None] - self.p0
            inside = (np.abs(rel @ self.u) < self.extent) & (np.abs(rel @ self.v) < self.extent)
            t = np.where(ok & inside, t, np.inf)
        return t

    def colour(self, pts):
        rel = pts - self.p0
        uv = np.stack([rel @ self.u, rel @ self.v], -1)                  # [...,2]
        arg = uv @ self.freq.T * (2 * np.pi)                            # [...,n_waves]
        tex = np.sin(arg[..., None, :] + self.phase) * self.amp         # [...,3,n_waves]
        return self.base + 110.0 * tex.sum(-1)


def _unit2(gen, n):
    a = gen.uniform(0, 2 * np.pi, size=n)
    return np.stack([np.cos(a), np.sin(a)], 1)


def _pose(t_idx, step, yaw_deg):
    """Camera-to-world: forward motion along +z with a slow yaw; returned as world->camera."""
    yaw = np.deg2rad(yaw_deg) * t_idx
    c, s = np.cos(yaw), np.sin(yaw)
    R_c2w = np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])
    pos = np.array([0.02 * t_idx, 0.0, step * t_idx])
    T_c2w = np.eye(4)
    T_c2w[:3, :3] = R_c2w
    T_c2w[:3, 3] = pos
    return np.linalg.inv(T_c2w)


def _render(planes, T_w2c, K, h, w, cam_offset_x):
    """Ray-cast one view.  cam_offset_x shifts the camera centre along its own +x (right camera)."""
    T_c2w = np.linalg.inv(T_w2c)
    R, pos = T_c2w[:3, :3], T_c2w[:3, 3]
    origin = pos + R @ np.array([cam_offset_x, 0.0, 0.0])
    ys, xs = np.meshgrid(np.arange(h, dtype=np.float64), np.arange(w, dtype=np.float64), indexing="ij")
    dirs_c = np.stack([(xs - K[0, 2]) / K[0, 0], (ys - K[1, 2]) / K[1, 1], np.ones_like(xs)], -1)
    dirs_w = dirs_c @ R.T
    best_t = np.full((h, w), np.inf)
    img = np.zeros((h, w, 3))
    for pl in planes:
        t = pl.intersect(origin, dirs_w)
        hit = (t > 1e-3) & (t < best_t)
        if not hit.any():
            continue
        pts = origin + dirs_w[hit] * t[hit][:, None]
        img[hit] = pl.colour(pts)
        best_t = np.where(hit, t, best_t)
    depth = best_t            # dirs_c has z = 1, so t IS the camera-frame depth
    return img, depth


def make_sequence(seed: int, n_frames: int = 10, height: int = 480, width: int = 640, max_disp: float = 192.0,
                  K: np.ndarray = TARTANAIR_K, baseline: float = TARTANAIR_BASELINE, step: float = 0.05,
                  yaw_deg: float = 0.5) -> Sequence:
    """Config 2 of BASELINE.json by default: 640x480, D=192, 10 frames, TartanAir intrinsics.
    `K` is scaled with the image size when height/width differ from 480x640 so the field of view is kept."""
    gen = np.random.Generator(np.random.Philox(key=int(seed)))
    K = np.array(K, np.float64)
    if (height, width) != (480, 640) and K is not KITTI_K and np.allclose(K, TARTANAIR_K):
        sx, sy = width / 640.0, height / 480.0
        K = K * np.array([[sx], [sy], [1.0]])
    fx = K[0, 0]
    z_min = fx * baseline / max_disp * 1.15        # keeps disparity < max_disp over the whole clip
    travel = step * (n_frames + 1)
    # a back wall, a slanted floor-ish plane, and 2-3 slanted foreground panels
    planes = [_Plane(gen, [0, 0, z_min * gen.uniform(9, 14) + travel], [gen.uniform(-.15, .15), gen.uniform(-.1, .1), -1])]
    # floor: low enough that its nearest visible point (bottom image row) stays beyond z_min
    y_floor = z_min * (height / 2.0) / K[1, 1] * gen.uniform(1.15, 1.6)
    planes.append(_Plane(gen, [0, y_floor, 0], [gen.uniform(-.05, .05), -1, gen.uniform(-.12, -.03)]))
    for _ in range(int(gen.integers(2, 4))):
        z = z_min * gen.uniform(1.6, 5.0) + travel
        planes.append(_Plane(gen, [gen.uniform(-1.2, 1.2) * z * 0.5, gen.uniform(-0.3, 0.2) * z * 0.5, z],
                             [gen.uniform(-.6, .6), gen.uniform(-.3, .3), -1], extent=z * gen.uniform(0.18, 0.4)))
    frames = []
    for t in range(n_frames):
        T = _pose(t, step, yaw_deg)
        left, depth = _render(planes, T, K, height, width, 0.0)
        right, _ = _render(planes, T, K, height, width, baseline)
        depth = np.maximum(depth, z_min)
        disp = fx * baseline / depth
        q = lambda im: np.clip(np.rint(im), 0, 255).astype(np.float32).transpose(2, 0, 1)
        frames.append(Frame(q(left), q(right), disp[None].astype(np.float32), T.astype(np.float32)))
    return Sequence(frames, K.astype(np.float32), float(baseline))


def make_pair(seed: int, height: int = 240, width: int = 320, max_disp: float = 64.0) -> Frame:
    """Config 1 of BASELINE.json: a single 320x240 pair, D=64."""
    return make_sequence(seed, n_frames=1, height=height, width=width, max_disp=max_disp).frames[0]
