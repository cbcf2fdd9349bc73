"""On-disk formats either side of the hot path: what the reference's
core/utils/frame_utils.py reads for the evaluation configs, restated with numpy only (no cv2,
imageio or scipy dependency) so that BASELINE configs 3 (TartanAir) and 5 (KITTI raw) can run when the
data is present on the box.  Every pose reader returns WORLD->CAMERA 4x4 matrices, the convention
`TCStereo.forward` expects in params['T'] (geo_utils.py:148-155).

Pinned to the reference: tools/make_goldens_formats.py runs the reference's own readers on constructed files and
tests/test_host.py::test_formats_match_reference_fixtures checks these functions against what they returned.
"""
from __future__ import annotations

import re
from typing import List

import numpy as np

TARTANAIR_FX_TIMES_BASELINE = 80.0       # fx 320 px * baseline 0.25 m (frame_utils.py:163-167)


def disp_from_tartanair_depth(depth: np.ndarray):
    """TartanAir depth (.npy, metres) -> (disparity, valid) (frame_utils.py:163-167)."""
    disp = TARTANAIR_FX_TIMES_BASELINE / (np.asarray(depth) + 1e-5)
    return disp, disp > 0


def read_disp_tartanair(path: str):
    return disp_from_tartanair_depth(np.load(path, allow_pickle=False))


def quat_to_matrix(qx, qy, qz, qw) -> np.ndarray:
    """Unit-normalised quaternion (x,y,z,w) -> rotation matrix (what scipy's Rotation.from_quat gives)."""
    q = np.array([qx, qy, qz, qw], np.float64)
    x, y, z, w = q / np.linalg.norm(q)
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


_NED_TO_CAMERA = np.array([[0, 1, 0, 0], [0, 0, 1, 0], [1, 0, 0, 0], [0, 0, 0, 1]], np.float64)


def tartanair_pose_to_world2cam(tx, ty, tz, qx, qy, qz, qw) -> np.ndarray:
    """One line of a TartanAir pose file (camera->world in NED axes: translation + quaternion) ->
    world->camera with z forward (frame_utils.py:231-259): invert the rigid motion, then permute axes."""
    R = quat_to_matrix(qx, qy, qz, qw)
    T = np.eye(4)
    T[:3, :3] = R.T
    T[:3, 3] = -R.T @ np.array([tx, ty, tz], np.float64)
    return _NED_TO_CAMERA @ T


def read_tartanair_extrinsic(path: str) -> List[np.ndarray]:
    out = []
    with open(path) as f:
        for line in f:
            vals = line.split()
            if not vals:
                continue
            if len(vals) != 7:
                raise ValueError(f"TartanAir pose lines have 7 numbers (t, quaternion), got {len(vals)}")
            out.append(tartanair_pose_to_world2cam(*map(float, vals)))
    return out


def read_kitti_extrinsic(path: str) -> List[np.ndarray]:
    """KITTI-style pose text: 12 numbers per line = 3x4 camera->world; returned inverted (frame_utils.py:274-284)."""
    out = []
    with open(path) as f:
        for line in f:
            vals = line.split()
            if not vals:
                continue
            if len(vals) != 12:
                raise ValueError(f"KITTI pose lines have 12 numbers, got {len(vals)}")
            P = np.vstack([np.array(vals, np.float64).reshape(3, 4), [0, 0, 0, 1]])
            out.append(np.linalg.inv(P))
    return out


def read_sceneflow_pose(path: str) -> List[np.ndarray]:
    """SceneFlow camera_data.txt: lines 'L m00 ... m33' hold the LEFT camera->world matrix; returned
    inverted (frame_utils.py:262-271).  'R' and 'Frame' lines are ignored."""
    out = []
    with open(path) as f:
        for line in f:
            tok = line.split()
            if tok and tok[0] == "L":
                out.append(np.linalg.inv(np.array(tok[1:], np.float64).reshape(4, 4)))
    return out


def read_pfm(path: str) -> np.ndarray:
    """PFM image (SceneFlow disparities): header 'PF'/'Pf', 'W H', scale (negative = little endian),
    rows stored bottom-up (frame_utils.py:44-79)."""
    with open(path, "rb") as f:
        kind = f.readline().rstrip()
        if kind not in (b"PF", b"Pf"):
            raise ValueError("not a PFM file")
        m = re.match(rb"^(\d+)\s(\d+)\s$", f.readline())
        if not m:
            raise ValueError("malformed PFM header")
        w, h = int(m.group(1)), int(m.group(2))
        scale = float(f.readline().rstrip())
        data = np.frombuffer(f.read(), dtype=("<f4" if scale < 0 else ">f4"))
    shape = (h, w, 3) if kind == b"PF" else (h, w)
    return np.flipud(data[: int(np.prod(shape))].reshape(shape)).astype(np.float32)


def read_gen_pfm(path: str) -> np.ndarray:
    """What the reference's `read_gen` returns for a .pfm (frame_utils.py:223-228): float32, and for a colour file the last
    channel dropped."""
    a = read_pfm(path)
    return a if a.ndim == 2 else a[:, :, :-1]


def write_pfm(path: str, img: np.ndarray) -> None:
    img = np.asarray(img, np.float32)
    with open(path, "wb") as f:
        f.write(b"PF\n" if img.ndim == 3 else b"Pf\n")
        f.write(f"{img.shape[1]} {img.shape[0]}\n".encode())
        f.write(b"-1.0\n")
        f.write(np.flipud(img).astype("<f4").tobytes())
