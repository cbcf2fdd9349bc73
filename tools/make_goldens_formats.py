#!/usr/bin/env python3
"""Generate tests/golden/formats.npz: the REFERENCE's on-disk readers (core/utils/frame_utils.py) run on constructed files.

Runs only in the build container (needs /root/reference).  frame_utils.py imports `imageio` and `cv2` at the top of the
file and calls `cv2.setNumThreads` / `cv2.ocl.setUseOpenCL` (frame_utils.py:6-11); neither package is installed here, and
none of the readers pinned below uses them (they are pure numpy / scipy / PIL), so two empty stub modules satisfy the
import — the same accommodation tools/make_goldens.py makes for `cupy`.  Nothing is fetched.

The fixture stores DATA only: the bytes of each constructed input file and what the reference's reader returned for it
  read_tartanair_extrinsic (frame_utils.py:231-259)   pose_left.txt, 7 numbers per line
  readDispTartanAir        (frame_utils.py:163-167)   depth .npy
  readPFM                  (frame_utils.py:44-79)     grey little-endian, grey big-endian, colour
  read_kitti_extrinsic     (frame_utils.py:274-284)   12 numbers per line
  readsceneflow_pose       (frame_utils.py:262-271)   camera_data.txt with Frame / L / R lines
  read_gen + np.array      (frame_utils.py:214-228, evaluate_stereo.py:150-157)  RGB and RGBA-free PNG
tests/test_host.py re-creates the files from the stored bytes and checks tcs_mi355/formats.py (and the harness' PNG reader)
against the stored outputs.
"""
import io
import os
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden", "formats.npz")


def import_frame_utils():
    cv2 = types.ModuleType("cv2")
    cv2.setNumThreads = lambda n: None
    cv2.ocl = types.SimpleNamespace(setUseOpenCL=lambda b: None)
    sys.modules["cv2"] = cv2
    sys.modules["imageio"] = types.ModuleType("imageio")
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_frame_utils", "/root/reference/core/utils/frame_utils.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def as_bytes(b: bytes) -> np.ndarray:
    return np.frombuffer(b, dtype=np.uint8).copy()


def main():
    fu = import_frame_utils()
    rng = np.random.Generator(np.random.Philox(key=20240611))
    out = {}
    with tempfile.TemporaryDirectory() as d:
        # ---- TartanAir poses: translation + (unnormalised on purpose) quaternion, plain "%f"-style text ----------------
        n = 6
        t = rng.normal(0, 5, (n, 3))
        q = rng.normal(0, 1, (n, 4))
        q[:3] /= np.linalg.norm(q[:3], axis=1, keepdims=True)          # half unit quaternions, half not (scipy normalises)
        txt = "".join(" ".join(f"{v:.6f}" for v in (*t[i], *q[i])) + "\n" for i in range(n))
        p = os.path.join(d, "pose_left.txt")
        open(p, "w").write(txt)
        out["tartanair_pose_txt"] = as_bytes(txt.encode())
        out["tartanair_pose_out"] = np.stack(fu.read_tartanair_extrinsic(p, "left")).astype(np.float64)

        # ---- TartanAir depth -> disparity -------------------------------------------------------------------------------
        depth = rng.uniform(0.4, 200.0, (12, 20)).astype(np.float32)
        depth[0, 0] = 0.0                                                # a zero depth: 80 / 1e-5
        p = os.path.join(d, "000000_left_depth.npy")
        np.save(p, depth)
        out["tartanair_depth_npy"] = as_bytes(open(p, "rb").read())
        disp, valid = fu.readDispTartanAir(p)
        out["tartanair_disp_out"] = np.asarray(disp)
        out["tartanair_valid_out"] = np.asarray(valid)

        # ---- PFM: grey little-endian, grey big-endian, colour ------------------------------------------------------------
        img = rng.normal(0, 30, (5, 7)).astype(np.float32)
        col = rng.normal(0, 30, (4, 6, 3)).astype(np.float32)
        for name, arr, head, dt in (("pfm_le", img, b"Pf\n7 5\n-1.0\n", "<f4"), ("pfm_be", img, b"Pf\n7 5\n1.0\n", ">f4"),
                                    ("pfm_color", col, b"PF\n6 4\n-1\n", "<f4")):
            raw = head + np.flipud(arr).astype(dt).tobytes()
            p = os.path.join(d, name + ".pfm")
            open(p, "wb").write(raw)
            out[name + "_file"] = as_bytes(raw)
            out[name + "_out"] = np.ascontiguousarray(fu.readPFM(p)).astype(np.float32)
            out[name + "_gen_out"] = np.ascontiguousarray(fu.read_gen(p)).astype(np.float32)    # colour: last channel dropped

        # ---- KITTI-style poses ----------------------------------------------------------------------------------------------
        def rigid():
            A = rng.normal(0, 1, (3, 3))
            Q, _ = np.linalg.qr(A)
            if np.linalg.det(Q) < 0:
                Q[:, 0] = -Q[:, 0]
            return Q, rng.normal(0, 10, 3)

        lines = []
        for _ in range(5):
            R, tt = rigid()
            lines.append(" ".join(f"{v:.9e}" for v in np.hstack([R, tt[:, None]]).reshape(-1)))
        txt = "\n".join(lines) + "\n"
        p = os.path.join(d, "poses.txt")
        open(p, "w").write(txt)
        out["kitti_pose_txt"] = as_bytes(txt.encode())
        out["kitti_pose_out"] = np.stack(fu.read_kitti_extrinsic(p)).astype(np.float64)

        # ---- SceneFlow camera_data.txt -------------------------------------------------------------------------------------
        lines = []
        for f in range(4):
            lines.append(f"Frame {f + 1}")
            for side in "LR":
                R, tt = rigid()
                M = np.eye(4)
                M[:3, :3], M[:3, 3] = R, tt
                lines.append(side + " " + " ".join(f"{v:.8f}" for v in M.reshape(-1)))
            lines.append("")
        txt = "\n".join(lines) + "\n"
        p = os.path.join(d, "camera_data.txt")
        open(p, "w").write(txt)
        out["sceneflow_pose_txt"] = as_bytes(txt.encode())
        out["sceneflow_pose_out"] = np.stack(fu.readsceneflow_pose(p)).astype(np.float64)

        # ---- images: what evaluate_stereo.py:150-157 hands the model (read_gen -> np.array -> permute -> float) -----------
        from PIL import Image
        rgb = rng.integers(0, 256, (9, 11, 3), dtype=np.uint8)
        buf = io.BytesIO()
        Image.fromarray(rgb).save(buf, format="PNG")
        p = os.path.join(d, "000000_left.png")
        open(p, "wb").write(buf.getvalue())
        out["png_file"] = as_bytes(buf.getvalue())
        out["png_out"] = np.ascontiguousarray(np.array(fu.read_gen(p)).transpose(2, 0, 1)).astype(np.float32)
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT}: {len(out)} arrays, {os.path.getsize(OUT)} bytes")
    for k, v in out.items():
        print(f"  {k}: {v.dtype} {v.shape}")


if __name__ == "__main__":
    main()
