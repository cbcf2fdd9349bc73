#!/usr/bin/env python3
"""Does a kernel's result depend on what runs beside it?  A "victim" launch is repeated on one stream while an "aggressor"
kernel loops on a second stream; every victim output is compared bit-for-bit with the output of a solo run.  The victim
alternates between two inputs, so a stale or foreign value cannot pass.  Any mismatch means a kernel reads or writes memory
(global or LDS) that is not its own at that time.  This is how the 7x7 stems' negative-base LDS reads were found
(tools/scan_ds_negative_base.py; tests/test_gpu_parity.py::test_stem7x7_beside_concurrent_kernels is the regression test).
usage: race_probe.py [trials]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import torch
from tcs_mi355 import ops, s16

dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
trials = int(sys.argv[1]) if len(sys.argv) > 1 else 150
R = lambda *s: torch.randn(*s, generator=gen).to(dev)
H, W = 120, 160


def conv_pc(cout, cin, k, math="f16x3"):
    return ops.pack_conv(R(cout, cin, k, k) * (2.0 / (cin * k * k)) ** 0.5, R(cout) * 0.1, math)


flows = [R(1, 1, H, W) * 20, R(1, 1, H, W) * 20]
corrs = [R(1, 36, H, W), R(1, 36, H, W)]
x64s = [s16.to_s16(R(1, 64, H, W)), s16.to_s16(R(1, 64, H, W))]
f32s = [R(1, 64, H, W), R(1, 64, H, W)]
pc_f1, pc_c1, pc_33, pc_g = conv_pc(64, 1, 7), conv_pc(64, 36, 1), conv_pc(64, 64, 3), conv_pc(32, 2, 3)
g2 = [R(1, 2, H, W), R(1, 2, H, W)]
o = {k: s16.zeros(1, c, H, W, dev) for k, c in (("f1", 64), ("c1", 64), ("33", 64), ("g", 32), ("cv", 64))}
pool_out = s16.zeros(1, 64, 60, 80, dev)
victims = [
    ("k_conv7x7<1> (static LDS)", lambda t: ops.conv2d(pc_f1, [flows[t & 1]], act="relu", out16=o["f1"]).data),
    ("k_conv_f16x3 1x1 fp32 src", lambda t: ops.conv2d(pc_c1, [corrs[t & 1]], act="relu", out16=o["c1"]).data),
    ("k_conv_f16x3_ws 3x3 2->32", lambda t: ops.conv2d(pc_g, [g2[t & 1]], act="relu", out16=o["g"]).data),
    ("k_conv_s16 3x3 64->64", lambda t: s16.conv2d(pc_33, [x64s[t & 1]], act="relu", out16=o["33"])[0].data),
    ("k_avgpool3s2_s16 (no LDS)", lambda t: s16.avgpool3s2(x64s[t & 1], out=pool_out).data),
    ("k_s16_from_f32 (no LDS)", lambda t: s16.to_s16(f32s[t & 1], out=o["cv"]).data),
]
pc_a = conv_pc(256, 128, 3)
xa, oa = s16.to_s16(R(1, 128, H, W)), s16.zeros(1, 256, H, W, dev)
aggressors = [("none", None)] + [(f"conv 128->256 cfg {c}", (lambda c=c: s16.conv2d(pc_a, [xa], out16=oa, tile_cfg=c)))
                                 for c in (101411, 1411, 101412, 101812, 101413)]
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
for vname, vrun in victims:
    ref = [vrun(0).clone(), vrun(1).clone()]
    torch.cuda.synchronize()
    line = f"{vname:28s}"
    for aname, arun in aggressors:
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        for t in range(trials):
            if arun is not None:
                with torch.cuda.stream(sb):
                    for _ in range(3):
                        arun()
            with torch.cuda.stream(sa):
                bad += (vrun(t) != ref[t & 1]).any().long()
        torch.cuda.synchronize()
        line += f" | {aname.replace('conv 128->256 ', '')}: {int(bad)}/{trials}"
    print(line, flush=True)
