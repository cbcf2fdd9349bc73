#!/usr/bin/env python3
"""Diagnostic (needs lib/libtcs_mi355_stamps.so, built with -DTCS_CONV_STAMPS): per-phase cycle shares of the
fp16-split conv K loop.  Run with TCS_MI355_LIB=.../libtcs_mi355_stamps.so."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; tcs_paths.add_product_path()
import numpy as np, torch
from tcs_mi355 import ops, native
dev = torch.device("cuda:0")
L = native.lib()
L.tcs_debug_read_conv_stamps.restype = ctypes.c_int
L.tcs_debug_read_conv_stamps.argtypes = [ctypes.c_void_p, ctypes.c_int]
gen = torch.Generator().manual_seed(0)
for name, cins, cout, H, W in (("conv128->128", (128,), 128, 120, 160), ("gru08.q-like 384->128", (128, 128, 128), 128, 120, 160),
                               ("one patch row 128->128", (128,), 128, 4, 160)):
    cin = sum(cins)
    w = (torch.randn(cout, cin, 3, 3, generator=gen) * 0.02).to(dev)
    xs = [torch.randn(1, c, H, W, generator=gen).to(dev) for c in cins]
    pc = ops.pack_conv(w, torch.zeros(cout, device=dev), "f16x3")
    for _ in range(3):
        ops.conv2d(pc, xs, act="relu")
    torch.cuda.synchronize()
    nblocks = ((H + 3) // 4) * ((W + 31) // 32) * ((cout + 31) // 32)
    nw = min(nblocks * 4, 16384)
    buf = np.zeros(nw * 8, np.uint64)
    assert L.tcs_debug_read_conv_stamps(buf.ctypes.data, nw) == 0
    b = buf.reshape(nw, 8).astype(np.float64)
    n = b[:, 5].mean()
    per = b[:, :5].mean(0) / max(n - 1, 1)
    print(f"{name}: blocks {nblocks}, chunks {n:.0f}; cycles per chunk: load-issue {per[0]:.0f}, mfma-phase {per[1]:.0f}, "
          f"barrier1 {per[2]:.0f}, convert+store {per[3]:.0f}, barrier2 {per[4]:.0f}, total {per.sum():.0f}", flush=True)
