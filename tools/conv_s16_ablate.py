#!/usr/bin/env python3
"""Phase ablation of k_conv_s16 (diagnostic build, timing only — results are wrong by construction).
  build (container, no GPU):  python tools/conv_s16_ablate.py --build     -> lib/libtcs_mi355_ablate.so (-DTCS_S16_ABLATE)
  run (GPU box):              python tools/conv_s16_ablate.py [cfg ...]   e.g. 1412 2412 1812
Prints us per launch for: full kernel | no input DMA | no weight DMA | no DMA at all | no operand reads + MFMAs (DMA + barriers only)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import tcs_paths; PKG = tcs_paths.add_product_path()
LIB = os.path.join(PKG, "lib", "libtcs_mi355_ablate.so")
if "--build" in sys.argv:
    from tcs_mi355 import build as b
    b.build(verbose=False)
    obj = os.path.join(b.LIB_DIR, "tcs_conv_s16_ablate.o")
    subprocess.run([b._hipcc(), *b.FLAGS, "-DTCS_S16_ABLATE", "-c", os.path.join(b.CSRC, "tcs_conv_s16.hip"), "-o", obj], check=True)
    objs = [os.path.join(b.LIB_DIR, s.replace(".hip", ".o")) for s in b.SOURCES if s != "tcs_conv_s16.hip"] + [obj]
    subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs], check=True)
    print(LIB)
    sys.exit(0)
os.environ["TCS_MI355_LIB"] = LIB
import torch
from tcs_mi355 import ops, s16
dev = torch.device("cuda:0")
gen = torch.Generator().manual_seed(0)
SH = [("gru08.zr", (128, 128, 128), 256, 3, 120, 160), ("conv128->128", (128,), 128, 3, 120, 160), ("conv64->64", (64,), 64, 3, 120, 160),
      ("gru16.zr", (128, 128, 128), 256, 3, 60, 80)]
cfgs = [int(c) for c in sys.argv[1:] if not c.startswith("-")] or [1412, 2412, 1812, 1413]


def timed(run, n=100, reps=3):
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    g, side = torch.cuda.CUDAGraph(), torch.cuda.Stream()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            for _ in range(n):
                run()
    g.replay(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        g.replay()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e) * 1e3 / (n * reps)


for name, cins, cout, k, H, W in SH:
    xs16 = [s16.to_s16(torch.randn(1, c, H, W, generator=gen).to(dev)) for c in cins]
    w = (torch.randn(cout, sum(cins), k, k, generator=gen) * 0.02).to(dev)
    pc = ops.pack_conv(w, None, "f16x3")
    out = s16.zeros(1, cout, H, W, dev)
    for cfg in cfgs:
        line = f"{name:14s} cfg {cfg:7d}:"
        for tag, abl in (("full", 0), ("no-in-dma", 1), ("no-w-dma", 2), ("no-dma", 3), ("dma-only", 4), ("barriers-only", 7)):
            t = timed(lambda: s16.conv2d(pc, xs16, act="relu", out16=out, tile_cfg=cfg + 1000000 * abl))
            line += f"  {tag} {t:6.1f}"
        print(line, flush=True)
        if "--stamps" in os.environ.get("ABL_FLAGS", ""):
            import ctypes, numpy as np
            from tcs_mi355 import native
            L = native.lib()
            s16.conv2d(pc, xs16, act="relu", out16=out, tile_cfg=cfg)
            torch.cuda.synchronize()
            L.tcs_debug_clear_s16_stamps()
            for rep_ in range(3):
                s16.conv2d(pc, xs16, act="relu", out16=out, tile_cfg=cfg)
                torch.cuda.synchronize()
            nb = 4096
            buf = (ctypes.c_ulonglong * (4 * nb))()
            L.tcs_debug_read_s16_stamps.restype = ctypes.c_int
            assert L.tcs_debug_read_s16_stamps(buf, nb) == 0
            st = np.frombuffer(buf, dtype=np.uint64).reshape(nb, 4).astype(np.int64)
            st = st[st[:, 3] > 0]
            t0 = st[:, 0].min()
            rel = (st - t0) * 0.01
            print(f"    stamps ({len(st)} blocks, us since first block start): start max {rel[:,0].max():.2f} | first-stage-landed avg {rel[:,1].mean():.2f} max {rel[:,1].max():.2f}"
                  f" | loop-done avg {rel[:,2].mean():.2f} max {rel[:,2].max():.2f} | end avg {rel[:,3].mean():.2f} max {rel[:,3].max():.2f}"
                  f" | per-block: fill {(rel[:,1]-rel[:,0]).mean():.2f} loop {(rel[:,2]-rel[:,1]).mean():.2f} epilogue {(rel[:,3]-rel[:,2]).mean():.2f}", flush=True)
