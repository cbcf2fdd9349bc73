#!/usr/bin/env python3
"""(Uses the CPU oracle, hence kept under tests/.)  GPU-side debugging aid: stage-by-stage HIP vs oracle differences on frames 0 and 1 of the bench clip."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import tcs_paths; tcs_paths.add_product_path()
import numpy as np, torch
import tcs_oracle as oracle
from bench import build_model
from tcs_mi355 import synth
from tcs_mi355.harness import InputPadder

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 6
H, W = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (480, 640)
dev = torch.device("cuda:0")
model, Wt = build_model(dev)
model.use_hip_graph = False
torch.set_num_threads(16)
seq = synth.make_sequence(2000, n_frames=2, height=H, width=W, max_disp=192.0 * W / 640)
K = torch.as_tensor(seq.K)[None]; base = torch.tensor([seq.baseline])
def d(a, b):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    return f"max {float((a-b).abs().max()):.3e} mean {float((a-b).abs().mean()):.3e} (|ref| mean {float(b.abs().mean()):.3e})"
state_g = state_o = None
for t, fr in enumerate(seq.frames):
    i1, i2, T = torch.as_tensor(fr.image1)[None], torch.as_tensor(fr.image2)[None], torch.as_tensor(fr.T)[None]
    pg = po = None
    if t > 0:
        fq, nets, fm, pT = state_g
        pg = dict(K=K.to(dev), T=T.to(dev), previous_T=pT.to(dev), last_disp=fq, last_net_list=nets, fmap1=fm, baseline=base.to(dev))
        # feed the ORACLE the GPU's previous state so that only this frame's arithmetic is compared
        po = dict(K=K, T=T, previous_T=pT, last_disp=fq.cpu(), last_net_list=[n.cpu() for n in nets], fmap1=fm.cpu(), baseline=base)
    tg, to = {}, {}
    model._trace = tg
    og = model(i1.to(dev), i2.to(dev), iters=iters, test_mode=True, params=pg)
    torch.cuda.synchronize()
    t0 = time.time(); oo = oracle.tc_stereo_forward(Wt, i1, i2, iters=iters, params=po, trace=to); print(f"frame {t}: oracle {time.time()-t0:.1f}s", flush=True)
    for k in ("sparse_disp", "cost", "sparse_mask", "disp_init"):
        print(f"  {k:12s} {d(tg[k], to[k])}")
    if t > 0:
        m = (tg['sparse_mask'].cpu() != to['sparse_mask']).sum().item(); print("  mask flips:", m)
    for s in range(3):
        print(f"  net0[{s}]      {d(tg['net0'][s], to['net0'][s])}")
    for it in range(iters):
        a, b = tg["iters"][it], to["iters"][it]
        print(f"  it{it:02d} corr {d(a['corr'], b['corr'])} | delta {d(a['delta'], b['delta'])} | refined {d(a['refined'], b['refined'])}")
    print(f"  flow   {d(og['flow'], oo['flow'])}")
    state_g = (og["flow_q"], og["net_list"], og["fmap1"], T)
