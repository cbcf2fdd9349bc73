#!/usr/bin/env python3
"""Reference-generated vectors for architecture switches beyond the shipped evaluation setting (run in the build
container only, like tools/make_goldens.py whose reference import it reuses): the slow-fast GRU schedule
(tc_stereo.py:182-187).  First frame of a small synthetic pair, 4 iterations.  Writes tests/golden/e2e_configs.npz.
(n_gru_layers < 3 is not a usable switch of this model: DispGradPredictor indexes three context levels, update.py:206-210.)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np, torch
import make_goldens as mg

def main():
    torch.set_num_threads(8)
    ref_tc, *_ = mg.import_reference()
    weights = mg.load_by_path("tcs_weights", os.path.join(mg.PKG, "tcs_mi355", "weights.py"))
    synth = mg.load_by_path("tcs_synth", os.path.join(mg.PKG, "tcs_mi355", "synth.py"))
    from argparse import Namespace
    out = {}
    for tag, kw in (("slow_fast", dict(slow_fast_gru=True)),):
        d = dict(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2, context_norm="none",
                 slow_fast_gru=False, n_gru_layers=3, mixed_precision=False, init_thres=0.5)
        d.update(kw)
        model = ref_tc.TCStereo(Namespace(**d)).eval()
        W = weights.synth_state_dict({k: list(v.shape) for k, v in model.state_dict().items()})
        model.load_state_dict(W, strict=True)
        fr = synth.make_sequence(23, n_frames=1, height=96, width=128, max_disp=32.0).frames[0]
        with torch.no_grad():
            o = model(mg.T(fr.image1)[None], mg.T(fr.image2)[None], iters=4, test_mode=True)
        out[f"{tag}_input_sha"] = np.frombuffer(mg.sha(fr.image1, fr.image2).encode(), dtype=np.uint8)
        out[f"{tag}_flow"], out[f"{tag}_flow_q"] = mg.N(o["flow"]), mg.N(o["flow_q"])
        print(tag, "|flow| mean", float(o["flow"].abs().mean()))
    np.savez_compressed(os.path.join(mg.OUT, "e2e_configs.npz"), **out)

if __name__ == "__main__":
    main()
