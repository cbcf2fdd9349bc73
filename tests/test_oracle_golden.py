"""CPU: the oracle (oracle/tcs_oracle.py) against vectors produced by the reference itself
(tools/make_goldens.py).  Tolerances per SURVEY.md §8c."""
import hashlib

import numpy as np
import pytest
import torch

from conftest import T, epe, maxdiff


def test_corr_build_pyramid_cost(oracle, ops_golden):
    g = ops_golden
    for tag in "ab":
        vol = oracle.corr_volume(T(g[f"corr{tag}_f1"]), T(g[f"corr{tag}_f2"]))
        pyr = oracle.corr_pyramid(vol, 4)
        for i in range(4):
            assert pyr[i].shape == g[f"corr{tag}_pyr{i}"].shape
            assert maxdiff(pyr[i], g[f"corr{tag}_pyr{i}"]) <= 1e-5
        assert maxdiff(oracle.masked_cost_volume(vol), g[f"corr{tag}_cost"]) <= 1e-5


def test_corr_lookup(oracle, ops_golden):
    g = ops_golden
    for tag in "ab":
        pyr = [T(g[f"corr{tag}_pyr{i}"]) for i in range(4)]
        out = oracle.corr_lookup(pyr, T(g[f"corr{tag}_coords"]), 4)
        assert out.shape == g[f"corr{tag}_lookup"].shape
        assert maxdiff(out, g[f"corr{tag}_lookup"]) <= 1e-5


def test_argmax(oracle, ops_golden):
    g = ops_golden
    for tag in "ab":
        d, c, m = oracle.argmax_disp(T(g[f"corr{tag}_cost"]))
        assert maxdiff(d, g[f"corr{tag}_sparse_disp"]) == 0
        assert maxdiff(m, g[f"corr{tag}_sparse_mask"]) == 0
        assert maxdiff(c, g[f"corr{tag}_sparse_cost"]) == 0


def test_geometry(oracle, ops_golden):
    g = ops_golden
    d, K, Ki, Tr, b = (T(g[k]) for k in ("geo_disp", "geo_K", "geo_Kinv", "geo_Trel", "geo_baseline"))
    assert maxdiff(oracle.backward_grid(d, Tr, K, Ki, b), g["geo_backward_grid"]) <= 2e-4
    assert maxdiff(oracle.disp_gradient_xy(d), g["geo_grad_xy"]) == 0
    ref = g["geo_grad_cands"]
    got = oracle.grad_candidates(d).numpy()
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all()
    assert np.abs(got[fin] - ref[fin]).max() <= 1e-5 * max(1.0, np.abs(ref[fin]).max())
    cd, valid, flow, metric = oracle.forward_warp_inputs(d, Tr, K, Ki, b)
    assert maxdiff(cd, g["geo_warp_disp"]) <= 1e-4
    assert maxdiff(valid, g["geo_warp_valid"]) == 0
    assert maxdiff(flow, g["geo_warp_flow"]) <= 2e-4
    assert maxdiff(metric, g["geo_warp_metric"]) <= 1e-4
    assert maxdiff(oracle.sample_bilinear(T(g["samp_img"]), T(g["samp_grid"])), g["samp_out"]) <= 1e-5


def test_forward_warp_restatement_pinned(oracle, ops_golden):
    """warp() through the stand-in splat: checks the wrapper maths around the splat
    (softsplat.py:232-274); the splat kernel itself is restatement-pinned."""
    g = ops_golden
    d, K, Ki, Tr, b = (T(g[k]) for k in ("geo_disp", "geo_K", "geo_Kinv", "geo_Trel", "geo_baseline"))
    wd, wf, wm = oracle.forward_warp(d, T(g["rp_warp_fmap_in"]), Tr, K, Ki, b)
    assert maxdiff(wm, g["rp_warp_mask"]) == 0
    assert maxdiff(wd, g["rp_warp_disp"]) <= 1e-4
    assert maxdiff(wf, g["rp_warp_fmap"]) <= 1e-4


def test_softsplat_properties(oracle):
    """Known-answer checks for the splat restated from softsplat.py:285-335."""
    x = torch.zeros(1, 2, 4, 5)
    x[0, :, 1, 1] = torch.tensor([2.0, 3.0])
    flow = torch.zeros(1, 2, 4, 5)
    out = oracle.softsplat_forward(x, flow)                 # zero flow: identity
    assert torch.equal(out, x)
    flow[0, 0, 1, 1], flow[0, 1, 1, 1] = 0.25, 0.5         # lands between 4 pixels
    out = oracle.softsplat_forward(x, flow)
    assert out[0, 0].sum().item() == pytest.approx(2.0)
    assert out[0, 0, 1, 1].item() == pytest.approx(2.0 * 0.75 * 0.5)
    assert out[0, 0, 2, 2].item() == pytest.approx(2.0 * 0.25 * 0.5)
    flow[0, 0, 1, 1] = 100.0                                # leaves the frame: dropped
    assert oracle.softsplat_forward(x, flow).abs().sum().item() == 0
    flow[0, 0, 1, 1] = float("nan")                         # non-finite target: skipped
    assert oracle.softsplat_forward(x, flow).abs().sum().item() == 0
    flow[0, 0, 1, 1] = -1.5                                 # target x=-0.5,y=1.5: only the two in-frame corners kept
    out = oracle.softsplat_forward(x, flow)
    assert out[0, 0, 1, 0].item() == pytest.approx(2.0 * 0.5 * 0.5)
    assert out[0, 0, 2, 0].item() == pytest.approx(2.0 * 0.5 * 0.5)
    assert out[0, 0].sum().item() == pytest.approx(1.0)


def test_stencils(oracle, ops_golden):
    g = ops_golden
    cand, mat = oracle.propagate_disparity(T(g["prop_grad"]), T(g["prop_disp"]))
    assert maxdiff(cand, g["prop_cand"]) <= 1e-6
    assert maxdiff(mat, g["prop_matrix"]) == 0
    assert maxdiff(oracle.convex_upsample(T(g["ups_flow"]), T(g["ups_mask"]), 4), g["ups_out"]) <= 2e-6


def test_cells_and_blocks(oracle, ops_golden, synth_weights):
    g, W = ops_golden, synth_weights
    net = [T(g["ub_h08"]), T(g["ub_h16"]), T(g["ub_h32"])]
    inp = [[T(g[f"ub_ctx{i}{n}"]) for n in "zrq"] for i in range(3)]
    out, delta = oracle.update_block(W, net, inp, T(g["ub_corr"]), T(g["ub_flow"]))
    assert maxdiff(delta, g["ub_delta"]) <= 1e-4
    for o, k in zip(out, ("ub_out08", "ub_out16", "ub_out32")):
        assert maxdiff(o, g[k]) <= 1e-4
    assert maxdiff(oracle.motion_encoder(W, T(g["ub_flow"]), T(g["ub_corr"])), g["enc_out"]) <= 1e-4
    assert maxdiff(oracle.gru_1x1(W, "previous_current_hideen_fuse.0", T(g["ub_h08"]), T(g["lf_x"])), g["lf_out"]) <= 1e-5
    assert maxdiff(oracle.hidden_state_update(W, T(g["ub_h08"]), T(g["hu_delta"])), g["hu_out"]) <= 1e-5
    grad, ctx = oracle.disp_grad_predictor(W, T(g["prop_grad"]), T(g["prop_disp"]), [T(g[f"dg_ctx{i}"]) for i in range(3)])
    assert maxdiff(grad, g["dg_grad"]) <= 1e-4
    assert maxdiff(ctx, g["dg_context"]) <= 1e-4
    ref, mask = oracle.disp_refine(W, T(g["prop_grad"]), T(g["prop_disp"]), T(g["ub_h08"]), T(g["dg_context"]), want_mask=True)
    assert maxdiff(ref, g["dr_refined"]) <= 1e-4
    assert maxdiff(mask, g["dr_mask"]) <= 1e-4
    comp, mono, w, nets = oracle.disparity_completor(W, T(g["dc_disp"]), T(g["dc_cost"]), T(g["dc_mask"]),
                                                     [T(g[f"dc_net{i}"]) for i in range(3)])
    assert maxdiff(comp, g["dc_completed"]) <= 1e-4
    assert maxdiff(mono, g["dc_mono"]) <= 1e-4
    assert maxdiff(w, g["dc_w"]) <= 1e-5
    for i in range(3):
        assert maxdiff(nets[i], g[f"dc_out{i}"]) <= 2e-4


def test_extractor(oracle, ops_golden, synth_weights):
    g, W = ops_golden, synth_weights
    x = 2 * (T(g["ext_img"]) / 255.0) - 1.0
    cl, trunk = oracle.context_encoder(W, x, "none")
    assert maxdiff(trunk, g["ext_trunk"]) <= 1e-4 * max(1.0, np.abs(g["ext_trunk"]).max())
    for i in range(3):
        assert maxdiff(cl[i][0], g[f"ext_net{i}"]) <= 1e-4 * max(1.0, np.abs(g[f"ext_net{i}"]).max())
        assert maxdiff(cl[i][1], g[f"ext_ctx{i}"]) <= 1e-4 * max(1.0, np.abs(g[f"ext_ctx{i}"]).max())


def _sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def test_e2e_c1_first_frame(oracle, e2e_golden, synth_weights):
    """Config 1: 320x240 pair, D=64, 8 iters, padded to 256 rows.  Tolerance 1e-4 EPE."""
    from tcs_mi355 import synth
    from tcs_mi355.harness import InputPadder
    pr = synth.make_pair(1)
    assert _sha(pr.image1, pr.image2) == bytes(e2e_golden["c1_input_sha"]).decode()
    i1, i2 = T(pr.image1)[None], T(pr.image2)[None]
    padder = InputPadder(i1.shape, divis_by=32)
    p1, p2 = padder.pad(i1, i2)
    out = oracle.tc_stereo_forward(synth_weights, p1, p2, iters=8)
    assert epe(out["flow_q"], e2e_golden["c1_flow_q"]) <= 1e-4
    assert epe(out["flow"], e2e_golden["c1_flow"]) <= 1e-4


def test_e2e_temporal_clip(oracle, e2e_golden, synth_weights):
    """3-frame 160x128 clip, 6 iters.  Frame 0 is reference-pinned; frames 1-2 went through the
    stand-in splat in the reference run (restatement-pinned for the splat, reference code for the rest)."""
    from tcs_mi355 import synth
    seq = synth.make_sequence(7, n_frames=3, height=128, width=160, max_disp=48.0)
    assert _sha(*[f.image1 for f in seq.frames], *[f.image2 for f in seq.frames]) == bytes(e2e_golden["clip_input_sha"]).decode()
    params, flow_q, fmap1, prevT, nets = {}, None, None, None, None
    K = T(seq.K)[None]
    bl = torch.tensor([seq.baseline])
    for t, fr in enumerate(seq.frames):
        Tt = T(fr.T)[None]
        params.update(K=K, T=Tt, previous_T=prevT, last_disp=flow_q, last_net_list=nets, fmap1=fmap1, baseline=bl)
        out = oracle.tc_stereo_forward(synth_weights, T(fr.image1)[None], T(fr.image2)[None], iters=6,
                                       params=params if flow_q is not None else None)
        flow_q, nets, fmap1, prevT = out["flow_q"], out["net_list"], out["fmap1"], Tt
        tag = "clip" if t == 0 else "rp_clip"
        assert epe(out["flow"], e2e_golden[f"{tag}_flow_{t}"]) <= 1e-4, t
        assert epe(out["flow_q"], e2e_golden[f"{tag}_flow_q_{t}"]) <= 1e-4, t


def test_slow_fast_schedule_golden(oracle, synth_weights):
    """The slow-fast GRU schedule (tc_stereo.py:182-187) against the reference's own output
    (tests/golden/e2e_configs.npz, tools/make_goldens_configs.py)."""
    import os
    import numpy as np
    import torch
    from conftest import GOLDEN, T, epe
    from tcs_mi355 import synth
    g = np.load(os.path.join(GOLDEN, "e2e_configs.npz"))
    fr = synth.make_sequence(23, n_frames=1, height=96, width=128, max_disp=32.0).frames[0]
    out = oracle.tc_stereo_forward(synth_weights, T(fr.image1)[None], T(fr.image2)[None], iters=4, args=oracle.default_args(slow_fast_gru=True))
    assert epe(out["flow"], g["slow_fast_flow"]) <= 1e-4
    assert epe(out["flow_q"], g["slow_fast_flow_q"]) <= 1e-4
