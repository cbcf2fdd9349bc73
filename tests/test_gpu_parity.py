"""GPU parity: the HIP path (through the C ABI) against reference-generated goldens and the CPU
oracle on identical inputs.  Tolerances are SURVEY.md §8c's: per-op <= 1e-5 abs for lookup / build /
stencils, <= 1e-4 for conv stacks, end-to-end EPE <= 1e-4 (8 iters) and <= 1e-3 (32 iters)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import T, epe, maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from tcs_mi355 import native
    native.lib()                                   # fail loudly when the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def model(dev, synth_weights):
    from argparse import Namespace
    from core.tc_stereo import TCStereo
    args = Namespace(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2,
                     context_norm="none", slow_fast_gru=False, n_gru_layers=3, mixed_precision=False, init_thres=0.5)
    m = TCStereo(args)
    m.load_state_dict(synth_weights, strict=True)
    return m.to(dev).eval()


def D(x, dev):
    return (x if torch.is_tensor(x) else T(x)).to(dev).contiguous()


# ------------------------------------------------------------------------------------------------
# correlation
# ------------------------------------------------------------------------------------------------
def test_corr_build_golden(dev, ops_golden):
    from tcs_mi355 import ops
    g = ops_golden
    for tag in "ab":
        p = ops.corr_build(D(g[f"corr{tag}_f1"], dev), D(g[f"corr{tag}_f2"], dev), argmax=True, cost_volume=True, natural=True)
        for i in range(4):
            assert maxdiff(p.natural[i], g[f"corr{tag}_pyr{i}"]) <= 1e-5, (tag, i)
        assert maxdiff(p.cost_volume, g[f"corr{tag}_cost"]) <= 1e-5
        d, c, m = p.sparse
        assert maxdiff(m, g[f"corr{tag}_sparse_mask"]) == 0
        assert maxdiff(d, g[f"corr{tag}_sparse_disp"]) == 0
        assert maxdiff(c, g[f"corr{tag}_sparse_cost"]) <= 1e-5


def test_corr_lookup_golden(dev, ops_golden):
    from tcs_mi355 import ops
    g = ops_golden
    for tag in "ab":
        p = ops.corr_build(D(g[f"corr{tag}_f1"], dev), D(g[f"corr{tag}_f2"], dev))
        out = ops.corr_lookup(p, D(g[f"corr{tag}_coords"], dev), 4)
        assert maxdiff(out, g[f"corr{tag}_lookup"]) <= 1e-5, tag


@pytest.mark.parametrize("shape", [(1, 256, 120, 160), (1, 256, 96, 312), (2, 32, 5, 19)])
def test_corr_full_size_vs_oracle(dev, oracle, shape):
    """C2 (120x160) and C5 (96x312) grid sizes + a ragged one; lookup with radius 4 and a generic radius."""
    from tcs_mi355 import ops
    B, Cc, H, W = shape
    gen = torch.Generator().manual_seed(H * W)
    f1, f2 = torch.randn(shape, generator=gen), torch.randn(shape, generator=gen)
    f2[..., 5:] = 0.5 * f2[..., 5:] + 0.5 * f1[..., :-5]
    p = ops.corr_build(D(f1, dev), D(f2, dev), argmax=True, natural=True)
    vol = oracle.corr_volume(f1, f2)
    pyr = oracle.corr_pyramid(vol)
    for i in range(4):
        assert maxdiff(p.natural[i], pyr[i]) <= 1e-5
    sd, sc, sm = oracle.argmax_disp(oracle.masked_cost_volume(vol))
    # argmax can legitimately flip on a near-tie of two fp32 sums; demand agreement on >= 99.9 % of pixels
    agree = (p.sparse[2].cpu() == sm).float().mean().item()
    assert agree >= 0.999
    same = (p.sparse[2].cpu() == sm) & (p.sparse[0].cpu() == sd)
    assert same.float().mean().item() >= 0.999
    xs = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
    coords = (xs - torch.rand(B, 1, H, W, generator=gen) * 40 + 4).contiguous()
    got = ops.corr_lookup(p, D(coords, dev), 4)
    assert maxdiff(got, oracle.corr_lookup(pyr, coords, 4)) <= 1e-5
    got3 = ops.corr_lookup(p, D(coords, dev), 3)
    assert maxdiff(got3, oracle.corr_lookup(pyr, coords, 3)) <= 1e-5


def test_corr_lookup_integer_coords_property(dev):
    """Size-independent property at full C2 size: at integer coordinates the centre tap of level 0
    is exactly V[b,h,w1,x] and out-of-range taps are exactly 0."""
    from tcs_mi355 import ops
    B, Cc, H, W = 1, 256, 120, 160
    gen = torch.Generator().manual_seed(3)
    f1, f2 = torch.randn(B, Cc, H, W, generator=gen).to(dev), torch.randn(B, Cc, H, W, generator=gen).to(dev)
    p = ops.corr_build(f1, f2, natural=True)
    xi = torch.randint(-8, W + 8, (B, 1, H, W), generator=gen)
    out = ops.corr_lookup(p, xi.float().to(dev), 4)
    vol = p.natural[0]
    inside = (xi >= 0) & (xi < W)
    want = torch.gather(vol, 3, xi.clamp(0, W - 1).to(dev).permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
    centre = out[:, 4:5]
    assert torch.equal(centre[inside.to(dev)], want[inside.to(dev)])
    assert (centre[~inside.to(dev)] == 0).all()


def test_corr_lookup_extreme_coords(dev, oracle):
    """Edge cases of the sampler (corr.py:33-52 / bilinear_sampler zero padding): coordinates far outside the row on
    either side give exact zeros, half-in windows match the oracle, huge magnitudes stay defined (no int overflow)."""
    from tcs_mi355 import ops
    B, Cc, H, W = 1, 32, 6, 40
    gen = torch.Generator().manual_seed(9)
    f1, f2 = torch.randn(B, Cc, H, W, generator=gen), torch.randn(B, Cc, H, W, generator=gen)
    p = ops.corr_build(D(f1, dev), D(f2, dev))
    pyr = oracle.corr_pyramid(oracle.corr_volume(f1, f2))
    coords = torch.zeros(B, 1, H, W)
    coords[0, 0, 0] = -1000.0
    coords[0, 0, 1] = 1000.0
    coords[0, 0, 2] = torch.linspace(-6.0, 3.0, W)           # windows straddling the left border
    coords[0, 0, 3] = torch.linspace(W - 4.0, W + 6.0, W)    # ... and the right border
    coords[0, 0, 4] = 3.0e9                                  # beyond int32
    coords[0, 0, 5] = -3.0e9
    got = ops.corr_lookup(p, D(coords, dev), 4)
    assert torch.isfinite(got).all()
    for row in (0, 1, 4, 5):
        assert (got[0, :, row] == 0).all(), row
    want = oracle.corr_lookup(pyr, coords, 4)
    assert maxdiff(got[0, :, 2:4], want[0, :, 2:4]) <= 1e-5


# ------------------------------------------------------------------------------------------------
# temporal warp
# ------------------------------------------------------------------------------------------------
def test_warp_geometry_and_grid_golden(dev, ops_golden):
    from tcs_mi355 import ops
    g = ops_golden
    d, K, Ki, Tr, b = (D(g[k], dev) for k in ("geo_disp", "geo_K", "geo_Kinv", "geo_Trel", "geo_baseline"))
    cd, va, fl, me = ops.warp_geometry(d, Tr, K, Ki, b)
    assert maxdiff(cd, g["geo_warp_disp"]) <= 1e-4
    assert maxdiff(va, g["geo_warp_valid"]) == 0
    assert maxdiff(fl, g["geo_warp_flow"]) <= 2e-4
    assert maxdiff(me, g["geo_warp_metric"]) <= 1e-4
    assert maxdiff(ops.backward_grid(d, Tr, K, Ki, b), g["geo_backward_grid"]) <= 2e-4
    assert maxdiff(ops.bilinear_sample(D(g["samp_img"], dev), D(g["samp_grid"], dev)), g["samp_out"]) <= 1e-5


def test_warp_forward_golden_and_api(dev, ops_golden):
    """warp() via the fused kernels vs the reference wrapper run through the stand-in splat
    (restatement-pinned), and the drop-in softsplat() wrapper vs the same numbers."""
    from core.utils.geo_utils import warp
    g = ops_golden
    d, K, Ki, Tr, b = (D(g[k], dev) for k in ("geo_disp", "geo_K", "geo_Kinv", "geo_Trel", "geo_baseline"))
    wd, wf, wm = warp(d, D(g["rp_warp_fmap_in"], dev), Tr, K, Ki, b)
    assert maxdiff(wm, g["rp_warp_mask"]) == 0
    # The splat is ill-conditioned in its inputs: a 1e-4 px change of a landing position (fp32 rounding
    # of the reprojection, test_warp_geometry_and_grid_golden) moves a bilinear weight by 1e-4, and the
    # disparities blended here differ by up to ~13 px -> max error ~1e-3, mean error far below.
    assert maxdiff(wd, g["rp_warp_disp"]) <= 2e-3 and epe(wd, g["rp_warp_disp"]) <= 2e-5
    assert maxdiff(wf, g["rp_warp_fmap"]) <= 2e-3 and epe(wf, g["rp_warp_fmap"]) <= 2e-5


def test_softsplat_vs_oracle(dev, oracle):
    from core.utils.splatting.softsplat import softsplat
    from tcs_mi355 import ops
    gen = torch.Generator().manual_seed(11)
    x = torch.randn(2, 7, 33, 45, generator=gen)
    flow = torch.randn(2, 2, 33, 45, generator=gen) * 3
    flow[0, 0, 0, 0] = float("nan")
    flow[1, 1, 2, 3] = 1e30
    flow[0, :, 5, 5] = torch.tensor([-100.0, 0.3])
    got = ops.softsplat_sum(D(x, dev), D(flow, dev))
    assert maxdiff(got, oracle.softsplat_forward(x, flow)) <= 1e-5
    metric = torch.randn(2, 1, 33, 45, generator=gen)
    valid = (torch.rand(2, 1, 33, 45, generator=gen) > 0.2).float()
    out, mask = softsplat(D(x, dev), D(flow, dev), D(metric, dev), "soft-clipeps", D(valid, dev))
    e = metric.exp()
    ref = oracle.softsplat_forward(torch.cat([x * valid * e, e * valid], 1), flow)
    assert maxdiff(mask, (ref[:, -1:] != 0).float()) == 0
    assert maxdiff(out, ref[:, :-1] / ref[:, -1:].clamp_min(1e-7)) <= 1e-4


def test_warp_full_size_vs_oracle(dev, oracle):
    """C2 grid (120x160, 256 channels): fused warp + cost vs the oracle."""
    from tcs_mi355 import ops
    gen = torch.Generator().manual_seed(5)
    B, Cc, H, W = 1, 256, 120, 160
    disp = torch.rand(B, 1, H, W, generator=gen) * 30 + 1
    disp[:, :, :, 80:] += 15                                  # a depth discontinuity
    fm, cur = torch.randn(B, Cc, H, W, generator=gen), torch.randn(B, Cc, H, W, generator=gen)
    K = torch.tensor([[[80.0, 0, 80.0], [0, 80.0, 60.0], [0, 0, 1.0]]])
    Ki = torch.linalg.inv(K)
    Tr = torch.eye(4)[None].clone()
    Tr[0, :3, 3] = torch.tensor([0.03, -0.01, -0.06])
    Tr[0, :3, :3] = torch.tensor([[0.9998, 0, 0.02], [0, 1, 0], [-0.02, 0, 0.9998]])
    b = torch.tensor([0.25])
    wd, wf, wm, wc = ops.warp_forward(D(disp, dev), D(fm, dev), D(Tr, dev), D(K, dev), D(Ki, dev), D(b, dev), cur_fmap=D(cur, dev))
    # (1) splat + normalise + cost arithmetic in isolation: feed the oracle the SAME landing positions
    #     (the HIP geometry, itself checked to 2e-4 px below and against the reference's numbers above)
    cd, va, fl, me = (t.cpu() for t in ops.warp_geometry(D(disp, dev), D(Tr, dev), D(K, dev), D(Ki, dev), D(b, dev)))
    e = me.exp()
    acc = oracle.softsplat_forward(torch.cat([torch.cat([cd, fm], 1) * va * e, e * va], 1), fl)
    om = (acc[:, -1:] != 0).float()
    o = acc[:, :-1] / acc[:, -1:].clamp_min(1e-7)
    od, of = o[:, :1], o[:, 1:]
    oc = (F.normalize(cur, dim=1) * F.normalize(of, dim=1)).sum(1, keepdim=True) * om
    assert maxdiff(wm, om) == 0
    assert maxdiff(wd, od) <= 1e-4          # values up to ~60 px: 2e-6 relative (atomic order + expf)
    assert maxdiff(wf, of) <= 1e-4
    assert maxdiff(wc, oc) <= 1e-5
    # (2) whole warp vs the oracle's own geometry: limited by the conditioning explained in
    #     test_warp_forward_golden_and_api (random 1..46 px disparities blended with 1e-4 weight noise)
    od2, of2, om2 = oracle.forward_warp(disp, fm, Tr, K, Ki, b)
    assert (wm.cpu() != om2).float().mean().item() <= 1e-4
    assert epe(wd, od2) <= 1e-4 and epe(wf, of2) <= 1e-4
    cdo, vao, flo, meo = oracle.forward_warp_inputs(disp, Tr, K, Ki, b)
    assert maxdiff(cd, cdo) <= 1e-4 and maxdiff(fl, flo) <= 3e-4 and maxdiff(va, vao) == 0
    grid = ops.backward_grid(D(disp, dev), D(Tr, dev), D(K, dev), D(Ki, dev), D(b, dev))
    assert maxdiff(grid, oracle.backward_grid(disp, Tr, K, Ki, b)) <= 2e-4
    nets = [torch.randn(1, 128, H >> i, W >> i, generator=gen) for i in range(3)]
    gcur = grid
    for i in range(3):
        # random features have O(1) gradients per pixel, so the sample is checked at the SAME grid ...
        assert maxdiff(ops.bilinear_sample(D(nets[i], dev), gcur), oracle.sample_bilinear(nets[i], gcur.cpu())) <= 1e-5
        if i < 2:
            # ... and the grid pyramid separately (tc_stereo.py:163); coordinates up to ~160 -> 1e-5 relative
            nxt = ops.grid_halve(gcur)
            want = 0.5 * F.interpolate(gcur.cpu(), scale_factor=0.5, mode="bilinear", align_corners=True)
            assert maxdiff(nxt, want) <= 2e-6 * max(1.0, float(want.abs().max()))
            gcur = nxt


# ------------------------------------------------------------------------------------------------
# stencils
# ------------------------------------------------------------------------------------------------
def test_stencils_golden(dev, ops_golden, model):
    from core.utils.geo_utils import disp2disp_grad_candidates, disp2disp_gradient_xy
    from tcs_mi355 import ops
    g = ops_golden
    d = D(g["geo_disp"], dev)
    assert maxdiff(disp2disp_gradient_xy(d)[0], g["geo_grad_xy"]) == 0
    ref = g["geo_grad_cands"]
    got = disp2disp_grad_candidates(d, level=2).cpu().numpy()
    fin = np.isfinite(ref)
    assert (np.isfinite(got) == fin).all()
    assert np.abs(got[fin] - ref[fin]).max() <= 1e-5 * max(1.0, np.abs(ref[fin]).max())
    cand, mat = model.disp_refine.propagate_disparity(D(g["prop_grad"], dev), D(g["prop_disp"], dev))
    assert maxdiff(cand, g["prop_cand"]) <= 1e-6
    assert maxdiff(mat, g["prop_matrix"]) == 0
    assert maxdiff(model.upsample_flow(D(g["ups_flow"], dev), D(g["ups_mask"], dev)), g["ups_out"]) <= 1e-5   # |values| up to ~60
    x = torch.randn(2, 6, 13, 17)
    assert maxdiff(ops.avgpool3s2(D(x, dev)), F.avg_pool2d(x, 3, stride=2, padding=1)) <= 1e-6
    assert maxdiff(ops.resize_bilinear(D(x, dev), 26, 34), F.interpolate(x, (26, 34), mode="bilinear", align_corners=True)) <= 1e-5


# ------------------------------------------------------------------------------------------------
# MFMA convolutions and the blocks built from them
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cfg", [
    dict(B=1, cins=(128, 128, 128), cout=256, k=3, H=12, W=40),       # gru08-like, 3 virtual sources
    dict(B=2, cins=(36,), cout=64, k=1, H=9, W=37),                    # convc1-like, ragged size, batch 2
    dict(B=1, cins=(64, 64), cout=127, k=3, H=8, W=24),                # encoder.conv: 127 outputs
    dict(B=1, cins=(1,), cout=64, k=7, H=10, W=33),                    # convf1: single input channel
    dict(B=2, cins=(3,), cout=64, k=7, H=13, W=70),                    # the extractors' 7x7 RGB stem (extractor.py:205,270)
    dict(B=1, cins=(27,), cout=96, k=1, H=7, W=65),                    # disp_f_stem: 27 inputs
    dict(B=1, cins=(256,), cout=1, k=3, H=6, W=32),                    # flow head conv2
    dict(B=1, cins=(128,), cout=384, k=3, H=30, W=40),                 # context_zqr conv: 3 cout tiles
])
@pytest.mark.parametrize("math", ["f32", "f16x3"])
def test_conv2d_vs_torch(dev, cfg, math):
    from tcs_mi355 import ops
    gen = torch.Generator().manual_seed(cfg["cout"] + cfg["k"])
    cin = sum(cfg["cins"])
    w = torch.randn(cfg["cout"], cin, cfg["k"], cfg["k"], generator=gen) * (2.0 / (cin * cfg["k"] ** 2)) ** 0.5
    b = torch.randn(cfg["cout"], generator=gen) * 0.1
    xs = [torch.randn(cfg["B"], c, cfg["H"], cfg["W"], generator=gen) for c in cfg["cins"]]
    add = torch.randn(cfg["B"], cfg["cout"], cfg["H"], cfg["W"], generator=gen)
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double(), padding=cfg["k"] // 2)
    pc = ops.pack_conv(D(w, dev), D(b, dev), math)
    assert pc.math == (ops.MATH_F16X3 if (math == "f16x3" and cin > 1 and cfg["k"] != 7) else ops.MATH_F32)
    got = ops.conv2d(pc, [D(x, dev) for x in xs])
    assert maxdiff(got, ref) <= 2e-5
    got = ops.conv2d(pc, [D(x, dev) for x in xs], act="relu", addend=D(add, dev), post_scale=0.25)
    assert maxdiff(got, 0.25 * torch.relu(ref + add.double())) <= 2e-5
    for act, fn in (("sigmoid", torch.sigmoid), ("tanh", torch.tanh), ("leaky", lambda t: F.leaky_relu(t, 0.01))):
        assert maxdiff(ops.conv2d(pc, [D(x, dev) for x in xs], act=act), fn(ref)) <= 2e-5


def test_cells_and_blocks_golden(dev, ops_golden, model):
    g = ops_golden
    net = [D(g["ub_h08"], dev), D(g["ub_h16"], dev), D(g["ub_h32"], dev)]
    inp = [[D(g[f"ub_ctx{i}{n}"], dev) for n in "zrq"] for i in range(3)]
    out, delta = model.update_block(net, inp, D(g["ub_corr"], dev), D(g["ub_flow"], dev))
    assert maxdiff(delta, g["ub_delta"]) <= 1e-4
    for o, k in zip(out, ("ub_out08", "ub_out16", "ub_out32")):
        assert maxdiff(o, g[k]) <= 1e-4, k
    assert maxdiff(model.update_block.encoder(D(g["ub_flow"], dev), D(g["ub_corr"], dev)), g["enc_out"]) <= 1e-4
    assert maxdiff(model.previous_current_hideen_fuse[0](D(g["ub_h08"], dev), D(g["lf_x"], dev)), g["lf_out"]) <= 1e-5
    # the same cell on pre-split tensors, as the frame's head runs it (in place on the hidden state)
    from core.update import pool_of
    from tcs_mi355 import s16
    h16 = s16.to_s16(D(g["ub_h08"], dev))
    model.previous_current_hideen_fuse[0].step16(pool_of(model), h16, [s16.to_s16(D(g["lf_x"], dev))])
    assert maxdiff(h16.float(), g["lf_out"]) <= 1e-4
    assert maxdiff(model.hiddenstate_update(D(g["ub_h08"], dev), D(g["hu_delta"], dev)), g["hu_out"]) <= 1e-5
    grad, ctx = model.disp_grad_refine(D(g["prop_grad"], dev), D(g["prop_disp"], dev), [D(g[f"dg_ctx{i}"], dev) for i in range(3)])
    assert maxdiff(grad, g["dg_grad"]) <= 1e-4
    assert maxdiff(ctx, g["dg_context"]) <= 1e-4
    ref, mask = model.disp_refine(D(g["prop_grad"], dev), D(g["prop_disp"], dev), D(g["ub_h08"], dev), D(g["dg_context"], dev), False)
    assert maxdiff(ref, g["dr_refined"]) <= 1e-4
    assert maxdiff(mask, g["dr_mask"]) <= 1e-4
    ref2, none = model.disp_refine(D(g["prop_grad"], dev), D(g["prop_disp"], dev), D(g["ub_h08"], dev), D(g["dg_context"], dev), True)
    assert none is None and maxdiff(ref2, ref) == 0
    comp, mono, w, nets = model.disp_completor(D(g["dc_disp"], dev), D(g["dc_cost"], dev), D(g["dc_mask"], dev),
                                               [D(g[f"dc_net{i}"], dev) for i in range(3)])
    assert maxdiff(comp, g["dc_completed"]) <= 1e-4
    for i in range(3):
        assert maxdiff(nets[i], g[f"dc_out{i}"]) <= 2e-4


# ------------------------------------------------------------------------------------------------
# end to end
# ------------------------------------------------------------------------------------------------
def test_e2e_c1_golden(dev, e2e_golden, model):
    """BASELINE config 1: 320x240, D=64, 8 iters, first frame, vs the reference's own output."""
    from tcs_mi355 import synth
    from tcs_mi355.harness import InputPadder
    pr = synth.make_pair(1)
    i1, i2 = D(pr.image1, dev)[None], D(pr.image2, dev)[None]
    p1, p2 = InputPadder(i1.shape, divis_by=32).pad(i1, i2)
    out = model(p1, p2, iters=8, test_mode=True)
    assert tuple(out["flow"].shape) == (1, 1, 256, 320)
    assert epe(out["flow_q"], e2e_golden["c1_flow_q"]) <= 1e-4
    assert epe(out["flow"], e2e_golden["c1_flow"]) <= 1e-4
    # per-channel sums over 5,120 pixels: 1e-2 absolute plus 1e-5 of the largest sum (fp32 summation-order noise)
    ref_sum = T(e2e_golden["c1_fmap1_sum"])
    assert maxdiff(out["fmap1"].sum((2, 3)), ref_sum) <= 1e-2 + 1e-5 * float(ref_sum.abs().max())
    assert float(out["flow"].max()) <= 0.0


def test_e2e_temporal_clip_golden(dev, e2e_golden, model):
    """3-frame clip through the evaluation harness: argmax branch then two warp-branch frames."""
    from tcs_mi355 import synth
    from tcs_mi355.harness import run_sequence
    seq = synth.make_sequence(7, n_frames=3, height=128, width=160, max_disp=48.0)
    preds = []
    stats = run_sequence(model, seq, iters=6, device=dev, collect=preds)
    assert len(stats.frames) == 3
    for t in range(3):
        tag = "clip" if t == 0 else "rp_clip"
        assert epe(-preds[t], e2e_golden[f"{tag}_flow_{t}"]) <= 1e-4, t


def test_e2e_c2_golden(dev, e2e_golden, model):
    """BASELINE config 2, frame 0: 640x480, 32 iterations, vs the reference.  1e-3 EPE (north_star)."""
    from tcs_mi355 import synth
    fr = synth.make_sequence(2000, n_frames=1).frames[0]
    out = model(D(fr.image1, dev)[None], D(fr.image2, dev)[None], iters=32, test_mode=True)
    assert epe(out["flow_q"], e2e_golden["c2_flow_q"]) <= 1e-3


def test_e2e_c2_temporal_32iters_vs_oracle(dev, oracle, synth_weights, model):
    """BASELINE config 2 beyond frame 0: the first TEMPORAL frame (warp branch) at 640x480 and 32 iterations against the
    CPU oracle on identical inputs (what bench.py reports as epe_vs_oracle_first_frames).  1e-3 EPE (north_star)."""
    from tcs_mi355 import synth
    from tcs_mi355.harness import run_sequence
    seq = synth.make_sequence(2000, n_frames=2, height=480, width=640, max_disp=192.0)
    got, ref = [], []
    run_sequence(model, seq, iters=32, device=dev, collect=got)
    torch.set_num_threads(min(16, len(__import__("os").sched_getaffinity(0))))
    run_sequence(lambda a, b, **kw: oracle.tc_stereo_forward(synth_weights, a, b, iters=kw["iters"], params=kw["params"]), seq,
                 iters=32, device=torch.device("cpu"), collect=ref)
    for t in range(2):
        e = epe(got[t], ref[t])
        print(f"C2 frame {t} (32 iters): EPE vs oracle {e:.2e}")
        assert e <= 1e-3, (t, e)


def test_c2_clip_teacher_forced_vs_oracle_and_domain_flags(dev, oracle, synth_weights, model):
    """BASELINE configs[1] in full: the 10-frame 640x480 clip at 32 iterations.

    (1) Parity on EVERY frame.  A free run cannot be compared beyond frame ~2: on the key-seeded random weights the recurrence
    amplifies last-bit differences (two runs of the SAME code differ by 0.15 px at frame 9).  So each frame
    t >= 1 is given the ORACLE's temporal state of frame t-1 (last_disp, last_net_list, fmap1, previous_T — what
    evaluate_stereo.py:182-197 threads through) and its output is compared with the oracle's frame t: 1e-3 EPE (north_star).
    (2) The free run of the same clip (HIP-graph replay, the bench's configuration, and eager launches) must stay finite with the
    S16 domain flags clean on every frame: the device-side replacement of the reference's NaN asserts (update.py:27-35,58-67,
    78-86,155-158).  Round 2's bench tripped them (a NaN from the fused hidden-state kernel's sigmoid, frame 9 iteration 30)."""
    import os
    from tcs_mi355 import s16, synth
    from tcs_mi355.harness import InputPadder, run_sequence
    seq = synth.make_sequence(2000, n_frames=10, height=480, width=640, max_disp=192.0)
    torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))
    K_raw = torch.as_tensor(seq.K, dtype=torch.float32)[None]
    baseline = torch.tensor([seq.baseline], dtype=torch.float32)
    s16.take_flags()
    prev = None                       # the oracle's outputs of the previous frame (CPU)
    prev_T = None
    worst = 0.0
    for t, fr in enumerate(seq.frames):
        im1, im2 = torch.as_tensor(fr.image1)[None], torch.as_tensor(fr.image2)[None]
        T = torch.as_tensor(fr.T)[None]
        padder = InputPadder(im1.shape, divis_by=32)
        (im1, im2), K = padder.pad(im1, im2, K=K_raw)
        params_cpu = params_gpu = None
        if prev is not None:
            params_cpu = dict(K=K, T=T, previous_T=prev_T, last_disp=prev["flow_q"], last_net_list=prev["net_list"], fmap1=prev["fmap1"],
                              baseline=baseline)
            params_gpu = {k: ([D(x, dev) for x in v] if isinstance(v, (list, tuple)) else D(v, dev)) for k, v in params_cpu.items()}
        want = oracle.tc_stereo_forward(synth_weights, im1, im2, iters=32, params=params_cpu)
        got = model(D(im1, dev), D(im2, dev), iters=32, test_mode=True, params=params_gpu)
        e = epe(padder.unpad(-got["flow"]), padder.unpad(-want["flow"]))
        eq = epe(got["flow_q"], want["flow_q"])
        print(f"C2 frame {t} (32 iters, oracle state in): EPE vs oracle {e:.2e} (1/4 res {eq:.2e})")
        worst = max(worst, e)
        assert e <= 1e-3, (t, e)
        assert all(bool(torch.isfinite(v).all()) for v in [got["flow"], got["flow_q"], *got["net_list"]]), t
        prev, prev_T = want, T
    assert s16.take_flags() == 0
    # (2) free runs: graph replay (twice over the clip: the second pass replays only) and eager launches
    for graph, passes in ((True, 2), (False, 1)):
        saved = getattr(model, "use_hip_graph", None)
        model.use_hip_graph = graph
        try:
            for _ in range(passes):
                outs = []
                st = run_sequence(model, seq, iters=32, device=dev, collect=outs)
                assert st.domain_flags == 0, (graph, hex(st.domain_flags))
                assert all(bool(torch.isfinite(o).all()) for o in outs) and max(float(o.abs().max()) for o in outs) < 640.0
        finally:
            model.use_hip_graph = saved


def test_e2e_kitti_shape_vs_oracle(dev, oracle, synth_weights, model):
    """BASELINE config 5's shape (KITTI raw 1242x375, padded to 1248x384 by InputPadder): first frame + one temporal
    frame at 2 iterations, and the first frame at the configuration's 32 iterations, HIP against the CPU oracle through the evaluation
    harness (un-padded outputs)."""
    from tcs_mi355 import synth
    from tcs_mi355.harness import run_sequence
    seq = synth.make_sequence(5, n_frames=2, height=375, width=1242, max_disp=192.0)
    got, want = [], []
    run_sequence(model, seq, iters=2, device=dev, collect=got)
    torch.set_num_threads(16)
    run_sequence(lambda a, b, **kw: oracle.tc_stereo_forward(synth_weights, a, b, iters=kw["iters"], params=kw["params"]), seq, iters=2,
                 device=torch.device("cpu"), collect=want)
    for t in range(2):
        assert tuple(got[t].shape[-2:]) == (375, 1242)
        assert epe(got[t], want[t]) <= 1e-4, t
    # ... and the configuration as BASELINE names it — 32 iterations — on the first frame (one oracle frame of this size is ~6 s of CPU):
    # the north-star bar, 1e-3 EPE
    one = type(seq)(seq.frames[:1], seq.K, seq.baseline)
    got32, want32 = [], []
    run_sequence(model, one, iters=32, device=dev, collect=got32)
    run_sequence(lambda a, b, **kw: oracle.tc_stereo_forward(synth_weights, a, b, iters=kw["iters"], params=kw["params"]), one, iters=32,
                 device=torch.device("cpu"), collect=want32)
    assert epe(got32[0], want32[0]) <= 1e-3, float(epe(got32[0], want32[0]))


@pytest.mark.parametrize("name,over", [("separate_fnet", dict(shared_backbone=False)),
                                        ("context_norm_batch", dict(context_norm="batch")),
                                        ("shared_backbone", dict(slow_fast_gru=True))])
def test_other_configurations_vs_oracle(dev, oracle, key_shapes, name, over):
    """The architecture switches of TCStereo(args) (tc_stereo.py:29-58) other than the shipped evaluation setting:
    separate feature network (instance norm at full resolution), batch-norm context network (stays on PyTorch-ROCm),
    slow-fast GRU schedule.  First frame + one temporal frame, 3 iterations, against the CPU oracle (whose slow-fast path
    is pinned by tests/golden/e2e_configs.npz).  n_gru_layers < 3 is not a usable switch of this model
    (update.py:206-210 indexes three context levels)."""
    from argparse import Namespace
    from core.tc_stereo import TCStereo
    from tcs_mi355 import synth
    from tcs_mi355.harness import run_sequence
    from tcs_mi355.weights import synth_state_dict
    W = synth_state_dict(key_shapes[name])
    base = dict(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2, context_norm="none",
                slow_fast_gru=False, n_gru_layers=3, mixed_precision=False, init_thres=0.5)
    base.update(over)
    m = TCStereo(Namespace(**base))
    m.load_state_dict(W, strict=True)
    m = m.to(dev).eval()
    seq = synth.make_sequence(17, n_frames=2, height=96, width=128, max_disp=32.0)
    got, want = [], []
    run_sequence(m, seq, iters=3, device=dev, collect=got)
    run_sequence(lambda a, b, **kw: oracle.tc_stereo_forward(W, a, b, iters=kw["iters"], params=kw["params"], args=oracle.default_args(**over)),
                 seq, iters=3, device=torch.device("cpu"), collect=want)
    for t in range(2):
        assert epe(got[t], want[t]) <= 1e-4, (over, t)


def test_slow_fast_schedule_golden(dev, synth_weights):
    """HIP against the reference's own slow-fast output (tests/golden/e2e_configs.npz)."""
    import os
    from argparse import Namespace
    from conftest import GOLDEN
    from core.tc_stereo import TCStereo
    from tcs_mi355 import synth
    g = np.load(os.path.join(GOLDEN, "e2e_configs.npz"))
    m = TCStereo(Namespace(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2, context_norm="none",
                           slow_fast_gru=True, n_gru_layers=3, mixed_precision=False, init_thres=0.5))
    m.load_state_dict(synth_weights, strict=True)
    m = m.to(dev).eval()
    fr = synth.make_sequence(23, n_frames=1, height=96, width=128, max_disp=32.0).frames[0]
    out = m(D(fr.image1, dev)[None], D(fr.image2, dev)[None], iters=4, test_mode=True)
    assert epe(out["flow"], g["slow_fast_flow"]) <= 1e-4
    assert epe(out["flow_q"], g["slow_fast_flow_q"]) <= 1e-4


def test_no_cpu_fallback(dev, model):
    with pytest.raises(RuntimeError):
        model(torch.zeros(1, 3, 64, 64), torch.zeros(1, 3, 64, 64), iters=1, test_mode=True)


def test_pose_prepare_vs_torch(dev):
    """Device-side camera algebra (tc_stereo.py:121-127,159) vs torch.linalg on the CPU in float64."""
    from tcs_mi355 import ops
    gen = torch.Generator().manual_seed(9)
    B = 3
    K = torch.tensor([[320.0, 0, 320.0], [0, 320.0, 240.0], [0, 0, 1.0]]).repeat(B, 1, 1)
    K[1, 0, 2] += 3.0

    def pose():
        q, _ = torch.linalg.qr(torch.randn(3, 3, generator=gen))
        T = torch.eye(4)
        T[:3, :3] = q * torch.sign(torch.linalg.det(q))
        T[:3, 3] = torch.randn(3, generator=gen)
        return T
    T = torch.stack([pose() for _ in range(B)])
    Tp = torch.stack([pose() for _ in range(B)])
    ks, ksi, trel, tback = ops.pose_prepare(D(K, dev), D(T, dev), D(Tp, dev), 0.25)
    Ks = K.double() * torch.tensor([0.25, 0.25, 1.0], dtype=torch.float64).view(1, 3, 1)
    assert maxdiff(ks, Ks) == 0
    assert maxdiff(ksi, torch.linalg.inv(Ks)) <= 1e-6
    assert maxdiff(trel, T.double() @ torch.linalg.inv(Tp.double())) <= 2e-6
    assert maxdiff(tback, Tp.double() @ torch.linalg.inv(T.double())) <= 2e-6
    ks2, ksi2, none1, none2 = ops.pose_prepare(D(K, dev))
    assert none1 is None and none2 is None and maxdiff(ks2, Ks) == 0


def test_graph_replay_matches_eager(dev, model):
    """The captured HIP graph and eager launches run the same kernels on the same data."""
    from tcs_mi355 import synth
    from tcs_mi355.harness import run_sequence
    seq = synth.make_sequence(11, n_frames=3, height=96, width=128, max_disp=32.0)
    model.use_hip_graph = False
    eager = []
    run_sequence(model, seq, iters=3, device=dev, collect=eager)
    model.use_hip_graph = True
    graphed, again = [], []
    run_sequence(model, seq, iters=3, device=dev, collect=graphed)      # captures
    run_sequence(model, seq, iters=3, device=dev, collect=again)        # pure replay
    assert model._graphs is not None and model._graphs.fell_back == 0 and all(v is not None for v in model._graphs.cache.values()), \
        "capture fell back to eager"
    for t in range(3):
        # not bit-identical on temporal frames: the splat's float atomics commit in a different order from run to run
        assert epe(graphed[t], eager[t]) <= 1e-5 and epe(again[t], eager[t]) <= 1e-5, t


def test_prefetch_overlaps_the_next_frames_extract_stage_without_changing_results(dev, model):
    """TCStereo.prefetch: the image-only stage of frame t+1 enqueued ahead of time (tcs_mi355/graph.py).  Same kernels on the same data: frame 0 (no splat atomics) must be bit-identical with and without it,
    temporal frames within the atomics' jitter; a prefetch for other images than the next call's is discarded, not used."""
    from tcs_mi355 import synth
    from tcs_mi355.harness import InputPadder
    seq = synth.make_sequence(13, n_frames=4, height=96, width=160, max_disp=32.0)
    K_raw = torch.as_tensor(seq.K, dtype=torch.float32, device=dev)[None]
    baseline = torch.tensor([seq.baseline], dtype=torch.float32, device=dev)
    frames = []
    for fr in seq.frames:
        im1, im2 = D(fr.image1, dev)[None], D(fr.image2, dev)[None]
        padder = InputPadder(im1.shape, divis_by=32)
        (im1, im2), K = padder.pad(im1, im2, K=K_raw)
        frames.append((im1.contiguous(), im2.contiguous(), K, D(fr.T, dev)[None]))

    def run(prefetch, wrong_first=False):
        outs, state = [], None
        for t, (i1, i2, K, T) in enumerate(frames):
            params = None if state is None else dict(K=K, T=T, previous_T=state[3], last_disp=state[0], last_net_list=state[1], fmap1=state[2],
                                                     baseline=baseline)
            o = model(i1, i2, iters=3, test_mode=True, params=params)
            if prefetch and t + 1 < len(frames):
                nxt = frames[t + 1] if not (wrong_first and t == 0) else frames[-1]         # a prefetch that will not be used
                model.prefetch(nxt[0], nxt[1], first=False, inputs_ready=(t % 2 == 0))      # both orderings of the extract stream
            state = (o["flow_q"], o["net_list"], o["fmap1"], T)
            outs.append(o["flow"].clone())
        return outs

    saved = getattr(model, "use_hip_graph", None)
    try:
        for graph in (True, False):
            model.use_hip_graph = graph
            plain = run(False)
            n0 = model._graphs.prefetched
            piped = run(True)
            assert model._graphs.prefetched - n0 == len(frames) - 1         # every frame but the first found its features waiting
            odd = run(True, wrong_first=True)
            assert torch.equal(piped[0], plain[0]) and torch.equal(odd[0], plain[0])
            for t in range(1, len(frames)):
                assert epe(piped[t], plain[t]) <= 1e-5 and epe(odd[t], plain[t]) <= 1e-5, (graph, t)
    finally:
        model.use_hip_graph = saved


def test_frame_graph_drop_is_fenced_and_unused_prefetches_age_out(dev, model):
    """FrameGraphs.drop() releases every captured graph at a safe point and REFUSES while a capture is recording (destroying a HIP graph during
    another stream's capture aborted the process in round 3); a prefetch nobody consumes gives its slot back after two calls (one
    mispredicted prefetch must not pin a slot, and with it the two-executable alternation, for the rest of the process)."""
    from tcs_mi355 import graph as gmod, synth
    pr_a, pr_b = synth.make_pair(5, height=96, width=128, max_disp=24.0), synth.make_pair(6, height=96, width=128, max_disp=24.0)
    a1, a2 = D(pr_a.image1, dev)[None], D(pr_a.image2, dev)[None]
    b1, b2 = D(pr_b.image1, dev)[None], D(pr_b.image2, dev)[None]
    saved = getattr(model, "use_hip_graph", None)
    model.use_hip_graph = True
    try:
        ref = model(a1, a2, iters=2, test_mode=True)["flow"].clone()
        fg = model._pipeline()
        assert fg.rf and fg.ex
        gmod._CAPTURING += 1
        try:
            with pytest.raises(RuntimeError):
                fg.drop()
        finally:
            gmod._CAPTURING -= 1
        assert fg.rf and fg.ex                                   # nothing was released by the refused call
        fg.drop()
        assert not fg.rf and not fg.ex
        again = model(a1, a2, iters=2, test_mode=True)["flow"]   # re-captures
        assert torch.equal(again, ref)
        # a prefetch for images that never come: fresh for one call, aged out by the second
        model.prefetch(b1, b2, first=True)
        assert sum(sl.fresh for sl in fg.slots) == 1
        model(a1, a2, iters=2, test_mode=True)
        assert sum(sl.fresh for sl in fg.slots) == 1
        out = model(a1, a2, iters=2, test_mode=True)["flow"]
        assert sum(sl.fresh for sl in fg.slots) == 0 and all(sl.token is None for sl in fg.slots)
        assert torch.equal(out, ref)
    finally:
        model.use_hip_graph = saved


def test_graph_recaptures_when_weights_change(dev, synth_weights):
    """A captured frame graph holds the packed weight images of its capture: after load_state_dict (or any in-place
    parameter write) the cache must be dropped, or replays would mix old packed weights with new biases."""
    from argparse import Namespace
    from core.tc_stereo import TCStereo
    from tcs_mi355 import synth
    args = Namespace(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2,
                     context_norm="none", slow_fast_gru=False, n_gru_layers=3, mixed_precision=False, init_thres=0.5)
    m = TCStereo(args)
    m.load_state_dict(synth_weights, strict=True)
    m = m.to(dev).eval()
    pr = synth.make_pair(3, height=96, width=128, max_disp=24.0)
    i1, i2 = D(pr.image1, dev)[None], D(pr.image2, dev)[None]
    m.use_hip_graph = True
    first = m(i1, i2, iters=2, test_mode=True)["flow"].clone()
    other = {k: (v * 1.25 if k.endswith("flow_head.conv1.weight") or k.endswith("gru08.convq.weight") else v) for k, v in synth_weights.items()}
    m.load_state_dict(other, strict=True)
    graphed = m(i1, i2, iters=2, test_mode=True)["flow"].clone()
    assert m._graphs.fell_back == 0 and m._graphs.captures == 2
    m.use_hip_graph = False
    eager = m(i1, i2, iters=2, test_mode=True)["flow"]
    assert epe(graphed, eager) <= 1e-5
    assert epe(graphed, first) > 1e-4, "the changed weights must change the output"
    with torch.no_grad():
        m.update_block.flow_head.conv1.weight.mul_(0.5)          # in-place write: _version changes
    m.use_hip_graph = True
    again = m(i1, i2, iters=2, test_mode=True)["flow"].clone()
    m.use_hip_graph = False
    assert epe(again, m(i1, i2, iters=2, test_mode=True)["flow"]) <= 1e-5 and m._graphs.captures == 3


def test_f16x3_split_is_fp32_grade(dev):
    """The fp16-split contraction must be as accurate as fp32 arithmetic, also for awkward magnitudes:
    tiny and large activations, weights far from 1, long K (gru08: 3456)."""
    from tcs_mi355 import ops
    gen = torch.Generator().manual_seed(1)
    B, cin, cout, H, W = 1, 384, 64, 8, 32
    x = torch.randn(B, cin, H, W, generator=gen)
    x[:, :64] *= 1e-4                                   # fp16-subnormal territory for the lo halves
    x[:, 64:128] *= 300.0                               # large values (disparities)
    w = torch.randn(cout, cin, 3, 3, generator=gen) * 0.02
    ref = F.conv2d(x.double(), w.double(), None, padding=1)
    scale = float(ref.abs().max())
    e32 = maxdiff(ops.conv2d(ops.pack_conv(D(w, dev), None, "f32"), [D(x, dev)]), ref) / scale
    e16 = maxdiff(ops.conv2d(ops.pack_conv(D(w, dev), None, "f16x3"), [D(x, dev)]), ref) / scale
    cpu = maxdiff(F.conv2d(x, w, None, padding=1), ref) / scale
    print(f"relative max error: fp32 MFMA {e32:.2e}, f16x3 {e16:.2e}, torch CPU fp32 {cpu:.2e}")
    # K = 3456 products per output: the fp32 MFMA is a sequential fmaf chain (error grows with K); the split
    # kernel adds exact fp16xfp16 products in fp32 and must be at least as good
    assert e32 <= 1e-5 and e16 <= 1e-5 and e16 <= 1.5 * e32


def test_graph_replay_full_size_many_frames(dev, model):
    """Regression: at 640x480 the temporal graph must stay correct on its 2nd, 3rd ... replay (the splat
    accumulator has to be re-cleared inside the graph).  4 frames, 2 iterations, graph vs eager."""
    from tcs_mi355 import synth
    from tcs_mi355.harness import run_sequence
    seq = synth.make_sequence(2000, n_frames=4, height=480, width=640, max_disp=192.0)
    model.use_hip_graph = False
    eager = []
    run_sequence(model, seq, iters=2, device=dev, collect=eager)
    model.use_hip_graph = True
    for rep in range(2):
        graphed = []
        run_sequence(model, seq, iters=2, device=dev, collect=graphed)
        errs = [epe(graphed[t], eager[t]) for t in range(4)]
        print(f"graph vs eager EPE per frame (replay {rep}): " + " ".join(f"{e:.2e}" for e in errs))
        for t in range(4):
            assert errs[t] <= 1e-5, (rep, t)


def test_frame_is_deterministic_with_parallel_branches(dev, model):
    """The loop's independent chains run on side streams / as parallel graph branches (tcs_mi355/streams.py).  No kernel may
    depend on what runs beside it: the same non-temporal frame (no splat atomics on that path) must come out bit-identical
    run after run, eagerly and from graph replays.  (Regression: the 7x7 stems once read their weights with negative-base
    LDS addressing and returned wrong values for a quarter wave when sharing a CU with another kernel's workgroups.)"""
    from tcs_mi355 import synth
    f = synth.make_sequence(2000, n_frames=1, height=480, width=640, max_disp=192.0).frames[0]
    i1, i2 = (torch.as_tensor(getattr(f, k)).to(dev).float()[None] for k in ("image1", "image2"))
    outs = []
    with torch.no_grad():
        for use_graph in (False, True):
            model.use_hip_graph = use_graph
            for _ in range(3):
                outs.append(model(i1, i2, iters=6, test_mode=True)["flow"].clone())
    model.use_hip_graph = True
    torch.cuda.synchronize()
    assert all(bool((o == outs[0]).all()) for o in outs[1:]), [float((o - outs[0]).abs().max()) for o in outs]


def test_stem7x7_beside_concurrent_kernels(dev):
    """k_conv7x7 on one stream while an MFMA convolution (LDS-DMA staging, 31-80 KB of LDS per workgroup) loops on another:
    every result must equal the solo run's bit for bit; the stem alternates between two inputs so that a stale or foreign
    value cannot pass.  tools/race_probe.py is the exploratory version of this test."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(0)
    R = lambda *sh: torch.randn(*sh, generator=gen).to(dev)
    H, W = 120, 160
    flows = [R(1, 1, H, W) * 20, R(1, 1, H, W) * 20]
    rgbs = [R(1, 3, 96, 128) * 50, R(1, 3, 96, 128) * 50]
    pc1 = ops.pack_conv(R(64, 1, 7, 7) * 0.2, R(64) * 0.1, "f16x3")
    pc3 = ops.pack_conv(R(64, 3, 7, 7) * 0.1, R(64) * 0.1, "f32")
    o1 = s16.zeros(1, 64, H, W, dev)
    victims = [lambda t: ops.conv2d(pc1, [flows[t & 1]], act="relu", out16=o1).data,
               lambda t: ops.conv2d(pc3, [rgbs[t & 1]], act="relu")]
    pca = ops.pack_conv(R(256, 128, 3, 3) * 0.03, R(256) * 0.1, "f16x3")
    xa, oa = s16.to_s16(R(1, 128, H, W)), s16.zeros(1, 256, H, W, dev)
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for vrun in victims:
        ref = [vrun(0).clone(), vrun(1).clone()]
        torch.cuda.synchronize()
        bad = torch.zeros((), dtype=torch.int64, device=dev)
        for cfg in (0, 101412):
            for t in range(60):
                with torch.cuda.stream(sb):
                    for _ in range(3):
                        s16.conv2d(pca, [xa], out16=oa, tile_cfg=cfg)
                with torch.cuda.stream(sa):
                    bad += (vrun(t) != ref[t & 1]).any().long()
        torch.cuda.synchronize()
        assert int(bad) == 0


@pytest.mark.parametrize("kind", ["none", "instance"])
@pytest.mark.parametrize("cin,cout,stride", [(64, 64, 1), (64, 96, 2), (96, 128, 1)])
def test_extractor_block_vs_oracle(dev, oracle, kind, cin, cout, stride):
    """ResidualBlock (extractor.py:5-58) on the HIP kernels against the CPU oracle's restatement of the same block
    (oracle._res_block, itself pinned end to end by the reference-generated e2e goldens) on identical weights and input."""
    from core.extractor import ResidualBlock
    torch.manual_seed(3)
    blk = ResidualBlock(cin, cout, kind, stride).eval()
    x = torch.randn(2, cin, 37, 70)                      # ragged size; negative inputs exercise the final ReLU
    W = {"blk." + k: v.detach().clone() for k, v in blk.state_dict().items()}
    with torch.no_grad():
        ref = oracle._res_block(W, "blk", x, kind, stride)
        got = blk.to(dev)(x.to(dev))
    assert got.shape == ref.shape
    assert maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_flow_step_grads_equals_separate_kernels(dev):
    """The fused opening of the gradient stage must equal tcs_flow_step + tcs_disp_gradient_xy + tcs_grad_candidates
    bit for bit (same arithmetic per pixel), also on ragged sizes and batch 2."""
    from tcs_mi355 import ops
    for B, H, W in ((1, 120, 160), (2, 7, 13)):
        gen = torch.Generator().manual_seed(H)
        xs = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W)
        coords1 = (xs - torch.rand(B, 1, H, W, generator=gen) * 30).contiguous().to(dev)
        delta = torch.randn(B, 1, H, W, generator=gen).to(dev)
        dq, g5, cands = ops.flow_step_grads(coords1, delta, scale=5.0)
        c_sep = coords1.clone()
        dq_sep = ops.flow_step(c_sep, delta)
        assert torch.equal(dq, dq_sep)
        assert torch.equal(g5, ops.disp_gradient_xy(dq_sep, scale=5.0))
        sep_c = ops.grad_candidates(dq_sep)
        both_nan = torch.isnan(cands) & torch.isnan(sep_c)           # 0/0 where three neighbours are collinear
        assert torch.equal(torch.where(both_nan, torch.zeros_like(cands), cands), torch.where(both_nan, torch.zeros_like(sep_c), sep_c))


@pytest.mark.parametrize("shape", [(1, 256, 120, 160), (2, 37, 7, 61), (1, 8, 3, 125)])
def test_conv3x3_cout1_vs_torch(dev, shape):
    """FlowHead.conv2 (update.py:13): the single-output 3x3 reduction kernel, full size and ragged sizes."""
    from tcs_mi355 import ops
    B, Cin, H, W = shape
    gen = torch.Generator().manual_seed(Cin)
    x = torch.randn(B, Cin, H, W, generator=gen)
    w = torch.randn(1, Cin, 3, 3, generator=gen) * 0.1
    bias = torch.randn(1, generator=gen)
    ref = F.conv2d(x.double(), w.double(), bias.double(), padding=1)
    got = ops.conv3x3_cout1(D(x, dev), D(w, dev), D(bias, dev))
    assert maxdiff(got, ref) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_batched_sequences_match_single(dev, model):
    """Independent sequences stacked on the batch dimension (bench.py --seqs-per-gpu) must give what each gives alone:
    every kernel indexes its batch element, per-sample poses / intrinsics / baselines included."""
    import bench
    from tcs_mi355 import synth
    seqs = [synth.make_sequence(40 + j, n_frames=3, height=96, width=128, max_disp=32.0) for j in range(2)]

    def run(group):
        r = bench.ClipRunner(model, group, dev, 3)
        return [r.step()["flow"].clone() for _ in range(3)]

    both = run(seqs)
    for j, q in enumerate(seqs):
        alone = run([q])
        for t in range(3):
            assert epe(both[t][j:j + 1], alone[t]) <= 1e-5, (j, t)


def test_stride2_deconv_instancenorm_vs_torch(dev):
    """The U-Net pieces on the fp32-tensor kernels: 3x3 stride-2 conv, ConvTranspose2d(4,2,1), InstanceNorm (+act, +addend)."""
    from tcs_mi355 import ops
    gen = torch.Generator().manual_seed(21)
    for (cin, cout, H, W) in ((64, 96, 120, 160), (96, 128, 60, 80), (40, 33, 9, 21)):
        x = torch.randn(1, cin, H, W, generator=gen)
        w = torch.randn(cout, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
        b = torch.randn(cout, generator=gen) * 0.1
        ref = F.relu(F.conv2d(x.double(), w.double(), b.double(), stride=2, padding=1))
        got = ops.conv2d(ops.pack_conv(D(w, dev), D(b, dev), "f16x3"), [D(x, dev)], act="relu", stride=2)
        assert tuple(got.shape) == tuple(ref.shape)
        assert maxdiff(got, ref) <= 2e-5, (cin, cout, H, W)
    for (cin, cout, H, W) in ((128, 96, 30, 40), (96, 64, 60, 80), (24, 10, 5, 7)):
        x = torch.randn(2, cin, H, W, generator=gen)
        wt = torch.randn(cin, cout, 4, 4, generator=gen) * (2.0 / (16 * cin)) ** 0.5
        ref = F.conv_transpose2d(x.double(), wt.double(), None, stride=2, padding=1)
        got = ops.deconv4x4s2(ops.pack_deconv4x4s2(D(wt, dev)), [D(x, dev)])
        assert tuple(got.shape) == (2, cout, 2 * H, 2 * W)
        assert maxdiff(got, ref) <= 2e-5, (cin, cout, H, W)
    x = torch.randn(2, 7, 33, 45, generator=gen) * 3 + 1
    add = torch.randn(2, 7, 33, 45, generator=gen)
    assert maxdiff(ops.instance_norm(D(x, dev)), F.instance_norm(x.double())) <= 1e-5
    assert maxdiff(ops.instance_norm(D(x, dev), act="leaky", addend=D(add, dev)),
                   F.leaky_relu(F.instance_norm(x.double()), 0.01) + add.double()) <= 1e-5
    assert maxdiff(ops.instance_norm(D(x, dev), act="relu"), F.relu(F.instance_norm(x.double()))) <= 1e-5


def test_rgb_stem_reads_raw_image_pairs(dev):
    """tcs_conv_desc.in_transform / src_batch2: the 7x7 RGB stem normalises 0..255 images to [-1, 1] and appends the right images to the
    left ones along the batch inside its input staging (tc_stereo.py:101-107): equal to the stem on the torch-prepared batch."""
    from tcs_mi355 import ops
    gen = torch.Generator().manual_seed(41)
    l = torch.randint(0, 256, (2, 3, 37, 70), generator=gen).float()
    r = torch.randint(0, 256, (2, 3, 37, 70), generator=gen).float()
    w = torch.randn(64, 3, 7, 7, generator=gen) * 0.1
    b = torch.randn(64, generator=gen) * 0.1
    pc = ops.pack_conv(D(w, dev), D(b, dev), "f32")
    both = torch.cat((2 * (D(l, dev) / 255.0) - 1.0, 2 * (D(r, dev) / 255.0) - 1.0), 0).contiguous()
    want = ops.conv2d(pc, [both], act="relu")
    got = ops.conv2d(pc, [D(l, dev)], act="relu", image_pair=D(r, dev), in_transform=1)
    # (not bit-equal to the torch-prepared batch: torch on the GPU divides by a scalar as a multiplication by its rounded reciprocal, the
    # kernel divides like the CPU reference does — one ulp of the normalised pixel)
    assert tuple(got.shape) == (4, 64, 37, 70) and maxdiff(got, want) <= 1e-5
    norm = torch.cat((2 * (l.double() / 255.0) - 1.0, 2 * (r.double() / 255.0) - 1.0), 0)
    ref = F.relu(F.conv2d(norm, w.double(), b.double(), padding=3))
    assert maxdiff(got, ref) <= 2e-5
    with pytest.raises(RuntimeError):                      # only the RGB stem honours it
        ops.conv2d(ops.pack_conv(D(torch.randn(8, 4, 3, 3), dev), None, "f32"), [D(torch.randn(1, 4, 8, 32), dev)], in_transform=1)
