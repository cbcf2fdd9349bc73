#!/usr/bin/env python3
"""Generate tests/golden/*.npz|json by running the REFERENCE implementation on CPU.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
The reference is imported as-is; the only accommodation is a stub `cupy` module so that the import
of core/utils/splatting/softsplat.py:4 succeeds (its CUDA kernels are never called on CPU), and —
for the temporal clip only — a CPU stand-in for `softsplat_func.apply`, because the reference kernel
hard-asserts on non-CUDA tensors (softsplat.py:347-348).  Vectors that went through that stand-in
are labelled "restatement-pinned" in the file (key prefix `rp_`); everything else is the
reference's own arithmetic.

Fixtures are DATA (inputs, expected outputs); no reference source text is stored.
"""
import argparse
import hashlib
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
import tcs_paths  # noqa: E402

PKG = tcs_paths.PKG_DIR
OUT = os.path.join(ROOT, "tests", "golden")


def import_reference():
    cupy = types.ModuleType("cupy")
    cupy.memoize = lambda **k: (lambda f: f)          # decorator at softsplat.py:219
    sys.modules["cupy"] = cupy
    sys.path.insert(0, "/root/reference")
    warnings.filterwarnings("ignore")
    import core.tc_stereo as ref_tc                   # noqa
    import core.corr as ref_corr                      # noqa
    import core.update as ref_update                  # noqa
    import core.utils.geo_utils as ref_geo            # noqa
    import core.utils.utils as ref_utils              # noqa
    import core.utils.splatting.softsplat as ref_splat  # noqa
    return ref_tc, ref_corr, ref_update, ref_geo, ref_utils, ref_splat


def load_by_path(name, path):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


def sha(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()[:16]


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def N(x):
    return x.detach().cpu().numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-c2", action="store_true", help="skip the 640x480 32-iter frame (about 10 s)")
    a = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    ref_tc, ref_corr, ref_update, ref_geo, ref_utils, ref_splat = import_reference()
    weights = load_by_path("tcs_weights", os.path.join(PKG, "tcs_mi355", "weights.py"))
    synth = load_by_path("tcs_synth", os.path.join(PKG, "tcs_mi355", "synth.py"))
    oracle = load_by_path("tcs_oracle", os.path.join(ROOT, "oracle", "tcs_oracle.py"))
    from argparse import Namespace

    def mk_args(**kw):
        d = dict(hidden_dims=[128] * 3, shared_backbone=True, corr_levels=4, corr_radius=4, n_downsample=2,
                 context_norm="none", slow_fast_gru=False, n_gru_layers=3, mixed_precision=False, init_thres=0.5)
        d.update(kw)
        return Namespace(**d)

    # ---- 1. state-dict key -> shape ----------------------------------------------------------
    keys = {}
    for tag, kw in (("shared_backbone", {}), ("separate_fnet", dict(shared_backbone=False)),
                    ("context_norm_batch", dict(context_norm="batch"))):
        m = ref_tc.TCStereo(mk_args(**kw))
        keys[tag] = {k: list(v.shape) for k, v in m.state_dict().items()}
        print(tag, len(keys[tag]), sum(int(np.prod(s)) if s else 1 for s in keys[tag].values()))
    with open(os.path.join(OUT, "state_dict_keys.json"), "w") as f:
        json.dump(keys, f, indent=0, sort_keys=True)

    model = ref_tc.TCStereo(mk_args()).eval()
    sd = weights.load_synth_weights(model)

    g = np.random.Generator(np.random.Philox(key=20241004))
    rnd = lambda *s: g.standard_normal(s, dtype=np.float32)
    ops = {}

    with torch.no_grad():
        # ---- 2a. correlation ops (corr.py) ---------------------------------------------------
        for tag, (B, C, H, W) in (("a", (1, 256, 6, 48)), ("b", (2, 64, 3, 41))):
            f1, f2 = rnd(B, C, H, W), rnd(B, C, H, W)
            f2[:, :, :, 3:] = 0.6 * f2[:, :, :, 3:] + 0.4 * f1[:, :, :, :-3]    # some real matches at d=3
            cb = ref_corr.CorrBlock1D(T(f1), T(f2), num_levels=4, radius=4)
            ops[f"corr{tag}_f1"], ops[f"corr{tag}_f2"] = f1, f2
            for i in range(4):
                ops[f"corr{tag}_pyr{i}"] = N(cb.corr_pyramid[i]).reshape(B, H, W, -1)
            ops[f"corr{tag}_cost"] = N(cb.get_cost_volume())
            sdp, mc, mk = cb.argmax_disp()
            ops[f"corr{tag}_sparse_disp"], ops[f"corr{tag}_sparse_cost"], ops[f"corr{tag}_sparse_mask"] = N(sdp), N(mc), N(mk)
            xs = np.arange(W, dtype=np.float32)[None, None, None, :].repeat(B, 0).repeat(H, 2)
            coords = xs - g.uniform(-6, 30, size=xs.shape).astype(np.float32)           # includes x<0 and negative disparity
            coords[:, :, 0, :4] = np.array([-7.5, -0.25, W - 0.5, W + 9.0], np.float32)  # border cases
            ops[f"corr{tag}_coords"] = coords
            ops[f"corr{tag}_lookup"] = N(cb(T(coords)))

        # ---- 2b. geometry (geo_utils.py, utils.py) -------------------------------------------
        B, H, W = 2, 12, 20
        disp = np.abs(rnd(B, 1, H, W)) * 6 + 0.5
        K = np.array([[[40.0, 0, 10.0], [0, 40.0, 6.0], [0, 0, 1]]], np.float32).repeat(B, 0)
        Kinv = np.linalg.inv(K).astype(np.float32)
        ang = 0.03
        Trel = np.eye(4, dtype=np.float32)[None].repeat(B, 0)
        Trel[:, :3, :3] = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]], np.float32)
        Trel[:, :3, 3] = np.array([0.02, -0.01, -0.08], np.float32)
        Trel[1, :3, 3] = np.array([-0.05, 0.0, 0.3], np.float32)
        base = np.array([0.25, 0.25], np.float32)
        ops.update(geo_disp=disp, geo_K=K, geo_Kinv=Kinv, geo_Trel=Trel, geo_baseline=base)
        ops["geo_backward_grid"] = N(ref_geo.get_backward_grid(T(disp), T(Trel), T(K), T(Kinv), T(base)))
        ops["geo_grad_xy"] = N(ref_geo.disp2disp_gradient_xy(T(disp))[0])
        ops["geo_grad_cands"] = N(ref_geo.disp2disp_grad_candidates(T(disp), level=2))
        # warp() up to the splat input, from the reference's own helpers (geo_utils.py:169-195)
        fx = T(K)[:, 0, 0]
        depth = ref_geo.disp2depth(T(disp), T(base), fx)
        P = ref_geo.pixel2point(depth, T(Kinv))
        Pc = ref_geo.relative_transform(P, T(Trel))
        cd = Pc[:, -1:]
        cdisp = ref_geo.depth2disp(cd, T(base), fx)
        valid = ((cdisp > 0) & (cdisp < W)).float()
        flow = ref_geo.point2pixel(Pc, cd, T(K)) - ref_utils.coords_grid(B, H, W)
        metric = (cdisp - cdisp.mean()).clamp(-50, 50)
        ops.update(geo_warp_disp=N(cdisp), geo_warp_valid=N(valid), geo_warp_flow=N(flow), geo_warp_metric=N(metric))
        img = rnd(B, 5, H, W)
        grid = np.stack([g.uniform(-2, W + 1, size=(B, H, W)), g.uniform(-2, H + 1, size=(B, H, W))], 1).astype(np.float32)
        ops.update(samp_img=img, samp_grid=grid,
                   samp_out=N(ref_utils.bilinear_sampler(T(img), T(grid).permute(0, 2, 3, 1))))
        # restatement-pinned: full warp() through the stand-in splat
        class _Splat:
            @staticmethod
            def apply(tin, tflow):
                return oracle.softsplat_forward(tin, tflow)
        ref_splat.softsplat_func = _Splat
        fm = rnd(B, 16, H, W)
        wd, wf, wm = ref_geo.warp(T(disp), T(fm), T(Trel), T(K), T(Kinv), T(base))
        ops.update(rp_warp_fmap_in=fm, rp_warp_disp=N(wd), rp_warp_fmap=N(wf), rp_warp_mask=N(wm))

        # ---- 2c. stencils + cells (update.py, tc_stereo.py) ----------------------------------
        H, W = 8, 24
        d0 = np.abs(rnd(1, 1, H, W)) * 4 + 1
        gxy = rnd(1, 2, H, W) * 0.3
        cand, mat = model.disp_refine.propagate_disparity(T(gxy), T(d0))
        ops.update(prop_disp=d0, prop_grad=gxy, prop_cand=N(cand), prop_matrix=N(mat))
        flow_lr = -np.abs(rnd(1, 1, H, W)) * 5
        upmask = rnd(1, 144, H, W)
        ops.update(ups_flow=flow_lr, ups_mask=upmask, ups_out=N(model.upsample_flow(T(flow_lr), T(upmask))))

        h08, h16, h32 = rnd(1, 128, H, W) * 0.5, rnd(1, 128, H // 2, W // 2) * 0.5, rnd(1, 128, H // 4, W // 4) * 0.5
        ctx = [[rnd(1, 128, H >> i, W >> i) * 0.5 for _ in range(3)] for i in range(3)]
        corr36 = rnd(1, 36, H, W) * 0.3
        flw = -np.abs(rnd(1, 1, H, W)) * 3
        net = [T(h08).clone(), T(h16).clone(), T(h32).clone()]
        net_out, delta = model.update_block(net, [[T(c) for c in cc] for cc in ctx], T(corr36), T(flw))
        ops.update(ub_h08=h08, ub_h16=h16, ub_h32=h32, ub_corr=corr36, ub_flow=flw, ub_delta=N(delta),
                   ub_out08=N(net_out[0]), ub_out16=N(net_out[1]), ub_out32=N(net_out[2]))
        for i in range(3):
            for j, nm in enumerate("zrq"):
                ops[f"ub_ctx{i}{nm}"] = ctx[i][j]
        ops["enc_out"] = N(model.update_block.encoder(T(flw), T(corr36)))
        x128 = rnd(1, 128, H, W) * 0.5
        ops.update(lf_x=x128, lf_out=N(model.previous_current_hideen_fuse[0](T(h08), T(x128))))
        dd = rnd(1, 1, H, W) * 0.2
        ops.update(hu_delta=dd, hu_out=N(model.hiddenstate_update(T(h08), T(dd))))
        gctx = [rnd(1, 64, H >> i, W >> i) * 0.5 for i in range(3)]
        g_ref, g_ctx = model.disp_grad_refine(T(gxy), T(d0), [T(c) for c in gctx])
        ops.update(dg_ctx0=gctx[0], dg_ctx1=gctx[1], dg_ctx2=gctx[2], dg_grad=N(g_ref), dg_context=N(g_ctx))
        r_t, m_none = model.disp_refine(T(gxy), T(d0), T(h08), g_ctx, True)      # reference test_mode=True -> no mask
        r_f, m_f = model.disp_refine(T(gxy), T(d0), T(h08), g_ctx, False)
        assert m_none is None
        ops.update(dr_refined=N(r_f), dr_mask=N(m_f))
        sp_mask = (g.uniform(size=(1, 1, H, W)) > 0.5).astype(np.float32)
        sp_disp = d0 * sp_mask
        sp_cost = np.abs(rnd(1, 1, H, W)) * 0.5 * sp_mask
        nets_in = [rnd(1, 128, H >> i, W >> i) * 0.5 for i in range(3)]
        dc, dm, dw, dn = model.disp_completor(T(sp_disp), T(sp_cost), T(sp_mask), [T(x) for x in nets_in])
        ops.update(dc_disp=sp_disp, dc_cost=sp_cost, dc_mask=sp_mask, dc_net0=nets_in[0], dc_net1=nets_in[1], dc_net2=nets_in[2],
                   dc_completed=N(dc), dc_mono=N(dm), dc_w=N(dw), dc_out0=N(dn[0]), dc_out1=N(dn[1]), dc_out2=N(dn[2]))
        # context / feature extractor (stays PyTorch in the product; pinned so the oracle is complete)
        im = (g.uniform(0, 255, size=(2, 3, 32, 64))).astype(np.float32)
        *cl, trunk = model.cnet(T(2 * (im / 255.0) - 1.0), dual_inp=True, num_layers=3)
        ops.update(ext_img=im, ext_trunk=N(trunk), ext_fmap=N(model.conv2(trunk)))
        for i, pair in enumerate(cl):
            ops[f"ext_net{i}"], ops[f"ext_ctx{i}"] = N(pair[0]), N(pair[1])
    np.savez_compressed(os.path.join(OUT, "ops_small.npz"), **{k: np.asarray(v) for k, v in ops.items()})
    print("ops_small:", len(ops), "arrays")

    # ---- 3. end-to-end ---------------------------------------------------------------------------
    e2e = {}
    with torch.no_grad():
        # C1: 320x240 pair, D=64, 8 iters, first frame, padded to 256 rows (utils.py:10-28)
        pr = synth.make_pair(1)
        im1, im2 = T(pr.image1)[None], T(pr.image2)[None]
        padder = ref_utils.InputPadder(im1.shape, divis_by=32)
        (p1, p2) = padder.pad(im1, im2)
        out = model(p1, p2, iters=8, test_mode=True)
        e2e.update(c1_input_sha=np.frombuffer(sha(pr.image1, pr.image2).encode(), dtype=np.uint8),
                   c1_flow=N(out["flow"]), c1_flow_q=N(out["flow_q"]), c1_fmap1_sum=N(out["fmap1"].sum((2, 3))),
                   c1_net0=N(out["net_list"][0]).astype(np.float16))
        print("C1 |flow| mean", float(out["flow"].abs().mean()), "GT mean", float(pr.disp_gt.mean()))

        # temporal clip: 3 frames 160x128, 6 iters (stand-in splat => restatement-pinned for frames 1,2)
        seq = synth.make_sequence(7, n_frames=3, height=128, width=160, max_disp=48.0)
        params, flow_q, fmap1, prevT, nets = {}, None, None, None, None
        Kt = T(seq.K)[None]
        bl = torch.tensor([seq.baseline])
        e2e["clip_input_sha"] = np.frombuffer(sha(*[f.image1 for f in seq.frames], *[f.image2 for f in seq.frames]).encode(), dtype=np.uint8)
        for t, fr in enumerate(seq.frames):
            i1, i2 = T(fr.image1)[None], T(fr.image2)[None]
            Tt = T(fr.T)[None]
            params.update(K=Kt, T=Tt, previous_T=prevT, last_disp=flow_q, last_net_list=nets, fmap1=fmap1, baseline=bl)
            out = model(i1, i2, iters=6, test_mode=True, params=params if flow_q is not None else None)
            flow_q, nets, fmap1, prevT = out["flow_q"], out["net_list"], out["fmap1"], Tt
            tag = "clip" if t == 0 else "rp_clip"
            e2e[f"{tag}_flow_{t}"] = N(out["flow"])
            e2e[f"{tag}_flow_q_{t}"] = N(out["flow_q"])
            print("clip frame", t, "|flow| mean", float(out["flow"].abs().mean()), "GT", float(fr.disp_gt.mean()))

        if not a.skip_c2:
            seq2 = synth.make_sequence(2000, n_frames=1)
            fr = seq2.frames[0]
            out = model(T(fr.image1)[None], T(fr.image2)[None], iters=32, test_mode=True)
            e2e.update(c2_input_sha=np.frombuffer(sha(fr.image1, fr.image2).encode(), dtype=np.uint8),
                       c2_flow_q=N(out["flow_q"]))
            print("C2 |flow_q| mean", float(out["flow_q"].abs().mean()))
    np.savez_compressed(os.path.join(OUT, "e2e.npz"), **e2e)
    for fn in sorted(os.listdir(OUT)):
        print(fn, os.path.getsize(os.path.join(OUT, fn)) // 1024, "KiB")


if __name__ == "__main__":
    main()
