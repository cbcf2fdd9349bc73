"""Drop-in for the reference's core/tc_stereo.py: `TCStereo` with the hot path on MI355X.

Same constructor (an argparse-style Namespace), same sub-module names (reference checkpoints load
with strict=True), same `forward(image1, image2, iters, params, test_mode, frame_id)` signature and
the same test-mode output dict, so evaluate_stereo.py's loop (evaluate_stereo.py:170-197) runs
unchanged.  One optional addition: `prefetch(next_image1, next_image2)` for video loops (the next frame's
image-only stage is enqueued ahead of time).  What differs is underneath: correlation build/lookup, the temporal warp, the GRU update
step, both U-Nets, every stencil and the feature extractor's convolutions are hand-written HIP kernels
for gfx950 (libtcs_mi355.so); PyTorch-ROCm owns tensors, streams and graph capture.

Inference only: `test_mode=False` (training outputs and losses, train_stereo.py) is out of scope.
There is no CPU path: tensors must live on a HIP device and the library must be built.
"""
import contextlib
import os

import torch
import torch.nn as nn

from core.corr import CorrBlock1D
from core.extractor import BasicEncoder, MultiBasicEncoder, ResidualBlock, hip_head
from core.update import (IN_SUM_SLOTS, BasicMultiUpdateBlock, DispGradPredictor, DispRefine, DisparityCompletor, HiddenstateUpdater,
                         Lightfuse, _X, hip_conv)
from core.utils.utils import coords_grid
from tcs_mi355 import ops, s16


class autocast(contextlib.AbstractContextManager):
    """evaluate_stereo.py imports `autocast` from here (evaluate_stereo.py:14).  The HIP path keeps fp32 tensors and fp32-grade
    contractions (fp16 hi/lo split operands with fp32 accumulation, or fp32 MFMA — DESIGN.md section 5) whatever `enabled` says, so
    this is a no-op context."""

    def __init__(self, enabled=False, **_):
        self.enabled = enabled

    def __exit__(self, *exc):
        return False


class TCStereo(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.args = args
        self.scale_rate = 1 / (2 ** args.n_downsample)
        hd = list(args.hidden_dims)
        n = args.n_gru_layers
        self.cnet = MultiBasicEncoder(output_dim=[hd, hd], norm_fn=args.context_norm, downsample=args.n_downsample)
        self.update_block = BasicMultiUpdateBlock(args, hidden_dims=hd)
        self.context_zqr_convs = nn.ModuleList([nn.Conv2d(hd[i], hd[i] * 3, 3, padding=1) for i in range(n)])
        if args.shared_backbone:
            self.conv2 = nn.Sequential(ResidualBlock(128, 128, "instance", stride=1), nn.Conv2d(128, 256, 3, padding=1))
        else:
            self.fnet = BasicEncoder(output_dim=256, norm_fn="instance", downsample=args.n_downsample)
        self.previous_current_hideen_fuse = nn.ModuleList([Lightfuse(hd[i], hd[i]) for i in range(n)])   # (sic)
        self.disp_completor = DisparityCompletor()
        self.disp_grad_refine = DispGradPredictor(args)
        self.disp_refine = DispRefine(args)
        self.context_zqr_convs_grad = nn.ModuleList([nn.Conv2d(hd[i], 64, 3, padding=1) for i in range(n)])
        self.hiddenstate_update = HiddenstateUpdater(hd[0])
        # one pool of pre-split ("S16") activation buffers for the whole model: allocated (zero-filled) on first use,
        # reused every iteration and frame (tcs_mi355/s16.py)
        pool = s16.S16Pool()
        for m in self.modules():
            m._s16pool = pool

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

    def _coords0(self, img):
        """The x-coordinate grid at feature resolution, built once per shape (a constant of the frame)."""
        key = (int(img.shape[0]), int(img.shape[2]), int(img.shape[3]), str(img.device))
        cache = self.__dict__.setdefault("_coords0_cache", {})
        if key not in cache:
            cache[key] = coords_grid(key[0], key[1], key[2], device=img.device)[:, :1].contiguous()
        return cache[key]

    def initialize_flow(self, img):
        """coords0, coords1: the x-coordinate grid at feature resolution, [N,1,H,W] each (tc_stereo.py:66-73)."""
        n, _, h, w = img.shape
        g = coords_grid(n, h, w, device=img.device)[:, :1].contiguous()
        return g, g.clone()

    def upsample_flow(self, flow, mask, scale=True):
        """Convex x4 upsampling (tc_stereo.py:75-88).  Returns the upsampled flow (not clipped)."""
        if not scale or self.args.n_downsample != 2:
            raise NotImplementedError("HIP convex upsampling covers factor 4 with scaling")
        up, _ = ops.convex_upsample((-flow).float().contiguous(), mask.float().contiguous(), clip=False)   # kernel takes disparity
        return up

    def fuse_previous_current_hidden_state(self, net_list, warp_net_list):
        return [fuse(n, w) for n, w, fuse in zip(net_list, warp_net_list, self.previous_current_hideen_fuse)]

    def _weights_epoch(self):
        """Changes whenever a parameter tensor is replaced or written in place (load_state_dict, .copy_, optimiser step)."""
        return hash(tuple((p.data_ptr(), p._version) for p in self.parameters()))

    # -----------------------------------------------------------------------------------------
    def _features(self, image1, image2):
        a = self.args
        if a.shared_backbone:
            *cnet_list, trunk = self.cnet(torch.cat((image1, image2), 0), dual_inp=True, num_layers=a.n_gru_layers)
            fm = hip_head(self.conv2, trunk)                     # trunk: fp32 tensor, or an S16 tensor from the all-HIP context network
            fmap1, fmap2 = fm.split(fm.shape[0] // 2, 0)
        else:
            cnet_list = self.cnet(image1, num_layers=a.n_gru_layers)
            fmap1, fmap2 = self.fnet([image1, image2])
        return list(cnet_list), fmap1.float().contiguous(), fmap2.float().contiguous()

    def _pipeline(self):
        if getattr(self, "_graphs", None) is None:
            from tcs_mi355.graph import FrameGraphs
            self._graphs = FrameGraphs(self._extract_stage, self._refine_head, self._refine_loop, epoch_fn=self._weights_epoch,
                                       strict=os.environ.get("TCS_MI355_GRAPH_STRICT", "0") == "1")
        return self._graphs

    def _graph_mode(self):
        if getattr(self, "use_hip_graph", None) is None:
            self.use_hip_graph = os.environ.get("TCS_MI355_GRAPH", "1") != "0"
        return bool(self.use_hip_graph)

    @torch.no_grad()
    def prefetch(self, image1, image2, first=False, inputs_ready=False):
        """Optional, for callers that know the next frame (a video loop): enqueue the part of a frame that depends on nothing but its
        two images — feature / context networks, correlation pyramid, context convolutions (tc_stereo.py:101-116,147-149) — right
        behind the frame in flight, so that its host-side launch work is done while the GPU is still busy with that frame (on this
        stack the stage cannot run BESIDE the loop on the GPU).  The next `forward` with the SAME image tensors
        (same objects, unmodified) picks the result up; any other call simply extracts again.  `first`: that frame will be called with
        params=None (start of a sequence: the arg-max prior is then built with the correlation volume).  Call it right AFTER the
        `forward` it follows.  `inputs_ready` is kept for callers of round 3's two-stream version and no longer changes anything when
        the stage runs on the caller's stream.  Results are identical with and without the call."""
        if not image1.is_cuda:
            raise RuntimeError("TCStereo.prefetch needs HIP device tensors; there is no CPU fallback")
        self._pipeline().prefetch(image1, image2, first=bool(first), use_graph=self._graph_mode(), inputs_ready=bool(inputs_ready))

    @torch.no_grad()
    def forward(self, image1, image2, iters=12, params=None, test_mode=False, frame_id=0):
        """Disparity of a stereo pair, optionally conditioned on the previous frame (`params`):
        K [b,3,3], T / previous_T [b,4,4] world->camera, baseline [b], last_disp (= previous 'flow_q'),
        last_net_list, fmap1.  Returns {'flow' [b,1,H,W] (negative disparity, clipped at 0),
        'flow_q' [b,1,H/4,W/4], 'net_list', 'fmap1'} (tc_stereo.py:96-244).

        A frame is two fixed launch sequences with no host round trip — the image-only stage (`_extract_stage`) and the
        state-dependent stage (`_refine_stage`) — so by default each is captured once per (shape, iters, branch) into a HIP
        graph and replayed (tcs_mi355.graph); set TCS_MI355_GRAPH=0 or `model.use_hip_graph = False` for eager launches."""
        if not test_mode:
            raise NotImplementedError("TCStereo on MI355X is inference-only: call with test_mode=True")
        if iters < 1:
            raise ValueError("iters must be >= 1")
        if not image1.is_cuda:
            raise RuntimeError("TCStereo.forward needs HIP device tensors; there is no CPU fallback")
        temporal = None
        if params is not None:
            temporal = (params["K"], params["T"], params["previous_T"], params["baseline"], params["last_disp"],
                        list(params["last_net_list"]), params["fmap1"])
        return self._pipeline()(image1, image2, iters, temporal, use_graph=self._graph_mode())

    def _frame(self, image1, image2, iters, temporal):
        """One frame as a pure launch sequence on the current stream (both stages back to back)."""
        feats = self._extract_stage(image1, image2, temporal is None)
        return self._refine_loop(feats, self._refine_head(feats, temporal), iters)

    def _extract_stage(self, image1, image2, first):
        """Everything of a frame that depends only on its two images (tensors in, tensors out; capturable): matching features,
        correlation pyramid (+ the arg-max prior on a first frame), per-scale context terms (tc_stereo.py:101-116,147-149)."""
        a = self.args

        def correlate(fmap1, fmap2):
            corr_fn = CorrBlock1D(fmap1, fmap2, radius=a.corr_radius, num_levels=a.corr_levels, thres=a.init_thres, want_argmax=first)
            return corr_fn, (corr_fn.argmax_disp() if first else None)

        def context(cnet_list, relu_done=False):
            """Per-scale context terms of the GRUs and of the gradient predictor (tc_stereo.py:151-156): once per frame.  cz | cr | cq
            stay the three thirds of the context convolution's output (channel-slice views; the GRU epilogues index them in place)."""
            inp = [x[1] if relu_done else torch.relu(x[1]) for x in cnet_list]
            grads = [hip_conv(conv, [i]) for i, conv in zip(inp, self.context_zqr_convs_grad)]
            zqr = [list(hip_conv(conv, [i]).chunk(3, 1)) for i, conv in zip(inp, self.context_zqr_convs)]
            return zqr, grads, [x[0] for x in cnet_list]

        if a.shared_backbone and self.cnet.can16(image1):
            # the matching side (feature head -> correlation build) and the context side (per-scale heads -> context convolutions)
            # both start from the shared trunk: parallel graph branches.  The 7x7 stem reads the raw left | right images: the
            # normalisation to [-1, 1] and the batch concatenation (tc_stereo.py:101-107) happen in its input staging.
            from tcs_mi355.streams import fork_join
            trunk = self.cnet.trunk16(image1, right=image2, raw_images=True)

            def matching_side():
                fm = hip_head(self.conv2, trunk)
                f1, f2 = (t.float().contiguous() for t in fm.split(fm.shape[0] // 2, 0))
                return f1, f2, correlate(f1, f2)

            (fmap1, fmap2, (corr_fn, prior)), (inp_list, grad_list, net_list) = fork_join(
                [matching_side, lambda: context(self.cnet.heads16(trunk, True, a.n_gru_layers, relu_context=True), relu_done=True)], site="frame")
        else:
            image1 = (2 * (image1 / 255.0) - 1.0).contiguous()
            image2 = (2 * (image2 / 255.0) - 1.0).contiguous()
            cnet_list, fmap1, fmap2 = self._features(image1, image2)
            corr_fn, prior = correlate(fmap1, fmap2)
            inp_list, grad_list, net_list = context(cnet_list)
        return {"fmap1": fmap1, "corr_fn": corr_fn, "prior": prior, "inp_list": inp_list, "grad_list": grad_list, "net_list": net_list}

    def _refine_head(self, feats, temporal):
        """The state-dependent head of a frame (tensors in, tensors out; capturable): prior from the arg-max or from the pose warp of the
        previous frame, disparity completion, hidden-state warp and fusion (tc_stereo.py:119-172) -> what the loop starts from."""
        first = temporal is None
        fmap1, corr_fn = feats["fmap1"], feats["corr_fn"]
        net_list = feats["net_list"]
        if first:
            last_net_list = None
            sparse_disp, cost, sparse_mask = feats["prior"] if feats["prior"] is not None else corr_fn.argmax_disp()
        else:
            K, T, previous_T, baseline, last_disp, last_net_list, last_fmap1 = temporal
            # K_scale, its inverse, T @ inv(previous_T), previous_T @ inv(T): one tiny kernel, no host sync
            K_scale, K_scale_inv, relative_T, back_T = ops.pose_prepare(K, T, previous_T, self.scale_rate)
            # warp + normalise + cosine cost in one launch sequence; the warped feature map is never materialised
            sparse_disp, _, sparse_mask, cost = ops.warp_forward(
                (-last_disp).float().contiguous(), last_fmap1.float().contiguous(), relative_T, K_scale,
                K_scale_inv, baseline, cur_fmap=fmap1, want_fmap=False)

        pool = self._s16pool
        s16_head = "dc32" not in _X
        if s16_head:
            disp_init, _, _, net_list = self.disp_completor.run16(pool, sparse_disp, cost, sparse_mask, [c.float().contiguous() for c in net_list],
                                                                  tanh_nets=True)
        else:
            disp_init, _, _, net_list = self.disp_completor(sparse_disp, cost, sparse_mask, net_list, tanh_nets=True)
        disp_init = disp_init.float().contiguous()

        if last_net_list is None:
            warped = None
        else:
            grid = ops.backward_grid(disp_init, back_T, K_scale, K_scale_inv, baseline)
            warped = []
            for i, net in enumerate(last_net_list):
                warped.append(ops.bilinear_sample(net.float().contiguous(), grid))
                if i + 1 < len(last_net_list):
                    grid = ops.grid_halve(grid)

        # previous / current hidden-state fusion (tc_stereo.py:167-168; tanh applied by the completor's last convolutions)
        if s16_head:
            # the three Lightfuse cells on S16 tensors, updating the completor's hidden states in place; a first frame fuses with zeros
            for i, (h, fuse) in enumerate(zip(net_list, self.previous_current_hideen_fuse)):
                if warped is None:
                    x = pool.get(("frame", "zero", i), h.B, h.C, h.H, h.W, h.device)          # never written: stays zero
                else:
                    x = s16.to_s16(warped[i], out=pool.get(("frame", "warped", i), h.B, h.C, h.H, h.W, h.device))
                fuse.step16(pool, h, [x])
        else:
            net_list = self.fuse_previous_current_hidden_state(net_list, [torch.zeros_like(x) for x in net_list] if warped is None else warped)

        coords0 = self._coords0(fmap1)
        coords1 = (coords0 - disp_init).contiguous()
        trace = getattr(self, "_trace", None)          # debugging hook (eager mode only): intermediate tensors
        if trace is not None:
            trace.update(sparse_disp=sparse_disp, cost=cost, sparse_mask=sparse_mask, disp_init=disp_init,
                         net0=[t.float().clone() for t in net_list], iters=[])
        return {"coords1": coords1, "net_list": net_list}

    def _refine_loop(self, feats, start, iters):
        """The refinement loop and the upsampling (tc_stereo.py:175-229) from the head's disparity / hidden states (capturable)."""
        a = self.args
        fmap1, corr_fn = feats["fmap1"], feats["corr_fn"]
        inp_list, grad_list = feats["inp_list"], feats["grad_list"]
        coords1, net_list = start["coords1"], start["net_list"]
        coords0 = self._coords0(fmap1)
        trace = getattr(self, "_trace", None)

        # ---- refinement loop on pre-split activations (tcs_mi355/s16.py): hidden states, context features and the motion
        # feature buffer live in S16 pool buffers; 1-2 channel geometry (coords, disparity, gradients) stays fp32 ----
        pool = self._s16pool
        n3 = a.n_gru_layers == 3
        # (hidden states: the S16 tensors the head's completor / Lightfuse cells left, updated in place by the loop; fp32 only on the
        # A/B path of the fp32-tensor head)
        nets = [t if isinstance(t, s16.S16) else
                s16.to_s16(t.float().contiguous(), out=pool.get(("frame", "net", i), t.shape[0], t.shape[1], t.shape[2], t.shape[3], t.device))
                for i, t in enumerate(net_list)]
        grads16 = [s16.to_s16(t.float().contiguous(), out=pool.get(("frame", "ctxg", i), t.shape[0], t.shape[1], t.shape[2], t.shape[3], t.device))
                   for i, t in enumerate(grad_list)]
        dg_pre = self.disp_grad_refine.prepare(pool, grads16)      # the context share of three convolutions: once per frame
        self.disp_grad_refine.begin_frame(pool, coords1.shape[0], coords1.device)
        refined = up_mask = None
        # coords1 - coords0, the motion encoder's flow input (tc_stereo.py:180): once here, afterwards the blend kernel writes
        # it for the next iteration — as a tensor for the 7x7 stem and into channel 127 of the motion feature buffer
        flows_x = (coords1 - coords0).contiguous()
        motion = pool.get(("frame", "motion"), coords1.shape[0], 128, coords1.shape[2], coords1.shape[3], coords1.device)
        s16.set_channel(flows_x, motion, 127)
        ub = self.update_block
        ub.begin_frame()
        from tcs_mi355.streams import fork_join, join, mark, spawn
        hu_delta = None              # the hidden-state update of iteration i-1 runs at the head of iteration i's coarse chain
        early32 = None               # gru32 of iteration i + the early share of gru16, launched during iteration i-1
        plain = not a.slow_fast_gru and n3
        # Schedule of one iteration.  The critical chain — blend(i-1) -> hidden-state update -> pool -> gru16's late
        # share -> interp -> gru08 -> flow head -> gradient predictor -> refinement -> blend(i) — is the FIRST branch of every fork, so
        # that it stays in one launch list of the captured graph (tcs_mi355/streams.py: capture order at a fork); what has slack hangs off it as side
        # branches: [corr lookup -> motion encoder] (needs only coords1 / the flow of the previous blend, due at gru08), and [gru32 of
        # the NEXT iteration -> the share of gru16 that reads only net16 / interp(net32)] (needs net16, due ~500 us later), forked at the
        # iteration's join but enqueued after gru08 / the flow head, so that gru08 and not this branch continues the launch list.
        # (Measured alternatives, profiles/r04_ab_logs.txt: the encoder chain as the first branch +1.3 ms; the early gru32 branch forked behind the
        # flow head's stencil instead of at the join +0.27 ms — on this stack the branch starts ~430 us into the iteration whichever node it hangs off.)
        for itr in range(iters):
            def enc_branch():
                corr = corr_fn(coords1)
                return corr, ub.encoder.run(pool, flows_x, corr, motion)

            def coarse_branch():
                if hu_delta is not None:
                    self.hiddenstate_update.run(pool, nets[0], hu_delta)
                if isinstance(up32_now, tuple):          # gru32 ran ahead together with the early share of gru16
                    return ub.gru16_late(pool, nets, up32_now[1])
                if n3 and a.slow_fast_gru:
                    ub.run_coarse(pool, nets, inp_list, iter16=False, iter32=True, want_up16=False)
                if a.n_gru_layers >= 2 and a.slow_fast_gru:
                    ub.run_coarse(pool, nets, inp_list, iter16=True, iter32=n3, want_up16=False)
                return ub.run_coarse(pool, nets, inp_list, iter16=a.n_gru_layers >= 2, iter32=n3, up32=up32_now)

            up32_now = join(early32)                     # (None on the first iteration: gru32 then runs inside the coarse branch)
            early32 = None
            up16, (corr, m) = fork_join([coarse_branch, enc_branch], site="iter")
            run_ahead = plain and trace is None and itr + 1 < iters
            # net16 is final for this iteration: gru32 of the NEXT iteration + gru16's early share (update.py) may start from here
            def ahead():
                up32 = ub.run_gru32(pool, nets, inp_list)
                return (up32, ub.gru16_early(pool, nets, inp_list, up32)) if "nog16split" not in _X else up32
            at_join = mark() if run_ahead else None                # forked HERE, enqueued behind gru08 / the flow head (never the join's first child)
            sums = getattr(self, "_checksums", None)       # debugging hook (tools/determinism_check.py): device-side sums, no sync
            lazy = trace is None and sums is None           # the hooks want the flow head's / residual head's outputs as tensors
            delta_flow = ub.run_fine(pool, nets, inp_list, m, up16, lazy=lazy)
            if at_join is not None:
                early32 = spawn(ahead, site="gru32", after=at_join)
            # disp_q = x - (coords1 + delta), 5 * disp2disp_gradient_xy (update.py:199) and the gradient candidates in one
            # launch (with the flow head's last convolution finished from its tap partials); coords1 is replaced by the blend
            # kernel's output below
            if isinstance(delta_flow, s16.Taps):
                disp_q, g5, cands = s16.flow_taps_step_grads(coords1, delta_flow, scale=5.0)
            else:
                disp_q, g5, cands = ops.flow_step_grads(coords1, delta_flow, scale=5.0)
            disp_grad, context = self.disp_grad_refine.run(pool, g5, cands, dg_pre, lazy=lazy, slot=itr if itr < IN_SUM_SLOTS else None)
            last = itr == iters - 1
            # (not on the last iteration: no lookup follows; "nowarm": A/B)
            warm = corr_fn._pyr if (not last and "nowarm" not in _X and a.corr_levels == 4) else None
            refined, up_mask, fused = self.disp_refine.run(pool, disp_grad, disp_q, nets[0], context, want_mask=last, motion=motion,
                                                           warm_pyramid=warm, warm_radius=a.corr_radius)
            hu_delta = fused["delta_disp"]
            coords1, flows_x = fused["coords1"], fused["flow_x"]
            if sums is not None:
                sums.append({k: v.double().sum() for k, v in dict(
                    corr=corr, motion=m.data, net0=nets[0].data, net1=nets[1].data, net2=nets[2].data, delta_flow=delta_flow,
                    disp_grad=disp_grad, context=context.data, refined=refined, coords1=coords1).items()})
            if trace is not None:
                self.hiddenstate_update.run(pool, nets[0], hu_delta)       # debugging hook: states as of the end of the iteration
                hu_delta = None
                trace["iters"].append(dict(corr=corr, delta=delta_flow, disp_q=disp_q, refined=refined,
                                           net=[t.float() for t in nets]))
        if hu_delta is not None:
            self.hiddenstate_update.run(pool, nets[0], hu_delta)
        net_list = [t.float() for t in nets]

        flow_up, flow_q = ops.convex_upsample(refined.contiguous(), up_mask)
        return {"flow": flow_up, "flow_q": flow_q, "net_list": [x.detach() for x in net_list], "fmap1": fmap1.detach()}
