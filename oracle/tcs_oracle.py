"""CPU oracle for the TC-Stereo inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (numpy + PyTorch CPU ops) of the reference
algorithm.  It exists to CHECK the HIP product path; it is never the thing measured or shipped.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product package (`temporally-consistent-stereo-matching_amd/`) must never import from `oracle/`.

Pinning status: reference-pinned.  `tools/make_goldens.py` (run in the build container, where
/root/reference is readable) imports the reference model and writes the vectors under
`tests/golden/`; `tests/test_oracle_golden.py` checks every function below against them.  The one
exception is `softsplat_forward`: the reference kernel is CUDA-only (softsplat.py:347-348 asserts on
CPU tensors), so that single function is "restatement-pinned" from the kernel text
(softsplat.py:285-335) and the wrapper maths (softsplat.py:232-274).

All citations are file:line into the reference tree (jiaxiZeng/Temporally-Consistent-Stereo-Matching).
Everything is functional over a flat state dict `W` (the reference's `state_dict()` key names), so
the same weights feed the reference, this oracle and the HIP path.
"""
from __future__ import annotations

import math
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "default_args", "corr_volume", "corr_pyramid", "masked_cost_volume", "corr_lookup", "argmax_disp",
    "softsplat_forward", "forward_warp", "backward_grid", "sample_bilinear", "warp_hidden_states",
    "disp_gradient_xy", "grad_candidates", "propagate_disparity", "convex_upsample", "conv_gru",
    "gru_1x1", "hidden_state_update", "motion_encoder", "update_block", "disp_grad_predictor",
    "disp_refine", "disparity_completor", "context_encoder", "tc_stereo_forward",
]


def default_args(**over):
    """Architecture flags of the shipped evaluation scripts (tartanair_evaluate.sh:1-7)."""
    a = dict(hidden_dims=[128, 128, 128], shared_backbone=True, corr_levels=4, corr_radius=4,
             n_downsample=2, context_norm="none", slow_fast_gru=False, n_gru_layers=3,
             mixed_precision=False, init_thres=0.5)
    a.update(over)
    return SimpleNamespace(**a)


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def _conv(W, name, x, stride=1, padding=None):
    w = W[name + ".weight"]
    b = W.get(name + ".bias")
    if padding is None:
        padding = w.shape[-1] // 2
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def _xy_grid(n, h, w, like):
    """coords_grid (core/utils/utils.py:100-103): channel 0 = x (column), channel 1 = y (row)."""
    ys, xs = torch.meshgrid(torch.arange(h, dtype=like.dtype), torch.arange(w, dtype=like.dtype), indexing="ij")
    return torch.stack([xs, ys], 0)[None].repeat(n, 1, 1, 1)


def _norm(W, name, x, kind):
    """Normalisation layers used by the extractor (core/extractor.py:15-37)."""
    if kind == "none":
        return x
    if kind == "instance":
        return F.instance_norm(x)
    if kind == "batch":  # eval mode: running statistics
        return F.batch_norm(x, W[name + ".running_mean"], W[name + ".running_var"], W[name + ".weight"],
                            W[name + ".bias"], training=False)
    if kind == "group":
        c = x.shape[1]
        groups = 8 if name.endswith("cnet.norm1") or name.endswith("fnet.norm1") else c // 8
        return F.group_norm(x, groups, W[name + ".weight"], W[name + ".bias"])
    raise ValueError(kind)


# ----------------------------------------------------------------------------------------------
# a2/a3: correlation volume, pyramid, masked cost volume            (core/corr.py:8-31,54-62)
# ----------------------------------------------------------------------------------------------
def corr_volume(fmap1, fmap2):
    """V[b,h,w1,w2] = <f1/|f1|, f2/|f2|> over channels; eps 1e-12 (corr.py:58-60, F.normalize default)."""
    n1 = fmap1 / fmap1.norm(dim=1, keepdim=True).clamp_min(1e-12)
    n2 = fmap2 / fmap2.norm(dim=1, keepdim=True).clamp_min(1e-12)
    # [B,C,H,W1] x [B,C,H,W2] -> [B,H,W1,W2]
    return torch.matmul(n1.permute(0, 2, 3, 1), n2.permute(0, 2, 1, 3))


def corr_pyramid(vol, num_levels=4):
    """Levels 0..num_levels-1 of the 1-D average pyramid along w2 (corr.py:20-23).  The reference
    stores num_levels+1 entries but only reads the first num_levels (corr.py:39).  A trailing odd
    element is dropped, like avg_pool2d([1,2])."""
    pyr = [vol]
    for _ in range(num_levels - 1):
        v = pyr[-1]
        w = v.shape[-1] // 2
        pyr.append(0.5 * (v[..., 0:2 * w:2] + v[..., 1:2 * w:2]))
    return pyr


def masked_cost_volume(vol):
    """cost[b,w2,h,w1] = V[b,h,w1,w2] * [w2 <= w1] (corr.py:25-31)."""
    b, h, w1, w2 = vol.shape
    keep = (torch.arange(w2).view(1, w2, 1, 1) <= torch.arange(w1).view(1, 1, 1, w1)).to(vol.dtype)
    return vol.permute(0, 3, 1, 2).contiguous() * keep


def corr_lookup(pyr, coords, radius=4):
    """a4 (corr.py:33-52 + utils.py:82-97).  coords [B,1,H,W1] = x position in the right image.
    out[b, i*(2r+1)+k, h, w] = lerp of level i at x/2^i + (k-r); taps outside [0, W2_i-1] read 0."""
    b, _, h, w = coords.shape
    outs = []
    for i, lv in enumerate(pyr):
        w2 = lv.shape[-1]
        x = coords[:, 0] / (2 ** i)                      # [B,H,W]
        x0 = torch.floor(x)
        a = (x - x0).unsqueeze(-1)                        # same fraction for all taps of a level
        idx = x0.long().unsqueeze(-1) + torch.arange(-radius, radius + 2).view(1, 1, 1, -1)  # 2r+2 taps
        ok = (idx >= 0) & (idx < w2)
        vals = torch.gather(lv, 3, idx.clamp(0, w2 - 1)) * ok.to(lv.dtype)
        outs.append((1 - a) * vals[..., :-1] + a * vals[..., 1:])
    return torch.cat(outs, -1).permute(0, 3, 1, 2).contiguous()


def argmax_disp(cost):
    """a5 (corr.py:67-79).  cost [B,W2,H,W1] (masked).  Returns sparse_disp, main_cost, mask, each
    [B,1,H,W1].  Second-best suppression writes 0 (not -inf); threshold is the literal 0.3."""
    b, w2, h, w1 = cost.shape
    main, idx = cost.max(dim=1, keepdim=True)
    j = torch.arange(w2).view(1, w2, 1, 1)
    near = (j >= idx - 1.5) & (j < idx + 1.5)
    sub = torch.where(near, torch.zeros_like(cost), cost).max(dim=1, keepdim=True)[0]
    mask = (main - sub > 0.3).to(cost.dtype)
    disp = (torch.arange(w1).view(1, 1, 1, w1) - idx).to(cost.dtype)
    return disp * mask, main * mask, mask


# ----------------------------------------------------------------------------------------------
# a6/a7: forward warp of previous disparity + features                (geo_utils.py:158-198)
# ----------------------------------------------------------------------------------------------
def softsplat_forward(inp, flow):
    """Summation splat, one contribution per (n,c,y,x) to its 4 bilinear corners
    (softsplat.py:285-335).  Deterministic order (numpy add.at), float32 like the kernel."""
    dt = inp.dtype
    x_np = inp.detach().cpu().numpy()
    f_np = flow.detach().cpu().numpy()
    n, c, h, w = x_np.shape
    out = np.zeros_like(x_np)
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    for b in range(n):
        fx = xs.astype(x_np.dtype) + f_np[b, 0]
        fy = ys.astype(x_np.dtype) + f_np[b, 1]
        fin = np.isfinite(fx) & np.isfinite(fy)
        fx = np.where(fin, fx, 0)
        fy = np.where(fin, fy, 0)
        x0 = np.floor(fx).astype(np.int64)
        y0 = np.floor(fy).astype(np.int64)
        one = x_np.dtype.type(1)
        wx1 = fx - x0.astype(x_np.dtype)      # weight of the east column
        wx0 = (x0.astype(x_np.dtype) + one) - fx
        wy1 = fy - y0.astype(x_np.dtype)
        wy0 = (y0.astype(x_np.dtype) + one) - fy
        corners = [(x0, y0, wx0 * wy0), (x0 + 1, y0, wx1 * wy0), (x0, y0 + 1, wx0 * wy1), (x0 + 1, y0 + 1, wx1 * wy1)]
        flat = out[b].reshape(c, h * w)
        src = x_np[b].reshape(c, h * w)
        for cx, cy, wt in corners:
            ok = (fin & (cx >= 0) & (cx < w) & (cy >= 0) & (cy < h)).ravel()
            tgt = (cy * w + cx).ravel()[ok]
            contrib = src[:, ok] * wt.ravel()[ok][None, :]
            np.add.at(flat, (slice(None), tgt), contrib)
    return torch.from_numpy(out).to(dt)


def _disp_to_points(disp, K, K_inv, baseline):
    """disp2depth + pixel2point (geo_utils.py:7-16,32-42)."""
    n, _, h, w = disp.shape
    fx = K[:, 0, 0].view(-1, 1, 1, 1)
    depth = baseline.view(-1, 1, 1, 1) * fx / disp.clamp_min(0.001)
    g = _xy_grid(n, h, w, disp)
    pix = torch.cat([g, torch.ones_like(depth)], 1).view(n, 3, -1)
    pts = depth.view(n, 1, -1) * torch.matmul(K_inv, pix)
    return pts, depth


def _rigid(pts, T):
    """relative_transform (geo_utils.py:135-145): homogeneous 4x4 applied to [N,3,HW]."""
    n = pts.shape[0]
    hom = torch.cat([pts, torch.ones_like(pts[:, :1])], 1)
    return torch.matmul(T, hom)[:, :3]


def _project(pts, depth_flat, K):
    """point2pixel (geo_utils.py:45-57): NaN/Inf -> -1."""
    pix = torch.matmul(K, pts) / depth_flat
    pix = torch.where(torch.isnan(pix) | torch.isinf(pix), -torch.ones_like(pix), pix)
    return pix[:, :2]


def forward_warp_inputs(disp, relative_T, K, K_inv, baseline):
    """Everything `warp` computes before the splat (geo_utils.py:169-195): new disparity, validity,
    forward flow, soft-splat metric."""
    n, _, h, w = disp.shape
    fx = K[:, 0, 0].view(-1, 1, 1, 1)
    pts, _ = _disp_to_points(disp, K, K_inv, baseline)
    cur = _rigid(pts, relative_T)
    cur_depth = cur[:, 2:3]
    cur_disp = baseline.view(-1, 1, 1, 1) * fx / cur_depth.reshape(n, 1, h, w)
    cur_disp = torch.where(torch.isnan(cur_disp) | torch.isinf(cur_disp), -torch.ones_like(cur_disp), cur_disp)
    valid = ((cur_disp > 0) & (cur_disp < w)).to(disp.dtype)
    pix = _project(cur, cur_depth, K).reshape(n, 2, h, w)
    flow = pix - _xy_grid(n, h, w, disp)
    metric = (cur_disp - cur_disp.mean()).clamp(-50, 50)   # GLOBAL mean (geo_utils.py:193)
    return cur_disp, valid, flow, metric


def forward_warp(disp, fmap, relative_T, K, K_inv, baseline):
    """a6+a7: warp() with softsplat mode 'soft-clipeps' (geo_utils.py:158-198, softsplat.py:232-274).
    Returns warped disparity [N,1,H,W], warped fmap [N,C,H,W], mask [N,1,H,W]."""
    cur_disp, valid, flow, metric = forward_warp_inputs(disp, relative_T, K, K_inv, baseline)
    feats = torch.cat([cur_disp, fmap], 1) * valid
    e = metric.exp()
    splat_in = torch.cat([feats * e, e * valid], 1)
    out = softsplat_forward(splat_in, flow)
    norm = out[:, -1:]
    mask = (norm != 0).to(disp.dtype)
    out = out[:, :-1] / norm.clamp_min(1e-7)
    return out[:, :1], out[:, 1:], mask


# ----------------------------------------------------------------------------------------------
# a10: backward grid + hidden-state warp          (geo_utils.py:201-236, tc_stereo.py:155-163)
# ----------------------------------------------------------------------------------------------
def backward_grid(disp, relative_T, K, K_inv, baseline):
    n, _, h, w = disp.shape
    pts, _ = _disp_to_points(disp.clamp_min(0.01), K, K_inv, baseline)
    prev = _rigid(pts, relative_T)
    pdepth = prev[:, 2:3]
    pix = _project(prev, pdepth, K)
    pix = torch.where(pdepth > 0, pix, -torch.ones_like(pix))
    return pix.reshape(n, 2, h, w)


def sample_bilinear(img, grid_xy):
    """bilinear_sampler (utils.py:82-97) in pixel coordinates: zeros padding, align_corners=True.
    img [N,C,H,W]; grid_xy [N,2,Ho,Wo] (x,y).  Written as an explicit 4-corner gather."""
    n, c, h, w = img.shape
    x = grid_xy[:, 0]
    y = grid_xy[:, 1]
    x0 = torch.floor(x)
    y0 = torch.floor(y)
    ax = (x - x0).unsqueeze(1)
    ay = (y - y0).unsqueeze(1)
    x0 = x0.long()
    y0 = y0.long()
    flat = img.reshape(n, c, h * w)

    def tap(xi, yi):
        ok = ((xi >= 0) & (xi < w) & (yi >= 0) & (yi < h)).unsqueeze(1).to(img.dtype)
        lin = (yi.clamp(0, h - 1) * w + xi.clamp(0, w - 1)).view(n, 1, -1).expand(n, c, -1)
        return torch.gather(flat, 2, lin).view(n, c, *xi.shape[1:]) * ok

    top = (1 - ax) * tap(x0, y0) + ax * tap(x0 + 1, y0)
    bot = (1 - ax) * tap(x0, y0 + 1) + ax * tap(x0 + 1, y0 + 1)
    return (1 - ay) * top + ay * bot


def warp_hidden_states(last_net_list, grid):
    """tc_stereo.py:159-163: sample level i at the grid, then grid <- 0.5 * bilinear half-size grid."""
    out = []
    for net in last_net_list:
        out.append(sample_bilinear(net, grid))
        grid = 0.5 * F.interpolate(grid, scale_factor=0.5, mode="bilinear", align_corners=True)
    return out


# ----------------------------------------------------------------------------------------------
# a15/a16/a18/a20 stencils
# ----------------------------------------------------------------------------------------------
def disp_gradient_xy(disp):
    """a15 (geo_utils.py:115-132): forward differences on a replicate-padded map -> [N,2,H,W]."""
    p = F.pad(disp, (1, 1, 1, 1), mode="replicate")
    c = p[:, :, 1:-1, 1:-1]
    return torch.cat([p[:, :, 1:-1, 2:] - c, p[:, :, 2:, 1:-1] - c], 1)


_CLOCKWISE = [(-1, -1), (-1, 0), (-1, 1), (0, 1), (1, 1), (1, 0), (1, -1), (0, -1)]   # (dv, du)


def grad_candidates(disp):
    """a16 (geo_utils.py:73-101, level=2): 16 neighbour vectors (dilation 1 then 2, clockwise from
    top-left, disparity ZERO-padded), cross product of vector k with vector (k+2) mod 16, returns
    (-nx/nz, -ny/nz) as [N,2,16,H,W]."""
    n, _, h, w = disp.shape
    vecs = []
    for s in (1, 2):
        p = F.pad(disp, (s, s, s, s))
        for dv, du in _CLOCKWISE:
            dd = p[:, 0, s + s * dv: s + s * dv + h, s + s * du: s + s * du + w] - disp[:, 0]
            vecs.append(torch.stack([torch.full_like(dd, float(s * du)), torch.full_like(dd, float(s * dv)), dd], 1))
    v = torch.stack(vecs, 2)                               # [N,3,16,H,W]
    r = torch.roll(v, shifts=-2, dims=2)
    nx = v[:, 1] * r[:, 2] - v[:, 2] * r[:, 1]
    ny = v[:, 2] * r[:, 0] - v[:, 0] * r[:, 2]
    nz = v[:, 0] * r[:, 1] - v[:, 1] * r[:, 0]
    return torch.stack([-nx / nz, -ny / nz], 1)


def propagate_disparity(grad, disp):
    """a18 part 1 (update.py:259-289).  9 neighbours k = 3v+u.  disparity replicate-padded,
    gradient zero-padded.  cand_k = d_n + gx_n*(1-u) + gy_n*(1-v); matrix[comp*9+k] = |g_c - g_n|."""
    n, _, h, w = disp.shape
    dp = F.pad(disp, (1, 1, 1, 1), mode="replicate")
    gp = F.pad(grad, (1, 1, 1, 1))
    cands, diffs = [], []
    for v in range(3):
        for u in range(3):
            dn = dp[:, 0, v:v + h, u:u + w]
            gn = gp[:, :, v:v + h, u:u + w]
            cands.append(dn + gn[:, 0] * float(1 - u) + gn[:, 1] * float(1 - v))
            diffs.append((grad - gn).abs())
    cand = torch.stack(cands, 1)                           # [N,9,H,W]
    d = torch.stack(diffs, 2)                              # [N,2,9,H,W]
    return cand, d.reshape(n, 18, h, w)


def convex_upsample(flow, mask, factor=4):
    """a20 (tc_stereo.py:75-88): mask channel = k*factor^2 + i*factor + j; softmax over k of the 9
    zero-padded neighbours of factor*flow."""
    n, _, h, w = flow.shape
    m = mask.view(n, 9, factor, factor, h, w)
    m = torch.softmax(m - m.max(dim=1, keepdim=True)[0], dim=1)
    fp = F.pad(factor * flow, (1, 1, 1, 1))
    nb = torch.stack([fp[:, 0, v:v + h, u:u + w] for v in range(3) for u in range(3)], 1)   # [N,9,H,W]
    up = (m * nb.view(n, 9, 1, 1, h, w)).sum(1)            # [N,f,f,H,W]
    return up.permute(0, 3, 1, 4, 2).reshape(n, 1, factor * h, factor * w)


# ----------------------------------------------------------------------------------------------
# a11/a13/a19 recurrent cells, a12 motion encoder, a14 flow head
# ----------------------------------------------------------------------------------------------
def conv_gru(W, name, h, cz, cr, cq, *xs):
    """ConvGRU (update.py:77-87): h <- (1-z) h + z q."""
    x = torch.cat(xs, 1)
    zr = _conv(W, name + ".convzr", torch.cat([h, x], 1))
    z, r = zr.chunk(2, 1)
    z = torch.sigmoid(z + cz)
    r = torch.sigmoid(r + cr)
    q = torch.tanh(_conv(W, name + ".convq", torch.cat([r * h, x], 1)) + cq)
    return (1 - z) * h + z * q


def gru_1x1(W, name, h, x):
    """Lightfuse / HiddenstateUpdater cell (update.py:26-36, 61-67): h <- z h + (1-z) q."""
    zr = _conv(W, name + ".convzr", torch.cat([h, x], 1))
    z, r = zr.chunk(2, 1)
    z = torch.sigmoid(z)
    r = torch.sigmoid(r)
    q = torch.tanh(_conv(W, name + ".convq", torch.cat([r * h, x], 1)))
    return z * h + (1 - z) * q


def hidden_state_update(W, h, delta_disp, name="hiddenstate_update"):
    """a19 (update.py:57-68)."""
    x = _conv(W, name + ".convs.0", delta_disp)
    x = F.leaky_relu(x, 0.01)
    x = _conv(W, name + ".convs.2", x)
    return gru_1x1(W, name, h, x)


def motion_encoder(W, flow, corr, name="update_block.encoder"):
    """a12 (update.py:103-111)."""
    c = F.relu(_conv(W, name + ".convc1", corr))
    c = F.relu(_conv(W, name + ".convc2", c))
    f = F.relu(_conv(W, name + ".convf1", flow))
    f = F.relu(_conv(W, name + ".convf2", f))
    o = F.relu(_conv(W, name + ".conv", torch.cat([c, f], 1)))
    return torch.cat([o, flow], 1)


def _pool2x(x):
    return F.avg_pool2d(x, 3, stride=2, padding=1)          # update.py:114-115 (count_include_pad)


def _interp(x, like):
    return F.interpolate(x, like.shape[2:], mode="bilinear", align_corners=True)   # update.py:122-124


def update_block(W, net, inp, corr=None, flow=None, name="update_block", iter08=True, iter16=True, iter32=True, update=True):
    """a13/a14 (update.py:145-168) for n_gru_layers=3 (the only depth the model's gradient U-Net accepts, update.py:206-210):
    gru32 -> gru16 -> encoder -> gru08 -> flow head; the iter* / update switches serve the slow-fast schedule
    (tc_stereo.py:182-187)."""
    net = list(net)
    if iter32:
        net[2] = conv_gru(W, name + ".gru32", net[2], *inp[2], _pool2x(net[1]))
    if iter16:
        net[1] = conv_gru(W, name + ".gru16", net[1], *inp[1], _pool2x(net[0]), _interp(net[2], net[1]))
    if iter08:
        mf = motion_encoder(W, flow, corr, name + ".encoder")
        net[0] = conv_gru(W, name + ".gru08", net[0], *inp[0], mf, _interp(net[1], net[0]))
    if not update:
        return net
    d = F.relu(_conv(W, name + ".flow_head.conv1", net[0]))
    return net, _conv(W, name + ".flow_head.conv2", d)


# ----------------------------------------------------------------------------------------------
# conv stacks: Conv2x_IN, a17 DispGradPredictor, a18 DispRefine, a9 DisparityCompletor
# ----------------------------------------------------------------------------------------------
def _conv2x_in(W, name, x, rem, use_in):
    """Conv2x_IN(deconv=True, concat=False) (basic_layers.py:38-77): conv1 ALWAYS has
    InstanceNorm + LeakyReLU(0.01); conv2 obeys `use_in`."""
    x = F.conv_transpose2d(x, W[name + ".conv1.conv.weight"], None, stride=2, padding=1)
    x = F.leaky_relu(F.instance_norm(x), 0.01)
    if x.shape != rem.shape:
        x = F.interpolate(x, size=rem.shape[-2:], mode="nearest")
    x = x + rem
    x = F.conv2d(x, W[name + ".conv2.conv.weight"], None, padding=1)
    if use_in:
        x = F.instance_norm(x)
    return F.leaky_relu(x, 0.01)


def disp_grad_predictor(W, grad, disp, clist, name="disp_grad_refine"):
    """a17 (update.py:198-214)."""
    n, _, h, w = disp.shape
    g5 = 5 * grad
    cands = grad_candidates(disp).reshape(n, 32, h, w)
    xg = _conv(W, name + ".conv_grad_stem.2", F.relu(_conv(W, name + ".conv_grad_stem.0", g5)))
    xc = _conv(W, name + ".conv_grad_candidate_stem.2", F.relu(_conv(W, name + ".conv_grad_candidate_stem.0", cands)))
    x4 = F.relu(_conv(W, name + ".conv_4_4.0", torch.cat([xg, xc, clist[0]], 1)))
    x8 = F.relu(_conv(W, name + ".conv_4_8.0", x4, stride=2))
    x8 = F.relu(_conv(W, name + ".conv_8_8.0", torch.cat([x8, clist[1]], 1)))
    x16 = F.relu(_conv(W, name + ".conv_8_16.0", x8, stride=2))
    x16 = F.relu(_conv(W, name + ".conv_16_16.0", torch.cat([x16, clist[2]], 1)))
    x8u = _conv2x_in(W, name + ".conv_16_8", x16, x8, use_in=False)
    x4u = _conv2x_in(W, name + ".conv_8_4", x8u, x4, use_in=False)
    res = _conv(W, name + ".residual_head.2", F.relu(_conv(W, name + ".residual_head.0", x4u)))
    ctx = F.relu(_conv(W, name + ".conv_out.0", x4u))
    return (g5 + res) / 5, ctx


def disp_refine(W, grad, disp, ctx_disp, ctx_grad, want_mask, name="disp_refine"):
    """a18 (update.py:291-305).  want_mask=False is the reference's test_mode=True (mask skipped)."""
    c = _conv(W, name + ".context_compress.2", F.relu(_conv(W, name + ".context_compress.0", torch.cat([ctx_disp, ctx_grad], 1))))
    cand, matrix = propagate_disparity(grad, disp)
    f = _conv(W, name + ".disp_f_stem.2", F.relu(_conv(W, name + ".disp_f_stem.0", torch.cat([cand, matrix], 1))))
    fused = F.relu(_conv(W, name + ".conv_fuse.0", torch.cat([f, c], 1)))
    fused = F.relu(_conv(W, name + ".conv_fuse.2", fused))
    logits = _conv(W, name + ".w_head.2", F.relu(_conv(W, name + ".w_head.0", fused)))
    wgt = torch.softmax(logits - logits.max(dim=1, keepdim=True)[0], dim=1)
    refined = (wgt * cand).sum(1, keepdim=True)
    mask = None
    if want_mask:
        mask = 0.25 * _conv(W, name + ".mask.2", F.relu(_conv(W, name + ".mask.0", fused)))
    return refined, mask


def _cin(W, name, x, stride=1):
    """conv -> InstanceNorm -> ReLU -> conv blocks of DisparityCompletor (update.py:325-367)."""
    x = F.relu(F.instance_norm(_conv(W, name + ".0", x, stride=stride)))
    return _conv(W, name + ".3", x)


def _mlp1x1(W, name, x):
    return _conv(W, name + ".2", F.relu(_conv(W, name + ".0", x)))


def disparity_completor(W, disp, cost, mask, ctx, name="disp_completor"):
    """a9 (update.py:369-399)."""
    m = mask - 0.5
    d = disp / 10
    x4d = _mlp1x1(W, name + ".conv_disp_fuse", torch.cat([
        _mlp1x1(W, name + ".conv_disp_stem", d), _mlp1x1(W, name + ".conv_cost_stem", cost),
        _mlp1x1(W, name + ".conv_mask_stem", m)], 1))
    x4 = _cin(W, name + ".conv_4_4", torch.cat([x4d, ctx[0]], 1))
    x8 = _cin(W, name + ".conv_4_8", x4, stride=2)
    x8 = _cin(W, name + ".conv_8_8", torch.cat([x8, ctx[1]], 1))
    x16 = _cin(W, name + ".conv_8_16", x8, stride=2)
    x16o = _cin(W, name + ".conv_16_16", torch.cat([x16, ctx[2]], 1))
    x8o = _conv2x_in(W, name + ".conv_16_8", x16o, x8, use_in=True)
    x4o = _conv2x_in(W, name + ".conv_8_4", x8o, x4, use_in=True)
    mono = _mlp1x1(W, name + ".disp_head", x4o)
    wgt = torch.sigmoid(_mlp1x1(W, name + ".w_head", x4o))
    completed = (wgt * d + (1 - wgt) * mono) * 10
    nets = [_cin(W, name + ".conv_out4_disp", torch.cat([x4o, ctx[0]], 1)),
            _cin(W, name + ".conv_out8_disp", torch.cat([x8o, ctx[1]], 1)),
            _cin(W, name + ".conv_out16_disp", torch.cat([x16o, ctx[2]], 1))]
    return completed, mono * 10, wgt, nets


# ----------------------------------------------------------------------------------------------
# feature / context extractor (stays PyTorch in the product too)       (core/extractor.py)
# ----------------------------------------------------------------------------------------------
def _res_block(W, name, x, kind, stride):
    y = F.relu(_norm(W, name + ".norm1", _conv(W, name + ".conv1", x, stride=stride), kind))
    y = F.relu(_norm(W, name + ".norm2", _conv(W, name + ".conv2", y), kind))
    if (name + ".downsample.0.weight") in W:
        x = _norm(W, name + ".downsample.1", _conv(W, name + ".downsample.0", x, stride=stride, padding=0), kind)
    return F.relu(x + y)


def _layer(W, name, x, kind, stride):
    return _res_block(W, name + ".1", _res_block(W, name + ".0", x, kind, stride), kind, 1)


def context_encoder(W, x, kind, n_heads=2, name="cnet", dual=True):
    """MultiBasicEncoder.forward(num_layers=3) (extractor.py:270-296); `dual` = dual_inp (the batch
    holds [left; right] and only the left half feeds the context heads).  Strides are hard-coded:
    outputs at 1/4, 1/8, 1/16 of the input."""
    x = F.relu(_norm(W, name + ".norm1", _conv(W, name + ".conv1", x), kind))
    x = _layer(W, name + ".layer1", x, kind, 1)
    x = _layer(W, name + ".layer2", x, kind, 2)
    x = _layer(W, name + ".layer3", x, kind, 2)
    trunk = x
    if dual:
        x = x[: x.shape[0] // 2]
    o08 = [_conv(W, f"{name}.outputs08.{i}.1", _res_block(W, f"{name}.outputs08.{i}.0", x, kind, 1)) for i in range(n_heads)]
    y = _layer(W, name + ".layer4", x, kind, 2)
    o16 = [_conv(W, f"{name}.outputs16.{i}.1", _res_block(W, f"{name}.outputs16.{i}.0", y, kind, 1)) for i in range(n_heads)]
    z = _layer(W, name + ".layer5", y, kind, 2)
    o32 = [_conv(W, f"{name}.outputs32.{i}", z) for i in range(n_heads)]
    return [o08, o16, o32], trunk


def feature_encoder(W, x, downsample, name="fnet"):
    """BasicEncoder (extractor.py:119-192) with norm_fn='instance' (tc_stereo.py:45)."""
    x = F.relu(F.instance_norm(_conv(W, name + ".conv1", x, stride=1 + (downsample > 2))))
    x = _layer(W, name + ".layer1", x, "instance", 1)
    x = _layer(W, name + ".layer2", x, "instance", 1 + (downsample > 1))
    x = _layer(W, name + ".layer3", x, "instance", 1 + (downsample > 0))
    return _conv(W, name + ".conv2", x, padding=0)


# ----------------------------------------------------------------------------------------------
# a1: TCStereo.forward, test_mode=True                                 (core/tc_stereo.py:96-244)
# ----------------------------------------------------------------------------------------------
@torch.no_grad()
def tc_stereo_forward(W, image1, image2, iters=12, params=None, args=None, trace=None):
    """Returns the reference's test-mode dict {'flow','flow_q','net_list','fmap1'}.  `trace`, when a
    dict, receives intermediate tensors (used by per-stage parity tests)."""
    args = args or default_args()
    dt = image1.dtype
    scale = 1.0 / (2 ** args.n_downsample)
    im1 = 2 * (image1 / 255.0) - 1.0
    im2 = 2 * (image2 / 255.0) - 1.0

    if args.shared_backbone:
        cnet_list, trunk = context_encoder(W, torch.cat([im1, im2], 0), args.context_norm)
        f = _conv(W, "conv2.1", _res_block(W, "conv2.0", trunk, "instance", 1))
        fmap1, fmap2 = f[: f.shape[0] // 2], f[f.shape[0] // 2:]
    else:
        cnet_list, _ = context_encoder(W, im1, args.context_norm, dual=False)
        f = feature_encoder(W, torch.cat([im1, im2], 0), args.n_downsample)
        fmap1, fmap2 = f[: f.shape[0] // 2], f[f.shape[0] // 2:]

    vol = corr_volume(fmap1, fmap2)
    pyr = corr_pyramid(vol, args.corr_levels)

    last_nets = None
    if params is not None:
        K = params["K"].to(dt)
        K_s = K * torch.tensor([scale, scale, 1.0], dtype=dt).view(1, 3, 1)
        K_si = torch.linalg.inv(K_s)
        T, T_prev = params["T"].to(dt), params["previous_T"].to(dt)
        rel = torch.matmul(T, torch.linalg.inv(T_prev))            # geo_utils.py:148-155
        base = params["baseline"].to(dt)
        last_nets = params["last_net_list"]
        sparse_disp, warped_f, sparse_mask = forward_warp(-params["last_disp"], params["fmap1"], rel, K_s, K_si, base)
        cost = (F.normalize(fmap1, dim=1) * F.normalize(warped_f, dim=1)).sum(1, keepdim=True) * sparse_mask
    else:
        sparse_disp, cost, sparse_mask = argmax_disp(masked_cost_volume(vol))

    ctx = [torch.relu(x[1]) for x in cnet_list]
    grad_ctx = [_conv(W, f"context_zqr_convs_grad.{i}", c) for i, c in enumerate(ctx)]
    inp = [list(_conv(W, f"context_zqr_convs.{i}", c).chunk(3, 1)) for i, c in enumerate(ctx)]
    net = [x[0] for x in cnet_list]

    disp_init, _, _, net = disparity_completor(W, sparse_disp, cost, sparse_mask, net)

    if last_nets is None:
        warped = [torch.zeros_like(x) for x in net]
    else:
        back = torch.matmul(T_prev, torch.linalg.inv(T))
        warped = warp_hidden_states(last_nets, backward_grid(disp_init, back, K_s, K_si, base))

    net = [gru_1x1(W, f"previous_current_hideen_fuse.{i}", torch.tanh(n_), w_) for i, (n_, w_) in enumerate(zip(net, warped))]

    n, _, h4, w4 = fmap1.shape
    coords0 = _xy_grid(n, h4, w4, fmap1)[:, :1]
    coords1 = coords0 - disp_init
    if trace is not None:
        trace.update(fmap1=fmap1, fmap2=fmap2, sparse_disp=sparse_disp, cost=cost, sparse_mask=sparse_mask,
                     disp_init=disp_init, net0=[t.clone() for t in net], inp=inp, grad_ctx=grad_ctx, pyr=pyr,
                     iters=[])

    up_mask = None
    refined = None
    for itr in range(iters):
        corr = corr_lookup(pyr, coords1, args.corr_radius)
        flow_x = coords1 - coords0
        if args.slow_fast_gru:                                     # tc_stereo.py:182-185: extra coarse-level sweeps
            net = update_block(W, net, inp, iter32=True, iter16=False, iter08=False, update=False)
            net = update_block(W, net, inp, iter32=True, iter16=True, iter08=False, update=False)
        net, delta = update_block(W, net, inp, corr, flow_x)
        coords1 = coords1 + delta
        disp_q = coords0 - coords1
        g = disp_gradient_xy(disp_q)
        g, gctx = disp_grad_predictor(W, g, disp_q, grad_ctx)
        last = itr == iters - 1
        refined, up_mask = disp_refine(W, g, disp_q, net[0], gctx, want_mask=last)
        net = [hidden_state_update(W, net[0], refined - disp_q), net[1], net[2]]
        coords1 = coords0 - refined
        if trace is not None:
            trace["iters"].append(dict(corr=corr, delta=delta, disp_q=disp_q, grad=g, refined=refined,
                                       net=[t.clone() for t in net]))

    up = convex_upsample(-refined, up_mask, 2 ** args.n_downsample)
    return {"flow": up.clamp(max=0), "flow_q": (-refined).clamp(max=0), "net_list": net, "fmap1": fmap1}
