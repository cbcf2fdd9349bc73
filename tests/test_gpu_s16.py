"""GPU parity of the pre-split ("S16") activation path — tcs_conv2d_s16 and its glue kernels — through the C ABI.

The S16 kernels have no counterpart in the reference: they are an internal representation of the same fp32 tensors.
Each test therefore checks against fp64 PyTorch arithmetic on the CPU (convolutions, pooling, normalisation) at the
tolerances of the fp32-tensor kernels they replace (tests/test_gpu_parity.py): <= 2e-5 for single convolutions,
<= 1e-5 for the memory-bound glue.  The blocks built from them are pinned to the reference by the module goldens in
test_gpu_parity.py (ub_*, dg_*, dr_*, hu_*), which run through these kernels."""
import pytest
import torch
import torch.nn.functional as F

from conftest import maxdiff

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from tcs_mi355 import native
    native.lib()
    return torch.device("cuda:0")


def D(x, dev):
    return x.to(dev).contiguous()


def test_s16_roundtrip_and_border(dev):
    from tcs_mi355 import s16
    gen = torch.Generator().manual_seed(0)
    x = torch.randn(2, 27, 9, 21, generator=gen)
    x[:, 3] *= 1e-5            # lo halves in fp16-subnormal territory
    x[:, 4] *= 3e3
    x[0, 5, 0, 0] = 7e4        # beyond fp16 range: saturates at 65504 (documented domain of the split)
    t = s16.to_s16(D(x, dev))
    assert t.G == 4 and tuple(t.data.shape) == (2, 4, 2, 11, 23, 8)
    back = t.float().cpu()
    ref = x.clamp(-65504, 65504)
    # x = hi + lo to 2^-22 relative, with the fp16 subnormal step (2^-24) / 2 as the absolute floor for tiny values
    assert bool(((back - ref).abs() <= 2.0 ** -22 * ref.abs() + 2.0 ** -25).all())
    d = t.data.float().cpu()
    assert float(d[:, :, :, 0].abs().max()) == 0 and float(d[:, :, :, -1].abs().max()) == 0       # zero border rows
    assert float(d[:, :, :, :, 0].abs().max()) == 0 and float(d[:, :, :, :, -1].abs().max()) == 0  # zero border columns
    assert float(d[:, 3, :, :, :, 3:].abs().max()) == 0                                             # padding channels 27..31
    # virtual concat: a second tensor into the upper groups of a wider buffer
    wide = s16.zeros(2, 64, 9, 21, dev)
    y = torch.randn(2, 32, 9, 21, generator=gen)
    s16.to_s16(D(x, dev), out=wide, group_offset=0)
    s16.to_s16(D(y, dev), out=wide, group_offset=4)
    assert maxdiff(s16.from_s16(wide, 32, group_offset=4), y) <= 1e-6


CONV_CASES = [
    dict(B=1, cins=(128, 128, 128), cout=256, k=3, H=12, W=40),       # gru08-like, 3 virtual sources
    dict(B=2, cins=(32, 64, 64), cout=64, k=3, H=9, W=37),            # conv_4_4: mixed source widths, ragged grid, batch 2
    dict(B=1, cins=(64, 64), cout=127, k=3, H=8, W=24),               # encoder.conv: 127 outputs (masked partial store)
    dict(B=1, cins=(27,), cout=96, k=1, H=7, W=65),                   # disp_f_stem: 27 inputs
    dict(B=1, cins=(256,), cout=1, k=3, H=6, W=32),                   # flow head conv2: one output channel
    dict(B=1, cins=(128,), cout=9, k=1, H=11, W=33),                  # w_head[2]
    dict(B=1, cins=(128, 64), cout=96, k=3, H=17, W=35),              # odd grid, 96 outputs (3 tiles)
    dict(B=1, cins=(128, 64), cout=256, k=1, H=9, W=35),              # a 1x1 gate convolution (Lightfuse-like): 64-channel tiles, 2/4 k-steps per stage
    dict(B=1, cins=(64,), cout=96, k=3, H=16, W=34, stride=2),        # conv_4_8
    dict(B=1, cins=(96,), cout=128, k=3, H=15, W=33, stride=2),       # stride 2 on an odd grid
]


@pytest.mark.parametrize("cfg", CONV_CASES)
def test_conv2d_s16_vs_torch(dev, cfg):
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(cfg["cout"] + cfg["k"] + cfg["H"])
    cin, stride = sum(cfg["cins"]), cfg.get("stride", 1)
    w = torch.randn(cfg["cout"], cin, cfg["k"], cfg["k"], generator=gen) * (2.0 / (cin * cfg["k"] ** 2)) ** 0.5
    b = torch.randn(cfg["cout"], generator=gen) * 0.1
    xs = [torch.randn(cfg["B"], c, cfg["H"], cfg["W"], generator=gen) for c in cfg["cins"]]
    ref = F.conv2d(torch.cat(xs, 1).double(), w.double(), b.double(), padding=cfg["k"] // 2, stride=stride)
    add = torch.randn(ref.shape, generator=gen)
    pc = ops.pack_conv(D(w, dev), D(b, dev), "f16x3")
    xs16 = [s16.to_s16(D(x, dev)) for x in xs]
    tiles = [0]
    # every tile configuration the library instantiates (launch_s16_cfg): MT*1000 + ROWS*100 + KSTEPS*10 + NSTAGE, +10000 row split,
    # +100000*CSPLIT block -> XCD mapping; 64-channel tiles (MT = 2) need an even number of 32-channel tiles
    if cfg["k"] == 3 and stride == 1:
        tiles += [1411, 1412, 1413, 1811, 1512, 101812, 201412]
        tiles += [21812, 21412, 21411, 121812]                          # two rows per wave (RS digit 2)
        tiles += [2411, 2412, 2413, 2512, 102812, 12412, 12413, 22812, 22412, 122812] if cfg["cout"] % 64 == 0 else []
    elif cfg["k"] == 1 and stride == 1:
        ksteps = [(c + 15) // 16 for c in cfg["cins"]]
        tiles += [1412]
        for kst in (2, 4):
            if all(k % kst == 0 for k in ksteps):
                tiles += [1400 + 10 * kst + 2, 101400 + 10 * kst + 2] + ([1423] if kst == 2 else [])
                tiles += ([2400 + 10 * kst + 2] + ([2423] if kst == 2 else [])) if cfg["cout"] % 64 == 0 else []
        tiles += [2412] if cfg["cout"] % 64 == 0 else []
    for tc in tiles:
        o16, o32 = s16.conv2d(pc, xs16, want32=True, stride=stride, tile_cfg=tc)
        assert maxdiff(o32, ref) <= 2e-5, tc
        out = s16.zeros(cfg["B"], cfg["cout"], ref.shape[2], ref.shape[3], dev)
        s16.conv2d(pc, xs16, act="relu", addend=D(add, dev), post_scale=0.25, out16=out, stride=stride, tile_cfg=tc)
        assert maxdiff(out.float(), 0.25 * torch.relu(ref + add.double())) <= 2e-5, tc
    for act, fn in (("leaky", lambda t: F.leaky_relu(t, 0.01)), ("tanh", torch.tanh), ("sigmoid", torch.sigmoid)):
        assert maxdiff(s16.conv2d(pc, xs16, act=act, stride=stride)[0].float(), fn(ref)) <= 2e-5


def test_conv2d_s16_residual_block_pieces(dev):
    """The residual blocks of the feature extractor on S16 tensors (extractor.py:37-57): relu(skip + relu(conv(y))) with the
    skip read as an S16 addend, the 1x1 stride-2 shortcut, and the instance-norm variant relu(skip + relu(IN(z)))."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(17)
    for B, C_, H, W in ((1, 64, 11, 37), (2, 96, 8, 24)):
        w = torch.randn(C_, C_, 3, 3, generator=gen) * (2.0 / (9 * C_)) ** 0.5
        b = torch.randn(C_, generator=gen) * 0.1
        y, skip = torch.randn(B, C_, H, W, generator=gen), torch.randn(B, C_, H, W, generator=gen)
        ref = torch.relu(skip.double() + torch.relu(F.conv2d(y.double(), w.double(), b.double(), padding=1)))
        pc = ops.pack_conv(D(w, dev), D(b, dev), "f16x3")
        for tc in (0, 1412, 101812):
            got, _ = s16.conv2d(pc, [s16.to_s16(D(y, dev))], act="relu_add_relu", addend16=s16.to_s16(D(skip, dev)), tile_cfg=tc)
            assert maxdiff(got.float(), ref) <= 2e-5, (C_, tc)
        # plain S16 addend with a linear epilogue
        got, _ = s16.conv2d(pc, [s16.to_s16(D(y, dev))], act="relu", addend16=s16.to_s16(D(skip, dev)))
        assert maxdiff(got.float(), torch.relu(F.conv2d(y.double(), w.double(), b.double(), padding=1) + skip.double())) <= 2e-5
        # downsample shortcut: 1x1, stride 2, ragged grid
        wd = torch.randn(C_ + 32, C_, 1, 1, generator=gen) * (1.0 / C_) ** 0.5
        bd = torch.randn(C_ + 32, generator=gen) * 0.1
        got, _ = s16.conv2d(ops.pack_conv(D(wd, dev), D(bd, dev), "f16x3"), [s16.to_s16(D(y, dev))], stride=2)
        assert (got.H, got.W) == ((H + 1) // 2, (W + 1) // 2)
        assert maxdiff(got.float(), F.conv2d(y.double(), wd.double(), bd.double(), stride=2)) <= 2e-5
        # instance-norm residual epilogue
        z = torch.randn(B, C_, H, W, generator=gen) * 2 + 0.5
        refn = torch.relu(skip.double() + torch.relu(F.instance_norm(z.double(), eps=1e-5)))
        z16 = s16.to_s16(D(z, dev))
        s16.instance_norm(z16, act="relu_add_relu", addend=s16.to_s16(D(skip, dev)), out=z16)
        assert maxdiff(z16.float(), refn) <= 1e-5


def test_conv2d_s16_k_split_accumulation(dev):
    """A layer evaluated as partial convolutions over slices of its input channels (tcs_mi355.h, addend_ctot): the first
    partial writes bias + conv_a(a) as fp32, the second adds conv_b(b) in place or applies the real epilogue with the sum
    as addend — for GRU_ZR with cz / cr being channel slices of ONE [B, 2*hidden, H, W] tensor (batch stride 2*hidden)."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(23)
    B, hid, H, W = 2, 64, 9, 37
    h, x = torch.tanh(torch.randn(B, hid, H, W, generator=gen)), torch.randn(B, 96, H, W, generator=gen)
    wzr = torch.randn(2 * hid, hid + 96, 3, 3, generator=gen) * 0.03
    bzr, ctx = torch.randn(2 * hid, generator=gen) * 0.1, torch.randn(B, 2 * hid, H, W, generator=gen) * 0.3
    pre = F.conv2d(torch.cat([h, x], 1).double(), wzr.double(), bzr.double(), padding=1) + ctx.double()
    z_ref, rh_ref = torch.sigmoid(pre[:, :hid]), torch.sigmoid(pre[:, hid:]) * h.double()
    h16, x16 = s16.to_s16(D(h, dev)), s16.to_s16(D(x, dev))
    part = torch.empty(B, 2 * hid, H, W, device=dev)
    # partial 1: the hidden-state share + bias + context -> fp32 sum
    s16.conv2d(ops.pack_conv(D(wzr[:, :hid].contiguous(), dev), D(bzr, dev), "f16x3"), [h16], addend=D(ctx, dev), out32=part)
    assert maxdiff(part, F.conv2d(h.double(), wzr[:, :hid].double(), bzr.double(), padding=1) + ctx.double()) <= 2e-5
    # final: the input share with the gate epilogue, addends = channel slices of the sum
    pc_x = ops.pack_conv(D(wzr[:, hid:].contiguous(), dev), None, "f16x3")
    z, rh = s16.gru_gates(pc_x, [x16], h16, part[:, :hid], part[:, hid:], addend_ctot=2 * hid)
    assert maxdiff(z, z_ref) <= 1e-5 and maxdiff(rh.float(), rh_ref) <= 1e-5
    # LINEAR: in-place accumulation of a second partial, and a sliced addend
    acc = part.clone()
    s16.conv2d(pc_x, [x16], addend=acc, out32=acc)
    assert maxdiff(acc, pre) <= 2e-5
    o, _ = s16.conv2d(ops.pack_conv(D(wzr[:hid, hid:].contiguous(), dev), None, "f16x3"), [x16], act="relu", addend=part[:, hid:],
                      addend_ctot=2 * hid)
    assert maxdiff(o.float(), torch.relu(F.conv2d(x.double(), wzr[:hid, hid:].double(), padding=1) + part[:, hid:].double().cpu())) <= 2e-5
    with pytest.raises(ValueError):
        s16.gru_gates(pc_x, [x16], h16, part[:, :hid].contiguous(), part[:, hid:], addend_ctot=2 * hid)   # not a slice of a wider tensor


def test_conv2d_s16_partial_store_keeps_foreign_channel(dev):
    """encoder.conv writes 127 channels into the 128-channel motion buffer whose channel 127 belongs to the blend kernel."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(3)
    w = torch.randn(127, 128, 3, 3, generator=gen) * 0.03
    x = torch.randn(1, 128, 10, 40, generator=gen)
    flow = torch.randn(1, 1, 10, 40, generator=gen) * 20
    buf = s16.zeros(1, 128, 10, 40, dev)
    s16.set_channel(D(flow, dev), buf, 127)
    s16.conv2d(ops.pack_conv(D(w, dev), None, "f16x3"), [s16.to_s16(D(x, dev))], act="relu", out16=buf)
    got = buf.float()
    assert maxdiff(got[:, :127], torch.relu(F.conv2d(x.double(), w.double(), padding=1))) <= 2e-5
    assert maxdiff(got[:, 127:], flow) <= 2.0 ** -21 * 20 * 4


def test_conv2d_s16_from_fp32_sources(dev):
    """The fp32-input kernels with an S16 epilogue (convc1 36->64 1x1, convf1 1->64 7x7, grad stem 2->32 3x3, fp32-MFMA 32->64)."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(5)
    for cin, cout, k, math in ((36, 64, 1, "f16x3"), (1, 64, 7, "f16x3"), (1, 64, 1, "f16x3"), (2, 32, 3, "f16x3"), (32, 64, 3, "f32")):
        w = torch.randn(cout, cin, k, k, generator=gen) * (2.0 / (cin * k * k)) ** 0.5
        b = torch.randn(cout, generator=gen) * 0.1
        x = torch.randn(2, cin, 9, 37, generator=gen)
        out = s16.zeros(2, cout, 9, 37, dev)
        ops.conv2d(ops.pack_conv(D(w, dev), D(b, dev), math), [D(x, dev)], act="relu", out16=out)
        assert maxdiff(out.float(), torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=k // 2))) <= 2e-5, (cin, cout, k)


def test_gru_s16_vs_torch(dev):
    """Both GRU epilogues (update.py:77-87 and :26-36) on S16 tensors, in-place state update, 3x3 and 1x1 cells."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(9)
    for k, keep_z, cx in ((3, False, (128, 128)), (1, True, (64,))):
        hid, H, W = 128, 10, 37
        cin = hid + sum(cx)
        wzr = torch.randn(2 * hid, cin, k, k, generator=gen) * (1.0 / (cin * k * k)) ** 0.5
        wq = torch.randn(hid, cin, k, k, generator=gen) * (1.0 / (cin * k * k)) ** 0.5
        bzr, bq = torch.randn(2 * hid, generator=gen) * 0.1, torch.randn(hid, generator=gen) * 0.1
        h = torch.tanh(torch.randn(1, hid, H, W, generator=gen))
        xs = [torch.randn(1, c, H, W, generator=gen) for c in cx]
        cz, cr, cq = (torch.randn(1, hid, H, W, generator=gen) * 0.3 for _ in range(3))
        hx = torch.cat([h, *xs], 1).double()
        zr = F.conv2d(hx, wzr.double(), bzr.double(), padding=k // 2)
        z, r = torch.sigmoid(zr[:, :hid] + cz), torch.sigmoid(zr[:, hid:] + cr)
        q = torch.tanh(F.conv2d(torch.cat([r * h, *xs], 1).double(), wq.double(), bq.double(), padding=k // 2) + cq)
        ref = z * h + (1 - z) * q if keep_z else (1 - z) * h + z * q
        h16, xs16 = s16.to_s16(D(h, dev)), [s16.to_s16(D(x, dev)) for x in xs]
        zz, rh = s16.gru_gates(ops.pack_conv(D(wzr, dev), D(bzr, dev), "f16x3"), [h16, *xs16], h16, D(cz, dev), D(cr, dev))
        assert maxdiff(zz, z) <= 1e-5 and maxdiff(rh.float(), r * h) <= 1e-5
        out = s16.gru_update(ops.pack_conv(D(wq, dev), D(bq, dev), "f16x3"), [rh, *xs16], h16, zz, D(cq, dev), keep_z=keep_z, out=h16)
        assert out is h16 and maxdiff(h16.float(), ref) <= 1e-5, k


def test_deconv_s16_vs_torch(dev):
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(11)
    for cin, cout, H, W in ((128, 96, 6, 20), (96, 64, 9, 17)):
        wt = torch.randn(cin, cout, 4, 4, generator=gen) * (1.0 / (cin * 4)) ** 0.5
        x = torch.randn(2, cin, H, W, generator=gen)
        ref = F.conv_transpose2d(x.double(), wt.double(), stride=2, padding=1)
        got = s16.deconv4x4s2(ops.pack_deconv4x4s2(D(wt, dev)), [s16.to_s16(D(x, dev))])
        assert (got.H, got.W) == (2 * H, 2 * W) and maxdiff(got.float(), ref) <= 2e-5


def test_glue_s16_vs_torch(dev):
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(13)
    x = torch.randn(2, 24, 13, 17, generator=gen)
    x16 = s16.to_s16(D(x, dev))
    assert maxdiff(s16.avgpool3s2(x16).float(), F.avg_pool2d(x, 3, stride=2, padding=1)) <= 1e-6
    assert maxdiff(s16.resize_bilinear(x16, 26, 34).float(), F.interpolate(x, (26, 34), mode="bilinear", align_corners=True)) <= 1e-5
    # InstanceNorm + LeakyReLU + skip at the two shapes of the loop's up-blocks (one / several plane slices)
    for C_, H, W in ((96, 12, 20), (64, 120, 160)):
        y = torch.randn(1, C_, H, W, generator=gen) * 3 + 1.5
        rem = torch.randn(1, C_, H, W, generator=gen)
        ref = F.leaky_relu(F.instance_norm(y.double(), eps=1e-5), 0.01) + rem.double()
        y16 = s16.to_s16(D(y, dev))
        got = s16.instance_norm(y16, act="leaky", addend=s16.to_s16(D(rem, dev)), out=y16)
        assert got is y16 and maxdiff(y16.float(), ref) <= 1e-5, (C_, H, W)
        assert maxdiff(s16.instance_norm(s16.to_s16(D(y, dev))).float(), F.instance_norm(y.double(), eps=1e-5)) <= 1e-5
    # candidate stencil: the S16 stem input and the fp32 candidates against the fp32 kernel (golden-pinned in test_gpu_parity)
    g, d = torch.randn(2, 2, 9, 21, generator=gen), torch.rand(2, 1, 9, 21, generator=gen) * 40
    f27 = ops.propagate_disparity(D(g, dev), D(d, dev))
    o16, c9 = s16.propagate_disparity(D(g, dev), D(d, dev))
    assert o16.G == 4 and maxdiff(o16.float(), f27) <= 2.0 ** -21 * 64 and maxdiff(c9, f27[:, :9]) == 0
    # blend: identical fp32 outputs to the fp32 entry point, plus the flow channel in the S16 buffer
    logits = torch.randn(2, 9, 9, 21, generator=gen)
    co, fx = torch.empty(2, 1, 9, 21, device=dev), torch.empty(2, 1, 9, 21, device=dev)
    ref_r, ref_d = ops.softmax_blend(D(logits, dev), f27, disp_q=D(d, dev), want_delta=True)
    buf = s16.zeros(2, 128, 9, 21, dev)
    r, dl = s16.softmax_blend(D(logits, dev), c9, D(d, dev), co, fx, flow_x_s16=buf, flow_x_channel=127)
    assert maxdiff(r, ref_r) == 0 and maxdiff(dl, ref_d) == 0
    xs = torch.arange(21, dtype=torch.float32).view(1, 1, 1, 21)
    assert maxdiff(co, xs - r.cpu()) <= 1e-6 and maxdiff(fx, co.cpu() - xs) <= 1e-5
    assert maxdiff(s16.from_s16(buf, 128)[:, 127:], fx) <= 2.0 ** -21 * 64
    assert float(s16.from_s16(buf, 128)[:, :127].abs().max()) == 0


def test_conv1x1_blend_equals_conv_then_blend(dev):
    """TCS_EPI_BLEND9: w_head's last 1x1 convolution with the softmax blend as its epilogue must give exactly what the two
    launches give (same logits, same formula): refined, delta, coords1, flow_x and the flow channel of the S16 buffer."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(31)
    for B, H, W in ((1, 11, 45), (2, 9, 70)):
        w = torch.randn(9, 128, 1, 1, generator=gen) * 0.1
        b = torch.randn(9, generator=gen) * 0.1
        x16 = s16.to_s16(D(torch.randn(B, 128, H, W, generator=gen), dev))
        cand = D(torch.rand(B, 9, H, W, generator=gen) * 50, dev)
        disp = D(torch.rand(B, 1, H, W, generator=gen) * 50, dev)
        pc = ops.pack_conv(D(w, dev), D(b, dev), "f16x3")
        outs = []
        for fused in (False, True):
            c1, fx = torch.empty(B, 1, H, W, device=dev), torch.empty(B, 1, H, W, device=dev)
            buf = s16.zeros(B, 128, H, W, dev)
            if fused:
                r, dl = s16.conv1x1_blend(pc, [x16], cand, disp, c1, fx, flow_x_s16=buf, flow_x_channel=127)
            else:
                logits = s16.conv2d(pc, [x16], want32=True)[1]
                r, dl = s16.softmax_blend(logits, cand, disp, c1, fx, flow_x_s16=buf, flow_x_channel=127)
            outs.append((r, dl, c1, fx, buf.data.clone()))
        for a_, b_ in zip(*outs):
            assert bool((a_ == b_).all())
        # the optional pyramid warm-up (blend_warm_pyr: the lane also issues the next lookup's tap loads and discards them) changes
        # no output — also with candidates that put the new coordinate far outside the row, NaN included (addresses are clamped)
        if W % 8 == 0 or True:
            pyr = ops.corr_build(D(torch.randn(B, 16, H, W, generator=gen), dev), D(torch.randn(B, 16, H, W, generator=gen), dev))
            wild = cand.clone()
            wild[:, :, 0, :4] = 1e9
            wild[:, :, 1, :4] = -1e9
            wild[:, :, 2, :2] = float("nan")
            for cnd in (cand, wild):
                res = []
                for warm in (None, pyr):
                    c1, fx = torch.empty(B, 1, H, W, device=dev), torch.empty(B, 1, H, W, device=dev)
                    r, dl = s16.conv1x1_blend(pc, [x16], cnd, disp, c1, fx, warm_pyramid=warm)
                    res.append((r, dl, c1, fx))
                torch.cuda.synchronize()
                for a_, b_ in zip(*res):
                    assert bool(((a_ == b_) | (a_.isnan() & b_.isnan())).all())
        # and against fp64 arithmetic
        logits = F.conv2d(x16.float().double().cpu(), w.double(), b.double())
        ref = (torch.softmax(logits, 1) * cand.double().cpu()).sum(1, keepdim=True)
        assert maxdiff(outs[1][0], ref) <= 2e-4


def test_hidden_update_fused_vs_torch(dev):
    """tcs_hidden_update_s16 (HiddenstateUpdater, update.py:57-68, one launch) against fp64 PyTorch, ragged grid, batch 2."""
    from tcs_mi355 import s16
    gen = torch.Generator().manual_seed(21)
    B, H, W = 2, 11, 45
    h = torch.tanh(torch.randn(B, 128, H, W, generator=gen))
    delta = torch.randn(B, 1, H, W, generator=gen) * 2
    w1, b1 = torch.randn(64, 1, 1, 1, generator=gen), torch.randn(64, generator=gen) * 0.1
    w2, b2 = torch.randn(64, 64, 1, 1, generator=gen) * 0.15, torch.randn(64, generator=gen) * 0.1
    wzr, bzr = torch.randn(256, 192, 1, 1, generator=gen) * 0.08, torch.randn(256, generator=gen) * 0.1
    wq, bq = torch.randn(128, 192, 1, 1, generator=gen) * 0.08, torch.randn(128, generator=gen) * 0.1
    x = F.conv2d(F.leaky_relu(F.conv2d(delta.double(), w1.double(), b1.double()), 0.01), w2.double(), b2.double())
    zr = torch.sigmoid(F.conv2d(torch.cat([h.double(), x], 1), wzr.double(), bzr.double()))
    z, r = zr[:, :128], zr[:, 128:]
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), wq.double(), bq.double()))
    ref = z * h + (1 - z) * q
    h16 = s16.to_s16(D(h, dev))
    out = s16.hidden_update(h16, D(delta, dev), D(w1.reshape(64), dev), D(b1, dev), s16.pack_frags(D(w2, dev), D(b2, dev), 64),
                            s16.pack_frags(D(wzr, dev), D(bzr, dev), 0), s16.pack_frags(D(wq, dev), D(bq, dev), 0))
    assert out is h16 and maxdiff(h16.float(), ref) <= 1e-5
    d = h16.data.float().cpu()
    assert float(d[:, :, :, 0].abs().max()) == 0 and float(d[:, :, :, :, -1].abs().max()) == 0        # border untouched


def test_hidden_update_fused_saturating_gates(dev):
    """Gate pre-activations far beyond +-88.7 (where exp overflows in fp32): torch.sigmoid (update.py:62-63) saturates at 0 / 1 and so must
    the fused kernel's hardware exp / rcp path — it returned NaN there (stored as 65504 into net08, domain flag 0x2) until the
    argument was clamped.  delta of +-100 and +-1000 px drives |pre| into the hundreds / thousands; every activation stays inside the
    S16 domain, so the flag word must stay clean."""
    from tcs_mi355 import s16
    gen = torch.Generator().manual_seed(22)
    B, H, W = 1, 8, 64
    h = torch.tanh(torch.randn(B, 128, H, W, generator=gen))
    delta = torch.randn(B, 1, H, W, generator=gen) * 2
    delta[0, 0, 0, :] = 100.0
    delta[0, 0, 1, :] = -100.0
    delta[0, 0, 2, :] = 1000.0
    delta[0, 0, 3, :] = -1000.0
    w1, b1 = torch.randn(64, 1, 1, 1, generator=gen), torch.randn(64, generator=gen) * 0.1
    w2, b2 = torch.randn(64, 64, 1, 1, generator=gen) * 0.15, torch.randn(64, generator=gen) * 0.1
    wzr, bzr = torch.randn(256, 192, 1, 1, generator=gen) * 0.08, torch.randn(256, generator=gen) * 0.1
    wq, bq = torch.randn(128, 192, 1, 1, generator=gen) * 0.08, torch.randn(128, generator=gen) * 0.1
    x = F.conv2d(F.leaky_relu(F.conv2d(delta.double(), w1.double(), b1.double()), 0.01), w2.double(), b2.double())
    pre = F.conv2d(torch.cat([h.double(), x], 1), wzr.double(), bzr.double())
    assert float(pre.min()) < -200 and float(pre.max()) > 200 and float(x.abs().max()) < 6.0e4        # the case is what it claims to be
    zr = torch.sigmoid(pre)
    z, r = zr[:, :128], zr[:, 128:]
    q = torch.tanh(F.conv2d(torch.cat([r * h, x], 1), wq.double(), bq.double()))
    ref = z * h + (1 - z) * q
    s16.take_flags()
    h16 = s16.to_s16(D(h, dev))
    s16.hidden_update(h16, D(delta, dev), D(w1.reshape(64), dev), D(b1, dev), s16.pack_frags(D(w2, dev), D(b2, dev), 64),
                      s16.pack_frags(D(wzr, dev), D(bzr, dev), 0), s16.pack_frags(D(wq, dev), D(bq, dev), 0))
    got = h16.float()
    assert bool(torch.isfinite(got).all()) and float(got.abs().max()) <= 1.0 + 1e-6
    # rows 4.. are ordinary data: 1e-5.  Rows 0-3 carry pre-activations of 1e3 (delta = 100) / 1e4 (1000) whose fp32 rounding alone is
    # ~1e-4 / ~1e-3 absolute (any fp32 implementation, the reference's included, has it): where a gate or q is not saturated the
    # result moves by that much; the point of these rows is finite values and correct saturation
    assert maxdiff(got[:, :, 4:], ref[:, :, 4:]) <= 1e-5
    assert maxdiff(got[:, :, 0:2], ref[:, :, 0:2]) <= 2e-4 and maxdiff(got[:, :, 2:4], ref[:, :, 2:4]) <= 2e-3
    assert s16.take_flags() == 0


def test_domain_guard_flags(dev):
    """The split's domain (|x| <= 65504, finite) is guarded by a device-side flag word instead of the reference's host-synchronising
    NaN asserts (update.py:27-35,...): saturation and non-finite values are reported, ordinary data is silent."""
    from tcs_mi355 import ops, s16
    s16.take_flags()                                                    # clear
    x = torch.randn(1, 16, 8, 32)
    s16.to_s16(D(x, dev))
    assert s16.take_flags() == 0
    big = x.clone()
    big[0, 3, 2, 5] = 1.0e5
    t = s16.to_s16(D(big, dev))
    assert s16.take_flags() == 1 and float(t.float()[0, 3, 2, 5]) == 65504.0
    nan = x.clone()
    nan[0, 1, 0, 0] = float("nan")
    s16.to_s16(D(nan, dev))
    assert s16.take_flags() & 2
    assert s16.take_flags() == 0                                        # read-and-clear
    # a convolution whose output overflows the domain marks the flag from its epilogue
    w = torch.full((32, 16, 3, 3), 500.0)
    s16.conv2d(ops.pack_conv(D(w, dev), None, "f16x3"), [s16.to_s16(D(torch.full((1, 16, 8, 32), 10.0), dev))])
    assert s16.take_flags() & 1


def test_conv2d_s16_two_outputs_equal_separate_launches(dev):
    """tcs_conv_s16_desc.out16b: two layers over the same input as one launch (residual_head[0] + conv_out[0] of the gradient
    predictor, core/update.py:212-214).  Same K loop per output channel, so the result is bit-equal to the two separate launches."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(5)
    for (ca, cb, cin, H, W) in ((128, 64, 64, 13, 37), (32, 96, 48, 9, 33)):
        x = torch.randn(1, cin, H, W, generator=gen)
        wa, wb = (torch.randn(c, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5 for c in (ca, cb))
        ba, bb = torch.randn(ca, generator=gen) * 0.1, torch.randn(cb, generator=gen) * 0.1
        x16 = s16.to_s16(D(x, dev))
        # the separate launches must use the joint weight scale to be bit-comparable: pack each from the joint tensor's slices
        pc = ops.pack_conv(D(torch.cat([wa, wb]), dev), D(torch.cat([ba, bb]), dev), "f16x3")
        oa, ob = s16.zeros(1, ca, H, W, dev), s16.zeros(1, cb, H, W, dev)
        s16.conv2d(pc, [x16], act="relu", out16=oa, out16b=ob, out16_split=ca)
        whole, _ = s16.conv2d(pc, [x16], act="relu")
        assert torch.equal(oa.float(), whole.float()[:, :ca]) and torch.equal(ob.float(), whole.float()[:, ca:])
        for o, w, b in ((oa, wa, ba), (ob, wb, bb)):
            assert maxdiff(o.float(), torch.relu(F.conv2d(x.double(), w.double(), b.double(), padding=1))) <= 2e-5
    with pytest.raises(RuntimeError):          # the split must be tile aligned
        s16.conv2d(pc, [x16], out16=oa, out16b=ob, out16_split=16)


def test_grouped_launch_equals_separate_launches(dev):
    """tcs_conv2d_s16_group: two independent layers as ONE launch (block-index ranges, possibly two tile instances) — the pairs the
    refinement loop groups instead of forking graph branches: two 3x3 64 -> 64 layers (BasicMotionEncoder.convc2 | convf2, the stems'
    second layers; update.py:105-108,200-205) and a 3x3 next to a 1x1 (DispRefine.context_compress | disp_f_stem, update.py:293-297).
    Bit-equal to the separate launches; a combination without a pair kernel falls back to two launches with the same result."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(11)

    def layer(cin, cout, k, H, W, B=1):
        x = torch.randn(B, cin, H, W, generator=gen)
        w = torch.randn(cout, cin, k, k, generator=gen) * (2.0 / (k * k * cin)) ** 0.5
        b = torch.randn(cout, generator=gen) * 0.1
        return x, w, b, s16.to_s16(D(x, dev)), ops.pack_conv(D(w, dev), D(b, dev), "f16x3")

    # (layer A, tile A, layer B, tile B, expect one launch); tiles: CSPLIT*100000 + MT*1000 + ROWS*100 + KSTEPS*10 + NSTAGE, 0 = heuristic
    cases = [
        ((64, 64, 3, 13, 37), 101412, (64, 64, 3, 13, 37), 101412, True),          # same instance, ragged small grid
        ((32, 32, 3, 120, 160), 0, (64, 64, 3, 120, 160), 0, True),                # the stems' second layers at C2 size, heuristic tiles
        ((192, 96, 3, 11, 40), 101812, (27, 96, 1, 11, 40), 101422, True),         # 3x3 8-row tile | 1x1 two-k-step tile
        ((27, 96, 1, 120, 160), 0, (64, 96, 3, 120, 160), 0, True),                # ... at C2 size, 1x1 first, heuristic tiles
        ((96, 96, 3, 120, 160), 0, (96, 96, 1, 120, 160), 0, True),                # context_compress[2] | disp_f_stem[2]
        ((192, 96, 3, 120, 160), 101411, (27, 96, 1, 120, 160), 0, True),          # ... with the 4-row single-stage 3x3 tile the loop uses
        ((64, 64, 3, 9, 33), 101411, (64, 64, 3, 9, 33), 101412, False),           # no pair kernel for this combination: two launches
    ]
    for la, ta, lb, tb, fused in cases:
        xa, wa, ba, xa16, pca = layer(*la)
        xb, wb, bb, xb16, pcb = layer(*lb)
        sep_a, _ = s16.conv2d(pca, [xa16], act="relu", tile_cfg=ta)
        sep_b, _ = s16.conv2d(pcb, [xb16], act="relu", tile_cfg=tb)
        with s16.grouped(report=True) as g:
            grp_a, _ = s16.conv2d(pca, [xa16], act="relu", tile_cfg=ta)
            grp_b, _ = s16.conv2d(pcb, [xb16], act="relu", tile_cfg=tb)
        assert g.fused == [fused], (la, lb, g.fused)
        assert torch.equal(grp_a.data, sep_a.data) and torch.equal(grp_b.data, sep_b.data), (la, lb)
        assert maxdiff(grp_a.float(), torch.relu(F.conv2d(xa.double(), wa.double(), ba.double(), padding=la[2] // 2))) <= 2e-5
        assert maxdiff(grp_b.float(), torch.relu(F.conv2d(xb.double(), wb.double(), bb.double(), padding=lb[2] // 2))) <= 2e-5
    # batch sizes must match for one launch; otherwise two launches, same results
    xa, wa, ba, xa16, pca = layer(64, 64, 3, 9, 33, B=2)
    xb, wb, bb, xb16, pcb = layer(64, 64, 3, 9, 33, B=1)
    with s16.grouped(report=True) as g:
        oa, _ = s16.conv2d(pca, [xa16], tile_cfg=101412)
        ob, _ = s16.conv2d(pcb, [xb16], tile_cfg=101412)
    assert g.fused == [False]
    assert maxdiff(oa.float(), F.conv2d(xa.double(), wa.double(), ba.double(), padding=1)) <= 2e-5
    assert maxdiff(ob.float(), F.conv2d(xb.double(), wb.double(), bb.double(), padding=1)) <= 2e-5


def test_deconv_fused_instance_norm_statistics(dev):
    """tcs_conv_s16_desc.in_stats: the transposed convolution adds fixed-point (sum x, sum x^2) of its own output to 64-bit accumulators
    (integer atomics), tcs_instance_norm_apply_s16 normalises with them (basic_layers.py:28-35,57).  The sums against fp64 on the CPU; the
    result against the two-launch tcs_instance_norm_s16 on the same tensor and against fp64; bit-identical from run to run (integer
    addition does not care which workgroup arrives first); a large mean next to a small spread."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(23)
    # (cin, cout, H, W, B, offset): the two up-blocks of the gradient predictor at 640x480, a ragged grid with batch 2, a biased input
    for cin, cout, H, W, B, off in ((128, 96, 30, 40, 1, 0.3), (96, 64, 60, 80, 1, 0.3), (64, 32, 7, 37, 2, 0.3), (32, 64, 9, 33, 1, 6.0)):
        wt = torch.randn(cin, cout, 4, 4, generator=gen) * (1.0 / (cin * 4)) ** 0.5
        if off > 1:
            wt = wt.abs()                               # every output far from zero: mean^2 >> variance
        x = torch.randn(B, cin, H, W, generator=gen) + off
        rem = torch.randn(B, cout, 2 * H, 2 * W, generator=gen)
        y_ref = F.conv_transpose2d(x.double(), wt.double(), stride=2, padding=1)
        ref = F.leaky_relu(F.instance_norm(y_ref, eps=1e-5), 0.01) + rem.double()
        pc = ops.pack_deconv4x4s2(D(wt, dev))
        x16, rem16 = s16.to_s16(D(x, dev)), s16.to_s16(D(rem, dev))
        first = {}
        for rep in range(2):
            for tc in (0, 1412, 101812):
                ws = s16.deconv_in_stats_workspace(B, cout, H, W, dev)
                y = s16.deconv4x4s2(pc, [x16], in_stats=ws, tile_cfg=tc)
                ys = y.float().cpu().double()
                sums = ws.cpu().double()
                n = 4.0 * H * W
                assert maxdiff(sums[..., 0] / 2 ** 20 / n, ys.mean((2, 3))) <= 2e-6 * max(1.0, off), (cout, tc)
                assert float(((sums[..., 1] / 2 ** 16 / n - (ys * ys).mean((2, 3))).abs() / (ys * ys).mean((2, 3))).max()) <= 1e-6, (cout, tc)
                assert torch.equal(ws, first.setdefault(tc, ws.clone())), (cout, tc, rep)     # same bits from run to run (per tile shape:
                #                                                                               the fp32 partial sums follow the tiling)
                two = s16.instance_norm(y, act="leaky", addend=rem16)                       # the two-launch reference path
                got = s16.instance_norm_apply(y, ws, act="leaky", addend=rem16, out=y)
                tol = 2e-6 if off < 1 else 2e-4          # (E[x^2] - mean^2 with mean^2 / var ~ 1e2: fp32 sums lose two digits)
                assert got is y and maxdiff(y.float(), two.float()) <= tol, (cout, tc)
                assert maxdiff(y.float(), ref) <= max(3e-5, tol), (cout, tc)
    with pytest.raises(ValueError):                      # fp32 workspaces are round 3's first design
        s16.deconv4x4s2(pc, [x16], in_stats=torch.zeros(B * cout * 4, device=dev))
    # small-magnitude channels (sigma ~ 1e-3 next to O(1) ones): the squares are accumulated in units of 2^-16 PER TILE SUM (not per element),
    # so a channel whose tile sums of x^2 are ~1e-4 still keeps ~3 digits; the normalised output must match the two-launch path
    cin, cout, H, W = 96, 64, 60, 80
    wt = torch.randn(cin, cout, 4, 4, generator=gen) * (1.0 / (cin * 4)) ** 0.5
    wt[:, ::4] *= 1e-3                                    # every fourth output channel three orders of magnitude smaller
    x = torch.randn(1, cin, H, W, generator=gen)
    pc = ops.pack_deconv4x4s2(D(wt, dev))
    x16 = s16.to_s16(D(x, dev))
    ws = s16.deconv_in_stats_workspace(1, cout, H, W, dev)
    y = s16.deconv4x4s2(pc, [x16], in_stats=ws)
    two = s16.instance_norm(y, act="leaky")
    got = s16.instance_norm_apply(y, ws, act="leaky")
    ref = F.leaky_relu(F.instance_norm(F.conv_transpose2d(x.double(), wt.double(), stride=2, padding=1), eps=1e-5), 0.01)
    d_small = (got.float()[:, ::4].cpu().double() - ref[:, ::4]).abs().max()
    d_big = (got.float()[:, 1::4].cpu().double() - ref[:, 1::4]).abs().max()
    # the small channels: var ~ 1e-6 is of the order of eps = 1e-5, so the normalised values are O(0.3) and an error of the fixed-point
    # sums shows up directly; 1e-3 absolute there, the usual 3e-5 on the ordinary channels
    assert float(d_big) <= 3e-5 and float(d_small) <= 1e-3, (float(d_big), float(d_small))
    assert maxdiff(got.float()[:, 1::4], two.float()[:, 1::4]) <= 2e-6


def test_gradient_predictor_instance_norm_slots_are_single_use(dev):
    """DispGradPredictor.run(slot=k) lets the transposed convolutions accumulate their InstanceNorm sums into set k of per-frame accumulators,
    which only ADD: a slot index may be used once between two begin_frame() calls.  A repeated index, an index beyond IN_SUM_SLOTS
    (iterations > 64) or a run without begin_frame() must fall back to the statistics launch and give the same result, never a silently
    wrong mean / rstd."""
    from argparse import Namespace
    from core.update import IN_SUM_SLOTS, DispGradPredictor, pool_of
    from tcs_mi355 import ops, s16
    torch.manual_seed(3)
    m = DispGradPredictor(Namespace()).to(dev).eval()
    pool = pool_of(m)
    gen = torch.Generator().manual_seed(4)
    B, H, W = 1, 32, 48
    g5 = D(torch.randn(B, 2, H, W, generator=gen), dev)
    disp = D(torch.rand(B, 1, H, W, generator=gen) * 20, dev)
    cands = ops.grad_candidates(disp)
    clist = [s16.to_s16(D(torch.randn(B, 64, H >> i, W >> i, generator=gen), dev)) for i in range(3)]
    with torch.no_grad():
        pre = m.prepare(pool, clist)
        ref_g, ref_c = m.run(pool, g5, cands, pre, slot=None)                 # statistics launches
        ref_g, ref_c = ref_g.clone(), ref_c.float().clone()
        m.begin_frame(pool, B, dev)                                            # (each call is compared right away: pool buffers are reused)
        for k in (0, 0, 1, IN_SUM_SLOTS, IN_SUM_SLOTS + 1, 1):
            g, c = m.run(pool, g5, cands, pre, slot=k)
            assert maxdiff(g, ref_g) <= 2e-5 and maxdiff(c.float(), ref_c) <= 2e-5, k
        m.__dict__.pop("_slots_used", None)                                    # a caller that never called begin_frame
        g, c = m.run(pool, g5, cands, pre, slot=3)
        assert maxdiff(g, ref_g) <= 2e-5 and maxdiff(c.float(), ref_c) <= 2e-5


def test_tap_partials_fold_a_narrow_3x3_convolution_into_its_producer(dev):
    """tcs_conv_s16_desc.tap_*: FlowHead.conv2 (256 -> 1) and residual_head[2] (128 -> 2) never run as launches (update.py:13-17,196,213).
    The producer's tap partials, summed by tcs_taps_sum, must equal conv2(relu(conv1(x))) (fp64 reference); the two fused consumers
    must equal the unfused kernels fed with that sum, bit for bit (same additions in the same order)."""
    from tcs_mi355 import ops, s16
    gen = torch.Generator().manual_seed(31)
    for (cin, cmid, nout, H, W, B, extra) in ((128, 256, 1, 21, 45, 1, 0), (64, 128, 2, 18, 37, 2, 64), (32, 40, 1, 9, 33, 1, 0)):
        x = torch.randn(B, cin, H, W, generator=gen)
        w1 = torch.randn(cmid + extra, cin, 3, 3, generator=gen) * (2.0 / (9 * cin)) ** 0.5
        b1 = torch.randn(cmid + extra, generator=gen) * 0.1
        w2 = torch.randn(nout, cmid, 3, 3, generator=gen) * (2.0 / (9 * cmid)) ** 0.5
        b2 = torch.randn(nout, generator=gen) * 0.1
        y = torch.relu(F.conv2d(x.double(), w1.double(), b1.double(), padding=1))
        ref = F.conv2d(y[:, :cmid], w2.double(), b2.double(), padding=1)
        pc = ops.pack_conv(D(w1, dev), D(b1, dev), "f16x3")
        ntile = (cmid + 31) // 32
        taps = s16.Taps(torch.empty(B, ntile, 9 * nout, H, W, device=dev), ntile, nout, D(b2, dev))
        x16 = s16.to_s16(D(x, dev))
        second = s16.zeros(B, extra, H, W, dev) if extra else None
        for tc in (0, 1412, 101812):
            taps.data.fill_(float("nan"))
            s16.conv2d(pc, [x16], act="relu", taps=taps, tap_weights=s16.pack_taps(D(w2, dev)), out16b=second, out16_split=cmid if extra else 0,
                       tile_cfg=tc)
            got = s16.taps_sum(taps)
            assert maxdiff(got, ref) <= 3e-5, (cmid, nout, tc)
            if extra:
                assert maxdiff(second.float(), y[:, cmid:]) <= 2e-5
        add = torch.randn(B, nout, H, W, generator=gen)
        assert maxdiff(s16.taps_sum(taps, addend=D(add, dev), scale=0.2), (ref + add.double()) * 0.2) <= 1e-5
        if nout == 1:
            c1 = torch.arange(W, dtype=torch.float32).view(1, 1, 1, W).expand(B, 1, H, W) - torch.rand(B, 1, H, W, generator=gen) * 30
            c1 = D(c1.contiguous(), dev)
            dq, g, c, dl = s16.flow_taps_step_grads(c1, taps, scale=5.0, want_delta=True)
            assert torch.equal(dl, got)
            for a, b in zip((dq, g, c), ops.flow_step_grads(c1, got, scale=5.0)):
                assert torch.equal(torch.nan_to_num(a, nan=7.0, posinf=8.0, neginf=9.0), torch.nan_to_num(b, nan=7.0, posinf=8.0, neginf=9.0))
        else:
            g5 = D(torch.randn(B, 2, H, W, generator=gen), dev)
            disp = D(torch.rand(B, 1, H, W, generator=gen) * 40, dev)
            grad_ref = s16.taps_sum(taps, addend=g5, scale=0.2)
            o16, c9, grad = s16.taps_propagate(taps, g5, 0.2, disp)
            r16, r9 = s16.propagate_disparity(grad_ref, disp)
            assert torch.equal(grad, grad_ref) and torch.equal(c9, r9) and torch.equal(o16.data, r16.data)
