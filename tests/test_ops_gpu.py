"""Kernel-level parity: every HIP op (through the C ABI / ctypes) against a plain PyTorch CPU reference of
the same op in fp64, forward and backward.  Tolerances are fp32-rounding sized and written per test."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as TF

pytestmark = pytest.mark.gpu


def _dev():
    return torch.device("cuda:0")


def _rand(*shape, seed=0, scale=1.0):
    g = np.random.Generator(np.random.PCG64(seed + sum(shape)))
    return torch.from_numpy(g.standard_normal(shape).astype(np.float32) * scale)


def _close(a, b, tol, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    scale = max(1.0, b.abs().max().item())
    err = (a - b).abs().max().item()
    assert err <= tol * scale, f"{what}: max err {err:.3e} (scale {scale:.3g}, tol {tol:g})"


CONV_CASES = [
    # N, H, W, Ci, Co, k, stride, pad, dil, bias, relu
    (2, 13, 17, 64, 96, 3, 1, 1, 1, True, True),
    (2, 15, 20, 32, 128, 3, 1, 2, 2, False, False),
    (1, 30, 40, 64, 64, 3, 1, 4, 4, False, False),
    (2, 16, 20, 64, 128, 1, 2, 0, 1, False, False),
    (2, 9, 11, 96, 40, 1, 1, 0, 1, True, False),
    (2, 12, 16, 32, 128, 5, 1, 2, 1, True, False),
    (3, 10, 12, 160, 192, 3, 1, 1, 1, True, False),
    (2, 17, 21, 32, 64, 3, 2, 1, 1, False, False),
    (2, 17, 23, 128, 192, 1, 1, 0, 1, False, False),       # pointwise convs: ragged M and N tiles
    (2, 18, 22, 256, 64, 1, 2, 0, 1, True, True),          # ... strided rows, bias, relu
    (2, 14, 18, 128, 288, 3, 1, 1, 1, True, False),      # Ci % 128 == 0: exercises the split-scheme wgrad (tr reads)
    (3, 9, 13, 256, 64, 3, 1, 2, 2, False, False),
    (2, 16, 20, 128, 256, 1, 2, 0, 1, False, False),
    # 64-pixel-wide maps: the halo build of h2_kernel (one activation block with a one-pixel halo per channel block for all 9 taps);
    # tiles at the top / bottom of an image, ragged output-channel tiles, several images
    (3, 8, 64, 96, 160, 3, 1, 1, 1, True, True),
    (2, 4, 64, 32, 64, 3, 1, 1, 1, False, False),
    (1, 12, 64, 64, 288, 3, 1, 1, 1, False, False),
]


@pytest.mark.parametrize("path", ["fp32_mfma", "bf16x3", "f16x2"])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(case, path, monkeypatch):
    """both GEMM back-ends against fp64 with the SAME fp32-rounding-sized bar (the 3xbf16-split kernel is fp32-faithful)"""
    from scanpaths_amd import functional as F
    monkeypatch.setattr(F, "USE_BF16X3", path != "fp32_mfma")
    monkeypatch.setattr(F, "SPLIT_SCHEME", path if path != "fp32_mfma" else "bf16x3")
    if path != "fp32_mfma":    # force the split kernels even where the cost model would not pick them
        monkeypatch.setattr(F, "_b3_pays", lambda M, N, K, Kc, nbatch=1, **kw: nbatch == 1 and Kc % 16 == 0)
        if path == "f16x2":    # the 2xfp16 weight-gradient kernel also takes column tiles that span several filter taps (Ci < 128)
            monkeypatch.setattr(F, "_w3_pays", lambda M, Co, K, Ci, nbatch=1, **kw: nbatch == 1 and Ci % 32 == 0 and Co % 32 == 0)
        else:
            monkeypatch.setattr(F, "_w3_pays", lambda M, Co, K, Ci, nbatch=1, **kw: nbatch == 1 and Ci % 128 == 0 and Co % 16 == 0)
    N, H, W, Ci, Co, k, s, p, d, has_b, relu = case
    x = _rand(N, Ci, H, W, seed=1)
    w = _rand(Co, Ci, k, k, seed=2, scale=1.0 / math.sqrt(Ci * k * k))
    b = _rand(Co, seed=3) if has_b else None
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    br = b.double().requires_grad_(True) if has_b else None
    yr = TF.conv2d(xr, wr, br, stride=s, padding=p, dilation=d)
    if relu:
        yr = TF.relu(yr)
    gy = _rand(*yr.shape, seed=4)
    yr.backward(gy.double())

    dev = _dev()
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    wg = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bg = b.to(dev).requires_grad_(True) if has_b else None
    y = F.conv2d(xg, wg, bg, stride=s, pad=p, dil=d, relu=relu)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    _close(y.permute(0, 3, 1, 2), yr, 2e-6, "y")
    _close(xg.grad.permute(0, 3, 1, 2), xr.grad, 2e-6, "dx")
    _close(wg.grad, wr.grad, 3e-6, "dw")
    if has_b:
        _close(bg.grad, br.grad, 3e-6, "db")


def test_stem_conv_7x7_s2_padded_channels():
    from scanpaths_amd import functional as F
    x = _rand(2, 3, 37, 45, seed=5)
    w = _rand(64, 3, 7, 7, seed=6, scale=0.1)
    xr, wr = x.double(), w.double().requires_grad_(True)
    yr = TF.conv2d(xr, wr, None, stride=2, padding=3)
    gy = _rand(*yr.shape, seed=7)
    yr.backward(gy.double())
    dev = _dev()
    wg = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    x4 = F.nchw_to_nhwc(x.to(dev), 4)
    w4 = F.pad_last(wg.permute(0, 2, 3, 1), 4).permute(0, 3, 1, 2)
    y = F.conv2d(x4, w4, None, stride=2, pad=3)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    _close(y.permute(0, 3, 1, 2), yr, 2e-6, "y")
    _close(wg.grad, wr.grad, 3e-6, "dw")


@pytest.mark.parametrize("layout", ["nk", "kn"])
@pytest.mark.parametrize("shape", [(5, 1200, 1200), (64, 512, 512), (1, 512, 512), (130, 36, 260),
                                   # the dense layers of the decode loop at the benchmark size (csrc/gemm_skinny.hip): spatial_embed,
                                   # the rank-1 filter contraction and its data gradient's shape, a row count that is no multiple of 16
                                   (64, 2560, 2560), (32, 512, 13824), (32, 13824, 512), (37, 528, 1216)])
def test_gemm(layout, shape):
    from scanpaths_amd import functional as F
    M, K, N = shape
    F.reset_fusion_counts()
    a = _rand(M, K, seed=8)
    b = _rand(N, K, seed=9) if layout == "nk" else _rand(K, N, seed=9)
    bias = _rand(N, seed=10)
    ar, br, biasr = a.double().requires_grad_(True), b.double().requires_grad_(True), bias.double().requires_grad_(True)
    cr = (ar @ (br.t() if layout == "nk" else br)) * 0.5 + biasr
    gc = _rand(M, N, seed=11)
    cr.backward(gc.double())
    dev = _dev()
    ag, bg, biasg = (t.to(dev).requires_grad_(True) for t in (a, b, bias))
    c = F.gemm(ag, bg, biasg, layout, alpha=0.5)
    c.backward(gc.to(dev))
    tol = 3e-6 * math.sqrt(K / 32)
    _close(c, cr, tol, "c")
    _close(ag.grad, ar.grad, tol, "da")
    _close(bg.grad, br.grad, tol, "db")
    _close(biasg.grad, biasr.grad, tol, "dbias")
    if M <= 64 and K % 64 == 0 and N % 64 == 0:          # forward AND data gradient on the skinny kernel (one "nk", one "kn")
        assert F.FUSION_COUNTS["skinny_gemm"] == 2, F.FUSION_COUNTS


def test_gemm_batched_kn_relu():
    from scanpaths_amd import functional as F
    Bt, M, K, N = 3, 2, 1200, 512
    a, b = _rand(Bt, M, K, seed=12).abs(), _rand(Bt, K, N, seed=13)
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    cr = TF.relu(torch.bmm(ar, br) / K)
    gc = _rand(Bt, M, N, seed=14)
    cr.backward(gc.double())
    dev = _dev()
    ag, bg = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    c = F.gemm(ag, bg, None, "kn", alpha=1.0 / K, relu=True)
    c.backward(gc.to(dev))
    _close(c, cr, 2e-6, "c")
    _close(ag.grad, ar.grad, 2e-6, "da")
    _close(bg.grad, br.grad, 2e-6, "db")


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("with_res", [True, False])
def test_bn_act(training, with_res):
    from scanpaths_amd import functional as F
    N, C, H, W = 3, 64, 9, 11
    x = _rand(N, C, H, W, seed=15) * 2 + 0.5
    res = _rand(N, C, H, W, seed=16) if with_res else None
    gamma, beta = _rand(C, seed=17).abs() + 0.5, _rand(C, seed=18)
    rm, rv = _rand(C, seed=19) * 0.1, _rand(C, seed=20).abs() + 0.5
    xr, gr, br = x.double().requires_grad_(True), gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rr = res.double().requires_grad_(True) if with_res else None
    rmr, rvr = rm.double().clone(), rv.double().clone()
    yr = TF.batch_norm(xr, rmr, rvr, gr, br, training, 0.1, 1e-5)
    if with_res:
        yr = yr + rr
    yr = TF.relu(yr)
    gy = _rand(N, C, H, W, seed=21)
    yr.backward(gy.double())
    dev = _dev()
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev)
    xg, gg, bg = nhwc(x).requires_grad_(True), gamma.to(dev).requires_grad_(True), beta.to(dev).requires_grad_(True)
    rg = nhwc(res).requires_grad_(True) if with_res else None
    rmg, rvg = rm.to(dev), rv.to(dev)
    y = F.bn_act(xg, gg, bg, rmg, rvg, rg, training=training, relu=True)
    y.backward(nhwc(gy))
    _close(y.permute(0, 3, 1, 2), yr, 3e-6, "y")
    _close(xg.grad.permute(0, 3, 1, 2), xr.grad, 1e-5, "dx")
    _close(gg.grad, gr.grad, 1e-5, "dgamma")
    _close(bg.grad, br.grad, 1e-5, "dbeta")
    if with_res:
        _close(rg.grad.permute(0, 3, 1, 2), rr.grad, 3e-6, "dres")
    _close(rmg, rmr, 2e-6, "running_mean")
    _close(rvg, rvr, 2e-6, "running_var")


@pytest.mark.parametrize("hw", [(120, 160), (17, 23), (16, 20)])
def test_maxpool_ceil_with_ties(hw):
    from scanpaths_amd import functional as F
    H, W = hw
    x = TF.relu(_rand(2, 8, H, W, seed=22))        # many exact-zero ties, like the post-ReLU stem
    xr = x.double().requires_grad_(True)
    yr = TF.max_pool2d(xr, 3, 2, 0, ceil_mode=True)
    gy = _rand(*yr.shape, seed=23)
    yr.backward(gy.double())
    dev = _dev()
    xg = x.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    y = F.maxpool3s2(xg)
    y.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
    assert torch.equal(y.permute(0, 3, 1, 2).cpu().double(), yr.detach())
    _close(xg.grad.permute(0, 3, 1, 2), xr.grad, 1e-6, "dx")


@pytest.mark.parametrize("fused", [False, True, "epilogue", "epilogue_planes", "epilogue_w64"])
def test_lstm_cell_and_gate_conv(fused):
    """fused=False: rank-1 gate terms accumulated by the batched GEMM of gate_conv; fused=True: plain h-conv + lstm_cell_rank1
    (rank-1 terms inside the pointwise kernel; 135 pixels = 2 full 64-pixel tiles + a tail: the path of map sizes whose pixel
    count is no multiple of 256); "epilogue": the whole cell as the epilogue of the h-gate conv (sp_gateconv_lstm_f16x2, the
    path of the 40x64 benchmark map; 3 samples x 256 pixels, 96 channels = 3 channel tiles of the gathered weight rows)"""
    from scanpaths_amd import functional as F
    # "epilogue_w64": the same on a 64-pixel-wide map (8 x 64, two tiles per sample): the halo build of the kernel, the benchmark's
    B, Hm, Wm, C, S = (2, 8, 64, 64, 2) if fused == "epilogue_w64" else (3, 16, 16, 96, 2) if str(fused).startswith("epilogue") \
        else (2, 9, 15, 64, 2) if fused else (2, 6, 8, 32, 2)
    KP = 20
    h, c = _rand(B, C, Hm, Wm, seed=24), _rand(B, C, Hm, Wm, seed=25)
    xg = _rand(B, 4 * C, Hm, Wm, seed=26)
    wh = _rand(4 * C, C, 3, 3, seed=27, scale=0.1)
    sp = _rand(S, B, Hm, Wm, seed=28)
    wr = [_rand(3 * C, C, 3, 3, seed=29 + s, scale=0.1) for s in range(S)]     # rank-1 conv weights (i,f,o)
    se = _rand(S, B, C, seed=33)
    dbl = lambda t: t.double().requires_grad_(True)
    hr, cr, xgr, whr, spr, ser = dbl(h), dbl(c), dbl(xg), dbl(wh), dbl(sp), dbl(se)
    wrr = [dbl(t) for t in wr]
    pre = xgr + TF.conv2d(hr, whr, padding=1)
    extra = 0
    for s in range(S):
        ss = spr[s].unsqueeze(1) * ser[s].unsqueeze(-1).unsqueeze(-1)
        extra = extra + TF.conv2d(ss, wrr[s], padding=1)
    pre = pre + torch.cat([extra, torch.zeros_like(extra[:, :C])], 1)
    i, f, o, g = pre[:, :C].sigmoid(), pre[:, C:2 * C].sigmoid(), pre[:, 2 * C:3 * C].sigmoid(), pre[:, 3 * C:].tanh()
    c2 = f * cr + i * g
    h2 = o * c2
    gh, gc = _rand(*h2.shape, seed=40), _rand(*c2.shape, seed=41)
    (h2 * gh.double()).sum().add((c2 * gc.double()).sum()).backward()

    dev = _dev()
    nhwc = lambda t: t.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
    hg_, cg_, xgg = nhwc(h), nhwc(c), nhwc(xg)
    whg = wh.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    spg, seg = sp.to(dev).requires_grad_(True), se.to(dev).requires_grad_(True)
    wrg = [t.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True) for t in wr]
    parts = []
    for s in range(S):
        wflat = wrg[s].permute(0, 2, 3, 1).reshape(3 * C * 9, C)
        parts.append(F.gemm(seg[s], wflat, None, "nk").view(B, 3 * C, 9))
    wc = torch.cat(parts + [torch.zeros(B, 3 * C, KP - 9 * S, device=dev)], 2)
    spcol = F.im2col3x3(spg, KP)
    if fused in ("epilogue", "epilogue_w64"):
        hn, cn = F.gateconv_lstm(hg_, whg, xgg, cg_, spcol, wc, {})
        assert float(hn._sp_amax[1]) == float(hn.abs().max())          # fused max|h| hint (float bits in slot 1)
    elif fused == "epilogue_planes":
        # with a bound of max|c_prev| on the state tensor (|c_t| <= t + 1 in the model) the epilogue also writes h's split operand
        cg_._sp_cbound = float(c.abs().max()) + 0.5
        hn, cn = F.gateconv_lstm(hg_, whg, xgg, cg_, spcol, wc, {})
        op = hn._sp_cache["f16x2"]
        bound = float(op.scale[1])
        assert bound == cg_._sp_cbound + 1.0 and float(hn.abs().max()) <= bound
        assert float((_decode_split(op, hn.shape) - hn.detach()).abs().max()) <= 2.0 ** -21 * bound
    elif fused:
        hn, cn = F.lstm_cell_rank1(xgg, F.conv2d(hg_, whg, None, pad=1), cg_, spcol, wc)
    else:
        hgate = F.gate_conv(hg_, whg, spcol, wc, (Hm, Wm))
        hn, cn = F.lstm_cell(xgg, hgate, cg_)
    (hn * nhwc(gh).detach()).sum().add((cn * nhwc(gc).detach()).sum()).backward()
    back = lambda t: t.permute(0, 3, 1, 2)
    _close(back(hn), h2, 3e-6, "h")
    _close(back(cn), c2, 3e-6, "c")
    _close(back(hg_.grad), hr.grad, 1e-5, "dh")
    _close(back(cg_.grad), cr.grad, 1e-5, "dc")
    _close(back(xgg.grad), xgr.grad, 1e-5, "dxg")
    _close(whg.grad, whr.grad, 1e-5, "dwh")
    _close(spg.grad, spr.grad, 1e-5, "dspatial")
    _close(seg.grad, ser.grad, 1e-5, "dsemantic")
    for s in range(S):
        _close(wrg[s].grad, wrr[s].grad, 1e-5, f"dwr{s}")


def test_list_attention_and_small_ops():
    from scanpaths_amd import functional as F
    T, R, D = 5, 6, 1200
    Lst, u = _rand(T, R, D, seed=42), _rand(D, seed=43, scale=0.05)
    Lr, ur = Lst.double().requires_grad_(True), u.double().requires_grad_(True)
    sc = (Lr * ur).sum(-1)
    memr = (Lr * sc.softmax(0).unsqueeze(-1)).sum(0)
    gm = _rand(R, D, seed=44)
    memr.backward(gm.double())
    dev = _dev()
    Lg, ug = Lst.to(dev).requires_grad_(True), u.to(dev).requires_grad_(True)
    mem = F.list_attention(Lg, ug)
    mem.backward(gm.to(dev))
    _close(mem, memr, 3e-6, "mem")
    _close(Lg.grad, Lr.grad, 1e-5, "dL")
    _close(ug.grad, ur.grad, 1e-5, "du")

    a, b = _rand(2, 3, 50, seed=45), _rand(3, 50, seed=46)
    ar, br = a.double().requires_grad_(True), b.double().requires_grad_(True)
    outr = TF.relu(ar * br)
    go = _rand(2, 3, 50, seed=47)
    outr.backward(go.double())
    ag, bg = a.to(dev).requires_grad_(True), b.to(dev).requires_grad_(True)
    out = F.mul_relu(ag, bg)
    out.backward(go.to(dev))
    _close(out, outr, 1e-6)
    _close(ag.grad, ar.grad, 1e-6)
    _close(bg.grad, br.grad, 1e-6)

    x = _rand(2, 5, 7, 512, seed=48)
    xr = x.double().requires_grad_(True)
    mr = xr.mean(-1)
    gmm = _rand(2, 5, 7, seed=49)
    mr.backward(gmm.double())
    xg = x.to(dev).requires_grad_(True)
    m = F.channel_mean(xg)
    m.backward(gmm.to(dev))
    _close(m, mr, 1e-6)
    _close(xg.grad, xr.grad, 1e-6)

    a2, b2 = _rand(4, 33, seed=50), _rand(4, 33, seed=51)
    sel = torch.tensor([True, False, False, True])
    ag, bg = a2.to(dev).requires_grad_(True), b2.to(dev).requires_grad_(True)
    o = F.select_rows(ag, bg, sel.to(dev))
    o.backward(torch.ones_like(o))
    assert torch.equal(o.cpu(), torch.where(sel[:, None], a2, b2))
    assert torch.equal(ag.grad.cpu(), sel[:, None].float().expand_as(a2))
    assert torch.equal(bg.grad.cpu(), (~sel)[:, None].float().expand_as(a2))


def test_loss_and_clip_adam_vs_oracle():
    from oracle import scanpath_oracle as O
    from scanpaths_amd import functional as F
    from scanpaths_amd.optim import FlatAdam  # noqa: F401  (import check)
    from scanpaths_amd.synth import make_batch
    B, T, A = 3, 5, 1201
    batch = make_batch("AiR", B, 240, 320, T, seed=9)
    z, mu = _rand(B, T, A, seed=52) * 2, _rand(B, T, seed=53)
    s2 = _rand(B, T, seed=54).abs() + 0.3
    zr, mur, s2r = z.double().requires_grad_(True), mu.double().requires_grad_(True), s2.double().requires_grad_(True)
    bd = {k: (v.double() if v.is_floating_point() else v) for k, v in batch.items()}
    lossr, lar, ldr = O.supervised_loss({"all_actions_prob": zr, "log_normal_mu": mur, "log_normal_sigma2": s2r}, bd)
    (lossr * 1.7).backward()
    dev = _dev()
    zg, mug, s2g = (t.to(dev).requires_grad_(True) for t in (z, mu, s2))
    loss, la, ld = F.scanpath_loss(zg, mug, s2g, batch["scanpaths"].to(dev), batch["action_masks"].to(dev),
                                   batch["durations"].to(dev), batch["duration_masks"].to(dev), 1.0)
    (loss * 1.7).backward()
    _close(torch.stack([loss, la, ld]), torch.stack([lossr, lar, ldr]), 2e-6, "loss")
    _close(zg.grad, zr.grad, 2e-6, "dz")
    _close(mug.grad, mur.grad, 2e-6, "dmu")
    _close(s2g.grad, s2r.grad, 2e-6, "dsigma2")


def test_flat_adam_matches_oracle():
    from oracle import scanpath_oracle as O
    from scanpaths_amd.optim import FlatAdam
    shapes = {"a.weight": (64, 32, 3, 3), "a.bias": (64,), "b.weight": (7, 5), "c.bias": (1,)}
    params = {k: _rand(*s, seed=60 + i) for i, (k, s) in enumerate(shapes.items())}
    grads = {k: _rand(*s, seed=70 + i) * 3 for i, (k, s) in enumerate(shapes.items())}
    ref = {k: v.double().clone() for k, v in params.items()}
    state = {}
    dev = _dev()
    plist = [torch.nn.Parameter(v.to(dev)) for v in params.values()]
    opt = FlatAdam(plist, lr=1e-3, weight_decay=5e-5, clip=12.5)
    for step in range(3):
        gstep = {k: g.double() * (step + 1) for k, g in grads.items()}
        tn_ref = O.clip_and_adam(ref, gstep, state, lr=1e-3, clip=12.5, weight_decay=5e-5)
        for p, g in zip(plist, grads.values()):
            p.grad = (g * (step + 1)).to(dev)
        tn = opt.step()
        assert abs(float(tn) - tn_ref) <= 1e-5 * tn_ref
    for p, k in zip(plist, params):
        _close(p, ref[k], 2e-6, k)


@pytest.mark.parametrize("hw,per_sample", [((30, 40), False), ((40, 64), False), ((12, 13), True)])
def test_direct_head_matches_two_conv_formulation(hw, per_sample):
    """compose11 + sal_gather + drt_direct + head_finish(dpre) against the literal predict_head order in fp64
    (5x5 conv with the composed filters, then the 7x7 stride-5 pad-2 tap sums with zero padding of the INTERMEDIATE map):
    maps, duration sites, and the gradients w.r.t. h, the composed filters, the composed biases and drt_layer_2."""
    from scanpaths_amd import functional as F
    Hm, Wm = hw
    B, C_, HC = 3, 32, 64
    nsrc = 3 if per_sample else 2
    nsel = 1 if per_sample else 2
    dh, dw = (Hm + 4 - 7) // 5 + 1, (Wm + 4 - 7) // 5 + 1
    S, P = dh * dw, Hm * Wm
    h = _rand(B, Hm, Wm, C_, seed=1)
    G = _rand(nsrc * HC, 5, 5, C_, seed=2, scale=1.0 / math.sqrt(25 * C_))
    G.view(nsrc, HC, 5, 5, C_)[:, 51:] = 0
    cbh = _rand(nsrc, HC, seed=3, scale=0.1)
    w2 = _rand(2, S, seed=4, scale=0.2)
    b2 = _rand(2, seed=5, scale=0.1)
    src = torch.tensor([[2], [0], [2]]) if per_sample else torch.arange(2).repeat(B, 1)      # [B, nsel]

    # ---- fp64 reference ----
    hr, Gr, cbr = h.double().requires_grad_(True), G.double().requires_grad_(True), cbh.double().requires_grad_(True)
    w2r, b2r = w2.double().requires_grad_(True), b2.double().requires_grad_(True)
    Zf = TF.conv2d(hr.permute(0, 3, 1, 2), Gr.permute(0, 3, 1, 2), None, padding=2)           # [B, nsrc*HC, Hm, Wm]
    Zf = Zf.view(B, nsrc, HC, Hm, Wm)
    logits_r, amap_r, mu_r, s2_r, dpre_r = [], [], [], [], []
    for i in range(nsel):
        lg_b, am_b, mu_b, s2_b, dp_b = [], [], [], [], []
        for b in range(B):
            k = int(src[b, i])
            z, c = Zf[b, k], cbr[k]
            term = z[0].mean() + c[0]
            am = torch.relu(z[1] + c[1]).reshape(P)
            inter = (z[2:51] + c[2:51].view(49, 1, 1))                                           # 49 "channels" = taps
            # 7x7 stride-5 pad-2 conv whose tap (ky,kx) reads channel ky*7+kx: a one-hot depth filter
            eye = torch.eye(49, dtype=torch.float64).view(1, 49, 7, 7)
            d = TF.conv2d(inter.unsqueeze(0), eye, None, stride=5, padding=2).reshape(S)
            dp_b.append(d)
            drt = torch.relu(d + c[51])
            mu_b.append((w2r[0] * drt).sum() + b2r[0])
            s2_b.append(torch.exp((w2r[1] * drt).sum() + b2r[1]))
            lg_b.append(torch.cat([term.view(1), am]))
            am_b.append(am)
        logits_r.append(torch.stack(lg_b)); amap_r.append(torch.stack(am_b)); mu_r.append(torch.stack(mu_b))
        s2_r.append(torch.stack(s2_b)); dpre_r.append(torch.stack(dp_b))
    logits_r, amap_r, mu_r, s2_r = torch.stack(logits_r), torch.stack(amap_r), torch.stack(mu_r), torch.stack(s2_r)

    # ---- HIP path ----
    dev = _dev()
    hd = h.to(dev).requires_grad_(True)
    Gd = G.to(dev).permute(0, 3, 1, 2).requires_grad_(True)           # logical OIHW view of the physical layout
    cbd = cbh.to(dev).requires_grad_(True)
    w2d, b2d = w2.to(dev).requires_grad_(True), b2.to(dev).requires_grad_(True)
    hmap = src.to(torch.int32).to(dev).contiguous()
    Gp = Gd.permute(0, 2, 3, 1).reshape(nsrc, HC, 25, C_)
    R = (nsrc * 50 + 63) // 64 * 64
    Wsal = torch.cat([Gp[:, :2].reshape(nsrc * 50, C_), torch.zeros(R - nsrc * 50, C_, device=dev)], 0).view(R, C_, 1, 1)
    W11, cbsum = F.compose11(Gd, cbd, nsrc, HC, (Hm, Wm))
    Z2 = F.sal_gather(F.conv2d(hd, Wsal, None, pad=0), hmap, nsel, nsrc)
    Dpre = F.drt_direct(hd, W11, cbsum, hmap, nsel)
    cb_arg = cbd.index_select(0, hmap[:, 0].long()).view(B, 1, HC) if per_sample else cbd
    logits, amap, mu, s2 = F.head_finish(Z2, cb_arg, w2d, b2d, nsel, HC, False, per_sample=per_sample, dpre=Dpre)
    _close(Dpre, torch.stack([d for d in dpre_r]), 2e-5, "Dpre")
    _close(logits, logits_r, 2e-5, "logits")
    _close(amap, amap_r, 2e-5, "amap")
    _close(mu, mu_r, 2e-5, "mu")
    _close(s2, s2_r, 5e-5, "sigma2")

    gl, gm, gs, ga = _rand(*logits.shape, seed=6), _rand(*mu.shape, seed=7), _rand(*s2.shape, seed=8), _rand(*amap.shape, seed=9)
    ((logits_r * gl.double()).sum() + (mu_r * gm.double()).sum() + (s2_r * gs.double()).sum()
     + (amap_r * ga.double()).sum()).backward()
    ((logits * gl.to(dev)).sum() + (mu * gm.to(dev)).sum() + (s2 * gs.to(dev)).sum() + (amap * ga.to(dev)).sum()).backward()
    _close(hd.grad, hr.grad, 3e-5, "dh")
    _close(Gd.grad.permute(0, 2, 3, 1), Gr.grad, 3e-5, "dG")
    _close(cbd.grad, cbr.grad, 3e-5, "dcb")
    _close(w2d.grad, w2r.grad, 3e-5, "dw2")
    _close(b2d.grad, b2r.grad, 3e-5, "db2")


def test_split_gemms_are_as_accurate_as_cpu_fp32():
    """fp32-faithfulness of the split GEMM back-ends on decoder-shaped data (h = o*c in (-1,1) with many near-zero entries,
    heavy-tailed gradients): rms error against fp64 of the forward conv, data gradient and weight gradient must not exceed
    1.5x the error of the same conv computed by torch on the CPU in fp32 (measured: 0.7-0.9x for 2xfp16, 0.3-1.1x for 3xbf16)."""
    from scanpaths_amd import functional as F
    g = torch.Generator().manual_seed(0)
    B, Hm, Wm, C = 2, 15, 20, 256
    h = torch.sigmoid(torch.randn(B, C, Hm, Wm, generator=g) * 2) * (torch.randn(B, C, Hm, Wm, generator=g) * 0.7)
    w = torch.randn(2 * C, C, 3, 3, generator=g) * (1.0 / math.sqrt(9 * C))
    gy = torch.randn(B, 2 * C, Hm, Wm, generator=g) * torch.rand(B, 2 * C, Hm, Wm, generator=g) ** 4
    hr, wr = h.double().requires_grad_(True), w.double().requires_grad_(True)
    yr = TF.conv2d(hr, wr, padding=1)
    yr.backward(gy.double())
    h32, w32 = h.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y32 = TF.conv2d(h32, w32, padding=1)
    y32.backward(gy)
    rel = lambda a, b: ((a.double().cpu() - b).pow(2).mean().sqrt() / b.pow(2).mean().sqrt()).item()
    cpu = {"y": rel(y32.detach(), yr.detach()), "dh": rel(h32.grad, hr.grad), "dw": rel(w32.grad, wr.grad)}
    dev = _dev()
    saved = (F.USE_BF16X3, F.SPLIT_SCHEME, F._b3_pays, F._w3_pays)
    try:
        F.USE_BF16X3 = True
        F._b3_pays = lambda M, N, K, Kc, nbatch=1, **kw: True
        F._w3_pays = lambda M, Co, K, Ci, nbatch=1, **kw: True
        for scheme in ("f16x2", "bf16x3"):
            F.SPLIT_SCHEME = scheme
            hd = h.permute(0, 2, 3, 1).contiguous().to(dev).requires_grad_(True)
            wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = F.conv2d(hd, wd, None, pad=1)
            y.backward(gy.permute(0, 2, 3, 1).contiguous().to(dev))
            got = {"y": rel(y.detach().permute(0, 3, 1, 2), yr.detach()), "dh": rel(hd.grad.permute(0, 3, 1, 2), hr.grad),
                   "dw": rel(wd.grad, wr.grad)}
            for k in got:
                assert got[k] <= 1.5 * cpu[k], (scheme, k, got[k], cpu[k])
    finally:
        F.USE_BF16X3, F.SPLIT_SCHEME, F._b3_pays, F._w3_pays = saved


def test_f16x2_scales_survive_extreme_operand_magnitudes():
    """per-tensor power-of-two scales of the 2xfp16 GEMMs: operands of magnitude 1e-30 (vanishing gradients) and 1e+20 must
    give the same relative accuracy as ordinary ones -- the product of the two scales leaves the fp32 range, each one alone
    does not"""
    from scanpaths_amd import functional as F
    saved = (F.USE_BF16X3, F.SPLIT_SCHEME, F._b3_pays, F._w3_pays)
    dev = _dev()
    try:
        F.USE_BF16X3, F.SPLIT_SCHEME = True, "f16x2"
        F._b3_pays = lambda M, N, K, Kc, nbatch=1, **kw: True
        F._w3_pays = lambda M, Co, K, Ci, nbatch=1, **kw: True
        x = _rand(2, 9, 11, 64, seed=1)
        w = _rand(128, 64, 3, 3, seed=2, scale=1.0 / math.sqrt(576))
        gy = _rand(2, 9, 11, 128, seed=3)
        for sx, sw, sg in ((1.0, 1.0, 1.0), (1e-30, 1.0, 1e-5), (1e-3, 1e20, 1e-10), (1e15, 1e-20, 1e8)):     # all results representable in fp32
            xr = (x.double() * sx).permute(0, 3, 1, 2).requires_grad_(True)
            wr = (w.double() * sw).requires_grad_(True)
            yr = TF.conv2d(xr, wr, padding=1)
            yr.backward((gy.double() * sg).permute(0, 3, 1, 2))
            xd = (x * sx).to(dev).requires_grad_(True)
            wd = (w * sw).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
            y = F.conv2d(xd, wd, None, pad=1)
            y.backward((gy * sg).to(dev))
            for got, ref, what in ((y.permute(0, 3, 1, 2), yr, "y"), (xd.grad.permute(0, 3, 1, 2), xr.grad, "dx"), (wd.grad, wr.grad, "dw")):
                ref = ref.detach()
                rel = float((got.detach().cpu().double() - ref).abs().max() / ref.abs().max())
                assert rel <= 3e-6, (sx, sw, sg, what, rel)
    finally:
        F.USE_BF16X3, F.SPLIT_SCHEME, F._b3_pays, F._w3_pays = saved



def _row_col_rel(got, ref):
    """worst relative rms error over the rows and over the columns of a 2-D result, each relative to that row's / column's OWN rms"""
    got, ref = got.detach().cpu().double(), ref.detach().cpu().double()
    d = got - ref
    rows = (d.pow(2).mean(1).sqrt() / ref.pow(2).mean(1).sqrt().clamp_min(1e-300)).max().item()
    cols = (d.pow(2).mean(0).sqrt() / ref.pow(2).mean(0).sqrt().clamp_min(1e-300)).max().item()
    return rows, cols


@pytest.mark.parametrize("path", ["f16x2", "bf16x3", "fp32_mfma"])
@pytest.mark.parametrize("k", [3, 1])
def test_conv_gemms_keep_every_row_and_column_over_30_binades(path, k, monkeypatch):
    """operands whose CHANNELS span 2^0 .. 2^-30 in magnitude -- activations and output gradients per channel,
    weights per output AND per input channel.  An output row / column of a GEMM that receives all of its contributions from a small
    slice of an operand (forward: output channel <- weight row; data gradient: input channel <- weight column; weight gradient: dW
    row <- dY channel, dW column <- X channel) must be as exact RELATIVE TO ITS OWN rms as any other: 3e-6, all three back-ends.
    (Round 3's per-tensor scale of the 2xfp16 operands left such slices with few or no significant bits; now: per-row scales for
    weight operands, per-channel scales for activation / gradient operands, absorbed by the weight operand where the GEMM contracts
    over channels.)  Reference semantics: plain fp32 F.conv2d, AiR/models/baseline_attention.py:212-215, 306-309."""
    from scanpaths_amd import functional as F
    monkeypatch.setattr(F, "USE_BF16X3", path != "fp32_mfma")
    monkeypatch.setattr(F, "SPLIT_SCHEME", path if path != "fp32_mfma" else "bf16x3")
    if path != "fp32_mfma":
        monkeypatch.setattr(F, "_b3_pays", lambda M, N, K, Kc, nbatch=1, **kw: nbatch == 1 and Kc % 16 == 0)
        monkeypatch.setattr(F, "_w3_pays", lambda M, Co, K, Ci, nbatch=1, **kw: nbatch == 1 and Ci % 128 == 0 and Co % 32 == 0)
    N, H, W, Ci, Co = 2, 12, 16, 128, 160
    g = np.random.Generator(np.random.PCG64(77))
    span = lambda n: torch.from_numpy(np.exp2(-30.0 * g.permutation(n) / (n - 1)).astype(np.float32))
    fx, fg, fwo, fwi = span(Ci), span(Co), span(Co), span(Ci)
    # dead channels (exact zeros: a ReLU channel that never fires, the gradient columns of a head no sample selected): their scale must
    # not let the weight entries that multiply them dominate the weight rows they are absorbed into
    fx[3] = fx[77] = 0.0
    fg[5] = fg[100] = 0.0
    x = _rand(N, H, W, Ci, seed=1) * fx
    w = _rand(Co, Ci, k, k, seed=2, scale=1.0 / math.sqrt(Ci * k * k)) * fwo.view(Co, 1, 1, 1) * fwi.view(1, Ci, 1, 1)
    gy = _rand(N, H, W, Co, seed=3) * fg
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    yr = TF.conv2d(xr, wr, padding=k // 2)
    yr.backward(gy.double().permute(0, 3, 1, 2))
    dev = _dev()
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    y = F.conv2d(xd, wd, None, pad=k // 2)
    y.backward(gy.to(dev))
    res = {"y": _row_col_rel(y.reshape(-1, Co), yr.permute(0, 2, 3, 1).reshape(-1, Co)),
           "dx": _row_col_rel(xd.grad.reshape(-1, Ci), xr.grad.permute(0, 2, 3, 1).reshape(-1, Ci)),
           "dw": _row_col_rel(wd.grad.permute(0, 2, 3, 1).reshape(Co, -1), wr.grad.permute(0, 2, 3, 1).reshape(Co, -1)),
           # the weight gradient per INPUT channel (its columns over all taps and output channels)
           "dw_ci": _row_col_rel(wd.grad.permute(1, 0, 2, 3).reshape(Ci, -1), wr.grad.permute(1, 0, 2, 3).reshape(Ci, -1))}
    print(f"30-binade spans [{path}, {k}x{k}]: worst (row, column) relative rms error "
          + ", ".join(f"{n} ({a:.1e}, {b:.1e})" for n, (a, b) in res.items()))
    for n, (a, b) in res.items():
        assert a <= 3e-6 and b <= 3e-6, (path, k, n, a, b)


@pytest.mark.parametrize("wo,co", [(64, 256), (20, 256), (64, 128)])
def test_deferred_weight_gradient_of_a_repeated_conv_matches_fp64(wo, co, monkeypatch):
    """F.DeferredWgrad: the weight gradient of a conv applied T times with the same weight (the ConvLSTM's h-gate conv, AiR/models/
    baseline_attention.py:37-56) is computed by ONE launch over all applications at the end of backward (sp_conv_wgrad_f16x2_multi:
    hw2_kernel, 256 x 256 tiles, single-level accumulation, per-application operand scales applied by the slab reduce) -- against
    fp64, with applications whose gradients differ by 2^-20 .. 2^+8 in scale.  (20-pixel-wide map / 128 output channels: the
    kernel's shape constraints do not hold, the recorded applications run one launch each: same result.)"""
    from scanpaths_amd import functional as F
    monkeypatch.setattr(F, "USE_BF16X3", True)
    monkeypatch.setattr(F, "SPLIT_SCHEME", "f16x2")
    monkeypatch.setattr(F, "_b3_pays", lambda M, N, K, Kc, nbatch=1, **kw: nbatch == 1 and Kc % 32 == 0)
    monkeypatch.setattr(F, "_w3_pays", lambda M, Co, K, Ci, nbatch=1, **kw: nbatch == 1 and Ci % 32 == 0 and Co % 32 == 0)
    T, N, H, Ci = 3, 4, 8, 256
    w = _rand(co, Ci, 3, 3, seed=5, scale=1.0 / math.sqrt(9 * Ci))
    xs = [_rand(N, H, wo, Ci, seed=10 + t) * (0.5 + t) for t in range(T)]
    gys = [_rand(N, H, wo, co, seed=20 + t) * s for t, s in zip(range(T), (1.0, 2.0 ** -20, 2.0 ** 8))]
    wr = w.double().requires_grad_(True)
    tot = 0.0
    for x, gy in zip(xs, gys):
        tot = tot + (TF.conv2d(x.double().permute(0, 3, 1, 2), wr, padding=1) * gy.double().permute(0, 3, 1, 2)).sum()
    tot.backward()
    dev = _dev()
    wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    cache = {"defer": F.DeferredWgrad()}
    F.reset_fusion_counts()
    loss = 0.0
    xds = [x.to(dev).requires_grad_(True) for x in xs]
    for xd, gy in zip(xds, gys):
        loss = loss + (F.conv2d(xd, wd, None, pad=1, wcache=cache) * gy.to(dev)).sum()
    loss.backward()
    fits = wo % 32 == 0 and co % 256 == 0
    assert F.FUSION_COUNTS["wgrad_multi"] == int(fits), F.FUSION_COUNTS
    assert not cache["defer"].items
    rel = float((wd.grad.detach().cpu().double() - wr.grad).norm() / wr.grad.norm())
    rows, cols = _row_col_rel(wd.grad.permute(0, 2, 3, 1).reshape(co, -1), wr.grad.permute(0, 2, 3, 1).reshape(co, -1))
    print(f"deferred weight gradient ({'one multi-application launch' if fits else 'one launch per application'}): relative error {rel:.2e}, "
          f"worst row {rows:.2e}, worst column {cols:.2e}")
    assert rel <= 1e-6 and rows <= 3e-6 and cols <= 3e-6, (rel, rows, cols)
    for xd, x, gy in zip(xds, xs, gys):      # the data gradients are untouched by the deferral
        xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
        (TF.conv2d(xr, w.double(), padding=1) * gy.double().permute(0, 3, 1, 2)).sum().backward()
        _close(xd.grad.permute(0, 3, 1, 2), xr.grad, 2e-6, "dx")



@pytest.mark.parametrize("nimg,last", [(5, [0, 3, 7, 1, -1]), (8, [9] * 8), (3, [-1, -1, -1]), (12, [0, 5, 2, 7, 7, 1, 4, 6, 0, 3, 5, 2]),
                                       (64, [(7 * i) % 9 for i in range(64)]), (66, [(5 * i) % 8 for i in range(66)])])
def test_row_sparse_data_gradient_in_live_first_tile_order_equals_the_dense_launch(nimg, last, monkeypatch):
    """h2_kernel<dgrad> with row_last / row_step (masked-step sparsity of the backward pass): the tiles of the samples that
    still receive loss gradient are dealt to the XCDs FIRST and evenly (H2Args::row_nimg), the other workgroups write zero tiles.  Which
    workgroup computes a tile must not change the tile: the result equals, bit for bit, the dense launch on a gradient whose dead
    samples' rows are zero -- 0 live samples, all live, live counts that are not multiples of 4 (no super-tiles), 64 samples (full
    ballot mask), 66 (more than one mask word: the plain order)."""
    from scanpaths_amd import functional as F
    monkeypatch.setattr(F, "USE_BF16X3", True)
    monkeypatch.setattr(F, "SPLIT_SCHEME", "f16x2")
    monkeypatch.setattr(F, "_b3_pays", lambda M, N, K, Kc, nbatch=1, **kw: nbatch == 1 and Kc % 32 == 0)
    dev = _dev()
    H, W, Ci, Co, step = 8, 64, 160, 64, 4                        # 2 tiles of 256 pixels per sample, halo build; 2 N-tiles (ragged)
    w = _rand(Co, Ci, 3, 3, seed=3, scale=1.0 / math.sqrt(9 * Ci)).to(dev)
    wp = w.permute(0, 2, 3, 1).contiguous()
    lastd = torch.tensor(last, dtype=torch.int32, device=dev)
    live = (lastd >= step).float().view(nimg, 1, 1, 1)
    dy = _rand(nimg, H, W, Co, seed=7).to(dev) * live              # dead samples: exactly zero rows (what the recurrence produces)
    x = torch.zeros(nimg, H, W, Ci, device=dev)
    outs = []
    for rows in (None, (F.RowsCtx(lastd), step)):
        dys = F.split_op(dy, channel=True)
        wT = F._weight_operand(wp, dys, {}, transposed=True)
        dx = torch.full_like(x, float("nan"))
        F._igemm_b3(dys, wT, None, dx, N_img=nimg, Hi=H, Wi=W, Kc=Co, ldx=Co, Ho=H, Wo=W, Nout=Ci, ldc=Ci, ldw=9 * Co, KH=3, KW=3,
                    stride=1, pad=1, dil=1, mode=1, beta=0, rows=rows)
        outs.append(dx)
    torch.cuda.synchronize()
    assert torch.isfinite(outs[1]).all()
    assert torch.equal(outs[0], outs[1])
    dead = [i for i, l in enumerate(last) if l < step]
    if dead:
        assert float(outs[1][dead].abs().max()) == 0.0
    ref = TF.conv_transpose2d(dy.cpu().double().permute(0, 3, 1, 2), w.cpu().double(), padding=1).permute(0, 2, 3, 1)
    _close(outs[1], ref, 2e-6, "row-sparse dx")


@pytest.mark.parametrize("batched", [False, True])
def test_deferred_weight_gradient_of_a_repeated_dense_layer(batched):
    """F.DeferredGemmWgrad: a dense layer applied T times with the same weight (spatial_embed / semantic_embed, AiR/models/
    baseline_attention.py:207-208,279-286; the contracted rank-1 filters, :40-50) gets its weight gradient from ONE GEMM over the
    concatenated rows of all applications; inputs, bias and data gradients are untouched."""
    from scanpaths_amd import functional as F
    dev = _dev()
    T, M, K, N, S = 4, 24, 96, 160, 2
    w = _rand(*((S, N, K) if batched else (N, K)), seed=3, scale=0.1)
    xs = [_rand(*((S, M, K) if batched else (M, K)), seed=10 + t) for t in range(T)]
    gs = [_rand(*((S, M, N) if batched else (M, N)), seed=20 + t) * (3.0 ** t) for t in range(T)]
    wr = w.double().requires_grad_(True)
    xr = [x.double().requires_grad_(True) for x in xs]
    sum((x @ wr.transpose(-1, -2) * g.double()).sum() for x, g in zip(xr, gs)).backward()
    wd = w.to(dev).requires_grad_(True)
    xd = [x.to(dev).requires_grad_(True) for x in xs]
    defer = F.DeferredGemmWgrad()
    loss = 0.0
    for x, g in zip(xd, gs):
        loss = loss + (F.gemm(x, wd, None, "nk", defer=defer) * g.to(dev)).sum()
    loss.backward()
    assert not defer.items
    _close(wd.grad, wr.grad, 3e-6, "dW")
    for a, b in zip(xd, xr):
        _close(a.grad, b.grad, 3e-6, "dx")


def test_fused_amax_hints_equal_the_separate_pass():
    """producers (BN apply / backward, LSTM cell forward / backward) leave max|output| behind for the 2xfp16 operand split:
    the hinted split must be bit-identical to the split that runs its own amax pass, and a tensor without a hint still works"""
    from scanpaths_amd import functional as F
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3:
        pytest.skip("2xfp16 back-end not active")
    dev = _dev()
    x = _rand(3, 7, 9, 64, seed=5, scale=3.0).to(dev).requires_grad_(True)
    ga, be = (_rand(64, seed=6) * 0.5 + 1).to(dev).requires_grad_(True), _rand(64, seed=7).to(dev).requires_grad_(True)
    rm, rv = torch.zeros(64, device=dev), torch.ones(64, device=dev)
    y = F.bn_act(x, ga, be, rm, rv, None, True, 0.1, 1e-5, True)
    assert getattr(y, "_sp_amax", None) is not None
    hinted = F.split_op(y)
    plain = F.split_op(y.detach().clone())                       # no hint: separate amax pass
    # train-mode BN leaves an upper BOUND of max|y| (per-channel extrema through the affine map): exact without a residual
    assert torch.equal(hinted.buf, plain.buf) and float(hinted.scale[0]) == float(plain.scale[0])
    amax = torch.tensor([hinted.scale[1].item()]).view(torch.int32).view(torch.float32)     # float bits of the bound
    assert float(amax) == float(y.abs().max())
    # backward hints: dx of BN (a bound: |gamma*invstd| * (max|d| + |k1| + |k2| max|xhat|)), and the LSTM cell pair
    gy = _rand(*y.shape, seed=8).to(dev)
    (dx,) = torch.autograd.grad(y, x, gy)
    if getattr(dx, "_sp_amax", None) is not None:                # identity of the grad tensor object is up to autograd
        bound, true = float(dx._sp_amax[1]), float(dx.abs().max())
        assert true <= bound <= 8 * true, (true, bound)
        a = F.split_op(dx)
        assert float((_decode_split(a, dx.shape) - dx).abs().max()) <= 2.0 ** -21 * bound
    B, Hm, Wm, C = 2, 5, 7, 64
    xg = _rand(B, Hm, Wm, 4 * C, seed=9).to(dev)
    spcol, wc = _rand(B, Hm * Wm, 12, seed=10).to(dev), _rand(B, 3 * C, 12, seed=11, scale=0.1).to(dev)
    h, c = F.lstm_cell_rank1(xg, None, None, spcol, wc)
    a = F.split_op(h); b = F.split_op(h.detach().clone())
    assert getattr(h, "_sp_amax", None) is not None and torch.equal(a.buf, b.buf)


def _decode_split(op, shape):
    """fp32 value of a 2xfp16 split operand [rows][K/16][2][16]"""
    n = 1
    for d in shape:
        n *= d
    planes = op.buf[:2 * n].view(-1, 2, 16).float()
    return ((planes[:, 0] + planes[:, 1]) / float(op.scale[0])).reshape(shape)


@pytest.mark.parametrize("relu", [True, False])
def test_bn_act_emits_the_split_operand_and_the_bit_mask(relu, monkeypatch):
    """train-mode BN with a residual whose passes write the consumer's 2xfp16 operand (forward: of y, backward: of dx) and keep the
    ReLU mask as bits: values against the plain three-kernel path (SP_BN_SPLIT=0) and fp64, operands against their fp32 tensors"""
    from scanpaths_amd import functional as F
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3:
        pytest.skip("2xfp16 back-end not active")
    dev = _dev()
    N, H, W, C = 3, 9, 11, 64                                    # 297 pixels x 64: a ragged last wave for the bit mask
    x0 = (_rand(N, H, W, C, seed=15) * 2 + 0.5).to(dev)
    r0 = _rand(N, H, W, C, seed=16).to(dev)
    gamma, beta = (_rand(C, seed=17).abs() + 0.5).to(dev), _rand(C, seed=18).to(dev)
    gy = _rand(N, H, W, C, seed=21).to(dev)

    def run(split):
        monkeypatch.setattr(F, "BN_SPLIT", split)
        x, r = x0.clone().requires_grad_(True), r0.clone().requires_grad_(True)
        ga, be = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
        rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
        x._sp_from_split = True                                  # as if a split-path conv had produced x: backward emits too
        res = F.bn_act(r, ga, be, torch.zeros(C, device=dev), torch.ones(C, device=dev), None, training=True, relu=False)
        y = F.bn_act(x, ga, be, rm, rv, res, training=True, relu=relu, emit_split=True)
        seen = {}
        x.register_hook(lambda g: seen.setdefault("dx", g))
        y.backward(gy)
        return y, x.grad, r.grad, ga.grad, be.grad, rm, rv, seen.get("dx")

    ys, dxs, drs, dgs, dbs, rms, rvs, dx_obj = run(True)
    assert isinstance(getattr(ys, "_sp_cache", None), dict) and "f16x2" in ys._sp_cache
    yp, dxp, drp, dgp, dbp, rmp, rvp, _ = run(False)
    assert getattr(yp, "_sp_cache", None) is None
    for a, b, tol, what in ((ys, yp, 1e-6, "y"), (dxs, dxp, 1e-5, "dx"), (drs, drp, 1e-5, "dres"), (dgs, dgp, 1e-5, "dgamma"),
                            (dbs, dbp, 1e-5, "dbeta"), (rms, rmp, 1e-6, "rmean"), (rvs, rvp, 1e-6, "rvar")):
        _close(a, b, tol, what)
    op = ys._sp_cache["f16x2"]
    bound, true = float(op.scale[1]), float(ys.abs().max())
    assert true <= bound <= 4 * true, (true, bound)
    assert float((_decode_split(op, ys.shape) - ys.detach()).abs().max()) <= 2.0 ** -21 * bound
    cache = getattr(dx_obj, "_sp_cache", None)
    if cache is not None:                                        # the gradient object as the producing conv's backward sees it
        opd = cache["f16x2"]
        b2, t2 = float(opd.scale[1]), float(dxs.abs().max())
        assert t2 <= b2 <= 8 * t2, (t2, b2)
        assert float((_decode_split(opd, dxs.shape) - dxs).abs().max()) <= 2.0 ** -21 * b2


@pytest.mark.parametrize("k", [3, 1])
def test_conv_epilogue_writes_the_batchnorm_statistics(k, monkeypatch, request):
    """conv2d(bn_stats=True) on the 2xfp16 path (k = 3: channel-block-major schedule; k = 1: the tap-major fallback): per M-tile
    (256 rows) and output column the epilogue leaves sum / sum of squares (fp64) and min / max (fp32) of the conv output -- exactly the first stage of
    bn_pool.hip's statistics; the BatchNorm behind it gives the same result with and without them (ragged last tile: 2 x 63 x 65 =
    8190 pixels)"""
    from scanpaths_amd import functional as F
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3:
        pytest.skip("2xfp16 back-end not active")
    dev = _dev()
    N, H, W, Ci, Co = 2, 63, 65, 128, 160
    TM = 256
    x = _rand(N, H, W, Ci, seed=51).to(dev).requires_grad_(True)
    w = (_rand(Co, Ci, k, k, seed=52, scale=0.05)).to(dev).contiguous(memory_format=torch.channels_last)
    if k == 1:
        monkeypatch.setattr(F, "_b3_pays", lambda M, N, K, Kc, nbatch=1, **kw: True)        # small for the cost model
    assert F.conv_takes_split(x.shape, w, pad=k // 2)
    y = F.conv2d(x, w, None, pad=k // 2, bn_stats=True)
    st = getattr(y, "_sp_bnstats", None)
    assert st is not None and st[2] == (N * H * W + TM - 1) // TM
    y2 = y.detach().reshape(-1, Co)
    xr = x.detach().permute(0, 3, 1, 2).double().cpu().requires_grad_(True)
    yr = TF.conv2d(xr, w.double().cpu(), padding=k // 2)
    _close(y.permute(0, 3, 1, 2), yr, 2e-6, "y")
    gy = _rand(*y.shape, seed=55).to(dev)
    (dx,) = torch.autograd.grad(y, x, gy)                      # k = 1: the pointwise kernel's data-gradient form
    yr.backward(gy.permute(0, 3, 1, 2).double().cpu())
    _close(dx.permute(0, 3, 1, 2), xr.grad, 2e-6, "dx")
    y = y.detach()
    for t in (0, 7, st[2] - 1):
        rows = y2[TM * t:TM * (t + 1)]
        assert torch.equal(st[1][t, 0], rows.min(0).values) and torch.equal(st[1][t, 1], rows.max(0).values)
        _close(st[0][t, 0], rows.double().sum(0), 1e-12, "tile sum")
        _close(st[0][t, 1], (rows.double() ** 2).sum(0), 1e-12, "tile sum of squares")
    ga, be = (_rand(Co, seed=53).abs() + 0.5).to(dev), _rand(Co, seed=54).to(dev)
    outs = []
    for use in (True, False):
        yy = y.detach().clone()
        if use:
            yy._sp_bnstats = st
        rm, rv = torch.zeros(Co, device=dev), torch.ones(Co, device=dev)
        outs.append((F.bn_act(yy, ga, be, rm, rv, None, training=True, relu=True), rm, rv))
    for a, b in zip(outs[0], outs[1]):
        _close(a, b, 1e-6, "bn with / without epilogue statistics")


@pytest.mark.parametrize("S,B,P,C", [(2, 3, 1200, 512), (1, 2, 333, 64), (2, 2, 2560, 512)])
def test_semantic_pool(S, B, P, C):
    """relu(mean_p(a * vf)) and both gradients against fp64"""
    from scanpaths_amd import functional as F
    a, vf = _rand(S, B, P, seed=1).abs(), _rand(B, P, C, seed=2)
    g = _rand(B, S, C, seed=3)
    ar, vr = a.double().requires_grad_(True), vf.double().requires_grad_(True)
    ref = torch.relu(torch.einsum("sbp,bpc->bsc", ar, vr) / P)
    (ref * g.double()).sum().backward()
    dev = _dev()
    ad, vd = a.to(dev).requires_grad_(True), vf.to(dev).requires_grad_(True)
    out = F.semantic_pool(ad, vd)
    (out * g.to(dev)).sum().backward()
    _close(out, ref, 3e-6, "pooled")
    _close(ad.grad, ar.grad, 3e-6, "d amaps")
    _close(vd.grad, vr.grad, 3e-6, "d vf")
    # the [S,B,C] row order the decode loop asks for (no transposed copies): the same numbers, bit for bit
    a2, v2 = a.to(dev).requires_grad_(True), vf.to(dev).requires_grad_(True)
    out2 = F.semantic_pool(a2, v2, sbc=True)
    assert out2.shape == (S, B, C) and torch.equal(out2.transpose(0, 1), out)
    (out2 * g.to(dev).transpose(0, 1)).sum().backward()
    assert torch.equal(a2.grad, ad.grad) and torch.equal(v2.grad, vd.grad)


def test_round6_launch_savers_equal_the_forms_they_replace():
    """three per-decode-step PyTorch copies removed in round 6, each against the form it replaces, bit for bit: the rank-1 filters'
    transposed split in one launch (was: transposed copy + row split); the saliency head's backward reading its logits gradient as
    a slice of the stacked gradient [nh, B, T, 1+P] (was: a contiguous copy of the slice)"""
    from scanpaths_amd import functional as F
    dev = _dev()
    B, N3, KP = 3, 768, 20
    wc = (_rand(B, N3, KP, seed=41) * torch.exp(_rand(B, 1, KP, seed=42))).to(dev)
    new = F._split_wcT(wc)
    old = F.split_w(wc.transpose(1, 2).contiguous().view(B * KP, N3), "f16x2")
    assert torch.equal(new.buf, old.buf) and torch.equal(new.scale, old.scale)
    nh, B, Hm, Wm, T = 2, 3, 10, 16, 4
    HC = 64
    Z = _rand(B, Hm, Wm, nh * 2, seed=43).to(dev)
    cb = _rand(nh, HC, seed=44).to(dev)
    gl = _rand(nh, B, T, Hm * Wm + 1, seed=45).to(dev)
    ga = _rand(nh, B, Hm * Wm, seed=46).to(dev)
    res = []
    for strided in (True, False):
        z, c = Z.clone().requires_grad_(True), cb.clone().requires_grad_(True)
        logits, amap = F.head_sal(z, c, nh, HC, False, per_sample=False)
        gsl = gl[:, :, 2]
        assert not gsl.is_contiguous()
        torch.autograd.backward([logits, amap], [gsl if strided else gsl.contiguous(), ga])
        res.append((z.grad, c.grad))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])


def test_batched_pointwise_gemm_with_per_item_weights_on_wider_rows():
    """the cell backward's spatial-tap gradient (AiR/models/baseline_attention.py:37-56, backward of ss+- x conv_pos/neg): a batched
    GEMM out[b] = X[b][:, :Kc] x W[b]^T through sp_conv_igemm_f16x2 -- one weight set per batch item, X rows wider than Kc (the
    first 3C of the 4C gate-gradient channels) -- against fp64; and the error codes of the shapes it does not take"""
    import ctypes as C
    from scanpaths_amd import functional as F, hip
    dev = _dev()
    g = torch.Generator().manual_seed(3)
    B, P, Kc, ldx, N = 3, 512, 96, 128, 20
    x = torch.randn(B, P, ldx, generator=g).to(dev)
    w = (torch.randn(B, N, Kc, generator=g) * 0.1).to(dev)
    xs, ws = F.split_op(x, "f16x2"), F.split_op(w, "f16x2")
    out = torch.full((B, P, N), float("nan"), device=dev)
    d = hip.ConvDesc(P, 1, 1, Kc, ldx, 1, 1, N, N, 1, 1, 1, 0, 1, 0, Kc, 1.0, 0, 0, B, P * ldx, N * Kc, P * N, 0, None)
    L = hip.lib()
    args = (hip.ptr(xs.buf), hip.ptr(xs.scale), hip.ptr(ws.buf), hip.ptr(ws.scale), None, hip.ptr(out), hip.stream())
    assert L.sp_conv_igemm_f16x2(C.byref(d), *args) == 0
    ref = torch.einsum("bpk,bnk->bpn", x[:, :, :Kc].double().cpu(), w.double().cpu())
    err = (out.double().cpu() - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-6, err
    for bad in (dict(P=500), dict(sW=N * Kc + 16), dict(ldx=Kc - 16)):      # partial tiles per item / padded weight sets / ldx < Kc
        Pb, sW, lx = bad.get("P", P), bad.get("sW", N * Kc), bad.get("ldx", ldx)
        db = hip.ConvDesc(Pb, 1, 1, Kc, lx, 1, 1, N, N, 1, 1, 1, 0, 1, 0, Kc, 1.0, 0, 0, B, Pb * lx, sW, Pb * N, 0, None)
        assert L.sp_conv_igemm_f16x2(C.byref(db), *args) == -1, bad


def test_batched_weight_gradient_with_padded_taps_on_wider_rows():
    """the cell backward's rank-1 filter gradient: one TN GEMM per sample, dwc[b] = dY[b][:, :Co]^T x X[b], through
    sp_conv_wgrad_f16x2 with nbatch > 1 -- dY rows wider than Co, X's KP columns padded to 32, only the first KP output columns
    stored (ldo = KP) -- against fp64, and the canary behind every item's result stays untouched"""
    import ctypes as C
    from scanpaths_amd import functional as F, hip
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    B, P, Co, ldy, KP = 3, 512, 80, 128, 20
    dy = torch.randn(B, P, ldy, generator=g).to(dev)
    x = torch.randn(B, P, KP, generator=g).to(dev)
    ys, xs = F.split_op(dy, "f16x2"), F.split_op(torch.nn.functional.pad(x, (0, 32 - KP)), "f16x2")
    out = torch.full((B * Co * KP + 64,), 7.0, device=dev)
    d = hip.WgradDesc(1, P // 64, 64, 32, 32, P // 64, 64, Co, ldy, 1, 1, 1, 0, 1, KP, 0, 1.0, B, P * 32, P * ldy, Co * KP)
    L = hip.lib()
    assert L.sp_conv_wgrad_f16x2_workspace(C.byref(d)) == 0
    assert L.sp_conv_wgrad_f16x2(C.byref(d), hip.ptr(xs.buf), hip.ptr(xs.scale), hip.ptr(ys.buf), hip.ptr(ys.scale), hip.ptr(out), None,
                                 hip.stream()) == 0
    ref = torch.einsum("bpc,bpk->bck", dy[:, :, :Co].double().cpu(), x.double().cpu())
    got = out[:B * Co * KP].view(B, Co, KP).double().cpu()
    err = (got - ref).abs().max().item() / ref.abs().max().item()
    assert err < 1e-6, err
    assert bool((out[B * Co * KP:] == 7.0).all())


def test_fan_in_of_split_only_gradients_and_cell_backward_without_fp32_output():
    """the x-gate fan-in when steps hand back their gate gradient as a split operand only (sp_sum_n_mixed; the fp32 tensor of such
    a contribution is poisoned here and must not be read), and sp_lstm_pointwise_bwd_split with dpre == NULL writing the same
    split operand as with it"""
    import ctypes as C
    from scanpaths_amd import functional as F, hip
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    n = 4096 * 16
    ts = [(torch.randn(n, generator=g) * s).to(dev) for s in (1.0, 30.0, 0.01)]
    ops = [F.split_op(t.view(-1, 256), "f16x2") for t in ts]
    poison = torch.full((n,), float("nan"), device=dev)
    f = (C.c_void_p * 3)(None, ts[1].data_ptr(), None)
    pl = (C.c_void_p * 3)(ops[0].buf.data_ptr(), None, ops[2].buf.data_ptr())
    sc = (C.c_void_p * 3)(ops[0].scale.data_ptr(), None, ops[2].scale.data_ptr())
    out = torch.empty(n, device=dev)
    amax = torch.zeros(2, dtype=torch.int32, device=dev)
    L = hip.lib()
    assert L.sp_sum_n_mixed(f, pl, sc, 3, n, hip.ptr(out), hip.ptr(amax), hip.stream()) == 0
    ref = ts[0].double() + ts[1].double() + ts[2].double()
    err = (out.double() - ref).abs().max().item()
    assert err <= 2e-6 * 30.0 * 5, err                         # split representation: 2^-22 of each tensor's maximum
    assert float(amax.view(torch.float32)[0]) == float(out.abs().max())                      # fused max|sum| (float bits)
    assert L.sp_sum_n_mixed(f, pl, sc, 3, n - 8, hip.ptr(out), None, hip.stream()) == -1
    del poison
    # masked-step sparsity (sp_sum_n_mixed_rows): a term of decode step s is exactly zero for samples whose last loss step is < s and
    # is then not read -- same result, bit for bit, as the dense pass over terms whose dead samples hold zeros (4 samples here)
    nb = 4
    last = torch.tensor([0, 2, 1, -1], dtype=torch.int32, device=dev)
    steps = [0, 1, 2]
    zs = []
    for t, st in zip(ts, steps):
        z = t.clone().view(nb, -1)
        z[(last < st).nonzero().flatten()] = 0.0
        zs.append(z.view(-1))
    zops = [F.split_op(z.view(-1, 256), "f16x2") for z in zs]
    f2 = (C.c_void_p * 3)(None, zs[1].data_ptr(), None)
    pl2 = (C.c_void_p * 3)(zops[0].buf.data_ptr(), None, zops[2].buf.data_ptr())
    sc2 = (C.c_void_p * 3)(zops[0].scale.data_ptr(), None, zops[2].scale.data_ptr())
    dense, sparse = torch.empty(n, device=dev), torch.empty(n, device=dev)
    assert L.sp_sum_n_mixed(f2, pl2, sc2, 3, n, hip.ptr(dense), None, hip.stream()) == 0
    for z, st in zip(zs, steps):                               # dead samples now hold NaN: reading them would show
        z.view(nb, -1)[(last < st).nonzero().flatten()] = float("nan")
    st_arr = (C.c_int * 3)(*steps)
    assert L.sp_sum_n_mixed_rows(f2, pl2, sc2, 3, n, hip.ptr(sparse), None, hip.ptr(last), st_arr, nb, hip.stream()) == 0
    assert torch.equal(dense, sparse)
    assert L.sp_sum_n_mixed_rows(f2, pl2, sc2, 3, n, hip.ptr(sparse), None, hip.ptr(last), None, nb, hip.stream()) == -2
    # cell backward: same planes with and without the fp32 output
    rows, Cc = 64, 256
    gates = torch.rand(rows, 4 * Cc, generator=g).to(dev)
    c_prev, c = torch.randn(rows, Cc, generator=g).to(dev), torch.randn(rows, Cc, generator=g).to(dev)
    dh, dc = torch.randn(rows, Cc, generator=g).to(dev), torch.randn(rows, Cc, generator=g).to(dev)
    am = lambda t: torch.tensor([0.0, float(t.abs().max())], device=dev).view(torch.int32)
    res = []
    for skip in (False, True):
        dpre, dcp = torch.empty_like(gates), torch.empty_like(c)
        planes = torch.empty(2 * gates.numel() + 32, dtype=torch.float16, device=dev)
        scale = torch.zeros(2, device=dev)
        dh_a, dc_a = am(dh), am(dc)
        rc = L.sp_lstm_pointwise_bwd_split(hip.ptr(dh), hip.ptr(dc), hip.ptr(gates), hip.ptr(c_prev), hip.ptr(c), rows, Cc,
                                           None if skip else hip.ptr(dpre), hip.ptr(dcp), None, None, hip.ptr(dh_a[1:]), hip.ptr(dc_a[1:]),
                                           4.0, 3.0, hip.ptr(planes), hip.ptr(scale), hip.stream())
        assert rc == 0
        res.append((planes.clone(), dcp.clone(), scale.clone()))
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1]) and torch.equal(res[0][2], res[1][2])



def test_fan_in_reads_split_only_contributions_from_the_alias_record_not_from_tensor_identity():
    """a consumer that leaves the fp32 form of its gradient unwritten records the split operand under its alias
    index (F.fanout's token); the fan-in must take that contribution from the record even when autograd hands it ANOTHER tensor object
    than the consumer returned (a hook that clones: attributes such as _sp_skipped are lost), and must refuse an unwritten gradient
    that arrives without a record."""
    from scanpaths_amd import functional as F
    dev = _dev()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(64, 256, generator=g).to(dev).requires_grad_(True)
    true_g = [torch.randn(64, 256, generator=g).to(dev) * s for s in (1.0, 7.0)]

    class SplitOnly(torch.autograd.Function):
        @staticmethod
        def forward(ctx, a, k, record):
            ctx.k, ctx.fan, ctx.record = k, a._sp_fan, record
            return a.clone()

        @staticmethod
        def backward(ctx, gout):
            poisoned = torch.full_like(gout, float("nan"))          # the "unwritten" fp32 gradient
            poisoned._sp_skipped = True
            if ctx.record:
                ctx.fan[0][ctx.fan[1]] = F.split_op(true_g[ctx.k], "f16x2")
            return poisoned, None, None

    a, b = F.fanout(x, 2)
    a.register_hook(lambda t: t.clone())                             # identity (and the attribute) lost on the way to the fan-in
    ya, yb = SplitOnly.apply(a, 0, True), b * 2.0
    (ya.sum() + (yb * true_g[1]).sum()).backward()
    ref = true_g[0].double() + 2.0 * true_g[1].double()
    assert torch.isfinite(x.grad).all()
    assert (x.grad.double() - ref).abs().max().item() <= 4e-6 * float(ref.abs().max())
    x2 = x.detach().clone().requires_grad_(True)
    a2, b2 = F.fanout(x2, 2)
    with pytest.raises(RuntimeError, match="without its record"):
        (SplitOnly.apply(a2, 0, False).sum() + b2.sum()).backward()


def test_product_library_has_no_timing_modes():
    """the shipped library cannot be switched into a wrong-result timing mode or another kernel schedule -- those
    selectors exist only in libscanpaths_amd_timing.so; the one process-wide switch left is "amax_reset"."""
    from scanpaths_amd import hip
    L = hip.lib()
    assert L.sp_timing_build() == 0 and hip.LIB_PATH.endswith("libscanpaths_amd.so")
    for name in (b"h2_dbg", b"hw_dbg", b"b3_dbg", b"hw_splits", b"h2_halo", b"h2_variant", b"nonsense"):
        assert L.sp_set_tuning(name, 1) == -1, name
    assert L.sp_set_tuning(b"amax_reset", 1) == 0


def test_c_abi_error_codes():
    """error behaviour of the C ABI: 0 = ok, SP_EINVAL (-1) for unsupported shapes, SP_ENULL (-2) for missing buffers -- never a
    crash, never a silent fallback; the Python layer turns them into exceptions (hip.check)"""
    import ctypes as C
    from scanpaths_amd import hip
    L = hip.lib()
    dev = _dev()
    x = torch.zeros(64, device=dev)
    st = hip.stream()
    assert L.sp_split2_f16(None, 64, hip.ptr(x), hip.ptr(x), 0, st) == -2
    assert L.sp_split2_f16(hip.ptr(x), 60, hip.ptr(x), hip.ptr(x), 0, st) == -1          # row length not a multiple of 16
    d = hip.ConvDesc(1, 4, 4, 48, 48, 4, 4, 64, 64, 1, 1, 1, 0, 1, 0, 48, 1.0, 0, 0, 1, 0, 0, 0, 0, None)
    assert L.sp_conv_igemm_f16x2(C.byref(d), hip.ptr(x), hip.ptr(x), hip.ptr(x), hip.ptr(x), None, hip.ptr(x), st) == -1   # Kc % 32
    assert L.sp_conv_igemm_f16x2(C.byref(d), None, hip.ptr(x), hip.ptr(x), hip.ptr(x), None, hip.ptr(x), st) == -2
    assert L.sp_lstm_rank1_fwd(hip.ptr(x), None, None, hip.ptr(x), hip.ptr(x), 1, 4, 48, 12, hip.ptr(x), hip.ptr(x), hip.ptr(x), None,
                               st) == -1                                                  # C % 64
    assert L.sp_sempool_fwd(hip.ptr(x), hip.ptr(x), 3, 1, 4, 64, 1.0, hip.ptr(x), hip.ptr(x), st) == -1      # S > 2
    assert L.sp_scanmatch_submatrix(0, 3, 3.5, hip.ptr(x), hip.ptr(x), st) == -1
    assert L.sp_head_num_classes(3, 3) >= 1 and L.sp_head_num_classes(100000, 8) == -1
    with pytest.raises(hip.HipError):
        hip.check(-1, "demo")
    with pytest.raises(hip.HipError):
        hip.ptr(torch.zeros(4))                                                            # CPU tensor: no CPU path


# (the two non-default back-ends at the FULL benchmark size run with SP_ALL_GPU_TESTS=1: test_conv2d_fwd_bwd and the tame model goldens hold
# all three back-ends to the same bars on every run; 12 s each of the suite's 650-s budget)
_all_tests = pytest.mark.skipif(not __import__("os").environ.get("SP_ALL_GPU_TESTS"), reason="non-default back-end at full size; set SP_ALL_GPU_TESTS=1")


@pytest.mark.parametrize("path", ["f16x2", pytest.param("bf16x3", marks=_all_tests), pytest.param("fp32_mfma", marks=_all_tests)])
def test_hgate_conv_at_benchmark_size_vs_fp64(path, monkeypatch):
    """The dominant GEMM of the bench line AT ITS SIZE (BASELINE.json config 2: bs 32, 40x64 map): h-gate conv 3x3 512->2048,
    implicit GEMM M = 81 920, N = 2048, K = 4608 -- forward, data gradient and weight gradient of every back-end against a
    row-/entry-subsampled fp64 reference on the host (AiR/models/baseline_attention.py:44-50 as hoisted in scanpath_model.py).
    Bars (rms error relative to rms of the exact result; a CPU fp32 GEMM of such data sits at 2.4e-7 / 2.9e-7 / 8.4e-7,
    DESIGN.md §5c): 1e-6 forward / data gradient, 2e-6 weight gradient (81 920-term sums); max error <= 8x the rms bar."""
    import json
    import os
    from scanpaths_amd import functional as F
    monkeypatch.setattr(F, "USE_BF16X3", path != "fp32_mfma")
    monkeypatch.setattr(F, "SPLIT_SCHEME", path if path != "fp32_mfma" else "f16x2")
    B, Hm, Wm, Ci, Co = 32, 40, 64, 512, 2048
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(5)
    # decoder-shaped data: h = o * c in (-T, T) with many small entries, weights fan-in scaled, gate gradients heavy-tailed
    h = (torch.randn(B, Hm, Wm, Ci, generator=g) * torch.rand(B, Hm, Wm, Ci, generator=g)).float()
    w = (torch.randn(Co, 3, 3, Ci, generator=g) / math.sqrt(9 * Ci)).float()               # physical [Co][KH][KW][Ci]
    dy = (torch.randn(B, Hm, Wm, Co, generator=g) * torch.rand(B, Hm, Wm, 1, generator=g) ** 4 * 1e-3).float()
    hg = h.to(dev).requires_grad_(True)
    wg = w.to(dev).permute(0, 3, 1, 2).requires_grad_(True)                               # logical OIHW, channels_last memory
    y = F.conv2d(hg, wg, None, pad=1)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    y_c, dx_c, dw_c = y.detach().cpu(), hg.grad.cpu(), wg.grad.permute(0, 2, 3, 1).contiguous().cpu()      # dw as [Co][KH][KW][Ci]

    rs = np.random.Generator(np.random.PCG64(9))
    pix = [(int(rs.integers(B)), int(rs.integers(Hm)), int(rs.integers(Wm))) for _ in range(96)]
    pix += [(0, 0, 0), (B - 1, Hm - 1, Wm - 1), (3, 0, Wm - 1), (7, Hm - 1, 0)]            # zero-padding corners
    hd, wd, dyd = h.double(), w.double(), dy.double()

    def patch(src, b, yy, xx, sign):        # [3,3,C] window around (yy, xx) with zero padding; sign=-1 mirrors for the data gradient
        out = torch.zeros(3, 3, src.shape[-1], dtype=torch.float64)
        for ky in range(3):
            for kx in range(3):
                iy, ix = yy + sign * (ky - 1), xx + sign * (kx - 1)
                if 0 <= iy < Hm and 0 <= ix < Wm:
                    out[ky, kx] = src[b, iy, ix]
        return out

    res = {}
    # forward: y[b,y,x,co] = sum_{ky,kx,ci} h[b,y+ky-1,x+kx-1,ci] w[co,ky,kx,ci]
    ref = torch.stack([torch.einsum("yxc,oyxc->o", patch(hd, *p, 1), wd) for p in pix])
    got = torch.stack([y_c[p] for p in pix]).double()
    res["fwd"] = ((got - ref).pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item(),
                  (got - ref).abs().max().item() / ref.pow(2).mean().sqrt().item())
    # data gradient: dx[b,y,x,ci] = sum_{ky,kx,co} dy[b,y-(ky-1),x-(kx-1),co] w[co,ky,kx,ci]
    ref = torch.stack([torch.einsum("yxo,oyxc->c", patch(dyd, *p, -1), wd) for p in pix])
    got = torch.stack([dx_c[p] for p in pix]).double()
    res["dgrad"] = ((got - ref).pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item(),
                    (got - ref).abs().max().item() / ref.pow(2).mean().sqrt().item())
    # weight gradient on a 48 x 48 block of (co, ci) for all 9 taps: dw[co,ky,kx,ci] = sum_p dy[p,co] h[p+(ky-1,kx-1),ci]
    cos = torch.from_numpy(rs.choice(Co, 48, replace=False)).long()
    cis = torch.from_numpy(rs.choice(Ci, 48, replace=False)).long()
    dys = dyd[..., cos]                                                    # [B,Hm,Wm,48]
    hp = torch.nn.functional.pad(hd[..., cis], (0, 0, 1, 1, 1, 1))         # zero-pad W and H by 1
    ref = torch.empty(48, 3, 3, 48, dtype=torch.float64)
    for ky in range(3):
        for kx in range(3):
            ref[:, ky, kx, :] = torch.einsum("bhwo,bhwc->oc", dys, hp[:, ky:ky + Hm, kx:kx + Wm])
    got = dw_c[cos][..., cis].double()
    res["wgrad"] = ((got - ref).pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item(),
                    (got - ref).abs().max().item() / ref.pow(2).mean().sqrt().item())
    print(f"h-gate conv M=81920 N=2048 K=4608 [{path}]: " + "  ".join(f"{k}: rms {v[0]:.2e} max {v[1]:.2e}" for k, v in res.items()))
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        d = os.path.join(root, "gpurun_out", "parity")
        os.makedirs(d, exist_ok=True)
        fn = os.path.join(d, "r03_hgate_fullsize_errors.json")
        old = json.load(open(fn)) if os.path.exists(fn) else {}
        old[path] = {k: {"rms_rel": v[0], "max_rel": v[1]} for k, v in res.items()}
        json.dump(old, open(fn, "w"), indent=1)
    except OSError:
        pass
    for k, bar in (("fwd", 1e-6), ("dgrad", 1e-6), ("wgrad", 2e-6)):
        assert res[k][0] <= bar and res[k][1] <= 8 * bar, (path, k, res[k])


def test_fused_gateconv_lstm_at_benchmark_size_vs_fp64():
    """The kernel 15 of the 16 forward h-gate launches of the bench line run -- the ConvLSTM cell as the epilogue of the h-gate conv
    (sp_gateconv_lstm_f16x2; M = 81 920 pixels, C = 512, N = 2048 gate columns, K = 4608) -- AT ITS SIZE against a row-subsampled
    fp64 evaluation of AiR/models/baseline_attention.py:37-56 in the hoisted form of scanpath_model.py: gates, c' = f c + i g,
    h' = o c', and h's split operand written by the same epilogue."""
    import json
    import os
    from scanpaths_amd import functional as F
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3 or F.THROUGHPUT_MODE:
        pytest.skip("2xfp16 back-end not active")
    B, Hm, Wm, C, KP, S = 32, 40, 64, 512, 20, 2
    P = Hm * Wm
    dev = _dev()
    g = torch.Generator(device="cpu").manual_seed(15)
    h = (torch.randn(B, Hm, Wm, C, generator=g) * torch.rand(B, Hm, Wm, C, generator=g)).float()        # h = o * c shaped
    c = (torch.randn(B, Hm, Wm, C, generator=g) * 1.5).float()
    xg = torch.randn(B, Hm, Wm, 4 * C, generator=g).float()
    w = (torch.randn(4 * C, 3, 3, C, generator=g) / math.sqrt(9 * C)).float()                            # physical [Co][KH][KW][Ci]
    spcol = torch.rand(B, P, KP, generator=g).float()
    spcol[..., 9 * S:] = 0
    wc = (torch.randn(B, 3 * C, KP, generator=g) * 0.1).float()
    hg, cg = h.to(dev), c.to(dev)
    cg._sp_cbound = float(c.abs().max()) + 0.5                                                           # -> the epilogue writes h's planes
    wg = w.to(dev).permute(0, 3, 1, 2).requires_grad_(True)
    assert F.gateconv_lstm_fusable(hg, wg, spcol.to(dev))
    F.reset_fusion_counts()
    hn, cn = F.gateconv_lstm(hg, wg, xg.to(dev), cg, spcol.to(dev), wc.to(dev), {})
    torch.cuda.synchronize()
    assert F.FUSION_COUNTS["gateconv_lstm"] == 1 and F.FUSION_COUNTS["gateconv_lstm_hplanes"] == 1
    gates = hn.grad_fn.saved_tensors[0]
    assert gates.shape == (B, Hm, Wm, 4 * C)
    op = hn._sp_cache["f16x2"]
    bound = float(op.scale[1])
    assert float(hn.abs().max()) <= bound
    assert float((_decode_split(op, hn.shape) - hn.detach()).abs().max()) <= 2.0 ** -21 * bound
    h_c, c_c, g_c = hn.detach().cpu(), cn.detach().cpu(), gates.cpu()

    rs = np.random.Generator(np.random.PCG64(19))
    pix = [(int(rs.integers(B)), int(rs.integers(Hm)), int(rs.integers(Wm))) for _ in range(120)]
    pix += [(0, 0, 0), (B - 1, Hm - 1, Wm - 1), (5, 0, Wm - 1), (9, Hm - 1, 0), (17, 3, 63), (17, 4, 0)]   # padding corners, tile seams
    hd, wd = h.double(), w.double()
    ref_g, ref_c, ref_h = [], [], []
    for (b, y, x) in pix:
        win = torch.zeros(3, 3, C, dtype=torch.float64)
        for ky in range(3):
            for kx in range(3):
                iy, ix = y + ky - 1, x + kx - 1
                if 0 <= iy < Hm and 0 <= ix < Wm:
                    win[ky, kx] = hd[b, iy, ix]
        pre = xg[b, y, x].double() + torch.einsum("yxc,oyxc->o", win, wd)
        pre[:3 * C] += wc[b].double() @ spcol[b, y * Wm + x].double()
        i, f, o, gg = pre[:C].sigmoid(), pre[C:2 * C].sigmoid(), pre[2 * C:3 * C].sigmoid(), pre[3 * C:].tanh()
        cc = f * c[b, y, x].double() + i * gg
        ref_g.append(torch.cat([i, f, o, gg]))
        ref_c.append(cc)
        ref_h.append(o * cc)
    res = {}
    for name, got, ref in (("gates", torch.stack([g_c[p] for p in pix]), torch.stack(ref_g)),
                           ("c", torch.stack([c_c[p] for p in pix]), torch.stack(ref_c)),
                           ("h", torch.stack([h_c[p] for p in pix]), torch.stack(ref_h))):
        d = got.double() - ref
        res[name] = (d.pow(2).mean().sqrt().item() / ref.pow(2).mean().sqrt().item(), d.abs().max().item() / ref.pow(2).mean().sqrt().item())
    print("fused gate conv + cell M=81920 C=512: " + "  ".join(f"{k}: rms {v[0]:.2e} max {v[1]:.2e}" for k, v in res.items()))
    root = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    try:
        d = os.path.join(root, "gpurun_out", "parity")
        os.makedirs(d, exist_ok=True)
        json.dump({k: {"rms_rel": v[0], "max_rel": v[1]} for k, v in res.items()},
                  open(os.path.join(d, "r03_fused_cell_fullsize_errors.json"), "w"), indent=1)
    except OSError:
        pass
    for k in res:
        assert res[k][0] <= 1e-6 and res[k][1] <= 8e-6, (k, res[k])


def test_rows_last_reads_the_horizon_off_the_output_gradients():
    """sp_rows_last (functional._OutputGate): last[b] = the last decode step at which ANY element of ANY output gradient of sample b
    is non-zero -- over both stacked heads, rows of different widths ([.., A] logits, scalar mu / sigma2, [.., P] action maps), with
    NaN counting as non-zero, -0.0 as zero, a gradient that never arrived (None) as zero, and -1 for a sample without any.  The
    reference multiplies the zeros these rows imply (AiR/models/loss.py:10-14,27-32); the gate lets the backward kernels skip them."""
    from scanpaths_amd import functional as F
    dev = _dev()
    nst, B, T, A, P = 2, 7, 6, 37, 24
    g = np.random.Generator(np.random.PCG64(5))
    z = torch.zeros(nst, B, T, A)
    mu = torch.zeros(nst, B, T)
    s2 = torch.zeros(nst, B, T)
    am = torch.zeros(nst, B, T, P)
    want = [-1] * B
    z[1, 0, 2, 36] = 1e-30; want[0] = 2                        # a single tiny element of the second head, last column
    mu[0, 1, 5] = -2.0; z[0, 1, 1, 0] = 3.0; want[1] = 5       # the scalar output decides
    am[1, 2, 0, 7] = float("nan"); want[2] = 0                 # NaN is a non-zero gradient
    z[0, 3, 4, 3] = -0.0; s2[1, 3, 3] = 1.0; want[3] = 3       # -0.0 is zero
    z[:, 4] = torch.from_numpy(g.standard_normal((nst, T, A)).astype(np.float32)); want[4] = T - 1
    am[0, 6, 4, 23] = 5.0; want[6] = 4                         # sample 5: nothing at all
    tok = F.DecodeRows()
    outs = [t.to(dev).requires_grad_(True) for t in (z * 0, am * 0, mu * 0, s2 * 0)]
    gated = F.output_gate(tok, outs)
    torch.autograd.backward([gated[0], gated[1], gated[2], gated[3]], [z.to(dev), am.to(dev), mu.to(dev), s2.to(dev)])
    assert tok.rc.last.tolist() == want, (tok.rc.last.tolist(), want)
    for o, ref in zip(outs, (z, am, mu, s2)):                  # the gate is an identity for the gradients themselves
        assert torch.equal(torch.nan_to_num(o.grad.cpu(), nan=7.0), torch.nan_to_num(ref, nan=7.0))
    # a gradient that never arrived: only the action maps are consumed (None for the other three)
    tok2 = F.DecodeRows()
    outs2 = [t.to(dev).requires_grad_(True) for t in (z * 0, am * 0, mu * 0, s2 * 0)]
    g2 = F.output_gate(tok2, outs2)
    am2 = torch.zeros(nst, B, T, P)
    am2[0, 5, 2, 0] = 1.0
    (g2[1] * am2.to(dev)).sum().backward()
    assert tok2.rc.last.tolist() == [-1, -1, -1, -1, -1, 2, -1] and outs2[0].grad is None


@pytest.mark.parametrize("KP,B,P,Cc", [(20, 3, 256, 256), (12, 2, 512, 256), (20, 2, 2560, 512)])
def test_rank1_grads_kernel_gives_both_gradients_of_the_rank1_gate_term(KP, B, P, Cc):
    """csrc/rank1_grads.hip: dsp[b] = dpre[b][:, :3C] x wc[b] and dwc[b] = dpre[b][:, :3C]^T x spcol[b] from ONE pass over the 2xfp16
    split gate gradient -- against fp64 on the operand the kernel actually reads (decoded planes: the split itself is tested
    elsewhere), against the two GEMM launches it replaces, with dead samples (row_last) giving exact zeros and run-to-run identical.
    Reference semantics: the rank-1 gate term of AiR/models/baseline_attention.py:40-50 (conv of the spatial memory per sample)."""
    from scanpaths_amd import functional as F, hip
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3:
        pytest.skip("2xfp16 back-end not active")
    dev, L = _dev(), hip.lib()
    C4, N3 = 4 * Cc, 3 * Cc
    assert L.sp_rank1_grads_applies(B, P, N3, KP, C4) == 1
    assert L.sp_rank1_grads_applies(B, P + 8, N3, KP, C4) == 0 and L.sp_rank1_grads_applies(B, P, N3, 16, C4) == 0
    dpre = (_rand(B, P, C4, seed=31) * torch.exp(_rand(B, P, 1, seed=32))).to(dev)       # rows of different magnitudes
    spcol, wc = _rand(B, P, KP, seed=33).to(dev), _rand(B, N3, KP, seed=34, scale=0.2).to(dev)
    ys = F.split_op(dpre)
    ws = F.split_w(wc.transpose(1, 2).contiguous().view(B * KP, N3), "f16x2")
    yd = _decode_split(ys, dpre.shape).double()[:, :, :N3]
    assert float((yd - dpre[:, :, :N3].double()).abs().max()) <= 2.0 ** -20 * float(dpre.abs().max())
    ref_dsp, ref_dwc = torch.bmm(yd, wc.double()), torch.bmm(yd.transpose(1, 2), spcol.double())
    wsp = torch.empty(L.sp_rank1_grads_workspace(B, P, N3, KP), dtype=torch.uint8, device=dev)

    def run(last=None, step=0):
        dsp, dwc = torch.full_like(spcol, float("nan")), torch.full_like(wc, float("nan"))
        F.check(L.sp_rank1_grads_f16x2(hip.ptr(ys.buf), hip.ptr(ys.scale), C4, hip.ptr(ws.buf), hip.ptr(ws.scale), hip.ptr(spcol), B, P, N3, KP,
                                       hip.ptr(dsp), hip.ptr(dwc), hip.ptr(wsp), hip.ptr(last) if last is not None else None, step,
                                       hip.stream()), "sp_rank1_grads_f16x2")
        torch.cuda.synchronize()
        return dsp, dwc
    dsp, dwc = run()
    # dsp: 3-product split GEMM (the weight operand is split per row: 2^-21 relative of the row's largest product sum); dwc: fp32
    # accumulation over P pixels of exact operands
    e_dsp = float((dsp.double() - ref_dsp).abs().max()) / float(ref_dsp.abs().max())
    e_dwc = float((dwc.double() - ref_dwc).abs().max()) / float(ref_dwc.abs().max())
    print(f"rank1_grads KP={KP} P={P} C={Cc}: dsp rel err {e_dsp:.2e}  dwc rel err {e_dwc:.2e}")
    assert e_dsp <= 4e-6 and e_dwc <= 4e-6, (e_dsp, e_dwc)
    d2, w2 = run()
    assert torch.equal(d2, dsp) and torch.equal(w2, dwc)
    # dead samples: sample 0 has no loss gradient at step 5 (its planes are not read: poison them)
    last = torch.tensor([2] + [9] * (B - 1), dtype=torch.int32, device=dev)
    keep = ys.buf[:2 * P * C4].clone()
    ys.buf[:2 * P * C4] = float("nan")
    d3, w3 = run(last, 5)
    ys.buf[:2 * P * C4] = keep
    assert float(d3[0].abs().max()) == 0.0 and float(w3[0].abs().max()) == 0.0
    assert torch.equal(d3[1:], dsp[1:]) and torch.equal(w3[1:], dwc[1:])
