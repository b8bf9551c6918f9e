"""HIP ops (through the C ABI) against goldens of the REAL reference MODULES (tests/golden/modules.npz from
tests/golden/make_golden_modules.py: ConvLSTM.forward AiR/models/baseline_attention.py:37-56, conv 5x5 -> predict_head :306-309 +
:149-174, spatial_att :111-124, semantic_att :77-88, dilated Bottleneck AiR/models/resnet.py:57-93) -- SURVEY.md §8c fixture plan (iv).

The op tests of tests/test_ops_gpu.py compare with test-authored fp64 restatements; these compare the SAME kernels, composed the way
scanpaths_amd.models.scanpath_model composes them, with what the reference's own module returned on the same bytes, so a kernel-level
failure is attributable to the reference rather than to a restatement.  Bar per array: max(TOL x scale, 8 x the reference's own
fp32-vs-fp64 distance stored with the golden), TOL = 1e-5 (outputs) / 3e-5 (gradients, norms relative)."""
import os

import numpy as np
import pytest
import torch

from module_cases import MODULE_CASES, case_inputs, cotangents

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
G = np.load(os.path.join(os.path.dirname(__file__), "golden", "modules.npz"))


def _params(name, spec):
    from scanpaths_amd.procedural import procedural_state_dict
    sd = procedural_state_dict(spec, MODULE_CASES[name]["seed"])
    return {k: v.to(DEV) for k, v in sd.items()}


def _leaf(t, channels_last=False):
    t = t.to(DEV).to(torch.float32)
    if channels_last and t.dim() == 4:
        t = t.contiguous(memory_format=torch.channels_last)
    return t.requires_grad_(True)


def _nhwc(t):          # reference NCHW (float64, CPU) -> NHWC fp32 leaf on the device
    return t.permute(0, 2, 3, 1).contiguous().to(DEV).to(torch.float32).requires_grad_(True)


def _check(name, key, got, tol, to_ref=lambda a: a):
    ref = G[f"{name}/ref64/{key}"]
    got = to_ref(got.detach().double().cpu()).numpy().reshape(ref.shape)
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(got - ref).max())
    floor = float(G[f"{name}/err32/{key}"])
    bar = max(tol * scale, 8.0 * floor)
    assert err <= bar, f"{name}:{key}: err {err:.3e} > bar {bar:.3e} (scale {scale:.3g}, reference fp32 floor {floor:.2e})"
    return err / bar


def _check_norm(name, pkey, grad, tol=3e-5):
    key = f"{name}/ref64/dnorm/{pkey}"
    ref = float(G[key])
    got = float(grad.detach().double().norm())
    floor = float(G[f"{name}/err32/dnorm/{pkey}"])
    assert abs(got - ref) <= max(tol * max(ref, 1e-30), 8.0 * floor), (key, got, ref, floor)


def _backward(name, outs_ref_layout):
    """contract the outputs (given in the reference's layout) with the generator's cotangents and run backward"""
    cots = cotangents(name, outs_ref_layout)
    tot = None
    for k, o in outs_ref_layout.items():
        term = (o * cots[k].to(DEV).to(torch.float32)).sum()
        tot = term if tot is None else tot + term
    tot.backward()


@pytest.mark.parametrize("fused", [True, False])
def test_convlstm_step_matches_the_reference_module(fused, monkeypatch):
    """ConvLSTM.forward (baseline_attention.py:37-56) as scanpath_model.decode() evaluates one step: hoisted x-gate conv with all gate
    biases, contracted rank-1 filters, im2col of the spatial memories, then the fused h-gate conv + cell kernel (or the unfused pair)."""
    from scanpaths_amd import functional as F
    name = "convlstm"
    cs = MODULE_CASES[name]
    B, C, H, W = cs["B"], cs["C"], cs["H"], cs["W"]
    monkeypatch.setattr(F, "FUSE_GATE_LSTM", fused)
    monkeypatch.setattr(F, "COST_M_SCALE", 81920.0 / (B * H * W))      # the benchmark's kernel decisions at this small batch
    spec = {}
    for g_ in ("input", "forget", "output", "memory"):
        for sfx in ("_x", "_h"):
            spec[f"lstm.{g_}{sfx}.weight"], spec[f"lstm.{g_}{sfx}.bias"] = (C, C, 3, 3), (C,)
    for g_ in ("input", "forget", "output"):
        for sfx in ("_pos", "_neg"):
            spec[f"lstm.{g_}{sfx}.weight"], spec[f"lstm.{g_}{sfx}.bias"] = (C, C, 3, 3), (C,)
    P_ = {k: _leaf(v, channels_last=True) for k, v in _params(name, spec).items()}
    ins = case_inputs(name)
    x, h, c = _nhwc(ins["x"]), _nhwc(ins["h"]), _nhwc(ins["c"])
    sp = [_leaf(ins["sp_pos"]), _leaf(ins["sp_neg"])]
    se = [_leaf(ins["se_pos"]), _leaf(ins["se_neg"])]
    streams = ["_pos", "_neg"]
    cat = lambda ws: torch.cat([w.permute(0, 2, 3, 1) for w in ws], 0).permute(0, 3, 1, 2)
    w = lambda k: P_[f"lstm.{k}.weight"]
    b = lambda k: P_[f"lstm.{k}.bias"]
    Wx = cat([w("input_x"), w("forget_x"), w("output_x"), w("memory_x")])
    bias = torch.cat([sum((b(g_ + s) for s in ["_h"] + streams), b(g_ + "_x")) for g_ in ("input", "forget", "output")]
                     + [b("memory_x") + b("memory_h")])
    Wh = cat([w("input_h"), w("forget_h"), w("output_h"), w("memory_h")])
    Wr = [torch.cat([w(g_ + s).permute(0, 2, 3, 1) for g_ in ("input", "forget", "output")], 0).reshape(3 * C * 9, C) for s in streams]
    KP = (9 * 2 + 3) // 4 * 4
    Xg = F.conv2d(x, Wx, bias, pad=1)
    parts = [F.gemm(se[s], Wr[s], None, "nk").view(B, 3 * C, 9) for s in range(2)]
    wc = torch.cat(parts + [torch.zeros(B, 3 * C, KP - 18, device=DEV)], 2)
    spcol = F.im2col3x3(torch.stack(sp, 0), KP)
    c._sp_cbound = float(c.detach().abs().max())          # (decode() carries the bound |c_t| <= t + 1 on the state; here the measured maximum)
    F.reset_fusion_counts()
    if fused:
        assert F.gateconv_lstm_fusable(h, Wh, spcol)
        h2, c2 = F.gateconv_lstm(h, Wh, Xg, c, spcol, wc, {})
        assert F.FUSION_COUNTS["gateconv_lstm"] == 1
    else:
        h2, c2 = F.lstm_cell_rank1(Xg, F.conv2d(h, Wh, None, pad=1), c, spcol, wc)
    nchw = lambda t: t.permute(0, 3, 1, 2)
    _check(name, "h", nchw(h2), 1e-5)
    _check(name, "c", nchw(c2), 1e-5)
    _backward(name, {"h": nchw(h2), "c": nchw(c2)})
    _check(name, "d_x", nchw(x.grad), 3e-5)
    _check(name, "d_h", nchw(h.grad), 3e-5)
    _check(name, "d_c", nchw(c.grad), 3e-5)
    for i, s in enumerate(("pos", "neg")):
        _check(name, f"d_sp_{s}", sp[i].grad, 3e-5)
        _check(name, f"d_se_{s}", se[i].grad, 3e-5)
    for k, p in P_.items():
        _check_norm(name, k, p.grad)


@pytest.mark.parametrize("name", ["head_train", "head_eval"])
def test_head_conv_and_predict_head_match_the_reference_modules(name):
    """nn.Conv2d(512, 512, 5) -> predict_head (baseline_attention.py:306-309, 149-174) as the model evaluates it: composed head filters,
    the saliency tap GEMM + 25-tap gather, the composite 11x11 duration windows, the head epilogue (model._heads_prepare / _heads_step)."""
    from scanpaths_amd.models.scanpath_model import ScanpathModel
    cs = MODULE_CASES[name]
    B, H, W = cs["B"], cs["H"], cs["W"]
    model = ScanpathModel("AiR", convLSTM_length=1, map_width=W, map_height=H).to(DEV)
    model.train(cs["training"])
    spec = {"performance_sal_layer.True.weight": (512, 512, 5, 5), "performance_sal_layer.True.bias": (512,),
            "object_head.sal_layer_2.weight": (1, 512, 1, 1), "object_head.sal_layer_2.bias": (1,),
            "object_head.sal_layer_3.weight": (1, 512, 1, 1), "object_head.sal_layer_3.bias": (1,),
            "object_head.drt_layer_1.weight": (1, 512, 7, 7), "object_head.drt_layer_1.bias": (1,),
            "object_head.drt_layer_2.weight": (2, 1, 6, 8), "object_head.drt_layer_2.bias": (2,)}
    vals = _params(name, spec)
    sd = model.state_dict()
    with torch.no_grad():
        for k, v in vals.items():
            sd[k].copy_(v)
    named = dict(model.named_parameters())
    h = _nhwc(case_inputs(name)["h"])
    hp = model._heads_prepare(B, H, W, DEV, None)
    logits, amap, mu, s2 = model._heads_step(hp, h, h, hp["Wsal"], hp["W11"], hp["cbsum"], hp["cb"], hp["w2"], hp["b2"])
    outs = {"actions": logits[0].view(B, 1, H * W + 1), "log_normal_mu": mu[0].view(B, 1), "log_normal_sigma2": s2[0].view(B, 1),
            "action_map": amap[0].view(B, 1, H, W)}          # head slot 0 = the "True" (good) head
    for k, v in outs.items():
        _check(name, k, v, 1e-5)
    _backward(name, outs)
    _check(name, "d_h_sub8", h.grad.permute(0, 3, 1, 2)[:, ::8], 3e-5)
    ref_n, got_n = float(G[f"{name}/ref64/dnorm_in/h"]), float(h.grad.double().norm())
    assert abs(got_n - ref_n) <= 3e-5 * ref_n, (got_n, ref_n)
    for k in spec:
        _check_norm(name, k, named[k].grad)
        key = f"{name}/ref64/dparam/{k}"
        if key in G.files:
            _check(name, f"dparam/{k}", named[k].grad, 3e-5)


def test_spatial_and_semantic_attention_match_the_reference_modules():
    """spatial_att / semantic_att (baseline_attention.py:111-124, 77-88): scores reduced to <entry, u> (the "cur" branch and the biases
    cancel along the softmax axis; model._attention_vectors), one list-attention kernel per memory."""
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.scanpath_model import ScanpathModel
    csp, cse = MODULE_CASES["spatial_att"], MODULE_CASES["semantic_att"]
    H, W, E = csp["H"], csp["W"], cse["E"]
    model = ScanpathModel("AiR", convLSTM_length=1, map_width=W, map_height=H).to(DEV).train()
    spec_sp = {"spatial_att.spatial_lists.weight": (1, 1, 3, 3), "spatial_att.spatial_lists.bias": (1,),
               "spatial_att.spatial_cur.weight": (1, 1, 3, 3), "spatial_att.spatial_cur.bias": (1,),
               "spatial_att.spatial_attention.weight": (1, 1, H, W), "spatial_att.spatial_attention.bias": (1,)}
    spec_se = {"semantic_att.semantic_lists.weight": (E, E), "semantic_att.semantic_lists.bias": (E,),
               "semantic_att.semantic_cur.weight": (E, E), "semantic_att.semantic_cur.bias": (E,),
               "semantic_att.semantic_attention.weight": (1, E), "semantic_att.semantic_attention.bias": (1,)}
    sd = model.state_dict()
    with torch.no_grad():
        for nm, spec in (("spatial_att", spec_sp), ("semantic_att", spec_se)):
            for k, v in _params(nm, spec).items():
                sd[k].copy_(v)
    named = dict(model.named_parameters())
    u_sem, u_spa = model._attention_vectors()
    # spatial: lists [N,T,H,W] -> entries stacked [T, N, P]
    ins = case_inputs("spatial_att")
    lists = _leaf(ins["lists"])
    N, T = lists.shape[:2]
    mem = F.list_attention(lists.permute(1, 0, 2, 3).reshape(T, N, H * W), u_spa)
    _check("spatial_att", "mem", mem.view(N, H, W), 1e-5)
    ins2 = case_inputs("semantic_att")
    lists2 = _leaf(ins2["lists"])
    N2, T2 = lists2.shape[:2]
    mem2 = F.list_attention(lists2.permute(1, 0, 2).reshape(T2, N2, E), u_sem)
    _check("semantic_att", "mem", mem2, 1e-5)
    cots = cotangents("spatial_att", {"mem": mem.view(N, H, W)})["mem"].to(DEV).float()
    cots2 = cotangents("semantic_att", {"mem": mem2})["mem"].to(DEV).float()
    ((mem.view(N, H, W) * cots).sum() + (mem2 * cots2).sum()).backward()
    _check("spatial_att", "d_lists", lists.grad, 3e-5)
    _check("semantic_att", "d_lists", lists2.grad, 3e-5)
    for nm, spec in (("spatial_att", spec_sp), ("semantic_att", spec_se)):
        for k in spec:
            g = named[k].grad
            ref = float(G[f"{nm}/ref64/dnorm/{k}"]) if f"{nm}/ref64/dnorm/{k}" in G.files else 0.0
            if "_cur." in k or k.endswith(("lists.bias", "attention.bias")):
                # parameters that cancel along the softmax axis: the reference produces rounding-level noise there, this build exact zeros
                assert g is None or float(g.abs().max()) == 0.0, k
                assert ref <= 1e-12, (k, ref)
            else:
                _check_norm(nm, k, g)


@pytest.mark.parametrize("name", ["bottleneck_l3_train", "bottleneck_l3_eval", "bottleneck_l2first_train", "bottleneck_l2first_eval"])
def test_dilated_bottleneck_matches_the_reference_module(name):
    """Bottleneck.forward (AiR/models/resnet.py:73-93) with the dilation surgery of baseline_attention.py:226-238, train-mode batch
    statistics (incl. the running-statistics update) and eval mode: conv2d / bn_act composed as ScanpathModel.encode composes a block."""
    from scanpaths_amd import functional as F
    cs = MODULE_CASES[name]
    p, i, pl, dil, tr = cs["prefix"], cs["inpl"], cs["planes"], cs["dil"], cs["training"]
    spec = {p + "conv1.weight": (pl, i, 1, 1), p + "conv2.weight": (pl, pl, 3, 3), p + "conv3.weight": (4 * pl, pl, 1, 1)}
    bns = [("bn1", pl), ("bn2", pl), ("bn3", 4 * pl)]
    if cs["down"]:
        spec[p + "downsample.0.weight"] = (4 * pl, i, 1, 1)
        bns.append(("downsample.1", 4 * pl))
    for b_, c_ in bns:
        for leaf in ("weight", "bias", "running_mean", "running_var"):
            spec[f"{p}{b_}.{leaf}"] = (c_,)
    vals = _params(name, spec)
    P_ = {k: (_leaf(v, channels_last=True) if not k.endswith(("running_mean", "running_var")) else v.clone()) for k, v in vals.items()}
    x = _nhwc(case_inputs(name)["x"])

    def bn(key, t, residual=None, relu=True):
        return F.bn_act(t, P_[p + key + ".weight"], P_[p + key + ".bias"], P_[p + key + ".running_mean"], P_[p + key + ".running_var"],
                        residual, training=tr, relu=relu)
    x_main, x_side = x, x
    o = bn("bn1", F.conv2d(x_main, P_[p + "conv1.weight"], None, stride=cs["stride"], bn_stats=tr))
    o = bn("bn2", F.conv2d(o, P_[p + "conv2.weight"], None, pad=dil, dil=dil, bn_stats=tr))
    o = F.conv2d(o, P_[p + "conv3.weight"], None, bn_stats=tr)
    idn = x_side
    if cs["down"]:
        idn = bn("downsample.1", F.conv2d(x_side, P_[p + "downsample.0.weight"], None, stride=cs["stride"], bn_stats=tr), relu=False)
    y = bn("bn3", o, residual=idn, relu=True)
    nchw = lambda t: t.permute(0, 3, 1, 2)
    _check(name, "y", nchw(y), 1e-5)
    if tr:
        for b_, _ in bns:
            for leaf in ("running_mean", "running_var"):
                k = f"{p}{b_}.{leaf}"
                _check(name, f"bn_after/{k}", P_[k], 1e-6)
    _backward(name, {"y": nchw(y)})
    _check(name, "d_x", nchw(x.grad), 3e-5)
    for k, v in P_.items():
        if v.requires_grad:
            _check_norm(name, k, v.grad)
            if f"{name}/ref64/dparam/{k}" in G.files:
                _check(name, f"dparam/{k}", v.grad, 3e-5)
