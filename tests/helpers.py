"""Shared helpers for the parity tests (oracle <-> golden <-> HIP)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    with open(os.path.join(GOLDEN, name + ".json")) as f:
        meta = json.load(f)
    data = dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return meta, data


def oracle_state(task, arch, seed, Hm=30, Wm=40, dtype=torch.float64, family="default"):
    from scanpaths_amd.procedural import procedural_state_dict
    from scanpaths_amd.spec import model_spec
    sd = procedural_state_dict(model_spec(task, arch, Hm, Wm), seed, family)
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}


def case_inputs(meta, dtype=torch.float64):
    from scanpaths_amd.synth import make_batch
    b = make_batch(meta["task"], meta["B"], meta["H"], meta["W"], meta["T"], seed=meta["seed"])
    return {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in b.items()}


def max_err(a, b):
    if a is None:
        a = torch.zeros_like(torch.as_tensor(b))
    a = torch.as_tensor(a).detach().cpu().to(torch.float64)
    b = torch.as_tensor(b).detach().cpu().to(torch.float64)
    return (a - b).abs().max().item()


# ---- validation-metric goldens (tests/golden/make_golden_eval.py) -------------------------------------------------------------
FV_DTYPE = {"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")}


def toy_multimatch(fv1, fv2, screensize):
    """A FIXED deterministic stand-in for multimatch_gaze.docomparison (third-party, absent here and not vendored by the
    reference): 5 values in (0, 1] from simple float64 statistics of the two scanpaths, NaN when either has fewer than 3 fixations
    (the package's rule, which is what makes the reference drop a pair, utils/evaluation.py:215-217).  The SAME function is the
    `multimatch_gaze` module of the golden run and the `multimatch=` argument of the build's functions, so the goldens pin
    everything around that column (pair enumeration, dropping, grouping, float32 collection, means / stds / best columns)."""
    import numpy as np
    if len(fv1) < 3 or len(fv2) < 3:
        return [float("nan")] * 5
    x1, y1, d1 = (np.asarray(fv1[k], dtype=np.float64) for k in ("start_x", "start_y", "duration"))
    x2, y2, d2 = (np.asarray(fv2[k], dtype=np.float64) for k in ("start_x", "start_y", "duration"))
    w, h = float(screensize[0]), float(screensize[1])
    return [float(1.0 / (1.0 + abs(x1.mean() - x2.mean()) / w)), float(1.0 / (1.0 + abs(y1.mean() - y2.mean()) / h)),
            float(min(len(x1), len(x2)) / max(len(x1), len(x2))), float(1.0 / (1.0 + abs(x1[0] - x2[0]) / w + abs(y1[0] - y2[0]) / h)),
            float(1.0 / (1.0 + abs(d1.sum() - d2.sum())))]


def unpack_scanpaths(fix, lens, counts=None):
    """[sum_len, 3] + per-scanpath lengths (+ per-image counts) -> (list of) lists of the reference's structured arrays"""
    import numpy as np
    out, o = [], 0
    for n in lens:
        a = np.zeros(int(n), dtype=FV_DTYPE)
        a["start_x"], a["start_y"], a["duration"] = fix[o:o + n, 0], fix[o:o + n, 1], fix[o:o + n, 2]
        out.append(a)
        o += int(n)
    if counts is None:
        return out
    grouped, k = [], 0
    for c in counts:
        grouped.append(out[k:k + int(c)])
        k += int(c)
    return grouped


EVAL_COLUMNS = [("MultiMatch", k) for k in ("vector", "direction", "length", "position", "duration")] + \
               [("ScanMatch", "w/o duration"), ("ScanMatch", "with duration"), ("VAME", "SED"), ("VAME", "STDE"),
                ("VAME", "SED_best"), ("VAME", "STDE_best")]
EVAL_CATEGORIES = ("all", "right_answer", "wrong_answer")


def metrics_table(d):
    """the reference's nested metric dict -> [3 categories, 11 columns] float64"""
    import numpy as np
    return np.array([[float(d[c][g][k]) for g, k in EVAL_COLUMNS] for c in EVAL_CATEGORIES], dtype=np.float64)


def kink_candidates(sd, tap, tol=1e-5, heads=("True", "False")):
    """ReLU sites of an oracle run (tap = the dict oracle.forward(..., tap=) filled) whose mask a fp32 implementation may take the other
    way -- |pre-activation| < tol x max|pre-activation| of that tensor (the HIP path's forward values are within ~1e-5 of scale of the
    fp64 ones) -- with the rank-one direction each flip adds to the gradient of the layer's weight (tests/test_model_gpu.py::_kink_residual):
      "sal_conv":                        {"rows": {channel c: [sites of c, 2048 * 9] im2col patches of the encoder output}}
      "performance_sal_layer.<head>":    {"u": sal_layer_3's weights [512], "V": [sites, 512 * 25] im2col patches of h_t}
    Modules without a candidate site are absent."""
    import numpy as np
    import torch.nn.functional as TF
    out = {}
    pre, enc = tap["sal_conv_pre"], tap["enc"]
    idx = (pre.abs() < tol * pre.abs().max()).nonzero().tolist()
    if idx:
        encp = TF.pad(enc, (1, 1, 1, 1))
        rows = {}
        for b, c, y, x in idx:
            rows.setdefault(c, []).append(encp[b, :, y:y + 3, x:x + 3].reshape(-1).double().numpy())
        out["sal_conv"] = {"rows": {c: np.stack(v) for c, v in rows.items()}, "sites": len(idx)}
    for hd in heads:
        V = []
        for (name, t), p3 in tap.get("sal3_pre", {}).items():
            if name != hd:
                continue
            hp = TF.pad(tap["h"][t], (2, 2, 2, 2))
            for b, _, y, x in (p3.abs() < tol * p3.abs().max()).nonzero().tolist():
                V.append(hp[b, :, y:y + 5, x:x + 5].reshape(-1).double().numpy())
        if V:
            mod = "performance_sal_layer." + hd if hd else "performance_sal_layer"
            out[mod] = {"u": sd["object_head.sal_layer_3.weight"].detach().reshape(-1).double().numpy(), "V": np.stack(V), "sites": len(V)}
    return out


def bench_bn_calibration():
    """{"<bn>.running_mean" / ".running_var": float64 array} of the bench-path case (see tests/golden/make_bn_calibration.py)"""
    import numpy as np
    return dict(np.load(os.path.join(GOLDEN, "bench_bn_calib.npz")))


BENCH_CASE = dict(Hm=40, Wm=64, T=16, NB=2, seed=21, H=320, W=512)


def start_bench_oracle(background=False):
    """Start the two host-side oracle runs of the bench-path parity case (fp64 and fp32: ~4-5 minutes of CPU work, nothing on the GPU) in
    two spawned worker processes and return (executor, {tag: future}).  tests/conftest.py calls this when the session's collection is
    final, so the runs overlap with the other GPU tests instead of standing in the suite's critical path; the
    bench-path test is moved to the end of the session and collects the results."""
    import concurrent.futures as cf
    import multiprocessing as mp
    ncpu = os.cpu_count() or 8
    # the fp64 run is 5-8 x the fp32 one (no oneDNN path): it gets the larger share of the host cores.  In the background of a whole
    # session the two runs have ten minutes and must leave the cores to the tests that run their own host oracles meanwhile (with 96
    # threads in the background those took 76 s instead of 49 s, 35 s instead of 15 s, ...)
    threads = ({"float64": max(4, min(32, ncpu // 4)), "float32": max(4, min(16, ncpu // 8))} if background else
               {"float64": max(4, min(96, ncpu // 2)), "float32": max(4, min(32, ncpu // 4))})
    c = BENCH_CASE
    ex = cf.ProcessPoolExecutor(2, mp_context=mp.get_context("spawn"))
    futs = {tag: ex.submit(oracle_bench_case, (dn, c["seed"], c["Hm"], c["Wm"], c["T"], c["NB"], c["H"], c["W"], threads[dn]))
            for dn, tag in (("float64", "ref64/"), ("float32", "ref32/"))}
    return ex, futs


def oracle_bench_case(args):
    """(runs in a spawned worker process) the oracle's train step (loss + gradients) and eval forward of one AiR case in ONE dtype.
    args = (dtype name, seed, Hm, Wm, T, NB, H, W, threads) -> ({"train/<key>": array, "eval/<key>": array}, {param: grad}, loss, kinks);
    kinks = kink_candidates of the train-mode run (float64 only, else None)"""
    import torch
    from oracle import scanpath_oracle as O
    from scanpaths_amd.spec import is_buffer
    from scanpaths_amd.synth import make_batch
    dtname, seed, Hm, Wm, T, NB, H, W, threads = args
    torch.set_num_threads(threads)
    dt = getattr(torch, dtname)
    b = make_batch("AiR", NB, H, W, T, seed=seed)
    sd = oracle_state("AiR", "resnet50", seed, Hm, Wm, dtype=dt, family="tame")
    bd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in b.items()}
    out = {}
    # eval mode: running statistics that fit the case's own inputs (tests/golden/bench_bn_calib.npz, make_bn_calibration.py), as a trained
    # checkpoint has them -- with the procedural ones the oracle's own fp32 run left its fp64 run after ~10 of the 16 decode steps
    sd_ev = dict(sd)
    for k, v in bench_bn_calibration().items():
        sd_ev[k] = torch.from_numpy(v).to(dt)
    with torch.no_grad():
        ev = O.forward(sd_ev, "AiR", bd["images"], bd["attention_maps"], training=False, T=T)
    for k, v in ev.items():
        out["eval/" + k] = v.double().numpy()
    for k, v in sd.items():
        if v.is_floating_point() and not is_buffer(k):
            v.requires_grad_(True)
    tap = {} if dtname == "float64" else None
    tr = O.forward(sd, "AiR", bd["images"], bd["attention_maps"], bd["performances"], training=True, T=T, tap=tap)
    loss, _, _ = O.supervised_loss(tr, bd)
    loss.backward()
    for k, v in tr.items():
        out["train/" + k] = v.detach().double().numpy()
    grads = {k: v.grad.numpy() for k, v in sd.items() if v.requires_grad and v.grad is not None}
    kinks = kink_candidates(sd, tap) if tap is not None else None
    return out, grads, float(loss.detach()), kinks


# ---- full-batch parity leg: BASELINE.json config 2's per-GPU workload itself, bs 32 at 320x512, T = 16 -------------
FULL_CASE = dict(Hm=40, Wm=64, T=16, NB=32, seed=23, H=320, W=512, fp64_samples=(0, 9, 22, 31))


def start_full_oracle(background=False):
    """Two spawned worker processes with the host-side oracle of the full-batch case, TRAIN-mode forward + loss under no_grad:
      ref32/  fp32, all 32 samples (the reference's own arithmetic at the timed configuration)
      ref64/  fp64: the encoder on the WHOLE batch (its BatchNorm statistics are where the batch size enters), the decoder -- per-sample
              in the reference, and 95 % of the literal FLOPs -- on FULL_CASE["fp64_samples"] (oracle.forward decode_samples)
    Returns (executor, {tag: future}).  ~2-4 minutes of CPU work in the background of the GPU session."""
    import concurrent.futures as cf
    import multiprocessing as mp
    ncpu = os.cpu_count() or 8
    # (in the background of a whole session the two runs have ten minutes; with 2 x 42 threads the process-spawning DDP tests of the session
    # took twice as long)
    threads = max(4, min(20, ncpu // 12)) if background else max(4, min(64, ncpu // 3))
    ex = cf.ProcessPoolExecutor(2, mp_context=mp.get_context("spawn"))
    futs = {tag: ex.submit(oracle_full_case, (dn, threads)) for dn, tag in (("float64", "ref64/"), ("float32", "ref32/"))}
    return ex, futs


def oracle_full_case(args):
    """(runs in a spawned worker process) -> ({output key: float64 array}, loss over the rows computed, encoder output of sample 0)"""
    import torch
    from oracle import scanpath_oracle as O
    from scanpaths_amd.synth import make_batch
    dtname, threads = args
    torch.set_num_threads(threads)
    dt = getattr(torch, dtname)
    c = FULL_CASE
    b = make_batch("AiR", c["NB"], c["H"], c["W"], c["T"], seed=c["seed"])
    sd = oracle_state("AiR", "resnet50", c["seed"], c["Hm"], c["Wm"], dtype=dt, family="tame")
    bd = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in b.items()}
    sub = list(c["fp64_samples"]) if dtname == "float64" else None
    eo = {}
    with torch.no_grad():
        tr = O.forward(sd, "AiR", bd["images"], bd["attention_maps"], bd["performances"], training=True, T=c["T"], enc_out=eo,
                       decode_samples=sub)
        rows = {k: (v[sub] if sub is not None else v) for k, v in bd.items() if k in ("scanpaths", "durations", "action_masks", "duration_masks")}
        loss, la, ld = O.supervised_loss(tr, rows)
    enc = eo["enc"]
    return ({k: v.double().numpy() for k, v in tr.items()}, (float(loss), float(la), float(ld)), enc[0].double().numpy())


# ---- the RL phase's eval-mode backward (tests/test_rl_gpu.py::test_eval_mode_backward_matches_oracle): oracle side ---------------------------
RL_EVAL_CASE = dict(T=1, NB=2, H=240, W=320, Hm=30, Wm=40, seed=21)


def rl_eval_objective_weights():
    import torch
    c = RL_EVAL_CASE
    g = torch.Generator().manual_seed(c["seed"])
    P1 = c["Hm"] * c["Wm"] + 1
    return {k: torch.randn(s, generator=g) for k, s in (("good_all_actions_prob", (c["NB"], c["T"], P1)), ("poor_all_actions_prob", (c["NB"], c["T"], P1)),
                                                         ("good_log_normal_mu", (c["NB"], c["T"])), ("poor_log_normal_sigma2", (c["NB"], c["T"])))}


def start_rl_eval_oracle(background=False):
    import concurrent.futures as cf
    import multiprocessing as mp
    ncpu = os.cpu_count() or 8
    threads = max(4, min(12, ncpu // 16)) if background else max(4, min(32, ncpu // 3))
    ex = cf.ProcessPoolExecutor(1, mp_context=mp.get_context("spawn"))
    return ex, ex.submit(rl_eval_oracle_case, threads)


def rl_eval_oracle_case(threads):
    """(runs in a spawned worker process) -> (calibrated running statistics, fp64 gradients, fp64 objective, fp32 gradients, fp32 objective)
    of a weighted sum of the EVAL-mode outputs (AiR/train.py:244-251), as numpy arrays.
    Random weights with the initial running statistics (mean 0, var 1) let activations explode layer by layer; the LSTM gates then saturate
    everywhere and the few gradients that survive depend on WHICH element is accidentally unsaturated (rounding noise of pre-activations of
    size 1e9).  The running statistics are calibrated to this batch instead (one train-mode pass of the oracle, momentum undone), as
    trained checkpoints have them."""
    import torch
    from oracle import scanpath_oracle as O
    from scanpaths_amd.spec import is_buffer
    from scanpaths_amd.synth import make_batch
    torch.set_num_threads(threads)
    c = RL_EVAL_CASE
    T = c["T"]
    b = make_batch("AiR", c["NB"], c["H"], c["W"], T, seed=c["seed"])
    w = rl_eval_objective_weights()
    base = oracle_state("AiR", "resnet50", c["seed"], c["Hm"], c["Wm"], dtype=torch.float64)
    bn_new = {}
    with torch.no_grad():
        O.forward(base, "AiR", b["images"].double(), b["attention_maps"].double(), b["performances"], training=True, T=T, bn_new=bn_new)
    calib = {k: (v - 0.9 * base[k]) / 0.1 for k, v in bn_new.items() if k.endswith("running_mean") or k.endswith("running_var")}
    out = []
    for dt in (torch.float64, torch.float32):
        sd = oracle_state("AiR", "resnet50", c["seed"], c["Hm"], c["Wm"], dtype=dt)
        sd.update({k: v.to(dt) for k, v in calib.items()})
        for k, v in sd.items():
            if v.is_floating_point() and not is_buffer(k):
                v.requires_grad_(True)
        pred = O.forward(sd, "AiR", b["images"].to(dt), b["attention_maps"].to(dt), None, training=False, T=T)
        val = sum((pred[k] * w[k].to(dt)).sum() for k in w)
        val.backward()
        out += [{k: v.grad.numpy() for k, v in sd.items() if v.requires_grad and v.grad is not None}, float(val.detach())]
    return ({k: v.numpy() for k, v in calib.items()},) + tuple(out)
