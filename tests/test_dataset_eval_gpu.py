"""SURVEY.md §8 "next" rows on the device: dataset targets (f4), beam search + the test loop (f1), validation metrics (f2)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
_FV = {"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")}


def test_collate_targets_match_reference_collate_func():
    """HIP collate kernel vs the reference's AiR.__getitem__ + collate_func outputs (tests/golden/collate.npz): bit-exact"""
    from scanpaths_amd.dataset import collate_targets
    g = np.load(os.path.join(GOLDEN, "collate.npz"))
    recs = json.loads(bytes(g["records"]).decode())
    # collate.npz was made under numpy >= 2 (float32 division): the kernel's float32 branch is what it pins; the default
    # (f64_div=True = the reference's pinned numpy 1.19) is pinned by collate_f64.npz below
    out = collate_targets(recs, 16, (30, 40), f64_div=False)
    for k in ("scanpaths", "durations", "action_masks", "duration_masks"):
        assert np.array_equal(out[k].cpu().numpy(), g[k]), k
    g64 = np.load(os.path.join(GOLDEN, "collate_f64.npz"))
    recs64 = json.loads(bytes(g64["records"]).decode())
    out64 = collate_targets(recs64, 16, (30, 40))
    for k in ("scanpaths", "durations", "action_masks", "duration_masks"):
        assert np.array_equal(out64[k].cpu().numpy(), g64[k]), ("f64 default", k)
    # numpy-1.x (float64 division) semantics vs the oracle restatement, on cell-boundary-rich inputs at 320x512 -> 40x64
    from oracle import sampling_oracle as SO
    rng = np.random.Generator(np.random.PCG64(2))
    recs2 = [{"X": list((rng.integers(0, 64, 12) * 8.0).astype(float)), "Y": list((rng.integers(0, 40, 12) * 8.0).astype(float)),
              "T_start": list(np.arange(12) * 100.0), "T_end": list(np.arange(12) * 100.0 + 77.0), "height": 320, "width": 512}
             for _ in range(5)]
    for f64 in (False, True):
        got = collate_targets(recs2, 16, (40, 64), f64_div=f64)
        want = SO.collate_targets(recs2, 16, (40, 64), f64_div=f64)
        for k, w in zip(("scanpaths", "durations", "action_masks", "duration_masks"), want):
            assert np.array_equal(got[k].cpu().numpy(), w), (k, f64)


def test_collate_func_keys_and_performance_rule():
    from scanpaths_amd.dataset import collate_func, normalise_attention
    g = np.load(os.path.join(GOLDEN, "collate.npz"))
    recs = json.loads(bytes(g["records"]).decode())
    samples = [{"image": torch.zeros(3, 8, 8), "fixation": r, "attention_map": np.full((1, 30, 40), 0.5, np.float32),
                "img_name": r["image_id"], "question_id": r["question_id"]} for r in recs]
    data = collate_func(samples, 16, (30, 40))
    assert set(data) == {"images", "scanpaths", "durations", "action_masks", "duration_masks", "attention_maps", "img_names",
                         "question_ids", "performances"}
    assert np.array_equal(data["performances"].cpu().numpy(), g["performances"])
    att = torch.rand(3, 1, 30, 40, device=DEV) * 7
    n = normalise_attention(att)
    assert torch.allclose(n.flatten(1).max(1).values, torch.ones(3, device=DEV))


def test_beam_search_matches_oracle_and_is_optimal():
    """HIP beam kernel == the numpy restatement (itself exhaustively checked on small cases, tests/test_sampling_oracle.py);
    at the bs-128 / 320x512 inference shape: beam 0 is the greedy path, scores are sorted, sequences distinct."""
    from oracle import sampling_oracle as SO
    from scanpaths_amd.models.sampling import Sampling
    rng = np.random.Generator(np.random.PCG64(6))
    B, T, A, K = 6, 16, 1201, 4
    p = rng.random((B, T, A)).astype(np.float32) ** 8
    p[0, 3, 0] = 50.0                  # an early terminate
    p[1, :, 5] = 0.0                   # exact zeros
    p[2, 7, 10] = p[2, 7, 20] = p[2, 7].max() * 2     # an exact tie between two actions
    p /= p.sum(-1, keepdims=True)
    probs = torch.from_numpy(p).to(DEV)
    mu = torch.zeros(B, T, device=DEV)
    for ml in (1, 2):
        s = Sampling(convLSTM_length=T, min_length=ml)
        out = s.beam_search(probs, mu, mu + 1, beam=K)
        acts, scores = out["selected_actions"].cpu().numpy(), out["scores"].cpu().numpy()
        for b in range(B):
            wa, ws = SO.beam_search(p[b], ml, K)
            assert np.array_equal(acts[b], wa), (ml, b)
            assert np.allclose(scores[b], ws, rtol=0, atol=1e-9)
        assert torch.equal(out["durations"], torch.ones(B, K, T, device=DEV))          # exp(mu) with mu = 0
    # full inference shape
    B, T, A = 128, 16, 1 + 40 * 64
    probs = torch.softmax(torch.randn(B, T, A, device=DEV) * 3, -1)
    s = Sampling(convLSTM_length=T, min_length=1, map_width=64, map_height=40, width=512, height=320)
    out = s.beam_search(probs, torch.zeros(B, T, device=DEV), torch.ones(B, T, device=DEV), beam=4)
    acts, sc = out["selected_actions"], out["scores"]
    assert (sc[:, :-1] >= sc[:, 1:]).all()
    # optimality against one known candidate: the greedy path (per-step arg-max, ended by its first terminate).  The best
    # sequence under sum-of-log-probabilities may well terminate earlier (every further step costs log p < 0), so beam 0 need
    # not EQUAL the greedy path -- but it can never score below it
    pm = probs.clone()
    pm[:, 0, 0] = 0                    # terminate masked at t < min_length
    gp, ga = pm.max(-1)
    alive = ((ga == 0).float().cumsum(1) - (ga == 0).float()) == 0            # steps up to and including the first terminate
    greedy_score = (gp.double().log() * alive).sum(1)
    assert (sc[:, 0] >= greedy_score - 1e-9).all()
    assert (acts[:, 0] != acts[:, 1]).any(1).all()
    # every returned sequence really has the score it claims
    idx = acts.clamp(min=0)
    lp = torch.gather(probs.unsqueeze(1).expand(B, 4, T, A), 3, idx.unsqueeze(-1)).squeeze(-1).double().log()
    live = ((acts == 0).float().cumsum(2) - (acts == 0).float()) == 0
    assert torch.allclose((lp * live).sum(2), sc, rtol=0, atol=1e-9)
    fix, am, dm = s.generate_scanpath(torch.zeros(B, 3, 2, 2, device=DEV), None, out["durations"][:, 0], acts[:, 0])
    assert len(fix) == B and am.shape == (B, T)


def _fv(n, rng):
    a = np.zeros(n, dtype=_FV)
    a["start_x"], a["start_y"], a["duration"] = rng.uniform(0, 320, n), rng.uniform(0, 240, n), rng.uniform(0.1, 0.6, n)
    return a


def test_evaluation_performance_related_matches_literal_restatement():
    """batched device scoring + the reference's grouping == the reference's nested loops on the CPU oracles
    (oracle/eval_oracle.py, utils/evaluation.py:188-359).  ScanMatch and SED columns are bit-exact, STDE <= 4 ulp; MultiMatch is
    the same restated callable on both sides (third-party package absent: that column is unpinned)."""
    from oracle import eval_oracle as EO
    from scanpaths_amd.utils.evaltools.multimatch import docomparison
    from scanpaths_amd.utils.evaluation import evaluation_performance_related
    rng = np.random.Generator(np.random.PCG64(12))
    n_img = 7
    gt = [[_fv(int(rng.integers(2, 12)), rng) for _ in range(int(rng.integers(2, 6)))] for _ in range(n_img)]
    perf = [[bool(rng.random() < 0.5) for _ in g] for g in gt]
    perf[0] = [True] * len(perf[0])
    perf[1] = [False] * len(perf[1])
    pred = [_fv(int(rng.integers(3, 14)), rng) for _ in range(n_img)]
    pred[2] = _fv(2, rng)                    # too short for MultiMatch: every pair of this image is dropped
    alloc = [True, False, True, True, False, True, False]
    cur, cur_std, scores = evaluation_performance_related(gt, pred, perf, alloc, multimatch=docomparison)
    mean_ref, std_ref, scores_ref = EO.evaluation_performance_related(gt, pred, perf, alloc, docomparison)
    names = [("MultiMatch", k) for k in ("vector", "direction", "length", "position", "duration")] + \
            [("ScanMatch", "w/o duration"), ("ScanMatch", "with duration"), ("VAME", "SED"), ("VAME", "STDE"),
             ("VAME", "SED_best"), ("VAME", "STDE_best")]
    for ci, cat in enumerate(("all", "right_answer", "wrong_answer")):
        for col, (grp, key) in enumerate(names):
            assert abs(float(cur[cat][grp][key]) - float(mean_ref[ci][col])) <= 1e-6, (cat, grp, key)
            assert abs(float(cur_std[cat][grp][key]) - float(std_ref[ci][col])) <= 1e-6, (cat, grp, key)
    assert len(scores) == n_img
    for a, b in zip(scores, scores_ref):
        assert np.allclose(a, b, rtol=0, atol=1e-9)
    assert scores[2] == list(np.zeros(9))



def test_multimatch_on_the_device_matches_the_host_restatement():
    """sp_scan_multimatch (one thread per pair, all pairs of a validation call in one launch: AiR/utils/evaluation.py:44-45,213 calls
    multimatch_gaze.docomparison per pair) against utils/evaltools/multimatch.docomparison, its checker: lengths 1 .. 24 (fewer than 3
    fixations -> five NaNs on both sides), coordinates on an 8-pixel lattice (equal alignment costs: the tie rule right > down >
    diagonal must agree), repeated fixations (zero-length saccades), equal durations.  Same alignment path -> vector / length /
    position / duration values to 1e-13 (python's ** 2 is pow()), direction to 1e-12 (atan2's last bit).  Then the two validation
    entry points with the device default against the same call with the host callable."""
    from scanpaths_amd.utils.evaltools.multimatch import docomparison, multimatch_pairs
    from scanpaths_amd.utils.evaluation import evaluation_performance_related
    rng = np.random.Generator(np.random.PCG64(31))

    def fv(n, lattice):
        a = np.zeros(n, dtype=_FV)
        if lattice:
            a["start_x"], a["start_y"] = rng.integers(0, 40, n) * 8.0, rng.integers(0, 30, n) * 8.0
            a["duration"] = rng.integers(1, 4, n) * 0.1
        else:
            a["start_x"], a["start_y"], a["duration"] = rng.uniform(0, 320, n), rng.uniform(0, 240, n), rng.uniform(0.05, 0.9, n)
        return a
    paths = [fv(int(rng.integers(1, 25)), bool(k % 2)) for k in range(60)]
    paths[5] = paths[4].copy()                                        # identical scanpaths: all differences zero
    paths[7]["start_x"][:], paths[7]["start_y"][:] = 100.0, 50.0       # one location: zero-length saccades
    pairs = [(int(a), int(b)) for a, b in rng.integers(0, len(paths), (400, 2))] + [(4, 5), (7, 7), (7, 8)]
    got = multimatch_pairs(paths, pairs, [320, 240])
    worst = np.zeros(5)
    nnan = 0
    for (a, b), g in zip(pairs, got):
        with np.errstate(all="ignore"):
            ref = np.asarray(docomparison(paths[a], paths[b], screensize=[320, 240]), dtype=np.float64)
        assert np.array_equal(np.isnan(ref), np.isnan(g)), (a, b, ref, g)
        if np.isnan(ref).any():
            nnan += 1
            continue
        worst = np.maximum(worst, np.abs(ref - g))
    print(f"MultiMatch device vs host restatement over {len(pairs)} pairs ({nnan} unscorable): worst |diff| per value {worst}")
    assert nnan > 10 and (worst[[0, 2, 3, 4]] <= 1e-13).all() and worst[1] <= 1e-12, worst
    n_img = 6
    gt = [[_fv(int(rng.integers(2, 12)), rng) for _ in range(int(rng.integers(2, 6)))] for _ in range(n_img)]
    perf = [[bool(rng.random() < 0.5) for _ in g] for g in gt]
    perf[0], perf[1] = [True] * len(perf[0]), [False] * len(perf[1])
    pred = [_fv(int(rng.integers(3, 14)), rng) for _ in range(n_img)]
    alloc = [True, False, True, True, False, True]
    dev_out = evaluation_performance_related(gt, pred, perf, alloc)                              # default: device MultiMatch
    host_out = evaluation_performance_related(gt, pred, perf, alloc, multimatch=docomparison)
    for cat in ("all", "right_answer", "wrong_answer"):
        for grp in ("MultiMatch", "ScanMatch", "VAME"):
            for key, v in host_out[0][cat][grp].items():
                assert abs(float(dev_out[0][cat][grp][key]) - float(v)) <= 1e-6, (cat, grp, key)
    for a, b in zip(dev_out[2], host_out[2]):
        assert np.allclose(a, b, rtol=0, atol=1e-9)


def test_run_test_loop_order_and_single_copy():
    """the reference's test loop (AiR/test.py:111-193) on the device: record order (per trial: N good, then N poor), counts,
    finite metrics; fixation vectors equal what generate_scanpath returns for the same draws"""
    from scanpaths_amd.inference import run_test_loop, sample_batch, to_fix_vectors
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.models.sampling import Sampling
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    from scanpaths_amd.utils.evaltools.multimatch import docomparison
    T, N, R = 6, 3, 2
    m = baseline(convLSTM_length=T, arch="resnet18")
    fill_module(m, 3, family="tame")
    m = m.to(DEV).eval()
    rng = np.random.Generator(np.random.PCG64(5))
    b = make_batch("AiR", N, 240, 320, T, seed=3)
    batch = {"images": b["images"], "attention_maps": b["attention_maps"],
             "fix_vectors": [[_fv(int(rng.integers(3, 9)), rng) for _ in range(3)] for _ in range(N)],
             "performances": [[True, False, True] for _ in range(N)], "question_ids": [f"q{i}" for i in range(N)],
             "img_names": [f"i{i}.jpg" for i in range(N)]}
    samp = Sampling(convLSTM_length=T, min_length=1, seed=9)
    cur, cur_std, scores, results = run_test_loop(m, samp, [batch], repeat_num=R, multimatch=docomparison)
    assert len(results) == 2 * R * N and len(scores) == 2 * R * N
    assert [r["performance"] for r in results] == ([True] * N + [False] * N) * R
    assert [r["repeat_id"] for r in results] == [1] * (2 * N) + [2] * (2 * N)
    assert all(r["length"] == len(r["X"]) == len(r["T"]) for r in results)
    for cat in cur:
        for grp in cur[cat]:
            for k, v in cur[cat][grp].items():
                assert np.isfinite(v), (cat, grp, k)
    # same seed -> the loop's fixation vectors equal generate_scanpath's for the same draws
    with torch.no_grad():
        pred = m(batch["images"].to(DEV), batch["attention_maps"].to(DEV))
    s2 = Sampling(convLSTM_length=T, min_length=1, seed=9)
    fix, nfix = sample_batch(pred, s2, 1)
    fv = to_fix_vectors(fix[0, 0].cpu().numpy().astype(np.float64), nfix[0, 0].cpu().numpy())
    s3 = Sampling(convLSTM_length=T, min_length=1, seed=9)
    d = s3.random_sample(pred["good_all_actions_prob"], pred["good_log_normal_mu"], pred["good_log_normal_sigma2"])
    ref, _, _ = s3.generate_scanpath(batch["images"].to(DEV), d["selected_actions_probs"], d["durations"], d["selected_actions"])
    for a, r in zip(fv, ref):
        assert np.array_equal(a["start_x"], r["start_x"]) and np.array_equal(a["duration"], r["duration"])
    assert results[0]["X"] == list(fv[0]["start_x"])


def test_validation_metrics_match_the_real_reference():
    """the HIP-scored evaluation_performance_related / human_evaluation against outputs of the REAL reference
    functions (/root/reference/AiR/utils/evaluation.py:188-359, :11-186; tests/golden/eval_metrics.npz from
    tests/golden/make_golden_eval.py, deterministic multimatch stand-in on both sides).  ScanMatch (columns 5, 6) and SED (7, 9)
    enter bit-exact, STDE <= 4 ulp; the reference collects rows in float32, so table entries are held to 1e-6 and the float64
    per-image score rows to 1e-9."""
    from helpers import metrics_table, toy_multimatch
    from test_oracle_golden import _eval_golden
    from scanpaths_amd.utils.evaluation import evaluation_performance_related, human_evaluation
    g, (gt, pred, perf, alloc), loader, qids = _eval_golden()
    cur, cur_std, scores = evaluation_performance_related(gt, pred, perf, alloc, multimatch=toy_multimatch)
    assert np.abs(metrics_table(cur) - g["epr_mean"]).max() <= 1e-6
    assert np.abs(metrics_table(cur_std) - g["epr_std"]).max() <= 1e-6
    assert np.abs(np.array(scores, dtype=np.float64) - g["epr_scores"]).max() <= 1e-9
    assert np.array_equal(np.array(scores)[:, 5:8], g["epr_scores"][:, 5:8])                # ScanMatch x2 and SED: bit-exact
    assert np.array_equal(metrics_table(cur)[:, 9], g["epr_mean"][:, 9])                     # SED_best
    hm, hs, hsc = human_evaluation(loader, multimatch=toy_multimatch)
    assert np.abs(metrics_table(hm) - g["hum_mean"]).max() <= 1e-6
    assert np.abs(metrics_table(hs) - g["hum_std"]).max() <= 1e-6
    good = np.array([hsc[q][True] for q in qids])
    poor = np.array([hsc[q][False] for q in qids])
    assert np.abs(good - g["hum_good"]).max() <= 1e-9 and np.abs(poor - g["hum_poor"]).max() <= 1e-9
    assert np.array_equal(good[:, 5:8], g["hum_good"][:, 5:8]) and np.array_equal(poor[:, 5:8], g["hum_poor"][:, 5:8])


def test_dataset_variants_match_reference_datasets():
    """SURVEY §8 f4 remainder on the device, against outputs of the REAL reference datasets (tests/golden/make_golden_dataset.py):
    numpy-1.x division semantics (the default; AiR/dataset/dataset.py:125-134), blur_sigma targets (:144-146; scipy agreement to
    float32 rounding, the normaliser is summed in another order), OSIE (OSIE/dataset/dataset.py:59-115, incl. blur_sigma = 2) and
    COCO-Search18 targets (COCO_Search18/dataset/dataset.py:88-128, out-of-frame fixations clamped)."""
    from scanpaths_amd import dataset as DS
    g = np.load(os.path.join(GOLDEN, "dataset_variants.npz"))
    g64 = np.load(os.path.join(GOLDEN, "collate_f64.npz"))
    jf = lambda arr: json.loads(bytes(arr).decode())
    keys = ("scanpaths", "durations", "action_masks", "duration_masks")
    recs = jf(g64["records"])
    out = DS.collate_targets(recs, 16, (30, 40))                         # default = the reference's pinned numpy semantics
    for k in keys:
        assert np.array_equal(out[k].cpu().numpy(), g64[k]), k

    def close(a, b, what):
        a = a.cpu().numpy()
        assert np.array_equal(a == 0, b == 0), what                       # same support (the filter's radius and reflection)
        assert np.abs(a - b).max() <= 2e-7 * np.abs(b).max(), (what, np.abs(a - b).max())
        assert np.abs(a.sum(-1) - 1).max() <= 1e-6, what                   # every target row still sums to one

    out = DS.collate_targets(recs, 16, (30, 40), blur_sigma=1)
    close(out["scanpaths"], g["blur_scanpaths"], "AiR blur")
    orecs, crecs = jf(g["osie_records"]), jf(g["coco_records"])
    out = DS.collate_targets_osie(orecs, 16, (30, 40))
    for k in keys:
        assert np.array_equal(out[k].cpu().numpy(), g["osie_" + k]), k
    close(DS.collate_targets_osie(orecs, 16, (30, 40), blur_sigma=2)["scanpaths"], g["osie_blur_scanpaths"], "OSIE blur")
    out = DS.collate_targets_coco(crecs, 16, (30, 40))
    for k in keys:
        assert np.array_equal(out[k].cpu().numpy(), g["coco_" + k]), k
    # the benchmark's map size: blur against the oracle (scipy, as the reference calls it)
    from oracle import sampling_oracle as SO
    rng = np.random.Generator(np.random.PCG64(4))
    recs2 = [{"X": list(rng.uniform(0, 511.9, 9)), "Y": list(rng.uniform(0, 319.9, 9)), "T_start": list(np.arange(9) * 100.0),
              "T_end": list(np.arange(9) * 100.0 + 60.0), "height": 320, "width": 512} for _ in range(4)]
    recs2[0]["X"][0], recs2[0]["Y"][0] = 0.0, 0.0                          # corner cells: both axes reflect
    recs2[1]["X"][0], recs2[1]["Y"][0] = 511.9, 319.9
    want = SO.collate_targets(recs2, 16, (40, 64), f64_div=True, blur_sigma=1.5)[0]
    close(DS.collate_targets(recs2, 16, (40, 64), blur_sigma=1.5)["scanpaths"], want, "40x64 blur")
