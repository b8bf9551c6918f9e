"""Inference path on the device (BASELINE.json config 5 / SURVEY.md §8 f1): sampling kernels and HIP-graph capture."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_generate_scanpath_matches_reference_known_answers():
    """index arithmetic, masks and the first-terminate scan are bit-exact with the reference (golden from models/sampling.py)"""
    from scanpaths_amd.models.sampling import Sampling
    g = dict(np.load(os.path.join(GOLDEN, "sampling.npz")))
    acts, durs = torch.from_numpy(g["acts"]).to(DEV), torch.from_numpy(g["durs"]).to(DEV)
    s = Sampling(convLSTM_length=16, min_length=1)
    fix, am, dm = s.generate_scanpath(torch.zeros(6, 3, 2, 2, device=DEV), torch.zeros(6, 16, device=DEV), durs, acts)
    assert np.array_equal(am.cpu().numpy(), g["action_masks"]) and np.array_equal(dm.cpu().numpy(), g["duration_masks"])
    for b, f in enumerate(fix):
        ref = g[f"fix{b}"]
        assert len(f) == len(ref)
        if len(f):
            assert np.array_equal(np.stack([f["start_x"], f["start_y"]], 1), ref[:, :2])       # exact pixel centres
            assert np.allclose(f["duration"], ref[:, 2], rtol=0, atol=1e-7)
    length, _, _, _, _ = s._scan(acts, durs)
    assert np.array_equal(length.cpu().numpy(), g["scanpath_length"])


def test_random_sample_distribution_and_masking():
    from scanpaths_amd.models.sampling import Sampling
    B, T, A = 4000, 3, 7
    p = torch.tensor([0.30, 0.05, 0.15, 0.0, 0.25, 0.20, 0.05])
    probs = p.expand(B, T, A).contiguous().to(DEV)
    mu = torch.full((B, T), -1.0, device=DEV)
    s2 = torch.full((B, T), 0.25, device=DEV)
    s = Sampling(convLSTM_length=T, min_length=1, seed=3)
    out = s.random_sample(probs, mu, s2)
    a = out["selected_actions"].cpu()
    assert (a[:, 0] != 0).all()                                  # terminate masked for t < min_length
    assert (a != 3).all()                                        # zero-probability action never drawn
    freq = torch.bincount(a[:, 1], minlength=A).double() / B
    assert (freq - p.double()).abs().max() < 4 * (0.25 / B) ** 0.5 + 0.01      # ~4 sigma of a binomial proportion
    q = p.clone(); q[0] = 0; q /= q.sum()
    freq0 = torch.bincount(a[:, 0], minlength=A).double() / B
    assert (freq0 - q.double()).abs().max() < 4 * (0.25 / B) ** 0.5 + 0.01
    assert torch.equal(out["selected_actions_probs"].cpu(), p[a])               # gathered from the UNMASKED distribution
    logd = out["durations"].log().cpu().double()                 # exp(eps*sigma2 + mu): mean mu, std sigma2 (reference quirk)
    assert abs(logd.mean().item() + 1.0) < 0.02 and abs(logd.std().item() - 0.25) < 0.02
    again = Sampling(convLSTM_length=T, min_length=1, seed=3).random_sample(probs, mu, s2)
    assert torch.equal(again["selected_actions"], out["selected_actions"])      # reproducible for a given seed
    # scanpath_length: first t > 0 with a terminate, else T (reference quirk incl. t = 0)
    L = out["scanpath_length"].squeeze(-1).cpu()
    ref = torch.full((B,), float(T))
    for b in range(B):
        for t in range(T):
            if a[b, t] == 0 and t > 0:
                ref[b] = t
                break
    assert torch.equal(L, ref)


def test_eval_forward_hip_graph_replay_is_bit_identical():
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    m = baseline(convLSTM_length=3)
    fill_module(m, 2)
    m = m.to(DEV).eval()
    b = make_batch("AiR", 2, 240, 320, 3, seed=2)
    img, att = b["images"].to(DEV), b["attention_maps"].to(DEV)
    with torch.no_grad():
        ref = m(img, att)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            m(img, att)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = m(img, att)
    img.copy_(make_batch("AiR", 2, 240, 320, 3, seed=5)["images"].to(DEV))      # new input through the static buffer
    g.replay()
    torch.cuda.synchronize()
    with torch.no_grad():
        ref2 = m(img, att)
    for k in ref2:
        assert torch.equal(out[k], ref2[k]), k
    assert not torch.equal(ref["good_all_actions_prob"], ref2["good_all_actions_prob"])
