"""Self-critical (REINFORCE) training step of the AiR model -- the reference's RL branch, AiR/train.py:212-345 -- on the HIP
path.  The pieces:

  * eval-mode forward WITH autograd (softmax heads, running-stat BatchNorm)            scanpaths_amd.models.*
  * 2 * rl_sample_number sampled scanpaths (good head, then poor head)                  Sampling.random_sample / generate_scanpath
  * ScanMatch rewards of every sample against the human scanpaths, batched on device   utils.evaluation.pairs_eval_...
  * -LogAction / -LogDuration of the samples                                           models.loss.LogAction / LogDuration
  * reward shaping: harmonic mean of the two ScanMatch variants, per-head mean baseline (scipy.stats.hmean on [2S, N] host
    arrays, as the reference), loss = sum(neg_log * (reward - baseline)) for actions + durations
  * backward, clip, Adam                                                                FlatAdam

Quirk kept from the reference: the ``+ args.lambda_5 * (...)`` terms of loss_actions / loss_duration stand on their own
source lines (train.py:332-340) and are therefore no-op expression statements -- lambda_5 does not influence the loss.
``rl_loss`` computes them as the reference's arithmetic would and returns them in ``info`` so a caller can opt in.

OSIE / COCO_Search18 (``rl_loss_single_head`` / ``rl_step_single_head``; OSIE/train.py:197-262, COCO_Search18/train.py:212-283): ONE head,
no performance split -- rl_sample_number samples per image, reward = harmonic mean of the two ScanMatch scores of a sample against
the image's human scanpaths (pairs_eval columns 5:7 / pairs_eval_scanmatch), baseline = the mean over the samples."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import scipy.stats
import torch

from . import functional as F
from .models.loss import LogAction, LogDuration
from .utils.evaltools.scanmatch import SequenceTooLong
from .utils.evaluation import (gtpairs_eval_scanmatch_performance_related, pairs_eval, pairs_eval_scanmatch,
                               pairs_eval_scanmatch_performance_related)


def _hmean(a: np.ndarray) -> np.ndarray:
    with np.errstate(divide="ignore", invalid="ignore"):
        return scipy.stats.hmean(a, axis=-1)


def _dot(a: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """sum(a * w) through the HIP GEMM (differentiable w.r.t. a); w is a constant"""
    n = a.numel()
    pad = (-n) % 4
    av, wv = a.reshape(1, n), w.reshape(1, n).to(a.dtype)
    if pad:
        z = torch.zeros(1, pad, device=a.device, dtype=a.dtype)
        av, wv = torch.cat([av, z], 1), torch.cat([wv, z], 1)
    return F.gemm(av, torch.cat([wv, torch.zeros(3, wv.shape[1], device=a.device, dtype=a.dtype)], 0), None, "nk")[0, 0]


def rl_loss(neg_log_actions: torch.Tensor, neg_log_durations: torch.Tensor, same_reward: np.ndarray, diff_reward: np.ndarray,
            gtpairs_good: np.ndarray, gtpairs_poor: np.ndarray, gtpairs_diff: np.ndarray, rl_sample_number: int,
            lambda_5: float = 0.0):
    """neg_log_* [2S, N] (S good-head samples, then S poor-head samples); *_reward [2S, N, 2] ScanMatch (w/o, w/ duration) of each
    sample against same-/different-performance human scanpaths; gtpairs_* [N, 2].  AiR/train.py:296-342."""
    S2, N = neg_log_actions.shape
    S = rl_sample_number
    assert S2 == 2 * S, (S2, S)
    same = np.array(same_reward, dtype=np.float32, copy=True)
    diff = np.array(diff_reward, dtype=np.float32, copy=True)
    same[np.isnan(same)] = 0
    diff[np.isnan(diff)] = 0
    same_h = np.asarray(_hmean(same), dtype=np.float32)                       # [2S, N]
    diff_h = np.asarray(_hmean(diff), dtype=np.float32)
    base_same = np.broadcast_to(same_h.reshape(2, S, N).mean(1, keepdims=True), (2, S, N)).reshape(2 * S, N)
    g, p, d = (np.array(x, dtype=np.float64, copy=True) for x in (gtpairs_good, gtpairs_poor, gtpairs_diff))
    for x in (g, p, d):
        x[np.isnan(x)] = 0
    g, p, d = _hmean(g), _hmean(p), _hmean(d)                                 # [N]
    gt_same = np.concatenate([np.tile(g, (S, 1)), np.tile(p, (S, 1))], 0)     # [2S, N]
    gt_diff = np.tile(d, (2 * S, 1))
    usable = ((gt_same != 0) * (gt_diff != 0)).astype(np.float32)
    difference_reward = np.abs((same_h - diff_h) - (gt_same - gt_diff)).astype(np.float32) * usable
    base_difference = np.broadcast_to(difference_reward.reshape(2, S, N).mean(1, keepdims=True), (2, S, N)).reshape(2 * S, N)
    dev = neg_log_actions.device
    adv = torch.from_numpy(np.ascontiguousarray(same_h - base_same)).to(dev)
    loss_actions = _dot(neg_log_actions, adv)
    loss_duration = _dot(neg_log_durations, adv)
    loss = F.add(loss_actions.reshape(1), loss_duration.reshape(1))[0]
    info = {"loss_actions": loss_actions, "loss_duration": loss_duration, "same_reward_hmean": same_h, "diff_reward_hmean": diff_h,
            "difference_reward": difference_reward, "baseline_difference_reward": base_difference, "advantage": same_h - base_same,
            "lambda_5_terms_are_noops_in_the_reference": True, "lambda_5": lambda_5}
    return loss, info


def rl_step(model, sampling, optimizer, images, attention_maps, gt_fix_vectors, performances, ScanMatchwithDuration,
            ScanMatchwithoutDuration, rl_sample_number: int = 5, lambda_5: float = 0.0, ablate_attention_info: bool = False,
            max_resamples: int = 100):
    """One RL iteration (AiR/train.py:221-345).  ``optimizer`` is a FlatAdam (clip folded into step()).  Returns (loss, info)."""
    model.eval()
    N = images.shape[0]
    S = rl_sample_number
    given_performance = [True] * S + [False] * S
    gt_good, gt_poor, gt_diff = gtpairs_eval_scanmatch_performance_related(gt_fix_vectors, ScanMatchwithDuration,
                                                                           ScanMatchwithoutDuration, performances)
    if ablate_attention_info:
        attention_maps = attention_maps * 0
    optimizer.zero_grad()
    predict = model(images, attention_maps)
    same_b: List[np.ndarray] = []
    diff_b: List[np.ndarray] = []
    nla: List[torch.Tensor] = []
    nld: List[torch.Tensor] = []
    trial = resamples = 0
    while trial < 2 * S:
        head = "good" if given_performance[trial] else "poor"
        prob, mu, s2 = predict[head + "_all_actions_prob"], predict[head + "_log_normal_mu"], predict[head + "_log_normal_sigma2"]
        samples = sampling.random_sample(prob, mu, s2)
        fix, action_masks, duration_masks = sampling.generate_scanpath(images, samples["selected_actions_probs"],
                                                                       samples["durations"], samples["selected_actions"])
        # a heavy-tailed duration draw (exp(eps*sigma2 + mu)) can be inf or map to more symbols than the scorer's buffers hold
        # (utils/evaltools/scanmatch.py MAX_SYMBOLS): the reference's int(round(inf)) raises there; here the sample is redrawn
        finite = all(np.isfinite(np.asarray(f["duration"], dtype=np.float64)).all() for f in fix)
        try:
            if not finite:
                raise SequenceTooLong("non-finite sampled duration")
            same, diff, accept = pairs_eval_scanmatch_performance_related(gt_fix_vectors, fix, ScanMatchwithDuration,
                                                                          ScanMatchwithoutDuration, performances,
                                                                          given_performance[trial])
        except SequenceTooLong:
            accept = False
        if not accept:
            resamples += 1
            if resamples > max_resamples:
                raise RuntimeError("rl_step: too many rejected samples (every reward of a group is NaN)")
            continue
        trial += 1
        same_b.append(same)
        diff_b.append(diff)
        nla.append(F.scale_const(LogAction(samples["selected_actions_probs"], action_masks), -1.0))
        nld.append(F.scale_const(LogDuration(samples["durations"].detach(), mu, s2, duration_masks), -1.0))
    loss, info = rl_loss(torch.stack(nla, 0), torch.stack(nld, 0), np.stack(same_b, 0), np.stack(diff_b, 0), gt_good, gt_poor,
                         gt_diff, S, lambda_5)
    loss.backward()
    info["grad_norm"] = optimizer.step()
    info["resamples"] = resamples
    return loss.detach(), info


def rl_loss_single_head(neg_log_actions: torch.Tensor, neg_log_durations: torch.Tensor, metrics_reward, task: str):
    """neg_log_* [S, N]; metrics_reward [S, N, 11] (OSIE: pairs_eval) or [S, N, 2] (COCO_Search18: pairs_eval_scanmatch), float32 as the
    reference collects them.  OSIE/train.py:248-258, COCO_Search18/train.py:269-279 -> (loss, info)"""
    assert task in ("OSIE", "COCO_Search18"), task
    r = np.asarray(metrics_reward, dtype=np.float32)
    cols = r[:, :, 5:7] if task == "OSIE" else r
    hm = np.asarray(_hmean(cols), dtype=np.float32)                           # [S, N]
    base = hm.mean(0, keepdims=True)
    adv = torch.from_numpy(np.ascontiguousarray(hm - base)).to(neg_log_actions.device)
    loss_actions = _dot(neg_log_actions, adv)
    loss_duration = _dot(neg_log_durations, adv)
    loss = F.add(loss_actions.reshape(1), loss_duration.reshape(1))[0]
    return loss, {"loss_actions": loss_actions, "loss_duration": loss_duration, "reward_hmean": hm, "baseline": base,
                  "advantage": hm - base, "metrics_for_reward": r.mean(0).mean(0)}


def rl_step_single_head(model, sampling, optimizer, images, gt_fix_vectors, ScanMatchwithDuration, ScanMatchwithoutDuration, task: str,
                        attention_maps=None, tasks=None, rl_sample_number: int = 5, ablate_attention_info: bool = False,
                        max_resamples: int = 100, multimatch=None):
    """One RL iteration of OSIE (OSIE/train.py:205-262: model(images)) or COCO_Search18 (COCO_Search18/train.py:219-283:
    model(images, attention_maps, tasks)).  ``optimizer`` is a FlatAdam (clip folded into step()).  Returns (loss, info)."""
    assert task in ("OSIE", "COCO_Search18"), task
    model.eval()
    optimizer.zero_grad()
    if task == "OSIE":
        predict = model(images)
    else:
        if ablate_attention_info:
            attention_maps = attention_maps * 0
        predict = model(images, attention_maps, tasks)
    prob, mu, s2 = predict["all_actions_prob"], predict["log_normal_mu"], predict["log_normal_sigma2"]
    rewards: List[np.ndarray] = []
    nla: List[torch.Tensor] = []
    nld: List[torch.Tensor] = []
    trial = resamples = 0
    while trial < rl_sample_number:
        samples = sampling.random_sample(prob, mu, s2)
        fix, action_masks, duration_masks = sampling.generate_scanpath(images, samples["selected_actions_probs"], samples["durations"],
                                                                       samples["selected_actions"])
        finite = all(np.isfinite(np.asarray(f["duration"], dtype=np.float64)).all() for f in fix)
        try:
            if not finite:
                raise SequenceTooLong("non-finite sampled duration")
            if task == "OSIE":
                reward = pairs_eval(gt_fix_vectors, fix, ScanMatchwithDuration, ScanMatchwithoutDuration, multimatch=multimatch)
            else:
                reward = pairs_eval_scanmatch(gt_fix_vectors, fix, ScanMatchwithDuration, ScanMatchwithoutDuration)
        except SequenceTooLong:
            reward = np.array([[np.nan]])
        if np.any(np.isnan(reward)):                         # (OSIE/train.py:236-237) the sample is redrawn
            resamples += 1
            if resamples > max_resamples:
                raise RuntimeError("rl_step_single_head: too many rejected samples (a reward row is NaN every time)")
            continue
        trial += 1
        rewards.append(np.asarray(reward, dtype=np.float32))
        nla.append(F.scale_const(LogAction(samples["selected_actions_probs"], action_masks), -1.0))
        nld.append(F.scale_const(LogDuration(samples["durations"].detach(), mu, s2, duration_masks), -1.0))
    loss, info = rl_loss_single_head(torch.stack(nla, 0), torch.stack(nld, 0), np.stack(rewards, 0), task)
    loss.backward()
    info["grad_norm"] = optimizer.step()
    info["resamples"] = resamples
    return loss.detach(), info
