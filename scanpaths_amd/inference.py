"""The reference's test / validation loop (AiR/test.py:111-197, AiR/train.py:382-450) on the device.

Per batch: ONE eval-mode forward, then ``repeat_num`` x (good head, poor head) sampled scanpaths -- 2 * repeat_num
``random_sample`` + scan launches that stay on the device -- and ONE device->host copy of all fixation vectors of the batch
(the reference does 2 * repeat_num * N * T ``.cpu().numpy()`` calls, models/sampling.py:55-72).  The lists handed to
``evaluation_performance_related`` and the ``predict_results`` records have the reference's order and content:
for every trial the N good-head scanpaths (allocated performance True) followed by the N poor-head ones (False)."""
from __future__ import annotations

from typing import Dict, Iterable, List, Tuple

import numpy as np
import torch

_FV = {"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")}


def sample_batch(predict: Dict[str, torch.Tensor], sampling, repeat_num: int, heads=("good", "poor")):
    """-> fix [repeat, len(heads), N, T, 3] float32 and nfix [repeat, len(heads), N] int32, still on the device"""
    fixs, nfs = [], []
    for _ in range(repeat_num):
        for head in heads:
            pre = head + "_" if head else ""
            s = sampling.random_sample(predict[pre + "all_actions_prob"], predict[pre + "log_normal_mu"],
                                       predict[pre + "log_normal_sigma2"])
            _, _, _, fix, nfix = sampling._scan(s["selected_actions"], s["durations"])
            fixs.append(fix)
            nfs.append(nfix)
    N, T = fixs[0].shape[0], fixs[0].shape[1]
    return torch.stack(fixs).view(repeat_num, len(heads), N, T, 3), torch.stack(nfs).view(repeat_num, len(heads), N)


def to_fix_vectors(fix_h: np.ndarray, n_h: np.ndarray) -> List[np.ndarray]:
    out = []
    for b in range(fix_h.shape[0]):
        fv = np.zeros(int(n_h[b]), dtype=_FV)
        fv["start_x"], fv["start_y"], fv["duration"] = fix_h[b, :n_h[b], 0], fix_h[b, :n_h[b], 1], fix_h[b, :n_h[b], 2]
        out.append(fv)
    return out


@torch.no_grad()
def run_test_loop(model, sampling, loader: Iterable[dict], repeat_num: int = 10, ablate_attention_info: bool = False,
                  multimatch=None) -> Tuple[dict, dict, list, list]:
    """loader yields the reference's evaluation batches: images, fix_vectors, performances, attention_maps, question_ids,
    img_names (AiR/test.py:119-123).  Returns (cur_metrics, cur_metrics_std, scores_of_each_images, predict_results)."""
    from .utils.evaluation import evaluation_performance_related
    model.eval()
    all_gt, all_pred, all_perf, all_alloc, predict_results = [], [], [], [], []
    for batch in loader:
        images, attention_maps = batch["images"].cuda(), batch["attention_maps"].cuda()
        gt_fix_vectors, performances = batch["fix_vectors"], batch["performances"]
        N = images.shape[0]
        if ablate_attention_info:
            attention_maps = attention_maps * 0
        predict = model(images, attention_maps)
        fix, nfix = sample_batch(predict, sampling, repeat_num)
        fix_h, n_h = fix.cpu().numpy().astype(np.float64), nfix.cpu().numpy()          # the batch's single device->host copy
        for trial in range(repeat_num):
            for hi, allocated in enumerate((True, False)):
                fvs = to_fix_vectors(fix_h[trial, hi], n_h[trial, hi])
                all_gt.extend(gt_fix_vectors)
                all_perf.extend(performances)
                all_alloc.extend([allocated] * N)
                all_pred.extend(fvs)
                for index in range(N):
                    arr = np.array(fvs[index].tolist()).reshape(-1, 3)
                    predict_results.append({"img_names": batch["img_names"][index], "qid": batch["question_ids"][index],
                                            "repeat_id": trial + 1, "performance": allocated, "X": list(arr[:, 0]),
                                            "Y": list(arr[:, 1]), "T": list(arr[:, 2] * 1000), "length": len(arr)})
    cur_metrics, cur_metrics_std, scores = evaluation_performance_related(all_gt, all_pred, all_perf, all_alloc, multimatch)
    return cur_metrics, cur_metrics_std, scores, predict_results
