"""Drop-in for ``models/sampling.py`` (Sampling.random_sample :16-46, .generate_scanpath :48-77), on the device.

Both methods run HIP kernels (csrc/sampling.hip): a masked categorical draw per (b,t) by inverse CDF with a Philox counter
RNG, log-normal durations, the first-terminate scan and the index -> pixel mapping; ``generate_scanpath`` does ONE device->host
copy for the list of numpy structured arrays the reference returns (the reference loops B*T ``.cpu().numpy()`` calls).
The random stream is this library's own (seeded per call from ``self.seed`` + a call counter); like the reference's CUDA
stream it is not torch's CPU stream.  All index arithmetic is bit-exact with the reference (tests/golden/sampling.npz)."""
import numpy as np
import torch

from .. import hip
from ..hip import check, ptr


class Sampling():
    def __init__(self, convLSTM_length=16, min_length=2, map_width=40, map_height=30, width=320, height=240, seed=0):
        self.convLSTM_length = convLSTM_length
        self.min_length = min_length
        self.map_width = map_width
        self.map_height = map_height
        self.width = width
        self.height = height
        self.x_granularity = float(self.width / self.map_width)
        self.y_granularity = float(self.height / self.map_height)
        self.seed = seed
        self._calls = 0

    def _scan(self, actions, durations):
        B, T = actions.shape
        dev = actions.device
        length = torch.empty(B, dtype=torch.float32, device=dev)
        am = torch.empty((B, T), dtype=torch.float32, device=dev)
        dm = torch.empty((B, T), dtype=torch.float32, device=dev)
        fix = torch.empty((B, T, 3), dtype=torch.float32, device=dev)
        nfix = torch.empty(B, dtype=torch.int32, device=dev)
        check(hip.lib().sp_generate_scanpath(ptr(actions), ptr(durations), B, T, self.map_width, self.map_height, self.width,
                                             self.height, ptr(length), ptr(am), ptr(dm), ptr(fix), ptr(nfix), hip.stream()),
              "sp_generate_scanpath")
        return length, am, dm, fix, nfix

    def random_sample(self, all_actions_prob, log_normal_mu, log_normal_sigma2):
        probs = all_actions_prob.detach().contiguous().float()
        mu = log_normal_mu.detach().contiguous().float()
        s2 = log_normal_sigma2.detach().contiguous().float()
        B, T, A = probs.shape
        dev = probs.device
        actions = torch.empty((B, T), dtype=torch.int64, device=dev)
        aprob = torch.empty((B, T), dtype=torch.float32, device=dev)
        dur = torch.empty((B, T), dtype=torch.float32, device=dev)
        self._calls += 1
        # one process per GPU: every rank must draw its own stream (the Philox counter is the LOCAL row index), so the rank
        # enters the key; single-process runs (rank 0) keep the key of earlier versions
        rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
        seed = (int(self.seed) * 0x9E3779B97F4A7C15 + rank * 0xD1B54A32D192ED03 + self._calls) & 0xFFFFFFFFFFFFFFFF
        check(hip.lib().sp_sample_actions(ptr(probs), ptr(mu), ptr(s2), B, T, A, int(self.min_length), seed, ptr(actions),
                                          ptr(aprob), ptr(dur), hip.stream()), "sp_sample_actions")
        length, _, _, _, _ = self._scan(actions, dur)
        if all_actions_prob.requires_grad:      # RL phase differentiates through the gathered probabilities (:22-23)
            aprob = torch.gather(all_actions_prob, 2, actions.unsqueeze(-1)).squeeze(-1)
        return {"scanpath_length": length.unsqueeze(-1), "durations": dur, "selected_actions_probs": aprob,
                "selected_actions": actions}

    def generate_scanpath(self, images, prob_sample_actions, durations, sample_actions):
        acts = sample_actions.detach().to(torch.int64).contiguous()
        durs = durations.detach().float().contiguous()
        if not acts.is_cuda:
            raise hip.HipError("Sampling.generate_scanpath needs device tensors (no CPU path)")
        _, am, dm, fix, nfix = self._scan(acts, durs)
        fix_h, n_h = fix.cpu().numpy().astype(np.float64), nfix.cpu().numpy()
        out = []
        for b in range(acts.shape[0]):
            fv = np.zeros(int(n_h[b]), dtype={"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")})
            fv["start_x"], fv["start_y"], fv["duration"] = fix_h[b, :n_h[b], 0], fix_h[b, :n_h[b], 1], fix_h[b, :n_h[b], 2]
            out.append(fv)
        return out, am.to(images.dtype), dm.to(images.dtype)

    def beam_search(self, all_actions_prob, log_normal_mu, log_normal_sigma2, beam=4):
        """BASELINE.json config 5 ("beam-4 scanpath sampling"); build-side decoder, the reference only samples (:16-46).
        The ``beam`` most probable action sequences per sample under sum_t log p_t(a_t) (terminate ends a sequence, allowed
        from t >= min_length; like any sum-of-log-probability decoder it prefers short scanpaths: every further fixation costs
        log p < 0) by csrc/sampling.hip ``beam_kernel``; durations are the medians of the predicted log-normals,
        exp(mu) (random_sample's exp(eps*sigma2 + mu) at eps = 0).  Returns {"selected_actions" [B,beam,T] int64,
        "scores" [B,beam] float64 (log-probability, best first), "durations" [B,beam,T], "scanpath_length" [B,beam]}; feed
        ``selected_actions[:, k]`` / ``durations[:, k]`` to ``generate_scanpath`` for fixation vectors."""
        probs = all_actions_prob.detach().contiguous().float()
        B, T, A = probs.shape
        dev = probs.device
        actions = torch.empty((B, beam, T), dtype=torch.int64, device=dev)
        scores = torch.empty((B, beam), dtype=torch.float64, device=dev)
        check(hip.lib().sp_beam_search(ptr(probs), B, T, A, int(self.min_length), int(beam), ptr(actions), ptr(scores),
                                       hip.stream()), "sp_beam_search")
        dur = torch.exp(log_normal_mu.detach().float()).unsqueeze(1).expand(B, beam, T).contiguous()
        length, _, _, _, _ = self._scan(actions.view(B * beam, T), dur.view(B * beam, T))
        return {"selected_actions": actions, "scores": scores, "durations": dur, "scanpath_length": length.view(B, beam)}
