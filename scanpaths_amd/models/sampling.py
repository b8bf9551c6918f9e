"""Drop-in for ``models/sampling.py`` (Sampling.random_sample :16-46, .generate_scanpath :48-77).

Round 1: host-side restatement of the post-hoc sampler ("next"); the per-step
categorical / log-normal draws use torch's RNG on the tensors' own device.  The device kernel for the
first-terminate scan + index->(x,y) mapping is scheduled after the train path (§8 f1)."""
import numpy as np
import torch


class Sampling():
    def __init__(self, convLSTM_length=16, min_length=2, map_width=40, map_height=30, width=320, height=240):
        self.convLSTM_length = convLSTM_length
        self.min_length = min_length
        self.map_width = map_width
        self.map_height = map_height
        self.width = width
        self.height = height
        self.x_granularity = float(self.width / self.map_width)
        self.y_granularity = float(self.height / self.map_height)

    def random_sample(self, all_actions_prob, log_normal_mu, log_normal_sigma2):
        probs = all_actions_prob.detach().clone()
        probs[:, :self.min_length, 0] = 0
        selected = torch.distributions.categorical.Categorical(probs=probs).sample()
        selected_probs = torch.gather(all_actions_prob, 2, selected.unsqueeze(-1)).squeeze(-1)
        eps = torch.randn(log_normal_mu.shape, device=log_normal_mu.device)
        durations = torch.exp(eps * log_normal_sigma2 + log_normal_mu)     # sigma2 used as the scale: reference quirk :27
        is_term = selected == 0
        first = torch.where(is_term.any(1), is_term.float().argmax(1), torch.full_like(selected[:, 0], 0)).to(
            all_actions_prob.dtype)
        first[first == 0] = self.convLSTM_length                           # terminate at t=0 or never -> T (:33)
        return {"scanpath_length": first.unsqueeze(-1), "durations": durations, "selected_actions_probs": selected_probs,
                "selected_actions": selected}

    def generate_scanpath(self, images, prob_sample_actions, durations, sample_actions):
        acts = sample_actions.detach().cpu().numpy()
        drts = durations.detach().cpu().numpy()
        N, T = acts.shape
        amask = np.zeros((N, T), dtype=np.float32)
        dmask = np.zeros((N, T), dtype=np.float32)
        fix_vectors = []
        for n in range(N):
            term = np.nonzero(acts[n] == 0)[0]
            L = int(term[0]) if len(term) else T
            amask[n, :min(L + 1, T)] = 1
            dmask[n, :L] = 1
            idx = acts[n, :L] - 1
            fv = np.zeros(L, dtype={"names": ("start_x", "start_y", "duration"), "formats": ("f8", "f8", "f8")})
            fv["start_x"] = (idx % self.map_width) * self.x_granularity + self.x_granularity / 2
            fv["start_y"] = (idx // self.map_width) * self.y_granularity + self.y_granularity / 2
            fv["duration"] = drts[n, :L]
            fix_vectors.append(fv)
        return fix_vectors, images.new_tensor(amask), images.new_tensor(dmask)
