"""GPU check of BASELINE.json config 5 (AiR inference, bs=128, 320x512, T=16): eager eval forward + sampling, and HIP-graph
capture of the whole eval forward (fixed shapes)."""
import os, sys, time, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd.models.baseline_attention import baseline
from scanpaths_amd.models.sampling import Sampling
from scanpaths_amd.procedural import fill_module
from scanpaths_amd.synth import make_batch
dev = torch.device("cuda:0")
B, H, W, T = int(os.environ.get("B", 128)), 320, 512, 16
m = baseline(convLSTM_length=T, map_width=W // 8, map_height=H // 8); fill_module(m, 0); m = m.to(dev).eval()
b = {k: v.to(dev) for k, v in make_batch("AiR", B, H, W, T, seed=0).items()}
with torch.no_grad():
    out = m(b["images"], b["attention_maps"]); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        out = m(b["images"], b["attention_maps"])
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
print(f"eager eval forward bs={B}: {dt*1e3:.1f} ms -> {B/dt:.1f} img/s; mem {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
s = Sampling(convLSTM_length=T, min_length=1, map_width=W // 8, map_height=H // 8, width=W, height=H)
smp = s.random_sample(out["good_all_actions_prob"], out["good_log_normal_mu"], out["good_log_normal_sigma2"])
fix, am, dm = s.generate_scanpath(b["images"], smp["selected_actions_probs"], smp["durations"], smp["selected_actions"])
print("sampled scanpath lengths (first 8):", [len(f) for f in fix[:8]], "prob row sums", float(out["good_all_actions_prob"].sum(-1).mean()))
# ---- HIP graph capture of the eval forward ----
try:
    static_img, static_att = b["images"].clone(), b["attention_maps"].clone()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), torch.no_grad():
        m(static_img, static_att)            # warm-up on the capture stream (workspaces, attribute sets)
    torch.cuda.current_stream().wait_stream(side)
    with torch.no_grad(), torch.cuda.graph(g):
        gout = m(static_img, static_att)
    g.replay(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2):
        g.replay()
    torch.cuda.synchronize()
    dtg = (time.perf_counter() - t0) / 2
    err = max(float((gout[k] - out[k]).abs().max()) for k in out)
    print(f"graph replay bs={B}: {dtg*1e3:.1f} ms -> {B/dtg:.1f} img/s; max |graph - eager| = {err:.3e}")
except Exception as e:
    print("graph capture failed:", type(e).__name__, str(e)[:300])
