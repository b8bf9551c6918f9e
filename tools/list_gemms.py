"""GPU diagnostic: list every GEMM launch shape of one AiR train step with its time (HIP events)."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import hip
from scanpaths_amd.models.baseline_attention import baseline
from scanpaths_amd.models.loss import supervised_loss
from scanpaths_amd.optim import FlatAdam
from scanpaths_amd.procedural import fill_module
from scanpaths_amd.synth import make_batch
dev = torch.device("cuda:0")
B, H, W, T = 32, 320, 512, 16
model = baseline(convLSTM_length=T, map_width=W // 8, map_height=H // 8); fill_module(model, 0); model = model.to(dev).train()
opt = FlatAdam(model.parameters(), lr=1e-4, weight_decay=5e-5, clip=12.5)
b = {k: v.to(dev) for k, v in make_batch("AiR", B, H, W, T, seed=0).items()}
def step():
    opt.zero_grad()
    pred = model(b["images"], b["attention_maps"], b["performances"])
    loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
    loss.backward(); opt.step()
step()
hip.TIMER = hip.KernelTimer(min_flops=0)
step(); torch.cuda.synchronize()
rows = sorted(hip.TIMER.summary().items(), key=lambda kv: -kv[1]["ms"])
tot = sum(d["ms"] for _, d in rows)
print(f"total GEMM ms {tot:.1f}")
for k, d in rows[:int(os.environ.get("NROWS", "45"))]:
    print(f"{str(k):70s} n={d['launches']:3d} avg {d['avg_ms']:8.3f} ms  total {d['ms']:8.2f}  {d['tflops']:7.1f} TF/s")
