"""Which host-side torch ops launch the tiny elementwise kernels of a training step (adds / copies / fills / cats of a few hundred
elements)?  One bs-32 bench step under torch.profiler with shapes; counts per (op, shapes).   python3 tools/small_ops_profile.py"""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench  # noqa: E402
from scanpaths_amd.models.loss import supervised_loss  # noqa: E402
from scanpaths_amd.optim import FlatAdam  # noqa: E402
from scanpaths_amd.synth import make_batch  # noqa: E402

sys.argv = ["bench.py", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"]
args = bench.parse()
dev = torch.device("cuda:0")
model = bench.build_model(args, dev)
model.train()
b = {k: v.to(dev) for k, v in make_batch("AiR", args.batch, args.height, args.width, args.T, seed=0).items()}
opt = FlatAdam(model.parameters(), lr=1e-4, weight_decay=5e-5, clip=12.5)


def step():
    opt.zero_grad()
    pred = bench.call_model(model, args, b, True)
    loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0, None)
    loss.backward()
    opt.step()


step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::clone", "aten::stack", "aten::zeros",
                  "aten::contiguous", "aten::sum", "aten::mul", "aten::index_select", "aten::select_backward", "aten::slice_backward"):
        shapes = str(e.input_shapes)[:70]
        st = [f for f in (e.stack or []) if "scanpaths_amd" in f or "bench.py" in f]
        cnt[(e.name, shapes, st[0][-70:] if st else "(autograd engine)")] += 1
for k, v in sorted(cnt.items(), key=lambda x: -x[1])[:45]:
    print(v, k)
