"""Timing + checksum of the duration-site head kernels (csrc/head_direct.hip) at the benchmark size (B 32, 40x64 map, C 512, two head
slots): forward, data gradient, weight gradient.    python3 tools/bench_drt.py [tag]      (SP_LIBRARY=timing loads the other build)"""
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from scanpaths_amd import config as _sp_config  # noqa: E402
_sp_config.honour_env_for_tools()      # SP_LIBRARY=timing of the command line
from scanpaths_amd import functional as F, hip  # noqa: E402

dev = torch.device("cuda:0")
B, Hm, Wm, C, nsel, nheads = 32, 40, 64, 512, 2, 2
g = torch.Generator(device="cpu").manual_seed(0)
ncls_y = len({tuple(0 <= 5 * s - 4 + 2 + k < Hm for k in range(7)) for s in range((Hm + 4 - 7) // 5 + 1)})
ncls_x = len({tuple(0 <= 5 * s - 4 + 2 + k < Wm for k in range(7)) for s in range((Wm + 4 - 7) // 5 + 1)})
S = ((Hm + 4 - 7) // 5 + 1) * ((Wm + 4 - 7) // 5 + 1)
h = torch.randn(B, Hm, Wm, C, generator=g).to(dev).requires_grad_(True)
hmap = torch.arange(nsel, dtype=torch.int32).repeat(B, 1).contiguous().to(dev)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


res = {"lib": os.path.basename(hip.LIB_PATH)}
for ncls in (ncls_y * ncls_x,):
    W11 = (torch.randn(nheads, ncls, 121, C, generator=g) * 0.02).to(dev).requires_grad_(True)
    cbsum = torch.randn(nheads, ncls, generator=g).to(dev).requires_grad_(True)
    D = F.drt_direct(h, W11, cbsum, hmap, nsel)
    dD = torch.randn(D.shape, generator=g).to(dev)
    with torch.no_grad():
        res["fwd_us"] = round(timed(lambda: F.drt_direct(h, W11, cbsum, hmap, nsel)), 1)
    L = hip.lib()
    dh = torch.empty_like(h)
    dW = torch.empty_like(W11)
    dcs = torch.empty_like(cbsum)
    ws = hip.workspace(L.sp_drt_direct_bwd_weight_workspace(B, Hm, Wm, C, nsel), dev, slot=0)
    p = hip.ptr
    res["bwd_data_us"] = round(timed(lambda: hip.check(L.sp_drt_direct_bwd_data(p(dD), p(W11), p(hmap), B, Hm, Wm, C, nsel, 0, p(dh), hip.stream()), "d")), 1)
    res["bwd_weight_us"] = round(timed(lambda: hip.check(L.sp_drt_direct_bwd_weight(p(dD), p(h), p(hmap), B, Hm, Wm, C, nsel, nheads, p(ws), p(dW), p(dcs),
                                                                                   hip.stream()), "w")), 1)
    torch.cuda.synchronize()
    res["sha_dW"] = hashlib.sha256(dW.detach().cpu().numpy().tobytes()).hexdigest()[:16]
    res["sha_dh"] = hashlib.sha256(dh.cpu().numpy().tobytes()).hexdigest()[:16]
    res["sha_D"] = hashlib.sha256(D.detach().cpu().numpy().tobytes()).hexdigest()[:16]
    # fp64 check of dW on one (head, class, tap): direct evaluation of the definition
    res["ncls"] = ncls
print(json.dumps(res))
