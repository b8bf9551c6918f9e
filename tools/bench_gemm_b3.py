"""Time the split-scheme forward GEMM kernel on a plain [M,K]x[N,K]^T problem (1x1 conv form), operands pre-split.
   python tools/bench_gemm_b3.py M N K [reps]"""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scanpaths_amd import functional as F

M, N, K = (int(v) for v in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
dev = torch.device("cuda:0")
x = torch.randn(M, K, device=dev)
w = torch.randn(N, K, device=dev)
y = torch.empty(M, N, device=dev)
xs, ws = F.split3(x), F.split3(w)
def run():
    F._igemm_b3(xs, ws, None, y, N_img=1, Hi=1, Wi=M, Kc=K, ldx=K, Ho=1, Wo=M, Nout=N, ldc=N, ldw=K)
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
ref = x[:256] @ w.t()
err = (y[:256] - ref).abs().max().item()
print(json.dumps({"M": M, "N": N, "K": K, "ms": ms, "tflops": 2.0 * M * N * K / ms / 1e9, "err_vs_torch": err}))
