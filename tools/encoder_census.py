"""GPU diagnostic: every GEMM launch of the dilated ResNet-50 encoder (forward + backward, bs 32, 320x512) with its shape,
HIP-event time, TFLOP/s, operand + result bytes, and the time an ideal kernel would need (max of the 3-product matrix time at the
practical 530 TFLOP/s of DESIGN 9e and the byte time at 5.5 TB/s).  Sorted by the gap to that ideal: where the encoder's time is.
    python tools/encoder_census.py [--batch 32]"""
import argparse, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--hw2-min-flops", type=float, default=None, help="experiment: route smaller weight gradients to hw2_kernel")
    ap.add_argument("--only", type=str, default=None, help="print only rows whose kind contains this")
    a = ap.parse_args()
    from scanpaths_amd import hip
    if a.hw2_min_flops is not None:
        from scanpaths_amd import functional as F
        F.HW2_SINGLE_MIN_FLOPS = a.hw2_min_flops
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    dev = torch.device("cuda:0")
    m = baseline(convLSTM_length=16, map_width=64, map_height=40)
    fill_module(m, seed=0)
    m = m.to(dev).train()
    x = make_batch("AiR", a.batch, 320, 512, 16, seed=0)["images"].to(dev)

    def fwdbwd():
        y = m.encode(x)
        y.backward(torch.ones_like(y))
        for p in m.parameters():
            p.grad = None

    fwdbwd(); fwdbwd(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); fwdbwd(); e1.record(); torch.cuda.synchronize()
    whole = e0.elapsed_time(e1)
    hip.TIMER = hip.KernelTimer(min_flops=0)
    fwdbwd(); torch.cuda.synchronize()
    rows = []
    for key, d in hip.TIMER.summary().items():
        kind, M, N, K = key[0], key[1], key[2], key[3]
        taps = int(key[4].split("x")[0]) ** 2
        if "wgrad" in kind:      # operands dY [M][N] and X [M][K / taps] as two fp16 planes each; result N x K fp32
            byts = 4.0 * M * N + 4.0 * M * (K // taps) + 4.0 * N * K
        else:                    # activation [M][K / taps] planes (stride-1 same-size maps), weights, fp32 result [M][N]
            byts = 4.0 * M * (K // taps) + 4.0 * N * K + 4.0 * M * N
        rows.append((kind, M, N, K, d["launches"], d["avg_ms"], d["ms"], d["tflops"], d["flops_per_launch"], byts))
    hip.TIMER = None
    tot = sum(r[6] for r in rows)
    print(f"encoder forward+backward {whole:.2f} ms; GEMM launches (event-bracketed, serialising) {tot:.2f} ms")
    gap = lambda r: r[6] - r[4] * max(r[8] / 530e12, r[9] / 5.5e12) * 1e3
    rows.sort(key=lambda r: -gap(r))
    print("sorted by (time - ideal); ideal = max(3-product matrix time at 530 TFLOP/s, bytes at 5.5 TB/s)")
    for r in rows:
        if a.only and a.only not in r[0]:
            continue
        kind, M, N, K, n, avg, ms, tf, fl, by = r
        ideal = max(fl / 530e12, by / 5.5e12) * 1e6
        print(f"{kind[:18]:18s} M={M:8d} N={N:5d} K={K:6d} n={n:3d} avg {avg*1e3:7.1f} us  total {ms:6.2f} ms  {tf:6.1f} TF/s  "
              f"{by/avg/1e9:6.2f} TB/s  ideal {ideal:6.1f} us ({'hbm' if by/5.5e12 > fl/530e12 else 'mfma'})  gap {gap(r):6.2f} ms")


if __name__ == "__main__":
    main()
