"""The north-star's "ResNet-50 conv stage" sub-metric: dilated ResNet-50 encoder alone, bs 32, 320x512,
forward and forward+backward, timed with HIP events; algorithmic FLOPs = 71.35 GMAC/img x 2 (x3 for training).
    python tools/bench_backbone.py [--batch 32] [--reps 5]"""
import argparse, json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    dev = torch.device("cuda:0")
    m = baseline(convLSTM_length=16, map_width=64, map_height=40)
    fill_module(m, seed=0)
    m = m.to(dev).train()
    x = make_batch("AiR", a.batch, 320, 512, 16, seed=0)["images"].to(dev)
    gmac = 71.35e9 * a.batch

    def fwd():
        return m.encode(x)

    def fwdbwd():
        y = m.encode(x)
        y.backward(torch.ones_like(y))
        for p in m.parameters():
            p.grad = None

    out = {}
    for name, fn, mult in (("forward", fwd, 1), ("forward+backward", fwdbwd, 3)):
        with torch.set_grad_enabled(name != "forward"):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / a.reps
        tf = 2 * gmac * mult / ms / 1e9
        out[name] = {"ms": ms, "tflops": tf, "frac_of_2xfp16_split_ceiling_833.3": tf / 833.3,
                     "frac_of_3xbf16_split_ceiling_416.7": tf / 416.7, "frac_of_fp32_mfma_157.3": tf / 157.3,
                     "frac_of_16bit_dense_2500": tf / 2500.0}
    print(json.dumps({"workload": f"dilated ResNet-50 encoder, bs {a.batch}, 320x512, train-mode BN, fp32-faithful", **out}))


if __name__ == "__main__":
    main()
