#!/usr/bin/env python3
"""bench.py -- AiR supervised train step (fwd + loss + bwd + clip + Adam [+ RCCL grad all-reduce]) on MI355X.

  python bench.py --gpus 1 --steps K --warmup W                      (single GPU)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Default workload = BASELINE.json configs[1]: AiR, ResNet-50 encoder, 16-step decode, bs=32 per GPU, synthetic 320x512 images
(attention map 40x64 instead of the reference's hard-coded 30x40; "14-token questions" reach the model only as the
attention map).  A "step" is one pass of the hot path over one synthetic batch already resident in HBM.
Weak scaling: every rank processes its own bs=32 shard; the only data-path exchange is the bucketed gradient all-reduce of
the flat buffer inside FlatAdam (overlapped with backward) plus two mask-sum scalars.  Rank 0 prints ONE JSON line.

Other BASELINE.json configurations through flags (not the headline line; same JSON schema, own metric string):
  --task osie --arch resnet18 --T 8 --batch 4 --height 240 --width 320     config 1 (the reference's CPU-runnable case)
  --task coco --batch 16 --T 6                                             config 4's per-GPU shard (bs 64 over 4 GPUs)
  --mode infer --batch 128                                                  config 5: eval forward + 10 sampled scanpaths/head
  --precision f16x1                                                         throughput mode (single fp16 plane, 1 MFMA product):
                                                                            separately labelled, NOT the fp32-faithful headline
"""
import argparse
import json
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# /opt/skills/guides/MI355X_MICROARCH.md §Matrix cores: dense peaks
PEAK_FP32_MFMA_TFLOPS = 157.3
PEAK_F16_MFMA_TFLOPS = 2500.0
PRODUCTS = {"h2": 3, "b3": 6, "h1": 1, "igemm": None, "wgrad": None}      # MFMA products per algorithmic multiply-add
KERNEL_NAMES = {
    "h2_fwd": "h2_kernel<fwd> (2xfp16 split, 3 MFMA products)", "h2_dgrad": "h2_kernel<dgrad> (2xfp16 split, 3 MFMA products)",
    "h2_wgrad": "hw_kernel (2xfp16 split weight gradient, 3 MFMA products)",
    "h2_wgrad_multi": "hw2_kernel (2xfp16 split weight gradient of ALL applications of the h-gate conv in one launch, 3 MFMA products)",
    "h2_wgrad_multi_rows": "hw2_kernel (2xfp16 split weight gradient of ALL applications of the h-gate conv in one launch, 3 MFMA products; "
                           "samples behind their last masked-in step skipped)",
    "h2_dgrad_rows": "h2_kernel<dgrad> (2xfp16 split, 3 MFMA products; zero tiles for samples behind their last masked-in step)",
    "h1_fwd": "h2_kernel<fwd, 1 plane> (fp16 in / fp32 acc, 1 product)", "h1_dgrad": "h2_kernel<dgrad, 1 plane>",
    "h1_wgrad": "hw_kernel<1 plane>",
    "b3_fwd": "b3_kernel<fwd> (3xbf16 split, 6 MFMA products)", "b3_dgrad": "b3_kernel<dgrad>", "b3_wgrad": "w3_kernel",
    "igemm_fwd": "igemm_kernel<fwd> (fp32 MFMA)", "igemm_dgrad": "igemm_kernel<dgrad> (fp32 MFMA)", "wgrad": "wgrad_kernel (fp32 MFMA)"}
PMC_PREFIX = {"h2_fwd": "h2_kernel<0,", "h2_dgrad": "h2_kernel<1,", "h2_wgrad": "hw_kernel", "h2_wgrad_multi": "hw2_kernel",
              "h2_dgrad_rows": "h2_kernel<1,", "h2_wgrad_multi_rows": "hw2_kernel"}      # kernel-name prefixes in profiles/*_pmc_hconv.json
# the forward launches of the h-gate shape are (15 of 16) the ConvLSTM-fused variant: its own PMC entry (7th template argument true)
PMC_FUSED_FWD = ", true, true, true"



def bucketer_overhead(opt, step, sync, steps, bucketed_ms):
    """The 1-GPU half of the scaling evidence: the step with the data-parallel machinery active in an RCCL world
    of one -- post-accumulate-grad hooks, ~32 MB buckets launched in descending order from the hooks, async ncclAllReduce on RCCL's
    stream (an identity at world 1, but the launches, the stream hand-offs and the waits are real), FlatAdam.step() waiting on the
    handles -- against the plain step of the SAME process (bucketer detached), interleaved plain / bucketed / plain; and the bucket
    timeline: how long before the END of backward each bucket's all-reduce was launched (= the window its xGMI transfer can hide
    in; the bucket of the encoder's first layers is launched last).  No scaling curve: that needs more than one device."""
    import torch
    bk = opt._bucketer

    def timed(n):
        sync()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        sync()
        return (time.perf_counter() - t0) / n * 1e3

    opt._bucketer = None
    plain_a = timed(steps)
    opt._bucketer = bk
    bucketed_b = timed(steps)
    opt._bucketer = None
    plain_b = timed(steps)
    opt._bucketer = bk
    # one instrumented step: events at every bucket launch (compute stream) and at the end of backward
    bk.record = True
    bk.host_launch_ms = bk.host_wait_ms = 0.0
    end_bwd = torch.cuda.Event(enable_timing=True)
    orig_step = opt.step

    def step_hook(*a, **k):
        end_bwd.record()
        return orig_step(*a, **k)
    opt.step = step_hook
    step()
    sync()
    opt.step = orig_step
    bk.record = False
    # what a bucket's all-reduce would take on an 8-GPU xGMI node (7 links x ~77 GB/s each way per GPU): a ring over ONE
    # link moves 2 (N - 1) / N of the bytes, a direct reduce-scatter + all-gather over all 7 links a seventh of that -- the part of it
    # that does not fit into the window before the end of backward is exposed
    def xgmi_ms(nbytes, links):
        return 2.0 * 7 / 8 * nbytes / (links * 77e9) * 1e3
    buckets = []
    for b, nb, ev in bk.last_ready_events:
        window = ev.elapsed_time(end_bwd)
        buckets.append({"bucket": b, "mb": round(nb / 2 ** 20, 1), "launched_ms_before_end_of_backward": round(window, 2),
                        "expected_allreduce_ms_8gpu": {"ring_one_link": round(xgmi_ms(nb, 1), 3), "direct_7_links": round(xgmi_ms(nb, 7), 3)},
                        "expected_exposed_ms_8gpu": {"ring_one_link": round(max(0.0, xgmi_ms(nb, 1) - window), 3),
                                                     "direct_7_links": round(max(0.0, xgmi_ms(nb, 7) - window), 3)}})
    plain = 0.5 * (plain_a + plain_b)
    bucketed = 0.5 * (bucketed_ms + bucketed_b)
    return {"world": 1, "backend": "nccl (RCCL)", "bucket_mb": 32, "first_bucket_mb": round((bk.ranges[0][1] - bk.ranges[0][0]) * 4 / 2 ** 20, 1),
            "n_buckets": len(bk.ranges),
            "plain_ms_per_step": [round(plain_a, 2), round(plain_b, 2)], "bucketed_ms_per_step": [round(bucketed_ms, 2), round(bucketed_b, 2)],
            "ddp_overhead_ms": round(bucketed - plain, 2),
            # where the overhead of a world of one goes: host time of the instrumented step inside the collectives' launches (on the autograd
            # thread, i.e. in front of the ~2300 kernel launches the backward still has to enqueue) and inside handle.wait(); the rest is
            # RCCL's own identity kernels / stream hand-offs on the device
            "host_ms_in_allreduce_launches": round(bk.host_launch_ms, 3), "host_ms_in_handle_waits": round(bk.host_wait_ms, 3),
            "bucket_timeline": buckets,
            "note": "order: bucketed (the timed region of this line), plain, bucketed, plain -- same process, same box; identity all-reduce "
                    "(world 1): launch / hook / stream hand-off cost only, no xGMI traffic; no scaling curve has been measured"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--task", type=str, default="air", choices=["air", "osie", "coco"])
    ap.add_argument("--mode", type=str, default="train", choices=["train", "infer"])
    ap.add_argument("--precision", type=str, default="f32", choices=["f32", "f16x1"],
                    help="f32: fp32-faithful split GEMMs (headline); f16x1: single fp16 plane, fp32 accumulate (throughput mode)")
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch")
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--T", type=int, default=16)
    ap.add_argument("--arch", type=str, default="resnet50")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dense-backward", action="store_true",
                    help="config row_sparsity off: multiply the exact zeros behind every sample's last masked-in decode step like the reference "
                         "does (default: the decoder's backward derives them from the gradient that reaches its outputs and skips them, results "
                         "identical -- the line reports the dense time of the same process too)")
    ap.add_argument("--dropin", action="store_true",
                    help="make the reference's literal call sequence (AiR/train.py:188-205: two loss calls, clip_grad_norm_, torch.optim.Adam, "
                         "LambdaLR) the timed region of the line instead of the fused-loss / FlatAdam step (default: timed as a secondary leg, "
                         "JSON key 'dropin')")
    ap.add_argument("--no-dropin-leg", action="store_true", help="do not time the drop-in call sequence after the timed region")
    ap.add_argument("--no-dense-leg", action="store_true",
                    help="do not time the dense backward after the timed region (kernel traces of the headline step only)")
    ap.add_argument("--force-bucketer", action="store_true",
                    help="1-GPU half of the scaling evidence: run the data-parallel machinery in an RCCL world of ONE (post-accumulate hooks, "
                         "32 MB buckets, async all-reduce on RCCL's stream) and report its cost next to the plain step (JSON key 'ddp')")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="threads for the CPU baseline; 0 = the best count of the recorded sweep over {32, 64, 128} on this pool's host "
                         "(profiles/r06_cpu_thread_sweep.json, written by --cpu-sweep), 32 when no sweep is committed")
    ap.add_argument("--cpu-batch", type=int, default=4, help="images per CPU-baseline step (BASELINE.md §4: bs 4; every step is MEASURED "
                                                             "at the full T, nothing is extrapolated)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU steps after one warm-up step (BASELINE.md §4: 3, median)")
    ap.add_argument("--cpu-sweep", action="store_true",
                    help="run ONLY the CPU-baseline thread sweep (32 / 64 / 128 threads, 1 warm-up + 1 timed step each at --cpu-batch images), "
                         "print it as one JSON line and exit -- the line is committed as profiles/r06_cpu_thread_sweep.json")
    ap.add_argument("--no-length-leg", action="store_true",
                    help="do not time the step on the second scanpath-length law (L ~ U{T/2..T}) after the timed region")
    return ap.parse_args()


def reduced_fwd_gmac(arch, H, W, T, task="air"):
    """SURVEY.md §8(d) / BASELINE.md §3, 'reduced' column (necessary work after the exact hoistings), scaled to the config:
    every term is proportional to the map area; decoder terms other than the hoisted x-gate conv are proportional to T."""
    s = H * W / (320.0 * 512.0)
    enc = {"resnet50": 71.35, "resnet18": 34.11}[arch] * s
    sal = 24.16 * s * (1.0 if arch == "resnet50" else 0.25)
    per_step_other = (1.36 + 4.18 + 0.4) / 16.0 * s * (1.0 if task == "air" else 0.5)
    return enc + sal + 24.16 * s + (T - 1) * 24.16 * s + T * per_step_other


def host_info():
    info = {"logical_cpus": os.cpu_count()}
    try:
        out = subprocess.run(["lscpu"], capture_output=True, text=True, timeout=10).stdout
        kv = {l.split(":", 1)[0].strip(): l.split(":", 1)[1].strip() for l in out.splitlines() if ":" in l}
        info["model"] = kv.get("Model name", "?")
        sockets, cps = int(kv.get("Socket(s)", "1")), int(kv.get("Core(s) per socket", "0") or 0)
        info["sockets"], info["physical_cores"] = sockets, sockets * cps if cps else None
    except Exception as e:      # lscpu missing: keep going with what os reports
        info["model"] = f"unknown ({type(e).__name__})"
    return info


def cpu_baseline(args):
    """The oracle (literal CPU restatement of the reference, kind "port", pinned to the real reference by tests/golden) timed on
    the host cores, BASELINE.md §4 protocol on a bounded sample: the SAME workload (task, image size, full T) at a small batch,
    one untimed warm-up step (thread pool / oneDNN primitive creation) then --cpu-steps MEASURED steps, median, with the
    forward / backward / clip+Adam split.  Nothing is extrapolated."""
    from oracle import scanpath_oracle as O
    from scanpaths_amd.procedural import procedural_state_dict
    from scanpaths_amd.spec import is_buffer, model_spec
    from scanpaths_amd.synth import make_batch
    hi = host_info()
    sweep = cpu_sweep_record()
    want = args.cpu_threads or (sweep or {}).get("best_threads") or 32
    cores = min(want, os.cpu_count())
    torch.set_num_threads(cores)
    task = {"air": "AiR", "osie": "OSIE", "coco": "COCO_Search18"}[args.task]
    Hm, Wm = args.height // 8, args.width // 8
    sd = procedural_state_dict(model_spec(task, args.arch, Hm, Wm), seed=0)
    NB, T = args.cpu_batch, args.T
    params = {k: v.clone().requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not is_buffer(k)}
    full = dict(sd)
    full.update(params)
    state = {}
    rows = []
    for it in range(1 + args.cpu_steps):
        batch = make_batch(task, NB, args.height, args.width, T, seed=it)
        for p in params.values():
            p.grad = None
        t0 = time.perf_counter()
        pred = O.forward(full, task, batch["images"], batch["attention_maps"], batch["performances"], batch["tasks"],
                         training=True, T=T, arch=args.arch)
        loss, _, _ = O.supervised_loss(pred, batch)
        t1 = time.perf_counter()
        loss.backward()
        t2 = time.perf_counter()
        grads = {k: p.grad for k, p in params.items()}
        with torch.no_grad():
            O.clip_and_adam({k: p.data for k, p in params.items()}, grads, state, lr=1e-4, clip=12.5,
                            weight_decay=5e-5 if task == "AiR" else 5e-4)
        t3 = time.perf_counter()
        if it > 0:
            rows.append((t3 - t0, t1 - t0, t2 - t1, t3 - t2))
    rows.sort()
    med = rows[len(rows) // 2]
    return {"value": NB / med[0], "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"oracle {task} train step (fwd+loss+bwd+clip+Adam), {NB} images {args.height}x{args.width}, {args.arch}, "
                      f"T={T} measured (no extrapolation): 1 warm-up + {len(rows)} timed steps, median {med[0]:.1f} s/step "
                      f"(fwd {med[1]:.1f} s, bwd {med[2]:.1f} s, clip+Adam {med[3]:.2f} s); all steps: "
                      + ", ".join(f"{r[0]:.1f}" for r in rows),
            "s_per_step": round(med[0], 2), "fwd_s": round(med[1], 2), "bwd_s": round(med[2], 2), "opt_s": round(med[3], 3),
            "threads_used": cores, "host": hi,
            "thread_choice": ("--cpu-threads" if args.cpu_threads else
                              (f"best of the recorded sweep {sweep['images_per_s_by_threads']} on {sweep.get('host', {}).get('model', '?')} "
                               "(profiles/r06_cpu_thread_sweep.json)" if sweep else "32 (no sweep committed)"))}


def cpu_sweep_record():
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "r06_cpu_thread_sweep.json")))
    except Exception:
        return None


def cpu_sweep(args):
    """one-off: the CPU leg at 32 / 64 / 128 threads (1 warm-up + 1 timed step each), so that the thread count of the reported baseline
    is the best this host offers and the claim is a recorded measurement"""
    import copy
    res = {}
    for th in (32, 64, 128):
        if th > (os.cpu_count() or 1):
            continue
        a = copy.copy(args)
        a.cpu_threads, a.cpu_steps = th, 1
        r = cpu_baseline(a)
        res[str(th)] = round(r["value"], 4)
        print(f"bench.py --cpu-sweep: {th} threads -> {r['value']:.4f} img/s ({r['s_per_step']} s/step)", file=sys.stderr)
    best = max(res, key=lambda k: res[k])
    return {"images_per_s_by_threads": res, "best_threads": int(best), "images": args.cpu_batch, "workload": f"{args.task} {args.height}x{args.width} T={args.T} {args.arch}",
            "protocol": "oracle train step (fwd+loss+bwd+clip+Adam), 1 warm-up + 1 timed step per thread count", "host": host_info()}


def build_model(args, dev):
    from scanpaths_amd.procedural import fill_module
    Hm, Wm = args.height // 8, args.width // 8
    if args.task == "air":
        from scanpaths_amd.models.baseline_attention import baseline
        model = baseline(convLSTM_length=args.T, map_width=Wm, map_height=Hm, arch=args.arch)
    elif args.task == "osie":
        from scanpaths_amd.models.baseline_attention import baseline_osie
        model = baseline_osie(convLSTM_length=args.T, map_width=Wm, map_height=Hm, arch=args.arch)
    else:
        from scanpaths_amd.models.baseline_attention_multihead import baseline
        model = baseline(convLSTM_length=args.T, map_width=Wm, map_height=Hm, arch=args.arch)
    fill_module(model, seed=0)                     # identical replicas on every rank (FlatAdam broadcasts rank 0's anyway)
    return model.to(dev)


def call_model(model, args, b, training):
    if args.task == "air":
        return model(b["images"], b["attention_maps"], b["performances"] if training else None)
    if args.task == "osie":
        return model(b["images"])
    return model(b["images"], b["attention_maps"], b["tasks"])


def dropin_step(args, model, b):
    """The reference's own supervised iteration on the HIP model, call for call (AiR/train.py:116-117 optimizer, :156-167 LambdaLR,
    :188-205 the iteration; OSIE/train.py and COCO_Search18/train.py have the same shape): stock torch.optim.Adam, the two separate loss
    calls summed by autograd, torch.nn.utils.clip_grad_norm_, LambdaLR.step.  Only the tensorboard scalars of :206-210 are left out
    (logging is outside SURVEY 8's path).  Returns the step function."""
    from scanpaths_amd.models.loss import CrossEntropyLoss, MLPLogNormalDistribution
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, betas=(0.9, 0.999), eps=1e-08,
                                 weight_decay=5e-5 if args.task == "air" else 5e-4)
    iters_per_epoch, warmup_epoch, start_rl_epoch = 1000, 1, 10

    def lr_lambda(iteration):          # :156-164 (supervised branches)
        if iteration <= iters_per_epoch * warmup_epoch:
            return iteration / (iters_per_epoch * warmup_epoch)
        return 1 - (iteration - iters_per_epoch * warmup_epoch) / (iters_per_epoch * (start_rl_epoch - warmup_epoch))
    for g_ in optimizer.param_groups:          # (LambdaLR with last_epoch >= 0 resumes: it wants the groups' initial_lr, as a loaded checkpoint has)
        g_.setdefault("initial_lr", g_["lr"])
    lr_scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=lr_lambda, last_epoch=iters_per_epoch // 2)
    scanpaths, durations = b["scanpaths"], b["durations"]
    action_masks, duration_masks = b["action_masks"], b["duration_masks"]
    clip, lambda_1 = 12.5, 1.0

    def step():
        optimizer.zero_grad()
        predicts = call_model(model, args, b, True)
        loss_actions = CrossEntropyLoss(predicts["actions" if "actions" in predicts else "all_actions_prob"], scanpaths, action_masks)
        loss_duration = MLPLogNormalDistribution(predicts["log_normal_mu"], predicts["log_normal_sigma2"], durations, duration_masks)
        loss = loss_actions + lambda_1 * loss_duration
        loss.backward()
        if clip > 0:
            torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        optimizer.step()
        lr_scheduler.step()
        return loss
    return step


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` (N > 1) without a torch.distributed launcher around it: THIS process never touches the GPU (no HIP
    call, no exec of an initialised process); it starts N ranks -- one process per GPU, RCCL -- as a child
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` and relays
    rank 0's JSON line and the exit code.  The reference's counterpart is nn.DataParallel(model, gpu_ids) (AiR/train.py:169-170)."""
    import socket
    ndev = torch.cuda.device_count()              # counting devices does not initialise the GPU on this image
    if ndev < args.gpus and os.environ.get("SP_DIST_BACKEND", "nccl") == "nccl":
        print(f"bench.py: --gpus {args.gpus} asked for but only {ndev} HIP device(s) are visible; refusing to report a "
              f"{args.gpus}-GPU number from fewer devices (RCCL needs one device per rank)", file=sys.stderr)
        return 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL's hipIpcGetMemHandle otherwise fails)
    env.setdefault("OMP_NUM_THREADS", "8")
    return subprocess.run(cmd, env=env).returncode


def main():
    args = parse()
    if args.cpu_sweep:
        print(json.dumps(cpu_sweep(args)))
        return
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')} rank(s)")
    from scanpaths_amd import config as sp_config
    if args.precision == "f16x1":
        sp_config.set(split_scheme="f16x1")              # prints one loud line; the JSON line is tagged below
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    ndev = torch.cuda.device_count()
    dev_index = (local_rank % max(ndev, 1)) if world > 1 else 0
    if world > 1:
        torch.cuda.set_device(dev_index)
        # "nccl" is RCCL on ROCm.  SP_DIST_BACKEND=gloo exists only to exercise the N>1 code path on a 1-GPU test box
        # (RCCL refuses two ranks on one device).
        backend = os.environ.get("SP_DIST_BACKEND", "nccl")
        if backend == "nccl":
            torch.distributed.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            torch.distributed.init_process_group(backend)
    dev = torch.device("cuda", dev_index)
    torch.cuda.set_device(dev)
    if args.force_bucketer:
        if world != 1 or args.mode != "train":
            raise SystemExit("bench.py: --force-bucketer is the single-GPU training measurement (RCCL world of one)")
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        torch.distributed.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", world_size=1, rank=0,
                                             device_id=torch.device("cuda", dev_index))

    from scanpaths_amd import hip
    if not os.path.exists(hip.LIB_PATH):        # the in-tree .so normally travels with the snapshot; build it if it did not
        if rank == 0:                           # (rank 0 builds, the others wait at the barrier)
            subprocess.run(["make", "-C", os.path.join(ROOT, "scanpaths_amd", "csrc"), "-j8"], check=True, stdout=subprocess.DEVNULL)
        if world > 1:
            torch.distributed.barrier()
    from scanpaths_amd import functional as F
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.synth import make_batch

    task = {"air": "AiR", "osie": "OSIE", "coco": "COCO_Search18"}[args.task]
    Hm, Wm = args.height // 8, args.width // 8
    model = build_model(args, dev)
    b = {k: v.to(dev) for k, v in make_batch(task, args.batch, args.height, args.width, args.T, seed=0, rank=rank).items()}

    if args.dense_backward:
        sp_config.set(row_sparsity=False)            # tagged in config.non_default_switches
    sparse_on = sp_config.settings["row_sparsity"]
    if args.mode == "train" and args.dropin:
        if world != 1:
            raise SystemExit("bench.py: --dropin times the reference's single-process call sequence (its DataParallel has no counterpart "
                             "here; the multi-GPU path is FlatAdam's bucketed all-reduce)")
        model.train()
        step = dropin_step(args, model, b)
    elif args.mode == "train":
        model.train()
        if world > 1:
            from scanpaths_amd.ddp import broadcast_module_state_
            broadcast_module_state_(model)      # BatchNorm buffers; FlatAdam broadcasts the flat parameter buffer itself
        opt = FlatAdam(model.parameters(), lr=1e-4, weight_decay=5e-5 if args.task == "air" else 5e-4, clip=12.5,
                       conditional_params=model.has_conditional_params,
                       reference_zero_grad=model.has_conditional_params,     # COCO heads: the reference's torch-1.6 zero-fill semantics
                       force_bucketer=args.force_bucketer)

        cur = {"b": b}          # (the length-law leg after the timed region swaps the batch)

        def step():
            b = cur["b"]
            opt.zero_grad()
            pred = call_model(model, args, b, True)
            mask_sums = None
            if world > 1:   # loss normalised by the GLOBAL mask sums, as DataParallel's gathered loss (AiR/train.py:190-197)
                from scanpaths_amd.ddp import global_mask_normaliser
                mask_sums = global_mask_normaliser(torch.cat([F.device_sum(b["action_masks"]),
                                                              F.device_sum(b["duration_masks"])]))
            loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0, mask_sums)
            loss.backward()
            opt.step()
            return loss
    else:
        # config 5: eval-mode forward + the reference's test loop sampling (AiR/test.py:140-193: eval_repeat_num = 10 sampled
        # scanpaths per head), all on device; one D2H at the end of the timed region is NOT taken (results stay in HBM)
        from scanpaths_amd.models.sampling import Sampling
        model.eval()
        sampler = Sampling(convLSTM_length=args.T, min_length=1, map_width=Wm, map_height=Hm, width=args.width, height=args.height)
        heads = ("good", "poor") if args.task == "air" else ("",)

        def step():
            with torch.no_grad():
                pred = call_model(model, args, b, False)
                last = None
                for hd in heads:
                    pre = hd + "_" if hd else ""
                    for _ in range(10):
                        last = sampler.random_sample(pred[pre + "all_actions_prob"], pred[pre + "log_normal_mu"],
                                                     pred[pre + "log_normal_sigma2"])
            return last["durations"].sum()

    dist_info = None
    if world > 1:
        # self-verification of the multi-GPU line (the first SCALE run must prove the collective saw N ranks): an all-reduce of ones,
        # the devices the ranks sit on, the RCCL version
        one = torch.ones(1, dtype=torch.float32, device=dev)
        torch.distributed.all_reduce(one)
        props = torch.cuda.get_device_properties(dev)
        mine = {"rank": rank, "local_rank": local_rank, "device_index": dev_index, "name": props.name,
                "uuid": str(getattr(props, "uuid", "")), "pci_bus_id": getattr(props, "pci_bus_id", None)}
        gathered = [None] * world
        torch.distributed.all_gather_object(gathered, mine)
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:
            ver = f"unavailable ({type(e).__name__})"
        dist_info = {"backend": torch.distributed.get_backend(), "world_size": torch.distributed.get_world_size(),
                     "allreduce_of_ones": float(one.item()), "allreduce_self_test_ok": float(one.item()) == float(world),
                     "rccl_version": ver, "ranks": gathered,
                     "distinct_devices": len({(g["device_index"], g["uuid"], g["pci_bus_id"]) for g in gathered})}
        if not dist_info["allreduce_self_test_ok"]:
            raise SystemExit(f"bench.py: all-reduce self-test failed: sum of ones over {world} ranks = {one.item()}")

    for _ in range(args.warmup):
        step()

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    hip.TIMER = hip.KernelTimer(min_flops=2e11 * args.batch / 32)    # bracket only the dominant GEMM launches with HIP events
    sync()
    t0 = time.perf_counter()
    host_s = 0.0
    for _ in range(args.steps):
        th = time.perf_counter()
        loss = step()
        host_s += time.perf_counter() - th          # host time to ENQUEUE the step (no synchronisation inside): how far the host runs ahead
    sync()
    dt = time.perf_counter() - t0
    timer, hip.TIMER = hip.TIMER, None
    # host time to enqueue ONE step onto an idle device (outside the timed region): inside the timed loop the host is held back by the
    # device's queue once it is far enough ahead, so host_s / steps tends to the device time of a step however fast the host is
    host_idle = []
    for _ in range(2):
        th = time.perf_counter()
        step()
        host_idle.append(time.perf_counter() - th)
        sync()
    if world > 1 and args.mode == "train":
        # after W + K optimiser steps every rank must hold the same parameters (same reduced gradients, same Adam state): a drifted
        # replica means a collective was skipped or mis-ordered on some rank
        from scanpaths_amd.ddp import assert_replicas_identical
        assert_replicas_identical(opt.flat_p)
        dist_info["replicas_identical_after_timed_region"] = True
    sparsity, dropin = None, None
    length_leg = None
    act_frac = None
    if args.mode == "train":
        # what the gate derived from the gradient of the last step: last[b] per sample -> the live fraction of (sample, step) pairs of steps 1..T-1
        rows_tok = getattr(model, "last_decode_rows", None)
        if sparse_on and rows_tok is not None and rows_tok.rc is not None:
            last = rows_tok.rc.last
            Tn = args.T
            act_frac = float((last.view(-1, 1) >= torch.arange(1, Tn, device=last.device).view(1, -1)).float().mean()) if Tn > 1 else 1.0
    if args.mode == "train" and world == 1 and sparse_on and not args.no_dense_leg:
        # the same step with the dense backward pass (what the reference computes), same process, after the timed region
        sp_config.set(row_sparsity=False)
        step()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        dense_ms = (time.perf_counter() - t1) / args.steps * 1e3
        sp_config.set(row_sparsity=True)
        sparsity = {"derived_from_output_gradient": True, "dense_backward_ms_per_step": round(dense_ms, 2),
                    "dense_backward_images_per_s": round(args.batch * 1e3 / dense_ms, 3),
                    "active_fraction_of_sample_steps": round(act_frac, 4) if act_frac is not None else None,
                    "note": "the loss multiplies by action_masks / duration_masks (AiR/models/loss.py:10-14,27-32): behind a sample's last "
                            "masked-in step every gradient of the decoder's recurrence is EXACTLY zero.  An identity node behind decode()'s "
                            "outputs reads last[b] off the gradient that arrives (any consumer set; no switch, no caller promise) and the cell "
                            "backward, the h-gate conv's data gradient, its deferred weight gradient and the fan-ins skip those (sample, step) "
                            "pairs instead of multiplying zeros -- gradients bit-identical to the dense backward "
                            "(tests/test_model_gpu.py::test_masked_step_sparsity...); forward, loss, clip and Adam are unchanged; synthetic "
                            "scanpath lengths are uniform in 1..T (SURVEY.md 8d), real length distributions skip less or more; "
                            "`--dense-backward` makes the dense form the line"}
    if args.mode == "train" and world == 1 and sparse_on and not args.dropin and not args.no_length_leg and not args.force_bucketer:
        # sensitivity of the line to the synthetic length law (the sparsity of the backward pass is the only thing that depends on it):
        # the same step on L ~ U{ceil(T/2)..T} -- every other tensor of the batch is identical -- same process, after the timed region
        b2 = {k: v.to(dev) for k, v in make_batch(task, args.batch, args.height, args.width, args.T, seed=0, rank=rank,
                                                  length_law="uniform_halfT_T").items()}
        cur["b"] = b2
        step()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        l_ms = (time.perf_counter() - t1) / args.steps * 1e3
        rows_tok = getattr(model, "last_decode_rows", None)
        l_frac = None
        if rows_tok is not None and rows_tok.rc is not None and args.T > 1:
            l_frac = float((rows_tok.rc.last.view(-1, 1) >= torch.arange(1, args.T, device=dev).view(1, -1)).float().mean())
        cur["b"] = b
        length_leg = {"length_law": "L ~ U{ceil(T/2)..T} (every other draw of the batch unchanged)", "ms_per_step": round(l_ms, 2),
                      "images_per_s": round(args.batch * 1e3 / l_ms, 3),
                      "active_fraction_of_sample_steps": round(l_frac, 4) if l_frac is not None else None}
    if args.mode == "train" and world == 1 and not args.dropin and not args.no_dropin_leg and not args.force_bucketer:
        # the reference's literal call sequence on a second, identically initialised model (the first one's parameters live in FlatAdam's
        # flat buffers): same process, same box, after the timed region
        model2 = build_model(args, dev).train()
        dstep = dropin_step(args, model2, b)
        for _ in range(max(args.warmup, 1)):
            dstep()
        sync()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            dloss = dstep()
        sync()
        d_ms = (time.perf_counter() - t1) / args.steps * 1e3
        dropin = {"dropin_ms_per_step": round(d_ms, 2), "dropin_images_per_s": round(args.batch * 1e3 / d_ms, 3),
                  "vs_headline_step": round(d_ms / (dt / args.steps * 1e3), 4), "loss": round(float(dloss.detach()), 5),
                  "sequence": "AiR/train.py:188-205 call for call on the HIP model: optimizer.zero_grad(); model(...); CrossEntropyLoss + "
                              "MLPLogNormalDistribution (two calls) ; loss.backward(); torch.nn.utils.clip_grad_norm_; torch.optim.Adam.step; "
                              "LambdaLR.step (tensorboard scalars left out); the masked-step sparsity applies by itself"}
        del model2, dstep
    ddp_info = None
    if args.force_bucketer:
        ddp_info = bucketer_overhead(opt, step, sync, args.steps, dt / args.steps * 1e3)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    value = args.batch * world * args.steps / dt
    summ = timer.summary()
    # launches that skip samples behind their last masked-in step ("_rows" keys) EXECUTE only the active fraction of the dense FLOPs
    # their key was priced with: the roofline below counts the executed ones
    if args.mode == "train" and act_frac is not None:
        for k_, d_ in summ.items():
            if k_[0].split("+")[0].endswith("_rows"):
                d_["flops_per_launch"] *= act_frac
                d_["tflops"] *= act_frac
                d_["executed_fraction"] = round(act_frac, 4)
    # ---- roofline of the dominant kernel = the timed GEMM kind + shape with the largest total time -------------
    roofline = None
    if summ:
        dom_key, dom = max(summ.items(), key=lambda kv: kv[1]["ms"])
        kind = dom_key[0].split("+")[0]          # ("+side": the launch ran on the side stream beside the backward chain, hip.KernelTimer)
        fam = kind.split("_")[0]
        nprod = PRODUCTS.get(fam)
        peak = PEAK_F16_MFMA_TFLOPS if nprod else PEAK_FP32_MFMA_TFLOPS
        traffic, traffic_src, mfma_busy = None, None, None
        # the committed PMC passes were taken on the h-gate conv shape (tools/bench_hconv_steps.py: M = 81920 pixels per application,
        # N = 2048, K = 4608; the deferred weight gradient hw2_kernel covers all T - 1 applications in one launch)
        hgate = (dom_key[2], dom_key[3]) == (2048, 4608) and dom_key[1] in (81920, 81920 * (args.T - 1))
        if args.batch == 32 and (args.height, args.width) == (320, 512) and kind in PMC_PREFIX and hgate:
            for fn in ("r06_pmc_hconv.json", "r05_pmc_hconv.json", "r04_pmc_hconv.json", "r03_pmc_hconv.json", "r02_pmc_hconv.json", "r01_pmc_hconv.json"):      # newest committed PMC passes first
                try:
                    pmc = json.load(open(os.path.join(ROOT, "profiles", fn)))["kernels"]
                    keys = [k for k in pmc if k.startswith(PMC_PREFIX[kind])]
                    fused = [k for k in keys if PMC_FUSED_FWD in k]
                    key = fused[0] if (kind == "h2_fwd" and fused) else keys[0]
                    traffic, traffic_src = round(pmc[key].get("hbm_side_bytes_per_launch_calibrated", pmc[key]["hbm_side_bytes_per_launch"])), fn
                    mfma_busy = pmc[key].get("mfma_busy_pct")
                    break
                except Exception:
                    continue
        M, N, K = dom_key[1], dom_key[2], dom_key[3]
        total_timed_ms = sum(d["ms"] for d in summ.values()) / args.steps
        by_kind = []
        for k, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"])[:6]:
            kb = k[0].split("+")[0]
            f = kb.split("_")[0]
            by_kind.append({"kernel": KERNEL_NAMES.get(kb, kb), "M": k[1], "N": k[2], "K": k[3],
                            # timed on the side stream BESIDE the current stream's ~45 small launches of the backward chain: includes that
                            # contention, not comparable with a stand-alone launch of the same kernel (round 4's figures were stand-alone)
                            **({"beside_backward_chain": True} if "+side" in k[0] else {}),
                            **({"executed_fraction_of_dense_flops": d["executed_fraction"]} if "executed_fraction" in d else {}),
                            "launches_per_step": d["launches"] / args.steps, "avg_launch_ms": round(d["avg_ms"], 4),
                            "ms_per_step": round(d["ms"] / args.steps, 2), "tflops": round(d["tflops"], 1),
                            "frac_of_2500": round(d["tflops"] / PEAK_F16_MFMA_TFLOPS, 4) if PRODUCTS.get(f) else None})
        roofline = {
            "bound": "mfma", "achieved": round(dom["tflops"], 2), "peak": peak, "unit": "TFLOP/s",
            "frac": round(dom["tflops"] / peak, 4),
            # what the counters say about "bound": the matrix pipe is busy for this share of the launch (same PMC passes as `traffic`); the rest is the
            # K loop waiting for its LDS-DMA ring -- two 48-KB K-tiles in flight per CU over ~2 us of fabric-side latency = ~45 GB/s per CU, the
            # rate every big GEMM kernel of this path sits at -- and, for the fused-cell forward, its epilogue (~0.5 ms of 3.4-3.6)
            "mfma_busy_pct_measured": round(mfma_busy, 1) if mfma_busy is not None else None,
            "traffic": traffic,
            "traffic_note": (f"bytes/launch = (2*FETCH_SIZE + WRITE_SIZE)*1024 from separate rocprofv3 --pmc passes (profiles/{traffic_src}); "
                             "includes Infinity-Cache hits; algorithmic bytes/launch = split operands once + fp32 output = 0.88 GB per application "
                             "(weight gradient: dY planes 671 MB + X planes 168 MB + dW 38 MB; the deferred hw2_kernel launch covers T - 1 "
                             "applications: x (T - 1)) / 1.05 GB (forward; the ConvLSTM-fused forward also reads the x-gates and c_prev and writes gates, "
                             "c, h and h's split operand: 2.2 GB = reads 1.05 + writes 1.15; measured fabric-side reads 3.04 GB -- the K loop's operand "
                             "re-reads beyond L2, 2.2 GB, plus the epilogue's 0.84 GB, whose 64-byte-run loads FETCH_SIZE tallies at ~0.85 of "
                             "their bytes instead of 1/2: calibrated with and without them, profiles/r05_fetch_calibration.log -- writes 1.20 GB)") if traffic else "no committed PMC pass for this kernel/shape",
            "kernel": f"{KERNEL_NAMES.get(kind, kind)}: implicit GEMM M={M} N={N} K={K} ({dom_key[4]} taps) -- the timed GEMM "
                      "kind+shape with the largest total time",
            "achieved_note": "ALGORITHMIC FLOPs (2*M*N*K of the fp32 GEMM the reference computes) / HIP-event launch time on the "
                             "launch stream; peak = dense MFMA peak of the dtype the kernel ISSUES (fp16: 2500 TFLOP/s)",
            "flops_per_launch": dom["flops_per_launch"], "avg_launch_ms": round(dom["avg_ms"], 4),
            "launches_timed": dom["launches"],
            "mfma_products_per_fma": nprod,
            "mfma_pipe_frac": round(dom["tflops"] * nprod / peak, 4) if nprod else None,
            "mfma_pipe_note": "issued MFMA FLOPs / peak = frac x products per algorithmic FMA (matrix-pipe occupancy of the scheme)",
            "practical_ceiling_note": "profiles/r03_wave_tile_probe.log, profiles/r04_hw2_probe.log: idealised loops of the kernels' structures (same "
                                      "MFMAs, fragment reads, LDS-DMA pieces; no epilogue, no address arithmetic) on random fp16 operands (DVFS under "
                                      "switching power): 256x128 tile / 64x64 wave tiles 3.3-3.7 ms per application of the h-gate shape, 256x256 tile / "
                                      "64x128 wave tiles (hw2_kernel, deferred launch) 3.0-3.2 ms; the shipped kernels run at 2.8 (hw2), 2.9-3.0 (data "
                                      "gradient) and 3.6-3.8 ms (fused forward incl. the cell epilogue)",
            "timed_gemms": by_kind,
            "all_big_gemms_ms_per_step": round(total_timed_ms, 2),
            "all_big_gemms_tflops": round(sum(d["flops_per_launch"] * d["launches"] for d in summ.values())
                                          / max(sum(d["ms"] for d in summ.values()), 1e-9) / 1e9, 2)}
        # end to end: reduced (necessary) FLOPs per image x images/s, SURVEY.md §8(d) line "roofline.achieved"
        gmac = reduced_fwd_gmac(args.arch, args.height, args.width, args.T, args.task)
        tflop_per_img = gmac * 2e9 * (3.0 if args.mode == "train" else 1.0) / 1e12
        # FLOPs the step EXECUTES: the masked-step sparsity removes the data and weight gradient of the h-gate conv (2 x 24.16 GMAC at
        # 320x512 per application) for the dead (sample, step) pairs of steps 1..T-1
        exec_per_img = tflop_per_img
        if args.mode == "train" and act_frac is not None:
            exec_per_img -= 2 * 24.16 * (args.height * args.width / (320.0 * 512.0)) * 2e9 * (args.T - 1) * (1.0 - act_frac) / 1e12
        roofline["end_to_end"] = {"reduced_tflop_per_image": round(tflop_per_img, 3),
                                  "executed_tflop_per_image": round(exec_per_img, 3),
                                  "achieved_tflops_executed": round(exec_per_img * value / world, 1),
                                  "frac_of_2500_executed": round(exec_per_img * value / world / PEAK_F16_MFMA_TFLOPS, 4),
                                  "achieved_tflops_dense_equivalent": round(tflop_per_img * value / world, 1),
                                  "frac_of_2500_dense_equivalent": round(tflop_per_img * value / world / PEAK_F16_MFMA_TFLOPS, 4),
                                  "note": "reduced-column FLOPs/img (BASELINE.md §3; train = 3x forward) x measured img/s per GPU; 'executed' "
                                          "subtracts the h-gate data / weight gradients of the (sample, step) pairs the sparse backward skips, "
                                          "'dense_equivalent' prices the step as if it had multiplied those zeros (what the reference does)"}

    headline = (args.task == "air" and args.mode == "train" and args.precision == "f32" and args.batch == 32 and args.T == 16
                and (args.height, args.width) == (320, 512) and args.arch == "resnet50")
    if headline:
        metric = "images/sec/GPU (AiR train step, bs=32, 320x512) at 1/2/4/8 MI355X"
    else:
        metric = (f"images/sec/GPU ({task} {'train step' if args.mode == 'train' else 'inference: eval forward + 10 sampled scanpaths per head'}"
                  f", bs={args.batch}, {args.height}x{args.width}{', THROUGHPUT MODE f16x1' if args.precision == 'f16x1' else ''})")
    arith = ("fp32 in / fp32 out / fp32 accumulation; GEMM operands as exact-scaled 2xfp16 splits with 3 MFMA products (error vs fp64 "
             "below a CPU fp32 GEMM, tools/gemm_error.py); no reduced-precision storage") if args.precision == "f32" else \
            ("THROUGHPUT MODE: large GEMM operands rounded to ONE fp16 plane (per-tensor power-of-two scale), 1 MFMA product, fp32 "
             "accumulation, fp32 storage everywhere else; does NOT meet the 1e-4 parity bar (see DESIGN.md §6)")
    out = {"metric": metric, "value": round(value, 3),
           "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32" if args.precision == "f32" else "f16-in/f32-acc", "data": "synthetic",
           "value_per_gpu": round(value / world, 3),
           "config": {"workload": f"{task} {'supervised train step (fwd+loss+bwd+clip+Adam)' if args.mode == 'train' else 'eval forward + sampling'}"
                                  f", {args.arch}, T={args.T}, {args.height}x{args.width}, per-GPU batch {args.batch}",
                      "global_batch": args.batch * world, "parallelism": f"dp{world}",
                      "loss": round(float(loss.detach()), 5) if args.mode == "train" else None,
                      "peak_hbm_gib": round(torch.cuda.max_memory_allocated(dev) / 2 ** 30, 1),
                      "host_enqueue_ms_per_step": round(host_s / args.steps * 1e3, 2),
                      "host_enqueue_ms_one_step_on_idle_device": round(min(host_idle) * 1e3, 2),
                      "arithmetic": arith,
                      # switches that differ from the package defaults (scanpaths_amd.config; environment variables are honoured
                      # only under SP_ALLOW_ENV_TUNING=1): {} for the headline line
                      "non_default_switches": sp_config.non_default()},
           "roofline": roofline}
    if sparsity is not None:
        # the same line's dense figure at top level: what the step costs when the zeros behind every sample's last masked-in step are
        # multiplied like the reference does; `value` is the step on SURVEY 8(d)'s length law L ~ U{1..T} with those zeros skipped
        out["value_dense"] = sparsity["dense_backward_images_per_s"]
        out["ms_per_step_dense"] = sparsity["dense_backward_ms_per_step"]
        out["value_note"] = ("value: masked-step sparsity of the backward pass on (exact, bit-identical gradients; derived from the gradient, "
                             "no switch), scanpath lengths L ~ U{1..T}; value_dense: same process, dense backward; length_law_sensitivity: "
                             "same process, L ~ U{ceil(T/2)..T}; cpu_baseline: the oracle computes the dense backward")
    if length_leg is not None:
        out["length_law_sensitivity"] = length_leg
    if dist_info is not None:
        out["distributed"] = dist_info
    if sparsity is not None:
        out["backward_sparsity"] = sparsity
    elif args.mode == "train":
        out["backward_sparsity"] = {"derived_from_output_gradient": bool(sparse_on),
                                    "active_fraction_of_sample_steps": round(act_frac, 4) if act_frac is not None else None}
    if dropin is not None:
        out["dropin"] = dropin
    if args.mode == "train" and args.dropin:
        out["metric"] = metric + " [--dropin: the reference's literal call sequence AiR/train.py:188-205]"
    if ddp_info is not None:
        out["ddp"] = ddp_info
        out["metric"] = metric + " [--force-bucketer: RCCL world of one, NOT the headline line]"
    if world == 1 and not args.no_cpu_baseline and args.mode == "train":
        out["cpu_baseline"] = cpu_baseline(args)
    if world > 1 or args.force_bucketer:
        torch.distributed.destroy_process_group()
    # RCCL writes a version banner to the C stdout buffer at initialisation; flushed at process exit it would land BEHIND the JSON line of
    # a redirected run.  Flush it now: the JSON line is the last thing this process prints.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
