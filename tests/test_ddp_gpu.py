"""Data-parallel step on the real HIP path: two processes (sharing the one GPU of the test box, gloo transport -- RCCL
refuses two ranks per device) each run one train step on the SAME shard; after the flat-gradient all-reduce (sum, averaged
inside the fused Adam kernel) and the global mask-sum normaliser, every rank must hold exactly the parameters a single
process gets from that shard."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _one_step(world, rank, port, q):
    torch.set_num_threads(8)
    import torch.distributed as dist
    from scanpaths_amd import functional as F
    from scanpaths_amd.ddp import global_mask_normaliser
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.models.scanpath_model import ScanpathModel
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    m = ScanpathModel("OSIE", convLSTM_length=2, arch="resnet18")
    fill_module(m, seed=8)
    m = m.to(dev).train()
    opt = FlatAdam(m.parameters(), lr=1e-3, weight_decay=5e-4, clip=12.5, bucket_mb=4)     # several buckets, overlapped
    b = {k: v.to(dev) for k, v in make_batch("OSIE", 2, 240, 320, 2, seed=8).items()}     # same shard on every rank
    opt.zero_grad()
    pred = m(b["images"])
    sums = torch.cat([F.device_sum(b["action_masks"]), F.device_sum(b["duration_masks"])])
    loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0,
                                 global_mask_normaliser(sums))
    loss.backward()
    tn = opt.step()
    torch.cuda.synchronize()
    q.put((rank, float(loss), float(tn), opt.flat_p.detach().cpu().numpy()))   # by value: a shared-memory tensor
    # handle would die with this process if the parent has not unpickled it yet
    if world > 1:
        dist.destroy_process_group()


# The worker processes of this file cost 20-50 s each in start-up (interpreter, torch, device context) and seconds of GPU work: the
# processes of one test start side by side, with eight host threads each.
def _job(request, name, fn):
    return fn()


def _run_two_rank():
    ctx = mp.get_context("spawn")
    q1, q = ctx.Queue(), ctx.Queue()
    p = ctx.Process(target=_one_step, args=(1, 0, 0, q1))       # (the three processes run side by side)
    p.start()
    port = _free_port()
    procs = [ctx.Process(target=_one_step, args=(2, r, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    single = q1.get(timeout=900)
    p.join(60)
    res = [q.get(timeout=900) for _ in range(2)]
    for pr in procs:
        pr.join(60)
    return single, res, [pr.exitcode for pr in procs]


def test_two_rank_step_equals_single_process(request):
    (_, loss1, tn1, flat1), res, codes = _job(request, "two_rank", _run_two_rank)
    assert codes == [0, 0], codes
    for rank, loss2, tn2, flat2 in res:
        assert abs(loss2 - loss1) <= 1e-6 * abs(loss1), (rank, loss1, loss2)
        assert abs(tn2 - tn1) <= 1e-5 * tn1
        assert float(abs(flat2 - flat1).max()) <= 1e-6, rank      # identical up to the all-reduce's (a+a)/2 rounding


def _coco_step(world, rank, port, q):
    """COCO_Search18: every rank sees DIFFERENT target categories, so different per-category heads receive gradients"""
    torch.set_num_threads(8)
    import torch.distributed as dist
    from scanpaths_amd import functional as F
    from scanpaths_amd.ddp import global_mask_normaliser
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.models.scanpath_model import ScanpathModel
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda:0")
    m = ScanpathModel("COCO_Search18", convLSTM_length=2, arch="resnet18")
    fill_module(m, seed=5 + 100 * rank)        # replicas start DIFFERENT: FlatAdam must broadcast rank 0's parameters
    m = m.to(dev).train()
    from scanpaths_amd.ddp import assert_replicas_identical, broadcast_module_state_
    broadcast_module_state_(m)                 # BatchNorm buffers too (what nn.DataParallel's per-forward replication does)
    opt = FlatAdam(m.parameters(), lr=1e-3, weight_decay=5e-4, clip=12.5, conditional_params=m.has_conditional_params,
                   bucket_mb=8)
    b = {k: v.to(dev) for k, v in make_batch("COCO_Search18", 2, 240, 320, 2, seed=5, rank=rank).items()}
    b["tasks"] = torch.tensor([1, 7] if rank == 0 else [7, 12], device=dev)
    assert_replicas_identical(opt.flat_p)
    opt.zero_grad()
    pred = m(b["images"], b["attention_maps"], b["tasks"])
    sums = torch.cat([F.device_sum(b["action_masks"]), F.device_sum(b["duration_masks"])])
    loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0,
                                 global_mask_normaliser(sums))
    loss.backward()
    opt.step()
    torch.cuda.synchronize()
    names = [n for n, _ in m.named_parameters()]
    stepped = [n for n, p in zip(names, m.parameters()) if float(opt.state[p]["step"]) > 0]
    q.put((rank, opt.flat_p.detach().cpu().numpy(), stepped))
    dist.destroy_process_group()


def _run_coco():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_coco_step, args=(2, r, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=900) for _ in range(2)], key=lambda r: r[0])
    for pr in procs:
        pr.join(60)
    return res, [pr.exitcode for pr in procs]


def test_ranks_with_different_coco_categories_stay_identical(request):
    """a per-category head is stepped when ANY rank produced a gradient for it (ddp.union_flags), as under the reference's single
    optimizer behind DataParallel: after one step both replicas hold bit-identical parameters and stepped the same heads"""
    res, codes = _job(request, "coco", _run_coco)
    assert codes == [0, 0], codes
    assert (res[0][1] == res[1][1]).all()
    assert res[0][2] == res[1][2]
    heads = sorted({n.split(".")[1] for n in res[0][2] if n.startswith("object_sal_layer.")})
    assert len(heads) == 3, heads            # categories 1, 7, 12 and no other


def _rccl_world1(use_dist, port, q):
    """two train steps; use_dist: torch.distributed over RCCL ("nccl") with a world of ONE and the GradBucketer forced on, so the
    post-accumulate-grad hook -> bucket -> async ncclAllReduce (RCCL's own stream) -> wait -> sp_sumsq / sp_clip_adam (ctypes
    launches on torch's current stream) chain runs exactly as it does on N GPUs, with an identity reduction"""
    torch.set_num_threads(8)
    import torch.distributed as dist
    from scanpaths_amd.models.loss import supervised_loss
    from scanpaths_amd.models.scanpath_model import ScanpathModel
    from scanpaths_amd.optim import FlatAdam
    from scanpaths_amd.procedural import fill_module
    from scanpaths_amd.synth import make_batch
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    if use_dist:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    m = ScanpathModel("AiR", convLSTM_length=3, arch="resnet50")
    fill_module(m, seed=8, family="tame")
    m = m.to(dev).train()
    opt = FlatAdam(m.parameters(), lr=1e-3, weight_decay=5e-5, clip=12.5, bucket_mb=16, force_bucketer=use_dist)
    nb = len(opt._bucketer.ranges) if opt._bucketer is not None else 0
    out = []
    for it in range(2):
        b = {k: v.to(dev) for k, v in make_batch("AiR", 2, 240, 320, 3, seed=8 + it).items()}
        opt.zero_grad()
        pred = m(b["images"], b["attention_maps"], b["performances"])
        loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0)
        loss.backward()
        tn = opt.step()
        out.append((float(loss), float(tn)))
    torch.cuda.synchronize()
    q.put((use_dist, nb, out, opt.flat_p.detach().cpu().numpy()))
    if use_dist:
        dist.destroy_process_group()


def _run_rccl():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    res = {}
    procs = [ctx.Process(target=_rccl_world1, args=(use_dist, _free_port(), q)) for use_dist in (False, True)]
    for p in procs:                                            # (side by side: two independent processes on the one device)
        p.start()
    for _ in procs:
        r = q.get(timeout=900)
        res[r[0]] = r
    for p in procs:
        p.join(120)
    return res, [p.exitcode for p in procs]


def test_rccl_world_of_one_bucketed_step_is_bit_identical_to_the_plain_step(request):
    """the `nccl` (= RCCL) branch on a HIP device.  A world of one makes every all-reduce the identity, so two
    training steps through hooks + buckets + RCCL streams must reproduce the non-distributed steps BIT for bit; a missing stream
    dependency between RCCL's stream and the ctypes-launched kernels (clip+Adam reading a bucket still being reduced, backward
    writing a bucket already launched) would show up as a difference.  Surface: nn.DataParallel, AiR/train.py:169-170, 190-202."""
    res, codes = _job(request, "rccl", _run_rccl)
    assert codes == [0, 0], codes
    assert res[True][1] >= 4, res[True][1]                 # 81 M parameters in 16 MB buckets: the bucketer really was active
    assert res[True][2] == res[False][2], (res[True][2], res[False][2])
    assert (res[True][3] == res[False][3]).all()


def _start_bench(extra, env_extra=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.update(env_extra or {})
    return subprocess.Popen([sys.executable, os.path.join(root, "bench.py")] + extra, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env)


_SMALL = ["--task", "osie", "--arch", "resnet18", "--T", "2", "--batch", "2", "--height", "240", "--width", "320", "--steps", "2",
          "--warmup", "1", "--no-cpu-baseline"]


def _run_bench_gpus():
    """the three bench.py invocations of the test below, side by side -> [(returncode, stdout, stderr)] (None for the one not run)"""
    ps = [_start_bench(["--gpus", "2"] + _SMALL) if torch.cuda.device_count() < 2 else None, _start_bench(["--gpus", "1"] + _SMALL),
          _start_bench(["--gpus", "2"] + _SMALL, {"SP_DIST_BACKEND": "gloo"})]
    out = []
    for p in ps:
        if p is None:
            out.append(None)
            continue
        so, se = p.communicate(timeout=1500)
        out.append((p.returncode, so, se))
    return out


def test_bench_gpus_n_launches_n_ranks_or_fails_loudly(request):
    """`python bench.py --gpus N` without a launcher around it (the form the driver uses): (i) with fewer devices than ranks it must
    FAIL, never print an `n_gpus: 1` line; (ii) with the gloo transport (two ranks sharing this box's GPU; RCCL refuses that) the
    parent starts two ranks through torch.distributed.run and relays rank 0's line with n_gpus = 2 and the aggregate rate."""
    import json
    rf, r1, r2 = _job(request, "bench_gpus", _run_bench_gpus)
    if rf is not None:
        assert rf[0] != 0 and '"n_gpus"' not in rf[1], (rf[0], rf[1][-500:])
        assert "only 1 HIP device" in rf[2], rf[2][-500:]
    assert r1[0] == 0, r1[2][-2000:]
    one = json.loads(r1[1].strip().splitlines()[-1])
    assert r2[0] == 0, r2[2][-2000:]
    two = json.loads([l for l in r2[1].strip().splitlines() if l.startswith("{")][-1])
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["config"]["global_batch"] == 4 and two["scaling"] == "weak"
    assert abs(two["value"] - 2 * two["value_per_gpu"]) < 1e-2 * two["value"]

