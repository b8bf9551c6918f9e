"""N>1 path on CPU: world_size-2 gloo processes exercise scanpaths_amd/ddp.py (the exchanges bench.py / FlatAdam use)
and check the data-parallel semantics against the single-process loss of the reference (oracle formulas)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import scanpath_oracle as O
    from scanpaths_amd.ddp import allreduce_sum_, global_mask_normaliser, shard_batch
    from scanpaths_amd.synth import make_batch
    B, T, A = 6, 5, 1201
    full = make_batch("AiR", B, 240, 320, T, seed=21)
    g = torch.Generator().manual_seed(4)
    z = torch.randn(B, T, A, generator=g, dtype=torch.float64)
    mu = torch.randn(B, T, generator=g, dtype=torch.float64)
    s2 = torch.rand(B, T, generator=g, dtype=torch.float64) + 0.3
    w = torch.randn(A, dtype=torch.float64, generator=g).requires_grad_(True)      # a shared "parameter"
    fd = {k: (v.double() if v.is_floating_point() else v) for k, v in full.items()}
    fd.update(z=z, mu=mu, s2=s2)
    # single-process reference loss over the whole batch (what DataParallel computes on the gathered outputs)
    la = O.cross_entropy_loss(z * w, fd["scanpaths"], fd["action_masks"])
    ld = O.lognormal_nll(mu, s2, fd["durations"], fd["duration_masks"])
    (gref,) = torch.autograd.grad(la + ld, w)
    # this rank's shard, normalised by the all-reduced mask sums
    sh = shard_batch(fd, rank, world)
    local = torch.stack([sh["action_masks"].sum(), sh["duration_masks"].sum()])
    norm = global_mask_normaliser(local)
    p = torch.softmax(sh["z"] * w, -1)
    la_r = -(sh["scanpaths"] * torch.log(p + O.EPS) * sh["action_masks"].unsqueeze(-1)).sum() / norm[0]
    (gr,) = torch.autograd.grad(la_r, w)
    flat = gr.clone()
    n = allreduce_sum_(flat)
    flat /= n
    loss_sum = la_r.detach().clone()
    dist.all_reduce(loss_sum)
    from scanpaths_amd.ddp import union_flags
    mine = [rank == 0, rank == 1, False, True]          # which "parameters" this rank touched
    assert union_flags(mine, torch.device("cpu")) == [True, True, False, True]
    q.put((rank, float((flat - gref).abs().max()), float(loss_sum / world - la), float(norm[0] * world - fd["action_masks"].sum())))
    dist.destroy_process_group()


def test_two_rank_gradient_and_loss_equal_single_process():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, gerr, lerr, serr in res:
        assert gerr < 1e-12 and abs(lerr) < 1e-12 and abs(serr) < 1e-9, (rank, gerr, lerr, serr)


def test_shard_batch_covers_everything_once():
    from scanpaths_amd.ddp import shard_batch
    b = {"x": torch.arange(10).view(10, 1), "y": torch.arange(10)}
    parts = [shard_batch(b, r, 4) for r in range(4)]
    assert torch.equal(torch.cat([p["y"] for p in parts]), b["y"])


def _bucket_worker(rank, world, port, q):
    """GradBucketer: buckets complete in a rank-dependent order (and one parameter never reports on rank 1), yet every rank
    issues the same descending sequence of collectives and ends with the summed buffer"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from scanpaths_amd.ddp import GradBucketer, assert_replicas_identical, broadcast_module_state_
    sizes = [40, 8, 120, 64, 16, 200, 4]
    offs, tot = [], 0
    for s in sizes:
        offs.append(tot)
        tot += s
    flat = torch.arange(tot, dtype=torch.float64) * (rank + 1)
    bk = GradBucketer(flat, offs, tot, bucket_bytes=100 * 8)
    assert len(bk.ranges) >= 3 and bk.ranges[0][0] == 0 and bk.ranges[-1][1] == tot
    order = list(range(len(sizes)))[::-1] if rank == 0 else [3, 6, 0, 5, 1, 2, 4]
    for i in order:
        if rank == 1 and i == 2:
            continue                     # a parameter without a gradient on this rank (COCO head of an absent category)
        bk.mark_ready(i)
    bk.finish()
    expect = torch.arange(tot, dtype=torch.float64) * sum(r + 1 for r in range(world))
    ok = torch.equal(flat, expect)
    # a second step reuses the bucketer
    for i in range(len(sizes)):
        bk.mark_ready(i)
    bk.finish()
    ok = ok and torch.equal(flat, expect * world)
    # replicas that start different are detected, then fixed by the broadcast
    lin = torch.nn.Linear(3, 2)
    with torch.no_grad():
        lin.weight.fill_(float(rank))
    bad = False
    try:
        assert_replicas_identical(torch.cat([p.detach().flatten() for p in lin.parameters()]))
    except RuntimeError:
        bad = True
    broadcast_module_state_(lin)
    assert_replicas_identical(torch.cat([p.detach().flatten() for p in lin.parameters()]))
    q.put((rank, ok, bad))
    dist.destroy_process_group()


def test_bucketed_allreduce_order_is_rank_independent():
    world, port = 3, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok and bad for _, ok, bad in res), res


def _shard4_worker(rank, world, port, q):
    """BASELINE.json config 4: COCO_Search18, global batch 64 on 4 ranks (16 per rank): shards partition the batch, the
    all-reduced mask sums equal the global ones, and the rank-averaged sharded loss equals the single-process loss"""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import scanpath_oracle as O
    from scanpaths_amd.ddp import global_mask_normaliser, shard_batch
    from scanpaths_amd.synth import make_batch
    B, T = 64, 6
    full = make_batch("COCO_Search18", B, 16, 16, T, seed=33)          # tiny images: only targets / masks matter here
    full["fix_vectors"] = [[(i, t) for t in range(i % 5)] for i in range(B)]          # python lists, as the RL batches carry
    sh = shard_batch(full, rank, world)
    assert sh["images"].shape[0] == 16 and len(sh["fix_vectors"]) == 16 and sh["fix_vectors"][0] == full["fix_vectors"][16 * rank]
    assert torch.equal(sh["tasks"], full["tasks"][16 * rank:16 * rank + 16])
    A = full["scanpaths"].shape[-1]
    g = torch.Generator().manual_seed(5)
    z = torch.randn(B, T, A, generator=g, dtype=torch.float64)
    la_full = O.cross_entropy_loss(z, full["scanpaths"].double(), full["action_masks"].double())
    local = torch.stack([sh["action_masks"].double().sum(), sh["duration_masks"].double().sum()])
    norm = global_mask_normaliser(local)
    zs = z[16 * rank:16 * rank + 16]
    p = torch.softmax(zs, -1)
    la_r = -(sh["scanpaths"].double() * torch.log(p + O.EPS) * sh["action_masks"].double().unsqueeze(-1)).sum() / norm[0]
    tot = la_r.clone()
    dist.all_reduce(tot)
    q.put((rank, abs(float(tot / world - la_full)), abs(float(norm[0] * world - full["action_masks"].sum()))))
    dist.destroy_process_group()


def test_four_rank_coco_bs64_sharding():
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_shard4_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, lerr, serr in res:
        assert lerr < 1e-12 and serr < 1e-9, (rank, lerr, serr)


def test_shard_batch_rejects_empty_shards_and_slices_lists():
    import pytest
    from scanpaths_amd.ddp import shard_batch
    with pytest.raises(ValueError):
        shard_batch({"x": torch.arange(3)}, 0, 4)
    with pytest.raises(ValueError):
        shard_batch({"x": torch.arange(9)}, 3, 4)          # ceil(9/4) = 3 per rank -> rank 3 would be empty
    out = shard_batch({"x": torch.arange(8), "l": list("abcdefgh")}, 1, 4)
    assert out["l"] == ["c", "d"] and out["x"].tolist() == [2, 3]


def test_second_backward_before_step_raises_instead_of_drifting():
    """GradBucketer supports one backward per step; a parameter reporting twice (gradient accumulation, two losses)
    would accumulate into a bucket whose all-reduce is already in flight -- it must raise, not let the replicas drift."""
    import pytest
    from scanpaths_amd.ddp import GradBucketer
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        flat = torch.zeros(64)
        bk = GradBucketer(flat, [0, 16, 32, 48], 64, bucket_bytes=128)      # two buckets of two parameters
        for i in (3, 2, 1, 0):
            bk.mark_ready(i)
        with pytest.raises(RuntimeError, match="twice"):
            bk.mark_ready(3)
        bk.finish()
        for i in (3, 2, 1, 0):                                               # after finish() the next step starts clean
            bk.mark_ready(i)
        bk.finish()
        # a double report inside a bucket that has NOT been launched yet is detected as well (it used to decrement the counter
        # again and launch the bucket one report early) ...
        bk.mark_ready(3)
        with pytest.raises(RuntimeError, match="twice"):
            bk.mark_ready(3)
        # ... and a backward whose step() was skipped (non-finite-loss guard, caught exception) is recoverable: drain() -- what
        # FlatAdam.zero_grad() calls -- waits for what is in flight and starts a fresh round
        bk.drain()
        assert bk.pending == bk.members and not bk.handles and not any(bk.reported)
        for i in (3, 2, 1, 0):
            bk.mark_ready(i)
        bk.finish()
    finally:
        dist.destroy_process_group()
