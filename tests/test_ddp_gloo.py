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
