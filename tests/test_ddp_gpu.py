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


def test_two_rank_step_equals_single_process():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_step, args=(1, 0, 0, q))
    p.start()
    _, loss1, tn1, flat1 = q.get(timeout=300)
    p.join(60)
    port = _free_port()
    procs = [ctx.Process(target=_one_step, args=(2, r, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = [q.get(timeout=300) for _ in range(2)]
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    for rank, loss2, tn2, flat2 in res:
        assert abs(loss2 - loss1) <= 1e-6 * abs(loss1), (rank, loss1, loss2)
        assert abs(tn2 - tn1) <= 1e-5 * tn1
        assert float(abs(flat2 - flat1).max()) <= 1e-6, rank      # identical up to the all-reduce's (a+a)/2 rounding


def _coco_step(world, rank, port, q):
    """COCO_Search18: every rank sees DIFFERENT target categories, so different per-category heads receive gradients"""
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


def test_ranks_with_different_coco_categories_stay_identical():
    """a per-category head is stepped when ANY rank produced a gradient for it (ddp.union_flags), as under the reference's single
    optimizer behind DataParallel: after one step both replicas hold bit-identical parameters and stepped the same heads"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_coco_step, args=(2, r, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = sorted([q.get(timeout=300) for _ in range(2)], key=lambda r: r[0])
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    assert (res[0][1] == res[1][1]).all()
    assert res[0][2] == res[1][2]
    heads = sorted({n.split(".")[1] for n in res[0][2] if n.startswith("object_sal_layer.")})
    assert len(heads) == 3, heads            # categories 1, 7, 12 and no other
