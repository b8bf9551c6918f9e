"""two gloo ranks on one GPU vs single process: flat parameter difference after one step, with / without bucketed overlap"""
import os, sys, socket
import torch, torch.multiprocessing as mp
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def run(world, rank, port, q, bucket_mb):
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
    opt = FlatAdam(m.parameters(), lr=1e-3, weight_decay=5e-4, clip=12.5, bucket_mb=bucket_mb)
    b = {k: v.to(dev) for k, v in make_batch("OSIE", 2, 240, 320, 2, seed=8).items()}
    opt.zero_grad()
    pred = m(b["images"])
    sums = torch.cat([F.device_sum(b["action_masks"]), F.device_sum(b["duration_masks"])])
    loss, _, _ = supervised_loss(pred, b["scanpaths"], b["durations"], b["action_masks"], b["duration_masks"], 1.0, global_mask_normaliser(sums))
    loss.backward()
    torch.cuda.synchronize()
    g_local = opt.flat_g.detach().clone()           # before the reduction completes this may already hold reduced buckets
    opt.step()
    torch.cuda.synchronize()
    q.put((world, rank, bucket_mb, opt.flat_p.detach().cpu().numpy(), opt.flat_g.detach().cpu().numpy(), g_local.cpu().numpy()))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=run, args=(1, 0, 0, q, 0.0)); p.start(); base = q.get(timeout=300); p.join()
    for mb in (0.0, 4.0):
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        ps = [ctx.Process(target=run, args=(2, r, port, q, mb)) for r in range(2)]
        [x.start() for x in ps]
        res = [q.get(timeout=300) for _ in range(2)]
        [x.join() for x in ps]
        for w, r, b, fp, fg, gl in sorted(res, key=lambda t: t[1]):
            dp = abs(fp - base[3]); dg = abs(fg / 2 - base[4])
            print(f"bucket_mb {b}: rank {r}: max |p - p_single| {dp.max():.3e} (n>1e-6: {(dp > 1e-6).sum()}), max |g/2 - g_single| {dg.max():.3e} "
                  f"(n>0: {(dg > 0).sum()}), first bad index {int(dp.argmax())}, local-grad-vs-single max {abs(gl - base[4]).max():.3e} / {abs(gl/2 - base[4]).max():.3e}", flush=True)
