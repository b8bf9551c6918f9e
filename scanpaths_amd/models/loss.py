"""Drop-in for the supervised losses of ``models/loss.py`` (the two the train step calls, AiR/train.py:192-195).

CrossEntropyLoss(input, gt, mask)                      AiR/models/loss.py:10-14
MLPLogNormalDistribution(mu, sigma2, gt, mask)         AiR/models/loss.py:27-32
Each is computed by the fused HIP loss kernel (value + gradient in one pass).  ``supervised_loss`` is the fused
form of ``loss_actions + lambda_1 * loss_duration`` (one launch, what bench.py times)."""
import torch

from .. import functional as F

epsilon = 1e-7


def supervised_loss(predicts, scanpaths, durations, action_masks, duration_masks, lambda_1=1.0, mask_sums=None):
    """loss = L_actions + lambda_1 * L_duration of the supervised phase (AiR/train.py:192-197) -> (loss, L_actions, L_duration); ONE launch
    for value and gradient.  Like the two separate calls below it sends exact zeros into the predictions of masked-out steps; the decoder's
    backward pass finds those in the gradient it receives and skips what they imply (functional._OutputGate) -- nothing to switch on."""
    z = predicts["actions"] if "actions" in predicts else predicts["all_actions_prob"]
    return F.scanpath_loss(z, predicts["log_normal_mu"], predicts["log_normal_sigma2"], scanpaths, action_masks, durations,
                           duration_masks, lambda_1, mask_sums)


def CrossEntropyLoss(input, gt, mask):
    B, T, _ = input.shape
    dummy = torch.ones(B, T, device=input.device)
    zero = torch.zeros(B, T, device=input.device)
    ones2 = torch.cat([F.device_sum(mask), torch.ones(1, device=input.device)])
    _, la, _ = _both(input, dummy, dummy, gt, mask, dummy, zero, ones2)
    return la


def MLPLogNormalDistribution(log_normal_mu, log_normal_sigma2, gt, mask):
    B, T = log_normal_mu.shape
    z = torch.zeros(B, T, 4, device=gt.device)
    g = torch.zeros(B, T, 4, device=gt.device)
    zero = torch.zeros(B, T, device=gt.device)
    sums = torch.cat([torch.ones(1, device=gt.device), F.device_sum(mask)])
    _, _, ld = _both(z, log_normal_mu, log_normal_sigma2, g, zero, gt, mask, sums)
    return ld


class _Split(torch.autograd.Function):
    """expose loss_actions / loss_duration of the fused kernel as separately differentiable scalars"""
    @staticmethod
    def forward(ctx, z, mu, s2, gt, am, dur, dm, sums):
        from ..functional import _ScanpathLoss
        with torch.enable_grad():
            zz, mm, ss = z.detach().requires_grad_(True), mu.detach().requires_grad_(True), s2.detach().requires_grad_(True)
            loss, la, ld = _ScanpathLoss.apply(zz, mm, ss, gt, am, dur, dm, 1.0, sums)
            gz, gm, gs = torch.autograd.grad(loss, (zz, mm, ss))
        ctx.save_for_backward(gz, gm, gs)
        return loss.detach(), la.detach(), ld.detach()

    @staticmethod
    def backward(ctx, g0, g1, g2):
        gz, gm, gs = ctx.saved_tensors
        # dz carries only the action term, dmu/dsigma2 only the duration term (lambda_1 = 1 inside)
        ga = (g0 + g1).reshape(1).float().contiguous()
        gd = (g0 + g2).reshape(1).float().contiguous()
        from .. import hip
        outs = []
        for t, g in ((gz, ga), (gm, gd), (gs, gd)):
            o = torch.empty_like(t)
            hip.check(hip.lib().sp_scale_by(hip.ptr(t), hip.ptr(g), t.numel(), hip.ptr(o), hip.stream()), "sp_scale_by")
            outs.append(o)
        return outs[0], outs[1], outs[2], None, None, None, None, None


def _both(z, mu, s2, gt, am, dur, dm, sums):
    return _Split.apply(z, mu, s2, gt.contiguous(), am.contiguous(), dur.contiguous(), dm.contiguous(), sums)


# ---- RL (self-critical) phase: per-sample log-probabilities of sampled scanpaths, AiR/models/loss.py:34-45 ----------------
class _LogAction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, mask):
        from ..hip import check, lib, ptr, stream
        p, m = input.contiguous().float(), mask.contiguous().float()
        B, T = p.shape
        out = torch.empty(B, device=p.device)
        coef = torch.empty_like(p)
        check(lib().sp_log_action(ptr(p), ptr(m), B, T, ptr(F.device_sum(m)), ptr(out), ptr(coef), stream()), "sp_log_action")
        ctx.save_for_backward(coef)
        return out

    @staticmethod
    def backward(ctx, g):
        from ..hip import check, lib, ptr, stream
        (coef,) = ctx.saved_tensors
        B, T = coef.shape
        dp = torch.empty_like(coef)
        check(lib().sp_rowscale(ptr(coef), ptr(g.contiguous()), B, T, ptr(dp), stream()), "sp_rowscale")
        return dp, None


class _LogDuration(torch.autograd.Function):
    @staticmethod
    def forward(ctx, input, mu, sigma2, mask):
        from ..hip import check, lib, ptr, stream
        d, mu_, s2, m = (t.contiguous().float() for t in (input, mu, sigma2, mask))
        B, T = d.shape
        out = torch.empty(B, device=d.device)
        dmu, ds2 = torch.empty_like(d), torch.empty_like(d)
        check(lib().sp_log_duration(ptr(d), ptr(mu_), ptr(s2), ptr(m), B, T, ptr(F.device_sum(m)), ptr(out), ptr(dmu), ptr(ds2),
                                    stream()), "sp_log_duration")
        ctx.save_for_backward(dmu, ds2)
        return out

    @staticmethod
    def backward(ctx, g):
        from ..hip import check, lib, ptr, stream
        dmu, ds2 = ctx.saved_tensors
        B, T = dmu.shape
        g = g.contiguous()
        gmu, gs2 = torch.empty_like(dmu), torch.empty_like(ds2)
        check(lib().sp_rowscale(ptr(dmu), ptr(g), B, T, ptr(gmu), stream()), "sp_rowscale")
        check(lib().sp_rowscale(ptr(ds2), ptr(g), B, T, ptr(gs2), stream()), "sp_rowscale")
        return None, gmu, gs2, None


def LogAction(input, mask):
    """[B] = sum_t log(p + eps) * mask / mask.sum()  -- probabilities of the sampled actions (AiR/models/loss.py:34-37)"""
    return _LogAction.apply(input, mask)


def LogDuration(input, log_normal_mu, log_normal_sigma2, mask):
    """[B] = sum_t logpdf_lognormal(duration; mu, sigma2) * mask / mask.sum()  (AiR/models/loss.py:39-45); the sampled
    durations carry no gradient (the reference passes durations.data)"""
    return _LogDuration.apply(input, log_normal_mu, log_normal_sigma2, mask)
