"""ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A literal CPU restatement (plain ``torch`` CPU ops, fp32 or fp64) of the one hot path of
chenxy99/Scanpaths: dilated ResNet -> sal_conv -> attentive ConvLSTM decoder -> heads -> loss
-> clip -> Adam.  "Literal" = the same operator sequence the reference executes (no hoisting,
no composition), written functionally over a state_dict so it is generic in
  * map size      (reference hard-codes 30x40 at AiR/models/baseline_attention.py:105,142,145,208,279)
  * encoder       (resnet50 | resnet18; reference always builds resnet50, :201)
  * task          ("AiR" two-stream good/poor, "OSIE" one stream, "COCO_Search18" per-task head)

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The product (scanpaths_amd) never does.

PARITY STATUS: **pinned** -- this restatement is checked against outputs of the reference itself,
imported in the survey container by ``tests/golden/make_golden.py`` (shims per SURVEY.md §8c) and
committed as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks it everywhere.
The reference's own tests pin nothing on this path (it has none).

Every function cites the reference lines it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

EPS = 1e-7  # AiR/models/loss.py:8

COCO_OBJECTS = ["bottle", "bowl", "car", "chair", "clock", "cup", "fork", "keyboard", "knife", "laptop",
                "microwave", "mouse", "oven", "potted plant", "sink", "stop sign", "toilet",
                "tv"]  # COCO_Search18/models/baseline_attention_multihead.py:203-205

RESNET_LAYERS = {"resnet18": ("basic", [2, 2, 2, 2]), "resnet50": ("bottleneck", [3, 4, 6, 3])}  # resnet.py:155-188


# --------------------------------------------------------------------------------------------
# encoder
# --------------------------------------------------------------------------------------------
def _bn(sd, pfx, x, training, bn_new):
    """nn.BatchNorm2d, eps 1e-5, momentum 0.1 (torch defaults; resnet.py:29,63).  In training mode the
    updated running statistics are returned through ``bn_new`` (functional: ``sd`` is never mutated)."""
    rm, rv = sd[pfx + ".running_mean"], sd[pfx + ".running_var"]
    if training:
        rm, rv = rm.clone(), rv.clone()
        y = F.batch_norm(x, rm, rv, sd[pfx + ".weight"], sd[pfx + ".bias"], True, 0.1, 1e-5)
        if bn_new is not None:
            bn_new[pfx + ".running_mean"] = rm
            bn_new[pfx + ".running_var"] = rv
            bn_new[pfx + ".num_batches_tracked"] = sd[pfx + ".num_batches_tracked"] + 1
        return y
    return F.batch_norm(x, rm, rv, sd[pfx + ".weight"], sd[pfx + ".bias"], False, 0.1, 1e-5)


def encoder(sd, images, arch="resnet50", training=False, bn_new=None, pfx="resnet."):
    """ResNet trunk with the SAM-style dilation surgery, avgpool/fc dropped.
    resnet.py:96-152 (trunk), :57-93 Bottleneck (stride on conv1 1x1), :25-54 BasicBlock,
    :104 MaxPool2d(3,2,pad 0,ceil_mode=True); baseline_attention.py:226-238 (layer2[0] & layer4[0]
    strides -> 1, layer3 conv2 dilation 2, layer4 conv2 dilation 4), :203 ([:-2])."""
    kind, counts = RESNET_LAYERS[arch]
    x = F.conv2d(images, sd[pfx + "0.weight"], None, stride=2, padding=3)
    x = F.relu(_bn(sd, pfx + "1", x, training, bn_new))
    x = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)
    for li, nblocks in enumerate(counts):
        seq = 4 + li                      # Sequential index of layer{li+1}
        first_stride = 2 if li == 2 else 1  # layer1: 1; layer2, layer4: forced to 1; layer3: 2
        dil = {2: 2, 3: 4}.get(li, 1)
        for bi in range(nblocks):
            p = f"{pfx}{seq}.{bi}."
            s = first_stride if bi == 0 else 1
            x = residual_block(sd, p, x, kind, s, dil, training, bn_new)
    return x


def residual_block(sd, p, x, kind, s, dil, training, bn_new=None):
    """one residual block with state_dict prefix p: Bottleneck.forward resnet.py:73-93 (stride on conv1, :61) / BasicBlock.forward
    :36-54, conv2 dilated as dilate_resnet leaves it (baseline_attention.py:232-238); pinned per module by tests/golden/modules.npz"""
    idn = x
    if kind == "bottleneck":
        o = F.conv2d(x, sd[p + "conv1.weight"], None, stride=s)
        o = F.relu(_bn(sd, p + "bn1", o, training, bn_new))
        o = F.conv2d(o, sd[p + "conv2.weight"], None, stride=1, padding=dil, dilation=dil)
        o = F.relu(_bn(sd, p + "bn2", o, training, bn_new))
        o = F.conv2d(o, sd[p + "conv3.weight"], None)
        o = _bn(sd, p + "bn3", o, training, bn_new)
    else:
        o = F.conv2d(x, sd[p + "conv1.weight"], None, stride=s, padding=1)
        o = F.relu(_bn(sd, p + "bn1", o, training, bn_new))
        o = F.conv2d(o, sd[p + "conv2.weight"], None, stride=1, padding=dil, dilation=dil)
        o = _bn(sd, p + "bn2", o, training, bn_new)
    if (p + "downsample.0.weight") in sd:
        idn = F.conv2d(x, sd[p + "downsample.0.weight"], None, stride=s)
        idn = _bn(sd, p + "downsample.1", idn, training, bn_new)
    return F.relu(o + idn)


# --------------------------------------------------------------------------------------------
# decoder pieces
# --------------------------------------------------------------------------------------------
def _conv(sd, name, x, **kw):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], **kw)


def _lin(sd, name, x):
    return F.linear(x, sd[name + ".weight"], sd[name + ".bias"])


def conv_lstm(sd, x, state, spatial: List[torch.Tensor], semantic: List[torch.Tensor], streams: List[str]):
    """ConvLSTM.forward: AiR baseline_attention.py:37-56 (streams ["_pos","_neg"]); OSIE/COCO
    OSIE/models/baseline_attention.py:33-48 (streams [""] -> lstm.input/forget/output).
    NB h' = o * c'  (no tanh on the cell, :53)."""
    h, c = state
    ss = [sp.unsqueeze(1) * se.unsqueeze(-1).unsqueeze(-1) for sp, se in zip(spatial, semantic)]
    pre = {}
    for g in ("input", "forget", "output"):
        a = _conv(sd, f"lstm.{g}_x", x, padding=1) + _conv(sd, f"lstm.{g}_h", h, padding=1)
        for sfx, s in zip(streams, ss):
            a = a + _conv(sd, f"lstm.{g}{sfx}", s, padding=1)
        pre[g] = a
    i, f, o = torch.sigmoid(pre["input"]), torch.sigmoid(pre["forget"]), torch.sigmoid(pre["output"])
    g = torch.tanh(_conv(sd, "lstm.memory_x", x, padding=1) + _conv(sd, "lstm.memory_h", h, padding=1))
    c2 = f * c + i * g
    h2 = o * c2
    return h2, (h2, c2)


def semantic_att(sd, lists, cur):
    """baseline_attention.py:77-88.  lists [N,t,E], cur [N,E]."""
    a = _lin(sd, "semantic_att.semantic_lists", lists) + _lin(sd, "semantic_att.semantic_cur", cur).unsqueeze(1)
    w = F.softmax(_lin(sd, "semantic_att.semantic_attention", a), 1)
    return (lists * w).sum(1)


def spatial_att(sd, lists, cur):
    """baseline_attention.py:111-124.  lists [N,t,H,W], cur [N,1,H,W].  The "attention" conv is a
    full-map (Hm x Wm) kernel -> one scalar per list entry (:105)."""
    n, t, hh, ww = lists.shape
    a = _conv(sd, "spatial_att.spatial_lists", lists.reshape(-1, 1, hh, ww), padding=1).view(n, t, hh, ww)
    a = a + _conv(sd, "spatial_att.spatial_cur", cur, padding=1)
    s = _conv(sd, "spatial_att.spatial_attention", a.reshape(-1, 1, hh, ww)).view(n, t, 1, 1)
    return (lists * F.softmax(s, 1)).sum(1)


def predict_head(sd, feat, training, tap=None, tag=None):
    """predict_head.forward, baseline_attention.py:149-174.  drt_layer_2's kernel spans the whole
    drt_layer_1 output ((6,8) at 30x40, :145); AvgPool over the whole map (:142).
    tap / tag (tests only): the pre-activations of the two ReLUs are recorded under tap["sal3_pre"][tag] / tap["drt1_pre"][tag]."""
    n = feat.shape[0]
    y = _conv(sd, "object_head.sal_layer_2", feat).squeeze(1)
    y = y.mean(dim=(1, 2), keepdim=False).view(n, 1, 1)
    t_pre = _conv(sd, "object_head.drt_layer_1", feat, stride=5, padding=2)
    t = F.relu(t_pre)
    t = _conv(sd, "object_head.drt_layer_2", t)
    mu = t[:, 0].reshape(n, -1)
    sigma2 = torch.exp(t[:, 1]).reshape(n, -1)
    x_pre = _conv(sd, "object_head.sal_layer_3", feat)
    if tap is not None:
        tap.setdefault("sal3_pre", {})[tag] = x_pre.detach()
        tap.setdefault("drt1_pre", {})[tag] = t_pre.detach()
    x = F.relu(x_pre)
    z = torch.cat([y, x.reshape(n, 1, -1)], dim=-1)
    if not training:
        z = F.softmax(z, -1)
    return {"actions": z, "log_normal_mu": mu, "log_normal_sigma2": sigma2, "action_map": x}


def _pool_spatial(a, vf):   # get_spatial_semantic, baseline_attention.py:240-244
    return (a.expand_as(vf) * vf).mean(1, keepdim=True)


def _pool_channel(a, vf):   # get_channel_semantic, baseline_attention.py:246-250
    return (a.expand_as(vf) * vf).flatten(2).mean(-1)


def sal_conv_key_in(sd):
    return sd["sal_conv.weight"].shape[1]


def forward(sd, task, images, attention_maps=None, performances=None, tasks=None, *, training, T=16,
            arch="resnet50", bn_new=None, tap=None, decode_samples=None, enc_out=None) -> Dict[str, torch.Tensor]:
    """baseline.forward -> training_process / inference.
    AiR: baseline_attention.py:253-493; OSIE: OSIE/models/baseline_attention.py:239-396;
    COCO: COCO_Search18/models/baseline_attention_multihead.py:246-406.
    tap (tests only, a dict): receives the inputs / pre-activations of the ReLUs whose parameters the gradient tests examine for
    mask flips -- "enc", "sal_conv_pre", per decode step t "h"[t] and, per (head name, t), "sal3_pre" / "drt1_pre".
    decode_samples (tests only, a list of sample indices): the encoder -- the only place where samples meet, through the train-mode
    BatchNorm statistics -- runs on the WHOLE batch, everything behind it (per-sample in the reference: no BatchNorm, per-sample
    memories) only on these samples; outputs then have len(decode_samples) rows.  Lets a full-batch check afford the fp64 decoder."""
    x = encoder(sd, images, arch, training, bn_new)
    if enc_out is not None:                 # (tests only, a dict: receives the encoder output of the whole batch)
        enc_out["enc"] = x.detach()
    if decode_samples is not None:
        idx = torch.as_tensor(list(decode_samples), dtype=torch.long)
        x, images = x[idx], images[idx]
        attention_maps = attention_maps[idx] if attention_maps is not None else None
        performances = performances[idx] if performances is not None else None
        tasks = tasks[idx] if tasks is not None else None
    n = images.shape[0]
    vf_pre = _conv(sd, "sal_conv", x, padding=1)
    if tap is not None:
        tap["enc"], tap["sal_conv_pre"] = x.detach(), vf_pre.detach()
    vf = F.relu(vf_pre)
    hm, wm = vf.shape[2], vf.shape[3]
    if task == "OSIE":                      # OSIE/...:261 zero attention map
        attention_maps = images.new_zeros((n, 1, hm, wm))
    streams = ["_pos", "_neg"] if task == "AiR" else [""]
    ns = len(streams)

    def memory_entry(amap):
        sp = F.relu(_pool_spatial(amap, vf))
        sp = _lin(sd, "spatial_embed", sp.view(n, 1, -1)).view(n, 1, hm, wm)
        se = _lin(sd, "semantic_embed", F.relu(_pool_channel(amap, vf)))
        return sp, se

    sp_lists = [[] for _ in range(ns)]
    se_lists = [[] for _ in range(ns)]
    sp_mem, se_mem = [None] * ns, [None] * ns

    def push(k, amap):
        sp, se = memory_entry(amap)
        sp_lists[k].append(sp)
        se_lists[k].append(se)
        sp_mem[k] = spatial_att(sd, torch.cat(sp_lists[k], 1), sp)
        se_mem[k] = semantic_att(sd, torch.stack(se_lists[k], 1), se)

    for k in range(ns):
        push(k, attention_maps)
    state = (torch.zeros_like(vf), torch.zeros_like(vf))
    per_head: List[List[dict]] = [[] for _ in range(ns)]
    for step in range(T):
        out, state = conv_lstm(sd, vf, state, sp_mem, se_mem, streams)
        if tap is not None:
            tap.setdefault("h", {})[step] = out.detach()
        if task == "AiR":
            heads = [predict_head(sd, _conv(sd, "performance_sal_layer.True", out, padding=2), training, tap, ("True", step)),
                     predict_head(sd, _conv(sd, "performance_sal_layer.False", out, padding=2), training, tap, ("False", step))]
        elif task == "OSIE":
            heads = [predict_head(sd, _conv(sd, "performance_sal_layer", out, padding=2), training, tap, ("", step))]
        else:   # per-sample batch-1 convs keyed by task id, ...multihead.py:285-288
            feats = [_conv(sd, "object_sal_layer." + COCO_OBJECTS[int(tasks[b])], out[b:b + 1], padding=2)
                     for b in range(n)]
            heads = [predict_head(sd, torch.cat(feats, 0), training)]
        for k in range(ns):
            per_head[k].append(heads[k])
            push(k, heads[k]["action_map"])

    cat = [{key: torch.cat([st[key] for st in per_head[k]], 1) for key in per_head[k][0]} for k in range(ns)]
    if task == "AiR":
        if training:                         # :360-383 per-sample select; key is LOGITS despite the name
            sel = performances.to(torch.bool)
            pick = lambda key: torch.where(sel.view(n, *([1] * (cat[0][key].dim() - 1))), cat[0][key], cat[1][key])
            return {"all_actions_prob": pick("actions"), "log_normal_mu": pick("log_normal_mu"),
                    "log_normal_sigma2": pick("log_normal_sigma2")}
        res = {}
        for name, c in zip(("good", "poor"), cat):   # :471-491
            res[name + "_all_actions_prob"] = c["actions"]
            res[name + "_log_normal_mu"] = c["log_normal_mu"]
            res[name + "_log_normal_sigma2"] = c["log_normal_sigma2"]
            res[name + "_action_map"] = c["action_map"]
        return res
    c = cat[0]
    if task == "OSIE" and training:          # OSIE/...:316-320
        return {"actions": c["actions"], "log_normal_mu": c["log_normal_mu"],
                "log_normal_sigma2": c["log_normal_sigma2"]}
    return {"all_actions_prob": c["actions"], "log_normal_mu": c["log_normal_mu"],
            "log_normal_sigma2": c["log_normal_sigma2"], "action_map": c["action_map"]}


# --------------------------------------------------------------------------------------------
# losses, optimiser step, sampling
# --------------------------------------------------------------------------------------------
def cross_entropy_loss(z, gt, mask):
    """AiR/models/loss.py:10-14 (soft target, eps inside the log, masked mean)."""
    p = F.softmax(z, dim=-1)
    return -(gt * torch.log(p + EPS) * mask.unsqueeze(-1)).sum() / mask.sum()


def lognormal_nll(mu, sigma2, gt, mask):
    """MLPLogNormalDistribution, AiR/models/loss.py:27-32."""
    logpdf = torch.log(1 / (gt + EPS) * 1 / torch.sqrt(2 * math.pi * sigma2)) \
        + (-(torch.log(gt + EPS) - mu) ** 2 / (2 * sigma2))
    return -(logpdf[mask == 1]).sum() / mask.sum()


def supervised_loss(pred, batch, lambda_1=1.0, task="AiR"):
    """AiR/train.py:192-197."""
    z = pred["actions"] if "actions" in pred else pred["all_actions_prob"]
    la = cross_entropy_loss(z, batch["scanpaths"], batch["action_masks"])
    ld = lognormal_nll(pred["log_normal_mu"], pred["log_normal_sigma2"], batch["durations"], batch["duration_masks"])
    return la + lambda_1 * ld, la, ld


def clip_and_adam(params: Dict[str, torch.Tensor], grads: Dict[str, torch.Tensor], state: dict, *, lr, clip=12.5,
                  betas=(0.9, 0.999), eps=1e-8, weight_decay=5e-5) -> float:
    """clip_grad_norm_(12.5) then torch.optim.Adam (L2 folded into the gradient; not AdamW).
    AiR/train.py:116-117,200-202; opts.py:15,25.  In-place on ``params``; returns the total norm."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values() if g is not None)).item()
    scale = min(1.0, clip / (total + 1e-6)) if clip > 0 else 1.0
    state["step"] = state.get("step", 0) + 1
    t = state["step"]
    b1, b2 = betas
    for k, p in params.items():
        if grads.get(k) is None:        # torch.optim.Adam skips parameters without a gradient (unused COCO heads)
            continue
        g = grads[k] * scale + weight_decay * p
        m = state.setdefault("m." + k, torch.zeros_like(p))
        v = state.setdefault("v." + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v.sqrt() / math.sqrt(1 - b2 ** t)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / (1 - b1 ** t))
    return total


def generate_scanpath(sample_actions, durations, map_w, map_h, width, height):
    """Sampling.generate_scanpath, models/sampling.py:48-77 (pure index arithmetic)."""
    xg, yg = float(width / map_w), float(height / map_h)
    n, T = sample_actions.shape
    amask = torch.zeros((n, T))
    dmask = torch.zeros((n, T))
    fix = []
    for b in range(n):
        v = []
        for t in range(T):
            a = int(sample_actions[b, t])
            amask[b, t] = 1
            if a == 0:
                break
            a -= 1
            v.append(((a % map_w) * xg + xg / 2, (a // map_w) * yg + yg / 2, float(durations[b, t])))
            dmask[b, t] = 1
        fix.append(v)
    return fix, amask, dmask


def scanpath_length(selected_actions, T):
    """The first-terminate scan of Sampling.random_sample, models/sampling.py:29-34.
    Quirk kept: a terminate at t=0 leaves length 0 -> overwritten by T."""
    n = selected_actions.shape[0]
    length = torch.zeros(n)
    for t in range(T):
        length[torch.logical_and(length == 0, selected_actions[:, t] == 0)] = t
    length[length == 0] = T
    return length.unsqueeze(-1)


# ---- RL (self-critical) phase log-probabilities, AiR/models/loss.py:34-45 ------------------------------------------------
def log_action(p, mask):
    return (torch.log(p + EPS) * mask).sum(dim=-1) / mask.sum()


def log_duration(d, mu, sigma2, mask):
    items = torch.log(1 / (d + EPS) * 1 / (torch.sqrt(2 * math.pi * sigma2))) + (-(torch.log(d + EPS) - mu) ** 2 / (2 * sigma2))
    return (items * mask).sum(dim=-1) / mask.sum()
