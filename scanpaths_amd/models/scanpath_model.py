"""MI355X-native scanpath model: dilated ResNet encoder + attentive ConvLSTM decoder.

Host-side mirror of the reference ``baseline`` modules
  AiR            AiR/models/baseline_attention.py:187-504            forward(images, attention_maps, performances)
  OSIE           OSIE/models/baseline_attention.py:179-414           forward(images)
  COCO_Search18  COCO_Search18/models/baseline_attention_multihead.py:179-424   forward(images, attention_maps, tasks)
with the same constructor arguments, ``self.training``-switched forward, output dict keys and
state_dict key names / shapes (scanpaths_amd/spec.py), so the reference's train.py / test.py and its
checkpoints work unchanged.  All arithmetic runs in hand-written HIP kernels (scanpaths_amd/functional.py).

The decoder is evaluated in an algebraically reduced form (exact in real arithmetic,
rounding-level differences in fp32):
  * x-gate convolutions act on the step-invariant visual feature -> computed once as one 512->2048 conv
    (reference recomputes 4 convs per step, :44-50,304);
  * conv3x3(W, spatial (x) semantic) of a rank-1 tensor == 9-tap single-channel conv with the per-sample
    contracted filter W.semantic -> two small GEMMs instead of six 512->512 convs per step (:40-50);
  * 5x5 head conv followed by the linear 1x1 / 7x7-stride-5 predict_head convs (no non-linearity in
    between, :306-309,152-158) -> weights composed once per forward: 5x5 512->51 per head instead of 512->512,
    with the zero-padding semantics of the intermediate map preserved in the epilogue kernel;
  * semantic_att / spatial_att: the "cur" branch and all biases are constant along the softmax axis and
    cancel (:82-86,117-121) -> scores are <entry, u> with u composed from the "lists" and "attention"
    weights; semantic_cur / spatial_cur / those biases receive exactly zero gradient (the reference
    produces rounding-level noise there);
  * get_spatial_semantic = a * mean_c(vf) (:240-244) -> mean_c(vf) computed once.
Map size is derived from the input (Hm = H/8, Wm = W/8) instead of the hard-coded 30x40 (:105,142,145,208,279).
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import functional as F
from ..spec import ARCHS, COCO_OBJECTS, drt_hw, encoder_channels

HC = 64       # padded column count of one composed head (51 used)
NTAP = 49


# ----------------------------------------------------------------------------------------------------
# parameter holders (give the state_dict the reference's key structure)
# ----------------------------------------------------------------------------------------------------
class _Conv(nn.Module):
    def __init__(self, cin, cout, kh, kw=None, bias=True):
        super().__init__()
        kw = kh if kw is None else kw
        self.weight = nn.Parameter(torch.empty(cout, cin, kh, kw).contiguous(memory_format=torch.channels_last))
        if bias:
            self.bias = nn.Parameter(torch.zeros(cout))
        else:
            self.register_parameter("bias", None)


class _Linear(nn.Module):
    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin))
        self.bias = nn.Parameter(torch.zeros(cout))


class _BN(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))
        self.register_buffer("running_mean", torch.zeros(c))
        self.register_buffer("running_var", torch.ones(c))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))


class _Block(nn.Module):
    def __init__(self, kind, inpl, planes, exp, with_down):
        super().__init__()
        if kind == "bottleneck":
            self.conv1 = _Conv(inpl, planes, 1, bias=False)
            self.bn1 = _BN(planes)
            self.conv2 = _Conv(planes, planes, 3, bias=False)
            self.bn2 = _BN(planes)
            self.conv3 = _Conv(planes, planes * 4, 1, bias=False)
            self.bn3 = _BN(planes * 4)
        else:
            self.conv1 = _Conv(inpl, planes, 3, bias=False)
            self.bn1 = _BN(planes)
            self.conv2 = _Conv(planes, planes, 3, bias=False)
            self.bn2 = _BN(planes)
        if with_down:
            self.downsample = nn.Sequential(_Conv(inpl, planes * exp, 1, bias=False), _BN(planes * exp))
        else:
            self.downsample = None


def _make_encoder(arch):
    kind, counts, exp = ARCHS[arch]
    mods = [_Conv(3, 64, 7, bias=False), _BN(64), nn.Identity(), nn.Identity()]
    inpl = 64
    for li, n in enumerate(counts):
        planes = 64 * 2 ** li
        nominal_stride = 1 if li == 0 else 2
        blocks = []
        for bi in range(n):
            down = bi == 0 and (nominal_stride != 1 or inpl != planes * exp)
            blocks.append(_Block(kind, inpl, planes, exp, down))
            inpl = planes * exp
        mods.append(nn.Sequential(*blocks))
    return nn.Sequential(*mods)


class _LSTMParams(nn.Module):
    def __init__(self, streams):
        super().__init__()
        for g in ("input", "forget", "output", "memory"):
            setattr(self, g + "_x", _Conv(512, 512, 3))
        for g in ("input", "forget", "output", "memory"):
            setattr(self, g + "_h", _Conv(512, 512, 3))
        for sfx in streams:
            for g in ("input", "forget", "output"):
                setattr(self, g + sfx, _Conv(512, 512, 3))


class _SemAtt(nn.Module):
    def __init__(self):
        super().__init__()
        self.semantic_lists = _Linear(512, 512)
        self.semantic_cur = _Linear(512, 512)
        self.semantic_attention = _Linear(512, 1)


class _SpaAtt(nn.Module):
    def __init__(self, Hm, Wm):
        super().__init__()
        self.spatial_lists = _Conv(1, 1, 3)
        self.spatial_cur = _Conv(1, 1, 3)
        self.spatial_attention = _Conv(1, 1, Hm, Wm)


class _Head(nn.Module):
    def __init__(self, Hm, Wm):
        super().__init__()
        dh, dw = drt_hw(Hm, Wm)
        self.sal_layer_2 = _Conv(512, 1, 1)
        self.sal_layer_3 = _Conv(512, 1, 1)
        self.drt_layer_1 = _Conv(512, 1, 7)
        self.drt_layer_2 = _Conv(1, 2, dh, dw)


# ----------------------------------------------------------------------------------------------------
class ScanpathModel(nn.Module):
    """Generic over task ("AiR" | "OSIE" | "COCO_Search18"), encoder ("resnet50" | "resnet18") and map size."""

    def __init__(self, task="AiR", embed_size=512, convLSTM_length=16, min_length=1, ratio=4, map_width=40, map_height=30,
                 arch="resnet50"):
        super().__init__()
        assert task in ("AiR", "OSIE", "COCO_Search18")
        self.task, self.arch = task, arch
        self.embed_size, self.ratio = embed_size, ratio
        self.convLSTM_length, self.min_length = convLSTM_length, min_length
        self.downsampling_rate = 8
        self.map_width, self.map_height = map_width, map_height
        self.streams = ["_pos", "_neg"] if task == "AiR" else [""]
        # per-category heads only receive gradients for the categories present in a batch -> FlatAdam(conditional_params=True)
        self.has_conditional_params = task == "COCO_Search18"
        Hm, Wm = map_height, map_width
        P = Hm * Wm
        self.resnet = _make_encoder(arch)
        self.sal_conv = _Conv(encoder_channels(arch), 512, 3)
        self.lstm = _LSTMParams(self.streams)
        self.semantic_embed = _Linear(512, 512)
        self.spatial_embed = _Linear(P, P)
        self.semantic_att = _SemAtt()
        self.spatial_att = _SpaAtt(Hm, Wm)
        if task == "AiR":
            self.performance_situation = ["False", "True"]
            self.performance_sal_layer = nn.ModuleDict({k: _Conv(512, 512, 5) for k in self.performance_situation})
        elif task == "OSIE":
            self.performance_sal_layer = _Conv(512, 512, 5)
        else:
            self.object_name = list(COCO_OBJECTS)
            self.int2object = {i: n for i, n in enumerate(self.object_name)}
            self.object_sal_layer = nn.ModuleDict({k: _Conv(512, 512, 5) for k in self.object_name})
        self.object_head = _Head(Hm, Wm)
        self.last_decode_rows = None          # functional.DecodeRows of the latest decode() with autograd on (see decode)
        self.init_weights()

    # ---- nn.DataParallel (AiR/train.py:169-170; AiR/opts.py:26 makes gpu_ids=[0, 1] the reference's DEFAULT) -------------------------
    def _replicate_for_data_parallel(self):
        """torch.nn.parallel.replicate() calls this once per replica when nn.DataParallel runs a forward on MORE than one device.
        Refused: this model's parameters are views of FlatAdam's flat buffers (optim.py), its kernels write leaf gradients in place and
        draw scratch from per-process workspaces (hip.workspace) -- thread-per-device replicas inside one process would share all of
        it.  One process per GPU is the supported form (ddp.py).  A DataParallel wrapper over ONE device never replicates and works."""
        raise RuntimeError(
            "scanpaths_amd: this model cannot be replicated by nn.DataParallel (its parameters are views of the optimiser's flat "
            "buffers and its HIP kernels use per-process workspaces).  Pass a single id in --gpu_ids (the reference's default is "
            "[0, 1], AiR/opts.py:26), or run one process per GPU: `python -m torch.distributed.run --nproc-per-node N "
            "--master-addr 127.0.0.1 train.py` with scanpaths_amd.ddp (shard_batch, global_mask_normaliser, broadcast_module_state_) "
            "and scanpaths_amd.optim.FlatAdam -- see INTEGRATION.md, 'multi-GPU'.")

    # ---- init: resnet.py:112-118 (He normal, BN 1/0); mmcv xavier_init (normal) for convs, normal_init(std=0.01) for
    #      Linear (baseline_attention.py:58-65,90-97,126-133,176-185,495-504) ----
    def init_weights(self):
        for name, m in self.named_modules():
            if isinstance(m, _Conv):
                co, ci, kh, kw = m.weight.shape
                if name.startswith("resnet."):
                    nn.init.normal_(m.weight, 0.0, (2.0 / (kh * kw * co)) ** 0.5)
                else:
                    nn.init.xavier_normal_(m.weight)
                    if m.bias is not None:
                        nn.init.zeros_(m.bias)
            elif isinstance(m, _Linear):
                nn.init.normal_(m.weight, 0.0, 0.01)
                nn.init.zeros_(m.bias)

    # ------------------------------------------------------------------------------------------------
    def _bn(self, bn: _BN, x, residual=None, relu=True, emit_split=False, res_store=None, skip_z=False):
        # skip_dx: every BatchNorm of the encoder is the only consumer of the conv output it normalises
        y = F.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual, training=self.training, relu=relu,
                     emit_split=emit_split, res_store=res_store, skip_dx=True, skip_z=skip_z)
        if self.training:
            self._bn_seen.append(bn.num_batches_tracked)       # one fused increment per forward (encode) instead of 53 launches
        return y

    def encode(self, images):
        """models/resnet.py:96-152 + dilate_resnet (baseline_attention.py:226-238), NHWC.  emit_split: a BatchNorm whose output
        feeds a conv on the 2xfp16 split path writes that conv's operand in its own pass (F.bn_act)."""
        r = self.resnet
        self._bn_seen = []
        x = F.nchw_to_nhwc(images, 4)
        w0 = F.pad_last(r[0].weight.permute(0, 2, 3, 1), 4).permute(0, 3, 1, 2)     # Cin 3 -> 4 (zero)
        x = F.conv2d(x, w0, None, stride=2, pad=3)
        x = self._bn(r[1], x)
        st = self.training            # conv epilogues write the first stage of the batch statistics of the BatchNorm behind them
        x = F.maxpool3s2(x)
        kind = ARCHS[self.arch][0]
        blocks = []
        for li in range(4):
            first_stride = 2 if li == 2 else 1       # layer2[0], layer4[0] strides forced to 1
            dil = {2: 2, 3: 4}.get(li, 1)
            for bi, blk in enumerate(r[4 + li]):
                blocks.append((blk, first_stride if bi == 0 else 1, dil))
        for k, (blk, s, dil) in enumerate(blocks):
            # the block input has two consumers (conv1; identity or downsample conv): conv1's data gradient adds into the other one's
            gm = F.GradMerge() if (F.GRAD_MERGE and torch.is_grad_enabled() and x.requires_grad) else None
            x, x_side = F.tap(x, gm) if gm is not None else (x, x)
            if kind == "bottleneck":
                o = F.conv2d(x, blk.conv1.weight, None, stride=s, bn_stats=st, grad_accum=gm)
                # bn1 / bn2 feed exactly one conv: where that conv runs from the split operand alone the fp32 output stays unwritten
                o = self._bn(blk.bn1, o, emit_split=F.conv_takes_split(o.shape, blk.conv2.weight, pad=dil, dil=dil),
                             skip_z=F.conv_runs_from_split(o.shape, blk.conv2.weight, pad=dil, dil=dil,
                                                           need_dw=blk.conv2.weight.requires_grad))
                o = F.conv2d(o, blk.conv2.weight, None, pad=dil, dil=dil, bn_stats=st)
                o = self._bn(blk.bn2, o, emit_split=F.conv_takes_split(o.shape, blk.conv3.weight),
                             skip_z=F.conv_runs_from_split(o.shape, blk.conv3.weight, need_dw=blk.conv3.weight.requires_grad))
                o = F.conv2d(o, blk.conv3.weight, None, bn_stats=st)
                last = blk.bn3
            else:
                o = F.conv2d(x, blk.conv1.weight, None, stride=s, pad=1, bn_stats=st, grad_accum=gm)
                o = self._bn(blk.bn1, o, emit_split=F.conv_takes_split(o.shape, blk.conv2.weight, pad=dil, dil=dil))
                o = F.conv2d(o, blk.conv2.weight, None, pad=dil, dil=dil, bn_stats=st)
                last = blk.bn2
            idn = x_side
            if blk.downsample is not None:
                idn = self._bn(blk.downsample[1], F.conv2d(x_side, blk.downsample[0].weight, None, stride=s, bn_stats=st,
                                                           grad_store=gm), relu=False)
            if k + 1 < len(blocks):          # consumers of the block output: conv1 (and the downsample conv) of the next block
                nb, ns, _ = blocks[k + 1]
                emit = F.conv_takes_split(o.shape, nb.conv1.weight, stride=ns, pad=0 if kind == "bottleneck" else 1) or \
                    (nb.downsample is not None and F.conv_takes_split(o.shape, nb.downsample[0].weight, stride=ns))
            else:
                emit = True                  # the decoder's 3x3 feature conv
            x = self._bn(last, o, residual=idn, relu=True, emit_split=emit,
                         res_store=gm if blk.downsample is None else None)
        if self._bn_seen:
            torch._foreach_add_(self._bn_seen, 1)
        self._bn_seen = []
        return x

    # ------------------------------------------------------------------------------------------------
    @staticmethod
    def _cat_phys(ws: List[torch.Tensor]) -> torch.Tensor:
        """concatenate conv weights along Cout; returns the logical-OIHW view of a [sumCo,KH,KW,Ci] buffer."""
        return torch.cat([w.permute(0, 2, 3, 1) for w in ws], 0).permute(0, 3, 1, 2)

    def _sum_bias(self, names):
        b = getattr(self.lstm, names[0]).bias
        for n in names[1:]:
            b = F.add(b, getattr(self.lstm, n).bias)
        return b

    def _compose_heads(self, convs: List[_Conv]):
        """Compose each 5x5 head conv with sal_layer_2 / sal_layer_3 / drt_layer_1 (predict_head, :149-158)."""
        oh = self.object_head
        dev = oh.sal_layer_2.weight.device
        z = lambda *s: torch.zeros(*s, dtype=torch.float32, device=dev)
        wcomp = torch.cat([oh.sal_layer_2.weight.view(1, 512), oh.sal_layer_3.weight.view(1, 512),
                           oh.drt_layer_1.weight.permute(0, 2, 3, 1).reshape(NTAP, 512), z(HC - 2 - NTAP, 512)], 0)
        own = torch.cat([oh.sal_layer_2.bias, oh.sal_layer_3.bias, z(NTAP), oh.drt_layer_1.bias, z(HC - 3 - NTAP)])
        Gs, cbs = [], []
        for cv in convs:
            w5 = cv.weight.permute(0, 2, 3, 1).reshape(512, 25 * 512)                 # [o][tap*ci]
            Gs.append(F.gemm(wcomp, w5, None, "kn"))                                   # [HC, 25*512]
            b4 = torch.cat([cv.bias.view(1, 512), z(3, 512)], 0)
            cbs.append(F.add(F.gemm(wcomp, b4, None, "nk")[:, 0].contiguous(), own))   # [HC]
        G = torch.cat(Gs, 0).view(len(convs) * HC, 5, 5, 512).permute(0, 3, 1, 2)
        return G, torch.stack(cbs, 0)

    def _attention_vectors(self):
        sa, pa = self.semantic_att, self.spatial_att
        dev = sa.semantic_lists.weight.device
        u_sem = F.gemm(sa.semantic_attention.weight, sa.semantic_lists.weight, None, "kn").view(512)   # W_l^T w_a
        Hm, Wm = self.map_height, self.map_width
        acol = F.im2col3x3(pa.spatial_attention.weight.view(1, 1, Hm, Wm), 12).view(Hm * Wm, 12)
        wl = torch.cat([pa.spatial_lists.weight.reshape(9).flip(0), torch.zeros(3, device=dev)]).view(1, 12)
        wl4 = torch.cat([wl, torch.zeros(3, 12, device=dev)], 0)
        u_spa = F.gemm(acol, wl4, None, "nk")[:, 0].contiguous()
        # parameters that cancel along the softmax axis: exact zero gradient, but still decayed / stepped like the reference
        u_sem = F.touch_zero_grad(u_sem, [sa.semantic_lists.bias, sa.semantic_cur.weight, sa.semantic_cur.bias,
                                          sa.semantic_attention.bias])
        u_spa = F.touch_zero_grad(u_spa, [pa.spatial_lists.bias, pa.spatial_cur.weight, pa.spatial_cur.bias,
                                          pa.spatial_attention.bias])
        return u_sem, u_spa

    # ------------------------------------------------------------------------------------------------
    def _heads_prepare(self, B, Hm, Wm, dev, tasks=None):
        """Per-forward operands of predict_head (baseline_attention.py:149-174) behind the 5x5 head convs (:306-309; COCO: the
        per-sample head selected by task id, ...multihead.py:285-288), composed once: the saliency tap GEMM's weight, the 11x11 composite
        duration windows per border class, composed biases, the slot -> source-head map."""
        S = len(self.streams)
        if self.task == "AiR":
            head_convs = [self.performance_sal_layer["True"], self.performance_sal_layer["False"]]   # good, poor
        elif self.task == "OSIE":
            head_convs = [self.performance_sal_layer]
        else:
            head_convs = None
        nh = S
        if head_convs is not None:
            G, cb = self._compose_heads(head_convs)
            cbt = None
            nsrc, per_sample = len(head_convs), False
            hmap = torch.arange(nh, dtype=torch.int32, device=dev).repeat(B, 1).contiguous()
        else:       # COCO: per-sample head selected by task id (...multihead.py:285-288); only the tasks present are composed
            tl = [int(t) for t in tasks.tolist()]            # host sync, as the reference's int(tasks[index])
            uniq = sorted(set(tl))
            G, cbt = self._compose_heads([self.object_sal_layer[self.int2object[t]] for t in uniq])
            slot = torch.tensor([uniq.index(t) for t in tl], device=dev)
            cb = cbt.index_select(0, slot).view(B, 1, HC)
            nsrc, per_sample = len(uniq), True
            hmap = slot.to(torch.int32).view(B, 1).contiguous()
        # the composed 5x5 head is never run as a dense conv (csrc/head_direct.hip): the two saliency maps per head come from a
        # 1x1 "tap partial" GEMM + 25-tap gather, the duration sites from composite 11x11 stride-5 windows per border class
        Gp = G.permute(0, 2, 3, 1).reshape(nsrc, HC, 25, 512)
        R = (nsrc * 50 + 63) // 64 * 64
        Wsal = torch.cat([Gp[:, :2].reshape(nsrc * 50, 512), torch.zeros(R - nsrc * 50, 512, device=dev)], 0).view(R, 512, 1, 1)
        W11, cbsum = F.compose11(G, cbt if per_sample else cb, nsrc, HC, (Hm, Wm))
        return dict(nh=nh, nsrc=nsrc, per_sample=per_sample, hmap=hmap, Wsal=Wsal, W11=W11, cbsum=cbsum, cb=cb,
                    w2=self.object_head.drt_layer_2.weight, b2=self.object_head.drt_layer_2.bias)

    def _heads_step(self, hp, h_sal, h_drt, Wsal, W11, cbsum, cb, w2, b2, step=None):
        """predict_head on one hidden state (two aliases of it: the saliency path and the duration path each return a gradient):
        -> logits [nh,B,1+P], amap [nh,B,P], mu [nh,B], sigma2 [nh,B]"""
        Z2 = F.sal_gather(F.conv2d(h_sal, Wsal, None, pad=0, step=step), hp["hmap"], hp["nh"], hp["nsrc"], step=step)
        Dpre = F.drt_direct(h_drt, W11, cbsum, hp["hmap"], hp["nh"], step=step)
        return F.head_finish(Z2, cb, w2, b2, hp["nh"], HC, not self.training, per_sample=hp["per_sample"], dpre=Dpre)

    # ------------------------------------------------------------------------------------------------
    def decode(self, enc, attention_maps, tasks=None):
        """T steps of the attentive ConvLSTM; returns per-head stacks
        logits [nh,B,T,A], mu [nh,B,T], sigma2 [nh,B,T], amap [nh,B,T,P]."""
        T = self.convLSTM_length
        S = len(self.streams)
        grad_on_outer = torch.is_grad_enabled()
        vf = F.conv2d(enc, self.sal_conv.weight, self.sal_conv.bias, pad=1, relu=True)       # :270
        B, Hm, Wm, Cc = vf.shape
        assert (Hm, Wm) == (self.map_height, self.map_width), \
            f"input gives a {Hm}x{Wm} map but the model was built for {self.map_height}x{self.map_width}"
        P = Hm * Wm
        dev = vf.device
        L = self.lstm
        # hoisted x-gates, all gate biases folded in
        Wx = self._cat_phys([L.input_x.weight, L.forget_x.weight, L.output_x.weight, L.memory_x.weight])
        bias = torch.cat([self._sum_bias([g + "_x", g + "_h"] + [g + s for s in self.streams])
                          for g in ("input", "forget", "output")] + [self._sum_bias(["memory_x", "memory_h"])])
        # vf feeds the x-gate conv, its channel mean and the semantic pooling of every memory push: one gradient fan-in pass
        # (sp_sum_n) instead of T + 1 read-read-write adds by autograd (T memory pushes: the one after the last step is not run)
        n_vf = T + 2
        vfs = list(F.fanout(vf, n_vf)) if vf.requires_grad and n_vf <= 32 else [vf] * n_vf
        Xg = F.conv2d(vfs.pop(), Wx, bias, pad=1)
        Wh = self._cat_phys([L.input_h.weight, L.forget_h.weight, L.output_h.weight, L.memory_h.weight])
        # contracted-filter weights of the rank-1 gate terms, one [3*512*9, 512] matrix per stream.  (Batching the S contractions of a
        # decode step into one GEMM was tried: the batched fp32 kernel has no split-K, its data gradient -- M = 32 rows, K = 13 824 --
        # then runs on 8 workgroups: 354 us instead of 2 x 25 us.)
        Wr = [torch.cat([getattr(L, g + s).weight.permute(0, 2, 3, 1) for g in ("input", "forget", "output")], 0)
              .reshape(3 * 512 * 9, 512) for s in self.streams]
        KP = (9 * S + 3) // 4 * 4
        mvf = F.channel_mean(vfs.pop()).view(B * P)
        u_sem, u_spa = self._attention_vectors()
        hp = self._heads_prepare(B, Hm, Wm, dev, tasks)
        nh, nsrc, per_sample, hmap = hp["nh"], hp["nsrc"], hp["per_sample"], hp["hmap"]
        Wsal, W11, cbsum, cb, w2, b2 = hp["Wsal"], hp["W11"], hp["cbsum"], hp["cb"], hp["w2"], hp["b2"]
        # masked-step sparsity of the backward pass: every step-tagged op below carries this decode's token; the gate at the end of
        # this function fills it from the gradient that reaches the outputs (functional._OutputGate) -- no caller promise involved
        rows = F.DecodeRows() if grad_on_outer else None
        self.last_decode_rows = rows          # (diagnostics / tests: after backward, .rc.last = the horizon the gate derived per sample)
        at = (lambda t: rows.at(t)) if rows is not None else (lambda t: t)
        sp_list: List[List[torch.Tensor]] = []      # per memory entry: its remaining aliases (see push)
        se_list: List[List[torch.Tensor]] = []

        def rep(t, n):
            """n aliases of a tensor that every decode step (or memory push) uses once: its T (+1) gradient contributions are
            summed by ONE sp_sum_n pass instead of T small read-read-write adds that autograd would launch one by one"""
            return list(F.fanout(t, n)) if (t is not None and t.requires_grad and 1 < n <= 32) else [t] * n

        # weights of dense layers applied once per memory update / decode step: their T weight gradients are ONE GEMM over the
        # concatenated rows at the end of backward (F.DeferredGemmWgrad) -- the weight itself goes to every application
        grad_on = torch.is_grad_enabled()
        sp_defer = F.DeferredGemmWgrad() if (grad_on and self.spatial_embed.weight.requires_grad) else None
        se_defer = F.DeferredGemmWgrad() if (grad_on and self.semantic_embed.weight.requires_grad) else None
        wr_defer = [F.DeferredGemmWgrad() if (grad_on and w.requires_grad) else None for w in Wr]
        spw = [self.spatial_embed.weight] * T if sp_defer is not None else rep(self.spatial_embed.weight, T)
        sew = [self.semantic_embed.weight] * T if se_defer is not None else rep(self.semantic_embed.weight, T)
        spb, seb = rep(self.spatial_embed.bias, T), rep(self.semantic_embed.bias, T)
        mvfs, u_spas, u_sems = rep(mvf, T), rep(u_spa, T), rep(u_sem, T)
        Wrs = [[w] * T if d is not None else rep(w, T) for w, d in zip(Wr, wr_defer)]
        # The duration branch of predict_head (:155-159) feeds nothing inside the recurrence: its forward runs ONCE for all T steps behind the
        # loop (F.DrtBatch / F.drt_heads_batched; config drt_batched), the loop keeps the saliency part whose action map the next memory
        # update reads.  The per-step backward launches stay inside the backward recurrence.
        dbatch = F.DrtBatch(T) if F.DRT_BATCHED else None
        Wsals, W11s, cbsums, cbs = rep(Wsal, T), rep(W11, T), rep(cbsum, T), rep(cb, T + (1 if dbatch is not None else 0))
        w2s, b2s = (None, None) if dbatch is not None else (rep(w2, T), rep(b2, T))
        dpres = []
        sal_cache = {}          # split form of the saliency tap GEMM's weight: produced once per forward, not per decode step

        def push(amaps, k):       # amaps [S,B,P]; memory update number k (:277-296 / :317-336)
            # entry k of the two memory lists is stacked by this update and every later one (T - k of them: the update after the
            # last step feeds nothing and is not run): T - k aliases -> ONE fan-in pass for its gradient instead of the
            # T - k - 1 small adds autograd issues for a tensor that sits in T - k stacks (~250 launches per step over both lists)
            spf = F.mul_relu(amaps, mvfs.pop())
            sp_list.append(rep(F.linear(spf.view(S * B, P), spw.pop(), spb.pop(), defer=sp_defer), T - k))
            vf4 = vfs.pop()           # (4-D as it is: the gradient it returns carries the memory update's row-sparsity mark to vf's fan-in)
            pooled = F.semantic_pool(amaps, vf4, step=at(k), sbc=True) if (S <= 2 and Cc <= 512) else \
                F.gemm(amaps.transpose(0, 1).contiguous(), vf4.view(B, P, Cc), None, "kn", alpha=1.0 / P, relu=True).transpose(0, 1).contiguous()
            se_list.append(rep(F.linear(pooled.view(S * B, Cc), sew.pop(), seb.pop(), defer=se_defer), T - k))          # pooled [S,B,C]
            sp_mem = F.list_attention(torch.stack([a.pop() for a in sp_list], 0), u_spas.pop())        # [S*B,P]
            se_mem = F.list_attention(torch.stack([a.pop() for a in se_list], 0), u_sems.pop())        # [S*B,C]
            return sp_mem, se_mem

        a0 = attention_maps.reshape(1, B, P).to(torch.float32)
        sp_mem, se_mem = push(a0.expand(S, B, P).contiguous(), 0)
        h = c = None
        outs = {"logits": [], "amap": [], "mu": [], "s2": []}
        zpad = torch.zeros(B, 3 * 512, KP - 9 * S, device=dev) if KP > 9 * S else None
        wh_cache = {}          # split forms of the h-gate weight, shared by the T applications (and their backward)
        if torch.is_grad_enabled() and Wh.requires_grad:
            wh_cache["defer"] = F.DeferredWgrad()      # its T - 1 weight gradients: ONE launch on the current stream at the end of backward
        Xg_t = F.fanout(Xg, T) if (T > 1 and Xg.requires_grad) else (Xg,) * T      # one gradient fan-in pass instead of T-1 adds
        for t_ in range(T):
            t = at(t_)          # (an int that also names this decode: the backward kernels of step t find the row context through it)
            se = se_mem.view(S, B, Cc).unbind(0)      # (unbind: ONE stack in backward instead of a zero-fill + copy per stream and an add)
            parts = [F.gemm(se[s], Wrs[s].pop(), None, "nk", defer=wr_defer[s]).view(B, 3 * 512, 9) for s in range(S)]
            wc = torch.cat(parts + ([zpad] if zpad is not None else []), 2)
            spcol = F.im2col3x3(sp_mem.view(S, B, Hm, Wm), KP)
            hslot = (dbatch, t_) if dbatch is not None else None          # h_t goes into slot t of ONE [T, B, Hm, Wm, C] buffer
            if F.gateconv_lstm_fusable(h, Wh, spcol):       # the cell as the epilogue of the h-gate conv: no h-gate tensor
                h, c = F.gateconv_lstm(h, Wh, Xg_t[t], c, spcol, wc, wh_cache, step=t, hslot=hslot)
            else:
                hg = F.conv2d(h, Wh, None, pad=1, wcache=wh_cache, step=t) if h is not None else None        # step 0: h == 0
                h, c = F.lstm_cell_rank1(Xg_t[t], hg, c, spcol, wc, step=t, hslot=hslot)
            # h has three consumers (two heads now, the h-gate conv of the next step): one fan-in pass for its gradient
            nuse = 3 if t + 1 < T else 2
            # (step=t: under the masked-step sparsity of the backward pass the gradients of step t's heads are exact zeros for the samples
            # whose last loss step is earlier -- their backward kernels and h's fan-in skip those samples, functional.rows_ctx)
            h_sal, h_drt, h = (tuple(F.fanout(h, nuse, step=t)) + (None,))[:3] if h.requires_grad else (h, h, h)
            if dbatch is not None:
                Z2 = F.sal_gather(F.conv2d(h_sal, Wsals.pop(), None, pad=0, step=t, wcache=sal_cache), hmap, nh, nsrc, step=t)
                logits, amap = F.head_sal(Z2, cbs.pop(), nh, HC, not self.training, per_sample=per_sample)
                dpres.append(F.drt_direct(h_drt, W11s.pop(), cbsums.pop(), hmap, nh, step=t, batch=dbatch))
            else:
                logits, amap, mu, s2 = self._heads_step(hp, h_sal, h_drt, Wsals.pop(), W11s.pop(), cbsums.pop(), cbs.pop(), w2s.pop(),
                                                        b2s.pop(), step=t)
                outs["mu"].append(mu)
                outs["s2"].append(s2)
            outs["logits"].append(logits)
            outs["amap"].append(amap)
            if t + 1 < T:
                sp_mem, se_mem = push(amap, t + 1)
        stacks = {k: torch.stack(v, 2) for k, v in outs.items() if v}
        if dbatch is not None:
            stacks["mu"], stacks["s2"] = F.drt_heads_batched(dbatch, dpres, cbs.pop(), w2, b2, HC, per_sample=per_sample)
        if rows is not None and any(v.requires_grad for v in stacks.values()):
            keys = list(stacks)
            stacks = dict(zip(keys, F.output_gate(rows, [stacks[k] for k in keys])))
        return stacks

    # ------------------------------------------------------------------------------------------------
    def forward(self, images, attention_maps=None, third=None):
        if not images.is_cuda:
            raise RuntimeError("scanpaths_amd runs on a HIP device only (no CPU path); move the model and inputs to cuda")
        B = images.shape[0]
        Hm, Wm = self.map_height, self.map_width
        if self.task == "OSIE":
            attention_maps = torch.zeros(B, 1, Hm, Wm, device=images.device)        # OSIE/...:261
        enc = self.encode(images)
        d = self.decode(enc, attention_maps, third if self.task == "COCO_Search18" else None)
        T = self.convLSTM_length
        if self.task == "AiR":
            if self.training:                                   # :360-383
                sel = third.to(images.device).to(torch.bool)
                pick = lambda t: F.select_rows(t[0].reshape(B, -1), t[1].reshape(B, -1), sel)
                return {"all_actions_prob": pick(d["logits"]).view(B, T, -1),
                        "log_normal_mu": pick(d["mu"]).view(B, T),
                        "log_normal_sigma2": pick(d["s2"]).view(B, T)}
            res = {}
            for i, name in enumerate(("good", "poor")):         # :471-491
                res[name + "_all_actions_prob"] = d["logits"][i]
                res[name + "_log_normal_mu"] = d["mu"][i]
                res[name + "_log_normal_sigma2"] = d["s2"][i]
                res[name + "_action_map"] = d["amap"][i].view(B, T, Hm, Wm)
            return res
        if self.task == "OSIE" and self.training:               # OSIE/...:316-320
            return {"actions": d["logits"][0], "log_normal_mu": d["mu"][0], "log_normal_sigma2": d["s2"][0]}
        return {"all_actions_prob": d["logits"][0], "log_normal_mu": d["mu"][0], "log_normal_sigma2": d["s2"][0],
                "action_map": d["amap"][0].view(B, T, Hm, Wm)}
