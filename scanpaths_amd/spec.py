"""state_dict key/shape specification of the reference models (the checkpoint boundary).

Keys and registration order follow the reference constructors:
  resnet.{0,1,4..7}.*        models/resnet.py:96-152 wrapped by nn.Sequential(children[:-2]) (baseline_attention.py:203)
  sal_conv, lstm.*, semantic_embed, spatial_embed, semantic_att.*, spatial_att.*,
  performance_sal_layer.{False,True} | performance_sal_layer | object_sal_layer.<18 names>,
  object_head.*              AiR/models/baseline_attention.py:188-218 (+ OSIE / COCO variants)
Shapes are PyTorch-native (OIHW for convs).  Map-size dependent entries (spatial_embed,
spatial_att.spatial_attention, object_head.drt_layer_2) are derived from (Hm, Wm) instead of the
reference's hard-coded 30x40.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Tuple

COCO_OBJECTS = ["bottle", "bowl", "car", "chair", "clock", "cup", "fork", "keyboard", "knife", "laptop",
                "microwave", "mouse", "oven", "potted plant", "sink", "stop sign", "toilet", "tv"]
ARCHS = {"resnet18": ("basic", [2, 2, 2, 2], 1), "resnet50": ("bottleneck", [3, 4, 6, 3], 4)}
LSTM_GATES_X = ["input_x", "forget_x", "output_x", "memory_x", "input_h", "forget_h", "output_h", "memory_h"]


def drt_hw(Hm: int, Wm: int) -> Tuple[int, int]:
    """Output size of drt_layer_1 (7x7, stride 5, pad 2) = kernel of drt_layer_2 ((6,8) at 30x40)."""
    return (Hm + 4 - 7) // 5 + 1, (Wm + 4 - 7) // 5 + 1


def _bn(spec, pfx, c):
    spec[pfx + ".weight"] = (c,)
    spec[pfx + ".bias"] = (c,)
    spec[pfx + ".running_mean"] = (c,)
    spec[pfx + ".running_var"] = (c,)
    spec[pfx + ".num_batches_tracked"] = ()


def encoder_spec(arch: str) -> "OrderedDict[str, tuple]":
    kind, counts, exp = ARCHS[arch]
    spec: "OrderedDict[str, tuple]" = OrderedDict()
    spec["resnet.0.weight"] = (64, 3, 7, 7)
    _bn(spec, "resnet.1", 64)
    inpl = 64
    for li, n in enumerate(counts):
        planes = 64 * 2 ** li
        nominal_stride = 1 if li == 0 else 2          # as constructed, before the dilation surgery
        for bi in range(n):
            p = f"resnet.{4 + li}.{bi}."
            if kind == "bottleneck":
                spec[p + "conv1.weight"] = (planes, inpl, 1, 1)
                _bn(spec, p + "bn1", planes)
                spec[p + "conv2.weight"] = (planes, planes, 3, 3)
                _bn(spec, p + "bn2", planes)
                spec[p + "conv3.weight"] = (planes * 4, planes, 1, 1)
                _bn(spec, p + "bn3", planes * 4)
            else:
                spec[p + "conv1.weight"] = (planes, inpl, 3, 3)
                _bn(spec, p + "bn1", planes)
                spec[p + "conv2.weight"] = (planes, planes, 3, 3)
                _bn(spec, p + "bn2", planes)
            if bi == 0 and (nominal_stride != 1 or inpl != planes * exp):
                spec[p + "downsample.0.weight"] = (planes * exp, inpl, 1, 1)
                _bn(spec, p + "downsample.1", planes * exp)
            inpl = planes * exp
    return spec


def encoder_channels(arch: str) -> int:
    return 512 * ARCHS[arch][2]


def model_spec(task: str, arch: str = "resnet50", Hm: int = 30, Wm: int = 40) -> "OrderedDict[str, tuple]":
    assert task in ("AiR", "OSIE", "COCO_Search18")
    spec = encoder_spec(arch)
    P = Hm * Wm

    def wb(name, w, b=None):
        spec[name + ".weight"] = w
        spec[name + ".bias"] = (w[0],) if b is None else b

    wb("sal_conv", (512, encoder_channels(arch), 3, 3))
    gates = list(LSTM_GATES_X)
    if task == "AiR":
        gates += ["input_pos", "forget_pos", "output_pos", "input_neg", "forget_neg", "output_neg"]
    else:
        gates += ["input", "forget", "output"]
    for g in gates:
        wb("lstm." + g, (512, 512, 3, 3))
    wb("semantic_embed", (512, 512))
    wb("spatial_embed", (P, P))
    wb("semantic_att.semantic_lists", (512, 512))
    wb("semantic_att.semantic_cur", (512, 512))
    wb("semantic_att.semantic_attention", (1, 512))
    wb("spatial_att.spatial_lists", (1, 1, 3, 3))
    wb("spatial_att.spatial_cur", (1, 1, 3, 3))
    wb("spatial_att.spatial_attention", (1, 1, Hm, Wm))
    if task == "AiR":
        for k in ("False", "True"):
            wb("performance_sal_layer." + k, (512, 512, 5, 5))
    elif task == "OSIE":
        wb("performance_sal_layer", (512, 512, 5, 5))
    else:
        for k in COCO_OBJECTS:
            wb("object_sal_layer." + k, (512, 512, 5, 5))
    wb("object_head.sal_layer_2", (1, 512, 1, 1))
    wb("object_head.sal_layer_3", (1, 512, 1, 1))
    wb("object_head.drt_layer_1", (1, 512, 7, 7))
    dh, dw = drt_hw(Hm, Wm)
    wb("object_head.drt_layer_2", (2, 1, dh, dw))
    return spec


def is_buffer(key: str) -> bool:
    return key.endswith(("running_mean", "running_var", "num_batches_tracked"))
