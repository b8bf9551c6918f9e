"""Shapes, seeds and seeded inputs of the per-module goldens (tests/golden/modules.npz, generated from the REAL reference modules by
tests/golden/make_golden_modules.py).  Shared by the generator, the CPU oracle test and the GPU op tests, so all three feed the same
bytes: inputs are numpy PCG64 draws in float64, laid out as the reference modules take them (NCHW)."""
import numpy as np
import torch

MODULE_CASES = {
    # ConvLSTM.forward (AiR/models/baseline_attention.py:37-56); P = 256 pixels per sample, C % 256 == 0: the shape class the fused cell
    # kernel and the split-emitting cell backward take at the benchmark size
    "convlstm": dict(kind="convlstm", seed=11, B=2, C=256, H=16, W=16),
    # performance_sal_layer conv 5x5 -> predict_head (:306-309, :149-174); the reference hard-codes the 30x40 map (:142, :145)
    "head_train": dict(kind="head", seed=12, B=1, H=30, W=40, training=True),
    "head_eval": dict(kind="head", seed=12, B=1, H=30, W=40, training=False),
    "spatial_att": dict(kind="spatial_att", seed=13, B=2, T=5, H=30, W=40),          # :111-124
    "semantic_att": dict(kind="semantic_att", seed=14, B=2, T=5, E=512),             # :77-88
    # Bottleneck.forward (AiR/models/resnet.py:57-93) after dilate_resnet (baseline_attention.py:226-238)
    "bottleneck_l3_train": dict(kind="bottleneck", seed=15, B=1, H=6, W=8, inpl=1024, planes=256, stride=1, dil=2, down=False,
                                prefix="resnet.6.1.", training=True),
    "bottleneck_l3_eval": dict(kind="bottleneck", seed=15, B=1, H=6, W=8, inpl=1024, planes=256, stride=1, dil=2, down=False,
                               prefix="resnet.6.1.", training=False),
    "bottleneck_l2first_train": dict(kind="bottleneck", seed=16, B=2, H=8, W=12, inpl=256, planes=128, stride=1, dil=1, down=True,
                                     prefix="resnet.5.0.", training=True),
    "bottleneck_l2first_eval": dict(kind="bottleneck", seed=16, B=2, H=8, W=12, inpl=256, planes=128, stride=1, dil=1, down=True,
                                    prefix="resnet.5.0.", training=False),
}


def case_inputs(name):
    """float64 input tensors of a case, in the reference's own layouts"""
    cs = MODULE_CASES[name]
    g = np.random.Generator(np.random.PCG64(1000 + cs["seed"]))
    n = lambda *s, scale=1.0: torch.from_numpy(g.standard_normal(s) * scale)
    u = lambda *s: torch.from_numpy(g.random(s))
    k = cs["kind"]
    if k == "convlstm":
        B, C, H, W = cs["B"], cs["C"], cs["H"], cs["W"]
        return {"x": n(B, C, H, W), "h": n(B, C, H, W, scale=0.5), "c": n(B, C, H, W, scale=0.7),
                "sp_pos": u(B, H, W), "sp_neg": u(B, H, W), "se_pos": n(B, C, scale=0.5), "se_neg": n(B, C, scale=0.5)}
    if k == "head":
        return {"h": n(cs["B"], 512, cs["H"], cs["W"], scale=0.5)}
    if k == "spatial_att":
        return {"lists": u(cs["B"], cs["T"], cs["H"], cs["W"]), "cur": u(cs["B"], 1, cs["H"], cs["W"])}
    if k == "semantic_att":
        return {"lists": n(cs["B"], cs["T"], cs["E"]), "cur": n(cs["B"], cs["E"])}
    if k == "bottleneck":
        return {"x": n(cs["B"], cs["inpl"], cs["H"], cs["W"]).abs()}          # a block input is a ReLU output
    raise KeyError(k)


def cotangents(name, outs):
    """the fixed cotangents the generator contracted the outputs with (same draws, same order): {output key: tensor}"""
    g = np.random.Generator(np.random.PCG64(MODULE_CASES[name]["seed"] + 977))
    return {k: torch.from_numpy(g.standard_normal(tuple(o.shape))) for k, o in outs.items()}
