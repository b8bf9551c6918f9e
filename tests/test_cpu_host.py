"""CPU-only checks: the C-ABI library loads and exports every symbol the header declares, the host-side mirror of the
reference surface (state_dict keys, CLI flags, sampling index arithmetic), the synthetic generators, and the refusal
to run without a HIP device."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "scanpaths_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(sp_[A-Za-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from scanpaths_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = ctypes.CDLL(hip.LIB_PATH)
    syms = _header_symbols()
    assert len(syms) >= 35
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/scanpaths_amd.h but not exported"
    assert sorted(hip.SIGNATURES) == syms, set(hip.SIGNATURES) ^ set(syms)
    # pure host call, no GPU needed: the library, the header and the ctypes binding carry the same ABI version
    hdr = int(re.search(r"#define SP_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "scanpaths_amd.h")).read()).group(1))
    assert lib.sp_abi_version() == hdr == hip.ABI_VERSION


def test_ctypes_signatures_match_the_header():
    """ABI drift guard: for every entry point the ctypes binding (hip.SIGNATURES) has as many arguments as the header declares,
    8-byte integers / doubles / floats where the header says so, and the return type (int / int64_t)."""
    from scanpaths_amd import hip
    txt = open(os.path.join(ROOT, "include", "scanpaths_amd.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    decls = dict()
    for m in re.finditer(r"\b(int64_t|int)\s+(sp_[A-Za-z0-9_]+)\s*\(([^;{]*?)\)\s*;", txt, flags=re.S):
        ret, name, args = m.group(1), m.group(2), " ".join(m.group(3).split())
        decls[name] = (ret, [] if args in ("", "void") else [a.strip() for a in args.split(",")])
    assert set(decls) == set(hip.SIGNATURES), set(decls) ^ set(hip.SIGNATURES)
    for name, (ret, args) in decls.items():
        cret, cargs = hip.SIGNATURES[name]
        assert len(cargs) == len(args), (name, len(cargs), args)
        assert (cret is ctypes.c_int64) == (ret == "int64_t"), (name, ret, cret)
        for a, c in zip(args, cargs):
            if "*" in a:
                assert c in (ctypes.c_void_p, ctypes.c_char_p) or issubclass(c, ctypes._Pointer), (name, a, c)      # void* or POINTER(desc struct)
            elif a.startswith("int64_t"):
                assert c is ctypes.c_int64, (name, a, c)
            elif a.startswith("uint64_t"):
                assert c is ctypes.c_uint64, (name, a, c)
            elif a.startswith("double"):
                assert c is ctypes.c_double, (name, a, c)
            elif a.startswith("float"):
                assert c is ctypes.c_float, (name, a, c)
            else:
                assert a.startswith("int") and c is ctypes.c_int, (name, a, c)


@pytest.mark.parametrize("case,task", [("air_train_T4", "AiR"), ("osie_r18_train_T8", "OSIE"), ("coco_train_T6", "COCO_Search18")])
def test_state_dict_keys_and_parameter_order_match_reference(case, task):
    """parameter registration order == the reference's model.named_parameters() (stored with the goldens): optimizer
    state in reference checkpoints is indexed by that order (utils/checkpointing.py:93-110)."""
    from scanpaths_amd.models.scanpath_model import ScanpathModel
    from scanpaths_amd.spec import is_buffer, model_spec
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", case + ".json")))
    m = ScanpathModel(task, convLSTM_length=2, arch=meta["arch"])
    assert [k for k, _ in m.named_parameters()] == meta["param_names"]
    spec = model_spec(task, meta["arch"])
    sd = m.state_dict()
    assert list(sd) == list(spec) and all(tuple(sd[k].shape) == tuple(spec[k]) for k in spec)
    assert [k for k in spec if not is_buffer(k)] == meta["param_names"]
    w = sd["lstm.input_x.weight"]
    assert w.permute(0, 2, 3, 1).is_contiguous()       # [Co][KH][KW][Ci] physical layout the kernels consume


def test_map_size_generic_spec():
    from scanpaths_amd.spec import drt_hw, model_spec
    assert drt_hw(30, 40) == (6, 8)                     # reference's hard-coded (6,8), baseline_attention.py:145
    s = model_spec("AiR", "resnet50", 40, 64)
    assert s["spatial_embed.weight"] == (2560, 2560) and s["object_head.drt_layer_2.weight"] == (2, 1, 8, 13)
    assert s["spatial_att.spatial_attention.weight"] == (1, 1, 40, 64)


def test_no_cpu_path():
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd import functional as F, hip
    m = baseline(convLSTM_length=1)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 3, 240, 320), torch.zeros(1, 1, 30, 40), torch.ones(1, dtype=torch.bool))
    with pytest.raises(hip.HipError):
        F.conv2d(torch.zeros(1, 4, 4, 32), torch.zeros(32, 32, 3, 3), None, pad=1)


def test_data_parallel_replication_is_refused_with_directions():
    """The reference wraps the model in nn.DataParallel whenever len(gpu_ids) > 1 (AiR/train.py:169-170) and gpu_ids defaults to
    [0, 1] (AiR/opts.py:26, mirrored by scanpaths_amd.opts).  Here replication inside one process is refused by the hook
    torch.nn.parallel.replicate() calls per replica, with a message that names the supported forms; the wrapper itself (which a
    single-device DataParallel never replicates through) constructs and keeps the reference's attribute access `model.module`."""
    from scanpaths_amd.models.baseline_attention import baseline
    from scanpaths_amd.opts import parse_opt
    assert parse_opt("AiR", []).gpu_ids == [0, 1]                   # the reference's default: the wrap IS the default code path
    m = baseline(convLSTM_length=1)
    with pytest.raises(RuntimeError) as ei:
        m._replicate_for_data_parallel()
    msg = str(ei.value)
    for needle in ("nn.DataParallel", "--gpu_ids", "torch.distributed.run", "scanpaths_amd.ddp", "FlatAdam", "INTEGRATION.md"):
        assert needle in msg, (needle, msg)
    # replicate() reaches the hook for the ROOT module first: the refusal comes before any replica exists.  torch's own modules
    # keep their default hook (only the model root refuses), so sub-modules stay usable inside other containers.
    assert type(m.sal_conv)._replicate_for_data_parallel is torch.nn.Module._replicate_for_data_parallel
    import inspect
    from torch.nn.parallel import replicate          # (the call site this hook belongs to exists in the installed torch)
    assert "_replicate_for_data_parallel()" in inspect.getsource(replicate)
    w = torch.nn.DataParallel(m, device_ids=None) if not torch.cuda.is_available() else torch.nn.DataParallel(m, device_ids=[0])
    assert w.module is m


def test_opts_flag_surface():
    from scanpaths_amd.opts import parse_opt
    a = parse_opt("AiR", [])
    assert (a.width, a.height, a.map_width, a.map_height) == (320, 240, 40, 30)
    assert (a.clip, a.batch, a.lr, a.weight_decay, a.lambda_1, a.lambda_5, a.max_length) == (12.5, 16, 1e-4, 5e-5, 1, -2.0, 16)
    assert parse_opt("OSIE", []).weight_decay == 5e-4 and not hasattr(parse_opt("OSIE", []), "lambda_5")
    c = parse_opt("COCO_Search18", ["--detector_threshold", "0.5", "--batch", "64"])
    assert c.detector_threshold == 0.5 and c.batch == 64 and not hasattr(c, "att_dir")
    assert parse_opt("AiR", ["--ablate_attention_info", "False"]).ablate_attention_info is True   # type=bool quirk kept
    assert parse_opt("AiR", ["--set_cfgs", "lr", "0.01"]).lr == 0.01                              # set_cfgs > defaults
    assert parse_opt("AiR", ["--set_cfgs", "lr", "0.01", "--lr", "0.5"]).lr == 0.5                 # explicit CLI > set_cfgs (:70)
    # --cfg: plain YAML is merged; a file that relies on the reference's _BASE_ inheritance (utils/config.py:15-144) is refused loudly
    import tempfile
    with tempfile.TemporaryDirectory() as d:
        plain, based = os.path.join(d, "a.yaml"), os.path.join(d, "b.yaml")
        open(plain, "w").write("lr: 0.003\nbatch: 8\n")
        open(based, "w").write("_BASE_: a.yaml\nlr: 0.004\n")
        a = parse_opt("AiR", ["--cfg", plain])
        assert a.lr == 0.003 and a.batch == 8
        with pytest.raises(ValueError, match="_BASE_"):
            parse_opt("AiR", ["--cfg", based])


def test_config_switchboard_ignores_stray_environment_variables():
    """SP_* tuning variables are honoured only under SP_ALLOW_ENV_TUNING=1; the throughput mode and the timing library
    announce themselves; bench.py can list the non-default switches"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import scanpaths_amd.functional as F, scanpaths_amd.config as c;"
            "print(F.SPLIT_SCHEME, F.THROUGHPUT_MODE, F.BN_SPLIT, c.non_default())")
    run = lambda env: subprocess.run([sys.executable, "-c", code], cwd=root, env={**os.environ, **env}, capture_output=True, text=True)
    base = {k: "" for k in ("SP_ALLOW_ENV_TUNING", "SP_SPLIT_SCHEME", "SP_BN_SPLIT", "SP_LIBRARY")}
    r = run({**base, "SP_SPLIT_SCHEME": "f16x1", "SP_BN_SPLIT": "0"})
    assert r.stdout.split()[:3] == ["f16x2", "False", "True"] and "IGNORING" in r.stderr, (r.stdout, r.stderr)
    r = run({**base, "SP_ALLOW_ENV_TUNING": "1", "SP_SPLIT_SCHEME": "f16x1", "SP_BN_SPLIT": "0"})
    assert r.stdout.split()[:3] == ["f16x2", "True", "False"] and "THROUGHPUT MODE" in r.stderr, (r.stdout, r.stderr)
    assert "'split_scheme': 'f16x1'" in r.stdout and "'bn_split': False" in r.stdout
    from scanpaths_amd import config
    with pytest.raises(KeyError):
        config.set(no_such_switch=1)
    with pytest.raises(ValueError):
        config.set(split_scheme="fp8")


def test_sampling_refuses_cpu():
    """the sampler is a device kernel now (tests/test_inference_gpu.py holds the known answers); no CPU path"""
    from scanpaths_amd import hip
    from scanpaths_amd.models.sampling import Sampling
    s = Sampling(convLSTM_length=16, min_length=1)
    with pytest.raises(hip.HipError):
        s.generate_scanpath(torch.zeros(2, 3, 2, 2), torch.zeros(2, 16), torch.ones(2, 16), torch.zeros(2, 16, dtype=torch.long))


def test_synth_and_procedural_are_deterministic():
    from scanpaths_amd.procedural import procedural_state_dict
    from scanpaths_amd.spec import model_spec
    from scanpaths_amd.synth import make_batch
    a, b = make_batch("AiR", 3, 240, 320, 16, seed=5), make_batch("AiR", 3, 240, 320, 16, seed=5)
    assert all(torch.equal(a[k], b[k]) for k in a)
    assert a["scanpaths"].shape == (3, 16, 1201) and torch.all(a["scanpaths"].sum(-1) == 1)
    assert torch.all(a["attention_maps"].flatten(1).max(1).values == 1)
    assert torch.all(a["action_masks"].sum(1) >= a["duration_masks"].sum(1))
    r1 = make_batch("AiR", 3, 240, 320, 16, seed=5, rank=1)
    assert not torch.equal(a["images"], r1["images"])
    spec = {k: v for k, v in list(model_spec("OSIE", "resnet18").items())[:40]}
    s1, s2 = procedural_state_dict(spec, 3), procedural_state_dict(dict(reversed(list(spec.items()))), 3)
    assert all(torch.equal(s1[k], s2[k]) for k in spec)          # value depends on key/shape/seed, not on order


def test_init_weights_match_reference_distributions():
    """ScanpathModel.init_weights against (i) the analytic laws of the reference's initialisers
    (models/resnet.py:112-118: N(0, sqrt(2/(k*k*Cout))), BN 1/0; mmcv xavier_init(distribution='normal') for decoder convs:
    N(0, sqrt(2/(fan_in+fan_out))), bias 0; normal_init(std=0.01) for Linears -- AiR/models/baseline_attention.py:58-65,
    90-97,126-133,176-185,495-504) and (ii) the statistics of the REAL reference's freshly built models
    (tests/golden/init_stats.json from make_golden_init.py).  Tolerances are ~6 sigma of the sampling error of a std estimate
    (std * sqrt(1/(2n))) and of a mean estimate (std/sqrt(n)) for BOTH samples."""
    import json
    import math
    import os

    import torch

    from scanpaths_amd.models.scanpath_model import ScanpathModel
    with open(os.path.join(os.path.dirname(__file__), "golden", "init_stats.json")) as f:
        ref_all = json.load(f)
    for task in ("AiR", "OSIE", "COCO_Search18"):
        torch.manual_seed(1)
        m = ScanpathModel(task)
        ref = ref_all[task]
        names = [k for k, _ in m.named_parameters()]
        assert names == list(ref.keys())                       # same parameters, same registration order
        mods = dict(m.named_modules())
        for k, p in m.named_parameters():
            v = p.detach().double()
            n = v.numel()
            rmean, rstd, rmax, rn = ref[k]
            assert rn == n, k
            owner = mods[k.rpartition(".")[0]]
            is_bn = hasattr(owner, "running_mean")
            if k.endswith(".bias") or is_bn:
                want = 1.0 if (is_bn and k.endswith(".weight")) else 0.0      # constants in the reference too
                assert float((v - want).abs().max()) == 0.0 and rstd == 0.0 and abs(rmean - want) == 0.0, k
                continue
            if v.dim() == 4:
                co, ci, kh, kw = v.shape
                std = math.sqrt(2.0 / (kh * kw * co)) if k.startswith("resnet.") else math.sqrt(2.0 / ((ci + co) * kh * kw))
            else:
                std = 0.01
            tol_std = 6.0 * std * math.sqrt(1.0 / (2 * n)) + 1e-12
            tol_mean = 6.0 * std / math.sqrt(n)
            got_std = float(v.std()) if n > 1 else None
            if n >= 64:          # tiny tensors (1x1x3x3 attention convs) carry no distributional information
                assert abs(got_std - std) <= tol_std, (k, got_std, std)
                assert abs(rstd - std) <= tol_std, ("reference", k, rstd, std)
                assert abs(float(v.mean())) <= tol_mean and abs(rmean) <= tol_mean, k
            assert float(v.abs().max()) <= 7.0 * std and rmax <= 7.0 * std, k


def _json_field(g, k):
    import json
    return json.loads(bytes(g[k]).decode())


def test_dataset_host_side_matches_reference_datasets():
    """SURVEY §8 f4 remainder, host-side parts, against outputs of the REAL reference datasets (tests/golden/dataset_variants.npz,
    collate_f64.npz from tests/golden/make_golden_dataset.py): the oracle's numpy-1.x (float64 division) and blur_sigma variants;
    AiR_evaluation's per-question grouping / fixation vectors / performances (AiR/dataset/dataset.py:236-241, 258-343); the COCO
    detector-box attention map (COCO_Search18/dataset/dataset.py:150-160) and the OSIE / COCO record adapters."""
    import numpy as np
    import torch
    from helpers import GOLDEN
    from oracle import sampling_oracle as SO
    from scanpaths_amd import dataset as DS
    g = np.load(os.path.join(GOLDEN, "dataset_variants.npz"))
    g64 = np.load(os.path.join(GOLDEN, "collate_f64.npz"))
    recs = _json_field(g64, "records")
    keys = ("scanpaths", "durations", "action_masks", "duration_masks")
    for k, w in zip(keys, SO.collate_targets(recs, 16, (30, 40), f64_div=True)):
        assert np.array_equal(w, g64[k]), k
    for k, w in zip(keys, SO.collate_targets(recs, 16, (30, 40), f64_div=True, blur_sigma=1)):
        assert np.array_equal(w, g["blur_" + k]), k
    # evaluation grouping
    ev = _json_field(g, "eval_records")
    groups = DS.group_by_question(ev, resize=(240, 320))
    assert [s["question_id"] for s in groups] == _json_field(g, "eval_qids")
    assert [s["img_name"] for s in groups] == _json_field(g, "eval_imgs")
    assert [len(s["fix_vectors"]) for s in groups] == list(g["eval_count"])
    fvs = [f for s in groups for f in s["fix_vectors"]]
    assert [len(f) for f in fvs] == list(g["eval_len"])
    got = np.concatenate([np.stack([f["start_x"], f["start_y"], f["duration"]], 1) for f in fvs], 0)
    assert np.array_equal(got, g["eval_fix"])                                    # float32 divisions stored as f8: bit-exact
    assert [int(p) for s in groups for p in s["performances"]] == list(g["eval_perf"])
    batch = DS.collate_evaluation([dict(s, image=torch.zeros(3, 4, 4), attention_map=np.zeros((1, 30, 40), np.float32)) for s in groups],
                                  device=torch.device("cpu"))
    assert set(batch) == {"images", "fix_vectors", "attention_maps", "img_names", "performances", "question_ids"}
    # COCO: detector boxes -> attention map (identity resize in the golden run), record adapter incl. the clamp
    crecs, dets = _json_field(g, "coco_records"), _json_field(g, "coco_detector")
    idx = DS.index_detections(dets, 0.6)
    maps = np.stack([DS.detector_box_map(idx.get(r["name"].split(".")[0], []), r["task"], (30, 40))[None] for r in crecs])
    # (skimage's resize -- and its identity stand-in of the golden run -- returns float64: the reference normalises in float64)
    norm = DS.normalise_attention(torch.from_numpy(maps.astype(np.float64)), eps=1e-7).numpy()
    assert np.array_equal(norm, g["coco_attention_maps"])
    assert float(norm[3].max()) == 0.0 and float(norm[0].max()) > 0.99          # below-threshold detections are ignored
    assert [DS.COCO_OBJECT_NAMES.index(r["task"]) for r in crecs] == list(g["coco_tasks"])
    ad = DS._with_durations(crecs, (320, 512), clamp_to=(40 * (512 / 40), 30 * (320 / 30)))
    for k, w in zip(keys, SO.collate_targets(ad, 16, (30, 40), f64_div=True)):
        assert np.array_equal(w, g["coco_" + k]), k
    orecs = _json_field(g, "osie_records")
    for sigma, tag in ((None, "osie_"), (2, "osie_blur_")):
        for k, w in zip(keys, SO.collate_targets(DS._with_durations(orecs, (600, 800)), 16, (30, 40), f64_div=True, blur_sigma=sigma)):
            assert np.array_equal(w, g[tag + k]), (tag, k)


def test_kernel_sources_read_no_environment():
    """the product library takes no behaviour from the environment: no getenv in scanpaths_amd/csrc; the schedule /
    timing selectors are compiled only under SP_TIMING_VARIANTS"""
    import glob
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for fn in glob.glob(os.path.join(root, "scanpaths_amd", "csrc", "*.h*")):
        assert "getenv" not in open(fn).read(), fn
    import ctypes
    from scanpaths_amd import hip
    if os.path.exists(hip.LIB_PATH):
        lib = ctypes.CDLL(hip.LIB_PATH)
        lib.sp_set_tuning.argtypes = [ctypes.c_char_p, ctypes.c_int]
        assert lib.sp_timing_build() == 0
        assert lib.sp_set_tuning(b"h2_dbg", 1) == -1 and lib.sp_set_tuning(b"amax_reset", 1) == 0


def test_halo_block_swizzle_is_bank_conflict_free_for_every_base():
    """csrc/conv_f16x2.hip halo_swz: a ds_read_b128 is served in lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31} per 32-lane half;
    the fragment rows of the halo activation block start at an arbitrary pixel slot (image row, tap row and tap column shift it), so
    the chunk swizzle must give 16 distinct 16-byte bank slots for EVERY base.  Enumerated here for bases 0..599: the halo swizzle
    has no conflict, the ring's swizzle (made for bases that are multiples of 16) conflicts on 3/4 of the groups."""
    def banks(slot, chunk, swz):
        addr = slot * 128 + ((chunk ^ swz(slot)) * 16)
        return {((addr // 4) + k) % 64 for k in range(4)}
    g1 = [0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27]
    g2 = [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]
    groups = [g1, g2, [x + 32 for x in g1], [x + 32 for x in g2]]

    def conflicts(bases, swz):
        bad = 0
        for base in bases:
            for pl in range(2):
                for g in groups:
                    used = set()
                    for lane in g:
                        l16, g4 = lane & 15, lane >> 4
                        bs = banks(base + l16, (g4 >> 1) * 4 + pl * 2 + (g4 & 1), swz)
                        if used & bs:
                            bad += 1
                            break
                        used |= bs
        return bad
    halo = lambda s: ((s >> 1) & 3) << 1
    ring = lambda s: (s >> 1) & 7
    assert conflicts(range(600), halo) == 0
    assert conflicts(range(0, 256, 16), ring) == 0            # the ring's fragment rows start at multiples of 16
    assert conflicts(range(600), ring) > 3000


def test_split_counts_of_the_weight_gradient_kernels_follow_the_round_quantisation_rule():
    """Host logic, no GPU: the split-K count of a weight-gradient launch minimises  rounds x (K-tiles of a split + fixed cost)  with
    rounds = ceil(tiles x splits / resident workgroups).  Read back through the workspace queries (bytes = splits x Co x
    ldo x 4): the few-tile shapes of the encoder fill ONE round instead of one and a fraction, the h-gate shapes keep 8 / 4 splits."""
    from scanpaths_amd import hip
    L = hip.lib()

    def desc(M_img, Ho, Wo, Ci, Co, k):
        return hip.WgradDesc(M_img, Ho, Wo, Ci, Ci, Ho, Wo, Co, Co, k, k, 1, k // 2, 1, k * k * Ci, 0, 1.0, 1, 0, 0, 0)

    def splits(fn, d, *a):
        Co, ldo = d.Co, d.ldo
        return fn(ctypes.byref(d), *a) // (Co * ldo * 4)

    hw = L.sp_conv_wgrad_f16x2_workspace
    hw2 = L.sp_conv_wgrad_f16x2_multi_workspace
    # hw_kernel: tiles = ceil(Co / 256) x ceil(K / 128), one workgroup per CU
    assert splits(hw, desc(32, 80, 128, 64, 64, 3)) == 51          # 5 tiles: 255 workgroups = one round (was 64 -> 320 = two)
    assert splits(hw, desc(32, 80, 128, 128, 128, 3)) == 28        # 9 tiles: 252 workgroups
    assert splits(hw, desc(32, 40, 64, 256, 1024, 1)) == 32        # 8 tiles: 256 workgroups
    assert splits(hw, desc(32, 40, 64, 512, 2048, 3)) == 8         # the h-gate conv: 288 tiles, 9 rounds, as before
    # hw2_kernel: tiles = (Co / 256) x (K / 256); chains <= 20480 pixels, >= 64 K-tiles per split
    assert splits(hw2, desc(32, 40, 64, 512, 512, 3), 1) == 7      # 36 tiles: 252 workgroups (was 40 splits = 5.6 rounds)
    assert splits(hw2, desc(32, 40, 64, 256, 256, 3), 1) == 28     # 9 tiles
    assert splits(hw2, desc(32, 40, 64, 512, 2048, 3), 15) == 15 * 4    # the deferred h-gate launch (15 segments x 4 splits): unchanged
    for d in (desc(32, 40, 64, 512, 2048, 3), desc(2, 40, 64, 512, 2048, 3)):
        M = d.N_img * d.Ho * d.Wo
        s = splits(hw2, d, 1)
        assert -(-M // s) <= 20480 and s >= 1, (M, s)             # single-level accumulation bounds the chain
    # fp32 wgrad_kernel: two workgroups per CU
    fp = L.sp_conv_wgrad_workspace
    assert splits(fp, desc(32, 80, 128, 64, 64, 1)) == 512         # one tile: 512 workgroups (was 128)


def test_post_accumulate_hook_fires_for_an_undefined_gradient():
    """functional._take_grad_view writes a leaf parameter's gradient straight into FlatAdam's flat buffer and hands autograd
    None; the gradient-ready bookkeeping (bucketed all-reduce launch order, ddp.py) relies on the parameter's AccumulateGrad node still
    running and firing the post-accumulate hook with that undefined gradient.  This is observed torch behaviour (2.10), not a documented
    contract: pinned here so that a torch upgrade that drops it is noticed (FlatAdam.step() then still reports the parameters whose hook
    never fired -- correct results, no overlap)."""
    class Fn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, w):
            return x * 1.0

        @staticmethod
        def backward(ctx, g):
            return g, None          # "the gradient of w was written elsewhere"

    w = torch.nn.Parameter(torch.ones(3))
    x = torch.ones(3, requires_grad=True)
    fired = []
    w.register_post_accumulate_grad_hook(lambda p: fired.append(p.grad))
    Fn.apply(x, w).sum().backward()
    assert fired == [None] and w.grad is None


def test_direct_gradient_writes_step_aside_for_user_tensor_hooks():
    """a parameter with a tensor hook (p.register_hook: scaling, clipping, logging) must receive its gradient through autograd so
    that the hook sees it -- functional._grad_slot declines the in-place write for it."""
    from scanpaths_amd import functional as F

    class Slot:
        pass
    p = torch.nn.Parameter(torch.ones(4))
    p._sp_flat = Slot()
    assert F._grad_slot(p) is not None
    h = p.register_hook(lambda g: g * 2)
    assert F._grad_slot(p) is None
    h.remove()
    assert F._grad_slot(p) is not None


def _device_kernel_footprints(so_path):
    """{mangled kernel name: (vgprs + agprs, static LDS bytes)} read from the AMDGPU code objects embedded in the shared library"""
    import struct
    import subprocess
    import tempfile
    readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not os.path.exists(readelf):
        pytest.skip("llvm-readelf not found")
    data = open(so_path, "rb").read()
    out, pos = {}, 0
    while True:
        i = data.find(b"\x7fELF", pos)
        if i < 0:
            break
        pos = i + 4
        if data[i + 4] != 2 or struct.unpack_from("<H", data, i + 18)[0] != 224:      # ELF64, EM_AMDGPU
            continue
        shoff = struct.unpack_from("<Q", data, i + 40)[0]
        shentsize, shnum = struct.unpack_from("<HH", data, i + 58)
        with tempfile.NamedTemporaryFile(suffix=".elf") as f:
            f.write(data[i:i + shoff + shentsize * shnum])
            f.flush()
            notes = subprocess.run([readelf, "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            agpr = int(blk.split()[0])
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", blk).group(1))
            lds = int(re.search(r"\.group_segment_fixed_size:\s+(\d+)", blk).group(1))
            out[name] = (vgpr + agpr, lds)
    return out


def test_backward_chain_kernels_fit_beside_a_resident_data_gradient_workgroup():
    """The two-stream backward runs the small launches of the recurrence while the h-gate conv's data gradient holds
    every CU with 2 x 216 registers per SIMD and 148 KB of LDS; a launch starts beside it only with <= 80 VGPRs + AGPRs per wave and
    <= 11 KB of LDS per workgroup (tools/probes/coresidency_probe.hip, profiles/r05_coresidency_probe.log), otherwise it waits 0.6 ms
    for a tile to end.  The kernels of that chain are built to the limits: hold them there (register pressure moves with every edit)."""
    from scanpaths_amd import hip
    if not os.path.exists(hip.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    fp = _device_kernel_footprints(hip.LIB_PATH)
    gemm = [v for k, v in fp.items() if "h2_kernelILi1ELi3ELb1ELb0ELb1ELi0EE" in k]
    assert gemm and gemm[0][0] <= 216, gemm            # the resident: more registers here shrink what is left for everybody else
    chain = ("skinny_kernelILi0E", "skinny_kernelILi1E", "skinny_reduce_kernel", "listatt_bwd_kernel", "sempool_bwd_kernel",
             "mulrelu_bwd_kernel", "head_bwd_kernel", "drt_bwd_data_kernel", "drt_bwd_weight_kernel", "drt_slab_reduce_kernel",
             "drt_dcbsum_kernel", "sal_gather_bwd_kernel", "colsum_partial", "colsum_final", "sum_n_kernel", "add_kernel",
             "col2im1_kernel", "colamax_partial_kernel", "colamax_final_kernel", "split2_cols_kernel")
    for want in chain:
        hits = {k: v for k, v in fp.items() if want in k}
        assert hits, want
        for k, (regs, lds) in hits.items():
            assert regs <= 80 and lds <= 11 * 1024, (k, regs, lds)
