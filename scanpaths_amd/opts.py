"""CLI flag surface of the reference's ``opts.py`` (AiR/opts.py:6-50; OSIE / COCO variants differ in --att_dir vs
--detector_dir/--detector_threshold and the default weight decay).  Same names, types and defaults, including the
reference's ``type=bool`` / ``type=list`` quirks, so launch scripts keep working.  The optional ``--cfg`` YAML
(yacs CfgNode in the reference, utils/config.py) is read with plain ``yaml`` here: precedence YAML < --set_cfgs < CLI."""
import argparse


def build_parser(task="AiR"):
    p = argparse.ArgumentParser(description="Scanpath prediction for images")
    a = p.add_argument
    a("--mode", type=str, default="train")
    a("--img_dir", type=str, default="./data/stimuli")
    a("--fix_dir", type=str, default="./data/fixations")
    if task == "COCO_Search18":
        a("--detector_dir", type=str, default="./data/detectors")
        a("--detector_threshold", type=float, default=0.8)
    else:
        a("--att_dir", type=str, default="./data/attention_reasoning")
    a("--width", type=int, default=320)
    a("--height", type=int, default=240)
    a("--map_width", type=int, default=40)
    a("--map_height", type=int, default=30)
    a("--blur_sigma", type=float, default=None)
    a("--clip", type=float, default=12.5)
    a("--batch", type=int, default=16)
    a("--epoch", type=int, default=10)
    a("--warmup_epoch", type=int, default=1)
    a("--start_rl_epoch", type=int, default=5)
    a("--rl_sample_number", type=int, default=5)
    a("--seed", type=int, default=0)
    a("--lr", type=float, default=1e-4)
    a("--rl_lr_initial_decay", type=float, default=0.5)
    a("--weight_decay", type=float, default=5e-5 if task == "AiR" else 5e-4)
    a("--gpu_ids", type=list, default=[0, 1])
    a("--log_root", type=str, default="./assets/")
    a("--resume_dir", type=str, default="")
    a("--center_bias", type=bool, default=True)
    a("--lambda_1", type=float, default=1)
    if task == "AiR":
        a("--lambda_5", type=float, default=-2.0)
    a("--eval_repeat_num", type=int, default=10)
    a("--min_length", type=int, default=1)
    a("--max_length", type=int, default=16)
    a("--ablate_attention_info", type=bool, default=False)
    a("--supervised_save", type=bool, default=True)
    a("--cfg", type=str, default=None)
    a("--set_cfgs", dest="set_cfgs", default=[], nargs="+")
    return p


def parse_opt(task="AiR", argv=None):
    parser = build_parser(task)
    args = parser.parse_args(argv)
    if args.cfg is not None or args.set_cfgs:
        import yaml
        cfg = {}
        if args.cfg is not None:
            with open(args.cfg) as f:
                cfg = yaml.safe_load(f) or {}
            if isinstance(cfg, dict) and "_BASE_" in cfg:
                # the reference's loader (utils/config.py:15-144 load_yaml_with_base) merges the named base file first; that config code
                # is not rebuilt here -- refuse instead of silently training with the base's values missing
                raise ValueError(f"{args.cfg}: '_BASE_' inheritance is not supported by scanpaths_amd.opts; flatten the file "
                                 f"(base {cfg['_BASE_']!r} merged first, then this file's keys)")
        it = iter(args.set_cfgs)
        for k, v in zip(it, it):
            cfg[k] = yaml.safe_load(v)
        for k, v in cfg.items():
            if not hasattr(args, k):
                print("Warning: key %s not in args" % k)
            setattr(args, k, v)
        args = parser.parse_args(argv, namespace=args)
    return args
