"""Global `hparams` dict + YAML config chaining -- the reference's config surface.

Mirror of reference utils/hparams.py:25-128: same CLI flags (--config --exp_name --hparams --infer --validate
--reset --remove --debug), same `base_config` semantics (depth-first, later files override earlier, a file is
loaded once, relative entries resolve against the including file, other entries against the cwd), same
`--hparams "a=1,b.c=2,d=[1 1 1]"` override syntax, same `checkpoints/<exp_name>/config.yaml` save/merge rules.
"""
import argparse
import ast
import os
import shutil

import yaml

hparams = {}
_printed = False


def _merge(dst, src):
    for k, v in src.items():
        if isinstance(v, dict) and isinstance(dst.get(k), dict):
            _merge(dst[k], v)
        else:
            dst[k] = v


def load_config_chain(path, _seen=None, _chain=None):
    """Resolve one YAML file and its `base_config` ancestors.  Returns (dict, [files in load order])."""
    seen = set() if _seen is None else _seen
    chain = [] if _chain is None else _chain
    if not os.path.exists(path):
        return {}, chain
    with open(path) as f:
        cfg = yaml.safe_load(f) or {}
    seen.add(path)
    out = {}
    bases = cfg.get("base_config", [])
    if not isinstance(bases, list):
        bases = [bases]
        cfg["base_config"] = bases
    for b in bases:
        if b.startswith("."):
            b = os.path.normpath(os.path.join(os.path.dirname(path), b))
        if b not in seen:
            sub, _ = load_config_chain(b, seen, chain)
            _merge(out, sub)
    _merge(out, cfg)
    chain.append(path)
    return out, chain


def _apply_overrides(cfg, spec):
    for item in spec.split(","):
        if not item:
            continue
        key, val = item.split("=", 1)
        val = val.strip("'\" ")
        node = cfg
        parts = key.split(".")
        for p in parts[:-1]:
            node = node[p]
        leaf = parts[-1]
        old = node.get(leaf)
        if val in ("True", "False") or isinstance(old, (bool, list, dict)):
            if isinstance(old, list):
                val = val.replace(" ", ",")
            node[leaf] = ast.literal_eval(val)
        elif old is None:
            try:
                node[leaf] = ast.literal_eval(val)
            except (ValueError, SyntaxError):
                node[leaf] = val
        else:
            node[leaf] = type(old)(val)


def set_hparams(config="", exp_name="", hparams_str="", print_hparams=True, global_hparams=True):
    global _printed
    if config == "" and exp_name == "":
        ap = argparse.ArgumentParser(description="")
        ap.add_argument("--config", type=str, default="configs/config_base.yaml")
        ap.add_argument("--exp_name", type=str, default="")
        ap.add_argument("--hparams", type=str, default="")
        ap.add_argument("--infer", action="store_true")
        ap.add_argument("--validate", action="store_true")
        ap.add_argument("--reset", action="store_true")
        ap.add_argument("--remove", action="store_true")
        ap.add_argument("--debug", action="store_true")
        args, _ = ap.parse_known_args()
    else:
        args = argparse.Namespace(config=config, exp_name=exp_name, hparams=hparams_str, infer=False, validate=False,
                                  reset=False, remove=False, debug=False)
    assert args.config != "" or args.exp_name != ""
    work_dir = f"checkpoints/{args.exp_name}" if args.exp_name else ""
    ckpt_cfg = f"{work_dir}/config.yaml"
    saved = {}
    if work_dir and os.path.exists(ckpt_cfg):
        with open(ckpt_cfg) as f:
            saved = yaml.safe_load(f) or {}
    cfg, chain = ({}, [])
    if args.config:
        cfg, chain = load_config_chain(args.config)
    if not args.reset:
        cfg.update(saved)
    cfg["work_dir"] = work_dir
    if args.hparams:
        _apply_overrides(cfg, args.hparams)
    if work_dir and args.remove:
        if input("REMOVE old checkpoint? Y/N [Default: N]: ").lower() == "y":
            shutil.rmtree(work_dir, ignore_errors=True)
    if work_dir and (not os.path.exists(ckpt_cfg) or args.reset) and not args.infer:
        os.makedirs(work_dir, exist_ok=True)
        with open(ckpt_cfg, "w") as f:
            yaml.safe_dump(cfg, f)
    cfg["infer"], cfg["debug"], cfg["validate"], cfg["exp_name"] = args.infer, args.debug, args.validate, args.exp_name
    if global_hparams:
        hparams.clear()
        hparams.update(cfg)
    if print_hparams and global_hparams and not _printed:
        print("| Hparams chains: ", chain)
        print("| Hparams: ")
        for i, (k, v) in enumerate(sorted(cfg.items())):
            print(f"\033[;33;m{k}\033[0m: {v}, ", end="\n" if i % 5 == 4 else "")
        print("")
        _printed = True
    return cfg
