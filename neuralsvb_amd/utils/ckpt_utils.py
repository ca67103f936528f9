"""Checkpoint discovery and partial loading -- the reference's on-disk layout (utils/ckpt_utils.py:8-68):
`<work_dir>/model_ckpt_steps_<N>.ckpt` holding {'state_dict': {<child name>: state_dict}, 'optimizer_states', ...};
older flat layouts ("model.a.b") are accepted too."""
import glob
import logging
import os
import re

import torch


def get_all_ckpts(work_dir, steps=None):
    pat = f"{work_dir}/model_ckpt_steps_{'*' if steps is None else steps}.ckpt"
    return sorted(glob.glob(pat), key=lambda p: -int(re.findall(r".*steps\_(\d+)\.ckpt", p)[0]))


def get_last_checkpoint(work_dir, steps=None):
    paths = get_all_ckpts(work_dir, steps)
    if not paths:
        return None, None
    logging.info(f"load module from checkpoint: {paths[0]}")
    return torch.load(paths[0], map_location="cpu", weights_only=False), paths[0]


def extract_state_dict(checkpoint, model_name):
    sd = checkpoint["state_dict"]
    if any("." in k for k in sd.keys()):                      # flat layout
        return {k[len(model_name) + 1:]: v for k, v in sd.items() if k.startswith(f"{model_name}.")}
    if "." not in model_name:
        return sd[model_name]
    head, rest = model_name.split(".", 1)
    return {k[len(rest) + 1:]: v for k, v in sd[head].items() if k.startswith(f"{rest}.")}


def load_ckpt(cur_model, ckpt_base_dir, model_name="model", force=True, strict=True):
    if os.path.isfile(ckpt_base_dir):
        base_dir, ckpt_path = os.path.dirname(ckpt_base_dir), ckpt_base_dir
        checkpoint = torch.load(ckpt_base_dir, map_location="cpu", weights_only=False)
    else:
        base_dir = ckpt_base_dir
        checkpoint, ckpt_path = get_last_checkpoint(ckpt_base_dir)
    if checkpoint is None:
        msg = f"| ckpt not found in {base_dir}."
        if force:
            raise FileNotFoundError(msg)
        print(msg)
        return
    sd = extract_state_dict(checkpoint, model_name)
    if not strict:
        cur = cur_model.state_dict()
        for key in [k for k, v in sd.items() if k in cur and cur[k].shape != v.shape]:
            print("| Unmatched keys: ", key, cur[key].shape, sd[key].shape)
            del sd[key]
    cur_model.load_state_dict(sd, strict=strict)
    print(f"| load '{model_name}' from '{ckpt_path}'.")
