"""TEST INFRASTRUCTURE ONLY -- procedural (seeded, key-addressed) weights.

The reference ships no checkpoints and no golden vectors, and real state_dicts are 20-70 M
parameters -- too large to commit.  Instead every parameter/buffer is a deterministic function of its
state_dict key and shape, so the golden generator (which loads them into the *reference* modules) and the
tests (which load them into the oracle restatement and into the HIP modules) regenerate identical weights
from the committed key/shape lists in tests/golden/ref_state_keys.json.
"""
import zlib

import torch


def tensor_for(key: str, shape, dtype=torch.float32, salt: int = 0) -> torch.Tensor:
    shape = tuple(shape)
    gen = torch.Generator().manual_seed((zlib.crc32(key.encode()) + 7919 * salt) % (2 ** 31))
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked" or dtype in (torch.int64, "int64"):
        return torch.zeros(shape, dtype=torch.int64)
    if leaf == "running_var":
        return torch.rand(shape, generator=gen) + 0.5
    if leaf == "running_mean":
        return 0.1 * torch.randn(shape, generator=gen)
    if leaf == "weight_g":
        return torch.rand(shape, generator=gen) * 0.5 + 0.5
    if leaf in ("weight_u",) or (leaf == "weight_v" and len(shape) == 1):
        v = torch.randn(shape, generator=gen)
        return v / v.norm()
    if leaf == "alpha" or len(shape) == 0:
        return torch.ones(shape)
    if len(shape) == 1:
        if leaf == "bias":
            return 0.1 * torch.randn(shape, generator=gen)
        return 1.0 + 0.1 * torch.randn(shape, generator=gen)        # norm scales
    fan_in = 1
    for s in shape[1:]:
        fan_in *= s
    w = torch.randn(shape, generator=gen) / (fan_in ** 0.5)
    if "embed" in key and len(shape) == 2 and leaf == "weight" and "proj" not in key:
        w = torch.randn(shape, generator=gen) * (shape[1] ** -0.5)
        w[0] = 0.0                                                   # padding_idx row (common_layers.py:63-68)
    return w


def state_dict_for(keys_and_shapes, prefix: str = "", salt: int = 0):
    """keys_and_shapes: list of [key, shape, dtype-string]."""
    sd = {}
    for key, shape, dt in keys_and_shapes:
        sd[key] = tensor_for(prefix + key, shape, torch.int64 if dt == "torch.int64" else torch.float32, salt)
    return sd
