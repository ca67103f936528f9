"""Mel critic (Discriminator) on the HIP conv path vs the golden vectors produced by the unmodified reference."""
import json
import os

import numpy as np
import torch

from oracle import procedural

G = os.path.join(os.path.dirname(__file__), "golden")
KEYS = json.load(open(os.path.join(G, "ref_state_keys.json")))


def test_mel_discriminator_matches_reference_golden(dev):
    from neuralsvb_amd.modules.mel_disc import Discriminator
    d = np.load(os.path.join(G, "mel_disc.npz"))
    disc = Discriminator(time_lengths=[32, 64, 128], freq_length=80, hidden_size=128, kernel=(3, 3), cond_size=0,
                         norm_type="in", reduction="stack")
    ref_keys = sorted((k, tuple(s)) for k, s, _ in KEYS["Discriminator"])
    assert sorted((k, tuple(v.shape)) for k, v in disc.state_dict().items()) == ref_keys
    disc.load_state_dict(procedural.state_dict_for(KEYS["Discriminator"], prefix="mel_disc."), strict=True)
    disc = disc.to(dev).eval()
    with torch.no_grad():
        o = disc(torch.from_numpy(d["x"]).to(dev), None, start_frames_wins=[list(s) for s in d["starts"]])
    assert o["y"].shape == d["y"].shape == (2, 1, 3)
    assert np.abs(o["y"].cpu().numpy() - d["y"]).max() < 3e-5
    from tests.golden.make_golden import fmap_stats
    for i, h in enumerate(o["h"]):
        np.testing.assert_allclose(fmap_stats(h.cpu()), d["h_stats"][i], atol=3e-5, rtol=2e-4)


def test_stacked_critic_calls_equal_separate_calls(dev):
    """Discriminator.forward_many (several calls as one stacked pass per tower, each with its own window starts) must
    return, per call, what separate forward() calls return -- outputs and gradients (Dropout2d off)."""
    from neuralsvb_amd.modules.mel_disc import Discriminator
    torch.manual_seed(0)
    disc = Discriminator(time_lengths=[32, 64], freq_length=80, hidden_size=16, kernel=(3, 3), cond_size=0,
                         norm_type="in", reduction="stack").to(dev)
    for m in disc.modules():
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    g = torch.Generator().manual_seed(1)
    xs = [torch.randn(2, 90, 80, generator=g).to(dev).requires_grad_(True) for _ in range(3)]
    starts = [[[5, 5], [11, 11]], [[40, 40], [0, 0]], [[58, 58], [26, 26]]]
    sep = [disc(x, None, start_frames_wins=[list(s) for s in st], longest=90)["y"] for x, st in zip(xs, starts)]
    loss_sep = sum(((y - 1) ** 2).mean() * (i + 1) for i, y in enumerate(sep))
    grads_sep = torch.autograd.grad(loss_sep, xs + list(disc.parameters()))
    many = disc.forward_many([(x, [list(s) for s in st], None, 90) for x, st in zip(xs, starts)], want_fmaps=False)
    assert many[0]["h"] == []                  # the towers' blocks handed each other the conv layout: no feature maps
    loss_many = sum(((o["y"] - 1) ** 2).mean() * (i + 1) for i, o in enumerate(many))
    grads_many = torch.autograd.grad(loss_many, xs + list(disc.parameters()))
    for a, b in zip(sep, many):
        assert (a - b["y"]).abs().max().item() < 5e-5 * max(1.0, a.abs().max().item())
    for a, b in zip(grads_sep, grads_many):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, a.abs().max().item())
