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


def test_fused_tower_node_equals_per_op_nodes(dev):
    """SF.critic_tower (a whole tower + score layer as ONE autograd node, the per-op bodies replayed on a private tape) must give
    bit for bit what the per-op autograd nodes give: scores, input gradients and every parameter gradient -- with Dropout2d ON
    (same draw order from the same generator state) and with the critic's parameters frozen (the generator's pass)."""
    from neuralsvb_amd import functional as SF
    from neuralsvb_amd.modules.mel_disc import Discriminator
    torch.manual_seed(0)
    disc = Discriminator(time_lengths=[32, 64], freq_length=80, hidden_size=16, kernel=(3, 3), cond_size=0,
                         norm_type="in", reduction="stack").to(dev)
    disc.train()
    g = torch.Generator().manual_seed(2)
    xs0 = [torch.randn(2, 90, 80, generator=g).to(dev) for _ in range(2)]
    starts = [[[5, 5], [11, 11]], [[40, 40], [0, 0]]]
    res = {}
    for frozen in (False, True):
        for p in disc.parameters():
            p.requires_grad_(not frozen)
        for fused in (True, False):
            SF.FUSE_CRITIC_TOWER = fused
            try:
                torch.manual_seed(7)          # the Dropout2d draws
                xs = [x.clone().requires_grad_(True) for x in xs0]
                outs = disc.forward_many([(x, [list(s) for s in st], None, 90) for x, st in zip(xs, starts)], want_fmaps=False)
                loss = sum(((o["y"] - 1) ** 2).mean() * (i + 1) for i, o in enumerate(outs))
                wrt = xs + ([] if frozen else list(disc.parameters()))
                res[(frozen, fused)] = ([o["y"].detach().clone() for o in outs], torch.autograd.grad(loss, wrt))
            finally:
                SF.FUSE_CRITIC_TOWER = True
        (ya, ga), (yb, gb) = res[(frozen, True)], res[(frozen, False)]
        for a, b in zip(ya, yb):
            assert torch.equal(a, b)
        assert len(ga) == len(gb)
        for a, b in zip(ga, gb):
            assert torch.equal(a, b)


def test_tower_executor_equals_python_issue_order(dev):
    """The tower's C executor (csrc/critic_tower.hip: one ABI call per direction) against the per-op Python bodies on the same
    tape, bf16x3: scores and input gradients bit for bit; parameter gradients accumulated into existing `.grad` buffers (the
    executor's weight gradients: partials reduced in-stream, gathered into the 3x3 kernels' buffers) bit for bit as well; with the
    critic frozen (the generator's pass: data gradients only); and the fall-back when there are no `.grad` buffers to
    accumulate into (torch.autograd.grad): the executor's forward tape must serve the per-op backward."""
    from neuralsvb_amd import functional as SF
    from neuralsvb_amd.modules.mel_disc import Discriminator
    torch.manual_seed(0)
    disc = Discriminator(time_lengths=[32], freq_length=80, hidden_size=16, kernel=(3, 3), cond_size=0,
                         norm_type="in", reduction="stack").to(dev)
    disc.train()
    g = torch.Generator().manual_seed(5)
    xs0 = [torch.randn(2, 60, 80, generator=g).to(dev) for _ in range(2)]
    starts = [[[5, 5]], [[22, 22]]]
    SF.set_precision("bf16x3")
    try:
        res = {}
        for mode in ("grad_buffers", "frozen", "autograd_grad"):
            for p in disc.parameters():
                p.requires_grad_(mode != "frozen")
            for ex in (True, False):
                SF.TOWER_EXECUTOR = ex
                torch.manual_seed(7)          # the Dropout2d draws
                xs = [x.clone().requires_grad_(True) for x in xs0]
                for p in disc.parameters():
                    p.grad = torch.full_like(p, 0.125) if mode == "grad_buffers" else None
                outs = disc.forward_many([(x, [list(s) for s in st], None, 60) for x, st in zip(xs, starts)], want_fmaps=False)
                loss = sum(((o["y"] - 1) ** 2).mean() * (i + 1) for i, o in enumerate(outs))
                if mode == "autograd_grad":
                    grads = torch.autograd.grad(loss, xs + list(disc.parameters()))
                else:
                    loss.backward()
                    grads = [x.grad for x in xs] + ([p.grad.clone() for p in disc.parameters()] if mode == "grad_buffers" else [])
                res[(mode, ex)] = ([o["y"].detach().clone() for o in outs], grads)
            (ya, ga), (yb, gb) = res[(mode, True)], res[(mode, False)]
            for a, b in zip(ya, yb):
                assert torch.equal(a, b), mode
            assert len(ga) == len(gb)
            for i, (a, b) in enumerate(zip(ga, gb)):
                assert torch.equal(a, b), (mode, i, (a - b).abs().max().item())
            if mode == "grad_buffers":
                assert any((gr - 0.125).abs().max().item() > 1e-3 for gr in ga[2:])     # the buffers did receive gradients
    finally:
        SF.TOWER_EXECUTOR = True
        SF.set_precision("fp32")
        for p in disc.parameters():
            p.grad = None


def test_critic_general_shapes_match_stock_torch(dev):
    """Reference-legal critic configurations outside the fast path (multi_window_disc.py:14-65): a number of mel bins whose
    halvings turn odd (60 -> 30 -> 15 -> 8) and a 5x5 kernel.  Both used to raise inside SF.critic_block; they now take the
    general conv + Dropout2d factor + stock norm.  Compared with the same towers applied through stock torch ops (eval)."""
    from neuralsvb_amd.modules.mel_disc import Discriminator
    for freq, kernel in ((60, (3, 3)), (80, (5, 5)), (60, (5, 5))):
        torch.manual_seed(3)
        disc = Discriminator(time_lengths=[32, 64], freq_length=freq, hidden_size=16, kernel=kernel, cond_size=0,
                             norm_type="in", reduction="stack").eval()
        g = torch.Generator().manual_seed(4)
        x = torch.randn(2, 70, freq, generator=g)
        starts = [[3, 3], [6, 6]]
        ref_scores, ref_fmaps = [], []
        with torch.no_grad():
            for tower, wl, st in zip(disc.discriminator.conv_layers, [32, 64], starts):
                h = x[:, None, st[0]:st[0] + wl]
                for blk in tower.model:
                    h = blk(h)
                    ref_fmaps.append(h)
                ref_scores.append(tower.adv_layer(h.flatten(1)))
        ref_y = torch.stack(ref_scores, -1)
        d2 = disc.to(dev)
        xd = x.clone().to(dev).requires_grad_(True)
        o = d2(xd, None, start_frames_wins=[list(s) for s in starts], longest=70)
        assert (o["y"].detach().cpu() - ref_y).abs().max().item() < 5e-5 * max(1.0, ref_y.abs().max().item()), (freq, kernel)
        for a, b in zip(o["h"], ref_fmaps):
            assert a.shape == b.shape and (a.detach().cpu() - b).abs().max().item() < 5e-5 * max(1.0, b.abs().max().item())
        many = d2.forward_many([(xd, [list(s) for s in starts], None, 70)], want_fmaps=False)
        assert (many[0]["y"].detach().cpu() - ref_y).abs().max().item() < 5e-5 * max(1.0, ref_y.abs().max().item())
        # gradients through the general path
        xr = x.clone().requires_grad_(True)
        disc_cpu = disc.cpu()
        hs = []
        for tower, wl, st in zip(disc_cpu.discriminator.conv_layers, [32, 64], starts):
            h = xr[:, None, st[0]:st[0] + wl]
            for blk in tower.model:
                h = blk(h)
            hs.append(tower.adv_layer(h.flatten(1)))
        (torch.stack(hs, -1) ** 2).sum().backward()
        gref = [p.grad.clone() for p in disc_cpu.parameters()]
        for p in disc_cpu.parameters():
            p.grad = None
        d2 = disc_cpu.to(dev)
        xd = x.clone().to(dev).requires_grad_(True)
        (d2(xd, None, start_frames_wins=[list(s) for s in starts], longest=70)["y"] ** 2).sum().backward()
        assert (xd.grad.cpu() - xr.grad).abs().max().item() < 1e-4 * max(1.0, xr.grad.abs().max().item())
        for p, gr in zip(d2.parameters(), gref):
            assert (p.grad.cpu() - gr).abs().max().item() < 2e-4 * max(1.0, gr.abs().max().item())
