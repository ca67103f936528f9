"""Step-level parity through the unmodified driver path: YAML config -> set_hparams -> SVBVAEMleTask -> Trainer.setup ->
dataset/collater -> run_training_batch (3-optimizer step) on the HIP kernels, against the CPU oracle port of the
same step (oracle/train_step_ref.py) fed with the same weights, batch and random draws.

Small dims so the emulator variant fits the CPU suite; the gpu variant runs the same thing on the MI355X.
"""
import os

import numpy as np
import pytest
import torch

from oracle import frontend as ofe
from oracle.train_step_ref import CpuStep

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ("hidden_size=32,fvae_enc_dec_hidden=32,latent_size=16,fvae_enc_n_layers=2,fvae_dec_n_layers=2,"
         "mel_disc_hidden_size=16,max_sentences=2,ds_workers=0,num_sanity_val_steps=0,endless_ds=False,"
         "audio_sample_rate=24000,fmax=12000,warmup_updates=4")


def _oracle_mel_fn(hp):
    def fn(wavs):
        return np.stack([ofe.wav2mel_offline(w, hp["fft_size"], hp["hop_size"], hp["win_size"], hp["audio_num_mel_bins"],
                                             hp["fmin"], hp["fmax"], hp["audio_sample_rate"])[1] for w in wavs])
    return fn


def _setup(tmp_path, dev, extra="", n_clips=2, seconds=0.71):
    from neuralsvb_amd.utils.hparams import set_hparams, hparams
    from neuralsvb_amd.utils import synth
    set_hparams(config=os.path.join(ROOT, "egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml"), exp_name="",
                hparams_str=SMALL + extra, print_hparams=False)
    hparams["binary_data_dir"] = str(tmp_path / "bin")
    hparams["pretrain_asr_ckpt"] = str(tmp_path / "asr")
    hparams["work_dir"] = str(tmp_path / "ckpt")
    torch.manual_seed(0)
    synth.write_binary_dataset(hparams["binary_data_dir"], hparams, _oracle_mel_fn(hparams), n_train=n_clips, n_valid=1,
                               seconds=seconds)
    synth.write_fake_asr_ckpt(hparams["pretrain_asr_ckpt"], 70, hparams)
    from neuralsvb_amd.tasks.svb_vae_task import SVBVAEMleTask
    from neuralsvb_amd.utils.trainer import Trainer, move_to_device
    trainer = Trainer(work_dir=hparams["work_dir"], num_sanity_val_steps=0, num_ckpt_keep=2)
    trainer.on_gpu = dev.type == "cuda"
    trainer.world_size, trainer.use_ddp = 1, False
    torch.manual_seed(1)
    task = trainer.setup(SVBVAEMleTask())
    for m in task.mel_disc.modules():          # Dropout2d draws cannot be matched across devices: disable for parity
        if isinstance(m, torch.nn.Dropout2d):
            m.p = 0.0
    task.train()
    loader = task.build_dataloader(task.dataset_cls("train", False), False, hparams["max_tokens"], n_clips)
    batch = move_to_device(next(iter(loader)), dev)
    return task, trainer, batch, hparams


def test_phase2_training_step_matches_cpu_oracle(dev, tmp_path):
    task, trainer, batch, hp = _setup(tmp_path, dev)
    assert batch["mels"].shape == (2, 132, 80) and batch["pitch"].dtype == torch.int64
    B, L = 2, hp["latent_size"]
    msd = {k: v.detach().cpu().clone() for k, v in task.model.state_dict().items()}
    dsd = {k: v.detach().cpu().clone() for k, v in task.mel_disc.state_dict().items()}
    oracle = CpuStep(msd, dsd, hp)
    cpu_batch = {k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in batch.items()}

    g = torch.Generator().manual_seed(3)
    for step in (1, 2):
        eps_a, eps_p = torch.randn(B, L, 1, generator=g), torch.randn(B, L, 1, generator=g)
        # record the discriminator window starts of the HIP step (drawn up front by task.begin_step)
        starts = []
        orig_begin = task.begin_step

        def rec(*a, _o=orig_begin, **kw):
            r = _o(*a, **kw)
            starts.extend([list(s) for s in d["starts"]] for d in r["disc"])
            return r
        task.begin_step = rec
        orig_run = task.run_model
        task.run_model = lambda *a, **k: orig_run(*a, eps_a2a=eps_a.to(dev), eps_p2p=eps_p.to(dev), **k)
        np.random.seed(100 + step)
        task.global_step = trainer.global_step = step
        pbar, _ = trainer.run_training_batch(0, batch)
        task.begin_step, task.run_model = orig_begin, orig_run
        np.random.seed(100 + step)
        spk_idx = np.random.randint(1, 5)
        assert len(starts) == 6
        sg = {"a2a": starts[0], "p2p": starts[1]}
        sd = {"a2a": {"real": starts[2], "fake": starts[3]}, "p2p": {"real": starts[4], "fake": starts[5]}}
        ref = oracle.step(cpu_batch, spk_idx, eps_a, eps_p, sg, sd, global_step=step)
        for k in ("a2a_kl", "p2p_kl", "ssima2a", "l1a2a", "ssimp2p", "l1p2p", "a2a_a", "p2p_a", "a2a_r", "a2a_f", "p2p_r",
                  "p2p_f"):
            got = float(pbar[k])
            assert abs(got - ref[k]) <= 2e-4 * max(1.0, abs(ref[k])), (step, k, got, ref[k])
    # after two optimizer steps the weights must still agree (AdamW + clipping + schedules)
    # (Adam normalises every gradient to ~+-lr in its first steps, so parameters whose true gradient is zero -- e.g.
    #  a conv bias in front of a train-mode BatchNorm -- move by +-lr with a sign decided by rounding noise; those
    #  are excluded by counting, not by name.)
    n_tot = n_bad = 0
    pairs = [(v, oracle.sd[k]) for k, v in task.model.state_dict().items()
             if v.is_floating_point() and "running" not in k and not k.startswith("vc_asr")]
    pairs += [(v, oracle.dsd[k]) for k, v in task.mel_disc.state_dict().items() if v.is_floating_point()]
    for v, r in pairs:
        d = (v.detach().cpu() - r.detach()).abs()
        n_tot += d.numel()
        n_bad += int((d > 1e-3).sum())
    assert n_bad / n_tot < 5e-3, (n_bad, n_tot)

    # checkpoint layout round trip (reference utils/trainer.py:397-436)
    trainer.save_checkpoint(epoch=0)
    from neuralsvb_amd.utils.ckpt_utils import get_last_checkpoint
    ck, path = get_last_checkpoint(hp["work_dir"])
    assert os.path.basename(path) == f"model_ckpt_steps_{trainer.global_step}.ckpt"
    assert set(ck.keys()) == {"epoch", "global_step", "checkpoint_callback_best", "optimizer_states", "state_dict"}
    assert set(ck["state_dict"].keys()) == {"model", "mel_disc"} and len(ck["optimizer_states"]) == 3
    assert "vae_model.encoder.wn.in_layers.0.weight_g" in ck["state_dict"]["model"]


@pytest.mark.gpu
def test_ppg_prefetch_graph_replay_equals_eager_launches(gpu_only, tmp_path):
    """`ppg_graph: true`: the look-ahead run of the frozen PPG encoder replayed from a captured hipGraph (one per input shape,
    captured at the second sight) must hand the following step bit-for-bit what the eager launches hand it -- six steps over two
    alternating batches, identical losses and weights, and the graph must actually have been replayed."""
    from neuralsvb_amd.modules import svb_vae
    dev = gpu_only
    res = {}
    for mode in ("eager", "graph"):
        task, trainer, batch, hp = _setup(tmp_path / mode, dev)
        svb_vae.PPG_GRAPH = mode == "graph"
        try:
            b2 = {k: (v.flip(0).contiguous() if isinstance(v, torch.Tensor) and v.dim() > 0 else v) for k, v in batch.items()}
            seq = [batch, b2] * 4
            logs = []
            for step in range(1, 7):
                np.random.seed(400 + step)
                torch.manual_seed(400 + step)
                task.global_step = trainer.global_step = step
                pbar, _ = trainer.run_training_batch(0, seq[step - 1], next_batch=seq[step])
                logs.append({k: float(v) for k, v in pbar.items() if isinstance(v, torch.Tensor)})
            trainer._join_critic_stream()
            torch.cuda.synchronize()
            st = task.model.__dict__.get("_ppg_graphs")
            if mode == "graph":
                assert st is not None and len(st["graphs"]) == 1 and not st["off"], st
            res[mode] = (logs, {k: v.detach().clone() for k, v in task.state_dict().items() if v.is_floating_point()})
        finally:
            svb_vae.PPG_GRAPH = False
    for a, b in zip(res["eager"][0], res["graph"][0]):
        assert a == b, (a, b)
    for k, v in res["eager"][1].items():
        assert torch.equal(v, res["graph"][1][k]), k


@pytest.mark.gpu
def test_prefetched_ppg_encoder_gives_the_same_steps(gpu_only, tmp_path):
    """Trainer.run_training_batch(..., next_batch=...) runs the frozen PPG encoder of the following batch one step early, on
    its side stream beside the current backward, and copies that batch to the device early; the following step must consume
    exactly that result.  Four steps over two alternating batches with and without the look-ahead: identical losses, identical
    weights (same kernels on the same inputs -- only their place on the time line differs)."""
    dev = gpu_only
    res = {}
    for mode in ("plain", "lookahead"):
        task, trainer, batch, hp = _setup(tmp_path / mode, dev)
        b2 = {k: (v.flip(0).contiguous() if isinstance(v, torch.Tensor) and v.dim() > 0 else v) for k, v in batch.items()}
        seq = [batch, b2, batch, b2]
        logs = []
        for step in range(1, 5):
            np.random.seed(300 + step)
            torch.manual_seed(300 + step)
            task.global_step = trainer.global_step = step
            nxt = seq[step] if (mode == "lookahead" and step < 4) else None
            pbar, _ = trainer.run_training_batch(0, seq[step - 1], next_batch=nxt)
            logs.append({k: float(v) for k, v in pbar.items() if isinstance(v, torch.Tensor)})
        if mode == "lookahead":
            assert not task.model.__dict__.get("_content_cache")          # every prefetched result was consumed
        res[mode] = (logs, {k: v.detach().clone() for k, v in task.state_dict().items() if v.is_floating_point()})
    for a, b in zip(res["plain"][0], res["lookahead"][0]):
        assert a == b
    for k, v in res["plain"][1].items():
        assert torch.equal(v, res["lookahead"][1][k]), k


def test_flat_adamw_matches_torch_clip_and_adamw(dev):
    """utils/flat_optim.FlatAdamW (csrc/optim.hip: the clip factor of clip_grad_norm_ + AdamW over flat parameter / gradient /
    moment buffers, two launches) against torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW on the same gradients for several
    steps -- clipping active and inactive, weight decay, a learning-rate change -- and the optimizer's state_dict keeps the
    reference layout (step / exp_avg / exp_avg_sq per parameter) and survives a save / load."""
    from neuralsvb_amd.utils.trainer import FlatGradSync
    from neuralsvb_amd.utils.flat_optim import FlatAdamW
    g = torch.Generator().manual_seed(5)
    shapes = [(7, 3, 5), (13,), (4, 6), (1,), (33, 2)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    kw = dict(lr=3e-3, betas=(0.5, 0.999), eps=1e-6, weight_decay=0.01)
    ref_o, our_o = torch.optim.AdamW(ref_p, **kw), torch.optim.AdamW(our_p, **kw)
    sync = FlatGradSync(our_p, 1)
    flat = FlatAdamW(our_o, sync)
    for step, (max_norm, scale) in enumerate([(5.0, 10.0), (5.0, 0.01), (0.0, 1.0), (1.0, 3.0)]):
        if step == 2:
            for o in (ref_o, our_o):
                o.param_groups[0]["lr"] = 1e-3
        grads = [torch.randn(s, generator=g) * scale for s in shapes]
        for p, q, gr in zip(ref_p, our_p, grads):
            p.grad = gr.clone()
            q.grad.copy_(gr.to(dev))
        norm = torch.nn.utils.clip_grad_norm_(ref_p, max_norm) if max_norm else None
        ref_o.step()
        flat.set_clip(max_norm)
        flat.step()
        if norm is not None:
            assert abs(float(flat.norm) - float(norm)) <= 1e-5 * float(norm)
        for p, q in zip(ref_p, our_p):
            assert (p.detach() - q.detach().cpu()).abs().max().item() <= 2e-6 * max(1.0, p.abs().max().item()), step
    flat.export_state()
    sd = our_o.state_dict()
    assert set(sd["state"][0].keys()) == {"step", "exp_avg", "exp_avg_sq"} and float(sd["state"][0]["step"]) == 4.0
    for i, p in enumerate(ref_p):
        assert (ref_o.state[p]["exp_avg_sq"] - sd["state"][i]["exp_avg_sq"].cpu()).abs().max().item() <= 1e-6
    our_o.load_state_dict({"state": {k: {kk: vv.clone() for kk, vv in v.items()} for k, v in sd["state"].items()},
                           "param_groups": sd["param_groups"]})
    flat.reattach()
    assert our_o.state[our_p[0]]["exp_avg"].data_ptr() == flat.m.data_ptr()


@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("wd", [0.0, 0.01])
def test_flat_adamw_parameter_without_gradient_is_left_alone_like_torch(dev, wd, fused):
    """torch.optim.AdamW skips a parameter whose `.grad` is None: no weight decay, no moment decay, its own `step` does not
    advance (reference tasks/singing/svb_vae_task.py:84-118 steps stock AdamW).  FlatAdamW + FlatGradSync(drop_autograd_grads)
    through `gather_adopted`: parameter 1 NEVER gets a gradient, parameter 3 gets one only in some passes (so it has history
    when it is skipped), parameter 0 is a kernel sink (always a flat view).  Weights, moments and per-parameter step counts
    must equal torch's after every step, with and without weight decay; the fast flat launch must still be the one that runs
    while it is exact (wd = 0, only the never-updated parameter skipped).  `fused`: the optimizer as the tasks build it on the GPU
    (`torch.optim.AdamW(..., fused=True)`, svb_vae_task.py / hifigan_task.py) -- the exact fallback must step it although this
    object keeps the per-parameter `step` entries as CPU tensors."""
    if fused and dev.type != "cuda":
        pytest.skip("fused AdamW needs CUDA parameters")
    from neuralsvb_amd.utils.trainer import FlatGradSync
    from neuralsvb_amd.utils.flat_optim import FlatAdamW
    g = torch.Generator().manual_seed(11)
    shapes = [(6, 3, 5), (13,), (4, 6), (9,), (2, 2)]
    ref_p = [torch.nn.Parameter(torch.randn(s, generator=g)) for s in shapes]
    our_p = [torch.nn.Parameter(p.detach().clone().to(dev)) for p in ref_p]
    our_p[0]._svb_sink = True                      # (what functional._gbuf marks: a kernel accumulates into its flat view)
    kw = dict(lr=2e-3, betas=(0.8, 0.99), eps=1e-8, weight_decay=wd)
    ref_o, our_o = torch.optim.AdamW(ref_p, **kw), torch.optim.AdamW(our_p, fused=fused, **kw)
    sync = FlatGradSync(our_p, 1, drop_autograd_grads=True)
    flat = FlatAdamW(our_o, sync)
    flat.set_clip(1.5)
    sync.zero()                                    # as the Trainer does after set-up: autograd-only parameters drop their views
    assert our_p[0].grad is not None and all(q.grad is None for q in our_p[1:])
    torch_steps = 0
    orig = flat._torch_step

    def spy(skipped):
        nonlocal torch_steps
        torch_steps += 1
        return orig(skipped)
    flat._torch_step = spy
    has_grad = [(0, 2, 3, 4), (0, 2, 3, 4), (0, 2, 4), (0, 2, 3, 4), (0, 2, 4)]      # parameter 1 never, parameter 3 not in passes 2, 4
    for step, idx in enumerate(has_grad):
        for i, (p, q) in enumerate(zip(ref_p, our_p)):
            if i in idx:
                gr = torch.randn(shapes[i], generator=g) * 2.0
                p.grad = gr.clone()
                if i == 0:
                    q.grad.copy_(gr.to(dev))       # the sink: written in place
                else:
                    q.grad = gr.clone().to(dev)    # adopted tensor, as autograd hands it over
            else:
                p.grad = None
        torch.nn.utils.clip_grad_norm_(ref_p, 1.5)
        ref_o.step()
        flat.step(sync.gather_adopted())
        sync.zero()
        for i, (p, q) in enumerate(zip(ref_p, our_p)):
            assert (p.detach() - q.detach().cpu()).abs().max().item() <= 3e-6 * max(1.0, p.abs().max().item()), (step, i)
        if step == 1:
            assert torch_steps == (0 if wd == 0.0 else 2)       # only the never-updated parameter skipped: flat launch iff wd == 0
    flat.export_state()
    sd = our_o.state_dict()["state"]
    for i, p in enumerate(ref_p):
        rs = ref_o.state[p]
        want = float(rs["step"]) if "step" in rs else 0.0
        assert float(sd[i]["step"]) == want, (i, float(sd[i]["step"]), want)
        if "exp_avg" in rs:
            assert (rs["exp_avg"] - sd[i]["exp_avg"].cpu()).abs().max().item() <= 1e-6
            assert (rs["exp_avg_sq"] - sd[i]["exp_avg_sq"].cpu()).abs().max().item() <= 1e-6
        else:
            assert float(sd[i]["exp_avg"].abs().max()) == 0.0 and float(sd[i]["exp_avg_sq"].abs().max()) == 0.0
    assert torch_steps >= 3                                       # passes 2..4: a parameter with history was skipped / steps diverged
    assert all(bool(g.get("fused")) == fused for g in our_o.param_groups)      # the fallback restores the group's flags
    # an optimizer FlatAdamW cannot represent is reported as not eligible (Trainer.setup then keeps optimizer.step())
    assert not FlatAdamW.eligible(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(2))], amsgrad=True))
    assert not FlatAdamW.eligible(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(2))], maximize=True))
    assert FlatAdamW.eligible(torch.optim.AdamW([torch.nn.Parameter(torch.zeros(2))]))


@pytest.mark.gpu
@pytest.mark.parametrize("graph_mode", ["step", "pass"])
def test_hipgraph_replay_matches_eager_steps(gpu_only, tmp_path, graph_mode):
    """Trainer hip_graph mode must produce the same losses and weights as issuing every launch eagerly.  `step` (round 4): the
    forward+backward of both optimizer passes as ONE multi-stream graph -- weight gradients and the PPG encoder on their side
    streams, the critic's pass forked off the generator's forward beside the generator's backward -- replayed with fresh host
    randoms staged to device buffers; `pass`: one single-stream graph per optimizer pass."""
    dev = gpu_only
    results = {}
    # Replay must reproduce the eager launches bit for bit, so both modes have to take the same arithmetic: graph mode crops
    # the critic's windows with device-side starts (index_select + cat), eager launches use the fused crop whose gradient sums
    # the three windows in another order -- a 1e-7 difference that AdamW's first steps (update = lr * sign-like) turn into
    # 1e-4 on the loss terms two steps later.  The eager run therefore uses the unfused crop here.
    from neuralsvb_amd.modules import mel_disc
    mel_disc.FUSED_CROP = False
    try:
        _hipgraph_vs_eager(dev, tmp_path, results, graph_mode)
    finally:
        mel_disc.FUSED_CROP = True


def _hipgraph_vs_eager(dev, tmp_path, results, graph_mode):
    for mode in ("eager", "graph"):
        task, trainer, batch, hp = _setup(tmp_path / mode, dev)
        trainer.hip_graph, trainer.hip_graph_warmup, trainer.hip_graph_mode = mode == "graph", 1, graph_mode
        for m in task.modules():
            if isinstance(m, torch.nn.Dropout):
                m.p = 0.0
        g = torch.Generator().manual_seed(3)
        L = hp["latent_size"]
        eps_a, eps_p = torch.randn(2, L, 1, generator=g).to(dev), torch.randn(2, L, 1, generator=g).to(dev)
        orig_run = task.run_model
        task.run_model = lambda *a, _o=orig_run, **k: _o(*a, eps_a2a=eps_a, eps_p2p=eps_p, **k)
        host_lens = {k: batch[k].cpu() for k in ("mel_lengths", "prof_mel_lengths")}
        logs = []
        for step in range(1, 7):
            np.random.seed(200 + step)
            task.global_step = trainer.global_step = step
            pbar, _ = trainer.run_training_batch(0, dict(batch, **host_lens))
            logs.append({k: float(v) for k, v in pbar.items() if isinstance(v, torch.Tensor)})
        if mode == "graph":
            n_graphs = sum(1 for e in trainer._graphs.values() if e["graph"] is not None)
            assert n_graphs == (1 if graph_mode == "step" else 2)        # the step  |  gen pass + critic pass
        results[mode] = (logs, {k: v.detach().clone() for k, v in task.state_dict().items() if v.is_floating_point()})
    for step, (le, lg) in enumerate(zip(results["eager"][0], results["graph"][0])):
        assert le.keys() == lg.keys()
        for k in le:
            assert abs(le[k] - lg[k]) <= 1e-5 * max(1.0, abs(le[k])), (step, k, le[k], lg[k])
    for k, v in results["eager"][1].items():
        assert torch.allclose(v, results["graph"][1][k], rtol=1e-4, atol=1e-6), k


def _grads_after_passes(task, trainer, batch, dev, global_step, eps, seed):
    """One training step at `global_step`; -> (logged terms per pass, gradient of every parameter right before each
    optimizer step)."""
    terms, grads = {}, {}
    o_ts, o_before, o_run = task._training_step, task.on_before_optimization, task.run_model

    def _ts(sample, batch_idx, opt_idx, _o=o_ts):
        ret = _o(sample, batch_idx, opt_idx)
        if ret is not None:
            terms[opt_idx] = {k: float(v) for k, v in ret[1].items()}
            terms[opt_idx]["total"] = float(ret[0])
        return ret

    def _before(opt_idx, _o=o_before):
        grads[opt_idx] = {n: p.grad.detach().cpu().clone() for n, p in task.named_parameters()
                          if p.grad is not None and p.requires_grad}
        return _o(opt_idx)
    task._training_step, task.on_before_optimization = _ts, _before
    task.run_model = lambda *a, **k: o_run(*a, eps_a2a=eps[0].to(dev), eps_p2p=eps[1].to(dev), **k)
    np.random.seed(seed)
    task.global_step = trainer.global_step = global_step
    try:
        trainer.run_training_batch(0, batch)
    finally:
        task._training_step, task.on_before_optimization, task.run_model = o_ts, o_before, o_run
    return terms, grads


@pytest.mark.gpu
@pytest.mark.parametrize("phase", [2, 3])
def test_fused_step_paths_equal_elementwise_paths_well_conditioned(gpu_only, tmp_path, phase):
    """The same comparison on a WELL-CONDITIONED fixture, with tight bounds everywhere (advisor finding, round 4: the loose bounds
    of the small fixture -- 1.5e-2 across the critic, 6e-3 on the latent map -- would let a real 1 % regression of a fused kernel
    pass): 6 clips of 2.3 s, so the critic's windows see 430 frames of varying content (no near-constant InstanceNorm plane) and the
    latent map's / pooling stack's train-mode BatchNorm1d normalise over 6 clips instead of 2.  MI355X only (the lane emulator needs
    an hour for it)."""
    _fused_vs_elementwise(gpu_only, tmp_path, phase, well_conditioned=True)


@pytest.mark.parametrize("phase", [2, 3, "2-no-critic"])
def test_fused_step_paths_equal_elementwise_paths(dev, tmp_path, phase):
    """The same on the SMALL fixture (2 clips x 0.7 s: runs on the lane emulator too).  Its bounds across the critic (1.5e-2; 1e-1 on
    the emulator) and on the latent map (6e-3) are loose because the fixture is ill-conditioned there, see the comments at the
    assertion; the tight bounds on those paths are held by the well-conditioned MI355X variant above (5e-4 / 5e-3 / 2e-4) and, for
    the generator's fused paths on every device, by the `2-no-critic` case (1e-3)."""
    _fused_vs_elementwise(dev, tmp_path, phase)


def _fused_vs_elementwise(dev, tmp_path, phase, well_conditioned=False):
    """The fused passes of a step (mel loss, latent head + KL, GroupNorm+ReLU+residual, window crops, conformer residual
    epilogues) against the same step with each of them switched back to its stock element-wise form: every logged term
    and every gradient of the pass, in phase 2 (generator + critic) and phase 3 (latent map incl. the a2p way)."""
    from neuralsvb_amd.modules import fs2_vae, mel_disc, svb_vae, vc_asr
    no_critic = phase == "2-no-critic"      # the generator pass without adversarial terms: nothing ill-conditioned in the way
    phase = 2 if no_critic else phase
    # (the latent map's speaker projection is Conv1d(256, .) on h_style: phase 3 needs the real hidden_size -- 5 minutes on
    #  the CPU lane emulator, so that variant only runs there on request; the MI355X variant always runs)
    if phase == 3 and dev.type == "cpu" and os.environ.get("SVB_SLOW_TESTS", "0") != "1":
        pytest.skip("phase-3 step at hidden_size 256 on the lane emulator: set SVB_SLOW_TESTS=1 (passes; ~5 min)")
    # phase 2 runs in the bench's arithmetic (bf16x3: also the deferred multi-tensor reduce of the weight gradients, which only
    # exists on that path), phase 3 in fp32
    task, trainer, batch, hp = _setup(tmp_path, dev, ",hidden_size=256" if phase == 3 else ",conv_precision=bf16x3",
                                      **(dict(n_clips=6, seconds=2.3) if well_conditioned else {}))
    nb = batch["mels"].shape[0]
    hp["phase_2_steps"] = 2 if phase == 3 else 10 ** 6
    if no_critic:
        hp["lambda_mel_adv"] = 0.0
    gs = 3 if phase == 3 else 2
    L = hp["latent_size"]
    g = torch.Generator().manual_seed(5)
    eps = (torch.randn(nb, L, 1, generator=g), torch.randn(nb, L, 1, generator=g))
    sd0 = {k: v.detach().clone() for k, v in task.state_dict().items()}
    opt0 = [None if o is None else {"state": {}, "param_groups": o.state_dict()["param_groups"]} for o in trainer.optimizers]

    def run(fused):
        if dev.type == "cuda":          # (the previous run's critic pass may still be in flight on the Trainer's critic stream)
            trainer._join_critic_stream()
            torch.cuda.synchronize()
        task.load_state_dict(sd0)
        from neuralsvb_amd import functional as SF
        SF.note_weights_updated()
        for o, s in zip(trainer.optimizers, opt0):
            if o is not None:
                o.load_state_dict(s)
        hp["fused_mel_loss"] = hp["defer_wgrad_reduce"] = fused
        fs2_vae.FUSED_HEAD = svb_vae.FUSED_GN = svb_vae.SPLIT_STACKED = mel_disc.FUSED_CROP = vc_asr.FOLD_RESIDUALS = fused
        from neuralsvb_amd import kernels as K
        n_deferred = [0]
        o_flush = K.flush_deferred_reduces

        def counting_flush(end=True):
            n_deferred[0] += len(K._DEFERRED["descs"]) if K._DEFERRED is not None else 0
            return o_flush(end)
        K.flush_deferred_reduces = counting_flush
        try:
            out = _grads_after_passes(task, trainer, batch, dev, gs, eps, 77)
            if phase == 2 and not no_critic:           # (bf16x3: the generator's weight gradients all go through the deferred reduce)
                # (the gated stacks batch their own reduces inside the C executor; what is counted here are the other convs)
                assert (n_deferred[0] > 10) == bool(fused), n_deferred
            return out
        finally:
            K.flush_deferred_reduces = o_flush
            hp["fused_mel_loss"], hp["defer_wgrad_reduce"] = True, True
            fs2_vae.FUSED_HEAD = svb_vae.FUSED_GN = svb_vae.SPLIT_STACKED = mel_disc.FUSED_CROP = vc_asr.FOLD_RESIDUALS = True
    try:
        t1, g1 = run(True)
        t0, g0 = run(False)
    finally:
        from neuralsvb_amd import functional as SF
        SF.set_precision("fp32")
    assert sorted(t1) == sorted(t0) and len(t1) >= 1
    assert (2 in t1) == (phase == 3)
    worst = {}
    for oi in t0:
        assert set(t1[oi]) == set(t0[oi]), (oi, set(t1[oi]) ^ set(t0[oi]))
        for k, v in t0[oi].items():
            assert abs(t1[oi][k] - v) <= 1e-5 * max(1.0, abs(v)), (oi, k, t1[oi][k], v)
        assert set(g1[oi]) == set(g0[oi])
        for n, r in g0[oi].items():
            err = (g1[oi][n] - r).abs().max().item()
            # (the latent map's BatchNorm1d normalises over the 2 clips of this batch, the critic's InstanceNorm over near-
            #  constant planes: both amplify a 1e-7 summation-order difference -- tests/test_step_golden.py documents the same
            #  conditioning of the reference's own fp32 gradients)
            # On the MI355X the passes that cross the critic are held to the bound tests/test_step_golden.py uses for them
            # (the two runs differ in the summation order of the window gradients, and the critic's backward amplifies it);
            # the generator pass without adversarial terms and every pass on the emulator keep the tight bound.
            through_critic = dev.type == "cuda" and not no_critic and oi in (0, 1)
            # (1e-3, not 5e-4: a GroupNorm output within an ulp of zero gates differently in the two forms -- the row-resident
            #  kernels of round 3 moved one element of pitch_embed.weight's gradient by 6.8e-4 of the largest one)
            # (round 4: the pooling stack's BatchNorm is csrc/batchnorm.hip, which accumulates in double like the reference's CPU
            #  batch_norm; an fp32-accumulating draft of it moved ONE critic gradient of this test to 5.8e-3 -- with double sums the
            #  worst element of the emulator run is 6.7e-5)
            # (oi == 2, the latent map: its BatchNorm1d layers normalise 2 clips x ONE position -- two values per channel; 2e-3
            #  until the PPG encoder's position scores moved into the attention kernel, which shifts h_content by 1e-6 and this
            #  pass's worst element to 3.7e-3 on the MI355X: 6e-3)
            tol = 1.5e-2 if through_critic else (6e-3 if oi == 2 else 1e-3)
            # Emulator, generator pass ACROSS the critic: 1e-1 (round 4).  This fixture (2 clips of 0.7 s, synthetic tones) puts a
            # critic window on a near-constant plane, and how far the two forms' 1e-7 differences are amplified through that
            # InstanceNorm depends on the exact input: with the PPG encoder's projections fused (vc_asr.FUSE_QKV, h_content moves
            # by 1e-7) the SAME two generator paths that agree to 5.9e-5 otherwise read 5.0e-2 (cosine 0.9998) on every parameter
            # the adversarial gradient reaches -- the fused form alone moves by 5 % between the two inputs, the element-wise form
            # by 1.4e-4; running the fused projection for its side effects only changes nothing (no aliasing, no corruption).
            # What this pass still pins: structure (same parameter set, same logged terms to 1e-5).  The generator's fused paths
            # are held to 1e-3 by the variant without adversarial terms (now also on the emulator), the critic's by the critic
            # pass below (1e-3) and tests/test_modules_disc.py.
            if dev.type == "cpu" and not no_critic and oi == 0:
                tol = 1e-1
            if well_conditioned:
                tol = WELL_COND_TOL[oi]
            worst[oi] = max(worst.get(oi, 0.0), err / max(r.abs().max().item(), 1e-3))
            assert err <= tol * max(r.abs().max().item(), 1e-3), (oi, n, err, r.abs().max().item())
    print(f"fused vs element-wise step paths, phase {phase}{' (well-conditioned fixture)' if well_conditioned else ''}: worst gradient "
          f"element error relative to the tensor's largest, per pass: { {k: float(f'{v:.2e}') for k, v in worst.items()} }")


# test_fused_step_paths_equal_elementwise_paths_well_conditioned: bound per optimizer pass (0 generator -- across the critic --,
# 1 critic, 2 latent map).  Measured on the MI355X (round 5): 5.9e-5 / 1.4e-3 / 1.1e-5, against 1.5e-2 / 1.5e-2 / 6e-3 that the small
# fixture needs: 30x tighter on the generator's and the latent map's fused paths, 3x on the critic's.
WELL_COND_TOL = {0: 5e-4, 1: 5e-3, 2: 2e-4}


# bands for test_bf16x3_trains_like_fp32: relative deviation of the 20-step moving averages over 300 steps.  Measured on the MI355X
# (profiles/r05_trajectory_fp32_vs_bf16x3.log): reconstruction terms 2.6 %, KL 5.6 %, the generator's adversarial terms 10.8 %, the
# critic's own terms 12 ... 29 %; the last-50-step means agree to 0.2 % (reconstruction) ... 4 % (critic).  A second fp32 run from
# the same seeds reproduces the first bit for bit (deviation 0), so the band is the arithmetic's, not run-to-run noise.
TRAJ_BAND = {"recon": 0.05, "kl": 0.15, "adv": 0.5}


@pytest.mark.gpu
def test_bf16x3_trains_like_fp32(gpu_only):
    """300 optimizer steps (generator + critic, reference tasks/singing/svb_vae_task.py:579-676) from identical weights,
    batches, seeds and draws, once with `conv_precision: fp32` and once with `bf16x3` (tools/train_trajectory.py).  The runs are
    chaotic, so not bit-comparable; asserted: both stay finite, the reconstruction terms fall, and every loss term's 20-step moving
    average of the bf16x3 run stays within a stated band of the fp32 run's (reconstruction 5 %, KL 15 %, adversarial / critic terms 50 %)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import train_trajectory as TT
    steps = 300
    a = TT.run("fp32", steps, gpu_only)
    b = TT.run("bf16x3", steps, gpu_only)
    cmp_ = TT.compare(a, b)
    for k in TT.TERMS:
        assert np.isfinite(a[k]).all() and np.isfinite(b[k]).all(), k
        kind = "recon" if k.startswith(("l1", "ssim")) else ("kl" if k.endswith("_kl") else "adv")
        print(f"{k:8s} first-step rel dev {cmp_[k]['first_step_rel_dev']:.2e}  worst smoothed rel dev {cmp_[k]['worst_rel_dev_smoothed']:.3e}  "
              f"last-50 means fp32 {cmp_[k]['last50_mean_fp32']:.5g} bf16x3 {cmp_[k]['last50_mean_bf16x3']:.5g}")
        assert cmp_[k]["worst_rel_dev_smoothed"] <= TRAJ_BAND[kind], (k, cmp_[k])
        assert cmp_[k]["first_step_rel_dev"] <= 1e-3, (k, cmp_[k])            # same weights, same draws: the first step agrees closely
    for k in ("l1a2a", "l1p2p"):
        for r in (a, b):
            assert np.mean(r[k][-20:]) < 0.8 * np.mean(r[k][:5]), (k, np.mean(r[k][:5]), np.mean(r[k][-20:]))     # it trains
