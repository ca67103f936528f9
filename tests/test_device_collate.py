"""SURVEY 8f3 -- the batch assembled on the GPU (tasks/device_collate.py + csrc/collate.hip) against the product's host
collater, which tests/test_step_golden.py pins to the reference's own collater digests: same keys, shapes, dtypes; integer and
copied float fields identical; the normalised F0 tracks identical up to the last fp32 bit (the device evaluates the same fp64
expressions; only log2's last fp64 ulp can differ between libm and the device library); energies to 1e-6."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import step_common as C  # noqa: E402


@pytest.fixture(scope="module")
def ds(tmp_path_factory):
    from neuralsvb_amd.utils.hparams import set_hparams, hparams
    tmp = tmp_path_factory.mktemp("devcollate")
    set_hparams(config=os.path.join(ROOT, "egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml"), exp_name="",
                hparams_str=C.STEP_HPARAMS, print_hparams=False)
    hparams["binary_data_dir"] = str(tmp / "bin")
    C.write_dataset(hparams["binary_data_dir"], hparams)
    from neuralsvb_amd.tasks.dataset import MultiSpkEmbDataset
    return MultiSpkEmbDataset("train", False)


def _ulp_diff(a, b):
    ia = a.contiguous().view(torch.int32).long()
    ib = b.contiguous().view(torch.int32).long()
    return (ia - ib).abs()


@pytest.mark.parametrize("group", [[0], [0, 1, 2, 3], [7, 3, 1], list(range(C.N_TRAIN))])
def test_device_collated_batch_equals_host_collater(dev, ds, group):
    from neuralsvb_amd.tasks.device_collate import DeviceCollater
    ref = ds.collater([ds[i] for i in group])
    got = DeviceCollater(ds, dev)([ds.raw_item(i) for i in group])
    assert set(got) == set(ref)
    for k, r in ref.items():
        g = got[k]
        if not isinstance(r, torch.Tensor):
            assert g == r, k
            continue
        g = g.cpu()
        assert g.shape == r.shape and g.dtype == r.dtype, (k, g.shape, r.shape, g.dtype, r.dtype)
        if k in ("f0", "prof_f0"):
            d = _ulp_diff(g, r)
            assert int(d.max()) <= 1 and float((d > 0).float().mean()) < 1e-3, (k, int(d.max()), float((d > 0).float().mean()))
        elif k in ("energy", "prof_energy"):
            assert float((g - r).abs().max()) <= 1e-6 * max(1.0, float(r.abs().max())), k
        else:
            assert torch.equal(g, r), k


def test_norm_interp_f0_kernel_edge_cases(dev):
    """all-unvoiced clip -> zeros; unvoiced head / tail -> constant extrapolation; single voiced frame; `standard` and no
    normalisation; padding zero-filled -- against the host implementation (reference utils/pitch_utils.py:160-177)."""
    from neuralsvb_amd import kernels as K
    from neuralsvb_amd.utils import pitch_utils
    rng = np.random.RandomState(3)
    tracks = [np.zeros(37), np.r_[np.zeros(5), rng.uniform(80, 600, 20), np.zeros(9)], np.r_[np.zeros(11), 220.0, np.zeros(4)],
              rng.uniform(80, 600, 64) * (rng.rand(64) > 0.4), rng.uniform(80, 600, 3)]
    lens = [len(t) for t in tracks]
    off = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)
    src = torch.from_numpy(np.concatenate(tracks)).to(dev)
    d_off, d_len = torch.from_numpy(off).to(dev), torch.tensor(lens, dtype=torch.int32).to(dev)
    for hp in ({"pitch_norm": "log", "use_uv": True}, {"pitch_norm": "standard", "use_uv": True, "f0_mean": 210.5, "f0_std": 63.25},
               {"pitch_norm": "none", "use_uv": False}):
        f0, uv = K.norm_interp_f0(src, d_off, d_len, max(lens) + 3, hp["pitch_norm"], hp.get("f0_mean"), hp.get("f0_std"),
                                  hp["use_uv"])
        for b, t in enumerate(tracks):
            rf, ru = pitch_utils.norm_interp_f0(t, hp)
            got = f0[b, :len(t)].cpu()
            assert int(_ulp_diff(got, torch.FloatTensor(rf)).max()) <= 1, (hp, b)
            assert torch.equal(uv[b, :len(t)].cpu(), torch.FloatTensor(ru)), (hp, b)
            assert float(f0[b, len(t):].abs().max()) == 0.0 and float(uv[b, len(t):].abs().max()) == 0.0


@pytest.mark.gpu
def test_build_dataloader_with_device_collate(ds, gpu_only):
    """hparams device_collate=true: BaseTask.build_dataloader hands out batches assembled on the GPU (workers only decode
    items); same batches, in the same order, as the default host loader."""
    from neuralsvb_amd.utils.hparams import hparams
    from neuralsvb_amd.tasks.svb_vae_task import SVBVAEMleTask
    from neuralsvb_amd.utils.trainer import move_to_device
    task = SVBVAEMleTask()
    prev = hparams.get("device_collate", False)
    try:
        hparams["device_collate"] = False
        host = list(task.build_dataloader(ds, False, hparams["max_tokens"], 4))
        hparams["device_collate"] = True
        devl = list(task.build_dataloader(ds, False, hparams["max_tokens"], 4))
    finally:
        hparams["device_collate"] = prev
    assert len(devl) == len(host) == 2
    for hb, db in zip(host, devl):
        hb = move_to_device(hb, gpu_only)
        assert db["item_name"] == hb["item_name"] and db["mels"].is_cuda
        for k in ("mels", "prof_mels", "pitch", "prof_pitch", "a2p_f0_alignment", "uv", "prof_uv", "multi_spk_emb"):
            assert torch.equal(db[k], hb[k]), k
        assert int(_ulp_diff(db["f0"].cpu(), hb["f0"].cpu()).max()) <= 1


@pytest.mark.gpu
def test_device_collate_multi_worker_soak(ds, gpu_only):
    """The loader as a training run uses it: two worker processes decode items, batches of changing size and length arrive
    back to back for several epochs (the pinned staging buffers and the H2D copies of batch n+1 are reused / issued while
    batch n's kernels may still run), every batch compared with the host collater's.  This is the soak behind
    `device_collate: true` being the YAML default."""
    from neuralsvb_amd.tasks.device_collate import DeviceCollateLoader
    from neuralsvb_amd.utils.trainer import move_to_device
    rng = np.random.RandomState(3)
    batches = []
    for _ in range(6):                               # 6 epochs of ragged batches (1..5 items), shuffled
        order = rng.permutation(C.N_TRAIN).tolist()
        while order:
            n = int(rng.randint(1, 6))
            batches.append(order[:n])
            order = order[n:]
    loader = DeviceCollateLoader(ds, batches, gpu_only, num_workers=2)
    seen = 0
    kept = []
    for idx, db in zip(batches, loader):
        hb = move_to_device(ds.collater([ds[i] for i in idx]), gpu_only)
        assert db["item_name"] == hb["item_name"]
        for k in ("mels", "prof_mels", "pitch", "prof_pitch", "a2p_f0_alignment", "uv", "prof_uv", "multi_spk_emb"):
            assert torch.equal(db[k], hb[k]), (seen, k)
        assert int(_ulp_diff(db["f0"].cpu(), hb["f0"].cpu()).max()) <= 1
        kept.append((db["mels"], hb["mels"]))        # earlier batches must survive the staging buffers' reuse
        seen += 1
    assert seen == len(batches) >= 12
    torch.cuda.synchronize()
    for a, b in kept:
        assert torch.equal(a, b)
