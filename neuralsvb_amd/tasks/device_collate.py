"""Batches assembled ON the GPU: the device-side twin of MultiSpkEmbDataset.__getitem__ + collater.

The reference builds a batch with ~40 small host tensor ops per clip (truncate, FloatTensor/LongTensor, norm_interp_f0 in numpy,
exp/sum/sqrt for the energy, clip of the alignment) followed by zero-padded collation on the host and one H2D copy per field
(tasks/singing/svb_vae_task.py:20-45, tasks/tts/dataset_utils.py:133-205, utils/__init__.py:118-161,
utils/pitch_utils.py:160-177).  Here the host only slices the decoded item arrays into three PINNED staging buffers (one per
dtype: float32 mels + speaker embeddings, float64 F0 tracks, int64 pitch bins + alignment) and ships them with three asynchronous
copies; padding, the alignment clamp, the F0 normalisation + gap interpolation and the frame energies are HIP kernels
(csrc/collate.hip).  The result has the keys, shapes, dtypes and values of `move_to_device(dataset.collater(samples))`
(tests/test_device_collate.py); clip lengths additionally stay on the host, where the task reads them without a sync.

Opt-in: hparams `device_collate: true` (BaseTask.build_dataloader), or call DeviceCollater directly.
"""
import numpy as np
import torch

from .. import kernels as K


class DeviceCollater:
    def __init__(self, dataset, device):
        self.ds, self.hp, self.device = dataset, dataset.hparams, torch.device(device)
        self._stage = {}
        self._copied = None          # event after the last batch's H2D copies: the staging buffers are reused

    # ---- pinned staging buffers, grown on demand and reused batch after batch -------------------------------------------
    def _buf(self, key, n, dtype):
        b = self._stage.get(key)
        if b is None or b.numel() < n:
            b = torch.empty(max(n, 1), dtype=dtype)
            if self.device.type == "cuda":
                b = b.pin_memory()
            self._stage[key] = b
        return b

    def _ship(self, key, arrays, dtype, np_dtype):
        n = sum(a.size for a in arrays)
        buf = self._buf(key, n, dtype)
        view = buf.numpy()
        o = 0
        for a in arrays:
            view[o:o + a.size] = np.asarray(a, dtype=np_dtype).reshape(-1)
            o += a.size
        return buf[:n].to(self.device, non_blocking=True)

    def __call__(self, raw_items):
        """raw_items: list of dicts from MultiSpkEmbDataset.raw_item()."""
        hp, dev = self.hp, self.device
        if len(raw_items) == 0:
            return {}
        if self._copied is not None:
            self._copied.synchronize()          # the previous batch's asynchronous copies read these buffers
        fm, mf = hp["frames_multiple"], hp["max_frames"]
        B = len(raw_items)
        n_a = [min(len(r["mel"]), mf) // fm * fm for r in raw_items]
        n_p = [min(len(r["prof_mel"]), mf) // fm * fm for r in raw_items]
        W = np.asarray(raw_items[0]["mel"]).shape[1]
        for r, na, npf in zip(raw_items, n_a, n_p):
            assert len(r["a2p_f0_alignment"][:npf]) == npf, ("a2p F0 alignment with unmatched shape: ",
                                                             len(r["a2p_f0_alignment"]), npf)
        # ---- three staging buffers ---------------------------------------------------------------------------------------
        f32 = self._ship("f32", [np.asarray(r["mel"])[:n] for r, n in zip(raw_items, n_a)] +
                         [np.asarray(r["prof_mel"])[:n] for r, n in zip(raw_items, n_p)] +
                         [np.asarray(r["multi_spk_emb"]) for r in raw_items], torch.float32, np.float32)
        f64 = self._ship("f64", [np.asarray(r["f0"])[:n] for r, n in zip(raw_items, n_a)] +
                         [np.asarray(r["prof_f0"])[:n] for r, n in zip(raw_items, n_p)], torch.float64, np.float64)
        i64 = self._ship("i64", [np.asarray(r["pitch"])[:n] for r, n in zip(raw_items, n_a)] +
                         [np.asarray(r["prof_pitch"])[:n] for r, n in zip(raw_items, n_p)] +
                         [np.asarray(r["a2p_f0_alignment"])[:n] for r, n in zip(raw_items, n_p)], torch.int64, np.int64)
        off_a = np.concatenate([[0], np.cumsum(n_a)[:-1]]).astype(np.int32)
        off_p = np.concatenate([[0], np.cumsum(n_p)[:-1]]).astype(np.int32)
        meta = self._ship("meta", [off_a, np.asarray(n_a, np.int32), off_p, np.asarray(n_p, np.int32),
                                   np.asarray(n_a, np.int32) - 1], torch.int32, np.int32).view(5, B)
        if dev.type == "cuda":
            self._copied = torch.cuda.Event()
            self._copied.record()
        d_off_a, d_len_a, d_off_p, d_len_p, d_clip = (meta[i] for i in range(5))
        ta, tp = max(n_a), max(n_p)
        sa, sp = sum(n_a), sum(n_p)
        # ---- device side -----------------------------------------------------------------------------------------------------
        mel_a, mel_p = f32[:sa * W].view(sa, W), f32[sa * W:(sa + sp) * W].view(sp, W)
        emb = f32[(sa + sp) * W:].view(B, *np.asarray(raw_items[0]["multi_spk_emb"]).shape).clone()
        mels = K.collate_pad(mel_a, d_off_a, d_len_a, ta, W, 0.0)
        prof_mels = K.collate_pad(mel_p, d_off_p, d_len_p, tp, W, 0.0)
        std = dict(pitch_norm=hp["pitch_norm"], mean=hp.get("f0_mean"), std=hp.get("f0_std"), use_uv=hp["use_uv"])
        f0, uv = K.norm_interp_f0(f64[:sa], d_off_a, d_len_a, ta, **std)
        pf0, puv = K.norm_interp_f0(f64[sa:sa + sp], d_off_p, d_len_p, tp, **std)
        pitch = K.collate_pad(i64[:sa], d_off_a, d_len_a, ta)
        ppitch = K.collate_pad(i64[sa:sa + sp], d_off_p, d_len_p, tp)
        align = K.collate_pad(i64[sa + sp:sa + 2 * sp], d_off_p, d_len_p, tp, clip_max=d_clip)
        return {
            "id": torch.tensor([r["id"] for r in raw_items], dtype=torch.int64, device=dev),
            "item_name": [r["item_name"] for r in raw_items], "nsamples": B, "text": [r.get("txt") for r in raw_items],
            "mels": mels, "mel_lengths": torch.LongTensor(n_a), "mel2ph": None,
            "energy": K.mel_energy(mels, d_len_a), "pitch": pitch, "f0": f0, "uv": uv,
            "prof_f0": pf0, "prof_pitch": ppitch, "prof_uv": puv, "prof_energy": K.mel_energy(prof_mels, d_len_p),
            "prof_mel2ph": None, "prof_mels": prof_mels, "prof_mel_lengths": torch.LongTensor(n_p),
            "a2p_f0_alignment": align, "multi_spk_emb": emb,
        }


class _RawItems(torch.utils.data.Dataset):
    """What DataLoader workers hand back in device-collate mode: the decoded items, untouched."""

    def __init__(self, dataset):
        self.ds = dataset

    def __len__(self):
        return len(self.ds)

    def __getitem__(self, index):
        return self.ds.raw_item(index)


class DeviceCollateLoader:
    """DataLoader twin: workers decode items (IndexedDataset read + unpickle), the consuming process assembles the batch on
    its GPU."""

    def __init__(self, dataset, batches, device, num_workers=0):
        self.collate = DeviceCollater(dataset, device)
        self.inner = torch.utils.data.DataLoader(_RawItems(dataset), collate_fn=list, batch_sampler=batches,
                                                 num_workers=num_workers)

    def __len__(self):
        return len(self.inner)

    def __iter__(self):
        for raw in self.inner:
            yield self.collate(raw)
