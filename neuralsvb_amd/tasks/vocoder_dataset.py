"""Wave-segment dataset of the vocoder task (BASELINE configs[2]).

The reference names `tasks.vocoder.hifigan.HifiGanTask` (egs/egs_bases/tts/vocoder/hifigan.yaml:2) but ships no
`tasks/vocoder/`: this dataset is composed for that task over the reference's binary format (utils/indexed_datasets.py:7-54,
items as written by the binarizer with `binarization_args.with_wav: true`, egs/egs_bases/tts/vocoder/base.yaml:2-3): per
item `mel [T,80]`, `wav [>= T*hop]`, `f0 [T]`.  A training sample is a random window of `max_samples` samples
(hifigan.yaml:22), cut on a frame boundary so that mel / f0 / wav stay aligned; validation uses whole clips (batch 1).
"""
import os

import numpy as np
import torch

from ..utils.hparams import hparams
from ..utils.indexed_datasets import IndexedDataset


class VocoderDataset(torch.utils.data.Dataset):
    def __init__(self, prefix, shuffle=False):
        super().__init__()
        self.hparams = hparams
        self.prefix, self.shuffle = prefix, shuffle
        self.data_dir = hparams["binary_data_dir"]
        self.is_infer = prefix == "test"
        self.batch_max_frames = 0 if self.is_infer else hparams["max_samples"] // hparams["hop_size"]
        self.hop = hparams["hop_size"]
        self.indexed_ds = None
        self.sizes = list(np.load(f"{self.data_dir}/{prefix}_lengths.npy"))
        self.avail_idxs = [i for i, s in enumerate(self.sizes) if s > self.batch_max_frames]
        self.sizes = [self.sizes[i] for i in self.avail_idxs]

    def __len__(self):
        return len(self.sizes)

    def num_tokens(self, index):
        return self.sizes[index]

    def ordered_indices(self):
        return np.random.permutation(len(self)) if self.shuffle else np.arange(len(self))

    @property
    def num_workers(self):
        return int(os.getenv("NUM_WORKERS", hparams["ds_workers"]))

    def _get_item(self, index):
        if self.indexed_ds is None:
            self.indexed_ds = IndexedDataset(f"{self.data_dir}/{self.prefix}")
        return self.indexed_ds[self.avail_idxs[index]]

    def __getitem__(self, index):
        item = self._get_item(index)
        mel = torch.FloatTensor(item["mel"])
        T = mel.shape[0]
        return {"id": index, "item_name": item["item_name"], "mel": mel,
                "wav": torch.FloatTensor(np.asarray(item["wav"], dtype=np.float32))[:T * self.hop],
                "f0": torch.FloatTensor(np.asarray(item["f0"], dtype=np.float32))[:T]}

    def collater(self, batch):
        """Random aligned windows of batch_max_frames frames (whole clips when 0: validation / test, batch 1)."""
        if len(batch) == 0:
            return {}
        mels, wavs, f0s, names = [], [], [], []
        for s in batch:
            T = s["mel"].shape[0]
            if self.batch_max_frames > 0:
                st = int(np.random.randint(0, T - self.batch_max_frames))          # sizes > batch_max_frames (avail_idxs)
                T = self.batch_max_frames
            else:
                st = 0
            mels.append(s["mel"][st:st + T].t())
            f0s.append(s["f0"][st:st + T])
            wavs.append(s["wav"][st * self.hop:(st + T) * self.hop][None])
            names.append(s["item_name"])
        n = min(m.shape[1] for m in mels)
        return {"mels": torch.stack([m[:, :n] for m in mels]), "wavs": torch.stack([w[:, :n * self.hop] for w in wavs]),
                "f0": torch.stack([f[:n] for f in f0s]), "item_name": names, "nsamples": len(batch)}
