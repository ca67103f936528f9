"""Paired amateur/professional dataset of the vae_global_mle_eng task.

Produces exactly the sample / batch dictionaries of the reference chain BaseTTSDataset -> FastSpeechDataset ->
FastSingingDataset -> MultiSpkEmbDataset (tasks/tts/dataset_utils.py:15-205, tasks/singing/neural_svb_task.py:10-62,
tasks/singing/svb_vae_task.py:20-45) for the hparams of the path (use_pitch_embed, no spk_id / spk_embed / cwt),
reading the reference's binary format (item schema: SURVEY §8a D1).
"""
import os

import numpy as np
import torch

from ..utils import pitch_utils
from ..utils.batching import collate_1d, collate_2d
from ..utils.hparams import hparams
from ..utils.indexed_datasets import IndexedDataset


class MultiSpkEmbDataset(torch.utils.data.Dataset):
    def __init__(self, prefix, shuffle=False, data_dir=None):
        super().__init__()
        self.hparams = hparams
        self.prefix, self.shuffle = prefix, shuffle
        self.sort_by_len = hparams["sort_by_len"]
        self.data_dir = hparams["binary_data_dir"] if data_dir is None else data_dir
        self.indexed_ds = None
        sizes = np.load(f"{self.data_dir}/{prefix}_lengths.npy")
        if (prefix == "test" or hparams.get("infer")) and hparams["num_test_samples"] > 0:
            avail = [x for x in range(hparams["num_test_samples"]) if x < len(sizes)]
            if len(hparams["test_ids"]) > 0:
                avail = hparams["test_ids"] + avail
        else:
            avail = list(range(len(sizes)))
        if hparams["min_frames"] > 0:
            avail = [x for x in avail if sizes[x] >= hparams["min_frames"]]
        self.avail_idxs = avail
        self.sizes = [sizes[i] for i in avail]
        stats = f"{self.data_dir}/train_f0s_mean_std.npy"
        if os.path.exists(stats):
            m, s = np.load(stats)
            hparams["f0_mean"], hparams["f0_std"] = float(m), float(s)
        else:
            hparams["f0_mean"] = hparams["f0_std"] = None

    # ---- sizes / ordering (tasks/base_task.py:70-92) ----
    def __len__(self):
        return len(self.sizes)

    def num_tokens(self, index):
        return min(self.sizes[index], hparams["max_frames"])

    def ordered_indices(self):
        if not self.shuffle:
            return np.arange(len(self))
        idx = np.random.permutation(len(self))
        if self.sort_by_len:
            idx = idx[np.argsort(np.array(self.sizes)[idx], kind="mergesort")]
        return idx

    @property
    def num_workers(self):
        return int(os.getenv("NUM_WORKERS", hparams["ds_workers"]))

    def _get_item(self, index):
        if self.indexed_ds is None:
            self.indexed_ds = IndexedDataset(f"{self.data_dir}/{self.prefix}")
        return self.indexed_ds[self.avail_idxs[index]]

    def raw_item(self, index):
        """The decoded item as stored (numpy arrays, untruncated) + its dataset id: input of tasks/device_collate.py."""
        item = dict(self._get_item(index))
        assert max(len(item["mel"]), len(item["prof_mel"])) == self.sizes[index], (len(item["mel"]), self.sizes[index])
        if self.hparams.get("normalize_pitch", False):
            raise NotImplementedError("normalize_pitch is false on this path (vc_ppg.yaml)")
        item["id"] = index
        return item

    def _voice(self, item, pre):
        """mel / f0 / uv / pitch of one voice, truncated to max_frames and a multiple of frames_multiple."""
        hp = self.hparams
        spec = torch.Tensor(item[pre + "mel"])[:hp["max_frames"]]
        n = spec.shape[0] // hp["frames_multiple"] * hp["frames_multiple"]
        spec = spec[:n]
        f0_raw = np.asarray(item[pre + "f0"])
        if hp.get("normalize_pitch", False):
            raise NotImplementedError("normalize_pitch is false on this path (vc_ppg.yaml)")
        pitch = torch.LongTensor(item[pre + "pitch"])[:n] if (pre + "pitch") in item else None
        f0, uv = pitch_utils.norm_interp_f0(f0_raw[:n], hp)
        return spec, torch.FloatTensor(f0), torch.FloatTensor(uv), pitch

    def __getitem__(self, index):
        item = self._get_item(index)
        assert max(len(item["mel"]), len(item["prof_mel"])) == self.sizes[index], (len(item["mel"]), self.sizes[index])
        mel, f0, uv, pitch = self._voice(item, "")
        pmel, pf0, puv, ppitch = self._voice(item, "prof_")
        al = torch.LongTensor(item["a2p_f0_alignment"])[:ppitch.shape[0]].clip(max=pitch.shape[0] - 1)
        assert al.shape == ppitch.shape, ("a2p F0 alignment with unmatched shape: ", al.shape, ppitch.shape)
        return {"id": index, "item_name": item["item_name"], "text": item.get("txt"), "mel": mel,
                "mel_nonpadding": mel.abs().sum(-1) > 0, "energy": (mel.exp() ** 2).sum(-1).sqrt(),
                "f0": f0, "uv": uv, "pitch": pitch, "prof_mel": pmel, "prof_f0": pf0, "prof_uv": puv,
                "prof_pitch": ppitch, "prof_energy": (pmel.exp() ** 2).sum(-1).sqrt(),
                "prof_mel_nonpadding": pmel.abs().sum(-1) > 0, "a2p_f0_alignment": al,
                "multi_spk_emb": torch.FloatTensor(item["multi_spk_emb"])}

    def collater(self, samples):
        if len(samples) == 0:
            return {}
        c1 = lambda k, pad=0: collate_1d([s[k] for s in samples], pad)
        return {
            "id": torch.LongTensor([s["id"] for s in samples]), "item_name": [s["item_name"] for s in samples],
            "nsamples": len(samples), "text": [s["text"] for s in samples],
            "mels": collate_2d([s["mel"] for s in samples], 0.0),
            "mel_lengths": torch.LongTensor([s["mel"].shape[0] for s in samples]),
            "mel2ph": None, "energy": c1("energy", 0.0), "pitch": c1("pitch"), "f0": c1("f0", 0.0), "uv": c1("uv"),
            "prof_f0": c1("prof_f0", 0.0), "prof_pitch": c1("prof_pitch"), "prof_uv": c1("prof_uv"),
            "prof_energy": c1("prof_energy", 0.0), "prof_mel2ph": None,
            "prof_mels": collate_2d([s["prof_mel"] for s in samples], 0.0),
            "prof_mel_lengths": torch.LongTensor([s["prof_mel"].shape[0] for s in samples]),
            "a2p_f0_alignment": c1("a2p_f0_alignment"),
            "multi_spk_emb": collate_2d([s["multi_spk_emb"] for s in samples]),
        }
