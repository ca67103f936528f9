"""Batched paired-clip binarizer on the MI355X: the offline step that produces the items the hot path trains on.

Replaces, for the `vae_global_mle_eng` data, `PopBuTFyENSpkEMBinarizer.process_item` / `process_data`
(reference data_gen/singing/binarize_para.py:116-217, 234-260) and what they call per clip:
  * `PWG.wav2spec` -> `process_utterance` (vocoders/pwg.py:106-122, data_gen/tts/data_gen_utils.py:93-147): librosa STFT + mel
    filterbank + log10, wav right-padded to a whole number of hops and cut to frames*hop            -> `svb_stft_mel`, one
    launch for every clip of a batch (amateur and professional together, ragged lengths);
  * `get_pitch` (data_gen_utils.py:150-184): the F0 track of an EXTERNAL extractor (Praat through parselmouth in the reference)
    left-padded by 2*pad_size frames, right-padded to the mel length, +-8 frames tolerated            -> host arithmetic, exact;
    `f0_to_coarse` (utils/pitch_utils.py:130-146)                                                     -> `svb_f0_to_coarse`,
    one launch for the batch, bit-exact bins;
  * `get_pitch_align` -> `EHSADTW` (binarize_para.py:171-186, modules/voice_conversion/dtw/enhance_sadtw.py:18-113): the
    shape-aware F0 DTW that gives `a2p_f0_alignment`                                                  -> `svb_f0_shape_hist`,
    `svb_hist_cost`, `svb_dtw_align` over all pairs of the batch;
  * the speaker-embedding pick of PopBuTFyENSpkEMBinarizer.process_item (:234-260): own embedding + `spk_emb_num` embeddings
    of shuffled pieces of the same song (python's `random`, as there)                                 -> host, same calls.
Items are written with the reference's `IndexedDatasetBuilder` format (utils/indexed_datasets.py) and the side files
`<prefix>_lengths.npy` / `<prefix>_f0s_mean_std.npy` of process_data (:155-165).

The F0 extractor itself is NOT rebuilt (it is a third-party dependency of the reference too): `f0_fn(wav, hparams)` returns the
raw track the way `parselmouth.Sound(wav, sr).to_pitch_ac(...).selected_array['frequency']` does; the default tries to import
parselmouth and raises if it is missing.  File walking / train-test splits / the Resemblyzer encoder stay outside as well:
callers hand over `metas` (item name, the two wavs, speaker id, the item names of the set).
"""
import os
import random
import re

import numpy as np
import torch

from .. import kernels as K
from ..modules.dtw import ehsadtw_batch
from ..modules.frontend import MelFrontend
from ..utils.indexed_datasets import IndexedDatasetBuilder


class BinarizationError(Exception):
    pass


class F0ExtractorMissing(RuntimeError):
    """Never swallowed by the per-item `except Exception` (that would silently skip a whole dataset)."""


def praat_f0(wav, hp):
    """The reference's extractor call (data_gen_utils.py:160-171): needs `parselmouth` (not part of this package)."""
    try:
        import parselmouth
    except ImportError as e:
        raise F0ExtractorMissing("no F0 extractor: install praat-parselmouth or pass f0_fn=... to ParaBinarizer") from e
    time_step = hp["hop_size"] / hp["audio_sample_rate"]
    return parselmouth.Sound(wav, hp["audio_sample_rate"]).to_pitch_ac(
        time_step=time_step, voicing_threshold=0.6, pitch_floor=80, pitch_ceiling=750).selected_array["frequency"]


def _read_wav(w, sr):
    if isinstance(w, str):
        from scipy.io import wavfile
        fs, data = wavfile.read(w)
        if fs != sr:
            raise BinarizationError(f"{w}: {fs} Hz, expected {sr} Hz (resampling is not part of this path)")
        # librosa.load's conventions (data_gen_utils.py:123 via wav2spec): integer PCM scaled to [-1, 1), 8-bit PCM is unsigned
        if data.dtype == np.uint8:
            data = (data.astype(np.float32) - 128.0) / 128.0
        elif data.dtype == np.int16:
            data = data.astype(np.float32) / 32768.0
        elif data.dtype == np.int32:
            data = data.astype(np.float32) / 2147483648.0
        elif data.dtype in (np.float32, np.float64):
            data = data.astype(np.float32)
        else:
            raise BinarizationError(f"{w}: unsupported sample type {data.dtype}")
        return data if data.ndim == 1 else data.mean(1)
    return np.asarray(w, dtype=np.float32).reshape(-1)


def align_f0_to_mel(f0_raw, n_frames, hop_size):
    """get_pitch's length arithmetic (data_gen_utils.py:172-183).  Raises like np.pad does when the track is too long."""
    if hop_size == 128:
        pad_size = 4
    elif hop_size == 256:
        pad_size = 2
    else:
        raise AssertionError("hop_size must be 128 or 256")
    f0 = np.asarray(f0_raw, dtype=np.float64)
    lpad = pad_size * 2
    rpad = n_frames - len(f0) - lpad
    if rpad < 0:
        raise ValueError("index can't contain negative values")         # (what np.pad raises in the reference: item skipped)
    f0 = np.pad(f0, [[lpad, rpad]], mode="constant")
    delta_l = n_frames - len(f0)
    assert abs(delta_l) <= 8
    if delta_l > 0:
        f0 = np.concatenate([f0, [f0[-1]] * delta_l], 0)
    return f0[:n_frames]


class ParaBinarizer:
    def __init__(self, hparams, device, f0_fn=None, batch_pairs=32, binarization_args=None):
        self.hp, self.device = hparams, torch.device(device)
        self.f0_fn = f0_fn or praat_f0
        self.batch_pairs = int(batch_pairs)
        self.args = dict(hparams.get("binarization_args", {}) if binarization_args is None else binarization_args)
        # the reference's wav2spec also applies these when configured (data_gen_utils.py:98-121: pyloudnorm to -22 LUFS + peak
        # clamp, silence trimming); this path does not -- refuse instead of writing a dataset that silently differs
        for key in ("loud_norm", "trim_long_sil"):
            if hparams.get(key):
                raise NotImplementedError(f"ParaBinarizer: hparams['{key}'] is not supported by the batched wav2spec "
                                          f"(the shipped vae_global_mle_eng config has it off)")
        self.fe = MelFrontend(hparams, self.device)
        self.eps = float(hparams.get("wav2spec_eps", 1e-10))

    # ---- wav2spec for a ragged batch: one STFT+mel launch -----------------------------------------------------------------
    def wav2spec_batch(self, wavs):
        """wavs: list of float32 arrays.  -> [(wav padded/cut to F*hop, mel [F,80] float32)], F = 1 + len // hop.
        The clips are zero-padded to the longest one: librosa.stft(center=True, pad_mode='constant') sees zeros beyond a
        clip's end as well, so the first F frames of a padded row are that clip's own frames."""
        hop = self.fe.hop
        n = [len(w) for w in wavs]
        buf = np.zeros((len(wavs), max(n)), np.float32)
        for i, w in enumerate(wavs):
            buf[i, :n[i]] = w
        mel = K.stft_mel(torch.from_numpy(buf).to(self.device), self.fe.window, self.fe.basis, self.fe.n_fft, hop, 0, self.eps)
        mel = mel.cpu().numpy()
        out = []
        for i, w in enumerate(wavs):
            F = 1 + n[i] // hop
            r_pad = (n[i] // hop + 1) * hop - n[i]
            out.append((np.pad(w, (0, r_pad), mode="constant", constant_values=0.0)[:F * hop], np.ascontiguousarray(mel[i, :F])))
        return out

    # ---- process_item for a list of pairs ---------------------------------------------------------------------------------
    def process_items(self, metas):
        """metas: dicts with item_name, wav_fn (path or array), spk_id, profwavfn (path or array), item_names (the set's names).
        Returns one item dict (reference layout and key order) or None (skipped, as the reference skips) per meta."""
        out = []
        for s in range(0, len(metas), self.batch_pairs):
            out += self._process_batch(metas[s:s + self.batch_pairs])
        return out

    def _process_batch(self, metas):
        hp = self.hp
        sr, hop = hp["audio_sample_rate"], hp["hop_size"]
        wavs = []
        for m in metas:
            wavs += [_read_wav(m["wav_fn"], sr), _read_wav(m["profwavfn"], sr)]
        specs = self.wav2spec_batch(wavs)
        res_all, live = [], []
        for i, m in enumerate(metas):
            (wav, mel), (pwav, pmel) = specs[2 * i], specs[2 * i + 1]
            res = {"item_name": m["item_name"], "wav_fn": m["wav_fn"] if isinstance(m["wav_fn"], str) else m["item_name"],
                   "spk_id": m["spk_id"],
                   "a2profwavfn": m["profwavfn"] if isinstance(m["profwavfn"], str) else m["item_name"] + "#prof"}
            gap = hp.get("max_mel_tech_gap")
            if gap is not None and abs(mel.shape[0] - pmel.shape[0]) > gap:
                os.makedirs(hp["binary_data_dir"], exist_ok=True)
                with open(hp["binary_data_dir"] + "/bad_case.txt", "a+") as wf:
                    wf.write("Gap is too large: " + m["item_name"] + str(mel.shape) + str(pmel.shape) + "\n")
                res_all.append(None)
                continue
            res.update({"mel": mel, "wav": wav, "prof_mel": pmel, "prof_wav": pwav, "sec": len(wav) / sr, "len": mel.shape[0],
                        "prof_sec": len(pwav) / sr, "prof_len": pmel.shape[0]})
            if self.args.get("with_f0", True):
                try:
                    for prefix in ("", "prof_"):
                        f0 = align_f0_to_mel(self.f0_fn(res[prefix + "wav"], hp), len(res[prefix + "mel"]), hop)
                        if sum(f0) == 0:
                            raise BinarizationError("Empty f0")
                        res[prefix + "f0"] = f0
                except F0ExtractorMissing:
                    raise
                except BinarizationError as e:
                    print(f"| Skip item ({e}). item_name: {m['item_name']}")
                    res_all.append(None)
                    continue
                except Exception:
                    print(f"| Skip item. item_name: {m['item_name']}")
                    res_all.append(None)
                    continue
            res_all.append(res)
            live.append(len(res_all) - 1)
        if live and self.args.get("with_f0", True):
            # pitch bins of every surviving track in one launch (bit-exact integers), then the F0 alignment of every pair
            tracks = [res_all[j][p + "f0"] for j in live for p in ("", "prof_")]
            flat = torch.from_numpy(np.concatenate(tracks)).to(self.device)
            coarse = K.f0_to_coarse(flat).cpu().numpy()
            off = 0
            for j in live:
                for p in ("", "prof_"):
                    n = len(res_all[j][p + "f0"])
                    res_all[j][p + "pitch"] = coarse[off:off + n].astype(np.int64)
                    off += n
                # (insertion order of the reference: f0, pitch, prof_f0, prof_pitch)
                r = res_all[j]
                for k in ("f0", "pitch", "prof_f0", "prof_pitch"):
                    r[k] = r.pop(k)
            aligns = ehsadtw_batch([res_all[j]["f0"] for j in live], [res_all[j]["prof_f0"] for j in live], self.device)
            for j, al in zip(live, aligns):
                res_all[j]["a2p_f0_alignment"] = [int(a) for a in al]
        # speaker embeddings of the item and of shuffled pieces of the same song (binarize_para.py:234-260)
        for i, m in enumerate(metas):
            res = res_all[i]
            if hp.get("spk_emb_num") is None:
                continue
            name = m["item_name"]
            song_name = name[:-re.search(r"_", name[::-1]).span()[0]]
            pieces = [s for s in m["item_names"] if song_name in s]
            random.shuffle(pieces)               # (the reference shuffles before it looks at the item: skipped items draw too)
            sel = pieces[:hp["spk_emb_num"]]
            if res is None:
                continue
            try:
                embs = [np.load(os.path.join(hp["spk_emb_data_dir"], name + ".npy"), allow_pickle=True)]
                for k in range(hp["spk_emb_num"]):
                    embs.append(np.load(os.path.join(hp["spk_emb_data_dir"], (sel[-1] if k >= len(sel) else sel[k]) + ".npy"),
                                        allow_pickle=True))
                res["multi_spk_emb"] = np.stack(embs, axis=0)
            except Exception:
                print(f"| Skip item. item_name: {name}")
                res_all[i] = None
        return res_all

    # ---- process_data: items -> <prefix>.data/.idx + the side files -------------------------------------------------------
    def process_data(self, prefix, metas, data_dir=None):
        data_dir = data_dir or self.hp["binary_data_dir"]
        os.makedirs(data_dir, exist_ok=True)
        builder = IndexedDatasetBuilder(f"{data_dir}/{prefix}")
        mel_lengths, f0s, total_sec, n = [], [], 0.0, 0
        for item in self.process_items(metas):
            if item is None:
                continue
            item["spk_embed"] = None                     # (with_spk_embed: the Resemblyzer encoder is outside this path)
            if not self.args.get("with_wav", False) and "wav" in item:
                del item["wav"]
                del item["prof_wav"]
            builder.add_item(item)
            mel_lengths.append(max(item["len"], item["prof_len"]))
            total_sec += item["sec"]
            if item.get("f0") is not None:
                f0s += [item["f0"], item["prof_f0"]]
            n += 1
        builder.finalize()
        np.save(f"{data_dir}/{prefix}_lengths.npy", mel_lengths)
        if f0s:
            f = np.concatenate(f0s, 0)
            f = f[f != 0]
            np.save(f"{data_dir}/{prefix}_f0s_mean_std.npy", [np.mean(f).item(), np.std(f).item()])
        print(f"| {prefix} total duration: {total_sec:.3f}s")
        return n
