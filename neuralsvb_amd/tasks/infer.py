"""`--infer`: PPG + pitch -> VAE mel (ways a2a, p2p, a2p) -> NSF-HifiGAN waveform -> files.

Mirror of reference tasks/singing/svb_vae_task.py:302-381 (`test_step`, `after_infer`) and tasks/singing/svb_para.py:338-353
(`save_result`): same output tree
  checkpoints/<exp>/generated_<global_step>_<gen_dir_name>/{wavs,mels}/[disable_map_]<key>/[000000][<item>][P]<text>.{wav,npy}
with <key> in gt_a, gt_p, a2a, p2p, a2p (`_wavout` / `_mel`), int16 wavs (utils/audio.py:11-16), running index reset on
every call (:325).  The reference asserts batch size 1; here the whole batch goes through the VAE and the vocoder
at once (BASELINE config #5) and is then written item by item.
"""
import os

import numpy as np
import torch
from scipy.io import wavfile

from ..utils.hparams import hparams
from ..utils.pitch_utils import denorm_f0
from ..vocoders.base_vocoder import get_vocoder_cls


def save_wav(wav, path, sr, norm=False):
    wav = np.asarray(wav, dtype=np.float32)
    if norm:
        wav = wav / np.abs(wav).max()
    wavfile.write(path, sr, (wav * 32767).astype(np.int16))


@torch.no_grad()
def synthesize(task, sample):
    """-> dict of numpy batches: '<key>_mel' [B,T,80] and '<key>_wavout' [B, T*hop] for gt_a, gt_p, a2a, p2p, a2p."""
    if task.vocoder is None:
        task.vocoder = get_vocoder_cls(hparams)()
    ways = ["a2a", "p2p", "a2p"]
    _, out = task.run_model(task.model, sample, concurrent_ways=ways, return_output=True, infer=True,
                            disable_map=hparams["disable_map"])
    f0_p = denorm_f0(sample["prof_f0"], sample["prof_uv"], hparams)
    f0_a = denorm_f0(sample["f0"], sample["uv"], hparams)
    f0s = {"gt_a": f0_a, "gt_p": f0_p, "a2a": f0_a, "p2p": f0_p, "a2p": f0_p}
    mels = {"gt_a": sample["mels"], "gt_p": sample["prof_mels"]}
    mels.update({w: out[w]["mel_out"] for w in ways})
    res = {}
    for key, mel in mels.items():
        res[f"{key}_mel"] = mel.detach().cpu().numpy()
        res[f"{key}_wavout"] = task.vocoder.spec2wav(mel.detach().contiguous(), f0=f0s[key])
    return res


def infer_and_save(task, sample, batch_idx):
    res = synthesize(task, sample)
    gen_dir = os.path.join(hparams["work_dir"], f'generated_{task.trainer.global_step}_{hparams["gen_dir_name"]}')
    pre = "disable_map_" if hparams["disable_map"] else ""
    lens_a, lens_p = sample["mel_lengths"].tolist(), sample["prof_mel_lengths"].tolist()
    hop = hparams["hop_size"]
    for i, item_name in enumerate(sample["item_name"]):
        text = sample["text"][i]
        base = f"[{0:06d}][{item_name}][P]" + (text.replace(":", "%3A")[:80] if text is not None else "")
        base = base.replace(" ", "_")
        for key in ("gt_a", "gt_p", "a2a", "p2p", "a2p"):
            n = lens_a[i] if key in ("gt_a", "a2a") else lens_p[i]
            for kind, sub, ext in (("wavout", "wavs", "wav"), ("mel", "mels", "npy")):
                d = f"{gen_dir}/{sub}/{pre}{key}_{kind}"
                os.makedirs(d, exist_ok=True)
                if kind == "wavout":
                    save_wav(res[f"{key}_wavout"][i][:n * hop], f"{d}/{base}.wav", hparams["audio_sample_rate"],
                             norm=hparams["out_wav_norm"])
                else:
                    np.save(f"{d}/{base}.npy", res[f"{key}_mel"][i][:n])
    return {"item_name": sample["item_name"], "text": sample["text"]}
