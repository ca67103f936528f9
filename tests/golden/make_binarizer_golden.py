"""Generate tests/golden/binarizer_ref.json by running the UNMODIFIED reference binarizer
(`PopBuTFyENSpkEMBinarizer.process_item`, data_gen/singing/binarize_para.py:187-260 with everything it calls:
`PWG.wav2spec` -> `process_utterance`, `get_pitch`, `f0_to_coarse`, `EHSADTW`, the speaker-embedding pick) on synthetic pairs.

Build-container only.  Usage:  python tests/golden/make_binarizer_golden.py
Third-party pieces the reference imports and this container lacks are supplied at run time (no reference source is modified):
`librosa.stft` / `librosa.filters.mel` = the oracle's restatements of librosa 0.8.0 (oracle/frontend.py: parity UNPINNED for that
arithmetic, as everywhere), `parselmouth.Sound(...).to_pitch_ac(...)` = binarizer_common.fake_extractor_f0 (a deterministic
function of the audio; both sides use it).  Everything else -- padding, trimming, frame counts, bins, alignment, embedding pick,
item layout -- is the reference's own code.
"""
import json
import os
import random
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import binarizer_common as BC  # noqa: E402
from oracle import frontend as ofe  # noqa: E402
from oracle import ref_shims  # noqa: E402


def main():
    ref_shims.install()
    from utils.hparams import set_hparams, hparams
    cwd = os.getcwd()
    os.chdir(ref_shims.REFERENCE_ROOT)
    try:
        set_hparams(config="egs/datasets/audio/PopBuTFy/para_bin.yaml", exp_name="", print_hparams=False)
    finally:
        os.chdir(cwd)
    tmp = tempfile.mkdtemp(prefix="bingold_")
    hparams.update(BC.OVERRIDES)
    hparams.update(binary_data_dir=os.path.join(tmp, "binary"), spk_emb_data_dir=os.path.join(tmp, "spk_emb"))
    os.makedirs(hparams["spk_emb_data_dir"])
    os.makedirs(hparams["binary_data_dir"])
    sr, hop = hparams["audio_sample_rate"], hparams["hop_size"]

    # ---- run-time stand-ins for the absent third-party packages ------------------------------------------------------------
    import librosa
    librosa.stft = ofe.librosa_stft
    librosa.filters.mel = ofe.librosa_mel_filterbank
    import parselmouth

    class _Pitch:
        def __init__(self, f0):
            self.selected_array = {"frequency": f0}

    class _Sound:
        def __init__(self, wav, sampling_frequency):
            self.wav, self.sr = wav, sampling_frequency

        def to_pitch_ac(self, time_step=None, **_):
            return _Pitch(BC.fake_extractor_f0(self.wav, int(round(time_step * self.sr)), self.sr))
    parselmouth.Sound = _Sound

    from data_gen.singing.binarize_para import PopBuTFyENSpkEMBinarizer as B
    pairs = BC.make_pairs(sr)
    names = [n for n, _, _ in pairs]
    for n in names:
        np.save(os.path.join(hparams["spk_emb_data_dir"], n + ".npy"), BC.spk_embedding(n))
    args = dict(hparams["binarization_args"])
    args["with_wav"] = True
    random.seed(BC.SHUFFLE_SEED)
    out = {"hparams": {k: hparams[k] for k in ("audio_sample_rate", "hop_size", "fft_size", "win_size", "fmin", "fmax",
                                                "audio_num_mel_bins", "max_mel_tech_gap", "spk_emb_num", "loud_norm")},
           "items": []}
    for i, (name, a, p) in enumerate(pairs):
        item = B.process_item(name, a, i % 3, p, names, args)
        out["items"].append(None if item is None else BC.digest(item))
        print(name, None if item is None else (item["len"], item["prof_len"], len(item["a2p_f0_alignment"])))
    bad = os.path.join(hparams["binary_data_dir"], "bad_case.txt")
    out["bad_case"] = open(bad).read() if os.path.exists(bad) else ""
    with open(os.path.join(HERE, "binarizer_ref.json"), "w") as f:
        json.dump(out, f)
    print("written", os.path.getsize(os.path.join(HERE, "binarizer_ref.json")), "bytes")


if __name__ == "__main__":
    main()
