"""the batched binarizer driver (neuralsvb_amd/data_gen/binarizer.py: svb_stft_mel -> f0 alignment arithmetic ->
svb_f0_to_coarse -> batched F0 DTW -> IndexedDataset writer) against the UNMODIFIED reference's
`PopBuTFyENSpkEMBinarizer.process_item` run on the same synthetic pairs (tests/golden/make_binarizer_golden.py ->
tests/golden/binarizer_ref.json), then through process_data and back through the task's dataset class."""
import json
import os
import random
import sys

import numpy as np
import pytest
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
sys.path.insert(0, G)


def _setup(tmp_path):
    import binarizer_common as BC
    ref = json.load(open(os.path.join(G, "binarizer_ref.json")))
    hp = dict(ref["hparams"])
    hp.update(binary_data_dir=str(tmp_path / "binary"), spk_emb_data_dir=str(tmp_path / "spk_emb"),
              binarization_args={"with_f0": True, "with_wav": True})
    os.makedirs(hp["spk_emb_data_dir"])
    pairs = BC.make_pairs(hp["audio_sample_rate"])
    names = [n for n, _, _ in pairs]
    for n in names:
        np.save(os.path.join(hp["spk_emb_data_dir"], n + ".npy"), BC.spk_embedding(n))
    metas = [{"item_name": n, "wav_fn": a, "spk_id": i % 3, "profwavfn": p, "item_names": names} for i, (n, a, p) in enumerate(pairs)]
    f0_fn = lambda wav, h: BC.fake_extractor_f0(wav, h["hop_size"], h["audio_sample_rate"])      # noqa: E731
    return BC, ref, hp, metas, f0_fn


def test_batched_binarizer_items_match_reference_process_item(dev, tmp_path):
    from neuralsvb_amd.data_gen.binarizer import ParaBinarizer
    BC, ref, hp, metas, f0_fn = _setup(tmp_path)
    random.seed(BC.SHUFFLE_SEED)
    items = ParaBinarizer(hp, dev, f0_fn=f0_fn, batch_pairs=3).process_items(metas)      # (7 pairs: batches of 3, 3, 1)
    assert len(items) == len(ref["items"])
    assert [it is None for it in items] == [r is None for r in ref["items"]]              # same items skipped (gap, empty f0)
    assert open(os.path.join(hp["binary_data_dir"], "bad_case.txt")).read() == ref["bad_case"]
    for it, r in zip(items, ref["items"]):
        if r is None:
            continue
        d = BC.digest(it)
        # layout: same keys in the same order, same dtypes / container types
        for k in ("keys", "item_name", "len", "prof_len", "spk_id", "f0_dtype", "pitch_dtype", "alignment_type",
                  "multi_spk_emb_dtype", "wav_len"):
            assert d[k] == r[k], (it["item_name"], k, d[k], r[k])
        assert d["sec"] == r["sec"] and d["prof_sec"] == r["prof_sec"]
        # bit-exact: frame counts, padded F0 tracks, pitch bins, alignment, embedding pick, the padded / cut wav
        for k in ("f0", "prof_f0", "pitch", "prof_pitch", "a2p_f0_alignment", "multi_spk_emb"):
            assert d[k] == r[k], (it["item_name"], k)
        assert abs(d["wav_sum"] - r["wav_sum"]) == 0.0
        # log10-mel: the HIP STFT/mel kernel against the numpy restatement the reference ran with (|mel| ~ 1..10)
        for k in ("mel", "prof_mel"):
            assert d[k]["shape"] == r[k]["shape"] and d[k]["dtype"] == r[k]["dtype"]
            assert np.abs(np.array(d[k]["samples"]) - np.array(r[k]["samples"])).max() < 2e-4, (it["item_name"], k)
            assert abs(d[k]["sum"] - r[k]["sum"]) < 2e-5 * r[k]["abs"], (it["item_name"], k)


def test_batched_binarizer_process_data_round_trip(dev, tmp_path):
    """process_data: <prefix>.data/.idx in the reference's IndexedDataset format + <prefix>_lengths.npy / _f0s_mean_std.npy
    (binarize_para.py:116-165), read back by the reader and collated by the task's dataset class."""
    from neuralsvb_amd.data_gen.binarizer import ParaBinarizer
    from neuralsvb_amd.utils.indexed_datasets import IndexedDataset
    BC, ref, hp, metas, f0_fn = _setup(tmp_path)
    hp["binarization_args"]["with_wav"] = False
    random.seed(BC.SHUFFLE_SEED)
    n = ParaBinarizer(hp, dev, f0_fn=f0_fn).process_data("train", metas)
    live = [r for r in ref["items"] if r is not None]
    assert n == len(live)
    ds = IndexedDataset(os.path.join(hp["binary_data_dir"], "train"))
    assert len(ds) == n
    lens = np.load(os.path.join(hp["binary_data_dir"], "train_lengths.npy"))
    assert list(lens) == [max(r["len"], r["prof_len"]) for r in live]
    f0s = np.concatenate([np.array(r[k]) for r in live for k in ("f0", "prof_f0")])
    f0s = f0s[f0s != 0]
    ms = np.load(os.path.join(hp["binary_data_dir"], "train_f0s_mean_std.npy"))
    assert np.allclose(ms, [f0s.mean(), f0s.std()], rtol=0, atol=1e-12)
    for i, r in enumerate(live):
        it = ds[i]
        assert "wav" not in it and "prof_wav" not in it and it["spk_embed"] is None
        assert it["item_name"] == r["item_name"] and it["a2p_f0_alignment"] == r["a2p_f0_alignment"]
        assert np.asarray(it["pitch"]).tolist() == r["pitch"]


def test_f0_length_arithmetic_edge_cases():
    """get_pitch's padding (data_gen_utils.py:172-183): 2*pad_size frames on the left, the rest on the right; a track longer
    than the mel raises (np.pad with a negative width) and the item is skipped."""
    from neuralsvb_amd.data_gen.binarizer import align_f0_to_mel
    f = np.arange(1.0, 11.0)
    out = align_f0_to_mel(f, 25, 128)
    assert out.shape == (25,) and out[:8].tolist() == [0.0] * 8 and out[8:18].tolist() == f.tolist() and out[18:].tolist() == [0.0] * 7
    assert align_f0_to_mel(f, 14, 256)[:4].tolist() == [0.0] * 4
    with pytest.raises(ValueError):
        align_f0_to_mel(f, 17, 128)
    with pytest.raises(AssertionError):
        align_f0_to_mel(f, 30, 100)
