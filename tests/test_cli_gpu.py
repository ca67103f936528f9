"""End-to-end drop-in check on the MI355X: the reference's CLI surface (`python tasks/run.py --config … --exp_name …`)
trains a few steps on a synthetic binary dataset, writes reference-layout checkpoints, resumes, and `--infer` writes the
reference's output tree through the NSF-HifiGAN vocoder plugin."""
import glob
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ("hidden_size=256,fvae_enc_dec_hidden=64,latent_size=16,fvae_enc_n_layers=2,fvae_dec_n_layers=2,"
         "mel_disc_hidden_size=32,max_sentences=2,ds_workers=0,num_sanity_val_steps=1,endless_ds=False,"
         "audio_sample_rate=24000,fmax=12000,max_updates=3,val_check_interval=2,phase_2_steps=2,tb_log_interval=1,"
         "max_valid_sentences=1")


@pytest.mark.gpu
def test_run_py_train_resume_infer(gpu_only, tmp_path):
    sys.path.insert(0, ROOT)
    from neuralsvb_amd.utils.hparams import set_hparams, hparams
    from neuralsvb_amd.utils import synth
    from neuralsvb_amd.modules.hifigan import HifiGanGenerator
    cfg = os.path.join(ROOT, "egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml")
    set_hparams(config=cfg, exp_name="", hparams_str=SMALL, print_hparams=False)
    data_dir, asr_dir, voc_dir = (str(tmp_path / d) for d in ("data/binary/synth", "checkpoints/asr", "checkpoints/voc"))
    synth.write_binary_dataset(data_dir, hparams, synth.mel_fn_hip(hparams, gpu_only), n_train=4, n_valid=2, seconds=1.1)
    synth.write_fake_asr_ckpt(asr_dir, 70, hparams)
    vcfg = {"resblock": "1", "upsample_rates": [8, 4, 2, 2], "upsample_kernel_sizes": [16, 8, 4, 4],
            "upsample_initial_channel": 32, "resblock_kernel_sizes": [3, 7, 11],
            "resblock_dilation_sizes": [[1, 3, 5], [1, 3, 5], [1, 3, 5]], "use_pitch_embed": True,
            "audio_sample_rate": 24000, "hop_size": 128}
    os.makedirs(voc_dir)
    yaml.safe_dump(vcfg, open(os.path.join(voc_dir, "config.yaml"), "w"))
    torch.manual_seed(0)
    torch.save({"state_dict": {"model_gen": HifiGanGenerator(vcfg).state_dict()}}, os.path.join(voc_dir, "model_ckpt_steps_1.ckpt"))
    hp = SMALL + f",binary_data_dir={data_dir},pretrain_asr_ckpt={asr_dir},vocoder_ckpt={voc_dir}"
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", PYTHONPATH=ROOT)
    base = [sys.executable, os.path.join(ROOT, "tasks/run.py"), "--config", cfg, "--exp_name", "t1"]

    r = subprocess.run(base + ["--reset", "--hparams", hp], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    ckpts = sorted(glob.glob(str(tmp_path / "checkpoints/t1/model_ckpt_steps_*.ckpt")))
    assert ckpts, r.stdout[-2000:]
    ck = torch.load(ckpts[-1], map_location="cpu", weights_only=False)
    assert set(ck["state_dict"]) == {"model", "mel_disc"} and len(ck["optimizer_states"]) == 3
    assert os.path.exists(tmp_path / "checkpoints/t1/config.yaml")
    step0 = ck["global_step"]

    # resume: picks the newest checkpoint up and continues to a larger max_updates
    r = subprocess.run(base + ["--hparams", hp.replace("max_updates=3", "max_updates=5")], cwd=tmp_path, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    newest = max(int(os.path.basename(p).split("_")[-1].split(".")[0]) for p in glob.glob(str(tmp_path / "checkpoints/t1/model_ckpt_steps_*.ckpt")))
    assert newest > step0

    r = subprocess.run(base + ["--infer", "--hparams", hp], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    gen = glob.glob(str(tmp_path / "checkpoints/t1/generated_*"))
    assert len(gen) == 1
    for key in ("gt_a", "gt_p", "a2a", "p2p", "a2p"):
        assert glob.glob(os.path.join(gen[0], "wavs", f"{key}_wavout", "*.wav")), key
        assert glob.glob(os.path.join(gen[0], "mels", f"{key}_mel", "*.npy")), key


@pytest.mark.gpu
def test_run_py_variable_length_batches_never_measure_tiles(gpu_only, tmp_path):
    """The reference's loader batches length-sorted clips by a token budget (utils/__init__.py:163-217, tasks/tts/tts.py:57-101):
    every batch of a real run has its own (B, T).  `tasks/run.py` at the REAL channel dimensions on a synthetic set of 24 clips of
    24 distinct lengths (0.7 ... 3.0 s), `max_tokens`-bounded batches, 50 optimizer steps: the conv launches of every batch shape
    take their tiles from the committed table -- exact hits or the nearest entry of their launch family -- and NOTHING is measured
    inside the run (round 5 measured 17 configurations x 11 launches with an event synchronise per unseen signature)."""
    import json
    sys.path.insert(0, ROOT)
    from neuralsvb_amd.utils.hparams import set_hparams, hparams
    from neuralsvb_amd.utils import synth
    cfg = os.path.join(ROOT, "egs/datasets/audio/PopBuTFy/vae_global_mle_eng.yaml")
    base_hp = ("max_sentences=6,max_tokens=900,ds_workers=0,num_sanity_val_steps=0,endless_ds=True,audio_sample_rate=24000,fmax=12000,"
               "max_updates=50,val_check_interval=100000,phase_2_steps=100000,tb_log_interval=1000,max_valid_sentences=1,"
               "conv_precision=bf16x3,sort_by_len=True")
    set_hparams(config=cfg, exp_name="", hparams_str=base_hp, print_hparams=False)
    data_dir, asr_dir = str(tmp_path / "data/binary/synth"), str(tmp_path / "checkpoints/asr")
    secs = [0.7 + 0.1 * i for i in range(24)]
    synth.write_binary_dataset(data_dir, hparams, synth.mel_fn_hip(hparams, gpu_only), n_train=24, n_valid=1, seconds=secs)
    lens = np.load(os.path.join(data_dir, "train_lengths.npy"))
    assert len(set(int(v) for v in lens)) >= 20
    synth.write_fake_asr_ckpt(asr_dir, 70, hparams)
    hp = base_hp + f",binary_data_dir={data_dir},pretrain_asr_ckpt={asr_dir}"
    env = dict(os.environ, CUDA_VISIBLE_DEVICES="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tasks/run.py"), "--config", cfg, "--exp_name", "vl", "--reset",
                        "--hparams", hp], cwd=tmp_path, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    info = json.load(open(tmp_path / "checkpoints/vl/tile_table_info.json"))
    assert info["global_step"] > 50 and info["entries"] > 300
    assert info["online_tuned_signatures"] == 0, info
    assert info["nearest_bucket_signatures"] >= 40, info          # many batch shapes x ~30 conv signatures each, none in the table
