"""Time the oracle CPU port of the training step for several thread counts (picks bench.py's --cpu-threads default)."""
import json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import procedural
from oracle.train_step_ref import CpuStep
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
KEYS = json.load(open(os.path.join(G, "ref_state_keys.json")))
HP = json.load(open(os.path.join(G, "ref_hparams_vae_global_mle_eng.json")))
B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 4, 1124
g = torch.Generator().manual_seed(0)
sample = {"mels": torch.randn(B, T, 80, generator=g) - 3, "prof_mels": torch.randn(B, T, 80, generator=g) - 3,
          "pitch": torch.randint(1, 255, (B, T), generator=g), "prof_pitch": torch.randint(1, 255, (B, T), generator=g),
          "multi_spk_emb": torch.randn(B, 5, 256, generator=g) / 16, "a2p_f0_alignment": torch.arange(T)[None].repeat(B, 1)}
msd = procedural.state_dict_for(KEYS["MleSVBVAE"], prefix="model.")
dsd = procedural.state_dict_for(KEYS["Discriminator"], prefix="mel_disc.")
st = {w: [[5, 5], [9, 9], [3, 3]] for w in ("a2a", "p2p")}
sd = {w: {"real": st[w], "fake": st[w]} for w in ("a2a", "p2p")}
for nt in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "16,32,64,128").split(",")]:
    torch.set_num_threads(nt)
    step = CpuStep(msd, dsd, HP)
    ts = []
    for i in range(3):
        eps = [torch.randn(B, 128, 1, generator=g) for _ in range(2)]
        t0 = time.perf_counter(); step.step(sample, 1, eps[0], eps[1], st, sd, global_step=1 + i); ts.append(time.perf_counter() - t0)
    print(json.dumps({"threads": nt, "B": B, "s_per_step": [round(t, 3) for t in ts], "audio_s_per_s": round(B * 5.995 / min(ts[1:]), 2)}), flush=True)
