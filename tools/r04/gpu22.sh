#!/bin/bash
# round 4, GPU call 22: BatchNorm kernel with double accumulation: the gradient test's worst element, then the whole suite + smoke
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 600 python - > gpurun_out/r04_g22_vae_grad.log 2>&1 <<'PY'
import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import test_modules_vae as T
from oracle import modules_ref as R
dev = torch.device("cuda:0")
d = np.load(os.path.join(T.G, "vae_mle.npz"))
model, sd = T.build_model(dev); model.train()
t = T.t
args = [t(d[k]) for k in ("mels", "prof_mels", "pitch", "prof_pitch", "spk", "a2p_alignment")]
sdr = {k: (v.clone().requires_grad_(True) if v.dtype == torch.float32 and not k.startswith("vc_asr") else v) for k, v in sd.items()}
ret, _, _ = R.mle_svb_vae(sdr, *args, ["a2a", "p2p"], t(d["eps_a2a"]), t(d["eps_p2p"]), T.HP, training=True)
loss_r = sum(ret[w]["kl"] * 0.001 + R.l1_loss(ret[w]["mel_out"], tg) for w, tg in (("a2a", args[0]), ("p2p", args[1])))
loss_r.backward()
out = model(amateur_mel=args[0].to(dev), prof_mel=args[1].to(dev), amateur_pitch=args[2].to(dev), prof_pitch=args[3].to(dev),
            amateur_spk_id=args[4].to(dev), prof_spk_id=args[4].to(dev), a2p_alignment=args[5].to(dev), infer=False,
            concurrent_ways=["a2a", "p2p"], eps_a2a=t(d["eps_a2a"]).to(dev), eps_p2p=t(d["eps_p2p"]).to(dev))
loss = sum(out[w]["kl"] * 0.001 + R.l1_loss(out[w]["mel_out"], tg.to(dev)) for w, tg in (("a2a", args[0]), ("p2p", args[1])))
loss.backward()
rows = []
for k, p in model.named_parameters():
    if k.startswith("vc_asr") or k.startswith("z_mapping_function"):
        continue
    gr = sdr[k].grad
    rows.append((((p.grad.cpu() - gr).abs().max() / gr.abs().max().clamp_min(1e-8)).item(), k))
rows.sort(reverse=True)
print("loss", loss.item(), loss_r.item())
for r, k in rows[:8]:
    print(f"{r:.3e}  {k}")
PY
grep -v "Warn\|warn\|amdgpu" gpurun_out/r04_g22_vae_grad.log | tail -10
timeout 600 python -m pytest tests/test_modules_vae.py -q -m gpu -s -k "bench_shape_gradients" 2>&1 | grep -E "norm|worst|passed|failed" > gpurun_out/r04_g22_bench_shape_grads.log; cat gpurun_out/r04_g22_bench_shape_grads.log | cut -c1-150
