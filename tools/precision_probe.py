"""Where does the bf16x3 split's rounding noise enter the generated mels?  Runs the golden MleSVBVAE forward (reference
goldens tests/golden/vae_mle.npz [B=2,T=64] and vae_mle_b16.npz [B=16,T=1124]) with parts of the model pinned to fp32
MFMA and prints the per-way mel-L1 against the reference.  GPU tool:  python tools/precision_probe.py"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import procedural  # noqa: E402  (tool: compares against reference goldens)

G = os.path.join(ROOT, "tests", "golden")
KEYS = json.load(open(os.path.join(G, "ref_state_keys.json")))
HP = json.load(open(os.path.join(G, "ref_hparams_vae_global_mle_eng.json")))
PARTS = ["vc_asr", "pitch_encoder", "upsample_layer", "encoded_embed_proj", "vae_model.g_pre_net", "vae_model.encoder.pre_net",
         "vae_model.encoder.wn", "vae_model.encoder.out_proj", "vae_model.encoder.poolings", "vae_model.decoder.pre_net",
         "vae_model.decoder.wn", "vae_model.decoder.out_proj"]


def t(a):
    return torch.from_numpy(np.asarray(a))


def main():
    from neuralsvb_amd import functional as SF
    from neuralsvb_amd.modules.layers import set_layer_precision
    from neuralsvb_amd.modules.svb_vae import MleSVBVAE
    dev = torch.device("cuda:0")
    model = MleSVBVAE(70, HP)
    model.load_state_dict(procedural.state_dict_for(KEYS["MleSVBVAE"], prefix="model."), strict=True)
    model.to(dev).train()
    cases = {}
    d = np.load(os.path.join(G, "vae_mle.npz"))
    cases["B2xT64"] = ({k: t(d[k]).to(dev) for k in ("mels", "prof_mels", "pitch", "prof_pitch", "spk", "a2p_alignment",
                                                      "eps_a2a", "eps_p2p")},
                       {w: t(d[f"{w}.mel_out"]) for w in ("a2a", "p2p", "a2p")}, 1)
    fb = os.path.join(G, "vae_mle_b16.npz")
    if os.path.exists(fb):
        import make_golden as M
        d16 = np.load(fb)
        inp = M.make_vae_inputs(B=16, T=1124, lens=tuple(int(x) for x in d16["lens"]), seed=21)
        inp["eps_a2a"], inp["eps_p2p"] = t(d16["eps_a2a"]), t(d16["eps_p2p"])
        cases["B16xT1124"] = ({k: v.to(dev) for k, v in inp.items()}, {w: t(d16[f"{w}.mel_out"]) for w in ("a2a", "p2p", "a2p")},
                              int(d16["frame_stride"]))

    def run(inp, ref, stride):
        with torch.no_grad():
            out = model(amateur_mel=inp["mels"], prof_mel=inp["prof_mels"], amateur_pitch=inp["pitch"],
                        prof_pitch=inp["prof_pitch"], amateur_spk_id=inp["spk"], prof_spk_id=inp["spk"],
                        a2p_alignment=inp["a2p_alignment"], infer=False, concurrent_ways=["a2a", "p2p", "a2p"],
                        eps_a2a=inp["eps_a2a"], eps_p2p=inp["eps_p2p"])
        return {w: (out[w]["mel_out"][:, ::stride].cpu() - ref[w]).abs().mean().item() for w in ref}

    def show(tag, r):
        print(f"{tag:54s} " + "  ".join(f"{w} {v:.2e}" for w, v in r.items()), flush=True)

    for cname, (inp, ref, stride) in cases.items():
        print(f"==== {cname}")
        for glob in ("fp32", "bf16x3"):
            SF.set_precision(glob)
            set_layer_precision(model, {"": None})
            show(f"all {glob}", run(inp, ref, stride))
        SF.set_precision("bf16x3")
        for part in PARTS:
            set_layer_precision(model, {"": None, part: "fp32"})
            show(f"bf16x3, fp32 in {part}", run(inp, ref, stride))
        for combo in (["vc_asr", "upsample_layer"], ["vc_asr", "upsample_layer", "pitch_encoder", "encoded_embed_proj"],
                      ["vae_model.encoder.out_proj", "vae_model.encoder.poolings"],
                      ["vae_model.encoder.wn", "vae_model.encoder.out_proj", "vae_model.encoder.poolings", "vae_model.encoder.pre_net",
                       "vae_model.g_pre_net"]):
            rules = {"": None}
            rules.update({p: "fp32" for p in combo})
            set_layer_precision(model, rules)
            show("bf16x3, fp32 in " + "+".join(c.split(".")[-1] for c in combo), run(inp, ref, stride))
        SF.set_precision("fp32")
        for part in PARTS:
            set_layer_precision(model, {"": None, part: "bf16x3"})
            show(f"fp32, bf16x3 in {part}", run(inp, ref, stride))
        set_layer_precision(model, {"": None})
    SF.set_precision("fp32")


if __name__ == "__main__":
    main()
