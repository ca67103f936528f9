"""HBM-side traffic (rocprofv3 PMC) of the step's dominant conv kernel, per shape -> profiles/<round>_pmc_traffic.json (GPU box; round tag from SVB_ROUND, default r06).

As MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), nothing but
--pmc on the rocprofv3 command line, units of KB, and -- because the gfx950 counters are only calibrated for 16-byte streaming
reads -- each counter is calibrated here on a launch of the SAME kernel whose traffic is known by construction (16 input
channels -> one 16-channel chunk, one M tile: every input element is fetched by exactly one workgroup once, every output
element written once; sized beyond the 256 MB Infinity Cache).  bench.py reads the result for `roofline.traffic`.

  python tools/pmc_traffic.py            # ~4 min of GPU time
"""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
CFG = int(os.environ.get("SVB_PMC_CFG", "2"))        # tile configuration of the dominant kernel (2 = 128x96)


def run_pmc(counter, conv_args):
    with tempfile.TemporaryDirectory(dir="/tmp") as d:
        cmd = ["rocprofv3", "--pmc", counter, "-d", d, "--output-format", "csv", "--", sys.executable,
               os.path.join(ROOT, "tools", "pmc_conv.py")] + [str(a) for a in conv_args]
        env = dict(os.environ, TMPDIR="/tmp", SVB_AUTOTUNE="0")
        r = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=300)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            raise RuntimeError(f"rocprofv3 failed ({r.returncode}): {r.stderr[-400:]}")
        vals, name = [], None
        for row in csv.DictReader(open(files[0])):
            if row["Counter_Name"] == counter and "svb_conv1d_" in row["Kernel_Name"] and "wgrad" not in row["Kernel_Name"]:
                vals.append(float(row["Counter_Value"]))
                name = row["Kernel_Name"]
        if not vals:
            raise RuntimeError("no conv dispatch in the counter file")
        vals = vals[1:] if len(vals) > 2 else vals          # first launch: cold caches
        return sum(vals) / len(vals) * 1024.0, name         # KB -> bytes


def main():
    import torch
    import bench
    from neuralsvb_amd import kernels as K
    import argparse
    args = argparse.Namespace(gpus=1, steps=1, warmup=1, batch=16, seconds=6.0, sample_rate=24000, bf16=False,
                              precision="bf16x3", graph=False)
    dev = torch.device("cuda:0")
    with tempfile.TemporaryDirectory() as tmp:
        task, trainer, batch, hp = bench.build_task(args, 0, 1, dev, tmp)
        bench.run_steps(trainer, task, batch, 3, 1)
        K.PROFILE = []
        bench.run_steps(trainer, task, batch, 1, 4)
        torch.cuda.synchronize()
        rec, K.PROFILE = K.PROFILE, None
    by = {}
    for name, fl, e0, e1, tag in rec:
        if not name.startswith("svb_conv1d_wgrad"):
            d = by.setdefault(name, {"t": 0.0, "tags": {}})
            d["t"] += e0.elapsed_time(e1)
            d["tags"][tag] = d["tags"].get(tag, 0) + 1
    dom = max(by, key=lambda n: by[n]["t"])
    tags = sorted(by[dom]["tags"].items(), key=lambda kv: -kv[1])
    print("dominant kernel:", dom, "shapes:", tags, flush=True)
    del task, trainer, batch
    torch.cuda.empty_cache()
    cfg = 1 + [i for i, n in enumerate(K._CFG_NAMES) if n.split("(")[1] == dom.split("(")[1]][0]
    # ---- calibration: known traffic, same kernel, same access pattern
    # (the pointwise kernel's domain is Cin % 64 == 0; its row tile is the first number of the configuration's "(MxN)")
    pw = dom.startswith("svb_conv1d_pw")
    cb, ccin, ccout, cT = 32, (64 if pw else 16), (int(dom.split("(")[1].split("x")[0]) if pw else 128), 32768
    known_fetch = 4.0 * cb * ccin * cT + 4.0 * ccout * ccin
    known_write = 4.0 * cb * ccout * cT
    cal = {}
    for counter, known in (("FETCH_SIZE", known_fetch), ("WRITE_SIZE", known_write)):
        raw, _ = run_pmc(counter, [cb, ccin, ccout, cT, 1, "fwd", cfg])
        cal[counter] = {"raw_bytes": raw, "known_bytes": known, "factor": known / raw}
        print("calibration", counter, cal[counter], flush=True)
    out = {"kernel": dom, "cfg": cfg, "calibration": cal, "method": __doc__.split("\n\n")[1], "shapes": {}}
    for tag, cnt in tags[:int(os.environ.get("SVB_PMC_SHAPES", "16"))]:
        op, B, ca, cb_, G, T, k, s, dil = tag
        if G != 1 or s != 1 or dil != 1 or op == "taps":
            continue
        what = "fwd" if op == "fwd" else "dgrad"
        cin, cout = (ca, cb_) if op == "fwd" else (cb_, ca)      # pmc_conv dgrad: dy [B,Cout,T] -> dx [B,Cin,T]
        ent = {"launches_per_step": cnt, "op": op}
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            raw, kname = run_pmc(counter, [B, cin, cout, T, k, what, cfg])
            ent[counter.lower() + "_raw_bytes"] = raw
            ent[counter.lower() + "_bytes"] = raw * cal[counter]["factor"]
            ent["kernel"] = kname
        ent["hbm_bytes"] = ent["fetch_size_bytes"] + ent["write_size_bytes"]
        ent["algorithmic_bytes"] = bench.conv_alg_bytes(tag)
        out["shapes"][json.dumps(list(tag))] = ent
        print(tag, ent, flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    rnd = os.environ.get("SVB_ROUND", "r06")
    for path in (os.path.join(ROOT, "gpurun_out", f"{rnd}_pmc_traffic.json"), os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")):
        with open(path, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
