"""Where do the cycles of the tile-walking conv kernel (csrc/conv1d_tw.hip) go?  GPU only; needs the instrumentation build
(`make -C neuralsvb_amd/csrc instr`): per-stage shader-clock stamps of workgroups 0..3 and timing-only ablations."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from neuralsvb_amd import _lib  # noqa: E402

_ABL = os.environ.get("SVB_TW_ABL", "")          # "" = stamps only; N = the library compiled with -DSVB_TW_ABLATE=N (`make abl`)
_lib.LIB_PATH = os.path.join(ROOT, "neuralsvb_amd", f"libsvb_hip_instr_a{_ABL}.so" if _ABL else "libsvb_hip_instr.so")
from neuralsvb_amd import kernels as K  # noqa: E402

CONS = ["t0>t1 reads+MFMAs(+drain)", "t1>t2 lgkm+barrier", "phase"]
PROD = ["t0>t1 setup+issue", "t1>t2 wait+split+store", "t2>t3 vmcnt/lgkm(0)", "t3>t4 barrier", "phase"]


def timeit(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="32,192,384,1124,5")
    ap.add_argument("--cfgs", default="13,14")
    a = ap.parse_args()
    B, ca, cb, T, k = [int(v) for v in a.shape.split(",")]
    dev = torch.device("cuda:0")
    lib = _lib.get_lib()
    x = torch.randn(B, ca, T, device=dev)
    w = torch.randn(cb, ca, k, device=dev) * 0.05
    pk = K.weight_pack_q(w, None, 1)[0]
    flops = 2.0 * B * ca * cb * T * k
    for cfg in [int(c) for c in a.cfgs.split(",")]:
        fn = lambda: K.conv1d_forward(x, pk, cb, k, 1, (k - 1) // 2, 1, 1, force_cfg=cfg)
        print(f"== cfg {cfg}: {K._CFG_NAMES[cfg - 1]}  shape B{B} {ca}->{cb} k{k} T{T}")
        for drain in (0, 2, 4, 16):
            lib.svb_debug_set_tw(None, 0, drain)
            us = timeit(fn)
            print(f"   ablate={_ABL or 0} drain>={drain:2d}  {us:8.1f} us  {flops / us / 1e6:6.0f} TF", flush=True)
        buf = torch.zeros(4 * 8 * 32 * 8, dtype=torch.int64, device=dev)
        lib.svb_debug_set_tw(buf.data_ptr(), 0, 0)
        fn()
        torch.cuda.synchronize()
        lib.svb_debug_set_tw(None, 0, 0)
        st = buf.cpu().view(4, 8, 32, 8)
        for wg in (0, 1):
            for wave, names in ((0, CONS), (4, PROD)):
                s = st[wg, wave]
                nph = int((s[:, 0] > 0).sum())
                if nph < 3:
                    continue
                rows = []
                for ph in range(1, nph - 1):
                    d = [int(s[ph, i + 1] - s[ph, i]) if s[ph, i + 1] > 0 and s[ph, i] > 0 else 0 for i in range(len(names) - 1)]
                    d.append(int(s[ph + 1, 0] - s[ph, 0]))
                    rows.append(d)
                print(f"   wg {wg} wave {wave} ({'consumer' if wave < 4 else 'producer'}): {nph} phases stamped; cycles per phase:")
                for i, name in enumerate(names):
                    vals = [r[i] for r in rows]
                    print(f"      {name:28s} mean {sum(vals) / len(vals):8.0f}   " + " ".join(f"{v:6d}" for v in vals[:14]))


if __name__ == "__main__":
    main()
