"""One conv shape, a few launches: target for `rocprofv3 --pmc ...` runs (GPU only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from neuralsvb_amd import kernels as K  # noqa: E402

B, Cin, Cout, T, k = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (32, 192, 384, 1124, 5))]
what = sys.argv[6] if len(sys.argv) > 6 else "fwd"
cfg = int(sys.argv[7]) if len(sys.argv) > 7 else 0          # tile configuration to force (0 = the launcher's choice)
K.AUTOTUNE = False
dev = torch.device("cuda:0")
x = torch.randn(B, Cin, T, device=dev)
w = torch.randn(Cout, Cin, k, device=dev) * 0.05
dy = torch.randn(B, Cout, T, device=dev)
qa, qb = K.weight_pack_q(w, None, 1)
pad = (k - 1) // 2
for _ in range(6):
    if what == "fwd":
        K.conv1d_forward(x, qa, Cout, k, 1, pad, 1, 1, force_cfg=cfg)
    elif what == "dgrad":
        K.conv1d_transposed(dy, qb, Cin, T, k, 1, pad, 1, 1, force_cfg=cfg)
    else:
        K.conv1d_wgrad(dy, x, k, 1, pad, 1, 1, bf16x3=True)
torch.cuda.synchronize()
