"""AdamW + gradient clipping of one optimizer as two kernel launches over flat buffers (csrc/optim.hip).

The reference steps `torch.optim.AdamW` per optimizer (tasks/singing/svb_vae_task.py:84-118) after
`clip_grad_norm_` (:390-404).  Here the optimizer OBJECT stays a torch.optim.AdamW -- its `state_dict()` keeps the reference's
checkpoint layout ('step', 'exp_avg', 'exp_avg_sq' per parameter), schedulers keep writing `param_groups[0]['lr']` -- but its
parameters, gradients (FlatGradSync) and both moments are views of four flat fp32 buffers with one offset table, and `step()` is
`svb_adamw_flat`: no per-parameter Python, no multi-tensor launches (1.9 ms of host time per train step before)."""
import math

import torch

from .. import _lib as L


class FlatAdamW:
    def __init__(self, optimizer, sync):
        groups = optimizer.param_groups
        if len(groups) != 1 or groups[0].get("amsgrad") or groups[0].get("maximize"):
            raise NotImplementedError("FlatAdamW: one parameter group, no amsgrad / maximize")
        params = [p for p in groups[0]["params"]]
        if [id(p) for p in params] != [id(p) for p in sync.params]:
            raise ValueError("FlatAdamW: the gradient buffer was laid out for another parameter list")
        self.opt, self.sync, self.params = optimizer, sync, params
        dev, n = sync.flat.device, sync.flat.numel()
        self.p = torch.zeros(n, device=dev, dtype=torch.float32)
        self.m = torch.zeros(n, device=dev, dtype=torch.float32)
        self.v = torch.zeros(n, device=dev, dtype=torch.float32)
        lib = L.get_lib()
        self.ws = torch.zeros(max(lib.svb_adamw_flat_workspace_floats(), 1), device=dev, dtype=torch.float32)
        self.norm = torch.zeros(1, device=dev, dtype=torch.float32)
        self.max_norm = 0.0
        self.t = 0
        off = 0
        with torch.no_grad():
            for p in params:
                k = p.numel()
                pv = self.p[off:off + k].view_as(p)
                pv.copy_(p.data)
                p.data = pv                                   # the parameter now lives in the flat buffer
                st = optimizer.state[p]
                mv, vv = self.m[off:off + k].view_as(p), self.v[off:off + k].view_as(p)
                if "exp_avg" in st:                           # resumed from a checkpoint
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                    self.t = max(self.t, int(float(st["step"])))
                st["exp_avg"], st["exp_avg_sq"] = mv, vv
                st["step"] = torch.tensor(float(self.t), dtype=torch.float32)
                off += (k + 3) // 4 * 4

    def set_clip(self, max_norm):
        self.max_norm = float(max_norm) if max_norm else 0.0

    def step(self):
        g = self.opt.param_groups[0]
        self.t += 1
        b1, b2 = g["betas"]
        bc1 = 1.0 - b1 ** self.t
        bc2s = math.sqrt(1.0 - b2 ** self.t)
        lib = L.get_lib()
        st = torch._C._cuda_getCurrentRawStream(self.p.device.index) if self.p.is_cuda else None
        L.check(lib.svb_adamw_flat(self.p.data_ptr(), self.sync.flat.data_ptr(), self.m.data_ptr(), self.v.data_ptr(),
                                   self.p.numel(), float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                   float(g["weight_decay"]), bc1, bc2s, self.max_norm, self.ws.data_ptr(),
                                   self.norm.data_ptr(), st), "svb_adamw_flat")
        self.opt._opt_called = True                            # (what lr schedulers check before their first step)

    def export_state(self):
        """Before `optimizer.state_dict()`: the per-parameter 'step' entries of the reference layout."""
        for p in self.params:
            self.opt.state[p]["step"] = torch.tensor(float(self.t), dtype=torch.float32)

    def reattach(self):
        """After `optimizer.load_state_dict()` replaced the state tensors: copy them into the flat moments again."""
        off = 0
        with torch.no_grad():
            for p in self.params:
                k = p.numel()
                st = self.opt.state.get(p, {})
                mv, vv = self.m[off:off + k].view_as(p), self.v[off:off + k].view_as(p)
                if "exp_avg" in st and st["exp_avg"].data_ptr() != mv.data_ptr():
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                    self.t = max(self.t, int(float(st["step"])))
                    st["exp_avg"], st["exp_avg_sq"] = mv, vv
                off += (k + 3) // 4 * 4
