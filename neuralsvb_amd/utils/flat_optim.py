"""AdamW + gradient clipping of one optimizer as two kernel launches over flat buffers (csrc/optim.hip).

The reference steps `torch.optim.AdamW` per optimizer (tasks/singing/svb_vae_task.py:84-118) after
`clip_grad_norm_` (:390-404).  Here the optimizer OBJECT stays a torch.optim.AdamW -- its `state_dict()` keeps the reference's
checkpoint layout ('step', 'exp_avg', 'exp_avg_sq' per parameter), schedulers keep writing `param_groups[0]['lr']` -- but its
parameters, gradients (FlatGradSync) and both moments are views of four flat fp32 buffers with one offset table, and `step()` is
`svb_adamw_flat`: no per-parameter Python, no multi-tensor launches (1.9 ms of host time per train step before).

Parameters WITHOUT a gradient in a pass (`p.grad is None`: only possible for the parameters FlatGradSync hands to autograd,
`drop_autograd_grads`) are treated exactly as torch.optim.AdamW treats them -- no weight decay, no moment decay, their own
`step` count does not advance.  The flat kernel is elementwise with ONE step count, so it is used while that is exact: every
parameter that ever had a gradient has one in this pass too, and the ones that never had one are untouched by it (weight decay
0, zero moments: the update of a zero gradient is then the identity).  Any other step -- a parameter with history skipped, weight
decay acting on a never-updated parameter, step counts that have diverged -- goes through `torch.optim.AdamW.step()` itself
on the same (flat-backed) tensors, after torch's own clip_grad_norm_.  A NaN / inf gradient norm: `fminf(1, nan)` gives a
clip factor of 1 where clip_grad_norm_ would spread the NaN over every gradient; both end in NaN parameters for the
offending elements, the flat path leaves the others finite."""
import math

import torch

from .. import _lib as L


class FlatAdamW:
    @staticmethod
    def eligible(optimizer):
        """What Trainer.setup checks before building one (anything else keeps `optimizer.step()`)."""
        groups = optimizer.param_groups
        return (isinstance(optimizer, torch.optim.AdamW) and len(groups) == 1
                and not any(groups[0].get(k) for k in ("amsgrad", "maximize", "capturable", "differentiable")))

    def __init__(self, optimizer, sync):
        groups = optimizer.param_groups
        if not self.eligible(optimizer):
            raise NotImplementedError("FlatAdamW: one parameter group, no amsgrad / maximize / capturable / differentiable")
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
        self.steps = [0] * len(params)        # per-parameter step counts (torch keeps one per parameter)
        off = 0
        with torch.no_grad():
            for i, p in enumerate(params):
                k = p.numel()
                pv = self.p[off:off + k].view_as(p)
                pv.copy_(p.data)
                p.data = pv                                   # the parameter now lives in the flat buffer
                st = optimizer.state[p]
                mv, vv = self.m[off:off + k].view_as(p), self.v[off:off + k].view_as(p)
                if "exp_avg" in st:                           # resumed from a checkpoint
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                    self.steps[i] = int(float(st["step"]))
                st["exp_avg"], st["exp_avg_sq"] = mv, vv
                st["step"] = torch.tensor(float(self.steps[i]), dtype=torch.float32)
                off += (k + 3) // 4 * 4
        self.t = max(self.steps) if self.steps else 0

    def set_clip(self, max_norm):
        self.max_norm = float(max_norm) if max_norm else 0.0

    def _flat_is_exact(self, skipped):
        """One step count for every parameter that is updated in this pass, and the skipped ones untouched by a zero-gradient
        update: never updated before (zero moments) and no weight decay (see the module docstring)."""
        skip = set(skipped)
        wd = float(self.opt.param_groups[0]["weight_decay"])
        for i, n in enumerate(self.steps):
            if (n != 0 or wd != 0.0) if i in skip else (n != self.t):
                return False
        return True

    def step(self, skipped=()):
        """`skipped`: indices (into the parameter list) of parameters without a gradient in this pass (FlatGradSync.gather_adopted
        returns them)."""
        g = self.opt.param_groups[0]
        if not self._flat_is_exact(skipped):
            return self._torch_step(skipped)
        skip = set(skipped)
        self.t += 1
        for i in range(len(self.steps)):
            if i not in skip:
                self.steps[i] = self.t
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

    def _torch_step(self, skipped):
        """The exact (slow) form: torch's own clipping and AdamW on the flat-backed tensors, for a pass in which the flat kernel's
        single step count would not reproduce torch (a parameter without a gradient that has history or weight decay)."""
        skip = set(skipped)
        self.export_state()
        if self.max_norm > 0.0:
            n = torch.nn.utils.clip_grad_norm_([p for i, p in enumerate(self.params) if i not in skip], self.max_norm)
            self.norm.copy_(n.reshape(1))
        saved = {}
        for i in skip:                                        # (a flat-backed zero view would make torch update it)
            saved[i], self.params[i].grad = self.params[i].grad, None
        # the tasks build their optimizers with fused=True on the GPU, and this object keeps every state['step'] as a CPU tensor
        # (the reference's state_dict layout): torch's fused kernel refuses that mix, its multi-tensor form takes it
        flags = [(g, g.get("fused"), g.get("foreach")) for g in self.opt.param_groups]
        for g, fused, _ in flags:
            if fused:
                g["fused"], g["foreach"] = False, True
        try:
            self.opt.step()
        finally:
            for g, fused, foreach in flags:
                g["fused"], g["foreach"] = fused, foreach
            for i, gr in saved.items():
                self.params[i].grad = gr
        for i, p in enumerate(self.params):
            if i not in skip:
                self.steps[i] += 1
        self.t = max(self.steps)

    def export_state(self):
        """Before `optimizer.state_dict()`: the per-parameter 'step' entries of the reference layout."""
        for p, n in zip(self.params, self.steps):
            self.opt.state[p]["step"] = torch.tensor(float(n), dtype=torch.float32)

    def reattach(self):
        """After `optimizer.load_state_dict()` replaced the state tensors: copy them into the flat moments again."""
        off = 0
        with torch.no_grad():
            for i, p in enumerate(self.params):
                k = p.numel()
                st = self.opt.state.get(p, {})
                mv, vv = self.m[off:off + k].view_as(p), self.v[off:off + k].view_as(p)
                if "exp_avg" in st and st["exp_avg"].data_ptr() != mv.data_ptr():
                    mv.copy_(st["exp_avg"])
                    vv.copy_(st["exp_avg_sq"])
                    self.steps[i] = int(float(st["step"]))
                    st["exp_avg"], st["exp_avg_sq"] = mv, vv
                off += (k + 3) // 4 * 4
        self.t = max(self.steps) if self.steps else 0
