"""LR schedules of the path (reference utils/common_schedulers.py:24-50)."""


class RSQRTSchedule:
    """lr = max(lr0 * min(step/warmup, 1) * max(warmup, step)^-0.5 * hidden^-0.5, 1e-7)"""

    def __init__(self, optimizer, hparams):
        self.optimizer = optimizer
        self.base_lr, self.warmup, self.hidden = hparams["lr"], hparams["warmup_updates"], hparams["hidden_size"]
        self.lr = self.base_lr
        self.step(0)

    def step(self, num_updates):
        scale = min(num_updates / self.warmup, 1.0) * max(self.warmup, num_updates) ** -0.5 * self.hidden ** -0.5
        self.lr = max(self.base_lr * scale, 1e-7)
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr
        return self.lr

    def get_lr(self):
        return self.optimizer.param_groups[0]["lr"]

    get_last_lr = get_lr


class NoneSchedule:
    def __init__(self, optimizer, hparams):
        self.optimizer = optimizer
        self.lr = hparams["lr"]
        self.step(0)

    def step(self, num_updates):
        for g in self.optimizer.param_groups:
            g["lr"] = self.lr
        return self.lr

    def get_lr(self):
        return self.optimizer.param_groups[0]["lr"]

    get_last_lr = get_lr
