"""Linear learning-rate warm-up (reference: utils/schedulers.py:1-19), stepped per batch."""


class LinearWarmupScheduler:
    def __init__(self, optimizer, warmup_steps, start_lr, target_lr):
        self.optimizer = optimizer
        self._step = 0
        self.warmup_steps = max(1, warmup_steps)
        self.start_lr = start_lr
        self.target_lr = target_lr

    def step(self):
        self._step += 1
        if self._step > self.warmup_steps:
            return
        frac = float(self._step) / self.warmup_steps
        lr = self.start_lr + frac * (self.target_lr - self.start_lr)
        for group in self.optimizer.param_groups:
            group["lr"] = lr
