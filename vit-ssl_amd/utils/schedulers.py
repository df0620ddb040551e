"""Learning-rate warm-up, advanced once per optimizer step.

Behaviour of the reference's `LinearWarmupScheduler` (utils/schedulers.py:1-19): after the
k-th call of `step()` (k = 1 .. warmup_steps) every parameter group runs at
start_lr + k / warmup_steps * (target_lr - start_lr); later calls leave the groups alone so the
per-epoch main scheduler owns them.  Written here around a pure function so the CPU oracle
(`oracle.vit_oracle.linear_warmup_lr`) and the host-logic tests can check the same numbers, and
with `state_dict` support so a resumed run continues the ramp.
"""


def warmup_value(done_steps: int, total_steps: int, start_lr: float, target_lr: float) -> float:
    """Learning rate after `done_steps` warm-up steps (clamped to the end of the ramp)."""
    total_steps = max(1, int(total_steps))
    progress = min(max(int(done_steps), 0), total_steps) / total_steps
    return start_lr + progress * (target_lr - start_lr)


class LinearWarmupScheduler:
    def __init__(self, optimizer, warmup_steps, start_lr, target_lr):
        self.optimizer = optimizer
        self.warmup_steps = max(1, warmup_steps)
        self.start_lr, self.target_lr = start_lr, target_lr
        self._step = 0

    @property
    def finished(self) -> bool:
        return self._step >= self.warmup_steps

    def step(self):
        self._step += 1
        if self._step <= self.warmup_steps:
            lr = warmup_value(self._step, self.warmup_steps, self.start_lr, self.target_lr)
            for group in self.optimizer.param_groups:
                group["lr"] = lr

    def state_dict(self):
        return {"step": self._step, "warmup_steps": self.warmup_steps, "start_lr": self.start_lr, "target_lr": self.target_lr}

    def load_state_dict(self, state):
        self._step = int(state["step"])
        self.warmup_steps = int(state["warmup_steps"])
        self.start_lr, self.target_lr = float(state["start_lr"]), float(state["target_lr"])
