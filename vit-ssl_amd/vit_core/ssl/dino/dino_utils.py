"""Host-side DINO schedules: EMA momentum and teacher temperature.

Numbers as in the reference (vit_core/ssl/dino/dino_utils.py:4-36; evaluated once per EPOCH by
utils/trainers/dino_trainer.py:46,80): value(step) moves from `start` to `end` along half a
cosine (or a straight line) and stays at `end` from `total_iters` on.  Both schedulers are thin
named views of one interpolation helper, which is also what oracle.vit_oracle checks against
the reference's golden values (tests/golden/dino_sched.npz).
"""
import math


def interpolate(start: float, end: float, step: int, total: int, kind: str = "cosine") -> float:
    """Scheduled value after `step` of `total` iterations; `kind` is "cosine" or "linear"."""
    if step >= total:
        return end
    if kind == "linear":
        return start + (end - start) * (step / total)
    half_cosine = 0.5 * (1.0 + math.cos(math.pi * step / total))        # 1 at step 0, 0 at the end
    return end - (end - start) * half_cosine


class DINOMomentumScheduler:
    """teacher EMA momentum, cosine from `m_start` to `m_end`"""

    def __init__(self, m_start: float, m_end: float, total_iters: int):
        self.m_start, self.m_end, self.total_iters = m_start, m_end, total_iters

    def get_momentum(self, current_step: int) -> float:
        return interpolate(self.m_start, self.m_end, current_step, self.total_iters)


class DINOTeacherTempScheduler:
    """teacher softmax temperature warm-up (`schedule_type` "cosine" or "linear")"""

    def __init__(self, temp_start: float, temp_end: float, total_iters: int, schedule_type: str = "cosine"):
        self.t_start, self.t_end, self.total_iters = temp_start, temp_end, total_iters
        self.schedule_type = schedule_type

    def get_temp(self, current_step: int) -> float:
        return interpolate(self.t_start, self.t_end, current_step, self.total_iters, self.schedule_type)
