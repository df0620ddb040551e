"""Host-side DINO schedules (reference: vit_core/ssl/dino/dino_utils.py:4-36): cosine
(or linear) interpolation clamped at total_iters.  Pure Python scalars."""
import math


def _cosine(start: float, end: float, step: int, total: int) -> float:
    if step >= total:
        return end
    return end - (end - start) * 0.5 * (1.0 + math.cos(math.pi * step / total))


class DINOMomentumScheduler:
    def __init__(self, m_start: float, m_end: float, total_iters: int):
        self.m_start, self.m_end, self.total_iters = m_start, m_end, total_iters

    def get_momentum(self, current_step: int) -> float:
        return _cosine(self.m_start, self.m_end, current_step, self.total_iters)


class DINOTeacherTempScheduler:
    def __init__(self, temp_start: float, temp_end: float, total_iters: int, schedule_type: str = "cosine"):
        self.t_start, self.t_end, self.total_iters = temp_start, temp_end, total_iters
        self.schedule_type = schedule_type

    def get_temp(self, current_step: int) -> float:
        if current_step >= self.total_iters:
            return self.t_end
        if self.schedule_type == "linear":
            return self.t_start + (self.t_end - self.t_start) * (current_step / self.total_iters)
        return _cosine(self.t_start, self.t_end, current_step, self.total_iters)
