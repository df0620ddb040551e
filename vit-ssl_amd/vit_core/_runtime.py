"""Shared plumbing of the drop-in vit_core modules: lazy flat-store materialisation on
the module's device and loud failure when there is no GPU / no HIP library."""
import os
import sys

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from vitssl_hip import _lib as L  # noqa: E402
from vitssl_hip import ops  # noqa: E402
from vitssl_hip.engine import EncoderStack, FlatStore, GradReducer, Workspace  # noqa: E402,F401

BF16 = torch.bfloat16
F32 = torch.float32


def require_gpu(t: torch.Tensor, who: str):
    if not t.is_cuda:
        raise L.VitsslError(
            f"{who}: input is on {t.device}. This build runs the vit_core hot path only through the "
            "MI355X HIP library; there is no CPU fallback (move the module and inputs to 'cuda').")
    L.lib()


def next_seed() -> int:
    """Per-forward dropout seed drawn from torch's default CPU generator (so that
    torch.manual_seed makes training runs reproducible, as with nn.Dropout)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


def as_f32(x: torch.Tensor) -> torch.Tensor:
    if x.dtype != F32:
        x = x.float()
    return x.contiguous()


class StepPacer:
    """Bounds how far the host may run ahead of the GPU (in whole train steps).

    Measured on MI355X / ROCm 7.2 (tools/host_time.py): with ~330 launches per step and an
    unbounded launch queue the GPU itself runs 12-25 % slower (the runtime's in-flight
    packet/signal pool is exhausted and refilled in bursts); with a sync after every step
    the GPU idles while the host draws the next masks.  Keeping at most `depth` steps in
    flight gets both right."""

    def __init__(self, depth=None):
        if depth is None:
            depth = int(os.environ.get("VITSSL_STEP_LOOKAHEAD", "1"))
        self.depth = max(0, depth)
        self.events = []

    def begin_step(self):
        while len(self.events) > self.depth:
            self.events.pop(0).synchronize()

    def end_step(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.events.append(ev)
