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


def check_saved_generation(who: str, graph_gen: int, live_gen: int):
    """The engine keeps the activations of ONE grad-enabled forward per runtime (step-persistent
    workspaces, not per-graph autograd storage).  A backward through an older graph would
    silently use the newer forward's activations: refuse it."""
    if graph_gen != live_gen:
        raise L.VitsslError(
            f"{who}: backward of forward #{graph_gen}, but a later grad-enabled forward (#{live_gen}) has replaced its "
            "saved activations. Call backward() before the next training forward (or run the extra forward under "
            "torch.no_grad()); gradient accumulation over micro-batches works as forward/backward pairs.")


def as_f32(x: torch.Tensor) -> torch.Tensor:
    if x.dtype != F32:
        x = x.float()
    return x.contiguous()


def limit_host_threads(n=None):
    """Cap torch's intra-op CPU pool while the GPU engine drives the step.

    A GPU job normally owns a small CPU share (cgroup quota).  torch sizes its OpenMP pool to
    the machine's core count and the workers spin-wait after every parallel CPU op, which
    burns the quota and gets the whole process throttled for the rest of the scheduler period
    (measured on the MI355X box: 70-85 ms stalls at arbitrary host calls, ViT-S 48 -> 24
    ms/step once capped).  `VITSSL_HOST_THREADS` overrides; returns the previous setting."""
    prev = torch.get_num_threads()
    if n is None:
        n = int(os.environ.get("VITSSL_HOST_THREADS", "4"))
    if n > 0:
        torch.set_num_threads(min(n, prev))
    return prev


class StepPacer:
    """Bounds how far the host may run ahead of the GPU (in whole train steps).

    Measured on MI355X / ROCm 7.2 (bench.py, ViT-B/16 and ViT-S/16 SimMIM, batch 256) once
    the host pool is capped (limit_host_threads): look-ahead 0 (sync every step) 50.0 / 26.1
    ms per step, 1: 46.8 / 23.6, 2: 45.1 / 22.9, 3: 44.8 / 21.3, 8: 44.8 / 21-22 (noisier).
    Three steps in flight hide the host's mask bookkeeping and its launch jitter completely
    (44.8 ms is also what a HIP-graph replay of the ViT-B step takes); deeper queues buy
    nothing and only hold more pinned mask slots and event objects alive."""

    def __init__(self, depth=None):
        if depth is None:
            depth = int(os.environ.get("VITSSL_STEP_LOOKAHEAD", "3"))
        self.depth = max(0, depth)
        self.events = []

    def begin_step(self):
        while len(self.events) > self.depth:
            self.events.pop(0).synchronize()

    def end_step(self):
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self.events.append(ev)
