"""Transformer MLP block on the HIP engine.

Module surface of the reference (vit_core/feed_forward.py:7-28): `FeedForwardBlock(d_model,
d_ff, dropout)` with `linear_in`, `linear_out` (ordinary nn.Linear parameters: same init,
same checkpoint keys) and `dropout`.  Semantics: linear_out(dropout(gelu_erf(linear_in(x)))).
Execution: two bf16 MFMA GEMMs; bias, exact-erf GELU and the counter-based dropout mask are
applied in the first GEMM's epilogue (which also stores keep*gelu'(u) for the backward), the
second GEMM adds its bias in fp32 (`ffn_apply`, a torch.autograd.Function).
"""
import torch
from torch import nn

from . import _runtime as R
from ._functions import ffn_apply


class FeedForwardBlock(nn.Module):
    def __init__(self, d_model: int = 512, d_ff: int = 2048, dropout: float = 0.1):
        super().__init__()
        self.linear_in = nn.Linear(d_model, d_ff)
        self.linear_out = nn.Linear(d_ff, d_model)
        self.dropout = nn.Dropout(dropout)

    def _drop_p(self) -> float:
        """dropout probability in effect: the module's p while training, 0 in eval"""
        return float(self.dropout.p) if self.training else 0.0

    def extra_repr(self) -> str:
        return f"fused GEMM+GELU+dropout epilogue, hidden {self.linear_in.out_features}"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [..., d_model] fp32 on the GPU -> same shape."""
        R.require_gpu(x, "FeedForwardBlock")
        fc1, fc2 = self.linear_in, self.linear_out
        return ffn_apply(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias, self._drop_p())
