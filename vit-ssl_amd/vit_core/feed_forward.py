"""FeedForwardBlock (reference: vit_core/feed_forward.py:7-28): Linear -> GELU(erf) ->
Dropout -> Linear.  Parameters are ordinary nn.Linear weights (same init / state_dict
keys); the forward is two MFMA GEMMs with the bias+GELU+dropout epilogue fused."""
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

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        R.require_gpu(x, "FeedForwardBlock")
        p = self.dropout.p if self.training else 0.0
        return ffn_apply(x, self.linear_in.weight, self.linear_in.bias, self.linear_out.weight, self.linear_out.bias, p)
