"""Classification head (reference: vit_core/mlp_head.py:6-15): LayerNorm -> Linear."""
import torch
from torch import nn

from . import _runtime as R
from ._functions import ln_linear_apply


class MLPHead(nn.Module):
    def __init__(self, d_model: int, num_classes: int):
        super().__init__()
        self.norm = nn.LayerNorm(d_model)
        self.linear = nn.Linear(d_model, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        R.require_gpu(x, "MLPHead")
        return ln_linear_apply(x, self.norm.weight, self.norm.bias, self.linear.weight, self.linear.bias, self.norm.eps)
