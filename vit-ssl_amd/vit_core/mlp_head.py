"""Classification head of the supervised ViT.

Same module surface as the reference (vit_core/mlp_head.py:6-15: `MLPHead(d_model,
num_classes)` holding `norm` = LayerNorm and `linear` = Linear, hence the same checkpoint
keys `classification_head.norm.*` / `classification_head.linear.*`).  The forward is not two
torch ops but one engine call: LayerNorm emits the bf16 GEMM operand directly and the Linear
runs as a bf16 MFMA GEMM with an fp32 epilogue (`ln_linear_apply`, backward included).
"""
import torch
from torch import nn

from . import _runtime as R
from ._functions import ln_linear_apply


class MLPHead(nn.Module):
    def __init__(self, d_model: int, num_classes: int):
        super().__init__()
        self.norm = nn.LayerNorm(d_model)
        self.linear = nn.Linear(d_model, num_classes)

    def extra_repr(self) -> str:
        return f"fused LayerNorm({self.norm.normalized_shape[0]}) -> Linear({self.linear.out_features}) on libvitssl_hip"

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x: [..., d_model] fp32 on the GPU (the CLS rows) -> logits [..., num_classes] fp32."""
        R.require_gpu(x, "MLPHead")
        ln, fc = self.norm, self.linear
        return ln_linear_apply(x, ln.weight, ln.bias, fc.weight, fc.bias, ln.eps)
