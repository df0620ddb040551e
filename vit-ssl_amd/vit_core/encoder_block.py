"""Pre-LN transformer encoder block (reference: vit_core/encoder_block.py:9-53).

Inside ViT / SimMIMViT / DINOViT the blocks are pure parameter containers: the owning
model runs all blocks through one EncoderStack schedule.  Called on its own, a block
builds a private one-block stack on first use."""
import torch
from torch import nn

from . import _runtime as R
from .attention import MultiHeadedAttention
from .feed_forward import FeedForwardBlock
from ._functions import StackRunner


class EncoderBlock(nn.Module):
    def __init__(self, d_model: int = 512, num_heads: int = 8, mlp_dim: int = 3072, dropout: float = 0.1):
        super().__init__()
        self.self_attention = MultiHeadedAttention(d_model, num_heads)
        self.feed_forward = FeedForwardBlock(d_model, mlp_dim, dropout)
        self.layer_norm1 = nn.LayerNorm(d_model)
        self.layer_norm2 = nn.LayerNorm(d_model)
        self.drop1 = nn.Dropout(dropout)
        self.drop2 = nn.Dropout(dropout)
        self._cfg = (d_model, num_heads, mlp_dim, dropout)
        self._runner = None

    def forward(self, x: torch.Tensor, return_attn=False):
        R.require_gpu(x, "EncoderBlock")
        if self._runner is None or not self._runner.valid_for(x.device):
            d, h, f, p = self._cfg
            object.__setattr__(self, "_runner", StackRunner(self, [""], d, h, f, p, x.device))
        return self._runner(x, self.training, return_attn)
