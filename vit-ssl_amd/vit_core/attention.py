"""Attention (reference: vit_core/attention.py).  ScaledDotProductAttention is one fused
HIP kernel (scores never reach HBM); MultiHeadedAttention adds the four bias-free
projections as MFMA GEMMs.  Return conventions match the reference: always a
``(tensor, attn_probs_or_None)`` tuple."""
import torch
from torch import nn

from . import _runtime as R
from ._functions import sdpa_apply, mha_apply


def ScaledDotProductAttention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, return_attn: bool = False):
    """query/key/value: [..., seq, d_k] with identical shapes (self-attention geometry)."""
    R.require_gpu(query, "ScaledDotProductAttention")
    return sdpa_apply(query, key, value, return_attn)


class MultiHeadedAttention(nn.Module):
    def __init__(self, d_model: int, num_heads: int):
        super().__init__()
        assert (
            d_model % num_heads == 0
        ), f"d_model({d_model}) must be cleanly divisible by num_heads({num_heads})!"
        self.d_model = d_model
        self.d_k = d_model // num_heads
        self.d_v = d_model // num_heads
        self.num_heads = num_heads
        self.w_query = nn.Linear(d_model, d_model, bias=False)
        self.w_key = nn.Linear(d_model, d_model, bias=False)
        self.w_value = nn.Linear(d_model, d_model, bias=False)
        self.final_linear = nn.Linear(d_model, d_model, bias=False)

    def forward(self, query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, return_attn: bool = False):
        R.require_gpu(query, "MultiHeadedAttention")
        return mha_apply(query, key, value, self.w_query.weight, self.w_key.weight, self.w_value.weight,
                         self.final_linear.weight, self.num_heads, return_attn)
