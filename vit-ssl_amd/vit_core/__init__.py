"""`vit_core` on MI355X.

Exports the public names a user of kristi700/ViT-SSL imports from its `vit_core` package
(model, encoder block, attention, feed-forward and the three patch embedders); the SSL models
live in `vit_core.ssl.simmim` / `vit_core.ssl.dino`.  Every module keeps the reference's
constructor signature and `state_dict` keys and runs its forward and backward through
`libvitssl_hip.so` (hand-written gfx950 kernels behind the C ABI of `include/vitssl_hip.h`);
tensors on the CPU are rejected -- there is no fallback path.
"""
from .attention import MultiHeadedAttention, ScaledDotProductAttention
from .encoder_block import EncoderBlock
from .feed_forward import FeedForwardBlock
from .patch_embedding import (
    ConvolutionalPatchEmbedding,
    DynamicPatchEmbedding,
    ManualPatchEmbedding,
)
from .vit import ViT

__all__ = [
    "ViT",
    "EncoderBlock",
    "FeedForwardBlock",
    "MultiHeadedAttention",
    "ScaledDotProductAttention",
    "ConvolutionalPatchEmbedding",
    "ManualPatchEmbedding",
    "DynamicPatchEmbedding",
]
