"""MI355X drop-in for the reference's ``vit_core`` package (same public names as
/vit_core/__init__.py:1-5 of kristi700/ViT-SSL); the compute runs in libvitssl_hip.so."""
from .vit import ViT
from .encoder_block import EncoderBlock
from .feed_forward import FeedForwardBlock
from .attention import MultiHeadedAttention, ScaledDotProductAttention
from .patch_embedding import ConvolutionalPatchEmbedding, ManualPatchEmbedding, DynamicPatchEmbedding
