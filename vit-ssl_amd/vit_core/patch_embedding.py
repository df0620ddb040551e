"""Patch embedders (reference: vit_core/patch_embedding.py): Conv2d(k=s=P), Unfold+Linear
and the variable-size variant with bicubic positional-embedding interpolation.  The
convolution IS a GEMM over gathered patches (feature order c,kh,kw); CLS prepend and the
positional add are fused into the GEMM epilogue."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _runtime as R
from ._functions import patch_embed_apply


class DynamicPatchEmbedding(nn.Module):
    """Handles variable input sizes by interpolating the positional embeddings."""

    def __init__(self, input_shape, embed_dim, patch_size):
        super().__init__()
        self.patch_size = patch_size
        self.grid_size = (input_shape[1] // patch_size, input_shape[2] // patch_size)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(input_shape[0], embed_dim, kernel_size=patch_size, stride=patch_size)
        self.cls_token = nn.Parameter(torch.rand(1, 1, embed_dim))
        self.positional_embedding = nn.Parameter(torch.rand(1, self.num_patches + 1, embed_dim))

    def interpolate_pos_encoding(self, x, w, h):
        """x: patch tokens [B, n, D] (only its shape is used); w, h: patch-grid size."""
        npatch = x.shape[1]
        if npatch == self.num_patches and w == h:
            return self.positional_embedding
        class_pos_embed = self.positional_embedding[:, 0]
        patch_pos_embed = self.positional_embedding[:, 1:]
        dim = x.shape[-1]
        patch_pos_embed = patch_pos_embed.reshape(1, self.grid_size[0], self.grid_size[1], dim).permute(0, 3, 1, 2)
        patch_pos_embed = F.interpolate(patch_pos_embed, size=(w, h), mode="bicubic")
        patch_pos_embed = patch_pos_embed.permute(0, 2, 3, 1).view(1, -1, dim)
        return torch.cat((class_pos_embed.unsqueeze(0), patch_pos_embed), dim=1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        batch_size, _, height, width = x.shape
        if height % self.patch_size != 0 or width % self.patch_size != 0:
            raise ValueError(
                f"Input image dimensions ({height}x{width}) must be divisible by patch size ({self.patch_size}).")
        R.require_gpu(x, "DynamicPatchEmbedding")
        w, h = height // self.patch_size, width // self.patch_size
        D = self.cls_token.shape[-1]
        shape_probe = torch.empty(1, w * h, D, device="meta")
        pos = self.interpolate_pos_encoding(shape_probe, w, h)
        return patch_embed_apply(x, self.proj.weight.reshape(D, -1), self.proj.bias, self.cls_token, pos[0], self.patch_size)


class ConvolutionalPatchEmbedding(nn.Module):
    """Conv2d based patch embedder"""

    def __init__(self, input_shape, embedding_dimension, patch_size):
        super().__init__()
        if input_shape[1] % patch_size != 0 or input_shape[2] % patch_size != 0:
            raise ValueError(
                f"Image dimensions H={input_shape[1]}, W={input_shape[2]} must be divisible by patch_size={patch_size}")
        self.patch_size = patch_size
        self.conv = nn.Conv2d(input_shape[0], embedding_dimension, kernel_size=patch_size, stride=patch_size)
        self.cls_token = nn.Parameter(torch.rand(1, 1, embedding_dimension))
        self.positional_embedding = nn.Parameter(torch.rand(1, (input_shape[1] // patch_size) ** 2 + 1, embedding_dimension))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        R.require_gpu(x, "ConvolutionalPatchEmbedding")
        D = self.cls_token.shape[-1]
        return patch_embed_apply(x, self.conv.weight.reshape(D, -1), self.conv.bias, self.cls_token,
                                 self.positional_embedding[0], self.patch_size)


class ManualPatchEmbedding(nn.Module):
    """Unfold + Linear patch embedder"""

    def __init__(self, input_shape, embedding_dimension, patch_size):
        super().__init__()
        if input_shape[1] % patch_size != 0 or input_shape[2] % patch_size != 0:
            raise ValueError(
                f"Image dimensions H={input_shape[1]}, W={input_shape[2]} must be divisible by patch_size={patch_size}")
        self.patch_size = patch_size
        self.unfold = nn.Unfold(kernel_size=(patch_size, patch_size), stride=patch_size)
        self.linear = nn.Linear(input_shape[0] * patch_size * patch_size, embedding_dimension)
        self.cls_token = nn.Parameter(torch.rand(1, 1, embedding_dimension))
        self.positional_embedding = nn.Parameter(torch.rand(1, (input_shape[1] // patch_size) ** 2 + 1, embedding_dimension))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        R.require_gpu(x, "ManualPatchEmbedding")
        return patch_embed_apply(x, self.linear.weight, self.linear.bias, self.cls_token, self.positional_embedding[0],
                                 self.patch_size)
