"""Patch embedders on the HIP engine.

Public surface = the reference's (vit_core/patch_embedding.py:10-128): the three module
classes, their constructor signatures, parameter names (`proj` / `conv` / `linear`,
`cls_token`, `positional_embedding` -> identical `state_dict` keys) and parameter creation
order (so a seeded construction draws the same initial values).  Everything behind that is
different: a patch projection is ONE GEMM over gathered patches (feature order c, kh, kw,
which is both Conv2d's flattened weight layout and Unfold's output order), and the CLS
prepend plus the positional add happen in that GEMM's epilogue (`patch_embed_apply`).
"""
import torch
import torch.nn as nn

from . import _runtime as R
from ._functions import bicubic_rows_apply, patch_embed_apply


def _grid_of(input_shape, patch_size, strict):
    """(rows, cols) of the patch grid; `strict` raises for images the patch size does not tile."""
    _, height, width = input_shape
    if strict and (height % patch_size or width % patch_size):
        raise ValueError(f"patch_size={patch_size} does not tile a {height}x{width} image")
    return height // patch_size, width // patch_size


class _TokenTable(nn.Module):
    """CLS token + learned positional table shared by the three embedders (uniform [0,1) init,
    CLS drawn first, as in the reference)."""

    def _make_tokens(self, num_patches, width):
        self.cls_token = nn.Parameter(torch.rand(1, 1, width))
        self.positional_embedding = nn.Parameter(torch.rand(1, num_patches + 1, width))

    def _embed(self, images, weight2d, bias, pos_rows):
        R.require_gpu(images, type(self).__name__)
        return patch_embed_apply(images, weight2d, bias, self.cls_token, pos_rows, self.patch_size)


class DynamicPatchEmbedding(_TokenTable):
    """Accepts any image size the patch size tiles; the positional table is resampled
    (bicubic, on the patch grid) when the token count differs from the construction size."""

    def __init__(self, input_shape, embed_dim, patch_size):
        super().__init__()
        self.patch_size = patch_size
        self.grid_size = _grid_of(input_shape, patch_size, strict=False)
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.proj = nn.Conv2d(input_shape[0], embed_dim, kernel_size=patch_size, stride=patch_size)
        self._make_tokens(self.num_patches, embed_dim)

    def _positions_for(self, rows, cols):
        """[1, 1 + rows*cols, D] positional rows for a rows x cols patch grid."""
        table = self.positional_embedding
        if rows == cols and rows * cols == self.num_patches:
            return table
        R.require_gpu(table, "DynamicPatchEmbedding")
        patch_rows = bicubic_rows_apply(table[0, 1:], self.grid_size, (rows, cols))  # ATen-compatible bicubic, HIP
        return torch.cat((table[0, :1], patch_rows), dim=0).unsqueeze(0)

    def interpolate_pos_encoding(self, x, w, h):
        """Reference-compatible spelling (patch tokens, grid rows, grid cols); only the token
        count of `x` matters."""
        if x.shape[1] == self.num_patches and w == h:
            return self.positional_embedding
        return self._positions_for(w, h)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        height, width = x.shape[-2:]
        if height % self.patch_size or width % self.patch_size:
            raise ValueError(f"patch_size={self.patch_size} does not tile a {height}x{width} input")
        rows, cols = height // self.patch_size, width // self.patch_size
        pos = self._positions_for(rows, cols)
        return self._embed(x, self.proj.weight.flatten(1), self.proj.bias, pos[0])


class ConvolutionalPatchEmbedding(_TokenTable):
    """Conv2d(kernel = stride = patch) parameters, executed as a patch GEMM."""

    def __init__(self, input_shape, embedding_dimension, patch_size):
        super().__init__()
        rows, cols = _grid_of(input_shape, patch_size, strict=True)
        self.patch_size = patch_size
        self.conv = nn.Conv2d(input_shape[0], embedding_dimension, kernel_size=patch_size, stride=patch_size)
        self._make_tokens(rows * rows, embedding_dimension)       # the reference sizes the table from H alone

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._embed(x, self.conv.weight.flatten(1), self.conv.bias, self.positional_embedding[0])


class ManualPatchEmbedding(_TokenTable):
    """Unfold + Linear parameters (exported by the reference, used by none of its models)."""

    def __init__(self, input_shape, embedding_dimension, patch_size):
        super().__init__()
        rows, cols = _grid_of(input_shape, patch_size, strict=True)
        self.patch_size = patch_size
        self.unfold = nn.Unfold(kernel_size=(patch_size, patch_size), stride=patch_size)
        self.linear = nn.Linear(input_shape[0] * patch_size * patch_size, embedding_dimension)
        self._make_tokens(rows * rows, embedding_dimension)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._embed(x, self.linear.weight, self.linear.bias, self.positional_embedding[0])
