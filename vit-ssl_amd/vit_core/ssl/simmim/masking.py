"""SimMIM random patch masking (reference: vit_core/ssl/simmim/masking.py:6-37).

The reference draws one ``torch.randperm(N)[:int(N*ratio)]`` per image.  The parity
contract is the CPU path: B sequential draws from torch's
default CPU generator, in batch order.  The draws therefore stay on the host (they are
index bookkeeping, 1 KiB per image) and only the resulting mask / index lists are
uploaded; the tensor work (mask-token substitution, target gather) runs in HIP."""
from typing import Tuple

import torch


def draw_mask(batch_size: int, num_patches: int, mask_ratio: float, generator=None) -> torch.Tensor:
    """Bool mask [B, N] on the CPU, bit-identical to the reference's CPU path."""
    num_masked = int(num_patches * mask_ratio)
    mask = torch.zeros(batch_size, num_patches, dtype=torch.bool)
    for b in range(batch_size):
        mask[b, torch.randperm(num_patches, generator=generator)[:num_masked]] = True
    return mask


def mask_indices(mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(idx int32 [n_masked] ascending flat (b, n) rows, inv int32 [B*N]: compact row or -1)."""
    flat = mask.reshape(-1)
    idx = flat.nonzero(as_tuple=False).squeeze(1).to(torch.int32)
    inv = torch.cumsum(flat.to(torch.int32), 0, dtype=torch.int32) - 1
    inv = torch.where(flat, inv, torch.full_like(inv, -1))
    return idx, inv


def simple_masking(patches: torch.Tensor, mask_ratio: float):
    """Same signature and returns as the reference: (patches, bool_mask [B,N], targets
    [B*nm, Pd] in ascending (b, n) order).  Stand-alone helper; SimMIMViT uses the fused
    HIP path (gather straight from the image) instead of materialising `patches`."""
    batch_size, num_patches, _ = patches.shape
    bool_mask = draw_mask(batch_size, num_patches, mask_ratio).to(patches.device)
    targets = patches[bool_mask]
    return patches, bool_mask, targets
