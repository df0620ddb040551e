"""SimMIM random patch masking (reference: vit_core/ssl/simmim/masking.py:6-37).

The reference draws one ``torch.randperm(N)[:int(N*ratio)]`` per image.  The parity
contract is the CPU path: B sequential draws from torch's
default CPU generator, in batch order.  The draws therefore stay on the host (they are
index bookkeeping, 1 KiB per image) and only the resulting mask / index lists are
uploaded; the tensor work (mask-token substitution, target gather) runs in HIP."""
from typing import Tuple

import numpy as np
import torch


def draw_mask(batch_size: int, num_patches: int, mask_ratio: float, generator=None) -> torch.Tensor:
    """Bool mask [B, N] on the CPU, bit-identical to the reference's CPU path.

    Only the B tiny ``torch.randperm(N)`` draws touch torch (they define the parity
    contract); the scatter into the mask is NumPy.  Any torch CPU op on the whole
    50k-element mask would wake the full OpenMP pool, whose spin-waiting threads exhaust
    the job's CPU quota on a GPU box (measured: 70-85 ms process stalls per step)."""
    num_masked = int(num_patches * mask_ratio)
    mask = np.zeros((batch_size, num_patches), dtype=np.bool_)
    for b in range(batch_size):
        mask[b, torch.randperm(num_patches, generator=generator)[:num_masked].numpy()] = True
    return torch.from_numpy(mask)


def mask_indices_np(mask: torch.Tensor):
    """NumPy version (single-threaded on purpose: torch CPU ops on 50k-element tensors fan
    out to every core via OpenMP, and the spinning workers starve the HIP runtime's
    signal-handling thread -- measured 7 ms of GPU idle per step on a 256-core host)."""
    flat = mask.numpy().reshape(-1)
    idx = np.flatnonzero(flat).astype(np.int32)
    inv = np.cumsum(flat, dtype=np.int32) - 1
    inv[~flat] = -1
    return idx, inv, flat.view(np.uint8)


def mask_indices(mask: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """(idx int32 [n_masked] ascending flat (b, n) rows, inv int32 [B*N]: compact row or -1)."""
    idx, inv, _ = mask_indices_np(mask.contiguous())
    return torch.from_numpy(idx), torch.from_numpy(inv)


def simple_masking(patches: torch.Tensor, mask_ratio: float):
    """Same signature and returns as the reference: (patches, bool_mask [B,N], targets
    [B*nm, Pd] in ascending (b, n) order).  Stand-alone helper; SimMIMViT uses the fused
    HIP path (gather straight from the image) instead of materialising `patches`."""
    batch_size, num_patches, _ = patches.shape
    bool_mask = draw_mask(batch_size, num_patches, mask_ratio).to(patches.device)
    targets = patches[bool_mask]
    return patches, bool_mask, targets
