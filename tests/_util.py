"""Shared helpers for the tests (golden loading, comparisons)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def split_prefix(arrs, prefix):
    return {k[len(prefix):]: torch.from_numpy(np.ascontiguousarray(v)) for k, v in arrs.items() if k.startswith(prefix)}


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def rel_l2(a: torch.Tensor, b: torch.Tensor) -> float:
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def max_abs(a, b) -> float:
    return float((a.detach().double().cpu() - b.detach().double().cpu()).abs().max())
