"""Deterministic, regenerable fill for the DINO head's large matrices.

DINOHead hard-wires hidden_dim=2048 (vit_core/ssl/dino/head.py:8), so a tiny
DINO fixture would still carry 2048x2048 matrices.  make_golden.py overwrites
those tensors in the reference model with this closed-form pattern and the
tests regenerate them instead of storing 60 MB of weights."""
import numpy as np

BIG_KEYS = ("mlp.0.weight", "mlp.2.weight", "mlp.4.weight")


def synth_weight(shape, c: int, scale: float) -> np.ndarray:
    i = np.arange(shape[0], dtype=np.int64)[:, None]
    j = np.arange(shape[1], dtype=np.int64)[None, :]
    h = (i * 131 + j * 71 + (i * j) % 97 * 13 + c * 29) % 257
    return ((h - 128).astype(np.float32) / 257.0 * scale).astype(np.float32)


def dino_big_weights(D: int, hidden: int = 2048):
    """{state_dict key: array} for teacher_head/student_head big matrices."""
    out = {}
    for hi, head in enumerate(("teacher_head.", "student_head.")):
        out[head + "mlp.0.weight"] = synth_weight((hidden, D), 1 + 10 * hi, 0.25)
        out[head + "mlp.2.weight"] = synth_weight((hidden, hidden), 2 + 10 * hi, 0.04)
        out[head + "mlp.4.weight"] = synth_weight((D, hidden), 3 + 10 * hi, 0.04)
    return out


def summarize(a: np.ndarray):
    """Slices + checksums standing in for a large tensor."""
    return dict(rows=a[:4].copy(), cols=a[:, :4].copy(),
                stats=np.array([a.sum(dtype=np.float64), np.square(a, dtype=np.float64).sum()]))
