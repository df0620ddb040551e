"""The counter-based dropout stream of the HIP kernels (vit-ssl_amd/csrc/common.h), restated in NumPy.

Not a reference algorithm: the reference draws its masks with ``nn.Dropout`` (vit_core/encoder_block.py:29-30,45,51,
vit_core/feed_forward.py:16,27), i.e. torch's Philox / mt19937 Bernoulli stream, which a fused GEMM epilogue cannot
replay.  What has to hold instead is (i) the exported mask is what every kernel applies (the parity tests feed the
exported masks to the oracle), and (ii) the stream is statistically sound.  This file holds (ii) on the CPU and, on the
GPU, checks that ``vitssl_dropout_mask`` equals this restatement bit for bit, so the statistics below are statistics of
the kernels' stream.

Round 3 replaced the 8-round add/rotate/xor network by a 32-bit multiply / xor-shift mixer (2.8x cheaper on MI355X:
``v_mul_lo_u32`` issues as fast as ``v_alignbit_b32``).  Candidates that FAILED these checks while it was designed:
reduced-round ARX on a 2-D (row, column-group) counter (lag correlations of z = 20 at 8 rounds, gross bias below 6),
one-multiply mixers (row-lag correlations, z = 5-6), mixers that xor the key in before the first multiply (lag (16, 0))."""
import itertools
import math

import numpy as np
import pytest
import torch

U32 = np.uint32
C0 = 0x9E3779B1


def _mul(a, c):
    return ((a.astype(np.uint64) * np.uint64(c)) & np.uint64(0xFFFFFFFF)).astype(np.uint32)


def drop_key(seed: int, site: int):
    """make_drop_key: (k0, k1) from (seed, site) -- a splitmix-style 64-bit mix."""
    m = (1 << 64) - 1
    s = (seed * 0x9E3779B97F4A7C15 + (site + 1) * 0xD1B54A32D192ED03) & m
    s ^= s >> 29
    s = (s * 0xBF58476D1CE4E5B9) & m
    s ^= s >> 32
    return U32(s & 0xFFFFFFFF), U32(s >> 32)


def drop_words(g, k0, k1):
    """drop_words: group counter g (uint32 array) -> the two state words (a, b)."""
    a = (_mul(g.astype(np.uint32), C0) + k0).astype(np.uint32)
    a = a ^ (a >> U32(15))
    a = _mul(a, 0x2C1B3C6D)
    a = a ^ (a >> U32(12))
    a = _mul(a, 0x297A2D39)
    a = a ^ (a >> U32(15))
    b = _mul(a ^ k1, 0xC2B2AE35)
    b = b ^ (b >> U32(15))
    return a, b


def keep_mask(rows: int, cols: int, p: float, seed: int, site: int) -> np.ndarray:
    """The uint8 [rows, cols] keep mask every kernel applies for (p, seed, site)."""
    thr = min(int(p * 65536.0 + 0.5), 65535) if p > 0 else 0
    if thr == 0:
        return np.ones((rows, cols), np.uint8)
    k0, k1 = drop_key(seed, site)
    g = np.arange(rows * cols // 4, dtype=np.uint64).astype(np.uint32)
    a, b = drop_words(g, k0, k1)
    u = np.stack([a & U32(0xFFFF), a >> U32(16), b & U32(0xFFFF), b >> U32(16)], -1).astype(np.int64)
    s16 = np.where(u >= 32768, u - 65536, u)                   # the 16 bits read as a signed number
    return (s16 >= thr - 32768).astype(np.uint8).reshape(rows, cols)


# --------------------------------------------------------------------------------------------- statistics (CPU)
@pytest.mark.parametrize("cols,seed,p", [(3072, 12345, 0.1), (768, 999, 0.1), (768, 7, 0.5), (1536, 31337, 0.25)])
def test_stream_statistics(cols, seed, p):
    rows = 2048
    keep = keep_mask(rows, cols, p, seed, site=4).astype(np.float64)
    n = keep.size
    pe = 1.0 - round(p * 65536) / 65536
    var = pe * (1 - pe)
    assert abs(keep.mean() - pe) < 4.5 * math.sqrt(var / n)                       # overall keep rate
    zr = (keep.mean(1) - pe) / math.sqrt(var / cols)
    zc = (keep.mean(0) - pe) / math.sqrt(var / rows)
    assert abs(zr.std() - 1) < 0.08 and abs(zc.std() - 1) < 0.08                  # per-row / per-column rates: N(0, 1) z-scores
    assert np.abs(zr).max() < 5.2 and np.abs(zc).max() < 5.2
    k = keep - pe
    lags = [(0, 1), (0, 2), (0, 3), (0, 4), (0, 5), (0, 6), (0, 7), (0, 8), (0, 12), (0, 16), (0, 64), (0, 256),
            (1, 0), (2, 0), (3, 0), (4, 0), (5, 0), (7, 0), (8, 0), (16, 0), (128, 0), (1, 1), (1, 2), (1, 3), (1, 4), (2, 4), (16, 4)]
    for dr, dc in lags:                                                           # pairwise lag correlations
        a, b = k[:rows - dr, :cols - dc], k[dr:, dc:]
        z = (a * b).mean() / var * math.sqrt(a.size)
        assert abs(z) < 4.8, ((dr, dc), z)
    for dr, dc in [(1, 4), (1, 1), (3, 8), (16, 4), (7, 64), (1, 5), (2, 12), (5, 4)]:   # rectangle (4-point) products
        a = k[:rows - dr, :cols - dc] * k[:rows - dr, dc:] * k[dr:, :cols - dc] * k[dr:, dc:]
        assert abs(a.mean() / var ** 2 * math.sqrt(a.size)) < 4.8, (dr, dc)
    gk = k.reshape(rows, cols // 4, 4)                                            # the four elements of one draw
    for r in (2, 3, 4):
        for idx in itertools.combinations(range(4), r):
            prod = np.prod(gk[..., list(idx)], axis=-1)
            assert abs(prod.mean() / var ** (r / 2) * math.sqrt(prod.size)) < 4.8, idx
    kk = k[:1024, :min(cols, 1024)]                                               # 2-D spectrum: no lattice peaks
    F = np.abs(np.fft.fft2(kk)) ** 2 / (kk.size * var)
    F[0, 0] = 0
    assert F.max() < 24.0                                                         # exponential(1) bins: max of 1e6 ~ 14


def test_16bit_values_are_uniform():
    k0, k1 = drop_key(2024, 1)
    g = np.arange(1 << 22, dtype=np.uint64).astype(np.uint32)
    a, b = drop_words(g, k0, k1)
    for w in (a, b):
        for half in (w & U32(0xFFFF), w >> U32(16)):
            cnt = np.bincount(half.astype(np.int64), minlength=65536)
            e = half.size / 65536
            chi = ((cnt - e) ** 2 / e).sum()
            assert abs(chi - 65535) / math.sqrt(2 * 65535) < 4.5


def test_streams_of_different_keys_are_uncorrelated():
    rows, cols, p = 1024, 768, 0.1
    pe = 1.0 - round(p * 65536) / 65536
    base = keep_mask(rows, cols, p, 1234, 3).astype(np.float64) - pe
    for seed, site in ((1235, 3), (1234, 4), (1234, 2), (0, 3)):
        other = keep_mask(rows, cols, p, seed, site).astype(np.float64) - pe
        z = (base * other).mean() / (pe * (1 - pe)) * math.sqrt(base.size)
        assert abs(z) < 4.5, (seed, site, z)


# --------------------------------------------------------------------------------------------- the kernels use exactly this stream
@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,p,seed,site", [(512, 768, 0.1, 1234, 3), (197, 3072, 0.1, 2 ** 61 + 5, 35), (64, 64, 0.5, 0, 0),
                                                   (33, 1024, 0.9999, 77, 1), (1000, 384, 0.25, 42, 7)])
def test_exported_mask_is_the_restated_stream(rows, cols, p, seed, site):
    from vitssl_hip import ops
    dev = torch.device("cuda:0")
    got = ops.dropout_mask(rows, cols, ops.make_dropout(p, seed=seed, site=site), dev).cpu().numpy()
    assert np.array_equal(got, keep_mask(rows, cols, p, seed, site))
