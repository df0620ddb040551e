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
    """make_drop_key: (k0, k1, k2) from (seed, site) -- a splitmix-style 64-bit mix, and a second round of it for k2."""
    m = (1 << 64) - 1
    s = (seed * 0x9E3779B97F4A7C15 + (site + 1) * 0xD1B54A32D192ED03) & m
    s ^= s >> 29
    s = (s * 0xBF58476D1CE4E5B9) & m
    s ^= s >> 32
    t = (s * 0x94D049BB133111EB) & m
    t ^= t >> 31
    return U32(s & 0xFFFFFFFF), U32(s >> 32), U32((t >> 16) & 0xFFFFFFFF)


def drop_words(g, k0, k1, k2):
    """drop_words: group counter g (uint32 array) -> the two state words (a, b).  Round 4: the key also enters word a -- its first
    multiplier is the odd number M1 ^ (k1 << 1) and k2 is xored in between the two multiplies.  With the key in word b only, the
    a-words of any two keys were shifted windows of ONE 2^32-long sequence (test_streams_of_different_keys_are_not_shifted_copies).
    Same instruction count on the GPU: the multiplier is a scalar operand either way and the xor fuses into v_xor3_b32."""
    a = (_mul(g.astype(np.uint32), C0) + k0).astype(np.uint32)
    a = a ^ (a >> U32(15))
    a = _mul(a, ((0x2C1B3C6D ^ (int(k1) << 1)) & 0xFFFFFFFF) | 1)
    a = a ^ (a >> U32(12)) ^ k2
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
    k0, k1, k2 = drop_key(seed, site)
    g = np.arange(rows * cols // 4, dtype=np.uint64).astype(np.uint32)
    a, b = drop_words(g, k0, k1, k2)
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
    k0, k1, k2 = drop_key(2024, 1)
    g = np.arange(1 << 22, dtype=np.uint64).astype(np.uint32)
    a, b = drop_words(g, k0, k1, k2)
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


def test_streams_of_different_keys_are_not_shifted_copies():
    """Round-3 advisor finding: word a (elements 0 and 1 of every group) was a fixed bijection of g * C0 + k0, so the a-streams of
    ANY two (seed, site) keys were the same 2^32-long sequence read at an offset of (k0' - k0) * C0^-1 groups -- with 36 sites per
    step and a new seed every step, exact shifted repeats of half the mask bits inside one tensor's range were frequent.  With k1
    mixed in between the two multiplies, the streams at exactly that offset must be unrelated."""
    c0_inv = pow(C0, -1, 1 << 32)
    n = 1 << 20
    p = 0.1
    thr_s = int(p * 65536.0 + 0.5) - 32768
    pe = 1.0 - round(p * 65536) / 65536
    rng = np.random.default_rng(0)
    pairs = [((1234, 3), (1234, 4)), ((1234, 3), (1235, 3)), ((7, 0), (2 ** 40 + 1, 35))]
    pairs += [((int(rng.integers(1 << 40)), int(rng.integers(36))), (int(rng.integers(1 << 40)), int(rng.integers(36)))) for _ in range(9)]
    for (sa, ia), (sb, ib) in pairs:
        ka, kb = drop_key(sa, ia), drop_key(sb, ib)
        shift = ((int(kb[0]) - int(ka[0])) * c0_inv) % (1 << 32)        # (g + shift) * C0 + k0a == g * C0 + k0b  (mod 2^32)
        g = np.arange(n, dtype=np.uint64)
        aa, ba = drop_words(((g + shift) & 0xFFFFFFFF).astype(np.uint32), *ka)
        ab, bb = drop_words(g.astype(np.uint32), *kb)
        assert (aa == ab).mean() < 1e-4 and (ba == bb).mean() < 1e-4
        for wa, wb in ((aa, ab), (ba, bb)):
            for sh in (0, 16):
                ua = ((wa >> U32(sh)) & U32(0xFFFF)).astype(np.int64)
                ub = ((wb >> U32(sh)) & U32(0xFFFF)).astype(np.int64)
                ka_ = (np.where(ua >= 32768, ua - 65536, ua) >= thr_s) - pe
                kb_ = (np.where(ub >= 32768, ub - 65536, ub) >= thr_s) - pe
                z = (ka_ * kb_).mean() / (pe * (1 - pe)) * math.sqrt(n)
                assert abs(z) < 4.5, (sa, ia, sb, ib, sh, z)


# --------------------------------------------------------------------------------------------- the kernels use exactly this stream
@pytest.mark.gpu
@pytest.mark.parametrize("rows,cols,p,seed,site", [(512, 768, 0.1, 1234, 3), (197, 3072, 0.1, 2 ** 61 + 5, 35), (64, 64, 0.5, 0, 0),
                                                   (33, 1024, 0.9999, 77, 1), (1000, 384, 0.25, 42, 7)])
def test_exported_mask_is_the_restated_stream(rows, cols, p, seed, site):
    from vitssl_hip import ops
    dev = torch.device("cuda:0")
    got = ops.dropout_mask(rows, cols, ops.make_dropout(p, seed=seed, site=site), dev).cpu().numpy()
    assert np.array_equal(got, keep_mask(rows, cols, p, seed, site))
