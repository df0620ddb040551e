"""CPU checks of the oracle's fp8 (OCP e4m3fn) emulation: the quantiser against a table built from the format's
definition, the power-of-two weight-scale rule, and the fp8 forward mode against the fp32 / bf16 modes."""
import math

import numpy as np
import torch

from oracle import vit_oracle as O


def _e4m3_table():
    """All finite e4m3fn values from the format definition: 1 sign, 4 exponent (bias 7), 3 mantissa bits;
    exponent 0 = subnormal (m / 8 * 2^-6); S.1111.111 is NaN, there are no infinities."""
    vals = []
    for code in range(256):
        s, e, m = code >> 7, (code >> 3) & 15, code & 7
        if e == 15 and m == 7:
            continue
        v = (m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
        vals.append(-v if s else v)
    return np.array(sorted(set(vals)), dtype=np.float64)


def test_q8_is_round_to_nearest_even_saturating():
    tab = _e4m3_table()
    assert tab.max() == 448.0 and tab.min() == -448.0 and 2.0 ** -9 in tab
    g = torch.Generator().manual_seed(0)
    x = torch.randn(20000, generator=g) * torch.exp2(torch.randint(-12, 11, (20000,), generator=g).float())
    x = torch.cat([x, torch.tensor([0.0, 448.0, 449.0, 1e6, -1e6, 2.0 ** -10, 3 * 2.0 ** -10, 1.0625, 1.1875, 464.0])])
    got = O.q8(x).double().numpy()
    xs = np.clip(x.double().numpy(), -448.0, 448.0)
    idx = np.searchsorted(tab, xs)
    lo, hi = tab[np.clip(idx - 1, 0, len(tab) - 1)], tab[np.clip(idx, 0, len(tab) - 1)]
    near = np.where(np.abs(xs - lo) < np.abs(hi - xs), lo, hi)
    tie = np.abs(xs - lo) == np.abs(hi - xs)
    # ties go to the value whose mantissa is even = the one that is a multiple of twice the local step
    step = hi - lo
    even_lo = np.where(step > 0, np.round(lo / np.where(step > 0, step, 1)) % 2 == 0, True)
    want = np.where(tie, np.where(even_lo, lo, hi), near)
    assert np.array_equal(got, want)
    assert np.all(np.isin(got, tab))


def test_fp8_scale_exponent_rule():
    for amax in [448.0, 447.9, 449.0, 224.0, 0.02, 1e-6, 3.0, 1.75, 1.7500001, 0.875, 1e30]:
        k = O.fp8_scale_exp(amax)
        a32 = float(torch.tensor(amax, dtype=torch.float32))
        assert a32 * 2.0 ** k <= 448.0 < a32 * 2.0 ** (k + 1), (amax, k)
    assert O.fp8_scale_exp(0.0) == 0 and O.fp8_scale_exp(float("inf")) == 0 and O.fp8_scale_exp(1e-45) == 120


def test_fp8_block_mode_sits_between_bf16_and_garbage():
    torch.manual_seed(0)
    D, H, F = 128, 2, 256
    sd = {}
    for n in ("w_query", "w_key", "w_value", "final_linear"):
        sd[f"self_attention.{n}.weight"] = torch.randn(D, D) * 0.05
    sd["feed_forward.linear_in.weight"], sd["feed_forward.linear_in.bias"] = torch.randn(F, D) * 0.05, torch.randn(F) * 0.05
    sd["feed_forward.linear_out.weight"], sd["feed_forward.linear_out.bias"] = torch.randn(D, F) * 0.05, torch.randn(D) * 0.05
    for n in ("layer_norm1", "layer_norm2"):
        sd[n + ".weight"], sd[n + ".bias"] = torch.ones(D), torch.zeros(D)
    x = torch.randn(2, 10, D)
    out = {}
    for emu in (None, "bf16", "fp8"):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        y, _ = O.encoder_block(x, leaves, "", H, emu)
        y.square().mean().backward()
        out[emu] = (y.detach(), {k: v.grad for k, v in leaves.items()})

    def rel(a, b):
        return float((a - b).norm() / b.norm())

    e16, e8 = rel(out["bf16"][0], out[None][0]), rel(out["fp8"][0], out[None][0])
    assert e16 < 3e-3 and e16 < e8 < 4e-2
    for k in sd:
        assert rel(out["fp8"][1][k], out[None][1][k]) < 0.1, k
    # the shared QKV exponent: scaling w_query alone changes the grid w_key is quantised on
    sd2 = dict(sd)
    sd2["self_attention.w_query.weight"] = sd["self_attention.w_query.weight"] * 64
    k1 = O.fp8_scale_exp(max(float(sd[f"self_attention.{n}.weight"].abs().max()) for n in ("w_query", "w_key", "w_value")))
    k2 = O.fp8_scale_exp(max(float(sd2[f"self_attention.{n}.weight"].abs().max()) for n in ("w_query", "w_key", "w_value")))
    assert k2 == O.fp8_scale_exp(64 * float(sd['self_attention.w_query.weight'].abs().max())) and k1 - k2 in (5, 6)
    assert math.isfinite(float(O.encoder_block(x, sd2, "", H, "fp8")[0].sum()))


def test_engine_scale_rule_matches_the_oracle_rule():
    """The host-side scale update of the fp8 engine (torch ops on device tensors; here on the CPU) follows the oracle's
    exponent rule and produces exact powers of two."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-ssl_amd"))
    from vitssl_hip.engine import EncoderStack
    amax = torch.tensor([448.0, 449.0, 447.9, 224.0, 0.02, 1e-6, 3.0, 1.75, 1.7500001, 0.875, 6e-8, 1e30])
    for margin in (0, 1):
        got = EncoderStack._scale_from_amax(amax, margin)
        want = torch.tensor([2.0 ** (O.fp8_scale_exp(float(a)) - margin) for a in amax])
        assert torch.equal(got, want), (margin, got, want)
        mant, _ = torch.frexp(got)
        assert bool((mant == 0.5).all())
