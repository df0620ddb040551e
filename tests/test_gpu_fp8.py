"""fp8 (OCP e4m3fn) operand path of the forward Linear layers -- BASELINE.json configs[4].

Integer / byte work is bit-exact (the quantisers against torch's own float8_e4m3fn conversion, the per-tensor
power-of-two weight scales against the oracle's exponent rule); the GEMM is compared with an fp32 product of the
SAME e4m3 operands (products of e4m3 values are exact in fp32, so only the summation order differs: 2e-5);
blocks and models are compared with the oracle's emu="fp8" mode at the bf16 path's tolerances, and with the pure
fp32 oracle at the tolerance e4m3's 3 mantissa bits allow (stated per assert)."""
import pytest
import torch

from _util import rel_l2

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")
FP8 = torch.float8_e4m3fn


def _q8_torch(x32):
    return x32.clamp(-448.0, 448.0).to(FP8)


def _f32(t8):
    """e4m3 -> fp32 on the host (exact), back on the tensor's device"""
    return t8.cpu().float().to(t8.device)


@pytest.fixture
def fp8_operands():
    from vitssl_hip import engine
    engine.set_linear_operands("fp8")
    yield
    engine.set_linear_operands("bf16")


def test_quantize_fp8_bit_exact():
    from vitssl_hip import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 16, generator=g) * torch.exp2(torch.randint(-12, 10, (1 << 16,), generator=g).float())
    edge = torch.tensor([0.0, -0.0, 448.0, -448.0, 449.0, 464.0, 480.0, 1e4, -1e4, 2.0 ** -9, 2.0 ** -10, 1.5 * 2.0 ** -10,
                         3.0 * 2.0 ** -10, 2.0 ** -6, 0.0029, 1.0625, 1.125, 1.1875, 17.9, -3.3, 2.0 ** -20])
    x = torch.cat([edge, x])[: (1 << 16) + 5].to(torch.bfloat16)      # odd length: exercises the tail
    xd = x.to(DEV)
    y = torch.empty(x.shape, dtype=FP8, device=DEV)
    ops.quantize_fp8(xd, y)
    want = _q8_torch(x.float())
    assert torch.equal(y.cpu().view(torch.uint8), want.view(torch.uint8))


def test_fp8_weight_images_bit_exact():
    from oracle import vit_oracle as O
    from vitssl_hip import ops
    g = torch.Generator().manual_seed(1)
    shapes = [(384, 128), (128, 128), (512, 128), (128, 512), (5, 7), (70, 130)]
    scales = [0.02, 3.0, 448.0 / 4.0, 1e-6, 0.5, 10.0]
    srcs = [(torch.randn(*s, generator=g) * sc) for s, sc in zip(shapes, scales)]
    srcs[2][0, 0] = 448.0            # amax exactly on the 0.875 boundary of the exponent rule
    srcs.append(torch.zeros(64, 64))  # all-zero tensor: scale exponent 0
    dev = [s.to(DEV) for s in srcs]
    dst = [torch.empty(s.shape, dtype=FP8, device=DEV) for s in srcs]
    dst_t = [torch.empty(s.shape[::-1], dtype=FP8, device=DEV) for s in srcs]
    dst[1] = None                     # only the transposed image / only the plain one
    dst_t[3] = None
    plan = ops.Fp8WeightPlan()
    plan.run(list(zip(dev, dst, dst_t)))
    plan.run(list(zip(dev, dst, dst_t)))     # cached job table
    alpha = plan.alpha.cpu()
    for i, s in enumerate(srcs):
        k = O.fp8_scale_exp(float(s.abs().max()))
        assert float(alpha[i]) == 2.0 ** -k, (i, float(alpha[i]), k)
        want = _q8_torch(s * (2.0 ** k))
        if dst[i] is not None:
            assert torch.equal(dst[i].cpu().view(torch.uint8), want.view(torch.uint8)), i
        if dst_t[i] is not None:
            assert torch.equal(dst_t[i].cpu().view(torch.uint8), want.view(torch.uint8).t().contiguous()), i
        assert float(want.float().abs().max()) <= 448.0 and (float(s.abs().max()) == 0 or float(want.float().abs().max()) > 224.0 * 0.9)


def test_quantize_fp8_scaled_and_amax():
    from vitssl_hip import ops
    torch.manual_seed(2)
    x = (torch.randn(70001) * 3e-4).to(torch.bfloat16)
    xd = x.to(DEV)
    y = torch.empty(x.shape, dtype=FP8, device=DEV)
    scale = torch.tensor([2.0 ** 17], device=DEV)
    amax = torch.tensor([1e-5], device=DEV)            # a smaller running maximum is replaced
    ops.quantize_fp8(xd, y, scale=scale, amax=amax)
    assert torch.equal(y.cpu().view(torch.uint8), _q8_torch(x.float() * 2.0 ** 17).view(torch.uint8))
    assert float(amax) == float(x.float().abs().max())
    big = torch.tensor([1.0], device=DEV)              # a larger one is kept
    ops.quantize_fp8(xd, y, scale=scale, amax=big)
    assert float(big) == 1.0


def _same_e4m3(a, b):
    """Byte equality of two e4m3 tensors, +0 and -0 taken as equal (a dropped element is an exact +0 in the kernels; the
    torch expression `x * keep` leaves the sign of x on its zero)."""
    a, b = a.cpu().view(torch.uint8), b.cpu().view(torch.uint8)
    return (a == b) | (((a & 0x7F) == 0) & ((b & 0x7F) == 0))


def test_layernorm_bwd_fp8_image():
    from vitssl_hip import ops
    torch.manual_seed(6)
    rows, cols = 500, 768
    x = torch.randn(rows, cols, device=DEV)
    dy = (torch.randn(rows, cols, device=DEV) * 1e-3).to(torch.bfloat16)
    gres = torch.randn(rows, cols, device=DEV) * 1e-3
    gamma = (1 + 0.1 * torch.randn(cols)).to(DEV)
    mean, rstd = x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
    drop = ops.make_dropout(0.1, 5, 2)
    outs = []
    for fp8 in (False, True):
        go, gm = torch.empty(rows, cols, device=DEV), torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
        dg, db, cs = (torch.zeros(cols, device=DEV) for _ in range(3))
        if fp8:
            gm8 = torch.empty(rows, cols, dtype=FP8, device=DEV)
            scale, amax = torch.tensor([2.0 ** 14], device=DEV), torch.zeros(1, device=DEV)
            ops.layernorm_bwd_fp8(dy, x, mean, rstd, gamma, gres, go, gm, gm8, scale, amax, dg, db, cs, drop)
        else:
            ops.layernorm_bwd(dy, x, mean, rstd, gamma, gres, go, gm, dg, db, cs, drop)
        outs.append((go, gm, dg, db, cs))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])   # g_out and gm: the bf16 kernel's result
    for a, b in zip(outs[0][2:], outs[1][2:]):
        assert rel_l2(a, b) < 1e-5                                      # column reductions: fp32 atomics, order varies
    keep = ops.dropout_mask(rows, cols, drop, DEV).float()
    gm32 = outs[1][0] * keep / 0.9                                      # what the kernel held in fp32
    assert abs(float(amax) - float(gm32.abs().max())) <= 1e-4 * float(amax)   # the reference is recomputed in another order
    want = _q8_torch((gm32 * 2.0 ** 14).cpu())
    same = _same_e4m3(gm8, want).float().mean()
    assert float(same) > 0.999
    # the standalone mask + cast form
    g = torch.randn(rows, cols, device=DEV) * 1e-3
    gm, gm8 = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV), torch.empty(rows, cols, dtype=FP8, device=DEV)
    amax.zero_()
    ops.grad_mask_cast_fp8(g, gm, gm8, scale, amax, None, drop)
    # (the kernel's survivor scale is 65536 / (65536 - round(p 65536)), not exactly 1 / 0.9: a rounding flips now and then)
    same = _same_e4m3(gm8, _q8_torch((g * keep / 0.9 * 2.0 ** 14).cpu())).float().mean()
    assert float(same) > 0.999
    assert abs(float(amax) - float((g * keep / 0.9).abs().max())) <= 1e-4 * float(amax)


@pytest.mark.parametrize("cols", [128, 384, 768, 1024])
def test_layernorm_fwd_fp8_images(cols):
    from vitssl_hip import ops
    torch.manual_seed(cols)
    rows = 301
    x = (torch.randn(rows, cols) * 3 + 0.5).to(DEV)
    gamma, beta = (1 + 0.2 * torch.randn(cols)).to(DEV), (0.1 * torch.randn(cols)).to(DEV)
    y = torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
    y2 = torch.empty_like(y)
    y8 = torch.empty(rows, cols, dtype=FP8, device=DEV)
    m, r = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    ops.layernorm_fwd(x, gamma, beta, y2, m, r)
    ops.layernorm_fwd_fp8(x, gamma, beta, y, y8, m, r)
    assert torch.equal(y, y2)                                          # the bf16 image is the bf16 kernel's
    ref = torch.nn.functional.layer_norm(x, (cols,), gamma, beta)
    want = _q8_torch(ref.cpu())
    same = (y8.cpu().view(torch.uint8) == want.view(torch.uint8)).float().mean()
    assert float(same) > 0.995                                         # fp32 ulp differences flip a rounding now and then
    assert rel_l2(_f32(y8), ref) < 4e-2                              # e4m3: relative step 2^-3, rms error ~ 2^-3 / sqrt(12)


def test_attention_fwd_fp8_image():
    from vitssl_hip import ops
    torch.manual_seed(4)
    B, N, H, dh = 3, 197, 4, 64
    qkv = torch.randn(B * N, 3 * H * dh).to(torch.bfloat16).to(DEV)
    out, out2 = (torch.empty(B * N, H * dh, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    out8 = torch.empty(B * N, H * dh, dtype=FP8, device=DEV)
    lse = torch.empty(B, H, N, device=DEV)
    ops.attn_fwd(qkv, out2, lse, B, N, H, dh)
    ops.attn_fwd(qkv, out, lse, B, N, H, dh, out_fp8=out8)
    assert torch.equal(out, out2)
    q, k, v = (t.transpose(1, 2) for t in qkv.float().view(B, N, 3, H, dh).unbind(2))
    ref = torch.softmax(q @ k.transpose(-1, -2) / 8.0, -1) @ v
    ref = ref.transpose(1, 2).reshape(B * N, H * dh)
    assert rel_l2(_f32(out8), ref) < 4e-2
    # the kernel rounds the probabilities to bf16 before P.V: compare bytes against its own bf16 output, which is
    # a finer rounding of the same fp32 values (a bf16 value re-rounded to e4m3 differs from the direct rounding
    # only at double-rounding ties)
    same = (out8.cpu().view(torch.uint8) == _q8_torch(out.cpu().float()).view(torch.uint8)).float().mean()
    assert float(same) > 0.95    # expected ~0.97: 1 in 2^5 values sits where the two roundings disagree


def test_attention_bwd_fp8_image():
    from vitssl_hip import ops
    torch.manual_seed(5)
    B, N, H, dh = 2, 197, 3, 64
    qkv = torch.randn(B * N, 3 * H * dh).to(torch.bfloat16).to(DEV)
    out = torch.empty(B * N, H * dh, dtype=torch.bfloat16, device=DEV)
    lse = torch.empty(B, H, N, device=DEV)
    ops.attn_fwd(qkv, out, lse, B, N, H, dh)
    dout = (torch.randn(B * N, H * dh) * 1e-3).to(torch.bfloat16).to(DEV)
    d0, d1 = (torch.empty(B * N, 3 * H * dh, dtype=torch.bfloat16, device=DEV) for _ in range(2))
    d8 = torch.empty(B * N, 3 * H * dh, dtype=FP8, device=DEV)
    delta = torch.empty(B, H, N, device=DEV)
    ops.attn_bwd(qkv, out, dout, lse, d0, delta, B, N, H, dh)
    scale, amax = torch.tensor([2.0 ** 13], device=DEV), torch.zeros(1, device=DEV)
    ops.attn_bwd(qkv, out, dout, lse, d1, delta, B, N, H, dh, dqkv_fp8=d8, scale=scale, amax=amax)
    assert torch.equal(d0, d1)
    # the image is made from the fp32 accumulators, the bf16 tensor is a finer rounding of the same values
    assert abs(float(amax) - float(d1.float().abs().max())) <= 2.0 ** -8 * float(amax)
    same = (d8.cpu().view(torch.uint8) == _q8_torch(d1.cpu().float() * 2.0 ** 13).view(torch.uint8)).float().mean()
    assert float(same) > 0.95
    assert rel_l2(_f32(d8) / 2.0 ** 13, d1) < 4e-2


SHAPES = [(300, 128, 128), (1000, 384, 256), (517, 264, 1024), (4096, 512, 512)]   # N % 8 == 0 (fp8 operands); ragged M, N


@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_fp8_epilogues(M, N, K):
    from vitssl_hip import _lib as L, ops
    torch.manual_seed(M + N + K)
    A = _q8_torch(torch.randn(M, K) * 2).to(DEV)
    Bw = _q8_torch(torch.randn(N, K) * 100).to(DEV)
    alpha = torch.tensor([2.0 ** -7], device=DEV)
    bias = torch.randn(N, device=DEV)
    ref = (_f32(A) @ _f32(Bw).t()) * alpha + bias
    out32 = torch.empty(M, N, device=DEV)
    ops.gemm_fp8_nt(A, Bw, out32, L.EPI_F32, alpha=alpha, bias=bias)
    assert rel_l2(out32, ref) < 2e-5
    out16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_fp8_nt(A, Bw, out16, L.EPI_BF16, alpha=alpha, bias=bias)
    assert rel_l2(out16, ref) < 4e-3 and torch.equal(out16, out32.to(torch.bfloat16))
    # no alpha / no bias
    ops.gemm_fp8_nt(A, Bw, out32, L.EPI_F32)
    assert rel_l2(out32, _f32(A) @ _f32(Bw).t()) < 2e-5
    # residual + dropout: the mask stream is the bf16 kernel's (same element index)
    res = torch.randn(M, N, device=DEV)
    drop = ops.make_dropout(0.25, 11, 3)
    keep = ops.dropout_mask(M, N, drop, DEV).float()
    ops.gemm_fp8_nt(A, Bw, out32, L.EPI_RESID, alpha=alpha, bias=bias, aux=res, drop=drop)
    assert rel_l2(out32, res + ref * keep / 0.75) < 2e-5
    # GELU: g' image, bf16 a, e4m3 a
    u = ref.to(torch.bfloat16).float()
    a_ref = torch.nn.functional.gelu(u) * keep / 0.75
    gp = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    a16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    a8 = torch.empty(M, N, dtype=FP8, device=DEV)
    ops.gemm_fp8_nt(A, Bw, gp, L.EPI_GELU, alpha=alpha, bias=bias, out1=a16, out_fp8=a8, drop=drop)
    assert rel_l2(a16, a_ref) < 4e-3
    assert rel_l2(_f32(a8), a_ref) < 4e-2
    same = _same_e4m3(a8, _q8_torch(a_ref.cpu())).float().mean()
    assert float(same) > 0.99
    cdf = 0.5 * (1 + torch.erf(u / 2 ** 0.5))
    gp_ref = (cdf + u * torch.exp(-0.5 * u * u) / (2 * torch.pi) ** 0.5) * keep / 0.75
    assert rel_l2(gp, gp_ref) < 4e-3
    a16b = torch.empty_like(a16)
    ops.gemm_fp8_nt(A, Bw, gp, L.EPI_GELU, alpha=alpha, bias=bias, out1=a16b, drop=drop)   # without the e4m3 image
    assert torch.equal(a16, a16b)


def test_gemm_fp8_refuses_n_not_multiple_of_8():
    """fp8 operands run the ping-pong kernel only; its line-shaped stores move 8 columns per lane (include/vitssl_hip.h)."""
    from vitssl_hip import _lib as L, ops
    A = _q8_torch(torch.randn(64, 128)).to(DEV)
    Bw = _q8_torch(torch.randn(260, 128)).to(DEV)
    out = torch.empty(64, 260, device=DEV)
    with pytest.raises(Exception, match="multiple of 8"):
        ops.gemm_fp8_nt(A, Bw, out, L.EPI_F32)


def test_gemm_fp8_identity_layout():
    """A = I against an asymmetric B: catches a row/column swap or a k-permutation that differs between operands."""
    from vitssl_hip import _lib as L, ops
    K = 256
    A = torch.eye(K)[:, :K].to(FP8).to(DEV)
    Bm = torch.zeros(384, K)
    for n in range(384):
        for j in range(4):
            Bm[n, (7 * n + 13 * j) % K] = float((n + 3 * j) % 15 + 1)
    out = torch.empty(K, 384, device=DEV)
    ops.gemm_fp8_nt(A, Bm.to(FP8).to(DEV), out, L.EPI_F32)
    assert torch.equal(out.cpu(), Bm.t().contiguous())


@pytest.mark.parametrize("M,N,K", [(600, 512, 256), (1000, 264, 128)])
def test_gemm_fp8_dgelu_with_scaled_image(M, N, K):
    """The input-gradient form: dY (e4m3, scaled) . W^T image, dGELU epilogue, column sums, scaled e4m3 image + max|.|."""
    from vitssl_hip import _lib as L, ops
    torch.manual_seed(M + N)
    s_in, s_out = 2.0 ** 12, 2.0 ** 9
    dY = torch.randn(M, K) * 2e-3
    A = _q8_torch(dY * s_in).to(DEV)
    Bw = _q8_torch(torch.randn(N, K) * 100).to(DEV)
    alpha, alpha2 = torch.tensor([2.0 ** -7], device=DEV), torch.tensor([1.0 / s_in], device=DEV)
    gp = (torch.rand(M, N) * 1.2).to(torch.bfloat16).to(DEV)
    ref = (_f32(A) @ _f32(Bw).t()) * (2.0 ** -7 / s_in) * gp.float()
    du = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    du8 = torch.empty(M, N, dtype=FP8, device=DEV)
    cs = torch.zeros(N, device=DEV)
    scale, amax = torch.tensor([s_out], device=DEV), torch.zeros(1, device=DEV)
    ops.gemm_fp8_nt(A, Bw, du, L.EPI_DGELU, alpha=alpha, alpha2=alpha2, aux=gp, colsum=cs, out_fp8=du8, out_scale=scale, out_amax=amax)
    assert rel_l2(du, ref) < 4e-3
    assert rel_l2(cs, ref.sum(0)) < 1e-4
    assert abs(float(amax) - float(ref.abs().max())) <= 1e-4 * float(amax)
    assert rel_l2(_f32(du8) / s_out, ref) < 4e-2
    same = (du8.cpu().view(torch.uint8) == _q8_torch((ref * s_out).cpu()).view(torch.uint8)).float().mean()
    assert float(same) > 0.99


@pytest.mark.parametrize("M,N1,N2", [(300, 128, 128), (1000, 384, 256), (4100, 768, 272), (25088, 1024, 1024)])
def test_gemm_fp8_tn_weight_gradient(M, N1, N2):
    """C += alpha alpha2 A8^T B8: exact e4m3 products, fp32 accumulation -- against an fp64 product of the same bytes."""
    from vitssl_hip import ops
    torch.manual_seed(M + N1)
    A = _q8_torch(torch.randn(M, N1) * 3).to(DEV)
    B = _q8_torch(torch.randn(M, N2) * 2).to(DEV)
    C0 = torch.randn(N1, N2)
    Cd = C0.clone().to(DEV)
    alpha, alpha2 = torch.tensor([2.0 ** -5], device=DEV), torch.tensor([0.75], device=DEV)
    ops.gemm_fp8_tn(A, B, Cd, alpha=alpha, alpha2=alpha2)
    ref = C0.double() + (A.cpu().float().double().t() @ B.cpu().float().double()) * (2.0 ** -5 * 0.75)
    assert rel_l2(Cd, ref) < 2e-5
    Cd2 = torch.zeros(N1, N2, device=DEV)
    ops.gemm_fp8_tn(A, B, Cd2)                       # no scalars
    assert rel_l2(Cd2, A.cpu().float().double().t() @ B.cpu().float().double()) < 2e-5


def test_gemm_fp8_rejects_bad_arguments():
    from vitssl_hip import _lib as L, ops
    A = torch.zeros(64, 192, dtype=FP8, device=DEV)
    Bw = torch.zeros(64, 192, dtype=FP8, device=DEV)
    with pytest.raises(L.VitsslError):
        ops.gemm_fp8_nt(A, Bw, torch.empty(64, 64, device=DEV), L.EPI_F32)               # K % 128
    A = torch.zeros(64, 128, dtype=FP8, device=DEV)
    Bw = torch.zeros(64, 128, dtype=FP8, device=DEV)
    with pytest.raises(L.VitsslError):
        ops.gemm_fp8_nt(A, Bw, torch.empty(64, 64, dtype=torch.bfloat16, device=DEV), L.EPI_DGELU)     # needs aux
    with pytest.raises(L.VitsslError):
        ops.gemm_fp8_nt(A, Bw, torch.empty(64, 64, device=DEV), L.EPI_EMBED)
    with pytest.raises(L.VitsslError):
        ops.gemm_fp8_nt(A.to(torch.bfloat16), Bw, torch.empty(64, 64, device=DEV), L.EPI_F32)


def test_encoder_block_fp8_matches_oracle(fp8_operands):
    from vit_core import EncoderBlock
    from oracle import vit_oracle as O
    torch.manual_seed(3)
    blk = EncoderBlock(d_model=128, num_heads=2, mlp_dim=256, dropout=0.0)
    sd = {k: v.detach().clone() for k, v in blk.state_dict().items()}
    blk = blk.to(DEV).train()
    x = torch.randn(3, 20, 128)
    scales = []
    for step in range(2):      # first backward: scales from the tensors themselves; second: delayed (previous step's max, one bit of headroom)
        blk.zero_grad(set_to_none=True)
        xd = x.to(DEV).requires_grad_(True)
        y, _ = blk(xd)
        y.square().mean().backward()
        gs = blk._runner.stack.fp8_grad_scales().cpu()
        scales.append(gs)
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = x.clone().requires_grad_(True)
        ry, _ = O.encoder_block(xr, leaves, "", 2, emu="fp8", fp8_gscales=gs[0].tolist())
        ry.square().mean().backward()
        assert rel_l2(y, ry) < 1e-2                      # same quantisation points: the bf16 path's tolerance
        assert rel_l2(xd.grad, xr.grad) < 5e-2
        for k, p in blk.named_parameters():
            assert rel_l2(p.grad, leaves[k].grad) < 5e-2, (step, k, rel_l2(p.grad, leaves[k].grad))
    assert torch.equal(scales[1], scales[0] / 2)         # same gradients, one bit of headroom
    assert bool((scales[0] > 1.0).all())                  # gradients of a mean-square loss are far below 1: they need the scale
    fy, _ = O.encoder_block(x, sd, "", 2)
    assert 1e-3 < rel_l2(y, fy) < 5e-2               # and against pure fp32: e4m3 operands cost ~1.5 % here (bf16: 0.1 %)
    # against the fp32 gradients: e4m3 gradient operands add their 2^-4 relative rounding
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fy, _ = O.encoder_block(x, leaves, "", 2)
    fy.square().mean().backward()
    for k, p in blk.named_parameters():
        assert rel_l2(p.grad, leaves[k].grad) < 0.12, (k, rel_l2(p.grad, leaves[k].grad))


def test_simmim_fp8_matches_oracle_and_trains(fp8_operands):
    from vit_core.ssl.simmim import SimMIMViT
    from vit_core.ssl.simmim.masking import draw_mask
    from vitssl_hip.optim import FusedAdamW
    from oracle import vit_oracle as O
    torch.manual_seed(11)
    model = SimMIMViT(num_blocks=2, input_shape=(3, 64, 64), embed_dim=128, patch_size=16, num_heads=2, mlp_dim=256,
                      dropout=0.0, mask_ratio=0.6)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    x = torch.rand(4, 3, 64, 64, generator=torch.Generator().manual_seed(5))
    torch.manual_seed(9)
    pred, tgt = model(x.to(DEV))
    loss = torch.nn.functional.l1_loss(pred, tgt)
    loss.backward()
    torch.manual_seed(9)
    mask = draw_mask(4, 16, 0.6)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pe, te = O.simmim_forward(leaves, x, mask, 16, 2, emu="fp8", fp8_gscales=model.runtime().stack.fp8_grad_scales().cpu().tolist())
    assert torch.equal(tgt.cpu(), te) and rel_l2(pred, pe) < 1e-2
    wl = O.l1_loss_mean(pe, te)
    wl.backward()
    assert abs(float(loss) - float(wl)) < 1e-2 * float(wl)
    for k, p in model.named_parameters():
        assert rel_l2(p.grad, leaves[k].grad) < 6e-2, (k, rel_l2(p.grad, leaves[k].grad))
    # the weight images follow the optimizer: a few fused steps on one batch reduce the loss
    opt = FusedAdamW(model.flat_store(), lr=1e-3, weight_decay=0.0)
    torch.manual_seed(9)
    l0 = float(model.train_step(x.to(DEV), opt))
    for _ in range(8):
        torch.manual_seed(9)
        l1 = float(model.train_step(x.to(DEV), opt))
    assert abs(l0 - float(loss)) < 1e-4 * abs(float(loss)) and l1 < l0


def test_vit_l_shaped_blocks_fp8(fp8_operands):
    """configs[4] geometry: D = 1024, 16 heads, F = 4096, N = 196, two blocks, two images, against the oracle."""
    from vit_core.ssl.simmim import SimMIMViT
    from vit_core.ssl.simmim.masking import draw_mask
    from oracle import vit_oracle as O
    torch.manual_seed(7)
    model = SimMIMViT(num_blocks=2, input_shape=(3, 224, 224), embed_dim=1024, patch_size=16, num_heads=16, mlp_dim=4096,
                      dropout=0.0, mask_ratio=0.6)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    x = torch.rand(2, 3, 224, 224, generator=torch.Generator().manual_seed(8))
    torch.manual_seed(9)
    pred, tgt = model(x.to(DEV))
    loss = torch.nn.functional.l1_loss(pred, tgt)
    loss.backward()
    torch.manual_seed(9)
    mask = draw_mask(2, 196, 0.6)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pe, te = O.simmim_forward(leaves, x, mask, 16, 16, emu="fp8", fp8_gscales=model.runtime().stack.fp8_grad_scales().cpu().tolist())
    assert torch.equal(tgt.cpu(), te) and rel_l2(pred, pe) < 1e-2
    wl = O.l1_loss_mean(pe, te)
    wl.backward()
    assert abs(float(loss) - float(wl)) < 1e-2 * float(wl)
    for k, p in model.named_parameters():
        assert rel_l2(p.grad, leaves[k].grad) < 6e-2, (k, rel_l2(p.grad, leaves[k].grad))
    with torch.no_grad():
        p32, _ = O.simmim_forward(sd, x, mask, 16, 16)
    assert rel_l2(pred, p32) < 8e-2                   # vs pure fp32: e4m3 operand rounding through two blocks


def test_dino_step_with_fp8_operands():
    """The DINO student / teacher backbones run on the same encoder stack: a fused step with e4m3 operands gives the
    bf16 step's loss within the operand rounding, and keeps training."""
    from vit_core.ssl.dino import DINOViT
    from vit_core.ssl.dino.loss import DINOLoss
    from vitssl_hip import engine
    from vitssl_hip.optim import FusedAdamW
    losses = {}
    for mode in ("bf16", "fp8"):
        engine.set_linear_operands(mode)
        try:
            torch.manual_seed(0)
            model = DINOViT(num_blocks=2, input_shape=(3, 64, 64), embed_dim=128, patch_size=16, num_heads=2, mlp_dim=256,
                            dropout=0.0, output_dim=512, center_momentum=0.9).to(DEV).train()
            g = torch.Generator().manual_seed(3)
            views = [torch.rand(4, 3, 64, 64, generator=g).to(DEV) for _ in range(2)] + [torch.rand(4, 3, 32, 32, generator=g).to(DEV) for _ in range(3)]
            crit = DINOLoss(0.04, 0.1)
            opt = FusedAdamW(model.trainable_store(), lr=1e-3, weight_decay=0.0)
            ls = [float(model.train_step(views, 2, crit, opt, None, teacher_momentum=0.99)) for _ in range(4)]
            losses[mode] = ls
        finally:
            engine.set_linear_operands("bf16")
    assert all(l == l and l > 0 for l in losses["fp8"])
    assert abs(losses["fp8"][0] - losses["bf16"][0]) < 5e-2 * abs(losses["bf16"][0])


def test_dino_fp8_gradients_follow_bf16_with_per_slot_scales():
    """DINO calls the student stack's backward twice per step (local crops, then global crops: different row counts and
    gradient magnitudes).  The delayed gradient scales are kept per slot, so the SECOND step (delayed scales in force for
    both passes) still gives the bf16 path's parameter gradients within the e4m3 operand band; weights are frozen
    (lr = 0) so both modes differentiate the same function."""
    from vit_core.ssl.dino import DINOViT
    from vit_core.ssl.dino.loss import DINOLoss
    from vitssl_hip import engine
    from vitssl_hip.optim import FusedAdamW
    grads, scales = {}, {}
    for mode in ("bf16", "fp8"):
        engine.set_linear_operands(mode)
        try:
            torch.manual_seed(0)
            model = DINOViT(num_blocks=2, input_shape=(3, 64, 64), embed_dim=128, patch_size=16, num_heads=2, mlp_dim=256,
                            dropout=0.0, output_dim=512, center_momentum=0.9).to(DEV).train()
            g = torch.Generator().manual_seed(3)
            # the local crops carry a 16x larger input scale: their gradients differ from the global ones by far more than 2x
            views = [torch.rand(4, 3, 64, 64, generator=g).to(DEV) for _ in range(2)] + \
                    [(16.0 * torch.rand(4, 3, 32, 32, generator=g)).to(DEV) for _ in range(3)]
            crit = DINOLoss(0.04, 0.1)
            st = model.trainable_store()
            opt = FusedAdamW(st, lr=0.0, weight_decay=0.0)
            for _ in range(2):
                model.train_step(views, 2, crit, opt, None, teacher_momentum=1.0)
            grads[mode] = {k: st.gview(k).clone() for k in st.names}
            if mode == "fp8":
                stack = model.runtime().bb["student"].stack
                scales = {slot: stack.fp8_grad_scales(slot) for slot in stack._gs_by_slot}
        finally:
            engine.set_linear_operands("bf16")
    assert sorted(scales) == ["g", "l"], list(scales)               # one state per backward pass of the step
    assert not torch.equal(scales["g"], scales["l"])                # and the two passes did settle on different scales
    for k, gb in grads["bf16"].items():
        if float(gb.norm()) == 0.0:
            continue
        assert rel_l2(grads["fp8"][k], gb) < 0.12, (k, rel_l2(grads["fp8"][k], gb))


def test_simmim_fp8_with_dropout_matches_oracle(fp8_operands):
    """Dropout on (p = 0.1, the engine's counter-based masks exported to the oracle) together with e4m3 operands:
    forward outputs and every gradient, second (delayed-scale) step."""
    from vit_core import _runtime as R
    from vit_core.ssl.simmim import SimMIMViT
    from vit_core.ssl.simmim.masking import draw_mask
    from vitssl_hip import ops
    from oracle import vit_oracle as O
    p, B, D, H, F, L, img, patch = 0.1, 4, 128, 2, 256, 2, 64, 16
    N = (img // patch) ** 2
    torch.manual_seed(21)
    model = SimMIMViT(num_blocks=L, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H, mlp_dim=F,
                      dropout=p, mask_ratio=0.6)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(DEV).train()
    x = torch.rand(B, 3, img, img, generator=torch.Generator().manual_seed(4))
    for step in range(2):
        model.zero_grad(set_to_none=True)
        torch.manual_seed(33)
        pred, tgt = model(x.to(DEV))
        loss = torch.nn.functional.l1_loss(pred, tgt)
        loss.backward()
    torch.manual_seed(33)
    mask = draw_mask(B, N, 0.6)
    seed = R.next_seed()
    keeps = [[ops.dropout_mask(B * N, cols, ops.make_dropout(p, seed, 3 * i + which), DEV).float().cpu().view(B, N, cols)
              for which, cols in ((0, D), (1, F), (2, D))] for i in range(L)]
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pe, te = O.simmim_forward(leaves, x, mask, patch, H, emu="fp8", keeps=keeps, p_drop=round(p * 65536) / 65536,
                              fp8_gscales=model.runtime().stack.fp8_grad_scales().cpu().tolist())
    assert torch.equal(tgt.cpu(), te) and rel_l2(pred, pe) < 1e-2
    wl = O.l1_loss_mean(pe, te)
    wl.backward()
    assert abs(float(loss.detach()) - float(wl.detach())) < 1e-2 * float(wl.detach())
    for k, prm in model.named_parameters():
        assert rel_l2(prm.grad, leaves[k].grad) < 6e-2, (k, rel_l2(prm.grad, leaves[k].grad))


def test_gemm_fp8_tn_batch_matches_single_launches():
    """vitssl_gemm_fp8_tn_batch: every job's C += alpha alpha2 A8^T B8 against one vitssl_gemm_fp8_tn per job (same e4m3 products,
    another split of the rows: fp32 summation order only) and against an fp64 product of the same bytes."""
    from vitssl_hip import ops
    torch.manual_seed(5)
    M = 2560
    dims = [(1024, 4096), (4096, 1024), (1024, 1024), (272, 1024)]
    jobs, singles, refs = [], [], []
    for k, (N1, N2) in enumerate(dims):
        A = _q8_torch(torch.randn(M, N1) * 3).to(DEV)
        B = _q8_torch(torch.randn(M, N2) * 2).to(DEV)
        C0 = torch.randn(N1, N2)
        al = torch.tensor([2.0 ** -(3 + k)], device=DEV) if k != 2 else None
        al2 = torch.tensor([2.0 ** -5], device=DEV)
        f = (float(al) if al is not None else 1.0) * float(al2)
        refs.append(C0.double() + f * (_f32(A).double().cpu().t() @ _f32(B).double().cpu()))
        jobs.append((A, B, C0.clone().to(DEV), al, al2))
        c1 = C0.clone().to(DEV)
        ops.gemm_fp8_tn(A, B, c1, alpha=al, alpha2=al2)
        singles.append(c1)
    ops.gemm_fp8_tn_batch(jobs)
    for (_, _, Cd, _, _), ref, c1 in zip(jobs, refs, singles):
        scale = float(ref.abs().max())
        assert float((Cd.double().cpu() - ref).abs().max()) < 2e-5 * scale
        assert float((Cd - c1).abs().max()) < 1e-5 * scale
