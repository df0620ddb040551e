"""Round-4 GPU checks: the stream-K tail of the forward / input-gradient GEMMs (csrc/gemm_nt.hip: in the last, under-filled tile
round the workgroups without a tile take the last K-tiles of the round's tiles off their owners; whichever piece of a tile arrives
last adds the other's partial tile and runs the epilogue).  The cut changes the fp32 summation order of the cut tiles and nothing
else."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.fixture()
def streamk():
    from vitssl_hip import ops
    ops.set_nt_streamk(True)
    yield ops
    ops.set_nt_streamk(True)


@pytest.mark.parametrize("extra_tiles,N,K", [(100, 256, 4096), (37, 512, 2048), (150, 768, 3072)])
def test_streamk_tail_matches_whole_tiles(streamk, extra_tiles, N, K):
    """Every epilogue, cut against uncut (same kernels, whole tiles) and against an fp64 product; ragged last row tile; the arrival
    counters must be back at zero after every launch (the launches are repeated)."""
    ops = streamk
    from vitssl_hip import _lib as L
    lib = L.lib()
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    tn = (N + 255) // 256
    M = 256 * ((cus + extra_tiles + tn - 1) // tn) - 37           # a little over one round of 256-row tiles, ragged last tile
    g = torch.Generator().manual_seed(11)
    A = (torch.randn(M, K, generator=g) * 0.5).to(DEV).to(torch.bfloat16)
    B = (torch.randn(N, K, generator=g) * (2.0 / K ** 0.5)).to(DEV).to(torch.bfloat16)
    bias = (torch.randn(N, generator=g) * 0.1).to(DEV)
    res = torch.randn(M, N, generator=g).to(DEV)
    ref = A.double() @ B.double().t() + bias.double()
    drop = ops.make_dropout(0.1, seed=5, site=2)
    keep = ops.dropout_mask(M, N, drop, DEV).double()
    scale = 65536.0 / (65536 - round(0.1 * 65536))

    def run_all():
        out = {}
        o16 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        cs = torch.zeros(N, device=DEV)
        ops.gemm_nt(A, B, o16, L.EPI_BF16, bias=bias, colsum=cs)
        out["bf16"], out["bf16_cs"], out["sk_bf16"] = o16, cs, lib.vitssl_debug_last_nt_streamk()
        o32 = torch.empty(M, N, device=DEV)
        ops.gemm_nt(A, B, o32, L.EPI_F32, bias=bias)
        out["f32"] = o32
        orr = torch.empty(M, N, device=DEV)
        ops.gemm_nt(A, B, orr, L.EPI_RESID, bias=bias, aux=res, drop=drop)
        out["resid"], out["sk_resid"] = orr, lib.vitssl_debug_last_nt_streamk()
        gp, a = (torch.empty(M, N, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        ops.gemm_nt(A, B, gp, L.EPI_GELU, bias=bias, out1=a, drop=drop)
        out["gelu_gp"], out["gelu_a"], out["sk_gelu"] = gp, a, lib.vitssl_debug_last_nt_streamk()
        du = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        cs2 = torch.zeros(N, device=DEV)
        ops.gemm_nt(A, B, du, L.EPI_DGELU, aux=gp, colsum=cs2)
        out["dgelu"], out["dgelu_cs"] = du, cs2
        return out

    cut = run_all()
    again = run_all()                                            # counters were left at zero, slots are reused
    assert cut["sk_bf16"] > 0 and cut["sk_resid"] > 0 and cut["sk_gelu"] > 0, "the shapes of this test must trigger the cut"
    ops.set_nt_streamk(False)
    whole = run_all()
    assert whole["sk_bf16"] == 0 and whole["sk_gelu"] == 0
    for o in (cut, again):
        assert _rel(o["f32"], ref) < 2e-6 and _rel(o["f32"], whole["f32"]) < 1e-6
        assert _rel(o["bf16"].float(), ref) < 3e-3
        assert (o["bf16"] != whole["bf16"]).float().mean() < 2e-3          # a different fp32 summation order flips few bf16 roundings
        assert _rel(o["bf16_cs"], whole["bf16_cs"]) < 1e-5
        assert _rel(o["resid"], res.double() + ref * keep * scale) < 2e-6
        assert (o["gelu_a"] != whole["gelu_a"]).float().mean() < 2e-3 and (o["gelu_gp"] != whole["gelu_gp"]).float().mean() < 2e-3
        want = (A.double() @ B.double().t()) * o["gelu_gp"].double()
        assert _rel(o["dgelu"].float(), want) < 3e-3 and _rel(o["dgelu_cs"], want.sum(0)) < 2e-3
    # the same slots and counters, other data every time: a partial tile of an earlier launch must never be read (stale cache
    # lines of another XCD would show up here)
    o32 = torch.empty(M, N, device=DEV)
    ops.set_nt_streamk(True)
    for i in range(6):
        Ai = (A.float() * (1.0 + 0.37 * i) * (-1.0 if i & 1 else 1.0)).to(torch.bfloat16)
        ops.gemm_nt(Ai, B, o32, L.EPI_F32, bias=bias)
        assert lib.vitssl_debug_last_nt_streamk() > 0
        assert _rel(o32, Ai.double() @ B.double().t() + bias.double()) < 2e-6, i


def test_streamk_fp8_operands(streamk):
    ops = streamk
    from vitssl_hip import _lib as L
    lib = L.lib()
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    M, N, K = 256 * (cus + 90) - 64, 256, 2048
    g = torch.Generator().manual_seed(3)
    A8 = (torch.randn(M, K, generator=g)).to(DEV).to(torch.float8_e4m3fn)
    B8 = (torch.randn(N, K, generator=g)).to(DEV).to(torch.float8_e4m3fn)
    alpha = torch.tensor([0.125], device=DEV)
    ref = (A8.double() @ B8.double().t()) * 0.125
    o32 = torch.empty(M, N, device=DEV)
    ops.gemm_fp8_nt(A8, B8, o32, L.EPI_F32, alpha=alpha)
    assert lib.vitssl_debug_last_nt_streamk() > 0
    assert _rel(o32, ref) < 3e-5                     # (the K = 128 e4m3 MFMA's own accumulation: tests/test_gpu_fp8.py uses 2e-5 at K = 1024)
    ops.set_nt_streamk(False)
    o32w = torch.empty(M, N, device=DEV)
    ops.gemm_fp8_nt(A8, B8, o32w, L.EPI_F32, alpha=alpha)
    assert lib.vitssl_debug_last_nt_streamk() == 0 and _rel(o32, o32w) < 1e-5


def test_streamk_needs_its_workspace_and_stream(streamk):
    """No workspace, another stream, or a buffer that is too small: the launch runs whole tiles (never an error, never a cut)."""
    ops = streamk
    from vitssl_hip import _lib as L
    lib = L.lib()
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    M, N, K = 256 * (cus + 100), 256, 1024
    A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
    B = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    ops.gemm_nt(A, B, out, L.EPI_BF16)
    assert lib.vitssl_debug_last_nt_streamk() > 0
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    main_key = ops._NT_WS_ACTIVE[0]
    ops._NT_WS_ENABLED[0] = False                              # keep the registration of the main stream in force
    with torch.cuda.stream(side):
        out2 = torch.empty_like(out)
        ops.gemm_nt(A, B, out2, L.EPI_BF16)
        assert lib.vitssl_debug_last_nt_streamk() == 0          # registered for another stream: whole tiles
    side.synchronize()
    ops._NT_WS_ENABLED[0] = True
    assert ops._NT_WS_ACTIVE[0] == main_key
    assert (out != out2).float().mean() < 2e-3
    small = torch.empty(1 << 20, dtype=torch.uint8, device=DEV)
    with pytest.raises(L.VitsslError, match="vitssl_nt_workspace_bytes"):
        L.call("vitssl_set_nt_workspace", small.data_ptr(), small.numel(), None)
