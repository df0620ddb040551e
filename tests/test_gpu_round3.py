"""Round-3 GPU checks: CU reservation as explicit library state."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_reserved_cus_take_effect_after_a_forward(monkeypatch):
    """Round-2 finding: the reserve was an environment variable cached at the first GEMM launch, so a reducer built after
    any forward was ignored.  Now: forward first (full grid), THEN a GradReducer on a 2-rank world -> the next persistent
    NT grid leaves 8 CUs alone; results unchanged; resetting restores the full grid."""
    import torch.distributed as dist
    from vitssl_hip import _lib as L, ops
    from vitssl_hip.engine import GradReducer
    lib = L.lib()
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    before = lib.vitssl_get_reserved_cus()
    try:
        L.call("vitssl_set_reserved_cus", 0)
        torch.manual_seed(0)
        M, N, K = 256 * (cus + 40), 256, 256            # more 256-row tiles than CUs: the persistent grid is capped by the CU count
        A = torch.randn(M, K, device=DEV).to(torch.bfloat16)
        B = (torch.randn(N, K, device=DEV) * 0.05).to(torch.bfloat16)
        out0 = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm_nt(A, B, out0, L.EPI_BF16)
        assert lib.vitssl_debug_last_nt_grid() == cus
        monkeypatch.delenv("VITSSL_RESERVE_CUS", raising=False)
        monkeypatch.setattr(dist, "is_initialized", lambda: True)
        monkeypatch.setattr(dist, "get_world_size", lambda group=None: 2)
        GradReducer(torch.zeros(1024, device=DEV))
        assert lib.vitssl_get_reserved_cus() == 8
        out1 = torch.empty_like(out0)
        ops.gemm_nt(A, B, out1, L.EPI_BF16)
        assert lib.vitssl_debug_last_nt_grid() == cus - 8
        assert torch.equal(out0, out1)
        # weight-gradient split count and the LayerNorm-backward grid follow the same state: results stay right
        dY = torch.randn(4096, 256, device=DEV).to(torch.bfloat16)
        X = torch.randn(4096, 128, device=DEV).to(torch.bfloat16)
        C = torch.zeros(256, 128, device=DEV)
        ops.gemm_tn(dY, X, C)
        ref = dY.float().t() @ X.float()
        assert float((C - ref).norm() / ref.norm()) < 1e-5
        # the persistent attention forward owns a CU per workgroup for the whole launch: its grid leaves the reserve alone too
        Bq, Hq, Nq = 40, 8, 196                          # 320 (batch, head) items > CUs
        qkv = (torch.randn(Bq * Nq, 3 * Hq * 64, device=DEV) * 0.5).to(torch.bfloat16)
        o1, o0 = (torch.empty(Bq * Nq, Hq * 64, dtype=torch.bfloat16, device=DEV) for _ in range(2))
        lse1, lse0 = (torch.empty(Bq, Hq, Nq, device=DEV) for _ in range(2))
        ops.attn_fwd(qkv, o1, lse1, Bq, Nq, Hq, 64)
        assert lib.vitssl_debug_last_attn_fwd_grid() == cus - 8
        L.call("vitssl_set_reserved_cus", 0)
        ops.attn_fwd(qkv, o0, lse0, Bq, Nq, Hq, 64)
        assert lib.vitssl_debug_last_attn_fwd_grid() == cus
        assert torch.equal(o0, o1) and torch.equal(lse0, lse1)
        ops.gemm_nt(A, B, out1, L.EPI_BF16)
        assert lib.vitssl_debug_last_nt_grid() == cus
        with pytest.raises(L.VitsslError, match="negative"):
            L.call("vitssl_set_reserved_cus", -3)
    finally:
        L.call("vitssl_set_reserved_cus", before)


def test_nt_line_shaped_epilogue_matches_plain_kernel_on_ragged_shapes():
    """The ping-pong kernel's epilogue moves every 16-row tile through a per-wave LDS window and stores / loads whole 128-byte lines
    (csrc/gemm_nt.hip, NT_LDS_T); N % 8 != 0 and tiny grids run the round-1 kernel with accumulator-layout accesses.  Shapes whose
    M, N are ragged against the 256 x 256 tile in every way (last row tile 1..255 rows, last column tile 8..248 columns, several
    tile rounds per workgroup at K = 64): bf16 / fp32 outputs against an fp64 product, residual + dropout against the exported mask,
    dGELU against the stored g', and the bf16 image must be the rounding of the fp32 image bit for bit."""
    from vitssl_hip import _lib as L, ops
    g = torch.Generator().manual_seed(7)
    cus = torch.cuda.get_device_properties(DEV).multi_processor_count
    shapes = [(257, 8, 64), (1000, 24, 128), (513, 72, 320), (4097, 520, 64), (777, 1032, 192), (256 * cus + 300, 264, 64), (33, 2056, 128)]
    for (M, N, K) in shapes:
        A = (torch.randn(M, K, generator=g) * 0.5).to(torch.bfloat16)
        B = (torch.randn(N, K, generator=g) * 0.5).to(torch.bfloat16)
        bias = torch.randn(N, generator=g)
        acc = (A.double() @ B.double().t()).float() + bias
        Ad, Bd, bd = A.to(DEV), B.to(DEV), bias.to(DEV)
        o32 = torch.full((M + 1, N), 7.0, device=DEV)                 # one guard row: nothing may be written past M
        ops.gemm_nt(Ad, Bd, o32[:M], L.EPI_F32, bias=bd)
        assert float((o32[:M].cpu() - acc).abs().max()) < 1e-3 * (1 + float(acc.abs().max())), (M, N, K)
        assert bool((o32[M] == 7.0).all()), (M, N, K)
        o16 = torch.full((M + 1, N), 7.0, dtype=torch.bfloat16, device=DEV)
        cs = torch.zeros(N, device=DEV)
        ops.gemm_nt(Ad, Bd, o16[:M], L.EPI_BF16, bias=bd, colsum=cs)
        assert torch.equal(o16[:M], o32[:M].to(torch.bfloat16)), (M, N, K)
        assert bool((o16[M] == 7.0).all()), (M, N, K)
        assert float((cs.cpu() - acc.sum(0)).abs().max()) < 2e-3 * (1 + float(acc.sum(0).abs().max())), (M, N, K)
        # residual + dropout (fp32 in, fp32 out) and dGELU (bf16 operand in, bf16 out)
        res = torch.randn(M, N, generator=g)
        drop = ops.make_dropout(0.2, seed=3, site=5)
        keep = ops.dropout_mask(M, N, drop, DEV).cpu().float()
        scale = 65536.0 / (65536 - round(0.2 * 65536))
        outr = torch.empty(M, N, device=DEV)
        ops.gemm_nt(Ad, Bd, outr, L.EPI_RESID, bias=bd, aux=res.to(DEV), drop=drop)
        ref = res + acc * keep * scale
        assert float((outr.cpu() - ref).abs().max()) < 1e-3 * (1 + float(ref.abs().max())), (M, N, K)
        gp = (torch.rand(M, N, generator=g) * 1.2).to(torch.bfloat16)
        du = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
        ops.gemm_nt(Ad, Bd, du, L.EPI_DGELU, aux=gp.to(DEV))
        refd = ((A.double() @ B.double().t()).float() * gp.float()).to(torch.bfloat16)
        d = (du.cpu().float() - refd.float()).abs()
        assert bool((d <= refd.float().abs() * 2.0 ** -7 + 1e-3).all()), (M, N, K)


@pytest.mark.parametrize("M,dims", [(5000, [(768, 3072), (3072, 768), (768, 768), (2304, 768)]),      # a ViT-B block's four gradients
                                    (1000, [(384, 1536), (264, 72)]),                                   # ragged tiles, short contraction
                                    (300, [(8, 8)]),                                                    # one tile, one job
                                    (64 * 40, [(2048, 4096), (4096, 2048)])])                           # more tiles than CUs: single owners, no partials
def test_gemm_tn_batch_matches_single_launches(M, dims):
    """vitssl_gemm_bf16_tn_batch: every job's C += A^T B, against an fp64 product and against one vitssl_gemm_bf16_tn per job."""
    from vitssl_hip import ops
    g = torch.Generator().manual_seed(M + len(dims))
    jobs, refs, singles = [], [], []
    for (N1, N2) in dims:
        A = (torch.randn(M, N1, generator=g) * 0.5).to(torch.bfloat16)
        B = (torch.randn(M, N2, generator=g) * 0.5).to(torch.bfloat16)
        C0 = torch.randn(N1, N2, generator=g)
        refs.append(C0.double() + A.double().t() @ B.double())
        Ad, Bd = A.to(DEV), B.to(DEV)
        jobs.append((Ad, Bd, C0.clone().to(DEV)))
        c1 = C0.clone().to(DEV)
        ops.gemm_tn(Ad, Bd, c1)
        singles.append(c1)
    ops.gemm_tn_batch(jobs)
    ops.gemm_tn_batch(jobs[:0])                       # empty list: nothing to do
    for (_, _, Cd), ref, c1 in zip(jobs, refs, singles):
        scale = float(ref.abs().max())
        assert float((Cd.double().cpu() - ref).abs().max()) < 2e-5 * scale
        assert float((Cd - c1).abs().max()) < 1e-5 * scale        # same products, a different split: fp32 summation order only
    # a second call accumulates on top
    ops.gemm_tn_batch(jobs)
    for (A, B, Cd), ref in zip(jobs, refs):
        ref2 = ref + A.double().cpu().t() @ B.double().cpu()
        assert float((Cd.double().cpu() - ref2).abs().max()) < 3e-5 * float(ref2.abs().max())
    with pytest.raises(Exception, match="row counts"):
        ops.gemm_tn_batch([jobs[0], (jobs[0][0][:-8], jobs[0][1][:-8], jobs[0][2])])


def test_gemm_tn_batch_with_helper_workgroups_at_full_size():
    """At M = 50 176 the ViT-B list (108 tiles x 2 splits = 216 units) leaves CUs idle and the planner hands the last rows of every
    tile to the idle workgroups (three partials per tile: csrc/gemm_tn.hip, tn_batch_plan).  Same products as one launch per
    gradient: only the fp32 summation order differs."""
    from vitssl_hip import ops
    M = 50176
    g = torch.Generator(device=DEV).manual_seed(11)
    jobs, singles = [], []
    for (N1, N2) in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]:
        A = (torch.randn(M, N1, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
        B = (torch.randn(M, N2, generator=g, device=DEV) * 0.5).to(torch.bfloat16)
        C0 = torch.randn(N1, N2, generator=g, device=DEV)
        jobs.append((A, B, C0.clone()))
        c1 = C0.clone()
        ops.gemm_tn(A, B, c1)
        singles.append(c1)
    ops.gemm_tn_batch(jobs)
    for (_, _, Cd), c1 in zip(jobs, singles):
        assert float((Cd - c1).abs().max()) < 2e-5 * float(c1.abs().max())
