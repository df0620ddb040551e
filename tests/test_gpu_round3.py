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
        L.call("vitssl_set_reserved_cus", 0)
        ops.gemm_nt(A, B, out1, L.EPI_BF16)
        assert lib.vitssl_debug_last_nt_grid() == cus
        with pytest.raises(L.VitsslError, match="negative"):
            L.call("vitssl_set_reserved_cus", -3)
    finally:
        L.call("vitssl_set_reserved_cus", before)
