"""DINO projection head on a flat store (reference: vit_core/ssl/dino/head.py:7-23):
3 GEMMs with fused bias+GELU epilogues -> row L2-normalise -> weight-normalised last
layer (W = g v/||v|| folded once per optimizer step into bf16 operands)."""
import torch

from ... import _runtime as R
from ..._runtime import BF16, F32, L, ops


class HeadRuntime:
    def __init__(self, store, prefix: str, D: int, K: int, hidden: int = 2048):
        self.store, self.prefix = store, prefix
        self.D, self.K, self.Hd = D, K, hidden
        p = prefix
        store.register_weight(p + "w0", lambda: store.view(p + "mlp.0.weight", (hidden, D)))
        store.register_weight(p + "w2", lambda: store.view(p + "mlp.2.weight", (hidden, hidden)))
        store.register_weight(p + "w4", lambda: store.view(p + "mlp.4.weight", (D, hidden)))
        self.g_name = p + "fully_connected.parametrizations.weight.original0"
        self.v_name = p + "fully_connected.parametrizations.weight.original1"
        self.b_name = p + "fully_connected.bias"
        self.ws = R.Workspace()
        self.rec = {}
        self._fold_key = None
        self.dwn = None

    def prepare(self):
        st = self.store
        key = st.weights_key()
        if key == self._fold_key:
            return
        dev = st.device
        K, D = self.K, self.D
        wf = self.ws.get("wn_f32", (K, D), F32, dev)
        self.inv_vnorm = self.ws.get("inv_vnorm", (K,), F32, dev)
        ops.weightnorm_fold(st.view(self.g_name), st.view(self.v_name, (K, D)), wf, self.inv_vnorm)
        self.wn = self.ws.get("wn", (K, D), BF16, dev)
        self.wnt = None if st.forward_only else self.ws.get("wnt", (D, K), BF16, dev)   # (operand of the backward only)
        ops.cast_transpose_bf16(wf, self.wn, self.wnt)
        self._fold_key = key

    def forward(self, feats, out_logits, save: bool, slot: str):
        """feats fp32 [R, D] -> writes logits fp32 into out_logits [R, K]."""
        st, ws, p = self.store, self.ws, self.prefix
        self.prepare()
        Rn = feats.shape[0]
        dev = feats.device
        tag = slot if save else slot + ".tmp"     # a no-save forward leaves a pending backward's buffers alone
        xb = ws.get(tag + ".xb", (Rn, self.D), BF16, dev)
        ops.cast_bf16(R.as_f32(feats), xb)
        u0 = ws.get(tag + ".u0", (Rn, self.Hd), BF16, dev)
        a0 = ws.get(tag + ".a0", (Rn, self.Hd), BF16, dev)
        ops.gemm_nt(xb, st.w(p + "w0"), u0, L.EPI_GELU, bias=st.view(p + "mlp.0.bias"), out1=a0)
        u1 = ws.get(tag + ".u1", (Rn, self.Hd), BF16, dev)
        a1 = ws.get(tag + ".a1", (Rn, self.Hd), BF16, dev)
        ops.gemm_nt(a0, st.w(p + "w2"), u1, L.EPI_GELU, bias=st.view(p + "mlp.2.bias"), out1=a1)
        z = ws.get(tag + ".z", (Rn, self.D), F32, dev)
        ops.gemm_nt(a1, st.w(p + "w4"), z, L.EPI_F32, bias=st.view(p + "mlp.4.bias"))
        zn = ws.get(tag + ".zn", (Rn, self.D), BF16, dev)
        inv = ws.get(tag + ".inv", (Rn,), F32, dev)
        ops.rownorm_fwd(z, zn, inv)
        ops.gemm_nt(zn, self.wn, out_logits, L.EPI_F32, bias=st.view(self.b_name))
        if save:
            self.rec[slot] = dict(R=Rn, xb=xb, u0=u0, a0=a0, u1=u1, a1=a1, zn=zn, inv=inv)

    def begin_backward(self):
        self.dwn = self.ws.get("dwn", (self.K, self.D), F32, self.store.device)
        self.dwn.zero_()

    def backward(self, dlogits_bf16, slot: str):
        """dlogits bf16 [R, K] -> returns dfeats fp32 [R, D]; accumulates parameter grads
        (the weight-norm pair is finished by finish_backward())."""
        st, ws, p, rec = self.store, self.ws, self.prefix, self.rec[slot]
        gv = st.gview
        Rn, D, Hd = rec["R"], self.D, self.Hd
        dev = dlogits_bf16.device
        ops.colsum_bf16(dlogits_bf16, gv(self.b_name))
        ops.gemm_tn(dlogits_bf16, rec["zn"], self.dwn)
        dzn = ws.get(slot + ".dzn", (Rn, D), F32, dev)
        ops.gemm_nt(dlogits_bf16, self.wnt, dzn, L.EPI_F32)
        dz = ws.get(slot + ".dz", (Rn, D), BF16, dev)
        ops.rownorm_bwd(dzn, rec["zn"], rec["inv"], dz)
        ops.colsum_bf16(dz, gv(p + "mlp.4.bias"))
        ops.gemm_tn(dz, rec["a1"], gv(p + "mlp.4.weight", (D, Hd)))
        du1 = ws.get(slot + ".du1", (Rn, Hd), BF16, dev)
        ops.gemm_nt(dz, st.w(p + "w4.T"), du1, L.EPI_DGELU, aux=rec["u1"], colsum=gv(p + "mlp.2.bias"))
        ops.gemm_tn(du1, rec["a0"], gv(p + "mlp.2.weight", (Hd, Hd)))
        du0 = ws.get(slot + ".du0", (Rn, Hd), BF16, dev)
        ops.gemm_nt(du1, st.w(p + "w2.T"), du0, L.EPI_DGELU, aux=rec["u0"], colsum=gv(p + "mlp.0.bias"))
        ops.gemm_tn(du0, rec["xb"], gv(p + "mlp.0.weight", (Hd, D)))
        dfeats = torch.empty(Rn, D, dtype=F32, device=dev)
        ops.gemm_nt(du0, st.w(p + "w0.T"), dfeats, L.EPI_F32)
        return dfeats

    def finish_backward(self):
        st = self.store
        ops.weightnorm_bwd(self.dwn, st.view(self.g_name), st.view(self.v_name, (self.K, self.D)), self.inv_vnorm,
                           st.gview(self.g_name), st.gview(self.v_name, (self.K, self.D)))
