"""DINOHead (reference: vit_core/ssl/dino/head.py:7-23).  Parameters are created with the
same torch modules (nn.Sequential MLP, parametrizations.weight_norm Linear) so init and
state_dict keys match; the forward runs on the HIP head runtime."""
import torch
from torch import nn
from torch.autograd import Function
from torch.nn.utils.parametrizations import weight_norm

from ... import _runtime as R
from ..._runtime import BF16, F32, ops
from ._head_runtime import HeadRuntime


class _HeadFn(Function):
    @staticmethod
    def forward(ctx, mod, x, need, *params):
        st, rt = mod._store, mod._rt
        st.refresh_weights()
        x2 = R.as_f32(x)
        out = torch.empty(x2.shape[0], rt.K, dtype=F32, device=x.device)
        rt.forward(x2, out, save=need, slot="a")
        ctx.mod = mod
        return out

    @staticmethod
    def backward(ctx, dout):
        mod = ctx.mod
        st, rt = mod._store, mod._rt
        st.gflat.zero_()
        d = R.as_f32(dout)
        db = torch.empty(d.shape, dtype=BF16, device=d.device)
        ops.cast_bf16(d, db)
        rt.begin_backward()
        dx = rt.backward(db, "a")
        rt.finish_backward()
        grads = [st.gview(n, p.shape).clone() if p.requires_grad else None for n, p in zip(st.names, st.params)]
        return (None, dx, None, *grads)


class DINOHead(nn.Module):
    def __init__(self, embed_dim, output_dim, hidden_dim=2048):
        super().__init__()
        self.mlp = nn.Sequential(
            nn.Linear(embed_dim, hidden_dim),
            nn.GELU(),
            nn.Linear(hidden_dim, hidden_dim),
            nn.GELU(),
            nn.Linear(hidden_dim, embed_dim),
        )
        self.fully_connected = weight_norm(nn.Linear(embed_dim, output_dim), name="weight")
        self._dims = (embed_dim, output_dim, hidden_dim)
        self._store = None
        self._rt = None

    def forward(self, x):
        R.require_gpu(x, "DINOHead")
        if self._store is None or not self._store.is_attached() or self._store.device != x.device:
            D, K, Hd = self._dims
            object.__setattr__(self, "_store", R.FlatStore(self, x.device))
            object.__setattr__(self, "_rt", HeadRuntime(self._store, "", D, K, Hd))
        need = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in self._store.params))
        return _HeadFn.apply(self, x, need, *self._store.params)
