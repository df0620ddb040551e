"""Flat fused AdamW: one HIP kernel over the model's flat parameter buffer
(torch.optim.AdamW semantics; reference builds AdamW reflectively at
utils/train_utils.py:25-29 with configs/base/training.yaml:10-15)."""
import torch

from . import ops
from .engine import FlatStore


class FusedAdamW(torch.optim.Optimizer):
    """Drop-in for torch.optim.AdamW over a FlatStore.

    * `step_flat(gscale)` -- one kernel over the whole flat buffer; used by the fused
      train step (gradients already live in the flat gradient buffer; `gscale` folds the
      1/world_size of data-parallel averaging).
    * `step()` -- generic torch-style step: gradients that autograd produced as separate
      tensors are first copied into the flat buffer; parameters whose .grad is None are
      skipped, as torch.optim.AdamW does.
    param_groups[0]['lr'] is honoured, so the reference's warm-up / cosine schedulers
    (utils/schedulers.py, torch.optim.lr_scheduler) drive it unchanged."""

    def __init__(self, store: FlatStore, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        params = [p for p in store.params if p.requires_grad]
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.store = store
        self.exp_avg = torch.zeros_like(store.flat)
        self.exp_avg_sq = torch.zeros_like(store.flat)
        self.step_count = 0
        self._all_trainable = len(params) == len(store.params)

    def _hyper(self):
        g = self.param_groups[0]
        return float(g["lr"]), g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"]

    @torch.no_grad()
    def step_flat(self, gscale: float = 1.0):
        if not self._all_trainable:
            # frozen parameters in the store: per-parameter launches over the trainable slices.
            # The fused step leaves its gradients in store.gflat only (p.grad stays None).
            return self.step(gscale=gscale, from_flat=True)
        lr, b1, b2, eps, wd = self._hyper()
        self.step_count += 1
        st = self.store
        ops.adamw(st.flat, st.gflat, self.exp_avg, self.exp_avg_sq, lr, b1, b2, eps, wd, self.step_count, gscale)
        st.mark_dirty()

    @torch.no_grad()
    def step(self, closure=None, gscale: float = 1.0, from_flat: bool = False):
        """torch-style step.  `from_flat=True`: the gradients already live in the store's flat
        gradient buffer (fused train_step), so p.grad is not consulted."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lr, b1, b2, eps, wd = self._hyper()
        self.step_count += 1
        st = self.store
        trainable = {id(p) for p in self.param_groups[0]["params"]}
        for name, p in zip(st.names, st.params):
            if id(p) not in trainable or (p.grad is None and not from_flat):
                continue
            o, n = st.offsets[name]
            gslice = st.gflat[o:o + n]
            if not from_flat and p.grad.data_ptr() != gslice.data_ptr():
                gslice.copy_(p.grad.reshape(-1))
            ops.adamw(st.flat[o:o + n], gslice, self.exp_avg[o:o + n], self.exp_avg_sq[o:o + n], lr, b1, b2, eps, wd,
                      self.step_count, gscale)
        st.mark_dirty()
        return loss

    # checkpoint compatibility: torch.optim.AdamW layout (per-parameter state)
    def state_dict(self):
        st = self.store
        state = {}
        trainable = [p for p in self.param_groups[0]["params"]]
        index = {id(p): i for i, p in enumerate(trainable)}
        for name, p in zip(st.names, st.params):
            if id(p) not in index:
                continue
            o, n = st.offsets[name]
            state[index[id(p)]] = {
                "step": torch.tensor(float(self.step_count)),
                "exp_avg": self.exp_avg[o:o + n].view(p.shape).clone(),
                "exp_avg_sq": self.exp_avg_sq[o:o + n].view(p.shape).clone(),
            }
        g = {k: v for k, v in self.param_groups[0].items() if k != "params"}
        g["params"] = list(range(len(trainable)))
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        st = self.store
        trainable = [p for p in self.param_groups[0]["params"]]
        index = {id(p): i for i, p in enumerate(trainable)}
        for k, v in sd["param_groups"][0].items():
            if k != "params":
                self.param_groups[0][k] = v
        for name, p in zip(st.names, st.params):
            i = index.get(id(p))
            if i is None or i not in sd["state"]:
                continue
            o, n = st.offsets[name]
            s = sd["state"][i]
            self.exp_avg[o:o + n].copy_(s["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + n].copy_(s["exp_avg_sq"].reshape(-1))
            self.step_count = int(float(s["step"]))
