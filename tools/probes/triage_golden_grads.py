"""Per-parameter gradient error of the golden SimMIM models (HIP vs the reference's fp32 gradients stored in the fixture)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from _util import rel_l2, t, split_prefix
import test_gpu_models as TM
for name in ("simmim_tiny", "simmim_n196"):
    g, model, sd, cfg = TM._simmim_from_golden(name)
    x = (t(g["x_u8"]).float() / 256.0).to(TM.DEV)
    model.train()
    torch.manual_seed(int(g["mask_seed"]))
    pred, tgt, mask = model(x, return_bool_mask=True)
    torch.nn.L1Loss()(pred, tgt).backward()
    ref = split_prefix(g, "grad/")
    print(name, cfg, "pred", round(rel_l2(pred, t(g["pred"])), 5))
    rows = sorted(((rel_l2(p.grad, ref[k]), k) for k, p in model.named_parameters()), reverse=True)
    for r, k in rows[:8]:
        print(f"   {k:55s} {r:.4f}")
    qk = [r for r, k in rows if "w_query" in k or "w_key" in k]
    print("   w_query / w_key: max", round(max(qk), 4), "mean", round(sum(qk) / len(qk), 4), " all params: median", round(rows[len(rows) // 2][0], 4))
