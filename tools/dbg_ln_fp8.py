import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
from vitssl_hip import ops
from test_gpu_fp8 import _q8_torch, FP8
DEV = torch.device("cuda:0")
torch.manual_seed(6)
rows, cols = 500, 768
x = torch.randn(rows, cols, device=DEV)
dy = (torch.randn(rows, cols, device=DEV) * 1e-3).to(torch.bfloat16)
gres = torch.randn(rows, cols, device=DEV) * 1e-3
gamma = (1 + 0.1 * torch.randn(cols)).to(DEV)
mean, rstd = x.mean(1), 1.0 / torch.sqrt(x.var(1, unbiased=False) + 1e-5)
drop = ops.make_dropout(0.1, 5, 2)
go, gm = torch.empty(rows, cols, device=DEV), torch.empty(rows, cols, dtype=torch.bfloat16, device=DEV)
dg, db, cs = (torch.zeros(cols, device=DEV) for _ in range(3))
gm8 = torch.empty(rows, cols, dtype=FP8, device=DEV)
scale, amax = torch.tensor([2.0 ** 14], device=DEV), torch.zeros(1, device=DEV)
ops.layernorm_bwd_fp8(dy, x, mean, rstd, gamma, gres, go, gm, gm8, scale, amax, dg, db, cs, drop)
keep = ops.dropout_mask(rows, cols, drop, DEV).float()
print("keep mean", float(keep.mean()))
kern_keep = (gm.float() != 0).float()
print("kernel-applied keep mean", float(kern_keep.mean()), "agree with exported", float((kern_keep == keep).float().mean()))
gm32 = go * keep / 0.9
want = _q8_torch((gm32 * 2.0 ** 14).cpu())
neq = (gm8.cpu().view(torch.uint8) != want.view(torch.uint8))
print("mismatch frac", float(neq.float().mean()))
idx = neq.nonzero()[:10]
for r, c in idx.tolist():
    print(r, c, "keep", float(keep[r, c]), "go", float(go[r, c]), "gm", float(gm[r, c]), "gm8", float(gm8[r, c].float()), "want", float(want[r, c].float()), "val*2^14", float(gm32[r, c]) * 2 ** 14)
print("mismatch by keep:", float(neq[keep.cpu() == 1].float().mean()), float(neq[keep.cpu() == 0].float().mean()))
