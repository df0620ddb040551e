"""LayerNorm forward / backward launch time and HBM rate at the ViT-B shape.  Developer probe."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-ssl_amd"))
import torch
from vitssl_hip import ops
dev = torch.device("cuda:0")
M, D = int(os.environ.get("M", 50176)), int(os.environ.get("D", 768))
x = torch.randn(M, D, device=dev); dy = (torch.randn(M, D, device=dev) * 0.5).bfloat16()
gres = torch.randn(M, D, device=dev); gout = torch.empty(M, D, device=dev); gm = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev); dgam = torch.zeros(D, device=dev); dbet = torch.zeros(D, device=dev)
cs = torch.zeros(D, device=dev)
y = torch.empty(M, D, device=dev, dtype=torch.bfloat16); mo = torch.empty(M, device=dev); ro = torch.empty(M, device=dev)
drop = ops.make_dropout(0.1, 1, 2)
def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
ops.layernorm_fwd(x, gamma, beta, y, mo, ro)
us = t(lambda: ops.layernorm_fwd(x, gamma, beta, y, mo, ro)); print(f"ln_fwd            {us:7.1f} us  {M*D*6/us/1e6:.2f} TB/s")
us = t(lambda: ops.layernorm_bwd(dy, x, mo, ro, gamma, gres, gout, gm, dgam, dbet)); print(f"ln_bwd            {us:7.1f} us  {M*D*16/us/1e6:.2f} TB/s")
us = t(lambda: ops.layernorm_bwd(dy, x, mo, ro, gamma, gres, gout, gm, dgam, dbet, cs, drop)); print(f"ln_bwd+cs+drop    {us:7.1f} us  {M*D*16/us/1e6:.2f} TB/s")
