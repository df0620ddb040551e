"""Do independent kernels on two HIP streams overlap on MI355X?  Pairs from the ViT-B step:
wgrad (TN) beside LayerNorm-backward / attention / another GEMM.  Developer probe."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vit-ssl_amd"))
import torch
from vitssl_hip import ops, _lib as L

dev = torch.device("cuda:0")
M, D, F = 50176, 768, 3072
B, N, H, dh = 256, 196, 12, 64
rb = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
a_tn, b_tn, c_tn = rb(M, F), rb(M, D), torch.zeros(F, D, device=dev)
a2, b2, c2 = rb(M, D), rb(M, D), torch.zeros(D, D, device=dev)
x = torch.randn(M, D, device=dev); dy = rb(M, D); gres = torch.randn(M, D, device=dev); gout = torch.empty(M, D, device=dev)
gm = torch.empty(M, D, device=dev, dtype=torch.bfloat16)
mean = x.mean(1); rstd = 1.0 / x.std(1); gamma = torch.ones(D, device=dev); beta = torch.zeros(D, device=dev)
dgam = torch.zeros(D, device=dev); dbet = torch.zeros(D, device=dev)
y = torch.empty(M, D, device=dev, dtype=torch.bfloat16); mo = torch.empty(M, device=dev); ro = torch.empty(M, device=dev)
qkv = rb(M, 3 * D); o = torch.empty(M, D, device=dev, dtype=torch.bfloat16); lse = torch.empty(B, H, N, device=dev)
do = rb(M, D); dqkv = torch.empty(M, 3 * D, device=dev, dtype=torch.bfloat16); dws = torch.empty(B, H, N, device=dev)
An, Wn, On = rb(M, D), rb(F, D), torch.empty(M, F, device=dev, dtype=torch.bfloat16)
On2 = torch.empty(M, F, device=dev, dtype=torch.bfloat16); bias = torch.randn(F, device=dev)
drop = ops.make_dropout(0.1, 1, 2)
ops._attn_fwd(qkv, o, lse, B, N, H, dh)

K = {
    "tn3072": lambda: ops.gemm_tn(a_tn, b_tn, c_tn),
    "tn768": lambda: ops.gemm_tn(a2, b2, c2),
    "ln_bwd": lambda: ops.layernorm_bwd(dy, x, mean, rstd, gamma, gres, gout, gm, dgam, dbet),
    "ln_fwd": lambda: ops.layernorm_fwd(x, gamma, beta, y, mo, ro),
    "attn_fwd": lambda: ops._attn_fwd(qkv, o, lse, B, N, H, dh),
    "attn_bwd": lambda: ops._attn_bwd(qkv, o, do, lse, dqkv, dws, B, N, H, dh),
    "nt_gelu": lambda: ops.gemm_nt(An, Wn, On, L.EPI_GELU, bias=bias, out1=On2, drop=drop),
    "nt_bf16": lambda: ops.gemm_nt(An, Wn, On, L.EPI_BF16),
}
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def run(fa, fb, par, reps=10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        if par:
            s1.wait_stream(torch.cuda.current_stream()); s2.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s1): fa()
            with torch.cuda.stream(s2): fb()
            torch.cuda.current_stream().wait_stream(s1); torch.cuda.current_stream().wait_stream(s2)
        else:
            fa(); fb()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for f in K.values():
    with torch.cuda.stream(s1): f()
    with torch.cuda.stream(s2): f()
    f()
for a, b in [("tn3072", "ln_bwd"), ("tn3072", "ln_fwd"), ("tn3072", "attn_bwd"), ("tn3072", "attn_fwd"), ("tn3072", "nt_gelu"),
             ("tn768", "ln_bwd"), ("nt_gelu", "ln_bwd"), ("nt_gelu", "attn_fwd"), ("nt_bf16", "ln_fwd"), ("attn_bwd", "ln_bwd")]:
    ser = run(K[a], K[b], False); par = run(K[a], K[b], True)
    print(f"{a:9s} + {b:9s}: serial {ser:7.1f} us   two streams {par:7.1f} us   ({par/ser:.2f}x)", flush=True)
