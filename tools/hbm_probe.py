"""What a plain streaming kernel reaches on this MI355X (torch copy / fill / read-reduce), to
judge how far the LayerNorm kernels are from the practical HBM ceiling.  Developer probe."""
import torch
dev = "cuda"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for mb in (154, 616, 2464):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    a.normal_()
    s = t(lambda: b.copy_(a)); print(f"copy  {mb:5d} MB f32: {2*n*4/s/1e12:.2f} TB/s (r+w)")
    s = t(lambda: b.fill_(1.0)); print(f"fill  {mb:5d} MB    : {n*4/s/1e12:.2f} TB/s (w)")
    s = t(lambda: a.sum()); print(f"sum   {mb:5d} MB    : {n*4/s/1e12:.2f} TB/s (r)")
    h = torch.empty(n, device=dev, dtype=torch.bfloat16)
    s = t(lambda: h.copy_(a)); print(f"cast  {mb:5d} MB f32->bf16: {n*6/s/1e12:.2f} TB/s (r+w)")
