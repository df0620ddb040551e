"""Throughput of the GPU multi-crop pipeline next to the Pillow-equivalent CPU
oracle: B source images 96x96 (STL10, configs/dino/data.yaml) -> 2 global + 8 local views.
Reports image sets/s, per-kernel time (HIP events) and achieved HBM GB/s against the
algorithmic byte count.  Developer tool; prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vit-ssl_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from data.multicrop import GPUMultiCrop, ViewSpec, pack_params, sample_view_params  # noqa: E402
from vitssl_hip import ops  # noqa: E402


def main():
    B = int(os.environ.get("B", 256))
    H = W = int(os.environ.get("SRC", 96))
    GS, LS = int(os.environ.get("GS", 224)), int(os.environ.get("LS", 96))
    dev = torch.device("cuda:0")
    from vit_core._runtime import limit_host_threads
    limit_host_threads()
    rng = np.random.default_rng(0)
    imgs = torch.from_numpy(rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)).to(dev)
    gs, ls = ViewSpec(size=GS, scale=(0.5, 1.0), gray_p=0.2), ViewSpec(size=LS, scale=(0.08, 0.4))
    mc = GPUMultiCrop(gs, ls, 10, 2)
    gen = torch.Generator().manual_seed(0)
    for _ in range(2):
        mc(imgs, gen)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        views = mc(imgs, gen)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / reps

    # kernel-only time per view size (params already on the device)
    per = {}
    for name, spec in (("global", gs), ("local", ls)):
        S = spec.size
        prm = [sample_view_params(spec, H, W, gen) for _ in range(B)]
        ip, fp = pack_params(prm, 7)
        ip_d, fp_d = torch.from_numpy(ip).to(dev), torch.from_numpy(fp).to(dev)
        tmp = torch.empty(B, H, S, 3, dtype=torch.uint8, device=dev)
        u8 = torch.empty(B, S, S, 3, dtype=torch.uint8, device=dev)
        out = torch.empty(B, 3, S, S, device=dev)
        stages = {"resized_crop": lambda: ops.aug_resized_crop_u8(imgs, ip_d, tmp, u8),
                  "color": lambda: ops.aug_color_u8(u8, ip_d, fp_d),
                  "blur_to_tensor": lambda: ops.aug_blur_to_tensor(u8, fp_d, out, 7)}
        # algorithmic HBM bytes: crop read <= B*H*W*3, tmp w+r, u8 write; colour r+w; blur read + f32 write
        alg = {"resized_crop": B * (H * W * 3 + 2 * H * S * 3 + S * S * 3), "color": 2 * B * S * S * 3,
               "blur_to_tensor": B * S * S * 3 * (1 + 4)}
        for k, fn in stages.items():
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            per[f"{name}.{k}"] = {"us": round(ms * 1e3, 1), "GB/s": round(alg[k] / ms / 1e6, 1)}
    gpu_kernel_ms = 2 * sum(v["us"] for k, v in per.items() if k.startswith("global")) / 1e3 + \
        8 * sum(v["us"] for k, v in per.items() if k.startswith("local")) / 1e3

    # CPU baseline: the Pillow-pinned oracle on a bounded sample (one image set)
    from oracle import augment_oracle as A
    img0 = imgs[0].cpu().numpy()
    t0 = time.perf_counter()
    nset = 0
    while time.perf_counter() - t0 < 10.0:
        for v in range(10):
            spec = gs if v < 2 else ls
            A.apply_view(img0, sample_view_params(spec, H, W, gen), spec.size)
        nset += 1
    cpu_sets = nset / (time.perf_counter() - t0)
    print(json.dumps({"workload": f"{B} images {H}x{W} -> 2x{GS}^2 + 8x{LS}^2 views", "image_sets_per_s": round(B / wall, 1),
                      "ms_per_batch_wall": round(wall * 1e3, 2), "ms_per_batch_kernels": round(gpu_kernel_ms, 2), "kernels": per,
                      "cpu_oracle_sets_per_s_1core": round(cpu_sets, 2)}))


if __name__ == "__main__":
    main()
