#!/usr/bin/env python3
"""Headline benchmark: ViT-B/16 SimMIM pre-training step (224x224, mask 0.6, bf16 MFMA
GEMMs, batch 256 per GPU, dropout 0.1, AdamW) on N MI355X GPUs, synthetic data.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A step = zero_grad -> mask draw -> forward -> L1 -> backward -> (RCCL gradient
all-reduce, overlapped) -> AdamW, i.e. SimMIMViT.train_step.  Rank 0 prints ONE JSON
line.  `value` is whole-job images/s; `roofline` is the dominant kernel family (the
bf16 MFMA GEMMs) timed live with HIP events on the launch stream; `cpu_baseline` is the
CPU oracle timed on this host's cores on a bounded sample (rank 0, N=1 only).  On one GPU the
default run then times the other BASELINE.json configurations (ViT-S/16 batch 256, ViT-L/16
batch 128 in bf16 and with the fp8 weight path, ViT-B/16 DINO batch 64; 5 + 10 steps each) and
appends them as `other_configs` -- after the headline's timed region, never inside it."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "vit-ssl_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402

MODELS = {  # standard ViT families
    "vit_tiny": dict(D=192, L=12, H=3, F=768),
    "vit_s": dict(D=384, L=12, H=6, F=1536),
    "vit_b": dict(D=768, L=12, H=12, F=3072),
    "vit_l": dict(D=1024, L=24, H=16, F=4096),
}
PEAK_BF16_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md
PEAK_FP8_TFLOPS = 5000.0   # dense fp8 MFMA peak (K = 128 f8f6f4 form), MI355X_MICROARCH.md
PEAK_HBM_TBS = 8.0         # HBM3E spec (6.29 TB/s measured copy ceiling), MI355X_MICROARCH.md


def cpu_quota():
    """CPU cores this process may actually use: cgroup v2 quota (cpu.max), else the affinity mask."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return max(1, int(float(q) / float(per) + 0.5)), f"cgroup cpu.max {q}/{per}"
    except (OSError, ValueError):
        pass
    n = len(os.sched_getaffinity(0))
    return n, "sched_getaffinity"


def train_flops_per_image(D, L, H, F, N, Pd, nm):
    """Algorithmic-FLOP convention of SURVEY.md section 8(d): dense contractions only,
    train = 3 x (blocks + head) + 2 x patch projection."""
    blocks = L * (3 * 2 * N * D * D + 2 * N * N * D + 2 * N * N * D + 2 * N * D * D + 4 * N * D * F)
    head = 2 * nm * D * Pd
    proj = 2 * N * Pd * D
    return 3 * (blocks + head) + 2 * proj


def kernel_sources_sha16():
    """Fingerprint of the kernel sources this build was made from (csrc/*.hip, *.cpp, *.h and the C header): the .git
    directory does not travel to the GPU box, the sources do.  tools/profile_round.sh stamps it into the PMC summary."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "vit-ssl_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "vit-ssl_amd", "csrc", "*.cpp"))
                   + glob.glob(os.path.join(ROOT, "vit-ssl_amd", "csrc", "*.h")) + glob.glob(os.path.join(ROOT, "include", "*.h")))
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic_per_launch(family="gemm_nt", profiles_dir=None):
    """HBM bytes per launch of the dominant kernel family from the committed rocprofv3 PMC
    passes (profiles/*_pmc_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate
    --pmc runs of this very command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
    for gfx950's wide coalesced reads; unit KB).  Only a summary stamped with the fingerprint of
    the kernel sources in this tree is used: a summary of other kernels must not label these.
    Returns (bytes or None, source-or-reason)."""
    import glob
    sha = kernel_sources_sha16()
    files = sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "*_pmc_traffic.json")))
    stale = None
    for f in reversed(files):
        try:
            d = json.load(open(f))
        except (OSError, ValueError):
            continue
        if d.get("kernel_sources_sha16") != sha:
            stale = stale or os.path.basename(f)
            continue
        try:
            ft, w = d["FETCH_SIZE"][family], d["WRITE_SIZE"][family]
            return (2.0 * ft["sum_kb"] / ft["dispatches"] + w["sum_kb"] / w["dispatches"]) * 1024.0, os.path.basename(f)
        except KeyError:
            continue
    if stale:
        return None, f"no PMC summary for kernel sources {sha} (newest committed: {stale}, other sources): run tools/profile_round.sh"
    return None, None


def pmc_clock_ghz(family="gemm_nt", profiles_dir=None):
    """The clock the chip held over the launches of a kernel family, from the same committed PMC summary (a third --pmc pass,
    SQ_BUSY_CYCLES: one instance per shader engine, 32 on the chip; clock = sum of busy cycles / 32 / sum of the launches'
    durations in that pass).  MFMA-dense loops on random data do not run at the 2.4 GHz the peak is quoted at
    (MI355X_MICROARCH.md, DVFS give-back).  Same stamping rule as pmc_traffic_per_launch; None when the
    summary has no such pass."""
    import glob
    sha = kernel_sources_sha16()
    for f in reversed(sorted(glob.glob(os.path.join(profiles_dir or os.path.join(ROOT, "profiles"), "*_pmc_traffic.json")))):
        try:
            d = json.load(open(f))
            if d.get("kernel_sources_sha16") != sha:
                continue
            c = d["SQ_BUSY_CYCLES"][family]
            if c["duration_ns"] > 0:
                return c["sum"] / 32.0 / c["duration_ns"]
        except (OSError, ValueError, KeyError, TypeError):
            continue
    return None


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cfg, img, P, ratio, seconds_budget=25.0):
    """Reference-equivalent fp32 CPU step (oracle/vit_oracle.py: zero_grad -> forward ->
    L1 -> backward -> AdamW, eager, dropout 0.1) on a bounded sample."""
    from oracle import vit_oracle as O
    from vit_core.ssl.simmim import SimMIMViT
    torch.manual_seed(42)
    cores, quota_src = cpu_quota()
    cores = min(cores, os.cpu_count() or cores)
    torch.set_num_threads(cores)                           # a pool larger than the quota only spins and gets throttled
    Bc = 8
    model = SimMIMViT(num_blocks=cfg["L"], input_shape=(3, img, img), embed_dim=cfg["D"], patch_size=P,
                      num_heads=cfg["H"], mlp_dim=cfg["F"], dropout=0.1, mask_ratio=ratio)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    del model
    x = torch.rand(Bc, 3, img, img)
    N = (img // P) ** 2
    opt_state = {}
    times = []
    t_start = time.perf_counter()
    step = 0
    while True:
        mask = O.simple_masking(Bc, N, ratio)
        t0 = time.perf_counter()
        O.simmim_train_step(sd, opt_state, x, mask, P, cfg["H"], step + 1, 1e-4, 1e-3, p_drop=0.1)
        dt = time.perf_counter() - t0
        step += 1
        if step > 1:          # first step = warm-up
            times.append(dt)
        if step >= 2 and (time.perf_counter() - t_start > seconds_budget or len(times) >= 3):
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(Bc / med, 3), "unit": "images/s", "cores": cores, "cores_source": quota_src,
            "host_logical_cpus": os.cpu_count(), "cpu": _cpu_model(), "kind": "port",
            "sample": f"same model/inputs shape, batch {Bc}, {len(times)} measured step(s) after 1 warm-up, fp32 eager, dropout 0.1, AdamW"}


def families_from(recs):
    """(fam, hbm, kernels) of one instrumented step: per kernel family algorithmic flops / live-event ms / launches; the
    HBM-bound kernels by algorithmic bytes; per-label detail of the MFMA families."""
    fam, hbm, kernels = {}, {}, {}
    for label, flops, e0, e1, nbytes in recs:
        ms = e0.elapsed_time(e1)
        k = label.split("[")[0].split(" ")[0]
        if nbytes:                                      # HBM-bound kernels: algorithmic bytes / time vs the 8 TB/s spec
            h = hbm.setdefault(k, [0.0, 0.0, 0])
            h[0] += nbytes
            h[1] += ms
            h[2] += 1
            continue
        for d, key in ((fam, k), (kernels, label)):
            a = d.setdefault(key, [0.0, 0.0, 0])
            a[0] += flops
            a[1] += ms
            a[2] += 1
    return fam, hbm, kernels


def peak_of(k):
    return PEAK_FP8_TFLOPS if k in ("gemm_fp8_nt", "gemm_fp8_tn") else PEAK_BF16_TFLOPS


KERNEL_NAMES = {"gemm_nt": "gemm_nt_pp_kernel", "gemm_tn": "gemm_tn_pp_kernel", "gemm_fp8_nt": "gemm_nt_pp_kernel<.., fp8>",
                "gemm_fp8_tn": "gemm_tn_fp8_kernel"}


def instrumented(step_fn):
    """One extra step with every GEMM / attention / LayerNorm / AdamW launch bracketed by HIP events on the launch stream."""
    from vitssl_hip import ops
    ops.PROFILE = []
    step_fn()
    torch.cuda.synchronize()
    recs, ops.PROFILE = ops.PROFILE, None
    return families_from(recs)


def short_roofline(fam):
    """Dominant MFMA family of a side configuration: achieved TFLOP/s against the dense peak of its operand type."""
    if not fam:
        return None
    name, (fl, ms, cnt) = max(fam.items(), key=lambda kv: kv[1][1])
    ach = fl / (ms * 1e-3) / 1e12
    return {"kernel": KERNEL_NAMES.get(name, name + "_kernel"), "bound": "mfma", "achieved": round(ach, 1), "peak": peak_of(name),
            "unit": "TFLOP/s", "frac": round(ach / peak_of(name), 4), "launches_per_step": cnt, "ms_per_step": round(ms, 3)}


def side_simmim(model_name, batch, dtype, dev, steps, warmup, img=224, patch=16, ratio=0.6, dropout=0.1):
    """One of the other BASELINE.json SimMIM configurations, same step as the headline (train_step, dropout 0.1, AdamW)."""
    from vit_core.ssl.simmim import SimMIMViT
    from vitssl_hip import engine
    from vitssl_hip.optim import FusedAdamW
    cfg = MODELS[model_name]
    engine.set_linear_operands(dtype)
    torch.manual_seed(42)
    model = SimMIMViT(num_blocks=cfg["L"], input_shape=(3, img, img), embed_dim=cfg["D"], patch_size=patch, num_heads=cfg["H"],
                      mlp_dim=cfg["F"], dropout=dropout, mask_ratio=ratio).to(dev).train()
    opt = FusedAdamW(model.flat_store(), lr=1e-4, weight_decay=1e-3)
    x = torch.rand(batch, 3, img, img, generator=torch.Generator().manual_seed(42)).to(dev)
    for _ in range(warmup):
        loss = model.train_step(x, opt, None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = model.train_step(x, opt, None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    final = float(loss)
    recomputed = float((model.last_pred - model.last_targets).abs().mean())
    if not (final == final and abs(recomputed - final) <= 1e-4 * max(1.0, abs(final))):
        raise RuntimeError(f"bench sanity ({model_name} {dtype}): fused loss {final} vs recomputed {recomputed}")
    fam, _, _ = instrumented(lambda: model.train_step(x, opt, None))
    N = (img // patch) ** 2
    fl = train_flops_per_image(cfg["D"], cfg["L"], cfg["H"], cfg["F"], N, 3 * patch * patch, int(N * ratio)) * batch
    name = model_name.replace("_", "-").upper().replace("VIT-", "ViT-")
    return {"workload": f"{name}/{patch} SimMIM {img}x{img} mask {ratio} dropout {dropout} AdamW, batch {batch}/GPU",
            "dtype": "bf16" if dtype == "bf16" else "fp8 e4m3 Linear operands", "steps": steps, "warmup": warmup,
            "ms_per_step": round(dt * 1e3, 3), "images_per_sec": round(batch / dt, 1), "unit": "images/s",
            "mfma_util": round(fl / dt / 1e12 / PEAK_BF16_TFLOPS, 4), "final_loss": round(final, 5), "roofline": short_roofline(fam)}


def side_dino(batch, dev, steps, warmup, dropout=0.1):
    """BASELINE configs[3]: ViT-B/16 DINO, 2 x 224^2 global + 8 x 96^2 local crops, K = 65 536, EMA 0.996, tau_s 0.1, tau_t 0.04,
    bf16 GEMMs, AdamW; views resident in HBM.  437.8 GF per image set."""
    from vit_core.ssl.dino import DINOViT
    from vit_core.ssl.dino.loss import DINOLoss
    from vitssl_hip import engine
    from vitssl_hip.optim import FusedAdamW
    engine.set_linear_operands("bf16")
    torch.manual_seed(42)
    m = DINOViT(12, (3, 224, 224), 768, 16, 12, 3072, dropout, 65536, 0.9).to(dev).train()
    opt = FusedAdamW(m.trainable_store(), lr=1e-4, weight_decay=1e-3)
    crit = DINOLoss(0.04, 0.1)
    views = [torch.rand(batch, 3, 224, 224, device=dev) for _ in range(2)] + [torch.rand(batch, 3, 96, 96, device=dev) for _ in range(8)]
    step = lambda: m.train_step(views, 2, crit, opt, None, 0.996)  # noqa: E731
    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    final = float(loss)
    if final != final:
        raise RuntimeError("bench sanity (DINO): loss is NaN")
    fam, _, _ = instrumented(step)
    return {"workload": f"ViT-B/16 DINO student+teacher, 2x224 + 8x96 crops, K=65536, EMA 0.996, dropout {dropout} AdamW, batch {batch} image sets/GPU",
            "dtype": "bf16", "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 3),
            "image_sets_per_sec": round(batch / dt, 1), "images_per_sec": round(10 * batch / dt, 1), "unit": "views/s (10 per image set)",
            "mfma_util": round(437.8e9 * batch / dt / 1e12 / PEAK_BF16_TFLOPS, 4), "final_loss": round(final, 5),
            "roofline": short_roofline(fam)}


def other_configs(dev, steps=10, warmup=5):
    """The BASELINE.json configurations other than the headline one, on this GPU, after the headline's timed region:
    configs[1] ViT-S, configs[4] ViT-L in bf16 and with the fp8 weight path, configs[3] DINO (per-GPU batch 64)."""
    import gc
    rows = []
    jobs = [lambda: side_simmim("vit_s", 256, "bf16", dev, steps, warmup),
            lambda: side_simmim("vit_l", 128, "bf16", dev, steps, warmup),
            lambda: side_simmim("vit_l", 128, "fp8", dev, steps, warmup),
            lambda: side_dino(64, dev, steps, warmup)]
    for job in jobs:
        try:
            rows.append(job())
        except Exception as e:                               # a side configuration must never cost the headline line
            rows.append({"error": f"{type(e).__name__}: {e}"[:300]})
        gc.collect()
        torch.cuda.empty_cache()
    return rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--model", default="vit_b", choices=sorted(MODELS))
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--img", type=int, default=224)
    ap.add_argument("--patch", type=int, default=16)
    ap.add_argument("--mask-ratio", type=float, default=0.6)
    ap.add_argument("--dropout", type=float, default=0.1)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp8"],
                    help="operand type of the Linear GEMMs of the encoder blocks (fp8 = OCP e4m3, BASELINE configs[4]; attention and "
                         "everything else stay bf16).  The headline metric is bf16.")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the side runs of the other BASELINE configurations (ViT-S, ViT-L bf16 / fp8, DINO) that follow the "
                         "headline's timed region on one GPU")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse the N>1 path on one GPU)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal: all ranks use cuda:0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with torch.distributed.run (one rank per GPU)")
    import torch.distributed as dist
    dev = torch.device("cuda:0" if args.share_gpu else f"cuda:{local_rank}")
    torch.cuda.set_device(dev)                               # bind the GPU before RCCL sees the process
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        from vitssl_hip.engine import configure_collectives
        configure_collectives()                              # RCCL channels <= the CUs the persistent GEMM grids leave alone
        kw = {"device_id": dev} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)

    from vit_core.ssl.simmim import SimMIMViT
    from vitssl_hip import ops
    from vitssl_hip import engine
    from vitssl_hip import _lib as _vl
    from vitssl_hip.engine import GradReducer
    from vitssl_hip.optim import FusedAdamW
    engine.set_linear_operands(args.dtype)

    cfg = MODELS[args.model]
    # The CPU baseline runs FIRST (rank 0, N = 1): the GPU phase then ends the process, where the
    # driver's utilisation sampler can see it.
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(cfg, args.img, args.patch, args.mask_ratio)
    # Host hygiene for the GPU phase: a GPU job gets a small CPU share; torch's default
    # (one OpenMP worker per core, spin-waiting after every parallel CPU op) exhausts it and
    # the whole process is throttled for tens of ms.  The CPU-baseline leg restores all cores.
    from vit_core._runtime import limit_host_threads
    host_threads = limit_host_threads()
    torch.manual_seed(42)                                   # identical init on every rank
    model = SimMIMViT(num_blocks=cfg["L"], input_shape=(3, args.img, args.img), embed_dim=cfg["D"], patch_size=args.patch,
                      num_heads=cfg["H"], mlp_dim=cfg["F"], dropout=args.dropout, mask_ratio=args.mask_ratio).to(dev).train()
    store = model.flat_store()
    reducer = None
    dp = None
    if world > 1:
        dist.broadcast(store.flat, 0)
        store.mark_dirty()
        reducer = GradReducer(store.gflat)
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)                               # what the collective library itself sees
        dp = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "allreduce_of_ones": float(ones),
              "grad_bytes": 4 * store.gflat.numel(), "bucket_mb": reducer.bucket_elems * 4 / 2 ** 20,
              "reserved_cus": int(_vl.lib().vitssl_get_reserved_cus()),      # the library's value in force, not the env
              "nccl_env": {k: os.environ.get(k) for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO")}}
        if float(ones) != world:
            raise SystemExit(f"all_reduce over {world} ranks returned {float(ones)}: the process group is not what torchrun launched")
    opt = FusedAdamW(store, lr=1e-4, weight_decay=1e-3)
    gen = torch.Generator().manual_seed(42 + rank)
    x = torch.rand(args.batch, 3, args.img, args.img, generator=gen).to(dev)
    torch.manual_seed(1000 + rank)                          # masks / dropout seeds are rank-local

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = model.train_step(x, opt, reducer)
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = model.train_step(x, opt, reducer)
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    final_loss = float(loss)
    # the fused loss of the last timed step equals mean|pred - target| recomputed by torch on its outputs
    recomputed = float((model.last_pred - model.last_targets).abs().mean())
    if not (final_loss == final_loss and abs(recomputed - final_loss) <= 1e-4 * max(1.0, abs(final_loss))):
        raise SystemExit(f"bench sanity: fused loss {final_loss} vs recomputed {recomputed}")
    if world > 1:
        # exposed communication = step with the overlapped all-reduce - the same step without any reduction
        # (diagnostic only, after the timed region; the un-reduced steps de-synchronise nothing: same data per rank)
        nb, nbytes = reducer.stats()
        k = max(3, min(10, args.steps))
        sync()
        t1 = time.perf_counter()
        for _ in range(k):
            model.train_step(x, opt, reducer)
        sync()
        with_comm = (time.perf_counter() - t1) / k
        t1 = time.perf_counter()
        for _ in range(k):
            model.train_step(x, opt, None)
        sync()
        without = (time.perf_counter() - t1) / k
        dist.broadcast(store.flat, 0)                       # replicas drifted during the un-reduced steps
        store.mark_dirty()
        # one more step with every bucket bracketed by events on the communication stream: (bytes, ms after the step began,
        # ms the all-reduce took) -- buckets that start late or run long are queueing behind the persistent GEMM grids
        reducer.timing = True
        model.train_step(x, opt, reducer)
        dp["buckets"] = reducer.bucket_times()
        reducer.timing = False
        sync()
        dp.update(buckets_per_step=nb, reduced_bytes_per_step=nbytes, ms_step_with_allreduce=round(with_comm * 1e3, 3),
                  ms_step_without_allreduce=round(without * 1e3, 3), exposed_comm_ms=round((with_comm - without) * 1e3, 3))
        print(f"[bench rank {rank}] {json.dumps(dp)}", file=sys.stderr, flush=True)

    N = (args.img // args.patch) ** 2
    Pd = 3 * args.patch * args.patch
    nm = int(N * args.mask_ratio)
    fl_img = train_flops_per_image(cfg["D"], cfg["L"], cfg["H"], cfg["F"], N, Pd, nm)
    ms_per_step = elapsed / args.steps * 1e3
    imgs_per_s = args.batch * world * args.steps / elapsed
    step_tflops = fl_img * args.batch / (ms_per_step * 1e-3) / 1e12      # per GPU

    # ---- live kernel timing of one instrumented step (HIP events on the launch stream)
    roofline = None
    kernels = {}
    if not args.no_kernel_timing:
        fam, hbm, kernels = instrumented(lambda: model.train_step(x, opt, reducer))
        dom = max(fam.items(), key=lambda kv: kv[1][1])
        name, (fl, ms, cnt) = dom
        ach = fl / (ms * 1e-3) / 1e12
        traffic, traffic_src = (pmc_traffic_per_launch(name) if args.model == "vit_b" and args.batch == 256 and args.dtype == "bf16"
                                else (None, None))
        clock = pmc_clock_ghz(name) if traffic is not None else None
        roofline = {"kernel": KERNEL_NAMES.get(name, name + "_kernel"),
                    "bound": "mfma", "achieved": round(ach, 1), "peak": peak_of(name),
                    "unit": "TFLOP/s", "frac": round(ach / peak_of(name), 4),
                    "traffic": None if traffic is None else round(traffic), "traffic_unit": "bytes/launch (HBM, PMC)",
                    "traffic_source": traffic_src,
                    # the clock held over this family's launches (PMC, same summary) and the fraction of the peak AT that clock:
                    # context for `frac`, which stays against the 2.4 GHz peak
                    "clock_ghz": None if clock is None else round(clock, 3),
                    "frac_at_clock": None if clock is None else round(ach / (peak_of(name) * clock / 2.4), 4),
                    "launches_per_step": cnt, "avg_launch_ms": round(ms / cnt, 4),
                    "alg_flops_per_launch": fl / cnt,
                    "families": {k: {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1), "ms_per_step": round(v[1], 3), "launches": v[2],
                                     "peak": peak_of(k)} for k, v in fam.items()},
                    "hbm": {k: {"achieved_tbs": round(v[0] / (v[1] * 1e-3) / 1e12, 2), "peak_tbs": PEAK_HBM_TBS,
                                "frac": round(v[0] / (v[1] * 1e-3) / 1e12 / PEAK_HBM_TBS, 3), "ms_per_step": round(v[1], 3),
                                "launches": v[2], "alg_bytes_per_launch": round(v[0] / v[2])} for k, v in hbm.items()}}

    # ---- the other BASELINE configurations, after (never inside) the headline's timed region; one GPU only
    others = None
    headline = (args.model, args.batch, args.dtype, args.img, args.patch) == ("vit_b", 256, "bf16", 224, 16)
    if world == 1 and headline and not args.no_other_configs:
        import gc
        last = (model.last_pred, model.last_targets)
        del model, store, opt, x, last
        gc.collect()
        torch.cuda.empty_cache()
        others = other_configs(dev)

    if rank == 0:
        out = {
            "metric": "images_per_sec", "value": round(imgs_per_s, 2), "unit": "images/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.dtype == "bf16" else "fp8 e4m3 operands in the Linear GEMMs of the encoder blocks (forward, dgrad, wgrad), bf16 elsewhere",
            "data": "synthetic",
            "config": {"workload": f"{args.model.replace('_', '-').upper().replace('VIT-', 'ViT-')}/{args.patch} SimMIM {args.img}x{args.img} "
                                   f"mask {args.mask_ratio} dropout {args.dropout} AdamW, batch {args.batch}/GPU"
                                   + (", fp8 weight path (forward, input and weight gradients)" if args.dtype == "fp8" else ""),
                       "global_batch": args.batch * world, "parallelism": f"dp{world}"},
            "per_gpu_images_per_sec": round(imgs_per_s / world, 2),
            "mfma_util": round(step_tflops / PEAK_BF16_TFLOPS, 4),
            "step_tflops_per_gpu": round(step_tflops, 1),
            "train_gflop_per_image": round(fl_img / 1e9, 2),
            "final_loss": round(final_loss, 5),
            "roofline": roofline, "cpu_baseline": cpu,
        }
        if dp is not None:
            out["data_parallel"] = dp
        if others is not None:
            out["other_configs"] = others
        if os.environ.get("BENCH_KERNELS"):
            out["kernels"] = {k: {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 1), "ms": round(v[1], 3), "n": v[2]} for k, v in kernels.items()}
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
