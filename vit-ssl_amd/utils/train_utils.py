"""Factories that turn the `training:` section of a config into objects.

Same entry points and config keys as the reference (utils/train_utils.py:12-51:
`setup_device`, `make_criterion`, `make_optimizer`, `make_schedulers`; keys `criterion`,
`optimizer`, `lr_scheduler.{main,warmup}`, `warmup_*`), different machinery: every section is
read through `_spec` into a (name, kwargs) pair, and an AdamW request over a model that exposes
its flat parameter store is answered with the fused flat-buffer optimizer kernel instead of
torch.optim.AdamW.  `get_transforms` returns callables with the reference's per-image semantics that also carry the
GPU multi-crop recipe (`.view_spec`, see `data.ViewSpec.from_config`).
"""
import logging
import os

import torch
from torch import nn, optim
from torch.optim import lr_scheduler

from ._config import cfg_get
from .schedulers import LinearWarmupScheduler

logger = logging.getLogger(__name__)
_FUSED_ADAMW_KEYS = {"lr", "betas", "eps", "weight_decay"}


def _spec(config, *path):
    """(name, kwargs) of a `{name: ..., params: {...}}` config node."""
    node = cfg_get(config, *path)
    return cfg_get(node, "name"), dict(cfg_get(node, "params", default={}) or {})


class _NeedsTorchvision:
    """Stand-in for one transform list on a host without torchvision: carries the GPU recipe (`view_spec`), and says what
    is missing when a dataset calls it the way the reference's datasets do (`self.transform(image)`,
    `self.transforms["globals"](image)`: data/datasets.py:36-38,119-123)."""

    def __init__(self, key, sequence, view_spec):
        self.key, self.sequence, self.view_spec = key, list(sequence), view_spec

    def __call__(self, image):
        from vitssl_hip import VitsslError
        names = ", ".join(str(e["name"]) for e in self.sequence)
        raise VitsslError(
            f"transforms[{self.key!r}] ({names}) was called on the CPU, but torchvision is not installed on this host. "
            "The reference builds these lists from torchvision.transforms (utils/train_utils.py:54-68). Either install "
            "torchvision, or feed uint8 [B,H,W,3] batches and let data.GPUMultiCrop render the views on the GPU from "
            "this object's .view_spec (INTEGRATION.md section 4).")

    def __repr__(self):
        return f"_NeedsTorchvision({self.key!r}, view_spec={self.view_spec})"


def _view_spec_or_none(sequence):
    """The GPU multi-crop recipe of a transform list, or None when the list is not a DINO view recipe."""
    from data import ViewSpec
    try:
        return ViewSpec.from_config(sequence)
    except (ValueError, KeyError, TypeError):
        return None


def get_transforms(config):
    """`transforms:` section -> {name: callable}, the reference's contract (utils/train_utils.py:54-68: every list becomes
    a torchvision Compose built with getattr(T, name)(**params), which the datasets call per image on the CPU).  Each
    returned callable ALSO carries `.view_spec`: the same list reduced to the numbers `data.GPUMultiCrop` needs to render
    the DINO views for a whole batch on the GPU (None for lists that are not crop / flip / jitter / gray / blur recipes).
    Without torchvision the callables refuse to run with a message naming the GPU route."""
    try:
        from torchvision import transforms as T
    except ImportError:
        T = None
    out = {}
    for key, sequence in dict(cfg_get(config, "transforms") or {}).items():
        sequence = [dict(name=cfg_get(e, "name"), params=dict(cfg_get(e, "params", default={}) or {})) for e in sequence]
        spec = _view_spec_or_none(sequence)
        if T is None:
            out[key] = _NeedsTorchvision(key, sequence, spec)
            continue
        pipeline = T.Compose([getattr(T, e["name"])(**e["params"]) for e in sequence])
        pipeline.view_spec = spec
        out[key] = pipeline
    return out


def init_data_parallel(device=None, backend=None):
    """Join the data-parallel job `torch.distributed.run` started, once: when WORLD_SIZE > 1 and no process group exists yet,
    size RCCL to the CUs the persistent GEMM grids leave free (`engine.configure_collectives`, must precede the
    communicator) and call `init_process_group` (backend "nccl" = RCCL, bound to `device`; VITSSL_DIST_BACKEND overrides,
    e.g. "gloo" for a rehearsal on CPU or on one GPU).  An already initialised group is left alone.  Returns the world size.
    The reference has no multi-GPU path (train.py:107 only calls setup_device()); this is the "DDP added" of SURVEY 8(b):
    with it `BaseTrainer` finds an initialised group and builds its GradReducer."""
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    if not dist.is_available():
        if world > 1:
            raise RuntimeError(f"WORLD_SIZE={world} but this torch build has no torch.distributed")
        return 1
    if dist.is_initialized():
        return dist.get_world_size()
    if world <= 1:
        return 1
    backend = backend or os.environ.get("VITSSL_DIST_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ or "RANK" not in os.environ:
        raise RuntimeError(f"WORLD_SIZE={world} but RANK / MASTER_PORT are not set: launch with "
                           "`python -m torch.distributed.run --nproc-per-node N train.py ...` (one process per GPU)")
    kwargs = {}
    if backend == "nccl":
        from vitssl_hip.engine import configure_collectives
        configure_collectives()
        if device is not None:
            kwargs["device_id"] = torch.device(device)
    dist.init_process_group(backend, rank=int(os.environ["RANK"]), world_size=world, **kwargs)
    logger.info("data parallel: rank %d of %d over %s", dist.get_rank(), dist.get_world_size(), backend)
    return dist.get_world_size()


def setup_device():
    """One process per GPU: binds this process to cuda:LOCAL_RANK (torch.distributed.run exports it; 0 when run alone),
    joins the data-parallel job when there is one (`init_data_parallel`: nothing else on the reference's entry path,
    train.py:107-128, would) and returns the device.  The reference returns bare "cuda" or "cpu"
    (utils/train_utils.py:12-16); the engine has no CPU path, so unlike the reference this refuses to hand out 'cpu'.
    VITSSL_SHARE_GPU=1 (rehearsal of N ranks on one card, with VITSSL_DIST_BACKEND=gloo) binds every rank to cuda:0."""
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this engine runs the vit_core hot path on MI355X only (no CPU fallback)")
    index = int(os.environ.get("LOCAL_RANK", "0") or 0)
    if os.environ.get("VITSSL_SHARE_GPU", "0") not in ("", "0"):
        index = 0
    if not 0 <= index < torch.cuda.device_count():
        raise RuntimeError(f"LOCAL_RANK={index} but {torch.cuda.device_count()} GPU(s) are visible")
    torch.cuda.set_device(index)                            # bind the GPU before RCCL sees the process
    device = torch.device(f"cuda:{index}")
    init_data_parallel(device)
    logger.info("Using device: %s", device)
    return device


def make_criterion(config):
    name, kwargs = _spec(config, "training", "criterion")
    return getattr(nn, name)(**kwargs)


def make_optimizer(config, model):
    name, kwargs = _spec(config, "training", "optimizer")
    if name == "AdamW" and hasattr(model, "flat_store") and set(kwargs) <= _FUSED_ADAMW_KEYS:
        from vitssl_hip.optim import FusedAdamW
        store = model.trainable_store() if hasattr(model, "trainable_store") else model.flat_store()
        return FusedAdamW(store, **kwargs)
    return getattr(optim, name)((p for p in model.parameters() if p.requires_grad), **kwargs)


def make_schedulers(config, optimizer, num_epochs, warmup_steps):
    """{"main": per-epoch torch scheduler, "warmup": per-step linear ramp}"""
    train = lambda key: cfg_get(config, "training", key)  # noqa: E731
    main_name, main_kwargs = _spec(config, "training", "lr_scheduler", "main")
    _, ramp_kwargs = _spec(config, "training", "lr_scheduler", "warmup")
    main_kwargs["T_max"] = num_epochs - train("warmup_epochs")           # cosine over the post-warm-up epochs
    ramp = LinearWarmupScheduler(optimizer, warmup_steps=warmup_steps, start_lr=train("warmup_initial_learning_rate"),
                                 target_lr=train("warmup_final_learning_rate"), **ramp_kwargs)
    return {"main": getattr(lr_scheduler, main_name)(optimizer, **main_kwargs), "warmup": ramp}
