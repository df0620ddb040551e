"""Factories that turn the `training:` section of a config into objects.

Same entry points and config keys as the reference (utils/train_utils.py:12-51:
`setup_device`, `make_criterion`, `make_optimizer`, `make_schedulers`; keys `criterion`,
`optimizer`, `lr_scheduler.{main,warmup}`, `warmup_*`), different machinery: every section is
read through `_spec` into a (name, kwargs) pair, and an AdamW request over a model that exposes
its flat parameter store is answered with the fused flat-buffer optimizer kernel instead of
torch.optim.AdamW.  Transform construction (`get_transforms`) is the data side: see
`data.ViewSpec.from_config` for the DINO view recipes.
"""
import logging

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


def get_transforms(config):
    """`transforms:` section -> {name: recipe}.  The reference turns every list into a
    torchvision Compose that runs per image on the CPU (utils/train_utils.py:54-68); here the
    DINO view lists (`globals`, `locals`: crop, flip, jitter, grayscale, blur, ToTensor) become
    `data.ViewSpec` recipes that `data.GPUMultiCrop` renders for a whole batch on the GPU.
    Lists with other transforms are data-loader business outside this package and are rejected
    by `ViewSpec.from_config`."""
    from data import ViewSpec
    return {name: ViewSpec.from_config(sequence) for name, sequence in dict(cfg_get(config, "transforms") or {}).items()}


def setup_device():
    """The engine has no CPU path, so unlike the reference this refuses to hand out 'cpu'."""
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this engine runs the vit_core hot path on MI355X only (no CPU fallback)")
    device = torch.device("cuda")
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
