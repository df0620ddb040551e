"""Reflective factories (reference: utils/train_utils.py:12-51).  AdamW over a model that
exposes a flat store maps to the fused flat-buffer kernel; anything else is stock torch."""
import logging

import torch
from torch import nn, optim
from torch.optim import lr_scheduler

from ._config import cfg_get
from .schedulers import LinearWarmupScheduler

logger = logging.getLogger(__name__)


def setup_device():
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: this engine runs the vit_core hot path on MI355X only (no CPU fallback)")
    device = torch.device("cuda")
    logger.info(f"Using device: {device}")
    return device


def make_criterion(config):
    crit = cfg_get(config, "training", "criterion")
    cls = getattr(nn, cfg_get(crit, "name"))
    return cls(**dict(cfg_get(crit, "params", default={}) or {}))


def make_optimizer(config, model):
    opt = cfg_get(config, "training", "optimizer")
    name = cfg_get(opt, "name")
    params = dict(cfg_get(opt, "params", default={}) or {})
    flat = getattr(model, "flat_store", None)
    if name == "AdamW" and flat is not None and set(params) <= {"lr", "betas", "eps", "weight_decay"}:
        from vitssl_hip.optim import FusedAdamW
        store = model.trainable_store() if hasattr(model, "trainable_store") else flat()
        return FusedAdamW(store, **params)
    trainable = [p for p in model.parameters() if p.requires_grad]
    return getattr(optim, name)(trainable, **params)


def make_schedulers(config, optimizer, num_epochs, warmup_steps):
    sched = cfg_get(config, "training", "lr_scheduler")
    main, warm = cfg_get(sched, "main"), cfg_get(sched, "warmup")
    main_cls = getattr(lr_scheduler, cfg_get(main, "name"))
    main_kwargs = dict(cfg_get(main, "params", default={}) or {}, T_max=num_epochs - cfg_get(config, "training", "warmup_epochs"))
    warm_kwargs = dict(cfg_get(warm, "params", default={}) or {}, warmup_steps=warmup_steps,
                       start_lr=cfg_get(config, "training", "warmup_initial_learning_rate"),
                       target_lr=cfg_get(config, "training", "warmup_final_learning_rate"))
    return {"main": main_cls(optimizer, **main_kwargs), "warmup": LinearWarmupScheduler(optimizer, **warm_kwargs)}
