"""Model factory (reference: utils/model_builder.py:11-184).  Same entry points and mode
dispatch; returns the bare module (the reference wraps it in torch.compile, whose
checkpoints carry an ``_orig_mod.`` key prefix -- accepted here on load)."""
import logging
import os

import torch

from vit_core.vit import ViT
from vit_core.ssl.dino.model import DINOViT
from vit_core.ssl.simmim.model import SimMIMViT

from ._config import cfg_get

logger = logging.getLogger(__name__)


_COMPILE_PREFIX = "_orig_mod."                 # key prefix of checkpoints saved from a torch.compile-wrapped model
_SSL_ONLY = ("simmim_head", "mask_token")      # tensors that exist only while pre-training


def strip_compile_prefix(state_dict):
    return {k.removeprefix(_COMPILE_PREFIX): v for k, v in state_dict.items()}


def _place(key, tensor, target):
    """Where a checkpoint tensor goes in a model with state `target`.
    Returns (destination key, tensor) or (None, reason)."""
    def fits(dst_key, t):
        if t.shape == target[dst_key].shape:
            return dst_key, t
        return None, f"shape {tuple(t.shape)} != model's {tuple(target[dst_key].shape)} for {dst_key}"

    if key in target:
        return fits(key, tensor)
    embedded = f"patch_embedding.{key}"
    if key.startswith("projection.") and embedded in target:           # SimMIM patch projection -> ViT embedder
        return fits(embedded, tensor)
    if key == "positional_embedding" and embedded in target:           # SimMIM table has no CLS row: add a zero one
        slot = target[embedded]
        if tensor.shape[1] + 1 == slot.shape[1] and tensor.shape[2] == slot.shape[2]:
            grown = torch.zeros_like(slot)
            grown[:, 1:] = tensor
            return embedded, grown
        return None, f"positional table {tuple(tensor.shape)} cannot seed {tuple(slot.shape)}"
    if any(tag in key for tag in _SSL_ONLY) or key.startswith(("teacher.", "center")):
        return None, "pre-training-only tensor"
    return None, "no such parameter in the model"


def load_weights(model, checkpoint_path: str):
    """Initialise `model` from a checkpoint written by the reference or by this package
    (reference: utils/model_builder.py:11-89).  Accepts bare state dicts and trainer
    checkpoints, with or without the torch.compile key prefix, and carries SimMIM
    pre-training weights over to the fine-tuning ViT layout (see `_place`)."""
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError(f"Checkpoint file not found: {checkpoint_path}")
    payload = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    source = strip_compile_prefix(payload.get("model_state_dict", payload))
    target = model.state_dict()
    accepted, skipped = {}, {}
    for key, tensor in source.items():
        dst_key, result = _place(key, tensor, target)
        if dst_key is None:
            skipped[key] = result
        else:
            accepted[dst_key] = result
    missing, unexpected = model.load_state_dict(accepted, strict=False)
    logger.info("loaded %d tensors from %s (%d skipped)", len(accepted), checkpoint_path, len(skipped))
    for key, why in skipped.items():
        logger.info("  skipped %s: %s", key, why)
    if missing:
        logger.warning("parameters left at their initial values: %s", list(missing))
    if unexpected:
        logger.warning("tensors the model did not take: %s", list(unexpected))
    return model


def freeze_backbone(model: ViT):
    """Linear-probe setting of the reference (utils/model_builder.py:92-101): only the
    classification head and the CLS token keep training."""
    frozen = list(model.encoder_blocks.parameters())
    frozen += [p for name, p in model.patch_embedding.named_parameters() if "cls_token" not in name]
    for p in frozen:
        p.requires_grad = False
    logger.info("backbone frozen: %d tensors", len(frozen))


def _resolve_mode(config) -> str:
    mode = cfg_get(config, "training", "type") or cfg_get(config, "eval", "mode")
    if mode is None:
        raise ValueError("Could not determine mode. Set either 'training.type' or 'eval.mode' in config.")
    return mode.lower()


def build_model(config):
    """config -> module, by mode (reference: utils/model_builder.py:104-184)."""
    mode = _resolve_mode(config)
    m = lambda key: cfg_get(config, "model", key)  # noqa: E731
    side = cfg_get(config, "data", "img_size")
    backbone = dict(input_shape=(m("in_channels"), side, side), patch_size=m("patch_size"), embed_dim=m("embed_dim"),
                    num_blocks=m("num_blocks"), num_heads=m("num_heads"), mlp_dim=m("mlp_dim"), dropout=m("dropout"))
    makers = {
        "supervised": lambda: ViT(num_classes=m("num_classes"), **backbone),
        "simmim": lambda: SimMIMViT(mask_ratio=m("mask_ratio"), **backbone),
        "dino": lambda: DINOViT(output_dim=m("output_dim"), center_momentum=m("center_momentum"), **backbone),
    }
    makers["finetune"], makers["eval_dino"] = makers["supervised"], makers["dino"]
    if mode not in makers:
        raise ValueError(f"Unknown model-building mode: {mode}")
    logger.info("building a %s model", mode)
    model = makers[mode]()
    if mode == "finetune":
        load_weights(model, cfg_get(config, "training", "pretrained_path"))
        if cfg_get(config, "training", "freeze_backbone"):
            freeze_backbone(model)
    elif mode == "eval_dino":
        load_weights(model, os.path.join(cfg_get(config, "eval", "experiment_path"), "best_model.pth"))
    return model
