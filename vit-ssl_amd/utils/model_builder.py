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


def strip_compile_prefix(state_dict):
    return {(k[len("_orig_mod."):] if k.startswith("_orig_mod.") else k): v for k, v in state_dict.items()}


def load_weights(model, checkpoint_path: str):
    """Load a (pre-training) checkpoint into `model`, remapping SimMIM keys to the
    fine-tuning ViT layout (projection.* -> patch_embedding.*, positional embedding
    gains a zero CLS slot) and skipping SSL-only tensors."""
    if not os.path.exists(checkpoint_path):
        raise FileNotFoundError(f"Checkpoint file not found: {checkpoint_path}")
    logger.info(f"Loading weights from: {checkpoint_path}")
    ckpt = torch.load(checkpoint_path, map_location="cpu", weights_only=False)
    src = strip_compile_prefix(ckpt.get("model_state_dict", ckpt))
    dst = model.state_dict()
    out = {}
    for k, v in src.items():
        if k in dst:
            if v.shape == dst[k].shape:
                out[k] = v
            else:
                logger.warning(f"Shape mismatch for '{k}': Pretrained {v.shape} vs Model {dst[k].shape}")
        elif k.startswith("projection.") and f"patch_embedding.{k}" in dst:
            nk = f"patch_embedding.{k}"
            if v.shape == dst[nk].shape:
                out[nk] = v
                logger.info(f"Remapped key '{k}' to '{nk}'")
            else:
                logger.warning(f"Shape mismatch for remapped key '{nk}' (from '{k}')")
        elif k == "positional_embedding" and "patch_embedding.positional_embedding" in dst:
            tgt = dst["patch_embedding.positional_embedding"]
            if v.shape[1] == tgt.shape[1] - 1 and v.shape[2] == tgt.shape[2]:
                pe = torch.zeros_like(tgt)
                pe[:, 1:, :] = v
                out["patch_embedding.positional_embedding"] = pe
            else:
                logger.warning(f"Cannot interpolate positional_embedding: Pretrained {v.shape} vs Model {tgt.shape}")
        elif "simmim_head" in k or "mask_token" in k or k.startswith("teacher.") or k.startswith("center"):
            logger.info(f"Skipping SSL-specific key: {k}")
        else:
            logger.warning(f"Key '{k}' from checkpoint not found in the model.")
    missing, unexpected = model.load_state_dict(out, strict=False)
    logger.info("Successfully loaded weights.")
    logger.warning(f"Missing keys in model: {missing}")
    logger.warning(f"Unexpected keys in model (from checkpoint but not used): {unexpected}")
    return model


def freeze_backbone(model: ViT):
    """Freeze everything but the classifier head and the CLS token."""
    logger.info("Freezing model backbone...")
    for p in model.encoder_blocks.parameters():
        p.requires_grad = False
    for name, p in model.patch_embedding.named_parameters():
        if "cls_token" not in name:
            p.requires_grad = False
    logger.info("Backbone frozen.")


def build_model(config):
    mode = cfg_get(config, "training", "type") or cfg_get(config, "eval", "mode")
    if mode is None:
        raise ValueError("Could not determine mode. Set either 'training.type' or 'eval.mode' in config.")
    mode = mode.lower()
    m = lambda k: cfg_get(config, "model", k)  # noqa: E731
    image_shape = (m("in_channels"), cfg_get(config, "data", "img_size"), cfg_get(config, "data", "img_size"))
    logger.info(f"Building model for mode: '{mode}'")
    common = dict(input_shape=image_shape, patch_size=m("patch_size"), embed_dim=m("embed_dim"), num_blocks=m("num_blocks"),
                  num_heads=m("num_heads"), mlp_dim=m("mlp_dim"), dropout=m("dropout"))
    if mode in ("supervised", "finetune"):
        model = ViT(num_classes=m("num_classes"), **common)
    elif mode == "simmim":
        model = SimMIMViT(mask_ratio=m("mask_ratio"), **common)
    elif mode in ("dino", "eval_dino"):
        model = DINOViT(output_dim=m("output_dim"), center_momentum=m("center_momentum"), **common)
    else:
        raise ValueError(f"Unknown model-building mode: {mode}")
    if mode == "finetune":
        model = load_weights(model, cfg_get(config, "training", "pretrained_path"))
        if cfg_get(config, "training", "freeze_backbone"):
            freeze_backbone(model)
    elif mode == "eval_dino":
        model = load_weights(model, os.path.join(cfg_get(config, "eval", "experiment_path"), "best_model.pth"))
    return model
