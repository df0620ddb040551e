"""Host-side logic that needs no GPU: mask stream vs the oracle, index bookkeeping,
state_dict key compatibility with the reference, model_builder dispatch / checkpoint
remapping, schedulers, config helpers."""
import os

import numpy as np
import pytest
import torch

from _util import load_golden, split_prefix
from oracle import vit_oracle as O


def test_draw_mask_is_the_reference_cpu_stream():
    from vit_core.ssl.simmim.masking import draw_mask, mask_indices, simple_masking
    g = load_golden("masking")
    for i in range(4):
        seed, B, N = (int(v) for v in g[f"args{i}"])
        torch.manual_seed(seed)
        m = draw_mask(B, N, float(g[f"ratio{i}"]))
        assert np.array_equal(m.numpy(), g[f"mask{i}"])
    torch.manual_seed(3)
    patches = torch.arange(2 * 9 * 2, dtype=torch.float32).reshape(2, 9, 2)
    p, bm, tg = simple_masking(patches, 0.5)
    assert p is patches and np.array_equal(bm.numpy(), g["order_mask"]) and np.array_equal(tg.numpy(), g["order_targets"])
    idx, inv = mask_indices(bm)
    flat = bm.reshape(-1)
    assert idx.dtype == torch.int32 and torch.equal(idx.long(), flat.nonzero().squeeze(1))
    assert torch.equal(inv[flat].long(), torch.arange(int(flat.sum())))
    assert bool((inv[~flat] == -1).all())
    # edge cases: ratio 0 and ratio 1
    assert int(draw_mask(3, 16, 0.0).sum()) == 0 and int(draw_mask(3, 16, 1.0).sum()) == 48
    idx0, inv0 = mask_indices(draw_mask(2, 4, 0.0))
    assert idx0.numel() == 0 and bool((inv0 == -1).all())


@pytest.mark.parametrize("name", ["simmim_tiny", "vit_tiny", "dino_tiny"])
def test_state_dict_keys_match_reference(name):
    from synth import BIG_KEYS
    g = load_golden(name)
    ref_keys = set(split_prefix(g, "sd/"))
    cfg = [int(v) for v in g["cfg"]]
    if name == "simmim_tiny":
        from vit_core.ssl.simmim import SimMIMViT
        B, img, patch, D, H, F, blocks = cfg
        model = SimMIMViT(blocks, (3, img, img), D, patch, H, F, 0.0, 0.6)
    elif name == "vit_tiny":
        from vit_core import ViT
        B, img, patch, D, H, F, blocks, C = cfg
        model = ViT(C, blocks, (3, img, img), D, patch, H, F, 0.0)
    else:
        from vit_core.ssl.dino import DINOViT
        B, gi, li, patch, D, H, F, blocks, K, G, Lv = cfg
        model = DINOViT(blocks, (3, gi, gi), D, patch, H, F, 0.0, K, 0.9)
        ref_keys |= {f"{h}_head.{k}" for h in ("teacher", "student") for k in BIG_KEYS}
    ours = model.state_dict()
    assert set(ours) == ref_keys
    for k, v in split_prefix(g, "sd/").items():
        assert tuple(ours[k].shape) == tuple(v.shape), k


def test_same_seed_gives_reference_init():
    """Constructing under the same torch seed reproduces the reference's initial weights
    (parameters are created by the same torch modules in the same order)."""
    from vit_core.ssl.simmim import SimMIMViT
    g = load_golden("simmim_tiny")
    B, img, patch, D, H, F, blocks = (int(v) for v in g["cfg"])
    torch.manual_seed(100)                       # make_golden.py: simmim_case(seed=100)
    model = SimMIMViT(blocks, (3, img, img), D, patch, H, F, 0.0, float(g["ratio"]))
    for k, v in split_prefix(g, "sd/").items():
        assert torch.equal(model.state_dict()[k], v), k


class _Cfg(dict):
    __getattr__ = dict.get


def _cfg(mode, **model):
    base = dict(in_channels=3, patch_size=8, embed_dim=128, num_blocks=1, num_heads=2, mlp_dim=192, dropout=0.1,
                num_classes=10, mask_ratio=0.6, output_dim=256, center_momentum=0.9)
    base.update(model)
    return _Cfg(training=_Cfg(type=mode), data=_Cfg(img_size=32), model=_Cfg(base), eval=_Cfg())


def test_build_model_dispatch_and_errors():
    from utils.model_builder import build_model
    from vit_core import ViT
    from vit_core.ssl.simmim import SimMIMViT
    from vit_core.ssl.dino import DINOViT
    assert isinstance(build_model(_cfg("supervised")), ViT)
    assert isinstance(build_model(_cfg("SimMIM")), SimMIMViT)          # case-insensitive like the reference
    assert isinstance(build_model(_cfg("dino")), DINOViT)
    assert isinstance(build_model({"eval": {"mode": "supervised"}, "data": {"img_size": 32}, "model": dict(_cfg("x")["model"])}), ViT)
    with pytest.raises(ValueError):
        build_model(_cfg("nope"))
    with pytest.raises(ValueError):
        build_model(_Cfg(training=_Cfg(), eval=_Cfg(), data=_Cfg(img_size=32), model=_Cfg()))
    with pytest.raises(ValueError):
        build_model(_cfg("supervised", patch_size=5))                   # 32 % 5 != 0 -> ValueError like the reference
    with pytest.raises(AssertionError):
        build_model(_cfg("simmim", num_heads=3))                        # 128 % 3 != 0


def test_load_weights_remaps_simmim_checkpoint(tmp_path):
    from utils.model_builder import build_model, load_weights, freeze_backbone
    sim = build_model(_cfg("simmim"))
    ckpt = {"model_state_dict": {"_orig_mod." + k: v.clone() for k, v in sim.state_dict().items()}}   # torch.compile spelling
    path = os.path.join(tmp_path, "best_model.pth")
    torch.save(ckpt, path)
    vit = build_model(_cfg("supervised"))
    before = vit.classification_head.linear.weight.clone()
    load_weights(vit, path)
    assert torch.equal(vit.encoder_blocks[0].feed_forward.linear_in.weight, sim.encoder_blocks[0].feed_forward.linear_in.weight)
    pe = vit.patch_embedding.positional_embedding
    assert torch.equal(pe[:, 1:], sim.positional_embedding) and bool((pe[:, 0] == 0).all())
    assert torch.equal(vit.classification_head.linear.weight, before)   # SSL-only keys skipped, head untouched
    freeze_backbone(vit)
    trainable = {n for n, p in vit.named_parameters() if p.requires_grad}
    assert trainable == {"patch_embedding.cls_token", "classification_head.norm.weight", "classification_head.norm.bias",
                         "classification_head.linear.weight", "classification_head.linear.bias"}
    with pytest.raises(FileNotFoundError):
        load_weights(vit, os.path.join(tmp_path, "missing.pth"))


def test_schedulers_match_reference_formulas():
    from utils.schedulers import LinearWarmupScheduler
    from vit_core.ssl.dino.dino_utils import DINOMomentumScheduler, DINOTeacherTempScheduler
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([p], lr=1.0)
    w = LinearWarmupScheduler(opt, warmup_steps=4, start_lr=1e-6, target_lr=1e-4)
    for step in range(1, 7):
        w.step()
        want = O.linear_warmup_lr(min(step, 4), 4, 1e-6, 1e-4)
        assert abs(opt.param_groups[0]["lr"] - want) < 1e-15
    g = load_golden("dino_sched")
    ms, ts, tl = DINOMomentumScheduler(0.996, 1.0, 100), DINOTeacherTempScheduler(0.04, 0.07, 30), DINOTeacherTempScheduler(0.04, 0.07, 30, "linear")
    for i, s in enumerate(g["steps"]):
        assert abs(ms.get_momentum(int(s)) - g["mom"][i]) < 1e-12
        assert abs(ts.get_temp(int(s)) - g["temp_cos"][i]) < 1e-12
        assert abs(tl.get_temp(int(s)) - g["temp_lin"][i]) < 1e-12


def test_factories():
    from utils.train_utils import make_criterion, make_schedulers
    cfg = {"training": {"criterion": {"name": "L1Loss", "params": {"reduction": "mean"}}, "warmup_epochs": 2,
                        "warmup_initial_learning_rate": 1e-6, "warmup_final_learning_rate": 1e-4,
                        "lr_scheduler": {"main": {"name": "CosineAnnealingLR", "params": {"eta_min": 1e-6}}, "warmup": {"params": {}}}}}
    assert isinstance(make_criterion(cfg), torch.nn.L1Loss)
    opt = torch.optim.AdamW([torch.nn.Parameter(torch.zeros(2))], lr=1e-4)
    s = make_schedulers(cfg, opt, num_epochs=10, warmup_steps=20)
    assert s["main"].T_max == 8 and s["warmup"].warmup_steps == 20
