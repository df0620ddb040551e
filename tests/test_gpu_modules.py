"""Stand-alone use of the drop-in modules and the trainer loop, on the GPU.  Follows the
reference's own test strategy: shapes/dtypes, tuple returns,
batch independence, ValueError on bad dims, dropout train != eval, input not modified --
plus value/gradient parity against the oracle (bf16 tolerance as in test_gpu_models)."""
import os

import pytest
import torch

from _util import load_golden, split_prefix, t, rel_l2, max_abs
from oracle import vit_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _grads_close(module, leaves, tol=6e-2):
    for k, p in module.named_parameters():
        assert p.grad is not None, k
        assert rel_l2(p.grad, leaves[k].grad) < tol, (k, rel_l2(p.grad, leaves[k].grad))


def test_encoder_block_matches_reference_golden():
    from vit_core import EncoderBlock
    g = load_golden("ops")
    blk = EncoderBlock(d_model=128, num_heads=2, mlp_dim=192, dropout=0.0)
    blk.load_state_dict(split_prefix(g, "blk_sd/"))
    blk = blk.to(DEV).train()
    x = t(g["blk_x"]).to(DEV).requires_grad_(True)
    x0 = x.detach().clone()
    y, probs = blk(x, return_attn=True)
    assert torch.equal(x.detach(), x0)                                        # input not modified
    assert y.shape == x.shape and probs.shape == (3, 2, 10, 10)
    assert rel_l2(y, t(g["blk_y"])) < 2e-2 and rel_l2(probs, t(g["blk_probs"])) < 2e-2
    y.square().sum().backward()
    assert rel_l2(x.grad, t(g["blk_dx"])) < 5e-2
    ref = split_prefix(g, "blk_grad/")
    for k, p in blk.named_parameters():
        assert rel_l2(p.grad, ref[k]) < 6e-2, (k, rel_l2(p.grad, ref[k]))
    y2, none = blk(x0)
    assert none is None and isinstance(y2, torch.Tensor)


def test_encoder_block_dropout_and_batch_independence():
    from vit_core import EncoderBlock
    torch.manual_seed(0)
    blk = EncoderBlock(d_model=128, num_heads=2, mlp_dim=256, dropout=0.5).to(DEV)
    x = torch.randn(4, 17, 128, device=DEV)
    blk.eval()
    with torch.no_grad():
        e1, _ = blk(x)
        e2, _ = blk(x)
        singles = torch.cat([blk(x[i:i + 1])[0] for i in range(4)])
    assert torch.equal(e1, e2)
    assert max_abs(e1, singles) < 1e-5                                        # batch independence
    blk.train()
    with torch.no_grad():
        t1, _ = blk(x)
        t2, _ = blk(x)
    assert not torch.equal(t1, e1) and not torch.equal(t1, t2)                # dropout active and re-drawn


def test_multi_headed_attention_and_sdpa():
    from vit_core import MultiHeadedAttention, ScaledDotProductAttention
    torch.manual_seed(1)
    mha = MultiHeadedAttention(128, 2).to(DEV)
    sd = {k: v.detach().cpu() for k, v in mha.state_dict().items()}
    x = torch.randn(3, 20, 128)
    xd = x.to(DEV).requires_grad_(True)
    out, probs = mha(xd, xd, xd, return_attn=True)
    assert out.shape == (3, 20, 128) and probs.shape == (3, 2, 20, 20)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ro, rp = O.mha(xr, leaves, "", 2, emu="bf16", return_attn=True)
    assert rel_l2(out, ro) < 1e-2 and rel_l2(probs, rp) < 1e-2
    out.square().sum().backward()
    ro.square().sum().backward()
    assert rel_l2(xd.grad, xr.grad) < 5e-2
    _grads_close(mha, leaves)
    # distinct (but equal-valued) query/key/value tensors take the three-GEMM path
    q, k, v = (x.to(DEV).clone().requires_grad_(True) for _ in range(3))
    out2, none = mha(q, k, v)
    assert none is None and rel_l2(out2, out) < 1e-3
    out2.square().sum().backward()
    assert rel_l2(q.grad + k.grad + v.grad, xr.grad) < 5e-2
    with pytest.raises(AssertionError):
        MultiHeadedAttention(100, 3)
    # functional SDPA: 3-D [batch, seq, d_k] as in the reference docstring, and 4-D
    qq, kk, vv = (torch.randn(2, 20, 64, device=DEV) for _ in range(3))
    ctx, p = ScaledDotProductAttention(qq, kk, vv, return_attn=True)
    rc, rp2 = O.sdpa(qq.cpu(), kk.cpu(), vv.cpu())
    assert rel_l2(ctx, rc) < 2e-2 and rel_l2(p, rp2) < 2e-2
    ctx2, none2 = ScaledDotProductAttention(qq, kk, vv)
    assert none2 is None and ctx2.shape == qq.shape


def test_feed_forward_block():
    from vit_core import FeedForwardBlock
    torch.manual_seed(2)
    ffn = FeedForwardBlock(d_model=128, d_ff=256, dropout=0.0).to(DEV)
    sd = {k: v.detach().cpu() for k, v in ffn.state_dict().items()}
    x = torch.randn(5, 9, 128)
    xd = x.to(DEV).requires_grad_(True)
    y = ffn(xd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ry = O.feed_forward(xr, leaves, "", emu="bf16")
    assert y.shape == x.shape and rel_l2(y, ry) < 1e-2
    y.square().sum().backward()
    ry.square().sum().backward()
    assert rel_l2(xd.grad, xr.grad) < 5e-2
    _grads_close(ffn, leaves)
    ffn.eval()
    with torch.no_grad():
        full = ffn(x.to(DEV))
        singles = torch.cat([ffn(x[i:i + 1].to(DEV)) for i in range(5)])
    assert max_abs(full, singles) < 1e-5
    drop = FeedForwardBlock(128, 256, dropout=0.5).to(DEV).train()
    with torch.no_grad():
        assert not torch.equal(drop(x.to(DEV)), drop(x.to(DEV)))


def test_mlp_head():
    from vit_core.mlp_head import MLPHead
    torch.manual_seed(3)
    head = MLPHead(128, 10).to(DEV)
    sd = {k: v.detach().cpu() for k, v in head.state_dict().items()}
    x = torch.randn(7, 128)
    xd = x.to(DEV).requires_grad_(True)
    y = head(xd)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    ry = O.linear(O.rnd(O.layer_norm(xr, leaves["norm.weight"], leaves["norm.bias"]), "bf16"), leaves["linear.weight"], leaves["linear.bias"], "bf16")
    assert y.shape == (7, 10) and rel_l2(y, ry) < 1e-2
    y.square().sum().backward()
    ry.square().sum().backward()
    assert rel_l2(xd.grad, xr.grad) < 5e-2
    _grads_close(head, leaves)


def test_patch_embeddings():
    from vit_core import ConvolutionalPatchEmbedding, ManualPatchEmbedding, DynamicPatchEmbedding
    torch.manual_seed(4)
    x = torch.rand(3, 3, 32, 32)
    conv = ConvolutionalPatchEmbedding((3, 32, 32), 128, 8).to(DEV)
    sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in conv.state_dict().items()}
    y = conv(x.to(DEV))
    ry = O.conv_patch_embed(x, sd["conv.weight"], sd["conv.bias"], sd["cls_token"], sd["positional_embedding"], 8, emu="bf16")
    assert y.shape == (3, 17, 128) and rel_l2(y, ry) < 1e-2
    y.square().sum().backward()
    ry.square().sum().backward()
    for k, p in conv.named_parameters():
        assert rel_l2(p.grad, sd[k].grad) < 5e-2, k
    man = ManualPatchEmbedding((3, 32, 32), 128, 8).to(DEV)
    msd = {k: v.detach().cpu() for k, v in man.state_dict().items()}
    ym = man(x.to(DEV))
    rm = torch.cat([msd["cls_token"].expand(3, -1, -1), O.linear(O.patchify(x, 8), msd["linear.weight"], msd["linear.bias"], "bf16")], 1) + msd["positional_embedding"]
    assert rel_l2(ym, rm) < 1e-2
    with torch.no_grad():                                                       # batch independence (reference tests)
        singles = torch.cat([conv(x[i:i + 1].to(DEV)) for i in range(3)])
    assert max_abs(conv(x.to(DEV)).detach(), singles) < 1e-5
    for cls in (ConvolutionalPatchEmbedding, ManualPatchEmbedding):
        with pytest.raises(ValueError):
            cls((3, 30, 32), 128, 8)
    dyn = DynamicPatchEmbedding((3, 32, 32), 128, 8).to(DEV)
    dsd = {"p." + k: v.detach().cpu().clone().requires_grad_(True) for k, v in dyn.state_dict().items()}
    for size in (32, 16, 48):
        xi = torch.rand(2, 3, size, size)
        yd = dyn(xi.to(DEV))
        rd = O.dynamic_patch_embed(xi, dsd, "p.", 8, (4, 4), emu="bf16")
        assert yd.shape == rd.shape and rel_l2(yd, rd) < 1e-2, size
    yd.square().sum().backward()                                                # size 48: gradient flows through the bicubic resize
    rd.square().sum().backward()
    assert rel_l2(dyn.positional_embedding.grad, dsd["p.positional_embedding"].grad) < 5e-2
    with pytest.raises(ValueError):
        dyn(torch.rand(1, 3, 30, 32, device=DEV))


def _train_cfg(tmp_path, mode="simmim"):
    return {"training": {"type": mode, "num_epochs": 2, "warmup_epochs": 1, "warmup_initial_learning_rate": 1e-6,
                         "warmup_final_learning_rate": 1e-3, "criterion": {"name": "L1Loss", "params": {"reduction": "mean"}},
                         "optimizer": {"name": "AdamW", "params": {"lr": 1e-3, "weight_decay": 1e-3}},
                         "lr_scheduler": {"main": {"name": "CosineAnnealingLR", "params": {"eta_min": 1e-6}}, "warmup": {"params": {}}}},
            "eval": {}, "data": {"img_size": 32},
            "model": {"in_channels": 3, "patch_size": 8, "embed_dim": 128, "num_blocks": 2, "num_heads": 2, "mlp_dim": 192,
                      "dropout": 0.1, "mask_ratio": 0.6, "num_classes": 10, "output_dim": 256, "center_momentum": 0.9}}


def test_simmim_trainer_fused_and_reference_style(tmp_path):
    from utils.model_builder import build_model
    from utils.trainers import SimMIMTrainer
    from vitssl_hip.optim import FusedAdamW
    torch.manual_seed(5)
    cfg = _train_cfg(tmp_path)
    data = [torch.rand(8, 3, 32, 32) for _ in range(6)]
    model = build_model(cfg).to(DEV)
    tr = SimMIMTrainer(model, str(tmp_path / "fused"), cfg, data, data[:2], DEV)
    assert isinstance(tr.optimizer, FusedAdamW) and tr._fused_ok()
    def val(trainer):
        torch.manual_seed(1234)                                                 # same masks for every evaluation
        return trainer.validate()["Loss"]
    first = val(tr)
    tr.fit(2)
    assert val(tr) < first                                                      # it learns
    ck = torch.load(tmp_path / "fused" / "last_model.pth", weights_only=False)
    assert set(ck) >= {"epoch", "model_state_dict", "optimizer_state_dict", "config"} and ck["epoch"] == 2
    assert set(ck["model_state_dict"]) == set(model.state_dict())
    assert os.path.exists(tmp_path / "fused" / "best_model.pth")
    # reference-style path: stock torch optimizer + autograd through the same kernels
    cfg2 = _train_cfg(tmp_path)
    cfg2["training"]["optimizer"] = {"name": "SGD", "params": {"lr": 0.05, "momentum": 0.9}}
    cfg2["training"]["warmup_final_learning_rate"] = 0.05
    cfg2["training"]["num_epochs"] = 4
    model2 = build_model(cfg2).to(DEV)
    tr2 = SimMIMTrainer(model2, str(tmp_path / "sgd"), cfg2, data, data[:2], DEV)
    assert not tr2._fused_ok()
    before = val(tr2)
    tr2.fit(4)
    assert val(tr2) < before
    # optimizer state round trip in torch.optim.AdamW layout
    osd = tr.optimizer.state_dict()
    assert set(osd) == {"state", "param_groups"} and set(osd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"}
    tr.optimizer.load_state_dict(osd)


def test_supervised_and_dino_trainers(tmp_path):
    from utils.model_builder import build_model
    from utils.trainers import SupervisedTrainer, DINOTrainer
    torch.manual_seed(6)
    cfg = _train_cfg(tmp_path, "supervised")
    cfg["training"]["criterion"] = {"name": "CrossEntropyLoss", "params": {}}
    data = [(torch.rand(8, 3, 32, 32), torch.randint(0, 10, (8,))) for _ in range(4)]
    tr = SupervisedTrainer(build_model(cfg).to(DEV), str(tmp_path / "sup"), cfg, data, data[:1], DEV)
    before = tr.validate()["Loss"]
    tr.fit(2)
    assert tr.validate()["Loss"] < before
    cfgd = _train_cfg(tmp_path, "dino")
    views = [[torch.rand(4, 3, 32, 32), torch.rand(4, 3, 32, 32), torch.rand(4, 3, 16, 16), torch.rand(4, 3, 16, 16)] for _ in range(3)]
    trd = DINOTrainer(build_model(cfgd).to(DEV), str(tmp_path / "dino"), cfgd, views, views[:1], DEV)
    assert trd._is_fused()
    trd.fit(1)
    m = trd.train_epoch(2)
    assert m["Loss"] == m["Loss"] and 0.04 <= m["TeacherTemp"] <= 0.07 and 0.996 <= m["Momentum"] <= 1.0


def test_dino_trainer_gpu_multicrop(tmp_path):
    """Loader yields raw uint8 [B,H,W,3] batches; the trainer builds the views on the GPU from
    the config's transforms.globals / transforms.locals recipes."""
    from utils.model_builder import build_model
    from utils.trainers import DINOTrainer
    torch.manual_seed(8)
    cfgd = _train_cfg(tmp_path, "dino")
    cfgd["training"].update(num_all_views=4, num_global_views=2)
    recipe = lambda size, scale: [                                              # noqa: E731  (configs/dino/{globals,locals}.yaml)
        {"name": "RandomResizedCrop", "params": {"size": size, "scale": scale}}, {"name": "RandomHorizontalFlip", "params": {}},
        {"name": "ColorJitter", "params": {"brightness": 0.4, "contrast": 0.4, "saturation": 0.2, "hue": 0.1}},
        {"name": "GaussianBlur", "params": {"kernel_size": 7, "sigma": [0.1, 2.0]}}, {"name": "ToTensor"}]
    cfgd["transforms"] = {"globals": recipe(32, [0.5, 1.0]) + [], "locals": recipe(16, [0.08, 0.4])}
    raw = [torch.randint(0, 256, (4, 40, 40, 3), dtype=torch.uint8) for _ in range(3)]
    trd = DINOTrainer(build_model(cfgd).to(DEV), str(tmp_path / "dino_mc"), cfgd, raw, raw[:1], DEV)
    views = trd._views(raw[0])
    assert [tuple(v.shape) for v in views] == [(4, 3, 32, 32)] * 2 + [(4, 3, 16, 16)] * 2 and views[0].is_cuda
    m = trd.train_epoch(1)
    assert m["Loss"] == m["Loss"]
    v = trd.validate()["Loss"]
    assert v == v                                                               # finite (fresh random views every call)
