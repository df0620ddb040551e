"""Pin the CPU oracle (oracle/vit_oracle.py) to vectors produced by the reference
itself (tests/golden/make_golden.py).  fp32 tolerance 1e-5 relative;
masks / index order bit-exact."""
import numpy as np
import pytest
import torch

from _util import load_golden, split_prefix, t, rel_l2, max_abs
from oracle import vit_oracle as O
from synth import dino_big_weights, summarize, BIG_KEYS

TOL = 2e-5


def test_masking_bit_exact():
    g = load_golden("masking")
    for i in range(4):
        seed, B, N = (int(v) for v in g[f"args{i}"])
        torch.manual_seed(seed)
        m = O.simple_masking(B, N, float(g[f"ratio{i}"]))
        assert np.array_equal(m.numpy(), g[f"mask{i}"]), i
        assert int(m.sum()) == B * int(N * float(g[f"ratio{i}"]))
    # target rows come out in ascending (b, n) order
    patches = torch.arange(2 * 9 * 2, dtype=torch.float32).reshape(2, 9, 2)
    assert np.array_equal(patches[t(g["order_mask"])].numpy(), g["order_targets"])


def test_sdpa_and_block():
    g = load_golden("ops")
    o, p = O.sdpa(t(g["q"]), t(g["k"]), t(g["v"]))
    assert rel_l2(o, t(g["o"])) < TOL and rel_l2(p, t(g["p"])) < TOL
    sd = split_prefix(g, "blk_sd/")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    x = t(g["blk_x"]).clone().requires_grad_(True)
    y, probs = O.encoder_block(x, leaves, "", 2, return_attn=True)
    assert rel_l2(y, t(g["blk_y"])) < TOL
    assert rel_l2(probs, t(g["blk_probs"])) < TOL
    y.square().sum().backward()
    assert rel_l2(x.grad, t(g["blk_dx"])) < 1e-4
    for k, gr in split_prefix(g, "blk_grad/").items():
        assert rel_l2(leaves[k].grad, gr) < 1e-4, k


def test_flash_delta_mode_is_the_bf16_mode_with_another_backward():
    """`with O.flash_delta():` changes only the BACKWARD of sdpa(emu="bf16") (delta = rowsum(dO * O) from the bf16 output, bf16
    operands -- the form of csrc/attention.hip): same outputs bit for bit, gradients of the reference's golden q / k / v within
    bf16 rounding of the autograd form, and the switch is scoped to the block."""
    g = load_golden("ops")
    bf = lambda a: t(a).to(torch.bfloat16).float()   # noqa: E731
    w = torch.randn(t(g["o"]).shape, generator=torch.Generator().manual_seed(3))
    grads = {}
    for mode in ("autograd", "flash"):
        q, k, v = (bf(g[n]).requires_grad_(True) for n in ("q", "k", "v"))
        if mode == "flash":
            with O.flash_delta():
                o, p = O.sdpa(q, k, v, "bf16")
            assert type(o.grad_fn).__name__ == "_SdpaFlashBwdBackward"
        else:
            o, p = O.sdpa(q, k, v, "bf16")
            assert type(o.grad_fn).__name__ != "_SdpaFlashBwdBackward"      # the switch does not leak out of the block
        (o * w).sum().backward()
        grads[mode] = (o.detach(), p.detach(), q.grad, k.grad, v.grad)
    assert torch.equal(grads["flash"][0], grads["autograd"][0]) and torch.equal(grads["flash"][1], grads["autograd"][1])
    for a, b in zip(grads["flash"][2:], grads["autograd"][2:]):
        assert rel_l2(a, b) < 1e-2
    # fp32 mode never takes it
    with O.flash_delta():
        o, _ = O.sdpa(t(g["q"]), t(g["k"]), t(g["v"]))
    assert rel_l2(o, t(g["o"])) < TOL


@pytest.mark.parametrize("name", ["simmim_tiny", "simmim_n196"])
def test_simmim(name):
    g = load_golden(name)
    B, img, patch, D, H, F, blocks = (int(v) for v in g["cfg"])
    x = t(g["x_u8"]).float() / 256.0
    N = (img // patch) ** 2
    torch.manual_seed(int(g["mask_seed"]))
    mask = O.simple_masking(B, N, float(g["ratio"]))
    assert np.array_equal(mask.numpy(), g["mask"])
    sd = split_prefix(g, "sd/")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    pred, tgt = O.simmim_forward(leaves, x, mask, patch, H)
    assert np.array_equal(tgt.numpy(), g["targets"])          # pure gather: bit-exact
    assert rel_l2(pred, t(g["pred"])) < TOL
    loss = O.l1_loss_mean(pred, tgt)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6 * max(1.0, abs(float(g["loss"])))
    loss.backward()
    for k, gr in split_prefix(g, "grad/").items():
        assert rel_l2(leaves[k].grad, gr) < 2e-4, k
    feat = O.simmim_inference(sd, x, patch, H)
    assert rel_l2(feat, t(g["feat"])) < TOL
    # the bf16-emulating mode stays close to fp32 (it is the GPU tests' tight comparator)
    pred_e, _ = O.simmim_forward(sd, x, mask, patch, H, emu="bf16")
    assert rel_l2(pred_e, t(g["pred"])) < 3e-2


def test_vit_supervised():
    g = load_golden("vit_tiny")
    B, img, patch, D, H, F, blocks, C = (int(v) for v in g["cfg"])
    x = t(g["x_u8"]).float() / 256.0
    sd = split_prefix(g, "sd/")
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    logits, attn = O.vit_forward(leaves, x, patch, H, return_attn=True)
    assert rel_l2(logits, t(g["logits"])) < TOL
    assert rel_l2(attn, t(g["attn"])) < TOL
    loss = O.cross_entropy_mean(logits, t(g["labels"]))
    assert abs(float(loss) - float(g["loss"])) < 1e-5
    loss.backward()
    for k, gr in split_prefix(g, "grad/").items():
        assert rel_l2(leaves[k].grad, gr) < 2e-4, k


def _dino_sd(g, D):
    sd = split_prefix(g, "sd/")
    for k, a in dino_big_weights(D).items():
        sd[k] = t(a)
    return sd


def test_dino():
    g = load_golden("dino_tiny")
    B, gi, li, patch, D, H, F, blocks, K, G, L = (int(v) for v in g["cfg"])
    sd = _dino_sd(g, D)
    views = [t(g[f"view{i}_u8"]).float() / 256.0 for i in range(G + L)]
    leaves = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items()}
    grid = (gi // patch, gi // patch)
    teacher, student, c1 = O.dino_forward(leaves, views, G, patch, H, grid, t(g["center0"]), 0.9)
    assert rel_l2(teacher, t(g["teacher"])) < 5e-5
    assert rel_l2(student, t(g["student"])) < 5e-5
    assert rel_l2(c1, t(g["center1"])) < 5e-5
    lt = teacher.view(G, B, K)
    ls = student.view(G + L, B, K)
    loss_n = O.dino_loss_naive(lt, ls, c1, 0.04, 0.1)
    loss_a = O.dino_loss_algebraic(lt, ls, c1, 0.04, 0.1)
    assert abs(float(loss_n) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    assert abs(float(loss_a) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    loss_a.backward()
    for k, gr in split_prefix(g, "grad/").items():
        assert leaves[k].grad is not None, k
        assert rel_l2(leaves[k].grad, gr) < 5e-4, k
    for k in leaves:
        if k.startswith("student_head.") and any(k.endswith(b) for b in BIG_KEYS):
            s = summarize(leaves[k].grad.numpy())
            assert rel_l2(t(s["rows"]), t(g[f"gradsum/{k}/rows"])) < 5e-4, k
            assert rel_l2(t(s["cols"]), t(g[f"gradsum/{k}/cols"])) < 5e-4, k
            assert abs(s["stats"][1] - g[f"gradsum/{k}/stats"][1]) < 1e-3 * g[f"gradsum/{k}/stats"][1], k
    # teacher gets no gradient
    assert all(leaves[k].grad is None for k in leaves if k.startswith("teacher_") and leaves[k].is_floating_point())
    # EMA
    for k, ref in split_prefix(g, "ema/").items():
        sk = k.replace("teacher_", "student_", 1)
        assert max_abs(O.ema_update(sd[k], sd[sk], 0.996), ref) < 1e-6, k
    # inference features = teacher backbone CLS
    # (golden feats were taken after momentum_update_teacher)
    sd_post = dict(sd)
    sd_post.update(split_prefix(g, "ema/"))
    feats = O.dino_backbone(sd_post, "teacher_backbone.", views[0], patch, H, grid)
    assert rel_l2(feats, t(g["feats"])) < 5e-5


def test_bicubic_matches_torch():
    torch.manual_seed(0)
    img = torch.randn(1, 5, 14, 14)
    for size in [(6, 6), (3, 3), (7, 7), (20, 20), (2, 2)]:
        ref = torch.nn.functional.interpolate(img, size=size, mode="bicubic")
        assert max_abs(O.bicubic_resize(img, *size), ref) < 1e-5, size


def test_dino_schedules():
    g = load_golden("dino_sched")
    for i, s in enumerate(g["steps"]):
        assert abs(O.dino_momentum(int(s), 0.996, 1.0, 100) - g["mom"][i]) < 1e-12
        assert abs(O.dino_teacher_temp(int(s), 0.04, 0.07, 30) - g["temp_cos"][i]) < 1e-12
        assert abs(O.dino_teacher_temp(int(s), 0.04, 0.07, 30, "linear") - g["temp_lin"][i]) < 1e-12


def test_adamw():
    g = load_golden("adamw")
    p = t(g["params"][0])
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for i in range(g["grads"].shape[0]):
        p, m, v = O.adamw_step(p, t(g["grads"][i]), m, v, i + 1, float(g["lr"]), wd=float(g["wd"]))
        assert max_abs(p, t(g["params"][i + 1])) < 1e-6


def _unpack(g, key):
    shape = tuple(int(v) for v in g[key + "_shape"])
    n = int(np.prod(shape))
    return torch.from_numpy(np.unpackbits(g[key])[:n].reshape(shape).astype(np.float32))


def test_encoder_block_dropout_sites_match_reference():
    """The three dropout sites of a block (vit_core/encoder_block.py:45-46,51-52,
    feed_forward.py:27), their placement and their 1/(1-p) scaling: the reference ran with
    dropout 0.1 in train mode and its own keep masks were captured (make_golden_r2.py)."""
    g = load_golden("ops_drop")
    p = float(g["p"])
    keep = [_unpack(g, k) for k in ("keep1", "keep_inner", "keep2")]
    assert 0.85 < float(keep[0].mean()) < 0.95
    leaves = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "sd/").items()}
    x = t(g["x"]).clone().requires_grad_(True)
    y, _ = O.encoder_block(x, leaves, "", 2, keep=keep, p_drop=p)
    assert rel_l2(y, t(g["y"])) < TOL
    y.square().sum().backward()
    assert rel_l2(x.grad, t(g["dx"])) < 1e-4
    for k, gr in split_prefix(g, "grad/").items():
        assert rel_l2(leaves[k].grad, gr) < 1e-4, k
    # and the masks matter: without them the oracle is far from the reference output
    y0, _ = O.encoder_block(t(g["x"]), split_prefix(g, "sd/"), "", 2)
    assert rel_l2(y0, t(g["y"])) > 1e-2


def test_simmim_with_dropout_matches_reference():
    g = load_golden("simmim_drop")
    B, img, patch, D, H, F, blocks = (int(v) for v in g["cfg"])
    x = t(g["x_u8"]).float() / 256.0
    torch.manual_seed(int(g["mask_seed"]))
    mask = O.simple_masking(B, (img // patch) ** 2, float(g["ratio"]))     # masking consumes the RNG first
    assert np.array_equal(mask.numpy(), g["mask"])
    keeps = [[_unpack(g, f"keep{i}_{j}") for j in range(3)] for i in range(blocks)]
    leaves = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "sd/").items()}
    pred, tgt = O.simmim_forward(leaves, x, mask, patch, H, keeps=keeps, p_drop=float(g["p"]))
    assert np.array_equal(tgt.numpy(), g["targets"])
    assert rel_l2(pred, t(g["pred"])) < TOL
    loss = O.l1_loss_mean(pred, tgt)
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-6
    loss.backward()
    for k, gr in split_prefix(g, "grad/").items():
        assert rel_l2(leaves[k].grad, gr) < 2e-4, k


def test_manual_patch_embedding_matches_reference():
    """ManualPatchEmbedding (vit_core/patch_embedding.py:122-128) == unfold . Linear + CLS + pos."""
    g = load_golden("manual_embed")
    B, img, patch, D = (int(v) for v in g["cfg"])
    x = t(g["x_u8"]).float() / 256.0
    leaves = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "sd/").items()}
    y = O.conv_patch_embed(x, leaves["linear.weight"], leaves["linear.bias"], leaves["cls_token"],
                           leaves["positional_embedding"], patch)
    assert rel_l2(y, t(g["y"])) < TOL
    (y * t(g["w"])).sum().backward()
    for k, gr in split_prefix(g, "grad/").items():
        assert rel_l2(leaves[k].grad, gr) < 1e-4, k
