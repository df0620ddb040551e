#!/usr/bin/env python3
"""Generate golden input/output vectors from the reference itself.

Run ONLY in the build container (needs /root/reference):

    python tests/golden/make_golden.py

Imports ``/root/reference/vit_core`` (pure torch; importable on CPU, SURVEY.md
section 8c), seeds, builds small models, and writes inputs / state_dict / masks /
outputs / loss / gradients as ``tests/golden/*.npz``.  The reference never
travels to the GPU box; these data files do.  All tensors are fp32 CPU results
(the reference's PyTorch-CPU path: autocast and GradScaler are inert on CPU).
"""
import os
import sys

import numpy as np
import torch

REF = os.environ.get("VITSSL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))

from vit_core.vit import ViT  # noqa: E402
from vit_core.attention import ScaledDotProductAttention  # noqa: E402
from vit_core.encoder_block import EncoderBlock  # noqa: E402
from vit_core.ssl.simmim.model import SimMIMViT  # noqa: E402
from vit_core.ssl.simmim.masking import simple_masking  # noqa: E402
from vit_core.ssl.dino.model import DINOViT  # noqa: E402
from vit_core.ssl.dino.loss import DINOLoss  # noqa: E402
from vit_core.ssl.dino.dino_utils import DINOMomentumScheduler, DINOTeacherTempScheduler  # noqa: E402

sys.path.insert(0, OUT)
from synth import BIG_KEYS, dino_big_weights, summarize  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path)/1024:.0f} KiB, {len(arrs)} arrays")


def img_u8(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)


def sd_arrays(model, prefix="sd/"):
    return {prefix + k: npy(v) for k, v in model.state_dict().items()}


def grad_arrays(model, prefix="grad/"):
    return {prefix + k: npy(p.grad) for k, p in model.named_parameters() if p.grad is not None}


def simmim_case(name, seed, B, img, patch, D, H, F, blocks, ratio):
    torch.manual_seed(seed)
    model = SimMIMViT(num_blocks=blocks, input_shape=(3, img, img), embed_dim=D, patch_size=patch,
                      num_heads=H, mlp_dim=F, dropout=0.0, mask_ratio=ratio)
    model.train()
    xu8 = img_u8((B, 3, img, img), seed + 1)
    x = xu8.float() / 256.0
    torch.manual_seed(seed + 2)  # the masking RNG state the oracle must replay
    pred, tgt, bm = model(x, return_bool_mask=True)
    loss = torch.nn.L1Loss(reduction="mean")(pred, tgt)
    loss.backward()
    with torch.no_grad():
        model.eval()
        feat = model.inference_forward(x)
    save(name, x_u8=npy(xu8), mask=npy(bm[..., 0]), pred=npy(pred), targets=npy(tgt), loss=npy(loss),
         feat=npy(feat),
         cfg=np.array([B, img, patch, D, H, F, blocks], dtype=np.int64), ratio=np.array(ratio),
         mask_seed=np.array(seed + 2), **sd_arrays(model), **grad_arrays(model))


def masking_case():
    out = {}
    for i, (seed, B, N, ratio) in enumerate([(0, 3, 196, 0.6), (7, 4, 16, 0.5), (123, 2, 49, 0.75), (9, 5, 196, 0.6)]):
        torch.manual_seed(seed)
        patches = torch.zeros(B, N, 4)
        _, bm, _ = simple_masking(patches, ratio)
        out[f"mask{i}"] = npy(bm)
        out[f"args{i}"] = np.array([seed, B, N], dtype=np.int64)
        out[f"ratio{i}"] = np.array(ratio)
    # target row order: ascending (b, n)
    torch.manual_seed(3)
    patches = torch.arange(2 * 9 * 2, dtype=torch.float32).reshape(2, 9, 2)
    _, bm, tg = simple_masking(patches, 0.5)
    out["order_mask"] = npy(bm)
    out["order_targets"] = npy(tg)
    save("masking", **out)


def ops_case():
    torch.manual_seed(11)
    q, k, v = (torch.randn(2, 3, 20, 64) for _ in range(3))
    o, p = ScaledDotProductAttention(q, k, v, return_attn=True)
    blk = EncoderBlock(d_model=128, num_heads=2, mlp_dim=192, dropout=0.0)
    x = torch.randn(3, 10, 128, requires_grad=True)
    y, probs = blk(x, return_attn=True)
    y.square().sum().backward()
    arrs = dict(q=npy(q), k=npy(k), v=npy(v), o=npy(o), p=npy(p), blk_x=npy(x), blk_y=npy(y),
                blk_probs=npy(probs), blk_dx=npy(x.grad))
    arrs.update({"blk_sd/" + k_: npy(v_) for k_, v_ in blk.state_dict().items()})
    arrs.update({"blk_grad/" + k_: npy(p_.grad) for k_, p_ in blk.named_parameters()})
    save("ops", **arrs)


def vit_case():
    torch.manual_seed(21)
    B, img, patch, D, H, F, blocks, C = 4, 32, 8, 128, 2, 192, 2, 10
    model = ViT(num_classes=C, num_blocks=blocks, input_shape=(3, img, img), embed_dim=D, patch_size=patch,
                num_heads=H, mlp_dim=F, dropout=0.0)
    xu8 = img_u8((B, 3, img, img), 22)
    x = xu8.float() / 256.0
    labels = torch.tensor([3, 0, 9, 3])
    logits, attn = model(x, return_attn=True)
    loss = torch.nn.CrossEntropyLoss()(logits, labels)
    loss.backward()
    save("vit_tiny", x_u8=npy(xu8), labels=npy(labels), logits=npy(logits), attn=npy(attn), loss=npy(loss),
         cfg=np.array([B, img, patch, D, H, F, blocks, C], dtype=np.int64), **sd_arrays(model), **grad_arrays(model))


def dino_case():
    torch.manual_seed(31)
    B, gi, li, patch, D, H, F, blocks, K = 2, 32, 16, 8, 64, 1, 128, 1, 256
    G, L = 2, 2
    model = DINOViT(num_blocks=blocks, input_shape=(3, gi, gi), embed_dim=D, patch_size=patch, num_heads=H,
                    mlp_dim=F, dropout=0.0, output_dim=K, center_momentum=0.9)
    # make student != teacher and the center non-trivial so every term is exercised
    with torch.no_grad():
        for p in list(model.student_backbone.parameters()) + list(model.student_head.parameters()):
            p.add_(0.02 * torch.randn_like(p))
        model.center.copy_(0.05 * torch.randn(1, K))
        big = dino_big_weights(D)
        sdm = model.state_dict()
        for k, a in big.items():
            sdm[k].copy_(torch.from_numpy(a))
    is_big = lambda k: any(k.endswith(b) for b in BIG_KEYS) and "_head." in k
    center0 = model.center.clone()
    views_u8 = [img_u8((B, 3, gi, gi), 40 + i) for i in range(G)] + [img_u8((B, 3, li, li), 50 + i) for i in range(L)]
    views = [v.float() / 256.0 for v in views_u8]
    sd0 = {"sd/" + k: npy(v).copy() for k, v in model.state_dict().items() if not is_big(k)}
    model.train()
    teacher, student = model(views, G)
    crit = DINOLoss(teacher_temp=0.04, student_temp=0.1)
    loss = crit(teacher.view(G, B, K), student.view(G + L, B, K), model.center)
    loss.backward()
    grads = {}
    for k, p in model.named_parameters():
        if p.grad is None:
            continue
        if is_big(k):
            for sk, sv in summarize(npy(p.grad)).items():
                grads[f"gradsum/{k}/{sk}"] = sv
        else:
            grads["grad/" + k] = npy(p.grad)
    center1 = model.center.clone()
    model.momentum_update_teacher(0.996)
    ema = {}
    for k, v in model.state_dict().items():
        if not k.startswith("teacher_"):
            continue
        if is_big(k):
            for sk, sv in summarize(npy(v)).items():
                ema[f"emasum/{k}/{sk}"] = sv
        else:
            ema["ema/" + k] = npy(v)
    feats = model.inference_forward(views[0], return_features=True)
    arrs = {f"view{i}_u8": npy(v) for i, v in enumerate(views_u8)}
    save("dino_tiny", teacher=npy(teacher), student=npy(student), center0=npy(center0), center1=npy(center1),
         loss=npy(loss), feats=npy(feats),
         cfg=np.array([B, gi, li, patch, D, H, F, blocks, K, G, L], dtype=np.int64), **arrs, **sd0, **grads, **ema)
    # schedulers
    ms = DINOMomentumScheduler(0.996, 1.0, 100)
    ts = DINOTeacherTempScheduler(0.04, 0.07, 30)
    tl = DINOTeacherTempScheduler(0.04, 0.07, 30, "linear")
    steps = np.arange(0, 121, 7)
    save("dino_sched", steps=steps, mom=np.array([ms.get_momentum(int(s)) for s in steps]),
         temp_cos=np.array([ts.get_temp(int(s)) for s in steps]),
         temp_lin=np.array([tl.get_temp(int(s)) for s in steps]))


def adamw_case():
    torch.manual_seed(61)
    p = torch.nn.Parameter(torch.randn(257))
    opt = torch.optim.AdamW([p], lr=1e-3, weight_decay=1e-3)
    hist = [npy(p).copy()]
    grads = []
    for i in range(4):
        g = torch.randn(257)
        p.grad = g.clone()
        opt.step()
        grads.append(npy(g))
        hist.append(npy(p).copy())
    save("adamw", params=np.stack(hist), grads=np.stack(grads), lr=np.array(1e-3), wd=np.array(1e-3))


if __name__ == "__main__":
    torch.set_num_threads(4)
    masking_case()
    ops_case()
    simmim_case("simmim_tiny", seed=100, B=3, img=32, patch=8, D=128, H=2, F=192, blocks=2, ratio=0.6)
    simmim_case("simmim_n196", seed=200, B=2, img=224, patch=16, D=64, H=1, F=128, blocks=1, ratio=0.6)
    vit_case()
    dino_case()
    adamw_case()
