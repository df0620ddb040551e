#!/usr/bin/env python3
"""Round-2 golden vectors generated from the reference itself (build container only;
needs /root/reference).  Adds to make_golden.py's fixtures:

  * ``ops_drop`` / ``simmim_drop``: an EncoderBlock and a SimMIMViT run in train mode
    with dropout = 0.1.  The reference's own ``F.dropout`` produces the outputs; the keep
    masks of its three sites per block (drop1, FFN inner, drop2) are captured by replaying the
    SAME generator state through the same ``F.dropout`` on a tensor of ones, so the masks are
    exactly the ones ATen drew (vit_core/encoder_block.py:45-46,51-52, feed_forward.py:27).
  * ``manual_embed``: ManualPatchEmbedding forward + every gradient
    (vit_core/patch_embedding.py:122-128).
  * ``dino_trainer_sched``: teacher temperature / momentum per epoch exactly as the reference
    trainer builds and evaluates them (utils/trainers/dino_trainer.py:16-29,46,80) on the
    shipped configs/dino/training.yaml values.
  * ``ckpt_ref_simmim.pth`` + ``ckpt_ref_expected``: a checkpoint written with the dict layout
    of utils/trainers/base_trainer.py:99-118 from a torch.compile-wrapped reference SimMIMViT
    (so keys carry the ``_orig_mod.`` prefix the reference really emits) after 3 AdamW steps,
    and the state the reference's own load_weights (utils/model_builder.py:11-89) produces
    from the un-prefixed tensors in a reference fine-tuning ViT.

    python tests/golden/make_golden_r2.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = os.environ.get("VITSSL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
OUT = os.path.dirname(os.path.abspath(__file__))

from vit_core.vit import ViT  # noqa: E402
from vit_core.encoder_block import EncoderBlock  # noqa: E402
from vit_core.patch_embedding import ManualPatchEmbedding  # noqa: E402
from vit_core.ssl.simmim.model import SimMIMViT  # noqa: E402
from vit_core.ssl.dino.dino_utils import DINOMomentumScheduler, DINOTeacherTempScheduler  # noqa: E402


def npy(t):
    return t.detach().cpu().numpy().copy()


def save(name, **arrs):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"{name}: {os.path.getsize(path)/1024:.0f} KiB, {len(arrs)} arrays")


def img_u8(shape, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, shape, generator=g, dtype=torch.uint8)


class CaptureDropout:
    """Context manager: every F.dropout call made by the reference modules is executed by
    the real F.dropout; its keep mask is recovered by re-running the same call on ones from
    the same generator state (same shape and dtype => same Bernoulli draws)."""

    def __init__(self):
        self.keeps = []

    def __enter__(self):
        self.orig = F.dropout

        def wrapped(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return self.orig(x, p, training, inplace)
            state = torch.get_rng_state()
            ones = self.orig(torch.ones_like(x), p, True, False)
            after = torch.get_rng_state()
            torch.set_rng_state(state)
            out = self.orig(x, p, True, inplace)
            assert torch.equal(torch.get_rng_state(), after)
            keep = ones != 0
            assert torch.equal(out != 0, keep & (x != 0))
            self.keeps.append(keep)
            return out

        F.dropout = wrapped
        torch.nn.functional.dropout = wrapped
        return self

    def __exit__(self, *a):
        F.dropout = self.orig
        torch.nn.functional.dropout = self.orig


def pack(keep):
    return np.packbits(npy(keep).astype(np.uint8).reshape(-1))


def ops_drop_case():
    torch.manual_seed(311)
    blk = EncoderBlock(d_model=128, num_heads=2, mlp_dim=192, dropout=0.1)
    blk.train()
    x = torch.randn(3, 10, 128, requires_grad=True)
    with CaptureDropout() as cap:
        y, _ = blk(x)
    assert len(cap.keeps) == 3          # call order: drop1, FFN inner, drop2
    y.square().sum().backward()
    arrs = dict(x=npy(x), y=npy(y), dx=npy(x.grad), p=np.array(0.1),
                keep1=pack(cap.keeps[0]), keep_inner=pack(cap.keeps[1]), keep2=pack(cap.keeps[2]),
                keep1_shape=np.array(cap.keeps[0].shape), keep_inner_shape=np.array(cap.keeps[1].shape),
                keep2_shape=np.array(cap.keeps[2].shape))
    arrs.update({"sd/" + k: npy(v) for k, v in blk.state_dict().items()})
    arrs.update({"grad/" + k: npy(p.grad) for k, p in blk.named_parameters()})
    save("ops_drop", **arrs)


def simmim_drop_case():
    seed, B, img, patch, D, H, Fd, blocks, ratio = 400, 3, 32, 8, 128, 2, 192, 2, 0.6
    torch.manual_seed(seed)
    model = SimMIMViT(num_blocks=blocks, input_shape=(3, img, img), embed_dim=D, patch_size=patch,
                      num_heads=H, mlp_dim=Fd, dropout=0.1, mask_ratio=ratio)
    model.train()
    xu8 = img_u8((B, 3, img, img), seed + 1)
    x = xu8.float() / 256.0
    torch.manual_seed(seed + 2)          # masking draws first, dropout after (reference order)
    with CaptureDropout() as cap:
        pred, tgt, bm = model(x, return_bool_mask=True)
    assert len(cap.keeps) == 3 * blocks
    loss = torch.nn.L1Loss(reduction="mean")(pred, tgt)
    loss.backward()
    arrs = dict(x_u8=npy(xu8), mask=npy(bm[..., 0]), pred=npy(pred), targets=npy(tgt), loss=npy(loss), p=np.array(0.1),
                cfg=np.array([B, img, patch, D, H, Fd, blocks], dtype=np.int64), ratio=np.array(ratio),
                mask_seed=np.array(seed + 2))
    for i, k in enumerate(cap.keeps):
        arrs[f"keep{i // 3}_{i % 3}"] = pack(k)
        arrs[f"keep{i // 3}_{i % 3}_shape"] = np.array(k.shape)
    arrs.update({"sd/" + k: npy(v) for k, v in model.state_dict().items()})
    arrs.update({"grad/" + k: npy(p.grad) for k, p in model.named_parameters() if p.grad is not None})
    save("simmim_drop", **arrs)


def manual_embed_case():
    torch.manual_seed(521)
    B, img, patch, D = 3, 32, 8, 128
    m = ManualPatchEmbedding((3, img, img), D, patch)
    xu8 = img_u8((B, 3, img, img), 522)
    x = xu8.float() / 256.0
    y = m(x)
    w = torch.randn_like(y)
    (y * w).sum().backward()
    arrs = dict(x_u8=npy(xu8), y=npy(y), w=npy(w), cfg=np.array([B, img, patch, D], dtype=np.int64))
    arrs.update({"sd/" + k: npy(v) for k, v in m.state_dict().items()})
    arrs.update({"grad/" + k: npy(p.grad) for k, p in m.named_parameters()})
    save("manual_embed", **arrs)


def dino_trainer_sched_case():
    """What DINOTrainer.__init__ / fit / train_epoch compute for configs/dino/training.yaml
    (teacher_temp 0.04 -> teacher_temp_final 0.07, cosine, momentum 0.996 -> 1, num_epochs 100),
    evaluated at epoch = 1 .. num_epochs + 2 (the trainers count epochs from 1)."""
    num_epochs = 100
    ts = DINOTeacherTempScheduler(0.04, 0.07, num_epochs, "cosine")
    tl = DINOTeacherTempScheduler(0.04, 0.07, num_epochs, "linear")
    t_const = DINOTeacherTempScheduler(0.05, 0.05, num_epochs, "cosine")      # teacher_temp_final absent -> teacher_temp
    ms = DINOMomentumScheduler(0.996, 1, num_epochs)
    ep = np.arange(1, num_epochs + 3)
    save("dino_trainer_sched", epochs=ep, num_epochs=np.array(num_epochs),
         temp_cos=np.array([ts.get_temp(int(e)) for e in ep]), temp_lin=np.array([tl.get_temp(int(e)) for e in ep]),
         temp_const=np.array([t_const.get_temp(int(e)) for e in ep]), mom=np.array([ms.get_momentum(int(e)) for e in ep]))


def _load_ref_model_builder():
    spec = importlib.util.spec_from_file_location("ref_model_builder", os.path.join(REF, "utils", "model_builder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def checkpoint_case():
    seed, B, img, patch, D, H, Fd, blocks = 700, 2, 32, 8, 64, 1, 128, 2
    torch.manual_seed(seed)
    model = SimMIMViT(num_blocks=blocks, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H,
                      mlp_dim=Fd, dropout=0.0, mask_ratio=0.6)
    compiled = torch.compile(model)          # what build_model returns (utils/model_builder.py:182-183); never run here
    opt = torch.optim.AdamW([p for p in compiled.parameters() if p.requires_grad], lr=1e-3, weight_decay=1e-3)
    x = img_u8((B, 3, img, img), seed + 1).float() / 256.0
    for step in range(3):                    # three real AdamW steps so the optimizer state is non-trivial
        opt.zero_grad(set_to_none=True)
        torch.manual_seed(seed + 10 + step)
        pred, tgt = model(x)                 # eager forward of the wrapped module (same parameters)
        torch.nn.L1Loss()(pred, tgt).backward()
        opt.step()
    config = {"training": {"type": "simmim"}, "model": {"embed_dim": D}}
    ckpt = {"epoch": 3, "model_state_dict": compiled.state_dict(), "optimizer_state_dict": opt.state_dict(),
            "best_val_loss": 0.125, "config": config}                       # base_trainer.py:99-105
    assert all(k.startswith("_orig_mod.") for k in ckpt["model_state_dict"])
    path = os.path.join(OUT, "ckpt_ref_simmim.pth")
    torch.save(ckpt, path)
    print(f"ckpt_ref_simmim.pth: {os.path.getsize(path)/1024:.0f} KiB")

    # ground truth of the fine-tune remap: the reference's own load_weights on the un-prefixed tensors
    mb = _load_ref_model_builder()
    plain = os.path.join(OUT, "_tmp_plain.pth")
    torch.save({"model_state_dict": model.state_dict()}, plain)
    torch.manual_seed(seed + 50)
    vit = ViT(num_classes=10, num_blocks=blocks, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H,
              mlp_dim=Fd, dropout=0.0)
    init = {k: v.clone() for k, v in vit.state_dict().items()}
    mb.load_weights(vit, plain)
    os.remove(plain)
    arrs = {"vit_init/" + k: npy(v) for k, v in init.items()}
    arrs.update({"vit_loaded/" + k: npy(v) for k, v in vit.state_dict().items()})
    arrs.update({"simmim/" + k: npy(v) for k, v in model.state_dict().items()})
    # one more AdamW step of the reference from the saved state: what a resumed optimizer must reproduce
    opt.zero_grad(set_to_none=True)
    torch.manual_seed(seed + 99)
    g = {k: torch.randn_like(p) for k, p in model.named_parameters()}
    for k, p in model.named_parameters():
        p.grad = g[k].clone()
    opt.step()
    arrs.update({"step4_grad/" + k: npy(v) for k, v in g.items()})
    arrs.update({"step4_param/" + k: npy(p) for k, p in model.named_parameters()})
    arrs["cfg"] = np.array([B, img, patch, D, H, Fd, blocks, 10], dtype=np.int64)
    arrs["lr_wd"] = np.array([1e-3, 1e-3])
    save("ckpt_ref_expected", **arrs)


if __name__ == "__main__":
    torch.set_num_threads(4)
    ops_drop_case()
    simmim_drop_case()
    manual_embed_case()
    dino_trainer_sched_case()
    checkpoint_case()
