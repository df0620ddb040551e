#!/usr/bin/env python3
"""Random-shape sweep of the op-parity checks (developer tool, GPU box): runs the bodies of tests/test_gpu_ops.py on
shapes the parametrised lists do not name -- ragged M / N, every N % 8 / N % 4 residue class the entry points accept,
token counts 1..256 -- and prints the first failing shape.

    SEED=1 BUDGET_S=150 [KINDS=nt,tn,attn,ln,nt8,tn8,tnb,simmim,vit,dino] [CASES=n] python tools/fuzz_ops.py
(exit code 1 on the first failure)"""
import contextlib
import os
import random
import sys
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "vit-ssl_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import test_gpu_ops as T  # noqa: E402
from _util import l1_backward_with_signs  # noqa: E402
import test_gpu_fp8 as T8  # noqa: E402
import test_gpu_round3 as T3  # noqa: E402
from vitssl_hip import ops  # noqa: E402


WORST = {}        # kind -> (largest gradient distance seen in this sweep, parameter, case); printed at the end of run()


def simmim_case(_ops, B, img, patch, D, H, F):
    """tests/test_gpu_models.py::test_edge_batches_against_oracle on one random configuration, judged against the oracle's
    modes (fp32, bf16 emulation, bf16 emulation with the flash-style attention backward) for the SAME d(loss)/d(pred):
    dL1/dpred = sign(pred - target) / n is discontinuous, so with a few hundred masked elements one element whose difference
    changes sign between two bf16 rounding orders moves every gradient by several per cent (triaged in round 4,
    tools/probes/triage_mask_token.py: 9 % against the bf16-emulating oracle, 0.3-1.2 % against the fp32 one, same run); the
    oracle's backward therefore starts from the engine's signs (l1_backward_with_signs).
    With the same dpred on both sides the largest distance of any gradient over several hundred random models, 4-token ones
    included, is under 1 % against every mode (the 6-8 % on the query / key weight gradients of 4-token models that an earlier
    version of this sweep put down to the flash-style delta were such sign flips too): the bar is 2e-2."""
    import torch
    from _util import rel_l2
    from oracle import vit_oracle as O
    from vit_core.ssl.simmim import SimMIMViT
    from vit_core.ssl.simmim.masking import draw_mask
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 100 + img)
    model = SimMIMViT(num_blocks=2, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H, mlp_dim=F,
                      dropout=0.0, mask_ratio=0.6)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    x = torch.rand(B, 3, img, img)
    N = (img // patch) ** 2
    torch.manual_seed(77)
    mask = draw_mask(B, N, 0.6)
    torch.manual_seed(77)
    pred, tgt, bm = model(x.to(dev), return_bool_mask=True)
    assert torch.equal(bm[..., 0].cpu(), mask)
    torch.nn.L1Loss()(pred, tgt).backward()
    worst = {}
    for emu in ("flash", "bf16", None):                      # "flash": the bf16 mode with the flash-style attention backward (oracle sdpa())
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        with (O.flash_delta() if emu == "flash" else contextlib.nullcontext()):
            pe, te = O.simmim_forward(leaves, x, mask, patch, H, emu="bf16" if emu == "flash" else emu)
        assert pred.shape == pe.shape and torch.equal(tgt.cpu(), te)
        assert rel_l2(pred, pe) < (1e-2 if emu else 2e-2), (emu, rel_l2(pred, pe))
        l1_backward_with_signs(pe, te, pred, tgt)
        worst[emu] = max((rel_l2(p.grad, leaves[k].grad), k) for k, p in model.named_parameters())
    if os.environ.get("FUZZ_VERBOSE"):
        print("  worst gradient per oracle mode:", {str(m): (round(w[0], 4), w[1]) for m, w in worst.items()}, flush=True)
    assert min(w[0] for w in worst.values()) < 2e-2, worst
    best = min(worst.values())
    if best[0] > WORST.get("simmim", (0.0,))[0]:
        WORST["simmim"] = (best[0], best[1], (B, img, patch, D, H, F))


def simmim_drop_case(_ops, B, img, patch, D, H, F, blocks, p):
    """Dropout ON through the fused train_step (the bench's path): the engine's masks are exported and handed to the oracle;
    prediction, loss and every parameter gradient (tests/test_gpu_models.py::test_simmim_dropout_backward_... on a random
    configuration; any oracle mode within 2e-2 for the engine's own d(loss)/d(pred), see simmim_case)."""
    import torch
    from _util import rel_l2
    from oracle import vit_oracle as O
    from vit_core import _runtime as R
    from vit_core.ssl.simmim import SimMIMViT
    from vit_core.ssl.simmim.masking import draw_mask
    from vitssl_hip.optim import FusedAdamW
    import test_gpu_models as TM
    dev = torch.device("cuda:0")
    N = (img // patch) ** 2
    torch.manual_seed(B * 100 + img + blocks)
    model = SimMIMViT(num_blocks=blocks, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H, mlp_dim=F,
                      dropout=p, mask_ratio=0.6)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    x = torch.rand(B, 3, img, img)
    torch.manual_seed(91)
    mask = draw_mask(B, N, 0.6)
    opt = FusedAdamW(model.flat_store(), lr=1e-4, weight_decay=0.0)
    torch.manual_seed(92)
    loss = model.train_step(x.to(dev), opt, mask_cpu=mask)
    st = model.flat_store()
    grads = {k: st.gview(k).view(sd[k].shape) for k in st.names}
    pred, tgt = model.last_pred, model.last_targets
    torch.manual_seed(92)
    seed = R.next_seed()
    keeps = TM._export_keeps(p, seed, blocks, B * N, D, F, (B, N))
    ref = {}
    for emu in ("flash", "bf16", None):                      # "flash": the bf16 mode with the flash-style attention backward (oracle sdpa())
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        with (O.flash_delta() if emu == "flash" else contextlib.nullcontext()):
            pe, te = O.simmim_forward(leaves, x, mask, patch, H, emu="bf16" if emu == "flash" else emu, keeps=keeps,
                                      p_drop=round(p * 65536) / 65536)
        assert torch.equal(tgt.cpu(), te) and rel_l2(pred, pe) < (1e-2 if emu else 2e-2), (emu, rel_l2(pred, pe))
        wl = O.l1_loss_mean(pe, te)
        l1_backward_with_signs(pe, te, pred, tgt)
        assert abs(float(loss) - float(wl.detach())) < 1e-2 * float(wl.detach())
        ref[emu] = {k: v.grad for k, v in leaves.items()}
    # A parameter on whose gradient the oracle's own two rounding modes disagree by s (same masks, same dpred) is allowed 3 s;
    # everywhere else the bar is 2e-2 (largest distance seen in the sweeps: 0.8 %).
    for k in ref[None]:
        spread = rel_l2(ref["bf16"][k], ref[None][k])
        errs = {m: rel_l2(grads[k], ref[m][k]) for m in ref}
        err = min(errs.values())
        if os.environ.get("FUZZ_VERBOSE"):                   # triage: every parameter's distance instead of the first failure
            print(f"  {k:60s} vs flash {errs['flash']:.4f} bf16 {errs['bf16']:.4f} fp32 {errs[None]:.4f} spread {spread:.4f} "
                  f"|g| {float(ref[None][k].norm()):.3e}", flush=True)
            continue
        assert err < max(2e-2, 3 * spread), (k, err, spread)
        if err > WORST.get("simdrop", (0.0,))[0]:
            WORST["simdrop"] = (err, k, (B, img, patch, D, H, F, blocks, p))


def simmim_fp8_case(_ops, B, img, patch, D, H, F):
    """e4m3 operands in every Linear GEMM of the blocks (BASELINE configs[4]'s path) on a random 2-block SimMIM model against
    the oracle's fp8 mode run with the gradient scales the engine used (tests/test_gpu_fp8.py::test_simmim_fp8_matches_oracle...)."""
    import torch
    from _util import rel_l2
    from oracle import vit_oracle as O
    from vit_core.ssl.simmim import SimMIMViT
    from vit_core.ssl.simmim.masking import draw_mask
    from vitssl_hip import engine
    dev = torch.device("cuda:0")
    N = (img // patch) ** 2
    engine.set_linear_operands("fp8")
    try:
        torch.manual_seed(B * 100 + img)
        model = SimMIMViT(num_blocks=2, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H, mlp_dim=F,
                          dropout=0.0, mask_ratio=0.6)
        sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
        model = model.to(dev).train()
        x = torch.rand(B, 3, img, img)
        torch.manual_seed(9)
        pred, tgt = model(x.to(dev))
        loss = torch.nn.functional.l1_loss(pred, tgt)
        loss.backward()
        torch.manual_seed(9)
        mask = draw_mask(B, N, 0.6)
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        pe, te = O.simmim_forward(leaves, x, mask, patch, H, emu="fp8", fp8_gscales=model.runtime().stack.fp8_grad_scales().cpu().tolist())
        assert torch.equal(tgt.cpu(), te) and rel_l2(pred, pe) < 2e-2, rel_l2(pred, pe)      # (tests/test_gpu_fp8.py holds 1e-2 on its fixed models; 1.07e-2 seen here on a random one)
        wl = O.l1_loss_mean(pe, te)
        l1_backward_with_signs(pe, te, pred, tgt, frac=3e-2, mag=1e-1)      # (e4m3 operands: predictions up to 2 % apart)
        assert abs(float(loss) - float(wl.detach())) < 1e-2 * float(wl.detach())
        # The bar is set by how much e4m3 operands move each gradient at all (fp8 mode against fp32 mode of the oracle, `spread`):
        # two implementations of the same quantised arithmetic differ where a rounding falls the other way (6 % of that element
        # per e4m3 flip), and in the small, cancelling gradients of 9-token models (query / key weights, position embedding)
        # such flips decorrelate the two results: their distance approaches sqrt(2) x spread (seen: 9.0 % at a spread of 13.6 %,
        # 9.5 % at 12.5 %, 8.4 % at 7.4 %).  So: within 1.5 x spread of the fp8 mode (8e-2 where the spread is smaller), and
        # as close to the fp32 gradient as the emulation is (1.5 x spread + 2e-2).
        l32 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        p32, t32 = O.simmim_forward(l32, x, mask, patch, H, emu=None)
        l1_backward_with_signs(p32, t32, pred, tgt, strict=False)   # (the fp32 prediction is up to 8 % away: more signs differ)
        for k, p in model.named_parameters():
            err, spread = rel_l2(p.grad, leaves[k].grad), rel_l2(leaves[k].grad, l32[k].grad)
            assert err < max(8e-2, 1.5 * spread), (k, err, spread)
            assert rel_l2(p.grad, l32[k].grad) < 1.5 * spread + 2e-2, (k, rel_l2(p.grad, l32[k].grad), spread)
    finally:
        engine.set_linear_operands("bf16")


def vit_case(_ops, B, img, patch, D, H, F, classes):
    """Supervised ViT (conv patch embedding + CLS token, MLP head): logits, attention maps, the cross-entropy value, and every
    gradient of a LINEAR functional sum(logits * R) against the oracle.  The linear functional hands both sides the same
    d(logits): with cross entropy on a few samples and classes, sum_b (softmax - onehot) cancels and a 0.15 % difference of the
    logits becomes 5-14 % on every bias-type gradient (triaged in round 4 on B = 5, 2 classes: tools/probes/triage_vit_case.py;
    the head bias gradient there is an exact fp32 column sum of autograd's own d(logits) and still sits 7 % from the oracle's)."""
    import torch
    from _util import rel_l2
    from oracle import vit_oracle as O
    from vit_core import ViT
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 1000 + img + classes)
    model = ViT(num_classes=classes, num_blocks=2, input_shape=(3, img, img), embed_dim=D, patch_size=patch, num_heads=H,
                mlp_dim=F, dropout=0.0)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    x = torch.rand(B, 3, img, img)
    labels = torch.randint(0, classes, (B,))
    R = torch.randn(B, classes)
    logits, attn = model(x.to(dev), return_attn=True)
    ce = torch.nn.CrossEntropyLoss()(logits, labels.to(dev))
    (logits * R.to(dev)).sum().backward()
    worst = {}
    for emu in ("bf16", None):
        leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        lo, pr = O.vit_forward(leaves, x, patch, H, emu=emu, return_attn=True)
        # (a handful of logits near zero: 2e-2 against the bf16-emulating mode, 4e-2 against fp32)
        assert rel_l2(logits, lo) < (2e-2 if emu else 4e-2) and rel_l2(attn, pr) < 2e-2, (emu, rel_l2(logits, lo), rel_l2(attn, pr))
        assert abs(float(ce) - float(O.cross_entropy_mean(lo, labels))) < 1e-2 * abs(float(ce)) + 1e-3
        (lo * R).sum().backward()
        worst[emu] = max((rel_l2(p.grad, leaves[k].grad), k) for k, p in model.named_parameters())
    assert min(w[0] for w in worst.values()) < 6e-2, worst


def dino_case(_ops, B, gi, li, patch, D, H, F, K, G, Lv):
    """DINOViT (teacher + student, bicubic positional embedding for the local crops, weight-normed head, centre update) and
    DINOLoss: outputs, centre, loss and every student gradient against the oracle."""
    import torch
    from _util import rel_l2
    from oracle import vit_oracle as O
    from vit_core.ssl.dino import DINOViT
    from vit_core.ssl.dino.loss import DINOLoss
    dev = torch.device("cuda:0")
    torch.manual_seed(B * 1000 + gi + li + K)
    model = DINOViT(num_blocks=2, input_shape=(3, gi, gi), embed_dim=D, patch_size=patch, num_heads=H, mlp_dim=F, dropout=0.0,
                    output_dim=K, center_momentum=0.9)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.to(dev).train()
    views = [torch.rand(B, 3, gi, gi) for _ in range(G)] + [torch.rand(B, 3, li, li) for _ in range(Lv)]
    teacher, student = model([v.to(dev) for v in views], G)
    crit = DINOLoss(teacher_temp=0.04, student_temp=0.1)
    loss = crit(teacher.view(G, B, K), student.view(G + Lv, B, K), model.center)
    loss.backward()
    worst = {}
    for emu in ("bf16", None):
        leaves = {k: (v.clone().requires_grad_(True) if k.startswith("student_") else v.clone()) for k, v in sd.items()}
        te, stu, c1 = O.dino_forward(leaves, views, G, patch, H, (gi // patch, gi // patch), sd["center"], 0.9, emu=emu)
        assert rel_l2(teacher, te) < 2e-2 and rel_l2(student, stu) < 2e-2 and rel_l2(model.center, c1) < 2e-2, (
            emu, rel_l2(teacher, te), rel_l2(student, stu), rel_l2(model.center, c1))
        lo = O.dino_loss_naive(te.view(G, B, K), stu.view(G + Lv, B, K), c1, 0.04, 0.1)
        assert abs(float(loss) - float(lo)) < 1e-2 * abs(float(lo)), (emu, float(loss), float(lo))
        lo.backward()
        worst[emu] = max((rel_l2(p.grad, leaves[k].grad), k) for k, p in model.named_parameters() if k.startswith("student_"))
    assert min(w[0] for w in worst.values()) < 8e-2, worst


def run(seed=0, kinds="nt,nt,tn,attn,ln,nt8,tn8,tnb", budget_s=120.0, max_cases=None):
    """Returns 0, or 1 after printing the first failing case.  `max_cases` bounds the sweep by count (a deterministic list of
    cases for a given seed: tests/test_gpu_fuzz.py), `budget_s` by time."""
    rng = random.Random(int(seed))
    budget = float(budget_s)
    t0 = time.time()
    n = 0
    cases = []
    while time.time() - t0 < budget and (max_cases is None or n < max_cases):
        kind = rng.choice(kinds.split(","))
        if kind == "nt":
            big = rng.random() < 0.15
            M = rng.randint(1, 40000) if big else rng.randint(1, 3000)
            N = 4 * rng.randint(1, 200 if big else 700)
            K = 64 * rng.randint(1, 3 if big else 16)
            args = (M, N, K)
            fn = T.test_gemm_nt_epilogues
        elif kind == "tn":
            M = rng.randint(1, 9000)
            N1 = 8 * rng.randint(1, 150)
            N2 = 8 * rng.randint(1, 150)
            args = (M, N1, N2)
            fn = T.test_gemm_tn
        elif kind == "nt8":                                  # e4m3 operands: N % 8 == 0, K % 128 == 0
            args = (rng.randint(1, 3000), 8 * rng.randint(1, 300), 128 * rng.randint(1, 8))
            fn = lambda _ops, *a: T8.test_gemm_fp8_epilogues(*a)  # noqa: E731
        elif kind == "tn8":                                  # e4m3 weight gradients: N1, N2 % 16 == 0
            args = (rng.randint(1, 9000), 16 * rng.randint(1, 70), 16 * rng.randint(1, 70))
            fn = lambda _ops, *a: T8.test_gemm_fp8_tn_weight_gradient(*a)  # noqa: E731
        elif kind == "tnb":                                  # several weight gradients in one launch
            args = (rng.randint(1, 6000), [(8 * rng.randint(1, 200), 8 * rng.randint(1, 200)) for _ in range(rng.randint(1, 5))])
            fn = lambda _ops, *a: T3.test_gemm_tn_batch_matches_single_launches(*a)  # noqa: E731
        elif kind == "simmim":                               # whole 2-block SimMIM model (forward, loss, every gradient) against the oracle
            patch = rng.choice([8, 16])
            H = rng.choice([1, 2, 3, 6])                     # (6 heads = 384 columns: ViT-S's width, the row-pair LayerNorm backward)
            args = (rng.randint(1, 6), patch * rng.randint(2, 16 if patch == 8 else 8), patch, 64 * H, H, 64 * rng.randint(1, 6))
            fn = simmim_case
        elif kind == "simdrop":
            patch = rng.choice([8, 16])
            H = rng.choice([1, 2, 3, 6])
            args = (rng.randint(1, 5), patch * rng.randint(2, 12 if patch == 8 else 8), patch, 64 * H, H, 64 * rng.randint(1, 6),
                    rng.randint(1, 3), rng.choice([0.05, 0.1, 0.25, 0.5]))
            fn = simmim_drop_case
        elif kind == "sim8":                                 # e4m3 operands: D and F multiples of 128
            patch = rng.choice([8, 16])
            H = rng.choice([2, 4])
            args = (rng.randint(1, 5), patch * rng.randint(3, 12 if patch == 8 else 8), patch, 64 * H, H, 128 * rng.randint(1, 4))
            fn = simmim_fp8_case
        elif kind == "vit":
            patch = rng.choice([8, 16])
            H = rng.randint(1, 3)
            args = (rng.randint(1, 6), patch * rng.randint(2, 15 if patch == 8 else 8), patch, 64 * H, H, 64 * rng.randint(1, 6), rng.randint(2, 40))
            fn = vit_case
        elif kind == "dino":
            patch = rng.choice([8, 16])
            H = rng.randint(1, 2)
            gi = patch * rng.randint(3, 8)
            li = patch * rng.randint(2, max(2, gi // patch - 1))
            args = (rng.randint(1, 3), gi, li, patch, 64 * H, H, 64 * rng.randint(1, 4), 64 * rng.randint(1, 24), rng.randint(1, 2), rng.randint(1, 4))
            fn = dino_case
        elif kind == "attn":
            args = (rng.randint(1, 256),)
            fn = T.test_attention_fwd_bwd
        else:                                                # a quarter of the cases on the row-pair backward (384 columns, even row count)
            args = (2 * rng.randint(1, 3000), 384) if rng.random() < 0.25 else (rng.randint(1, 2000), 4 * rng.randint(1, 512))
            fn = T._layernorm_case
        cases.append((kind, args))
        try:
            fn(ops, *args)
        except Exception:
            print(f"FAILED {kind}{args}", flush=True)
            traceback.print_exc()
            return 1
        n += 1
        if n % 10 == 0:
            print(f"{n} cases ok ({time.time() - t0:.0f} s); last: {kind}{args}", flush=True)
    print(f"all {n} cases ok: " + " ".join(f"{k}{a}" for k, a in cases[-12:]))
    for kind, (err, name, case) in sorted(WORST.items()):
        print(f"  largest gradient distance of the sweep, {kind}: {err:.4f} ({name}, {kind}{case})")
    return 0


def main():
    mc = os.environ.get("CASES")
    return run(os.environ.get("SEED", "0"), os.environ.get("KINDS") or "nt,nt,tn,attn,ln,nt8,tn8,tnb",
               float(os.environ.get("BUDGET_S", "120")), int(mc) if mc else None)


if __name__ == "__main__":
    sys.exit(main())
