"""Round-2 parity cases on the GPU:
  * weight-staleness: any torch optimizer / load_state_dict after a forward must be seen by the
    bf16 GEMM operand caches (values checked against the oracle on the NEW weights);
  * a backward through a graph whose saved activations were replaced raises instead of
    returning wrong gradients; gradient accumulation as forward/backward pairs stays exact;
  * frozen parameters + fused train_step still update the trainable ones;
  * ManualPatchEmbedding forward and every gradient vs the reference golden;
  * reference-written checkpoint -> load_weights into the HIP ViT, optimizer state -> FusedAdamW,
    next step equal to the reference's torch.optim.AdamW;
  * a launch on a tensor of another device index is refused."""
import os

import numpy as np
import pytest
import torch

from _util import GOLDEN, load_golden, split_prefix, t, rel_l2, max_abs
from oracle import vit_oracle as O

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def _vit_from_golden():
    from vit_core import ViT
    g = load_golden("vit_tiny")
    B, img, patch, D, H, F, blocks, C = (int(v) for v in g["cfg"])
    model = ViT(C, blocks, (3, img, img), D, patch, H, F, 0.0)
    sd = split_prefix(g, "sd/")
    model.load_state_dict(sd)
    return g, model.to(DEV), sd, (patch, H)


def test_torch_optimizer_and_load_state_dict_refresh_gemm_weights():
    g, model, sd, (patch, H) = _vit_from_golden()
    x = (t(g["x_u8"]).float() / 256.0)
    labels = t(g["labels"])
    model.train()
    logits = model(x.to(DEV))
    assert rel_l2(logits, t(g["logits"])) < 2e-2
    torch.nn.functional.cross_entropy(logits, labels.to(DEV)).backward()
    opt = torch.optim.SGD(model.parameters(), lr=0.5)            # big step: stale GEMM weights would show
    opt.step()
    new_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    moved = rel_l2(new_sd["encoder_blocks.0.feed_forward.linear_in.weight"], sd["encoder_blocks.0.feed_forward.linear_in.weight"])
    assert moved > 1e-3
    with torch.no_grad():
        after = model(x.to(DEV))
    want = O.vit_forward(new_sd, x, patch, H, emu="bf16")
    stale = O.vit_forward({**new_sd, **{k: v for k, v in sd.items() if k.endswith("weight") and v.dim() == 2 and "norm" not in k}},
                          x, patch, H, emu="bf16")                # new LN/bias/pos, OLD matrices: the bug's signature
    assert rel_l2(after, want) < 1e-2
    assert rel_l2(after, stale) > 5 * rel_l2(after, want)
    # load_state_dict after a forward: back to the golden weights
    model.load_state_dict(sd)
    with torch.no_grad():
        back = model(x.to(DEV))
    assert rel_l2(back, t(g["logits"])) < 2e-2
    # in-place edit of one parameter through torch
    with torch.no_grad():
        model.classification_head.linear.weight.mul_(0.0)
        model.encoder_blocks[0].self_attention.w_query.weight.mul_(2.0)
        z = model(x.to(DEV))
    sd2 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    assert rel_l2(z, O.vit_forward(sd2, x, patch, H, emu="bf16")) < 1e-2


def test_simmim_sgd_step_matches_oracle_on_new_weights():
    from vit_core.ssl.simmim import SimMIMViT
    g = load_golden("simmim_tiny")
    B, img, patch, D, H, F, blocks = (int(v) for v in g["cfg"])
    model = SimMIMViT(blocks, (3, img, img), D, patch, H, F, 0.0, float(g["ratio"]))
    model.load_state_dict(split_prefix(g, "sd/"))
    model = model.to(DEV).train()
    x = t(g["x_u8"]).float() / 256.0
    torch.manual_seed(int(g["mask_seed"]))
    pred, tgt = model(x.to(DEV))
    torch.nn.functional.l1_loss(pred, tgt).backward()
    torch.optim.SGD(model.parameters(), lr=3.0).step()
    new_sd = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    torch.manual_seed(int(g["mask_seed"]))
    with torch.no_grad():
        pred2, _ = model(x.to(DEV))
    pe, _ = O.simmim_forward(new_sd, x, t(g["mask"]), patch, H, emu="bf16")
    assert rel_l2(pred2, pe) < 1e-2
    assert rel_l2(pred2, pred) > 5 * rel_l2(pred2, pe)                          # and the step did change the output


def test_stale_graph_backward_is_refused_and_accumulation_pairs_work():
    from vitssl_hip import VitsslError
    g, model, sd, (patch, H) = _vit_from_golden()
    x = (t(g["x_u8"]).float() / 256.0).to(DEV)
    labels = t(g["labels"]).to(DEV)
    model.train()
    l1 = torch.nn.functional.cross_entropy(model(x[:2]), labels[:2])
    l2 = torch.nn.functional.cross_entropy(model(x[2:]), labels[2:])            # replaces the saved activations of l1
    with pytest.raises(VitsslError, match="saved activations"):
        (l1 + l2).backward()
    # forward/backward pairs accumulate exactly like one batch
    model.zero_grad(set_to_none=True)
    for sl in (slice(0, 2), slice(2, 4)):
        (0.5 * torch.nn.functional.cross_entropy(model(x[sl]), labels[sl])).backward()
    acc = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(model(x), labels).backward()
    for k, p in model.named_parameters():
        assert rel_l2(acc[k], p.grad) < 2e-2, k
    # an extra forward under no_grad (other data) between loss and backward changes nothing
    full = {k: p.grad.clone() for k, p in model.named_parameters()}
    model.zero_grad(set_to_none=True)
    out = model(x)
    with torch.no_grad():
        model(torch.rand_like(x))
    torch.nn.functional.cross_entropy(out, labels).backward()
    for k, p in model.named_parameters():
        assert rel_l2(p.grad, full[k]) < 1e-5, k          # (bias / LN sums use fp32 atomics: order-dependent last bits)
    # stand-alone encoder stack: slot reuse after wrap-around is detected
    from vit_core import EncoderBlock
    blk = EncoderBlock(128, 2, 192, 0.0).to(DEV)
    xs = torch.randn(2, 5, 128, device=DEV, requires_grad=True)
    first, _ = blk(xs)
    for _ in range(4):
        blk(xs)
    with pytest.raises(VitsslError, match="saved activations"):
        first.sum().backward()


def test_fused_step_with_frozen_parameters_updates_the_rest():
    from vit_core.ssl.simmim import SimMIMViT
    from vitssl_hip.optim import FusedAdamW
    torch.manual_seed(3)
    model = SimMIMViT(2, (3, 32, 32), 128, 8, 2, 192, 0.0, 0.6).to(DEV).train()
    for p in model.encoder_blocks[0].parameters():
        p.requires_grad = False
    before = {k: v.detach().clone() for k, v in model.state_dict().items()}
    opt = FusedAdamW(model.flat_store(), lr=1e-2, weight_decay=0.0)
    x = torch.rand(4, 3, 32, 32, device=DEV)
    model.train_step(x, opt)
    after = model.state_dict()
    for k in before:
        changed = not torch.equal(before[k], after[k])
        assert changed == (not k.startswith("encoder_blocks.0.")), k


def test_manual_patch_embedding_grads_match_reference_golden():
    from vit_core import ManualPatchEmbedding
    g = load_golden("manual_embed")
    B, img, patch, D = (int(v) for v in g["cfg"])
    m = ManualPatchEmbedding((3, img, img), D, patch)
    m.load_state_dict(split_prefix(g, "sd/"))
    m = m.to(DEV)
    x = (t(g["x_u8"]).float() / 256.0).to(DEV)
    y = m(x)
    assert y.shape == (B, (img // patch) ** 2 + 1, D) and rel_l2(y, t(g["y"])) < 1e-2
    (y * t(g["w"]).to(DEV)).sum().backward()
    ref = split_prefix(g, "grad/")
    assert set(ref) == {k for k, _ in m.named_parameters()}
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, ref[k]) < 2e-2, (k, rel_l2(p.grad, ref[k]))
    # and against the oracle with the same rounding points
    leaves = {k: v.clone().requires_grad_(True) for k, v in split_prefix(g, "sd/").items()}
    ry = O.conv_patch_embed(x.cpu(), leaves["linear.weight"], leaves["linear.bias"], leaves["cls_token"],
                            leaves["positional_embedding"], patch, emu="bf16")
    assert rel_l2(y, ry) < 5e-3
    (ry * t(g["w"])).sum().backward()
    for k, p in m.named_parameters():
        assert rel_l2(p.grad, leaves[k].grad) < 1e-2, k


def test_reference_checkpoint_roundtrip_on_gpu():
    """f-2: checkpoint written by the reference (`_orig_mod.` keys) -> load_weights into the HIP
    ViT on the GPU; its optimizer state -> FusedAdamW; the next AdamW step equals the reference's."""
    from utils.model_builder import load_weights, strip_compile_prefix
    from vit_core import ViT
    from vit_core.ssl.simmim import SimMIMViT
    from vitssl_hip.optim import FusedAdamW
    g = load_golden("ckpt_ref_expected")
    B, img, patch, D, H, F, blocks, C = (int(v) for v in g["cfg"])
    path = os.path.join(GOLDEN, "ckpt_ref_simmim.pth")
    vit = ViT(C, blocks, (3, img, img), D, patch, H, F, 0.0)
    vit.load_state_dict(split_prefix(g, "vit_init/"))
    vit = vit.to(DEV)
    vit(torch.rand(2, 3, img, img, device=DEV))                 # materialise the flat store and the bf16 caches first
    load_weights(vit, path)
    for k, v in split_prefix(g, "vit_loaded/").items():
        assert torch.equal(vit.state_dict()[k].cpu(), v), k
    # the loaded weights are the ones the kernels use
    x = torch.rand(2, 3, img, img)
    with torch.no_grad():
        got = vit(x.to(DEV))
    want = O.vit_forward(split_prefix(g, "vit_loaded/"), x, patch, H, emu="bf16")
    assert rel_l2(got, want) < 1e-2
    # resume: model + optimizer state, then the reference's 4th step
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sim = SimMIMViT(blocks, (3, img, img), D, patch, H, F, 0.0, 0.6)
    sim.load_state_dict(strip_compile_prefix(ckpt["model_state_dict"]))
    sim = sim.to(DEV)
    lr, wd = (float(v) for v in g["lr_wd"])
    opt = FusedAdamW(sim.flat_store(), lr=lr, weight_decay=wd)
    opt.load_state_dict(ckpt["optimizer_state_dict"])
    assert opt.step_count == 3
    ref_state = ckpt["optimizer_state_dict"]["state"]
    names = [k for k, _ in sim.named_parameters()]
    mine = opt.state_dict()["state"]
    for i, k in enumerate(names):
        assert max_abs(mine[i]["exp_avg"], ref_state[i]["exp_avg"]) == 0.0, k
        assert max_abs(mine[i]["exp_avg_sq"], ref_state[i]["exp_avg_sq"]) == 0.0, k
    grads = split_prefix(g, "step4_grad/")
    for k, p in sim.named_parameters():
        p.grad = grads[k].to(DEV)
    opt.step()
    for k, v in split_prefix(g, "step4_param/").items():
        assert max_abs(sim.state_dict()[k], v) < 2e-6, (k, max_abs(sim.state_dict()[k], v))


def test_wrong_device_index_is_refused(monkeypatch):
    """A tensor of another GPU than the current one must never reach a kernel (launches go to the CURRENT device's
    stream).  With two visible GPUs the real situation is built; on a one-GPU box the process's current device is
    made to read as cuda:1 while the tensors live on cuda:0 -- the guard in ops._chk sees exactly the mismatch it
    exists for, and no launch happens."""
    from vitssl_hip import ops, VitsslError
    if torch.cuda.device_count() >= 2:
        with torch.cuda.device(0):
            x = torch.zeros(64, 64, device="cuda:1")
            y = torch.empty(64, 64, dtype=torch.bfloat16, device="cuda:1")
            with pytest.raises(VitsslError, match="current device"):
                ops.cast_bf16(x, y)
        return
    x = torch.ones(64, 64, device=DEV)
    y = torch.full((64, 64), 7.0, dtype=torch.bfloat16, device=DEV)
    monkeypatch.setattr(torch.cuda, "current_device", lambda: 1)
    with pytest.raises(VitsslError, match="current device is cuda:1"):
        ops.cast_bf16(x, y)
    monkeypatch.undo()
    assert float(y.float().min()) == 7.0          # nothing was launched
    ops.cast_bf16(x, y)                           # and the same call goes through once the devices agree
    assert float(y.float().max()) == 1.0


def test_reserved_cus_knob_keeps_results():
    """VITSSL_RESERVE_CUS shrinks the persistent GEMM grids (CUs left to the collective library under data
    parallelism); it is read once per process, so the check runs in a child process."""
    import os
    import subprocess
    import sys
    code = r"""
import sys, torch
sys.path.insert(0, sys.argv[1])
from vitssl_hip import _lib as L, ops
dev = torch.device("cuda:0")
torch.manual_seed(0)
M, N, K = 3000, 768, 512
A = torch.randn(M, K, device=dev).to(torch.bfloat16)
B = (torch.randn(N, K, device=dev) * 0.05).to(torch.bfloat16)
out = torch.empty(M, N, device=dev)
ops.gemm_nt(A, B, out, L.EPI_F32)
ref = A.float() @ B.float().t()
assert float((out - ref).norm() / ref.norm()) < 1e-5
dY = torch.randn(M, 256, device=dev).to(torch.bfloat16)
C = torch.zeros(256, K, device=dev)
ops.gemm_tn(dY, A, C)
ref = dY.float().t() @ A.float()
assert float((C - ref).norm() / ref.norm()) < 1e-5
print("ok")
"""
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vit-ssl_amd")
    env = dict(os.environ, VITSSL_RESERVE_CUS="24")
    r = subprocess.run([sys.executable, "-c", code, pkg], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
