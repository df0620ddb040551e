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


def test_mirror_packages_do_not_hide_reference_modules(tmp_path):
    """`vit-ssl_amd/` in front of a reference checkout on PYTHONPATH:
    utils.model_builder / utils.trainers / vit_core come from this package, while the modules it
    does not replace (utils.schemas, utils.logger, data.data_builder -> .datasets: what the
    reference's train.py:7-11 imports) still resolve to the checkout behind it."""
    import subprocess
    import sys
    from conftest import PKG
    ref = tmp_path / "fake_reference"
    (ref / "utils" / "schemas" / "training_schemas").mkdir(parents=True)
    (ref / "data").mkdir()
    (ref / "utils" / "__init__.py").write_text("raise ImportError('the reference utils/__init__ must not run')\n")
    (ref / "utils" / "model_builder.py").write_text("WHO = 'reference'\n")
    (ref / "utils" / "logger.py").write_text("class Logger:\n    WHO = 'reference'\n")
    (ref / "utils" / "schemas" / "__init__.py").write_text("")
    (ref / "utils" / "schemas" / "training_schemas" / "__init__.py").write_text("class TrainConfig:\n    WHO = 'reference'\n")
    (ref / "data" / "__init__.py").write_text("raise ImportError('the reference data/__init__ must not run')\n")
    (ref / "data" / "datasets.py").write_text("class STL10Dataset:\n    WHO = 'reference'\n")
    (ref / "data" / "data_builder.py").write_text("from .datasets import STL10Dataset\n\ndef prepare_dataloaders():\n    return STL10Dataset.WHO\n")
    code = (
        "from utils.model_builder import build_model\n"
        "import utils.model_builder as mb, utils.trainers as tr, vit_core\n"
        "from data.data_builder import prepare_dataloaders\n"
        "from utils.schemas.training_schemas import TrainConfig\n"
        "from utils.logger import Logger\n"
        "from data import GPUMultiCrop\n"
        "assert not hasattr(mb, 'WHO') and hasattr(tr, 'SimMIMTrainer')\n"
        "assert prepare_dataloaders() == 'reference' and TrainConfig.WHO == 'reference' and Logger.WHO == 'reference'\n"
        "print('ok', mb.__file__)\n")
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([PKG, str(ref)]))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.startswith("ok") and PKG in r.stdout


def test_dino_trainer_schedules_match_reference_trainer():
    """Keys, horizon and evaluation point of the two DINO schedules as the reference trainer has
    them (utils/trainers/dino_trainer.py:16-29,46,80) on the shipped configs/dino/training.yaml."""
    from utils.trainers.dino_trainer import DINOTrainer
    g = load_golden("dino_trainer_sched")
    n = int(g["num_epochs"])
    shipped = {"training": {"type": "dino", "student_temp": 0.1, "teacher_temp": 0.04, "teacher_temp_final": 0.07,
                            "teacher_temp_scheduler": "cosine", "teacher_momentum_start": 0.996, "teacher_momentum_final": 1,
                            "num_epochs": n}}
    temp, mom = DINOTrainer.build_schedules(shipped, n)
    assert DINOTrainer.teacher_temp0(shipped) == 0.04
    for i, e in enumerate(g["epochs"]):
        assert abs(temp.get_temp(int(e)) - g["temp_cos"][i]) < 1e-12
        assert abs(mom.get_momentum(int(e)) - g["mom"][i]) < 1e-12
    assert temp.get_temp(31) < 0.05 and abs(temp.get_temp(n) - 0.07) < 1e-12      # reaches 0.07 at epoch 100, not 31
    lin = dict(shipped["training"], teacher_temp_scheduler="linear")
    temp, _ = DINOTrainer.build_schedules({"training": lin}, n)
    for i, e in enumerate(g["epochs"]):
        assert abs(temp.get_temp(int(e)) - g["temp_lin"][i]) < 1e-12
    const = {k: v for k, v in shipped["training"].items() if k != "teacher_temp_final"}
    const["teacher_temp"] = 0.05                                                    # no final -> constant teacher_temp
    temp, _ = DINOTrainer.build_schedules({"training": const}, n)
    for i, e in enumerate(g["epochs"]):
        assert abs(temp.get_temp(int(e)) - g["temp_const"][i]) < 1e-12


def test_reference_written_checkpoint_loads_like_the_reference(tmp_path):
    """tests/golden/ckpt_ref_simmim.pth was written by the REFERENCE (torch.compile wrapper ->
    `_orig_mod.` keys, base_trainer.py:99-105 layout).  load_weights must put into a fine-tuning
    ViT exactly what the reference's own load_weights puts there from the same tensors
    (utils/model_builder.py:39-72), and the SimMIM model must take the state dict whole."""
    from _util import GOLDEN
    from utils.model_builder import load_weights, strip_compile_prefix
    from vit_core import ViT
    from vit_core.ssl.simmim import SimMIMViT
    g = load_golden("ckpt_ref_expected")
    B, img, patch, D, H, F, blocks, C = (int(v) for v in g["cfg"])
    path = os.path.join(GOLDEN, "ckpt_ref_simmim.pth")
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ckpt) == {"epoch", "model_state_dict", "optimizer_state_dict", "best_val_loss", "config"}
    assert all(k.startswith("_orig_mod.") for k in ckpt["model_state_dict"])
    vit = ViT(C, blocks, (3, img, img), D, patch, H, F, 0.0)
    vit.load_state_dict(split_prefix(g, "vit_init/"))
    load_weights(vit, path)
    want = split_prefix(g, "vit_loaded/")
    assert set(vit.state_dict()) == set(want)
    for k, v in want.items():
        assert torch.equal(vit.state_dict()[k], v), k
    sim = SimMIMViT(blocks, (3, img, img), D, patch, H, F, 0.0, 0.6)
    sim.load_state_dict(strip_compile_prefix(ckpt["model_state_dict"]), strict=True)
    for k, v in split_prefix(g, "simmim/").items():
        assert torch.equal(sim.state_dict()[k], v), k


@pytest.mark.skipif(not os.path.isdir("/root/reference/vit_core"), reason="needs the reference checkout (build container only)")
def test_checkpoint_written_here_loads_into_the_reference(tmp_path):
    """The other direction: a trainer checkpoint of THIS package (same dict layout) is taken by
    the reference modules' load_state_dict(strict=True).  Runs the reference in a subprocess."""
    import subprocess
    import sys
    from vit_core.ssl.simmim import SimMIMViT
    torch.manual_seed(5)
    sim = SimMIMViT(2, (3, 32, 32), 64, 8, 2, 128, 0.1, 0.6)
    path = os.path.join(tmp_path, "last_model.pth")
    torch.save({"epoch": 1, "model_state_dict": sim.state_dict(), "optimizer_state_dict": {}, "config": {}}, path)
    code = (
        "import sys, torch\n"
        "sys.path.insert(0, '/root/reference')\n"
        "from vit_core.ssl.simmim.model import SimMIMViT\n"
        "m = SimMIMViT(num_blocks=2, input_shape=(3, 32, 32), embed_dim=64, patch_size=8, num_heads=2, mlp_dim=128, dropout=0.1, mask_ratio=0.6)\n"
        f"ck = torch.load({path!r}, map_location='cpu', weights_only=False)\n"
        "r = m.load_state_dict(ck['model_state_dict'], strict=True)\n"
        "assert not r.missing_keys and not r.unexpected_keys\n"
        "print('ok', float(sum(p.double().sum() for p in m.parameters())))\n")
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    total = float(sum(p.double().sum() for p in sim.parameters()))
    assert abs(float(r.stdout.split()[1]) - total) < 1e-9 * max(1.0, abs(total))


def test_weight_cache_is_hit_until_a_parameter_changes(monkeypatch):
    """FlatStore.refresh_weights(): two calls with no parameter change in between launch ONE cast pass; an in-place
    update through a Parameter, a load_state_dict or mark_dirty() (raw-pointer writers: HIP AdamW / EMA) triggers the
    next one.  (Round-2 advisor finding: the cache key was overwritten by a loop variable, so every forward re-cast.)"""
    from vitssl_hip import engine, ops
    runs = []

    class FakePlan:
        def run(self, jobs):
            runs.append(len(jobs))

    monkeypatch.setattr(ops, "CastPlan", FakePlan)
    lin = torch.nn.Sequential(torch.nn.Linear(64, 128), torch.nn.Linear(128, 64))
    st = engine.FlatStore(lin, torch.device("cpu"))
    st.register_weight("w0", lambda: st.view("0.weight", (128, 64)))
    st.register_weight("w1", lambda: st.view("1.weight", (64, 128)), transposed_too=False)
    st.refresh_weights()
    st.refresh_weights()
    assert runs == [2]
    assert st.w("w0").shape == (128, 64) and st.w("w0.T").shape == (64, 128) and st.w("w1").shape == (64, 128)
    with torch.no_grad():
        lin[0].weight.mul_(0.5)
    st.refresh_weights()
    st.refresh_weights()
    assert runs == [2, 2]
    st.mark_dirty()
    st.refresh_weights()
    assert runs == [2, 2, 2]
    lin.load_state_dict({k: v.clone() for k, v in lin.state_dict().items()})
    st.refresh_weights()
    st.refresh_weights()
    assert runs == [2, 2, 2, 2]
    torch.optim.SGD(lin.parameters(), lr=0.1)   # building an optimizer touches nothing
    st.refresh_weights()
    assert runs == [2, 2, 2, 2]


_DINO_TRANSFORMS = {"transforms": {
    "globals": [{"name": "RandomResizedCrop", "params": {"size": 224, "scale": [0.5, 1.0]}},
                {"name": "RandomHorizontalFlip", "params": {"p": 0.5}},
                {"name": "ColorJitter", "params": {"brightness": 0.4, "contrast": 0.4, "saturation": 0.2, "hue": 0.1}},
                {"name": "GaussianBlur", "params": {"kernel_size": 7, "sigma": [0.1, 2.0]}},
                {"name": "ToTensor"}],
    "train": [{"name": "Resize", "params": {"size": 96}}, {"name": "ToTensor", "params": None}]}}


def test_get_transforms_returns_reference_style_callables(monkeypatch):
    """utils.train_utils.get_transforms keeps the reference's contract (utils/train_utils.py:54-68: one
    torchvision Compose per list, built with getattr(T, name)(**params)) -- the un-mirrored reference datasets call
    `self.transform(image)` / `self.transforms["globals"](image)` (data/datasets.py) -- and each callable also carries
    the GPU multi-crop recipe.  torchvision is not installed in the build image: a stand-in module records the calls."""
    import sys
    import types
    built = []

    class _Op:
        def __init__(self, **kw):
            self.kw = kw
            built.append((type(self).__name__, kw))

        def __call__(self, img):
            return img + [type(self).__name__]

    class Compose:
        def __init__(self, ops):
            self.transforms = ops

        def __call__(self, img):
            for op in self.transforms:
                img = op(img)
            return img

    T = types.ModuleType("torchvision.transforms")
    T.Compose = Compose
    for name in ("RandomResizedCrop", "RandomHorizontalFlip", "ColorJitter", "GaussianBlur", "ToTensor", "Resize"):
        setattr(T, name, type(name, (_Op,), {}))
    tv = types.ModuleType("torchvision")
    tv.transforms = T
    monkeypatch.setitem(sys.modules, "torchvision", tv)
    monkeypatch.setitem(sys.modules, "torchvision.transforms", T)
    from utils.train_utils import get_transforms
    tf = get_transforms(_DINO_TRANSFORMS)
    assert set(tf) == {"globals", "train"}
    assert tf["train"]([]) == ["Resize", "ToTensor"]                      # what a reference dataset does per image
    assert tf["globals"]([])[0] == "RandomResizedCrop" and len(tf["globals"]([])) == 5
    assert ("Resize", {"size": 96}) in built and ("ToTensor", {}) in built
    assert ("ColorJitter", {"brightness": 0.4, "contrast": 0.4, "saturation": 0.2, "hue": 0.1}) in built
    spec = tf["globals"].view_spec                                          # the GPU route's recipe rides along
    assert spec.size == 224 and tuple(spec.scale) == (0.5, 1.0) and spec.blur_kernel == 7
    assert tf["train"].view_spec is None                                    # not a multi-crop recipe


def test_get_transforms_without_torchvision_names_the_gpu_route(monkeypatch):
    import sys
    from utils.train_utils import get_transforms
    from vitssl_hip import VitsslError
    monkeypatch.setitem(sys.modules, "torchvision", None)                  # import torchvision -> ImportError
    tf = get_transforms(_DINO_TRANSFORMS)
    assert tf["globals"].view_spec.size == 224
    with pytest.raises(VitsslError, match="INTEGRATION.md section 4"):
        tf["globals"](object())
    with pytest.raises(VitsslError, match="torchvision is not installed"):
        tf["train"](object())


def test_setup_device_binds_local_rank(monkeypatch):
    """One process per GPU: setup_device() binds cuda:LOCAL_RANK (the reference returns bare "cuda",
    utils/train_utils.py:12-16, which would put every rank on GPU 0)."""
    from utils import train_utils
    bound = []
    monkeypatch.setattr(torch.cuda, "is_available", lambda: True)
    monkeypatch.setattr(torch.cuda, "device_count", lambda: 8)
    monkeypatch.setattr(torch.cuda, "set_device", lambda i: bound.append(i))
    monkeypatch.setenv("LOCAL_RANK", "5")
    assert train_utils.setup_device() == torch.device("cuda:5") and bound == [5]
    monkeypatch.delenv("LOCAL_RANK")
    assert train_utils.setup_device() == torch.device("cuda:0") and bound == [5, 0]
    monkeypatch.setenv("LOCAL_RANK", "9")
    with pytest.raises(RuntimeError, match="LOCAL_RANK=9"):
        train_utils.setup_device()
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    with pytest.raises(RuntimeError, match="no GPU visible"):
        train_utils.setup_device()


def test_configure_collectives_caps_channels_at_the_reserve(monkeypatch):
    """RCCL runs one workgroup per channel; the persistent GEMM grids leave `reserve` CUs free, so the channel count is
    capped there unless the user chose otherwise (engine.configure_collectives, called before init_process_group)."""
    from vitssl_hip import engine
    monkeypatch.delenv("NCCL_MAX_NCHANNELS", raising=False)
    monkeypatch.delenv("NCCL_MIN_NCHANNELS", raising=False)
    assert engine.configure_collectives() == {"NCCL_MAX_NCHANNELS": "8", "NCCL_MIN_NCHANNELS": "8"}
    monkeypatch.setenv("NCCL_MAX_NCHANNELS", "32")
    monkeypatch.delenv("NCCL_MIN_NCHANNELS")
    assert engine.configure_collectives(16) == {"NCCL_MAX_NCHANNELS": "32", "NCCL_MIN_NCHANNELS": "16"}   # the user's choice wins


def test_bench_quotes_pmc_traffic_only_for_these_kernel_sources(tmp_path):
    """roofline.traffic comes from a committed rocprofv3 PMC summary; a summary taken from OTHER kernel sources must not
    label this build (the lexicographically last file was used whatever it described)."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    sha = bench.kernel_sources_sha16()
    assert len(sha) == 16 and sha == bench.kernel_sources_sha16()
    fam = {"gemm_nt": {"sum_kb": 1000.0, "dispatches": 10}}
    (tmp_path / "r01_final_pmc_traffic.json").write_text(json.dumps({"FETCH_SIZE": fam, "WRITE_SIZE": fam}))      # no stamp at all
    (tmp_path / "r09_final_pmc_traffic.json").write_text(json.dumps({"FETCH_SIZE": fam, "WRITE_SIZE": fam, "kernel_sources_sha16": "0" * 16}))
    val, why = bench.pmc_traffic_per_launch("gemm_nt", str(tmp_path))
    assert val is None and "r09_final_pmc_traffic.json" in why and sha in why
    (tmp_path / "r05_final_pmc_traffic.json").write_text(json.dumps({"FETCH_SIZE": fam, "WRITE_SIZE": fam, "kernel_sources_sha16": sha}))
    val, src = bench.pmc_traffic_per_launch("gemm_nt", str(tmp_path))
    assert src == "r05_final_pmc_traffic.json" and val == (2 * 100.0 + 100.0) * 1024.0
    assert bench.pmc_traffic_per_launch("gemm_nt", str(tmp_path / "nothing")) == (None, None)
    # the clock held over the family's launches: same stamping rule; absent pass -> None
    assert bench.pmc_clock_ghz("gemm_nt", str(tmp_path)) is None
    busy = {"gemm_nt": {"sum": 32 * 1.9 * 5000.0, "duration_ns": 5000.0, "dispatches": 3}}
    (tmp_path / "r09_final_pmc_traffic.json").write_text(json.dumps({"SQ_BUSY_CYCLES": {"gemm_nt": {"sum": 1.0, "duration_ns": 1.0}},
                                                                     "kernel_sources_sha16": "0" * 16}))
    (tmp_path / "r05_final_pmc_traffic.json").write_text(json.dumps({"FETCH_SIZE": fam, "WRITE_SIZE": fam, "SQ_BUSY_CYCLES": busy,
                                                                     "kernel_sources_sha16": sha}))
    assert abs(bench.pmc_clock_ghz("gemm_nt", str(tmp_path)) - 1.9) < 1e-9
    assert bench.pmc_clock_ghz("attn", str(tmp_path)) is None
