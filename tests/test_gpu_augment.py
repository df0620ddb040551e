"""GPU multi-crop input pipeline through the C ABI against the Pillow-pinned
oracle: uint8 stages bit-exact, final float32 views exact (same float32 summation order as
the oracle), and within 1 LSB of the Pillow / ATen fixtures."""
import numpy as np
import pytest
import torch

from _util import load_golden
from oracle import augment_oracle as A

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


@pytest.fixture(scope="module")
def ops():
    from vitssl_hip import ops as o
    return o


def _images(B, H, W, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:H, 0:W]
    out = []
    for b in range(B):
        base = np.stack([127 + 120 * np.sin(xx / (5.0 + b) + yy / 11.0), 127 + 120 * np.cos(xx / 5.0 - yy / (3.0 + b)),
                         (xx * yy + 17 * b) % 256], -1)
        out.append(np.clip(base + rng.normal(0, 25, base.shape), 0, 255).astype(np.uint8))
    return np.stack(out)


def _spec(size, scale, gray_p):
    from data.multicrop import ViewSpec
    return ViewSpec(size=size, scale=scale, gray_p=gray_p)


@pytest.mark.parametrize("H,W,S,scale", [(96, 96, 224, (0.5, 1.0)), (96, 96, 96, (0.08, 0.4)), (96, 96, 48, (0.08, 0.4)),
                                         (160, 120, 96, (0.3, 1.0))])
def test_stages_bit_exact_against_oracle(ops, H, W, S, scale):
    from data.multicrop import GPUMultiCrop, pack_params, sample_view_params
    B = 6
    imgs = _images(B, H, W, S)
    spec = _spec(S, scale, 0.5)
    gen = torch.Generator().manual_seed(S + H)
    prm = [sample_view_params(spec, H, W, gen) for _ in range(B)]
    ip, fp = pack_params(prm, 7)
    d_img = torch.from_numpy(imgs).to(DEV)
    ip_d, fp_d = torch.from_numpy(ip).to(DEV), torch.from_numpy(fp).to(DEV)
    tmp = torch.empty(B, H, S, 3, dtype=torch.uint8, device=DEV)
    u8 = torch.empty(B, S, S, 3, dtype=torch.uint8, device=DEV)
    ops.aug_resized_crop_u8(d_img, ip_d, tmp, u8)
    ref1 = [A.resized_crop_u8(imgs[b], p["top"], p["left"], p["h"], p["w"], S, S, p["flip"]) for b, p in enumerate(prm)]
    got1 = u8.cpu().numpy()
    for b in range(B):
        assert np.array_equal(got1[b], ref1[b]), ("resized_crop", b, prm[b])
    ops.aug_color_u8(u8, ip_d, fp_d)
    got2 = u8.cpu().numpy()
    ref2 = []
    for b, p in enumerate(prm):
        x = ref1[b]
        for fn in p["order"]:
            x = (A.adjust_brightness(x, p["brightness"]) if fn == 0 else A.adjust_contrast(x, p["contrast"]) if fn == 1
                 else A.adjust_saturation(x, p["saturation"]) if fn == 2 else A.adjust_hue(x, p["hue"]))
        if p["gray"]:
            x = A.to_grayscale3(x)
        ref2.append(x)
        assert np.array_equal(got2[b], x), ("colour chain", b, p)
    out = torch.empty(B, 3, S, S, device=DEV)
    ops.aug_blur_to_tensor(u8, fp_d, out, 7)
    got3 = out.cpu().numpy()
    for b, p in enumerate(prm):
        ref3 = A.to_tensor(A.gaussian_blur_u8(ref2[b], 7, p["sigma"]))
        assert np.array_equal(got3[b], ref3), ("blur+ToTensor", b, p["sigma"])
    # and the assembled path gives the same thing
    mc = GPUMultiCrop(spec, spec, 2, 1)
    assert torch.equal(mc.render(d_img, prm, spec), out)


def test_golden_views_from_pillow(ops):
    """the three whole views of tests/golden/augment.npz (Pillow + ATen): 1 LSB on <= 1e-4 of the pixels"""
    from data.multicrop import GPUMultiCrop, ViewSpec
    g = load_golden("augment")
    img = torch.from_numpy(g["img"][None]).to(DEV)
    for n in range(3):
        p, f = g[f"view{n}_params"], g[f"view{n}_factors"]
        prm = dict(top=int(p[0]), left=int(p[1]), h=int(p[2]), w=int(p[3]), flip=bool(p[4]), order=[int(v) for v in p[5:9]],
                   gray=bool(p[9]), brightness=float(f[0]), contrast=float(f[1]), saturation=float(f[2]), hue=float(f[3]),
                   sigma=float(f[4]))
        spec = ViewSpec(size=int(p[10]), scale=(0.08, 1.0))
        got = GPUMultiCrop(spec, spec, 1, 1).render(img, [prm], spec)[0].cpu().numpy()
        d = np.abs(np.rint(got * 255.0) - np.rint(g[f"view{n}"] * 255.0))
        assert d.max() <= 1 and (d > 0).sum() <= max(1, d.size // 10000), (n, d.max(), int((d > 0).sum()))


def test_view_list_layout_and_statistics(ops):
    """__call__: globals first, shapes per view, values in [0,1]; flips / gray hits happen at
    roughly their configured rates (sampler smoke test, not a parity claim)."""
    from data.multicrop import GPUMultiCrop, sample_view_params
    B = 16
    imgs = torch.from_numpy(_images(B, 96, 96, 1)).to(DEV)
    gs, ls = _spec(224, (0.5, 1.0), 0.2), _spec(96, (0.08, 0.4), 0.0)
    mc = GPUMultiCrop(gs, ls, 6, 2)
    views = mc(imgs, torch.Generator().manual_seed(0))
    assert [tuple(v.shape) for v in views] == [(B, 3, 224, 224)] * 2 + [(B, 3, 96, 96)] * 4
    for v in views:
        assert v.dtype == torch.float32 and float(v.min()) >= 0.0 and float(v.max()) <= 1.0
    from data.multicrop import pack_params, params_as_list, sample_batch_params
    gen = torch.Generator().manual_seed(1)
    arrs = sample_batch_params(gs, 96, 96, 2000, gen)
    ip_a, fp_a = pack_params(arrs, 7)                      # array path == per-view path, bit for bit
    ip_l, fp_l = pack_params(params_as_list(arrs), 7)
    assert np.array_equal(ip_a, ip_l) and np.array_equal(fp_a, fp_l)
    for draws in ([sample_view_params(gs, 96, 96, gen) for _ in range(2000)], params_as_list(arrs)):
        assert 0.45 < np.mean([d["flip"] for d in draws]) < 0.55
        assert 0.16 < np.mean([d["gray"] for d in draws]) < 0.24
        areas = np.array([d["h"] * d["w"] for d in draws]) / (96 * 96)
        assert areas.min() >= 0.45 and areas.max() <= 1.0 + 1e-9
        assert 0.66 < areas.mean() < 0.78                                   # U(0.5, 1) minus the rejected (too wide / tall) boxes
        assert all(0.6 <= d["brightness"] <= 1.4 and 0.8 <= d["saturation"] <= 1.2 and -0.1 <= d["hue"] <= 0.1 for d in draws)
        assert all(0.1 <= d["sigma"] <= 2.0 and sorted(d["order"]) == [0, 1, 2, 3] for d in draws)
        first = np.bincount([d["order"][0] for d in draws], minlength=4) / len(draws)
        assert first.min() > 0.2 and first.max() < 0.3                      # every op leads about a quarter of the time
        assert all(0 <= d["top"] <= 96 - d["h"] and 0 <= d["left"] <= 96 - d["w"] for d in draws)


def test_rejects_unsupported(ops):
    from vitssl_hip import _lib as L
    from data.multicrop import GPUMultiCrop
    spec = _spec(48, (0.5, 1.0), 0.0)
    with pytest.raises(L.VitsslError):
        GPUMultiCrop(spec, spec, 2, 1).render(torch.zeros(1, 96, 96, 3, dtype=torch.uint8), [], spec)       # CPU tensor
    big = torch.zeros(1, 230, 230, 3, dtype=torch.uint8, device=DEV)
    with pytest.raises(L.VitsslError):
        ops.aug_color_u8(big, torch.zeros(1, 11, dtype=torch.int32, device=DEV), torch.zeros(1, 10, device=DEV))
