"""The NumPy restatement of the DINO multi-crop image arithmetic (oracle/augment_oracle.py)
against fixtures produced by Pillow / ATen themselves (tests/golden/make_augment_golden.py):
uint8 results must be bit-exact, the final ToTensor view exact in float32."""
import numpy as np
import pytest

from _util import load_golden
from oracle import augment_oracle as A


@pytest.fixture(scope="module")
def g():
    return load_golden("augment")


def test_resized_crop_and_flip(g):
    img = g["img"]
    for n, (t, l, h, w, S) in enumerate(g["crops"]):
        assert np.array_equal(A.resized_crop_u8(img, t, l, h, w, S, S), g[f"crop{n}"]), (n, t, l, h, w, S)
    assert np.array_equal(A.resized_crop_u8(img, 0, 0, 96, 96, 96, 96, flip=True), g["flip"])


def test_colour_ops(g):
    img = g["img"]
    assert np.array_equal(A.rgb_to_l(img), g["L"])
    assert np.array_equal(A.to_grayscale3(img), g["gray3"])
    for n, f in enumerate(g["bc_factors"]):
        assert np.array_equal(A.adjust_brightness(img, float(f)), g[f"brightness{n}"]), f
        assert np.array_equal(A.adjust_contrast(img, float(f)), g[f"contrast{n}"]), f
    for n, f in enumerate(g["sat_factors"]):
        assert np.array_equal(A.adjust_saturation(img, float(f)), g[f"saturation{n}"]), f
    for n, f in enumerate(g["hue_factors"]):
        assert np.array_equal(A.adjust_hue(img, float(f)), g[f"hue{n}"]), f


def test_hsv_round_trip_tables(g):
    assert np.array_equal(A.rgb2hsv_u8(g["rnd"]), g["rnd_hsv"])
    assert np.array_equal(A.hsv2rgb_u8(g["rnd"]), g["rnd_as_hsv_to_rgb"])


def _blur_close(got_u8, ref_u8):
    """The blur is a float32 sum of 49 products; ATen's conv2d and a plain loop add them in
    different orders, so a value that lands within rounding error of x.5 may round the other
    way: allow 1 LSB on at most 1 pixel in 10 000 (observed: 2 of 150 528)."""
    d = np.abs(got_u8.astype(np.int32) - ref_u8.astype(np.int32))
    return d.max() <= 1 and (d > 0).sum() <= max(1, d.size // 10000)


def test_gaussian_blur(g):
    for n, s in enumerate(g["sigmas"]):
        assert _blur_close(A.gaussian_blur_u8(g["img"], 7, float(s)), g[f"blur{n}"]), s


def test_whole_views(g):
    for n in range(3):
        p = g[f"view{n}_params"]
        f = g[f"view{n}_factors"]
        prm = dict(top=int(p[0]), left=int(p[1]), h=int(p[2]), w=int(p[3]), flip=bool(p[4]), order=[int(v) for v in p[5:9]],
                   gray=bool(p[9]), brightness=float(f[0]), contrast=float(f[1]), saturation=float(f[2]), hue=float(f[3]),
                   sigma=float(f[4]))
        got = A.apply_view(g["img"], prm, int(p[10]))
        assert got.dtype == np.float32
        assert _blur_close(np.rint(got * 255.0), np.rint(g[f"view{n}"] * 255.0)), n
        same = np.rint(got * 255.0) == np.rint(g[f"view{n}"] * 255.0)
        assert np.array_equal(got[same], g[f"view{n}"][same])          # ToTensor itself is exact


def test_hsv_against_pillow_dense():
    """Every 3rd of all 2^24 RGB / HSV triples straight against Pillow (the full sweep was run
    once: 0 mismatches in either direction)."""
    Image = pytest.importorskip("PIL.Image")
    v = np.arange(0, 1 << 24, 3, dtype=np.uint32)
    n = (v.size // 1024) * 1024
    tri = np.stack([(v >> 16) & 255, (v >> 8) & 255, v & 255], -1).astype(np.uint8)[:n].reshape(-1, 1024, 3)
    assert np.array_equal(A.rgb2hsv_u8(tri), np.array(Image.fromarray(tri).convert("HSV")))
    assert np.array_equal(A.hsv2rgb_u8(tri), np.array(Image.fromarray(tri, "HSV").convert("RGB")))
