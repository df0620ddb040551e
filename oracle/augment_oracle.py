"""CPU restatement (NumPy) of the per-view image arithmetic of the reference's DINO multi-crop
input pipeline -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench cpu leg); the product path
never imports this module.

Reference path: data/datasets.py:80-123 (STL10DINODataset._get_dino_views) applies, per view,
the torchvision transform list of configs/dino/globals.yaml / locals.yaml, built by
utils/train_utils.py:54-68 (`getattr(torchvision.transforms, name)(**params)`) on PIL images:
RandomResizedCrop -> RandomHorizontalFlip -> ColorJitter -> [RandomGrayscale] -> GaussianBlur
-> ToTensor.

Third-party arithmetic.  torchvision (unpinned in requirements.txt, absent from this
container) only orchestrates; on PIL inputs its functional ops call Pillow (present here,
12.2.0), which is where the arithmetic lives:
  resized_crop   = img.crop(box).resize(size, BILINEAR)        -> Pillow Resample.c (8bpc path)
  hflip          = img.transpose(FLIP_LEFT_RIGHT)
  brightness / contrast / saturation = ImageEnhance.*(img).enhance(f) = Image.blend(deg, img, f)
  hue            = HSV split, uint8 wrap-around add on H, merge, convert back (Convert.c)
  grayscale      = img.convert("L") replicated to 3 channels
  gaussian_blur  = pil_to_tensor -> float32 conv2d(reflect pad) -> round -> uint8 (torchvision
                   _functional_tensor.gaussian_blur), kernel = outer(k1d, k1d), k1d = normalised
                   exp(-0.5 (x/sigma)^2) on linspace(-(k-1)/2, (k-1)/2, k)
  to_tensor      = uint8 / 255 -> float32 [C,H,W]
Parity status: every function below is pinned bit-exactly against Pillow itself run in this
container (tests/golden/make_augment_golden.py writes the fixtures, tests/test_augment_oracle.py
checks them); the blur is pinned against the same float32 convolution evaluated with
torch.nn.functional.conv2d (summation order differs: 1 LSB on <= 1 pixel in 10 000).  The RANDOM PARAMETER SAMPLING of the torchvision classes
(`get_params`) is restated from torchvision's published source in vit-ssl_amd/data/multicrop.py
and is "parity unpinned" (no torchvision here to draw from).

All images are uint8 arrays [H, W, 3] (PIL's RGB layout) unless noted.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2      # Pillow Resample.c


# ----------------------------------------------------------------------------- resize
def _bilinear(x):
    x = abs(x)
    return 1.0 - x if x < 1.0 else 0.0


def resample_coeffs(in_size, out_size, in0=0.0, in1=None):
    """Pillow precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter (support 1).
    Returns (xmin[out], count[out], kk[out, ksize] int32)."""
    if in1 is None:
        in1 = float(in_size)
    scale = (in1 - in0) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    xmins = np.zeros(out_size, np.int32)
    counts = np.zeros(out_size, np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bilinear((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = sum(w)
        for x in range(xmax):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmins[xx], counts[xx] = xmin, xmax
    return xmins, counts, kk


def _resample_axis(img, out_size, axis):
    """one 8bpc pass along `axis` (0 = vertical, 1 = horizontal) of an [H, W, C] uint8 image"""
    in_size = img.shape[axis]
    xmins, counts, kk = resample_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for xx in range(out_size):
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(counts[xx]):
            acc += src[xmins[xx] + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img, out_h, out_w):
    """Image.resize((out_w, out_h), BILINEAR) of an RGB image: horizontal pass first (uint8
    intermediate), then vertical; a pass is skipped when that size does not change."""
    if img.shape[1] != out_w:
        img = _resample_axis(img, out_w, 1)
    if img.shape[0] != out_h:
        img = _resample_axis(img, out_h, 0)
    return img


def resized_crop_u8(img, top, left, h, w, out_h, out_w, flip=False):
    """torchvision F.resized_crop (PIL path) followed by the optional horizontal flip."""
    out = resize_bilinear_u8(np.ascontiguousarray(img[top:top + h, left:left + w]), out_h, out_w)
    return out[:, ::-1] if flip else out


# ----------------------------------------------------------------------------- colour
def rgb_to_l(img):
    """Pillow L24 macro: (R*19595 + G*38470 + B*7471 + 0x8000) >> 16"""
    x = img.astype(np.int64)
    return ((x[..., 0] * 19595 + x[..., 1] * 38470 + x[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend_u8(deg, img, factor):
    """Pillow ImagingBlend(deg, img, factor): float32 arithmetic, truncation toward zero;
    clipped when extrapolating (factor outside [0, 1])."""
    a = deg.astype(np.float32)
    b = img.astype(np.float32)
    t = a + np.float32(factor) * (b - a)
    if 0.0 <= factor <= 1.0:
        return t.astype(np.uint8)
    return np.where(t <= 0.0, 0, np.where(t >= 255.0, 255, t)).astype(np.uint8)


def adjust_brightness(img, f):
    return blend_u8(np.zeros_like(img), img, f)


def contrast_mean(img):
    """int(ImageStat.Stat(img.convert('L')).mean[0] + 0.5)"""
    lum = rgb_to_l(img)
    return int(float(lum.astype(np.int64).sum()) / lum.size + 0.5)


def adjust_contrast(img, f):
    return blend_u8(np.full_like(img, contrast_mean(img)), img, f)


def adjust_saturation(img, f):
    lum = rgb_to_l(img)
    return blend_u8(np.repeat(lum[..., None], 3, axis=2), img, f)


def rgb2hsv_u8(img):
    """Pillow Convert.c rgb2hsv_row.  The C code mixes float variables with double literals:
    `bc - gc` is a float operation, `2.0 + rc - bc` is evaluated in double and rounded to
    float on assignment, `fmod(h / 6.0 + 1.0, 1.0)` is double rounded to float, and the
    final `h * 255.0` / `s * 255.0` are double products truncated to int."""
    r, g, b = (img[..., i].astype(np.int32) for i in range(3))
    maxc = np.maximum(r, np.maximum(g, b))
    minc = np.minimum(r, np.minimum(g, b))
    cr = (maxc - minc).astype(np.float32)
    safe = np.where(cr == 0, np.float32(1), cr)
    s = cr / np.where(maxc == 0, 1, maxc).astype(np.float32)
    rc = (maxc - r).astype(np.float32) / safe
    gc = (maxc - g).astype(np.float32) / safe
    bc = (maxc - b).astype(np.float32) / safe
    h1 = bc - gc
    h2 = (2.0 + rc.astype(np.float64) - bc.astype(np.float64)).astype(np.float32)
    h3 = (4.0 + gc.astype(np.float64) - rc.astype(np.float64)).astype(np.float32)
    h = np.where(r == maxc, h1, np.where(g == maxc, h2, h3)).astype(np.float32)
    h = np.fmod(h.astype(np.float64) / 6.0 + 1.0, 1.0).astype(np.float32)
    uh = np.clip((h.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    us = np.clip((s.astype(np.float64) * 255.0).astype(np.int32), 0, 255)
    gray = maxc == minc
    out = np.empty(img.shape, np.uint8)
    out[..., 0] = np.where(gray, 0, uh)
    out[..., 1] = np.where(gray, 0, us)
    out[..., 2] = maxc
    return out


def hsv2rgb_u8(hsv):
    """Pillow Convert.c hsv2rgb (float32 intermediates, round half away from zero)."""
    h = hsv[..., 0].astype(np.float32)
    s = hsv[..., 1]
    v = hsv[..., 2].astype(np.float32)
    hf = h * np.float32(6.0) / np.float32(255.0)
    i = np.floor(hf)
    f = hf - i
    fs = s.astype(np.float32) / np.float32(255.0)

    def rnd(x):                                                   # C round()
        return np.clip(np.floor(x.astype(np.float64) + 0.5), 0, 255).astype(np.uint8)

    p = rnd(v * (np.float32(1.0) - fs))
    q = rnd(v * (np.float32(1.0) - fs * f))
    t = rnd(v * (np.float32(1.0) - fs * (np.float32(1.0) - f)))
    vv = hsv[..., 2]
    sel = i.astype(np.int32) % 6
    r = np.choose(sel, [vv, q, p, p, t, vv])
    g = np.choose(sel, [t, vv, vv, q, p, p])
    b = np.choose(sel, [p, p, t, vv, vv, q])
    out = np.stack([r, g, b], axis=-1).astype(np.uint8)
    gray = s == 0
    out[gray] = vv[gray][:, None]
    return out


def adjust_hue(img, hue_factor):
    """torchvision _functional_pil.adjust_hue: uint8 wrap-around shift of the H channel"""
    hsv = rgb2hsv_u8(img)
    hsv[..., 0] = (hsv[..., 0].astype(np.int32) + int(np.uint8(int(hue_factor * 255) & 0xFF))).astype(np.uint8)
    return hsv2rgb_u8(hsv)


def to_grayscale3(img):
    return np.repeat(rgb_to_l(img)[..., None], 3, axis=2)


# ----------------------------------------------------------------------------- blur + ToTensor
def gaussian_kernel1d(ksize, sigma):
    half = (ksize - 1) * 0.5
    x = np.linspace(-half, half, ksize, dtype=np.float32)
    pdf = np.exp(np.float32(-0.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
    return (pdf / pdf.sum(dtype=np.float32)).astype(np.float32)


def gaussian_blur_u8(img, ksize, sigma):
    """float32 2-D convolution with reflect padding, round-half-even, uint8"""
    k1 = gaussian_kernel1d(ksize, sigma)
    k2 = np.outer(k1, k1).astype(np.float32)
    pad = ksize // 2
    x = np.pad(img.astype(np.float32), ((pad, pad), (pad, pad), (0, 0)), mode="reflect")
    H, W = img.shape[:2]
    acc = np.zeros(img.shape, np.float32)
    for dy in range(ksize):
        for dx in range(ksize):
            acc += k2[dy, dx] * x[dy:dy + H, dx:dx + W]
    return np.clip(np.rint(acc), 0, 255).astype(np.uint8)


def to_tensor(img):
    """uint8 [H,W,3] -> float32 [3,H,W] in [0,1]"""
    return (img.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1)


# ----------------------------------------------------------------------------- one view
def apply_view(img, prm, out_size, ksize=7):
    """prm: dict(top, left, h, w, flip, order[4], brightness, contrast, saturation, hue,
    gray, sigma) -- one sampled parameter set (vit-ssl_amd/data/multicrop.py).  Returns
    float32 [3, S, S]."""
    x = resized_crop_u8(img, prm["top"], prm["left"], prm["h"], prm["w"], out_size, out_size, prm["flip"])
    for fn in prm["order"]:
        if fn == 0:
            x = adjust_brightness(x, prm["brightness"])
        elif fn == 1:
            x = adjust_contrast(x, prm["contrast"])
        elif fn == 2:
            x = adjust_saturation(x, prm["saturation"])
        else:
            x = adjust_hue(x, prm["hue"])
    if prm["gray"]:
        x = to_grayscale3(x)
    x = gaussian_blur_u8(x, ksize, prm["sigma"])
    return to_tensor(x)
