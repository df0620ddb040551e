"""Writes tests/golden/augment.npz: inputs and expected outputs of the per-view image
arithmetic of the reference's DINO multi-crop pipeline, produced by the libraries the
reference itself calls for it -- Pillow (crop / resize / ImageEnhance / HSV / convert('L'))
and torch.nn.functional.conv2d (torchvision's tensor Gaussian blur) -- run in this container.
torchvision is not installed here, so its thin PIL wrappers (torchvision/transforms/
_functional_pil.py, _functional_tensor.gaussian_blur) are spelled out below from their
published source; the arithmetic is Pillow's / ATen's own.

    python tests/golden/make_augment_golden.py
"""
import os

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image, ImageEnhance

HERE = os.path.dirname(os.path.abspath(__file__))


def make_image(h, w, seed):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    base = np.stack([127 + 120 * np.sin(xx / 7.0 + yy / 11.0), 127 + 120 * np.cos(xx / 5.0 - yy / 3.0), (xx * yy) % 256], -1)
    return np.clip(base + rng.normal(0, 20, base.shape), 0, 255).astype(np.uint8)


def pil_hue(p, hf):                       # torchvision _functional_pil.adjust_hue
    h, s, v = p.convert("HSV").split()
    nh = np.array(h, dtype=np.uint8)
    with np.errstate(over="ignore"):
        nh += np.array(hf * 255).astype(np.uint8)
    return Image.merge("HSV", (Image.fromarray(nh, "L"), s, v)).convert("RGB")


def tv_blur(p, ksize, sigma):             # torchvision _functional_tensor.gaussian_blur on pil_to_tensor(img)
    half = (ksize - 1) * 0.5
    x = torch.linspace(-half, half, steps=ksize)
    pdf = torch.exp(-0.5 * (x / sigma).pow(2))
    k1 = pdf / pdf.sum()
    k2 = torch.mm(k1[:, None], k1[None, :])
    t = torch.from_numpy(np.array(p)).permute(2, 0, 1)[None].to(torch.float32)
    pad = ksize // 2
    t = F.pad(t, [pad, pad, pad, pad], mode="reflect")
    o = F.conv2d(t, k2.expand(3, 1, ksize, ksize), groups=3)
    return Image.fromarray(torch.round(o).to(torch.uint8)[0].permute(1, 2, 0).numpy())


def main():
    out = {}
    img = make_image(96, 96, 0)
    pil = Image.fromarray(img)
    out["img"] = img
    crops = [(0, 0, 96, 96, 224), (5, 7, 60, 80, 224), (10, 20, 33, 17, 96), (3, 3, 90, 91, 48), (0, 0, 96, 96, 48), (2, 1, 13, 11, 96),
             (0, 0, 96, 96, 96)]
    out["crops"] = np.array(crops, np.int32)
    for n, (t, l, h, w, S) in enumerate(crops):
        out[f"crop{n}"] = np.array(pil.crop((l, t, l + w, t + h)).resize((S, S), Image.BILINEAR))
    out["flip"] = np.array(pil.transpose(Image.FLIP_LEFT_RIGHT))
    out["L"] = np.array(pil.convert("L"))
    out["gray3"] = np.dstack([np.array(pil.convert("L"))] * 3)
    bc = [0.6, 0.83, 1.0, 1.27, 1.4]
    sat = [0.8, 0.93, 1.0, 1.11, 1.2]
    hue = [-0.1, -0.03, 0.0, 0.05, 0.1]
    out["bc_factors"], out["sat_factors"], out["hue_factors"] = np.array(bc), np.array(sat), np.array(hue)
    for n, f in enumerate(bc):
        out[f"brightness{n}"] = np.array(ImageEnhance.Brightness(pil).enhance(f))
        out[f"contrast{n}"] = np.array(ImageEnhance.Contrast(pil).enhance(f))
    for n, f in enumerate(sat):
        out[f"saturation{n}"] = np.array(ImageEnhance.Color(pil).enhance(f))
    for n, f in enumerate(hue):
        out[f"hue{n}"] = np.array(pil_hue(pil, f))
    rnd = np.random.default_rng(5).integers(0, 256, (64, 64, 3), dtype=np.uint8)
    out["rnd"] = rnd
    out["rnd_hsv"] = np.array(Image.fromarray(rnd).convert("HSV"))
    out["rnd_as_hsv_to_rgb"] = np.array(Image.fromarray(rnd, "HSV").convert("RGB"))
    sig = [0.1, 0.7, 2.0]
    out["sigmas"] = np.array(sig)
    for n, s in enumerate(sig):
        out[f"blur{n}"] = np.array(tv_blur(pil, 7, s))

    # whole views: the transform list of configs/dino/globals.yaml / locals.yaml with FIXED
    # parameters (the sampled quantities), evaluated with the calls above, then ToTensor
    views = [
        dict(top=4, left=9, h=70, w=81, flip=1, order=[2, 0, 3, 1], brightness=1.21, contrast=0.77, saturation=1.13, hue=-0.06,
             gray=0, sigma=1.3, size=224),
        dict(top=0, left=0, h=96, w=96, flip=0, order=[0, 1, 2, 3], brightness=0.64, contrast=1.38, saturation=0.85, hue=0.09,
             gray=1, sigma=0.35, size=224),
        dict(top=40, left=31, h=29, w=37, flip=1, order=[3, 2, 1, 0], brightness=1.05, contrast=1.0, saturation=1.19, hue=0.0,
             gray=0, sigma=1.9, size=96),
    ]
    for n, v in enumerate(views):
        p = pil.crop((v["left"], v["top"], v["left"] + v["w"], v["top"] + v["h"])).resize((v["size"], v["size"]), Image.BILINEAR)
        if v["flip"]:
            p = p.transpose(Image.FLIP_LEFT_RIGHT)
        for fn in v["order"]:
            if fn == 0:
                p = ImageEnhance.Brightness(p).enhance(v["brightness"])
            elif fn == 1:
                p = ImageEnhance.Contrast(p).enhance(v["contrast"])
            elif fn == 2:
                p = ImageEnhance.Color(p).enhance(v["saturation"])
            else:
                p = pil_hue(p, v["hue"])
        if v["gray"]:
            p = Image.fromarray(np.dstack([np.array(p.convert("L"))] * 3))
        p = tv_blur(p, 7, v["sigma"])
        out[f"view{n}"] = (np.array(p).astype(np.float32) / 255.0).transpose(2, 0, 1)     # ToTensor
        out[f"view{n}_params"] = np.array([v["top"], v["left"], v["h"], v["w"], v["flip"], *v["order"], v["gray"], v["size"]], np.int32)
        out[f"view{n}_factors"] = np.array([v["brightness"], v["contrast"], v["saturation"], v["hue"], v["sigma"]], np.float64)
    np.savez_compressed(os.path.join(HERE, "augment.npz"), **out)
    print("wrote", os.path.join(HERE, "augment.npz"), os.path.getsize(os.path.join(HERE, "augment.npz")), "bytes")


if __name__ == "__main__":
    main()
