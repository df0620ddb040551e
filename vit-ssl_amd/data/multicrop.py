"""DINO multi-crop views produced on the GPU (reference: data/datasets.py:80-123,
configs/dino/globals.yaml, configs/dino/locals.yaml, utils/train_utils.py:54-68).

The reference builds, per image and on the CPU, `num_global_views` global and
`num_all_views - num_global_views` local views by running a torchvision transform list on a
PIL image.  Here a whole batch of decoded uint8 images goes through three HIP kernels per
view group (crop+resize+flip, colour chain, blur+ToTensor); the image arithmetic is
bit-identical to Pillow's (oracle/augment_oracle.py).  What stays on the host is the drawing of
the random parameters -- a few dozen scalars per view.

Sampling (`sample_view_params`) is restated from torchvision's published `get_params`
methods (torchvision is not installed here: "parity unpinned" for the draw ORDER; the
distributions are the documented ones):
  RandomResizedCrop.get_params : up to 10 tries of  area ~ U(scale)*H*W,
        log-ratio ~ U(log 3/4, log 4/3), w = round(sqrt(area*ratio)), h = round(sqrt(area/ratio)),
        accepted when it fits, then top ~ randint(0, H-h+1), left ~ randint(0, W-w+1);
        fallback = centre crop clamped to the ratio range
  RandomHorizontalFlip         : torch.rand(1) < 0.5
  ColorJitter.get_params       : order = randperm(4); brightness, contrast, saturation
        ~ U(max(0, 1-x), 1+x); hue ~ U(-x, x)
  RandomGrayscale              : torch.rand(1) < p
  GaussianBlur.get_params      : sigma ~ U(sigma_min, sigma_max)
No CPU fallback: the views are produced by libvitssl_hip or not at all.
"""
import math
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from vitssl_hip import _lib as L
from vitssl_hip import ops


@dataclass
class ViewSpec:
    """One transform list of configs/dino/{globals,locals}.yaml, reduced to its numbers."""
    size: int
    scale: Tuple[float, float]
    ratio: Tuple[float, float] = (3.0 / 4.0, 4.0 / 3.0)
    flip_p: float = 0.5
    brightness: float = 0.4
    contrast: float = 0.4
    saturation: float = 0.2
    hue: float = 0.1
    gray_p: float = 0.0
    blur_kernel: int = 7
    blur_sigma: Tuple[float, float] = (0.1, 2.0)

    @classmethod
    def from_config(cls, sequence) -> "ViewSpec":
        """sequence: the list of {name, params} entries of a transform YAML."""
        kw = {}
        for entry in sequence:
            name, prm = entry["name"], (entry.get("params") or {})
            if name == "RandomResizedCrop":
                kw["size"] = int(prm["size"])
                kw["scale"] = tuple(prm.get("scale", (0.08, 1.0)))
                if "ratio" in prm:
                    kw["ratio"] = tuple(prm["ratio"])
            elif name == "RandomHorizontalFlip":
                kw["flip_p"] = float(prm.get("p", 0.5))
            elif name == "ColorJitter":
                for k in ("brightness", "contrast", "saturation", "hue"):
                    kw[k] = float(prm.get(k, 0.0))
            elif name == "RandomGrayscale":
                kw["gray_p"] = float(prm.get("p", 0.1))
            elif name == "GaussianBlur":
                kw["blur_kernel"] = int(prm["kernel_size"])
                kw["blur_sigma"] = tuple(prm.get("sigma", (0.1, 2.0)))
            elif name == "ToTensor":
                pass
            else:
                raise ValueError(f"GPUMultiCrop: transform {name!r} is not part of the DINO view recipe")
        if "size" not in kw:
            raise ValueError("GPUMultiCrop: a view recipe needs RandomResizedCrop")
        return cls(**kw)


def _uniform(lo, hi, gen):
    return float(torch.empty(1).uniform_(float(lo), float(hi), generator=gen))


def gaussian_kernel1d(ksize: int, sigma: float) -> np.ndarray:
    """torchvision _get_gaussian_kernel1d in float32 (host side; handed to the blur kernel)."""
    half = (ksize - 1) * 0.5
    x = np.linspace(-half, half, ksize, dtype=np.float32)
    pdf = np.exp(np.float32(-0.5) * (x / np.float32(sigma)) ** 2).astype(np.float32)
    return (pdf / pdf.sum(dtype=np.float32)).astype(np.float32)


def sample_view_params(spec: ViewSpec, height: int, width: int, generator: Optional[torch.Generator] = None) -> dict:
    """One view's random parameters, drawn in the order the transform list consumes them."""
    area = height * width
    log_ratio = (math.log(spec.ratio[0]), math.log(spec.ratio[1]))
    box = None
    for _ in range(10):
        target_area = area * _uniform(spec.scale[0], spec.scale[1], generator)
        aspect = math.exp(_uniform(log_ratio[0], log_ratio[1], generator))
        w = int(round(math.sqrt(target_area * aspect)))
        h = int(round(math.sqrt(target_area / aspect)))
        if 0 < w <= width and 0 < h <= height:
            top = int(torch.randint(0, height - h + 1, (1,), generator=generator))
            left = int(torch.randint(0, width - w + 1, (1,), generator=generator))
            box = (top, left, h, w)
            break
    if box is None:                                      # fallback: centre crop inside the ratio range
        in_ratio = width / height
        if in_ratio < spec.ratio[0]:
            w, h = width, int(round(width / spec.ratio[0]))
        elif in_ratio > spec.ratio[1]:
            h, w = height, int(round(height * spec.ratio[1]))
        else:
            w, h = width, height
        box = ((height - h) // 2, (width - w) // 2, h, w)
    flip = bool(torch.rand(1, generator=generator) < spec.flip_p)
    order = [int(v) for v in torch.randperm(4, generator=generator)]
    jit = {}
    for k in ("brightness", "contrast", "saturation"):
        v = getattr(spec, k)
        jit[k] = _uniform(max(0.0, 1.0 - v), 1.0 + v, generator) if v > 0 else 1.0
    jit["hue"] = _uniform(-spec.hue, spec.hue, generator) if spec.hue > 0 else 0.0
    gray = bool(torch.rand(1, generator=generator) < spec.gray_p) if spec.gray_p > 0 else False
    sigma = _uniform(spec.blur_sigma[0], spec.blur_sigma[1], generator)
    return dict(top=box[0], left=box[1], h=box[2], w=box[3], flip=flip, order=order, gray=gray, sigma=sigma, **jit)


def sample_batch_params(spec: ViewSpec, height: int, width: int, n: int,
                        generator: Optional[torch.Generator] = None) -> dict:
    """`n` independent parameter sets with the distributions of `sample_view_params`, drawn
    from ONE block of uniforms and computed with NumPy array arithmetic (the scalar sampler
    costs ~30 us per view: 75 ms of host time for a batch of 256 image sets).  Returns a dict
    of arrays (same keys as the scalar sampler's dict; `order` is [n, 4]).  The mapping of
    random numbers to parameters differs from the scalar sampler; the distributions do not:
    integers are floor(u * range), the jitter order is the argsort of four uniforms."""
    u = torch.rand(n, 32, generator=generator, dtype=torch.float64).numpy()
    area = float(height * width)
    lr0, lr1 = math.log(spec.ratio[0]), math.log(spec.ratio[1])
    hh = np.zeros(n, np.int64)
    ww = np.zeros(n, np.int64)
    done = np.zeros(n, bool)
    for t in range(10):                                   # torchvision's 10 rejection tries
        target = area * (spec.scale[0] + (spec.scale[1] - spec.scale[0]) * u[:, 2 * t])
        aspect = np.exp(lr0 + (lr1 - lr0) * u[:, 2 * t + 1])
        w = np.rint(np.sqrt(target * aspect)).astype(np.int64)      # rint == round() here (half-even both)
        h = np.rint(np.sqrt(target / aspect)).astype(np.int64)
        ok = (~done) & (w > 0) & (w <= width) & (h > 0) & (h <= height)
        hh[ok], ww[ok] = h[ok], w[ok]
        done |= ok
    top = np.minimum((u[:, 20] * (height - hh + 1)).astype(np.int64), height - hh)
    left = np.minimum((u[:, 21] * (width - ww + 1)).astype(np.int64), width - ww)
    if not done.all():                                    # fallback: centre crop inside the ratio range
        in_ratio = width / height
        if in_ratio < spec.ratio[0]:
            fw, fh = width, int(round(width / spec.ratio[0]))
        elif in_ratio > spec.ratio[1]:
            fh, fw = height, int(round(height * spec.ratio[1]))
        else:
            fw, fh = width, height
        miss = ~done
        hh[miss], ww[miss], top[miss], left[miss] = fh, fw, (height - fh) // 2, (width - fw) // 2

    def jitter(x, col):
        if x <= 0:
            return np.ones(n)
        lo = max(0.0, 1.0 - x)
        return lo + (1.0 + x - lo) * u[:, col]

    return dict(top=top, left=left, h=hh, w=ww, flip=u[:, 22] < spec.flip_p,
                brightness=jitter(spec.brightness, 23), contrast=jitter(spec.contrast, 24), saturation=jitter(spec.saturation, 25),
                hue=(-spec.hue + 2.0 * spec.hue * u[:, 26]) if spec.hue > 0 else np.zeros(n),
                order=np.argsort(u[:, 27:31], axis=1), gray=(u[:, 31] < spec.gray_p) if spec.gray_p > 0 else np.zeros(n, bool),
                sigma=spec.blur_sigma[0] + (spec.blur_sigma[1] - spec.blur_sigma[0]) * u[:, 19])


def params_as_list(arrs: dict) -> List[dict]:
    """array dict of `sample_batch_params` -> list of per-view dicts (tests, debugging)"""
    n = len(arrs["top"])
    return [{k: (v[i].tolist() if k == "order" else v[i].item()) for k, v in arrs.items()} for i in range(n)]


def _kernels1d(ksize: int, sigma: np.ndarray) -> np.ndarray:
    """row-wise `gaussian_kernel1d` (float32, same operation order per row)"""
    half = (ksize - 1) * 0.5
    x = np.linspace(-half, half, ksize, dtype=np.float32)[None, :]
    pdf = np.exp(np.float32(-0.5) * (x / sigma.astype(np.float32)[:, None]) ** 2).astype(np.float32)
    return (pdf / pdf.sum(axis=1, dtype=np.float32)[:, None]).astype(np.float32)


def pack_params(params, ksize: int):
    """list of per-view dicts, or the array dict of `sample_batch_params`
    -> (iparams int32 [B,11], fparams float32 [B,10]) NumPy arrays in the kernels' layout"""
    if isinstance(params, dict):
        n = len(params["top"])
        ip = np.zeros((n, ops.AUG_IP), np.int32)
        fp = np.zeros((n, ops.AUG_FP), np.float32)
        for j, k in enumerate(("top", "left", "h", "w", "flip")):
            ip[:, j] = params[k]
        ip[:, 5:9] = params["order"]
        ip[:, 9] = params["gray"]
        ip[:, 10] = (np.asarray(params["hue"]) * 255).astype(np.int64) & 0xFF       # np.uint8(hue_factor * 255): truncate, wrap
        for j, k in enumerate(("brightness", "contrast", "saturation")):
            fp[:, j] = params[k]
        fp[:, 3:3 + ksize] = _kernels1d(ksize, np.asarray(params["sigma"], np.float64))
        return ip, fp
    ip = np.zeros((len(params), ops.AUG_IP), np.int32)
    fp = np.zeros((len(params), ops.AUG_FP), np.float32)
    for i, p in enumerate(params):
        ip[i, :5] = (p["top"], p["left"], p["h"], p["w"], int(p["flip"]))
        ip[i, 5:9] = p["order"]
        ip[i, 9] = int(p["gray"])
        ip[i, 10] = int(p["hue"] * 255) & 0xFF            # np.uint8(hue_factor * 255): truncate, wrap
        fp[i, :3] = (p["brightness"], p["contrast"], p["saturation"])
        fp[i, 3:3 + ksize] = gaussian_kernel1d(ksize, p["sigma"])
    return ip, fp


class GPUMultiCrop:
    """images uint8 [B, H, W, 3] on the GPU -> list of V float32 tensors [B, 3, S_v, S_v]
    (globals first, as STL10DINODataset._get_dino_views orders them)."""

    def __init__(self, global_spec: ViewSpec, local_spec: ViewSpec, num_all_views: int, num_global_views: int):
        if not 0 < num_global_views <= num_all_views:
            raise ValueError("GPUMultiCrop: need 0 < num_global_views <= num_all_views")
        for s in (global_spec, local_spec):
            if s.blur_kernel != 7:
                raise ValueError("GPUMultiCrop: the blur kernel is built for kernel_size 7 (configs/dino/*.yaml)")
        self.global_spec, self.local_spec = global_spec, local_spec
        self.num_all_views, self.num_global_views = num_all_views, num_global_views
        self._buf = {}

    def _scratch(self, key, shape, dtype, dev):
        t = self._buf.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.device != dev:
            t = torch.empty(shape, dtype=dtype, device=dev)
            self._buf[key] = t
        return t

    def render(self, images: torch.Tensor, params, spec: ViewSpec) -> torch.Tensor:
        """One view for every image of the batch with explicit parameters."""
        if images.device.type != "cuda":
            raise L.VitsslError("GPUMultiCrop: images are on the CPU; move the uint8 batch to 'cuda' (no CPU fallback)")
        B, H, W, _ = images.shape
        S = spec.size
        dev = images.device
        ip, fp = pack_params(params, spec.blur_kernel)
        ip_d = torch.from_numpy(ip).to(dev, non_blocking=True)
        fp_d = torch.from_numpy(fp).to(dev, non_blocking=True)
        tmp = self._scratch(("tmp", S), (B, H, S, 3), torch.uint8, dev)
        u8 = self._scratch(("u8", S), (B, S, S, 3), torch.uint8, dev)
        out = torch.empty(B, 3, S, S, dtype=torch.float32, device=dev)
        ops.aug_resized_crop_u8(images, ip_d, tmp, u8)
        ops.aug_color_u8(u8, ip_d, fp_d)
        ops.aug_blur_to_tensor(u8, fp_d, out, spec.blur_kernel)
        return out

    def __call__(self, images: torch.Tensor, generator: Optional[torch.Generator] = None) -> List[torch.Tensor]:
        B, H, W, _ = images.shape
        views = []
        for v in range(self.num_all_views):
            spec = self.global_spec if v < self.num_global_views else self.local_spec
            views.append(self.render(images, sample_batch_params(spec, H, W, B, generator), spec))
        return views
