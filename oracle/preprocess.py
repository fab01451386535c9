"""CPU restatement of the image preprocessing that feeds the hot path (TEST INFRASTRUCTURE ONLY — never imported by the product).

Reference call chain (datasets/LazySupervisedDataset.py:535-556):
  image_rgb (uint8 HWC)
    -> ResizeLongestSide(256 | 336).apply_image          (model/segment_anything/utils/transforms.py:25-34: torchvision
                                                           `resize(to_pil_image(image), (h, w))` = PIL `Image.resize((w, h), BILINEAR)`)
    -> SAM:  (x - pixel_mean) / pixel_std, then centre zero-pad to 256 x 256        (LazySupervisedDataset.py:480-496, :394-395)
    -> CLIP: centre pad with the integer CLIP mean to 336 x 336 (uint8), then HF CLIPImageProcessor.preprocess = rescale 1/255 +
             (x - mean) / std (its resize / centre crop are no-ops on a 336 x 336 input)  (LazySupervisedDataset.py:498-500, :546-553)

The arithmetic provider of the resize is PIL's ImagingResample (Pillow `src/libImaging/Resample.c`, 8 bits per channel path):
two separable passes (horizontal first, 8-bit intermediate), coefficients computed in double, normalised, converted to 22-bit
fixed point, accumulated from 1 << 21 and shifted.  torchvision is not installed here, so the torchvision glue (one call) is
restated; the resampler itself is PINNED against the real PIL in this container (tests/test_preprocess.py runs both on random
images and sizes and requires byte equality; Pillow 12.2.0).  The CLIP normalisation follows transformers' image_transforms
`rescale` / `normalize` (4.31: `image * scale` in float64 cast to float32, then float32 `(image - mean) / std`); transformers 4.31
is not installed (5.15 is): that leg is restated, tolerance-free because it is a per-value table.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
SAM_PIXEL_MEAN = (123.675, 116.28, 103.53)          # LazySupervisedDataset.py:394
SAM_PIXEL_STD = (58.395, 57.12, 57.375)             # :395
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)     # :398 (OPENAI_CLIP_MEAN)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)     # :399


def get_preprocess_shape(oldh, oldw, long_side_length):
    """transforms.py:98-108."""
    scale = long_side_length * 1.0 / max(oldh, oldw)
    newh, neww = oldh * scale, oldw * scale
    return int(newh + 0.5), int(neww + 0.5)


def bilinear_coeffs(in_size, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0) over the whole axis.
    -> (bounds int32 [out, 2] = (xmin, count), coeffs int32 [out, ksize])."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(xmax, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = (x + xmin - center + 0.5) * ss
            if a < 0.0:
                a = -a
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w = w / ww
        for x in range(xmax):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _resample_axis(img, out_size, axis):
    """One 8 bpc pass along `axis` (0 = vertical, 1 = horizontal) of an HWC uint8 array."""
    in_size = img.shape[axis]
    bounds, kk = bilinear_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, cnt = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(cnt):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def pil_resize_bilinear(img, out_h, out_w):
    """PIL Image.resize((out_w, out_h), BILINEAR) of a uint8 HWC (or HW) array: horizontal pass, then vertical (each only when
    that size changes; an unchanged size returns the image as is, Image.py resize)."""
    squeeze = img.ndim == 2
    x = img[:, :, None] if squeeze else img
    if x.shape[1] != out_w:
        x = _resample_axis(x, out_w, 1)
    if x.shape[0] != out_h:
        x = _resample_axis(x, out_h, 0)
    return x[:, :, 0] if squeeze else x


def resize_longest_side(img, target):
    """ResizeLongestSide(target).apply_image (transforms.py:25-34)."""
    h, w = get_preprocess_shape(img.shape[0], img.shape[1], target)
    return pil_resize_bilinear(img, h, w)


def sam_value_table():
    """(x - pixel_mean) / pixel_std for x = 0..255 per channel, in torch float32 like LazySupervisedDataset.preprocess:484."""
    import torch
    x = torch.arange(256, dtype=torch.uint8).view(1, 256)
    mean = torch.Tensor(list(SAM_PIXEL_MEAN)).view(3, 1)
    std = torch.Tensor(list(SAM_PIXEL_STD)).view(3, 1)
    return ((x - mean) / std).numpy()                       # [3, 256] float32


def clip_pad_values():
    """LazySupervisedDataset.py:398: (mean * 255).clamp(0, 255).to(int) -> (122, 116, 104)."""
    import torch
    return [int(v) for v in (torch.Tensor(list(CLIP_MEAN)) * 255).clamp(0, 255).to(torch.int)]


def clip_value_table():
    """HF CLIPImageProcessor on uint8 input: rescale (float64 product cast to float32), then float32 (x - mean) / std."""
    x = np.arange(256, dtype=np.uint8)
    r = (x * (1 / 255)).astype(np.float32)                  # image_transforms.rescale
    mean = np.array(CLIP_MEAN, dtype=np.float32)[:, None]
    std = np.array(CLIP_STD, dtype=np.float32)[:, None]
    return ((r[None, :] - mean) / std).astype(np.float32)   # [3, 256]


def _pad_center(chw, size, pad_values):
    """pad_tensor_channelwise (LazySupervisedDataset.py:446-477): top = pad_h // 2, left = pad_w // 2."""
    c, h, w = chw.shape
    out = np.empty((c, size, size), dtype=chw.dtype)
    for i in range(c):
        out[i] = pad_values[i]
    top, left = (size - h) // 2, (size - w) // 2
    out[:, top:top + h, left:left + w] = chw
    return out


def preprocess_sam(img_rgb, size=256):
    """uint8 HWC RGB -> (float32 [3, size, size], (resize_h, resize_w))."""
    r = resize_longest_side(img_rgb, size)
    tab = sam_value_table()
    chw = np.stack([tab[c][r[:, :, c]] for c in range(3)])
    return _pad_center(chw, size, [0.0, 0.0, 0.0]), r.shape[:2]


def preprocess_clip(img_rgb, size=336):
    """uint8 HWC RGB -> float32 [3, size, size] (image_aspect_ratio == 'pad')."""
    r = resize_longest_side(img_rgb, size)
    padded = _pad_center(np.ascontiguousarray(r.transpose(2, 0, 1)), size, clip_pad_values())
    tab = clip_value_table()
    return np.stack([tab[c][padded[c]] for c in range(3)])


def preprocess_region_mask(mask, size=336):
    """Region masks (LazySupervisedDataset.py:516-517): resize like the CLIP image, centre zero-pad, uint8 HW."""
    r = resize_longest_side(mask.astype(np.uint8), size)
    return _pad_center(r[None], size, [0])[0]
