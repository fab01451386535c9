"""Device-side image preprocessing for the hot path's inputs: what the reference does per sample on the CPU
inside DataLoader workers (datasets/LazySupervisedDataset.py:535-556), as three byte kernels on the GPU.

    img = torch.from_numpy(rgb_uint8_hwc).cuda()
    images, resize = preprocess_sam(img)          # [3, 256, 256] f32, (h, w) for resize_list
    images_clip = preprocess_clip(img)            # [3, 336, 336]
    region = preprocess_region_mask(mask_u8)      # [336, 336] uint8

The resize is bit-exact with `ResizeLongestSide.apply_image` (torchvision resize of a PIL image = PIL's 8-bit bilinear
ImagingResample; model/segment_anything/utils/transforms.py:25-34); the normalisations go through 256-entry per-channel tables
built here in the reference's own operation order, so every output value is the reference's, bit for bit.
There is no CPU fallback: inputs must be CUDA uint8 tensors."""
import ctypes
import functools

import numpy as np
import torch

from . import ops
from ._lib import lib

SAM_PIXEL_MEAN = (123.675, 116.28, 103.53)          # LazySupervisedDataset.py:394-395
SAM_PIXEL_STD = (58.395, 57.12, 57.375)
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)     # :398-399 / OPENAI_CLIP_MEAN, OPENAI_CLIP_STD
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def get_preprocess_shape(oldh, oldw, long_side_length):
    """ResizeLongestSide.get_preprocess_shape (transforms.py:98-108)."""
    scale = long_side_length * 1.0 / max(oldh, oldw)
    newh, neww = oldh * scale, oldw * scale
    return int(newh + 0.5), int(neww + 0.5)


def bilinear_coeffs_host(in_size, out_size):
    """PIL's window bounds and fixed-point coefficients of one axis (host C code in the library) -> (bounds [out, 2], coefs [out, k]) int32."""
    L = lib()
    k = L._raw_mp_pil_bilinear_ksize(in_size, out_size)
    bounds = np.empty((out_size, 2), dtype=np.int32)
    coefs = np.empty((out_size, k), dtype=np.int32)
    L.call("mp_pil_bilinear_coeffs", in_size, out_size, bounds.ctypes.data_as(ctypes.c_void_p), coefs.ctypes.data_as(ctypes.c_void_p), k)
    return bounds, coefs


@functools.lru_cache(maxsize=256)
def _axis_tables(in_size, out_size, device):
    b, c = bilinear_coeffs_host(in_size, out_size)
    return (torch.from_numpy(b).to(device), torch.from_numpy(c).to(device), c.shape[1])


def resize_bilinear_u8(img, out_h, out_w):
    """PIL Image.resize((out_w, out_h), BILINEAR) of a CUDA uint8 [H, W, C] (or [H, W]) tensor, bit-exact: horizontal pass, then
    vertical, each only when that size changes."""
    assert img.is_cuda and img.dtype == torch.uint8 and img.dim() in (2, 3), "resize_bilinear_u8: CUDA uint8 HWC / HW tensor"
    x = img.contiguous()
    squeeze = x.dim() == 2
    if squeeze:
        x = x[:, :, None]
    H, W, C = x.shape
    dev = str(x.device)
    if W != out_w:
        b, c, k = _axis_tables(W, out_w, dev)
        y = torch.empty((H, out_w, C), dtype=torch.uint8, device=x.device)
        lib().call("mp_resample_axis_u8", ops._p(x), ops._p(y), H, W, out_w, C, ops._p(b), ops._p(c), k, ops._stream())
        x = y
    if H != out_h:
        b, c, k = _axis_tables(H, out_h, dev)
        y = torch.empty((out_h, out_w, C), dtype=torch.uint8, device=x.device)
        lib().call("mp_resample_axis_u8", ops._p(x), ops._p(y), 1, H, out_h, out_w * C, ops._p(b), ops._p(c), k, ops._stream())
        x = y
    return x[:, :, 0] if squeeze else x


class ResizeLongestSide:
    """Device twin of model/segment_anything/utils/transforms.py:ResizeLongestSide (image part)."""

    def __init__(self, target_length):
        self.target_length = target_length

    def apply_image(self, image):
        h, w = get_preprocess_shape(image.shape[0], image.shape[1], self.target_length)
        return resize_bilinear_u8(image, h, w)

    get_preprocess_shape = staticmethod(get_preprocess_shape)


@functools.lru_cache(maxsize=8)
def _tables(kind, device):
    """(table [3, 256] f32, pad [3] f32) on the device, in the reference's operation order."""
    if kind == "sam":           # (uint8 tensor - float mean) / std in torch float32 (LazySupervisedDataset.preprocess:484); zero pad after
        x = torch.arange(256, dtype=torch.uint8).view(1, 256)
        tab = (x - torch.Tensor(list(SAM_PIXEL_MEAN)).view(3, 1)) / torch.Tensor(list(SAM_PIXEL_STD)).view(3, 1)
        pad = torch.zeros(3)
    elif kind == "clip":        # integer-mean pad BEFORE normalise (:398, :500), then CLIPImageProcessor: rescale in float64 -> float32,
        x = np.arange(256, dtype=np.uint8)          # float32 (x - mean) / std
        r = (x * (1 / 255)).astype(np.float32)
        tab = torch.from_numpy(((r[None, :] - np.array(CLIP_MEAN, dtype=np.float32)[:, None]) /
                                np.array(CLIP_STD, dtype=np.float32)[:, None]).astype(np.float32))
        pad_int = (torch.Tensor(list(CLIP_MEAN)) * 255).clamp(0, 255).to(torch.int)
        pad = torch.stack([tab[c, int(pad_int[c])] for c in range(3)])
    else:
        raise ValueError(kind)
    return tab.contiguous().to(device), pad.contiguous().to(device)


def _table_pad(img, size, kind, out_dtype):
    h, w, C = img.shape
    assert C == 3 and h <= size and w <= size
    tab, pad = _tables(kind, str(img.device))
    out = torch.empty((3, size, size), dtype=out_dtype, device=img.device)
    lib().call("mp_image_table_pad_chw", ops._p(img), h, w, 3, ops._p(tab), ops._p(pad), ops._p(out), size, size, (size - h) // 2,
               (size - w) // 2, ops._dt(out_dtype), ops._stream())
    return out


def preprocess_sam(img_rgb, size=256, out_dtype=torch.float32):
    """CUDA uint8 [H, W, 3] RGB -> (`images` entry [3, size, size], (resize_h, resize_w))  (LazySupervisedDataset.py:535-541)."""
    r = ResizeLongestSide(size).apply_image(img_rgb)
    return _table_pad(r, size, "sam", out_dtype), (int(r.shape[0]), int(r.shape[1]))


def preprocess_clip(img_rgb, size=336, out_dtype=torch.float32):
    """CUDA uint8 [H, W, 3] RGB -> `images_clip` entry [3, size, size] (image_aspect_ratio == 'pad'; :546-553)."""
    r = ResizeLongestSide(size).apply_image(img_rgb)
    return _table_pad(r, size, "clip", out_dtype)


def preprocess_region_mask(mask, size=336):
    """CUDA uint8 [H, W] region mask -> uint8 [size, size]: resized like the CLIP image, centre zero pad (:516-517)."""
    r = ResizeLongestSide(size).apply_image(mask)
    out = torch.zeros((size, size), dtype=torch.uint8, device=mask.device)
    h, w = r.shape
    top, left = (size - h) // 2, (size - w) // 2
    out[top:top + h, left:left + w] = r
    return out


ICL_TINT = (118.0, 158.0, 224.0)                    # ICLLazySupervisedDataset._overlay_mask (:47)


def overlay_mask(img_rgb, mask):
    """CUDA uint8 [H, W, 3] image, uint8 [H, W] mask -> the example image with the mask tinted in (ICL overlay mode, :46-50)."""
    assert img_rgb.dtype == torch.uint8 and mask.dtype == torch.uint8 and img_rgb.is_cuda and mask.is_cuda
    img_rgb, mask = img_rgb.contiguous(), mask.contiguous()
    assert img_rgb.shape[:2] == mask.shape and img_rgb.shape[2] == 3
    out = torch.empty_like(img_rgb)
    lib().call("mp_overlay_mask_u8", ops._p(img_rgb), ops._p(mask), ops._p(out), mask.numel(), *ICL_TINT, ops._stream())
    return out
