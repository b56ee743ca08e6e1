"""CPU restatement of the image pre-processing of the hot path's callers.  TEST INFRASTRUCTURE.

Reference pipeline: ``model/data_loader.py:255-275`` (train/val) and ``autoagents/image_agent.py:71-78,132-136``:
``Crop([top, bottom])`` (``model/augmenter.py:43-49``: rows ``[top:-bottom]`` -> ``PIL.Image``) ->
``torchvision.transforms.Resize((h, w))`` -> ``ToTensor()``.  torchvision (0.9.1 in the reference's requirements, not
installed here) implements Resize on a PIL image as ``img.resize((w, h), Image.BILINEAR)`` and ToTensor as
``uint8 HWC -> float32 CHW / 255``.  The arithmetic therefore lives in Pillow's ``ImagingResample`` (libImaging/Resample.c),
restated here: separable two-pass (horizontal, then vertical) convolution with the triangle filter whose support is
stretched by the down-scaling factor (that is Pillow's antialiasing), coefficients normalised in double precision and
rounded to 22-bit fixed point, 8-bit intermediate.  ``tests/test_preprocess_cpu.py`` pins it bit-exactly against the
Pillow installed in the build container (fixtures under tests/golden/prep_*.npz carry Pillow's own outputs).
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def precompute_coeffs(in_size, out_size):
    """Resample.c:precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR filter over the whole axis.
    -> ksize, bounds [out_size, 2] (first tap, tap count), coeffs int32 [out_size, ksize]."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale                     # bilinear_filter support = 1.0
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        ww = 0.0
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w = 1.0 - t if t < 1.0 else 0.0
            kk[xx, x] = w
            ww += w
        if ww != 0.0:
            kk[xx, :xmax] /= ww
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, (-0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64),
                     (0.5 + kk * (1 << PRECISION_BITS)).astype(np.int64)).astype(np.int32)
    return ksize, bounds, fixed


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resample_axis(img, bounds, coeffs, axis):
    """one pass of ImagingResampleHorizontal/Vertical_8bpc along ``axis`` of a uint8 [H, W, C] array."""
    img = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((bounds.shape[0],) + img.shape[1:], dtype=np.uint8)
    for xx, (xmin, cnt) in enumerate(bounds):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(cnt):
            acc += img[xmin + x] * int(coeffs[xx, x])
        out[xx] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def preprocess_frame(frame, crop, size):
    """frame uint8 [H0, W0, 3] -> float32 [3, h, w]: Crop -> Resize (Pillow BILINEAR) -> ToTensor."""
    top, bottom = crop
    img = frame[top:frame.shape[0] - bottom]
    h, w = size
    if img.shape[1] != w:                              # Pillow runs the horizontal pass first, then the vertical one
        _, b, k = precompute_coeffs(img.shape[1], w)
        img = resample_axis(img, b, k, axis=1)
    if img.shape[0] != h:
        _, b, k = precompute_coeffs(img.shape[0], h)
        img = resample_axis(img, b, k, axis=0)
    return np.ascontiguousarray(img.transpose(2, 0, 1)).astype(np.float32) / np.float32(255.0)


def preprocess_mask(mask, crop, size):
    """single-channel class-id image uint8 [H0, W0] -> int64 [h, w]: Crop -> Resize (BILINEAR on a mode-'L' PIL image, as
    the reference's ``transform_mask`` does, data_loader.py:282-286) -> MaskPILToTensor (augmenter.py:52-54)."""
    top, bottom = crop
    img = mask[top:mask.shape[0] - bottom][:, :, None]
    h, w = size
    if img.shape[1] != w:
        _, b, k = precompute_coeffs(img.shape[1], w)
        img = resample_axis(img, b, k, axis=1)
    if img.shape[0] != h:
        _, b, k = precompute_coeffs(img.shape[0], h)
        img = resample_axis(img, b, k, axis=0)
    return np.ascontiguousarray(img[:, :, 0]).astype(np.int64)
