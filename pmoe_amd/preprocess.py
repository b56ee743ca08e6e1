"""GPU input pipeline: drop-in for the transform chain of ``model/data_loader.py:255-275`` and
``autoagents/image_agent.py:71-78,132-136`` -- ``Crop([top, bottom])`` -> ``transforms.Resize((h, w))`` -> ``ToTensor()`` --
on raw uint8 camera frames that are already on the device.  Bit-exact with Pillow's BILINEAR resample (which is what
torchvision's Resize runs on a PIL image): the coefficient tables are computed here exactly as Pillow's
``precompute_coeffs`` / ``normalize_coeffs_8bpc`` do (double precision, 22-bit fixed point) and the two integer passes run
in ``csrc/preprocess.hip``.
"""
import ctypes as C
import math

import torch

from .hip import check, load, stream_ptr

PRECISION_BITS = 32 - 8 - 2


def _coeffs(in_size, out_size):
    """Resample.c:precompute_coeffs for the triangle filter over the whole axis -> ksize, bounds, fixed-point weights."""
    scale = filterscale = in_size / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    bounds, kk = [], []
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = []
        for x in range(xmax):
            t = abs((x + xmin - center + 0.5) * ss)
            w.append(1.0 - t if t < 1.0 else 0.0)
        ww = sum(w)                                   # left-to-right, like the C loop
        if ww != 0.0:
            w = [v / ww for v in w]
        w += [0.0] * (ksize - xmax)
        kk.append([int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS)) for v in w])
        bounds.append([xmin, xmax])
    return ksize, bounds, kk


class FramePreprocessor:
    """``pre = FramePreprocessor(crop=(125, 90), size=(224, 224))``; ``pre(frames)`` with ``frames`` a uint8 device tensor
    ``[..., H0, W0, 3]`` (RGB, HWC: what ``imread`` / the CARLA camera deliver) returns float32 ``[..., 3, h, w]`` in
    [0, 1] -- the tensor ``transform(img)`` of the reference produces for each frame, stacked."""

    def __init__(self, crop=(125, 90), size=(224, 224)):
        self.top, self.bottom = int(crop[0]), int(crop[1])
        self.size = (int(size[0]), int(size[1]))
        self._tables = {}

    def _table(self, in_size, out_size, device):
        key = (in_size, out_size, str(device))
        t = self._tables.get(key)
        if t is None:
            ksize, bounds, kk = _coeffs(in_size, out_size)
            t = self._tables[key] = (ksize, torch.tensor(bounds, dtype=torch.int32, device=device),
                                     torch.tensor(kk, dtype=torch.int32, device=device))
        return t

    def __call__(self, frames):
        if not isinstance(frames, torch.Tensor) or frames.dtype != torch.uint8:
            raise TypeError("FramePreprocessor: expected a uint8 tensor [..., H0, W0, C]")
        if not frames.is_cuda:
            raise RuntimeError("FramePreprocessor: frames must be on the MI355X (cuda) device; pmoe_amd has no CPU path")
        if frames.dim() < 3:
            raise ValueError("FramePreprocessor: expected [..., H0, W0, C]")
        return self._run(frames, labels=False)

    def labels(self, masks):
        """Label pipeline of stage-1 training (``data_loader.py:282-286,305-309``: ``Crop`` -> ``Resize`` ->
        ``MaskPILToTensor``): ``masks`` uint8 class-id images ``[..., H0, W0]`` on the device -> int64 ``[..., h, w]``.
        Like the reference, the resize is BILINEAR (a mode-'L' PIL image): ids are blended at region borders."""
        if not isinstance(masks, torch.Tensor) or masks.dtype != torch.uint8:
            raise TypeError("FramePreprocessor.labels: expected a uint8 tensor [..., H0, W0]")
        if not masks.is_cuda:
            raise RuntimeError("FramePreprocessor: masks must be on the MI355X (cuda) device; pmoe_amd has no CPU path")
        if masks.dim() < 2:
            raise ValueError("FramePreprocessor.labels: expected [..., H0, W0]")
        out = self._run(masks.unsqueeze(-1), labels=True)
        return out.squeeze(-3)

    def _run(self, frames, labels):
        lead = frames.shape[:-3]
        H0, W0, Cc = frames.shape[-3:]
        rows = H0 - self.top - self.bottom
        if rows < 1:
            raise ValueError(f"crop ({self.top}, {self.bottom}) leaves no rows of a {H0}-row frame")
        h, w = self.size
        src = frames.contiguous().view(-1, H0, W0, Cc)
        n = src.shape[0]
        dev = src.device
        kh, bh, ch = self._table(W0, w, dev)
        kv, bv, cv = self._table(rows, h, dev)
        tmp = torch.empty(n, rows, w, Cc, dtype=torch.uint8, device=dev)
        out = torch.empty(n, Cc, h, w, dtype=torch.int64 if labels else torch.float32, device=dev)
        p = lambda t: C.c_void_p(t.data_ptr())      # noqa: E731
        check(load().pmoe_resample_u8_horizontal(p(src), p(tmp), n, H0, W0, self.top, rows, Cc, w, p(bh), p(ch), kh,
                                                 stream_ptr()), "pmoe_resample_u8_horizontal")
        vert = load().pmoe_resample_u8_vertical_to_i64 if labels else load().pmoe_resample_u8_vertical_to_f32
        check(vert(p(tmp), p(out), n, rows, w, Cc, h, p(bv), p(cv), kv, stream_ptr()), "pmoe_resample_u8_vertical")
        return out.view(*lead, Cc, h, w)
