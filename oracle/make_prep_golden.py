#!/usr/bin/env python3
"""Fixtures for the image pre-processing row: outputs of the reference's own transform chain --
Crop (model/augmenter.py:43-49) -> PIL resize BILINEAR (what torchvision.transforms.Resize does on a PIL image) -> uint8
HWC before ToTensor's /255 -- produced by the Pillow installed in the build container.  Inputs are re-derived from seeds.
Usage: python oracle/make_prep_golden.py   ->  tests/golden/prep.npz"""
import sys
from pathlib import Path

import numpy as np
from PIL import Image

REPO = Path(__file__).resolve().parents[1]
CASES = [  # name, H0, W0, crop(top, bottom), size(h, w), seed
    ("agent_600x800_to_224", 600, 800, (125, 90), (224, 224), 11),       # image_agent.py:71-78 / stage_2*.yaml:41-46
    ("train_600x800_to_256", 600, 800, (125, 90), (256, 256), 12),       # BASELINE's 256x256 frames
    ("small_240x320_to_128", 240, 320, (60, 40), (128, 128), 13),
    ("upsample_100x90_to_224x160", 100, 90, (5, 5), (224, 160), 14),
    ("odd_301x203_to_97x65", 301, 203, (7, 3), (97, 65), 15),
]


MASK_CASES = [  # name, H0, W0, crop, size, seed   (stage-1 label images: data_loader.py:282-286, conf/stage_1.yaml:41-46)
    ("mask_600x800_to_224", 600, 800, (125, 90), (224, 224), 21),
    ("mask_odd_301x203_to_97x65", 301, 203, (7, 3), (97, 65), 22),
]


def mask(H0, W0, seed, classes=23, block=9):
    """single-channel class-id image, piecewise constant (what cv2.imread(..., IMREAD_UNCHANGED) returns for a label PNG)"""
    rng = np.random.default_rng(seed)
    coarse = rng.integers(0, classes, ((H0 + block - 1) // block, (W0 + block - 1) // block), dtype=np.uint8)
    return np.ascontiguousarray(np.repeat(np.repeat(coarse, block, 0), block, 1)[:H0, :W0])


def frame(H0, W0, seed):
    return np.random.default_rng(seed).integers(0, 256, (H0, W0, 3), dtype=np.uint8)


def main():
    out = {"pillow_version": np.array(Image.__version__ if hasattr(Image, "__version__") else "?")}
    import PIL
    out["pillow_version"] = np.array(PIL.__version__)
    for name, H0, W0, crop, size, seed in CASES:
        fr = frame(H0, W0, seed)
        img = Image.fromarray(fr[crop[0]:-crop[1]])
        res = np.asarray(img.resize((size[1], size[0]), Image.BILINEAR))
        out[name] = res
        out[name + "__meta"] = np.array([H0, W0, crop[0], crop[1], size[0], size[1], seed])
    for name, H0, W0, crop, size, seed in MASK_CASES:
        # Crop (augmenter.py:43-49: Image.fromarray of the row slice -> mode 'L') -> Resize -> MaskPILToTensor (np.array)
        img = Image.fromarray(mask(H0, W0, seed)[crop[0]:-crop[1]])
        assert img.mode == "L"
        out[name] = np.asarray(img.resize((size[1], size[0]), Image.BILINEAR))
        out[name + "__meta"] = np.array([H0, W0, crop[0], crop[1], size[0], size[1], seed])
    np.savez_compressed(REPO / "tests" / "golden" / "prep.npz", **out)
    print("wrote tests/golden/prep.npz", {k: v.shape for k, v in out.items() if not k.endswith("__meta")})


if __name__ == "__main__":
    sys.exit(main())
