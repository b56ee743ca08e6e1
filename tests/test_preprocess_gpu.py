"""GPU input pipeline (pmoe_amd/preprocess.py, csrc/preprocess.hip) against the Pillow fixtures: bit exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.make_prep_golden import CASES, MASK_CASES, frame, mask  # noqa: E402
from pmoe_amd.preprocess import FramePreprocessor  # noqa: E402


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_frames_match_pillow_bit_for_bit(golden_dir, case):
    name, H0, W0, crop, size, seed = case
    g = np.load(golden_dir / "prep.npz")
    want = torch.from_numpy(g[name].transpose(2, 0, 1).astype(np.float32) / np.float32(255.0))
    fr = torch.from_numpy(frame(H0, W0, seed)).cuda()
    got = FramePreprocessor(crop, size)(fr)
    assert got.shape == want.shape and got.dtype == torch.float32
    assert torch.equal(got.cpu(), want)


def test_batched_frames_feed_the_model_input_contract(golden_dir):
    """[B, T, H0, W0, 3] uint8 -> [B, T, 3, h, w] f32: the `images` argument of forward (train_2.py:141-149)."""
    name, H0, W0, crop, size, seed = CASES[0]
    g = np.load(golden_dir / "prep.npz")
    want = torch.from_numpy(g[name].transpose(2, 0, 1).astype(np.float32) / np.float32(255.0))
    one = torch.from_numpy(frame(H0, W0, seed)).cuda()
    batch = torch.stack([torch.stack([one, one.flip(1), one, one.flip(0)]), torch.stack([one.flip(0), one, one, one])])
    out = FramePreprocessor(crop, size)(batch)
    assert out.shape == (2, 4, 3) + tuple(size)
    assert torch.equal(out[0, 0].cpu(), want) and torch.equal(out[1, 3].cpu(), want)
    assert not torch.equal(out[0, 1].cpu(), want)


@pytest.mark.parametrize("case", MASK_CASES, ids=[c[0] for c in MASK_CASES])
def test_label_images_match_pillow_bit_for_bit(golden_dir, case):
    """stage-1 label pipeline (data_loader.py:282-286): uint8 class-id images -> int64 [h, w], batched [B, F, H0, W0]."""
    name, H0, W0, crop, size, seed = case
    g = np.load(golden_dir / "prep.npz")
    want = torch.from_numpy(g[name].astype(np.int64))
    m = torch.from_numpy(mask(H0, W0, seed)).cuda()
    pre = FramePreprocessor(crop, size)
    got = pre.labels(m)
    assert got.dtype == torch.int64 and torch.equal(got.cpu(), want)
    batch = torch.stack([torch.stack([m, m.flip(1)]), torch.stack([m.flip(0), m])])
    out = pre.labels(batch)
    assert out.shape == (2, 2) + tuple(size) and torch.equal(out[1, 1].cpu(), want) and not torch.equal(out[0, 1].cpu(), want)
    with pytest.raises(TypeError):
        pre.labels(m.float())
