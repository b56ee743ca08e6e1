"""Image pre-processing row: the numpy restatement of Pillow's BILINEAR resample
(oracle/preprocess_oracle.py) against fixtures produced by Pillow itself through the reference's transform chain."""
import numpy as np
import pytest

from oracle.make_prep_golden import CASES, MASK_CASES, frame, mask
from oracle.preprocess_oracle import precompute_coeffs, preprocess_frame, preprocess_mask


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_pillow_fixture(golden_dir, case):
    name, H0, W0, crop, size, seed = case
    g = np.load(golden_dir / "prep.npz")
    assert tuple(g[name + "__meta"]) == (H0, W0, crop[0], crop[1], size[0], size[1], seed)
    got = preprocess_frame(frame(H0, W0, seed), crop, size)
    want = g[name].transpose(2, 0, 1).astype(np.float32) / np.float32(255.0)      # ToTensor
    assert got.dtype == np.float32 and got.shape == (3,) + tuple(size)
    assert np.array_equal(got, want)                                              # bit exact


@pytest.mark.parametrize("case", MASK_CASES, ids=[c[0] for c in MASK_CASES])
def test_oracle_matches_pillow_mask_fixture(golden_dir, case):
    """stage-1 label images (data_loader.py:282-286): Crop -> Resize -> MaskPILToTensor."""
    name, H0, W0, crop, size, seed = case
    g = np.load(golden_dir / "prep.npz")
    got = preprocess_mask(mask(H0, W0, seed), crop, size)
    assert got.dtype == np.int64 and np.array_equal(got, g[name].astype(np.int64))


def test_coefficients_are_normalised_fixed_point():
    for n_in, n_out in ((385, 224), (800, 224), (90, 160), (224, 224)):
        ks, bounds, kk = precompute_coeffs(n_in, n_out)
        assert bounds.shape == (n_out, 2) and kk.shape == (n_out, ks)
        assert (bounds[:, 0] >= 0).all() and (bounds[:, 0] + bounds[:, 1] <= n_in).all()
        assert np.abs(kk.astype(np.int64).sum(1) - (1 << 22)).max() <= ks          # rows sum to 1.0 in 22-bit fixed point


def test_product_coefficient_tables_equal_the_oracle():
    """pmoe_amd/preprocess.py computes the tap tables on the host (no oracle import in the product): same integers."""
    from pmoe_amd.preprocess import _coeffs
    for n_in, n_out in ((385, 224), (800, 224), (385, 256), (800, 256), (90, 160), (224, 224), (291, 97), (203, 65)):
        k1, b1, c1 = _coeffs(n_in, n_out)
        k2, b2, c2 = precompute_coeffs(n_in, n_out)
        assert k1 == k2 and np.array_equal(np.array(b1), b2) and np.array_equal(np.array(c1), c2)


def test_preprocessor_rejects_cpu_and_float_inputs():
    import torch
    from pmoe_amd.preprocess import FramePreprocessor
    pre = FramePreprocessor()
    with pytest.raises(RuntimeError, match="no CPU path"):
        pre(torch.zeros(600, 800, 3, dtype=torch.uint8))
    with pytest.raises(TypeError):
        pre(torch.zeros(600, 800, 3))
