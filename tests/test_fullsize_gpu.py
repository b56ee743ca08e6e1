"""BASELINE.json's headline configuration at FULL size (E=4, B=64, 256x256, the bench.py workload), where the CPU oracle
is too slow to run: parity is carried by size-independent properties of the reference's computation
(moe.py:140-158: experts are independent until the gate softmax; loss.py:121-132: the loss is linear in its
coefficients; plain BatchNorm2d: statistics are permutation-invariant over the batch)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from pmoe_amd.loss import moe_loss  # noqa: E402
from pmoe_amd.model.moe import get_model  # noqa: E402
from pmoe_amd.utils import stage2_model_cfg  # noqa: E402

E, B, S = 4, 64, 256


def _batch(seed=1234):
    g = torch.Generator().manual_seed(seed)
    images = torch.rand(B, 4, 3, S, S, generator=g).cuda()
    speed, target = torch.rand(B, 1, generator=g).cuda(), torch.rand(B, 1, generator=g).cuda()
    command = torch.nn.functional.one_hot(torch.randint(0, 6, (B,), generator=g), 6).float().cuda()
    control = (torch.rand(B, 2, generator=g) * 2 - 1).cuda()
    return images, speed, command, control, target


def _model(dtype=torch.bfloat16, seed=0):
    torch.manual_seed(seed)
    m = get_model(stage2_model_cfg("moe", E, dropout=0.0)).cuda()
    m.compute_dtype = dtype
    return m.train()


def _step(m, batch, coefs=(0.7, 0.3)):
    images, speed, command, control, target = batch
    m.zero_grad(set_to_none=True)
    dist, speeds = m(images, speed, command)
    loss = moe_loss(dist, speeds, control, target, list(coefs))
    loss.backward()
    return [t.detach().clone() for t in dist.hip_params] + [speeds.detach().clone()], loss.detach().clone(), \
           {k: p.grad.detach().clone() for k, p in m.named_parameters()}


def test_forward_is_deterministic_and_gradients_reproducible():
    m, batch = _model(), _batch()
    sd = copy.deepcopy(m.state_dict())
    o1, l1, g1 = _step(m, batch)
    m.load_state_dict(sd)                      # undo the BatchNorm running-statistics update
    o2, l2, g2 = _step(m, batch)
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)               # fixed-order reductions everywhere on the forward path
    assert torch.equal(l1, l2)
    for k in g1:                               # no float atomics on the backward path either (fixed-order K-split folds)
        assert torch.equal(g1[k], g2[k]), k


def test_loss_scale_is_exact_in_the_gradients():
    """loss.py:132 is linear in loss_coefs; a factor of 2 is exact in binary floating point, so every kernel of the
    backward pass must reproduce it bit for bit (deterministic reductions; power-of-two scaling commutes with every
    bf16 / f32 rounding on the path)."""
    m, batch = _model(), _batch()
    sd = copy.deepcopy(m.state_dict())
    _, l1, g1 = _step(m, batch, (0.7, 0.3))
    m.load_state_dict(sd)
    _, l2, g2 = _step(m, batch, (1.4, 0.6))
    assert abs(l2.item() - 2 * l1.item()) <= 1e-6 * abs(l1.item())
    for k in g1:
        assert torch.equal(g2[k], 2 * g1[k]), k


def test_experts_are_independent_until_the_gate():
    """Permuting the experts permutes mean / std / speeds bit for bit (moe.py:141-149 stacks per-expert results) and the
    gate probabilities up to the softmax's summation order; gradients travel with their expert."""
    m, batch = _model(), _batch()
    sd = copy.deepcopy(m.state_dict())
    (p1, mu1, sd1, sp1), _, g1 = _step(m, batch)
    perm = [2, 0, 3, 1]
    sd2 = {}
    for k, v in sd.items():
        e = int(k.split(".")[1])
        sd2[k.replace(f"moe.{e}.", f"moe.{perm.index(e)}.", 1)] = v
    m.load_state_dict(sd2)
    (p2, mu2, sdd2, sp2), _, g2 = _step(m, batch)
    for a, b in ((mu1, mu2), (sd1, sdd2), (sp1, sp2)):
        assert torch.equal(a[:, perm], b)
    assert (p1[:, perm] - p2).abs().max() <= 1e-6
    for k in g1:
        e = int(k.split(".")[1])
        k2 = k.replace(f"moe.{e}.", f"moe.{perm.index(e)}.", 1)
        assert (g1[k] - g2[k2]).norm() <= 1e-4 * g1[k].norm() + 1e-10, k


def test_batch_permutation_equivariance():
    """Train-mode BatchNorm statistics do not depend on the order of the samples: permuting the batch permutes the
    outputs (summation order of the statistics changes: agreement to bf16 rounding, not bitwise) and leaves the loss."""
    m, batch = _model(), _batch()
    sd = copy.deepcopy(m.state_dict())
    o1, l1, _ = _step(m, batch)
    idx = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    m.load_state_dict(sd)
    o2, l2, _ = _step(m, tuple(t[idx] for t in batch))
    for a, b in zip(o1, o2):
        assert ((a[idx] - b).abs() / (1 + a[idx].abs())).max() <= 3e-2
        assert ((a[idx] - b).abs() / (1 + a[idx].abs())).median() <= 2e-3
    assert abs(l1.item() - l2.item()) <= 1e-2 * abs(l1.item())


def test_bf16_path_agrees_with_exact_f32_path_at_full_size():
    """The two arithmetic paths of the same kernels (bf16 MFMA + bf16 activations vs exact-f32 MFMA) on the headline
    workload: the f32 path is the one pinned to the reference at 1e-4 on the small goldens."""
    batch = _batch()
    mb, mf = _model(torch.bfloat16), _model(torch.float32)
    mf.load_state_dict(mb.state_dict())
    ob, lb, gb = _step(mb, batch)
    of, lf, gf = _step(mf, batch)
    for a, b in zip(ob, of):
        err = (a - b).abs() / (1 + b.abs())
        assert err.max() <= 6e-2 and err.median() <= 1e-2, (err.max().item(), err.median().item())
    assert abs(lb.item() - lf.item()) <= 3e-2 * max(1.0, abs(lf.item()))
    cos = [torch.nn.functional.cosine_similarity(gb[k].flatten(), gf[k].flatten(), dim=0).item()
           for k in gb if gb[k].numel() >= 1024]
    cos.sort()
    assert cos[len(cos) // 2] >= 0.9, cos[len(cos) // 2]
    tot_b = sum(v.norm().item() ** 2 for v in gb.values()) ** 0.5
    tot_f = sum(v.norm().item() ** 2 for v in gf.values()) ** 0.5
    assert abs(tot_b - tot_f) <= 0.2 * tot_f


# ------------------------------------------------------------------------------------------------
# BASELINE config 3's per-GPU shard (E=8, B=64) and config 4 (PU-Net expert, T=4, F=6, B=64, 256x256) at full size
def test_eight_experts_full_size_properties():
    """E=8, B=64, 256x256 (the shard one GPU of the 8-GPU config runs): deterministic forward, exact x2 loss scaling."""
    torch.manual_seed(0)
    m = get_model(stage2_model_cfg("moe", 8, dropout=0.0)).cuda()
    m.compute_dtype = torch.bfloat16
    m.train()
    batch = _batch()
    sd = copy.deepcopy(m.state_dict())
    o1, l1, g1 = _step(m, batch, (0.7, 0.3))
    m.load_state_dict(sd)
    o2, l2, g2 = _step(m, batch, (1.4, 0.6))
    assert o1[0].shape == (B, 8) and o1[1].shape == (B, 8, 2)
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)
    assert abs(l2.item() - 2 * l1.item()) <= 1e-6 * abs(l1.item())
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        assert (g2[k] - 2 * g1[k]).norm() <= 2e-5 * g1[k].norm() + 1e-12, k


def test_punet_expert_full_size_properties(tmp_path):
    """Config 4 (`punet`: frozen PU-Net -> 138-channel ResNet stem -> tanh head) at B=64, 256x256, bf16: deterministic
    forward, loss linear in its coefficients (exact factor 2 through every backward kernel), frozen PU-Net without
    gradients, train-mode BatchNorm buffers of the frozen U-Nets updated (4 passes of `unet`, F of `pred_unet`)."""
    from pmoe_amd.loss import punet_loss
    from tests.punet_util import build_product
    torch.manual_seed(0)
    m = build_product(tmp_path, dict(type="punet", n_experts=2, future_frames=6)).cuda()
    m.compute_dtype = torch.bfloat16
    m.train()
    images, speed, command, control, target = _batch()

    def step(coefs):
        m.zero_grad(set_to_none=True)
        a, s = m(images, speed, command)
        loss = punet_loss(a, s, control, target, list(coefs))
        loss.backward()
        return a.detach().clone(), s.detach().clone(), loss.detach().clone(), \
            {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}

    sd = copy.deepcopy(m.state_dict())
    a1, s1, l1, g1 = step((0.7, 0.3))
    nb = int(m.state_dict()["punet.unet.dwn_1.1.num_batches_tracked"]) - int(sd["punet.unet.dwn_1.1.num_batches_tracked"])
    nbp = int(m.state_dict()["punet.pred_unet.dwn_1.1.num_batches_tracked"]) - int(sd["punet.pred_unet.dwn_1.1.num_batches_tracked"])
    assert (nb, nbp) == (4, 6)
    m.load_state_dict(sd)
    a2, s2, l2, g2 = step((1.4, 0.6))
    assert a1.shape == (B, 2) and a1.abs().max() <= 1 and torch.equal(a1, a2) and torch.equal(s1, s2)
    assert abs(l2.item() - 2 * l1.item()) <= 1e-6 * abs(l1.item())
    assert g1 and not any(k.startswith("punet.") for k in g1), "the frozen PU-Net must not receive gradients"
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        assert (g2[k] - 2 * g1[k]).norm() <= 2e-5 * g1[k].norm() + 1e-12, k


def test_fp8_config_full_size_properties():
    """BASELINE config 5 at full size (E=4, B=128, 256x256, `fp8_weights=True`: e4m3 weights + activations on the block-scaled
    fp8 matrix instruction for the policy's convolutions, oracle/fp8_policy.py): bit-identical forward AND gradients on repeat,
    exact x2 scaling of every gradient with the loss coefficients (the quantisers use power-of-two scales, so a factor of 2
    commutes with every rounding on the path), and outputs within the EMULATED policy's own distance of the bf16 path's on
    the same weights and batch -- the bound is the sum of the worst emulated fp8-policy error and the worst emulated bf16
    error over the train-mode golden cases of tests/golden/bf16_bounds.pt (both are distances to float64)."""
    from tests.parity_util import GOLDEN, emul_worst
    Bc = 128
    g = torch.Generator().manual_seed(4321)
    images = torch.rand(Bc, 4, 3, S, S, generator=g).cuda()
    speed, target = torch.rand(Bc, 1, generator=g).cuda(), torch.rand(Bc, 1, generator=g).cuda()
    command = torch.nn.functional.one_hot(torch.randint(0, 6, (Bc,), generator=g), 6).float().cuda()
    control = (torch.rand(Bc, 2, generator=g) * 2 - 1).cuda()
    batch = (images, speed, command, control, target)
    m8 = _model()
    m8.fp8_weights = True
    sd = copy.deepcopy(m8.state_dict())
    o1, l1, g1 = _step(m8, batch, (0.7, 0.3))
    assert any(c.w_f8 is not None for c in m8._engine().all_convs), "the fp8 policy selected no convolution"
    m8.load_state_dict(sd)
    o1b, l1b, g1b = _step(m8, batch, (0.7, 0.3))
    m8.load_state_dict(sd)
    o2, l2, g2 = _step(m8, batch, (1.4, 0.6))
    for a, b, c in zip(o1, o1b, o2):
        assert torch.equal(a, b) and torch.equal(a, c)
    assert torch.equal(l1, l1b) and abs(l2.item() - 2 * l1.item()) <= 1e-6 * abs(l1.item())
    for k in g1:
        assert torch.isfinite(g1[k]).all(), k
        assert torch.equal(g1[k], g1b[k]), k
        assert torch.equal(g2[k], 2 * g1[k]), k
    del m8, g1b, g2
    mb = _model()
    mb.load_state_dict(sd)
    ob, lb, gb = _step(mb, batch, (0.7, 0.3))
    bounds = torch.load(GOLDEN / "bf16_bounds.pt", weights_only=False)["forward"]
    cases = [c for c, r in bounds.items() if "emul_fp8" in r and "eval" not in c]
    assert cases
    for name, a, b in zip(("probs", "mean", "std", "speeds"), o1, ob):
        lim = max(emul_worst(bounds[c]["emul_fp8"], name) for c in cases) + max(emul_worst(bounds[c]["emul"], name) for c in cases)
        err = (a - b).abs() / (1 + b.abs())
        print(f"C5 full size: fp8 vs bf16 {name}: max {err.max().item():.2e} median {err.median().item():.2e} (bound {lim:.2e})")
        assert err.max().item() <= lim and err.median().item() <= lim / 4, (name, err.max().item(), err.median().item(), lim)
    cos = sorted(torch.nn.functional.cosine_similarity(g1[k].flatten(), gb[k].flatten(), dim=0).item()
                 for k in gb if gb[k].numel() >= 1024)
    assert cos[len(cos) // 2] >= 0.8, cos[len(cos) // 2]           # straight-through gradients still point the bf16 path's way
