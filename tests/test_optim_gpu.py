"""Fused optimizer tail: pmoe_amd.optim against torch's own clip_grad_norm_ / Adam(amsgrad) /
AveragedModel on the same tensors, and the reference's 5-step H1 trajectory (train_2.py:149-165) from the goldens."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from pmoe_amd import optim  # noqa: E402
from tests.parity_util import GOLDEN, build_pair  # noqa: E402

SHAPES = [(64, 12, 3, 3), (64,), (1, 1, 3), (512, 1536), (5,), (70001,), (3, 16385), (512, 512, 3, 3)]


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter((torch.randn(s, generator=g) * 0.1).cuda()) for s in SHAPES]


@pytest.mark.parametrize("amsgrad,wd,max_norm", [(True, 0.0, 1.0), (False, 0.01, 0.0), (True, 0.05, 1e4)])
def test_fused_adam_and_clip_match_torch(amsgrad, wd, max_norm):
    ref, got = _params(0), _params(0)
    o_ref = torch.optim.Adam(ref, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, amsgrad=amsgrad)
    o_got = optim.FusedAdam(got, lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=wd, amsgrad=amsgrad)
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        for a, b in zip(ref, got):
            gr = (torch.randn(a.shape, generator=g) * (3.0 if step % 2 else 0.01)).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        if step == 3:                      # a parameter without gradient is skipped and keeps its own step count
            ref[1].grad = got[1].grad = None
        if max_norm > 0:
            n_ref = torch.nn.utils.clip_grad_norm_(ref, max_norm)
            if step % 2:                   # (a) clip in place like torch, (b) fused into the update
                n_got = optim.clip_grad_norm_(got, max_norm)
                for a, b in zip(ref, got):
                    if a.grad is not None:
                        torch.testing.assert_close(b.grad, a.grad, rtol=2e-6, atol=1e-9)
                o_got.step()
            else:
                n_got = optim.clip_grad_norm_(got, max_norm, scale=False)
                o_got.step(clip=n_got)
            assert n_got.is_cuda and abs(n_got.item() - n_ref.item()) <= 2e-6 * n_ref.item()
        else:
            o_got.step()
        o_ref.step()
        for a, b in zip(ref, got):
            torch.testing.assert_close(b, a, rtol=1e-5, atol=2e-7)
    for a, b in zip(ref, got):
        for k in ("exp_avg", "exp_avg_sq") + (("max_exp_avg_sq",) if amsgrad else ()):
            want = o_ref.state[a][k]
            torch.testing.assert_close(o_got.state[b][k], want, rtol=1e-4, atol=1e-5 * want.abs().max().item())
        assert float(o_got.state[b]["step"]) == float(o_ref.state[a]["step"])
    # checkpoint interchange: the fused optimizer's state loads into torch's Adam and vice versa
    o_ref.load_state_dict(o_got.state_dict())
    o_got.load_state_dict(o_ref.state_dict())


def test_fused_averaged_model_matches_torch():
    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.ps = torch.nn.ParameterList(_params(3))
    m = M()
    a_ref, a_got = torch.optim.swa_utils.AveragedModel(m), optim.FusedAveragedModel(m)
    g = torch.Generator().manual_seed(2)
    for _ in range(4):
        with torch.no_grad():
            for p in m.parameters():
                p.add_((torch.randn(p.shape, generator=g) * 0.05).cuda())
        a_ref.update_parameters(m)
        a_got.update_parameters(m)
        for x, y in zip(a_ref.module.parameters(), a_got.module.parameters()):
            torch.testing.assert_close(y, x, rtol=1e-6, atol=1e-8)
    assert int(a_got.n_averaged) == int(a_ref.n_averaged) == 4


def test_optim_rejects_cpu_tensors():
    p = torch.nn.Parameter(torch.zeros(4))
    p.grad = torch.ones(4)
    with pytest.raises(RuntimeError, match="no CPU path"):
        optim.clip_grad_norm_([p], 1.0)
    with pytest.raises(RuntimeError, match="no CPU path"):
        optim.FusedAdam([p], lr=1e-3).step()


def test_h1_trajectory_matches_reference():
    """The reference's own step recipe (train_2.py:149-165) on golden g1: forward, moe_loss, zero_grad, backward,
    clip_grad_norm_(1.0), Adam(lr 2e-4, amsgrad) -- 5 steps, loss and gradient norm of every step and the parameter
    norms afterwards, all produced by the imported reference (tests/golden/g1 'h1')."""
    from pmoe_amd.loss import moe_loss
    g = torch.load(GOLDEN / "g1_moe_e4_b2_128.pt", weights_only=False)
    ocfg, _, model, inp = build_pair(g, torch.float32)
    dev = {k: v.cuda() for k, v in inp.items()}
    opt = optim.FusedAdam([p for p in model.parameters() if p.requires_grad], lr=2e-4, betas=(0.9, 0.999), eps=1e-8,
                          weight_decay=0, amsgrad=True)
    # two f32 implementations of a chaotic 20-layer network drift apart step by step (the lr 2e-4 Adam steps move the loss
    # from 1.9 to 5.4 and back): the bound grows with the step index
    for k_step, ref in enumerate(g["h1"]["traj"]):
        dist, speeds = model(dev["images"], dev["speed"], dev["command"])
        loss = moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs)
        opt.zero_grad()
        loss.backward()
        gn = optim.clip_grad_norm_(model.parameters(), 1.0, scale=False)
        opt.step(clip=gn)
        assert loss.item() == pytest.approx(ref["loss"], rel=1e-3 * (1 + k_step)), (loss.item(), ref)
        assert gn.item() == pytest.approx(ref["grad_norm"], rel=1e-2 * (1 + k_step)), (gn.item(), ref)
    named = dict(model.named_parameters())
    for k, v in g["h1"]["param_l2"].items():
        assert named[k].norm().item() == pytest.approx(v, rel=1e-4), k
