"""PU-Net / PMoE model types on cuda:0 through the C-ABI kernels, against the
golden vectors of the imported reference and the live CPU oracle (tests/punet_parity.py explains the conditioning-
aware bounds of the chaotic train-mode cases)."""
import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.punet_parity import (GOLDEN, build_pair, run_pmoe_case, run_punet_case, run_punet_forced,  # noqa: E402
                                 run_punet_per_pass, run_punet_teacher_forced_bf16)


def test_punet_eval_parity_f32(tmp_path):
    """B=1, eval mode, 4 past + 6 predicted frames (the agent's configuration): plain 1e-4 bound."""
    run_punet_case(tmp_path, "p2_punet_b1_64_f6_eval", torch.float32, strict=True)


def test_punet_eval_parity_bf16(tmp_path):
    run_punet_case(tmp_path, "p2_punet_b1_64_f6_eval", torch.bfloat16)


def test_punet_inter_train_parity_f32(tmp_path):
    """punet_inter (PU-Net bottleneck vector as the image feature): well conditioned -> plain 1e-4 forward bound,
    every trainable gradient within 5e-3 of the oracle, reference gradient slices and BN buffers."""
    r = run_punet_case(tmp_path, "p3_punetinter_b2_64_f2", torch.float32, strict=True)
    assert r["grad_median_rel_l2"] <= 1e-3 and r["grad_worst"][0] <= 5e-3, r
    assert r["golden_slices_worst"] <= 5e-3 and r["bn_running_worst"] <= 1e-4, r


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_punet_train_chain_per_pass_teacher_forced(tmp_path, dtype):
    """each of the T + F train-mode U-Net passes of the frozen PU-Net on the float64 oracle's inputs for
    that pass (batch-of-8 golden p6) -- the tight test of the kernels config 4 spends 97 % of its time in.  f32: every logit
    within 1e-4 x (1 + |ref|), BatchNorm running buffers within 1e-4.  bf16: within 1.25 x the bf16-storage-emulating CPU
    oracle's own error on the same pass inputs (max and rms metric; that emulation is 0.3-0.4 off float64 per PASS in the max
    metric, so the flat 1e-2 is unreachable for any bf16-storage implementation and is only printed)."""
    run_punet_per_pass(tmp_path, "p6_punet_b8_96_f2", dtype)


@pytest.mark.parametrize("name", ["p4_punet_b3_96_f3", "p6_punet_b8_96_f2"])          # (p1, the smaller sibling, runs in bf16 below)
def test_punet_train_parity_f32(tmp_path, name):
    """PUNetExpert with the 138/69-channel ResNet stem in train mode: forward within 5x the f32 oracle's own drift from
    float64, the typical gradient tensor within 4x that drift, directions and total norm preserved, frozen PU-Net
    parameters without gradients, BN buffers (4 updates per step for `unet`, F for `pred_unet`) as in the reference."""
    r = run_punet_case(tmp_path, name, torch.float32)
    assert r["grad_cond_median"] <= 1.0, r
    assert r["grad_median_cos"] >= 0.9 and r["grad_total_rel"] <= 2e-2, r
    assert r["bn_running_worst"] <= max(1e-4, 5 * r["f32_oracle_drift"]), r


def test_punet_train_bf16(tmp_path):
    """bf16 storage through 6-7 chained train-mode U-Nets decorrelates the tiny-batch cases end to end (the f32 oracle already
    drifts 4e-4 from f64): the forward of p3 / p1 is held to 1.25 x the measured error of the bf16-storage-emulating oracle
    (tests/golden/bf16_bounds.pt), the well-conditioned punet_inter case also to aligned gradients.  What the bf16 KERNELS of
    the trainable half do is pinned by the teacher-forced test below."""
    r = run_punet_case(tmp_path, "p3_punetinter_b2_64_f2", torch.bfloat16)
    assert r["grad_median_cos"] >= 0.9 and r["grad_total_rel"] <= 0.2, r
    run_punet_case(tmp_path, "p1_punet_b2_64_f2", torch.bfloat16)
    # batch of 8 (round 3's p6): the chain is still chaotic end to end in bf16 (emulation's worst draw 0.37 on the actions);
    # what pins the bf16 U-Net kernels is test_punet_train_chain_per_pass_teacher_forced, the trainable half the test below
    run_punet_case(tmp_path, "p6_punet_b8_96_f2", torch.bfloat16)


@pytest.mark.parametrize("name", ["p1_punet_b2_64_f2", "p4_punet_b3_96_f3", "p6_punet_b8_96_f2"])
def test_punet_train_f32_forced(tmp_path, name):
    """p1 in f32, and the batch-of-8 case p6.  The float64 oracle receives the HIP path's own predicted
    masks and ReLU / max-pool decisions (tests/punet_parity.py:run_punet_forced): actions and speeds within 1e-4, EVERY
    trainable gradient tensor within 1e-3, every decision disagreement a near-tie."""
    r = run_punet_forced(tmp_path, name)
    assert r["grad_worst"][0] <= 1e-3, r


@pytest.mark.parametrize("name", ["p1_punet_b2_64_f2", "p6_punet_b8_96_f2"])
def test_punet_train_bf16_teacher_forced(tmp_path, name):
    """VERDICT r2 item 2c (the old p1 assertion passed at a median gradient cosine of 0.027): with the oracle's predicted masks
    fed to the HIP backbone the chained U-Nets are out of the loop, and the bf16 trainable half must produce the oracle's
    actions within the bf16 tolerance and gradients that point the oracle's way."""
    r = run_punet_teacher_forced_bf16(tmp_path, name)
    assert r["actions"] <= 3e-2 and r["speeds"] <= 3e-2, r
    assert r["grad_median_cos"] >= 0.9 and r["grad_total_rel"] <= 0.2, r


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pmoe_parity(tmp_path, dtype):
    """PMoE (moe.py:326-363): frozen mixture + PU-Net expert + lat/long blend on the reference's own draw."""
    run_pmoe_case(tmp_path, "p5_pmoe_e2_b2_64_f2", dtype)      # bf16: measured emulation bounds (tests/punet_parity.py)


def test_punet_stem_fold_matches_explicit_path(tmp_path):
    """138-channel stem: conv1 / eca1 gradients from per-image filter gradients vs the explicit dgrad + ECA backward
    (same forward, so this is tight even though the case itself is ill conditioned)."""
    from pmoe_amd.loss import punet_loss
    g = torch.load(GOLDEN / "p4_punet_b3_96_f3.pt", weights_only=False)
    grads = []
    for fold in (True, False):
        _, _, model, inp = build_pair(tmp_path, g, torch.float32)
        model._engine().fold_stem_input = fold
        dev = {k: v.cuda() for k, v in inp.items()}
        a, s = model(dev["images"], dev["speed"], dev["command"])
        punet_loss(a, s, dev["control"], dev["target_speed"], [0.7, 0.3]).backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if "backbone.conv1.layer1" in k})
    assert len(grads[0]) == 4
    for k in grads[0]:
        e = ((grads[0][k] - grads[1][k]).norm() / (grads[1][k].norm() + 1e-20)).item()
        assert e <= 1e-3, (k, e)


def test_punet_contract(tmp_path):
    """frame-count assertion (punet.py:84-86), unfrozen PU-Net refused, sample() == forward()[0], deepcopy."""
    import copy
    g = torch.load(GOLDEN / "p1_punet_b2_64_f2.pt", weights_only=False)
    _, _, model, inp = build_pair(tmp_path, g, torch.float32)
    dev = {k: v.cuda() for k, v in inp.items()}
    with pytest.raises(AssertionError, match="past frames"):
        model(dev["images"][:, :3], dev["speed"], dev["command"])
    model.eval()
    with torch.no_grad():
        a = model.sample(dev["images"], dev["speed"], dev["command"])
        b, s = copy.deepcopy(model)(dev["images"], dev["speed"], dev["command"])
    assert torch.equal(a, b) and s.shape == (2, 1)
    model.punet.unet.out.bias.requires_grad_(True)
    with pytest.raises(NotImplementedError, match="frozen"):
        model(dev["images"], dev["speed"], dev["command"])


def test_punet_fused_forward_paths_match_the_unfused_ones(tmp_path):
    """Round 4's fused forward paths of the frozen U-Nets -- BatchNorm + ReLU applied on load by the second convolution of a
    64-channel block and by the 1x1 layers behind a block (PMOE_RES_INBN), MaxPool2d written by the BatchNorm pass, ConvTranspose2d
    scattering its own 2x2 blocks --
    against the engine with each switch off: same kernels' arithmetic, so actions, speed and every BatchNorm running buffer of a
    train-mode step are BIT-identical (bf16, batch 4, 128 x 128, T = 4 past + F = 2 predicted frames)."""
    from oracle import weights as W
    g = torch.load(GOLDEN / "p6_punet_b8_96_f2.pt", weights_only=False)
    _, _, model, _ = build_pair(tmp_path, g, torch.bfloat16)
    inp = W.make_inputs(4, 128, 128, seed=11)            # (power-of-two sides: the fused ConvTranspose2d store serves those)
    eng = model._engine()
    args = [inp[k].cuda() for k in ("images", "speed", "command")]
    state0 = {k: v.clone() for k, v in model.state_dict().items()}

    def run(**switches):
        model.load_state_dict(state0)
        for k, v in switches.items():
            assert hasattr(type(eng), k), k
            setattr(eng, k, v)
        try:
            with torch.no_grad():
                act, sp = model(*args)
            torch.cuda.synchronize()
            return act.clone(), sp.clone(), {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
        finally:
            for k in switches:
                delattr(eng, k)                      # back to the class default

    ref = run()
    import pmoe_amd.ops as ops
    ops.profile_begin()
    run()
    names = [meta.get("kernel") for name, meta, _ in ops.profile_end() if name == "conv2d"]
    # the fused kernels really ran: BatchNorm + ReLU on load in the 64-channel 3x3 kernel (1267) and in the 1x1 direct kernel, plain
    # (1412 | 1414: the classifier) and with the ConvTranspose2d scatter (1462 | 1464)
    assert 1267 in names and any(c in names for c in (1412, 1414)) and any(c in names for c in (1462, 1464)), sorted(set(map(str, names)))
    for sw in ("fuse_in_bn", "fuse_in_bn_1x1", "fuse_bn_pool", "fuse_upconv_shuffle"):
        got = run(**{sw: False})
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), sw
        assert got[2].keys() == ref[2].keys() and all(torch.equal(got[2][k], ref[2][k]) for k in ref[2]), sw
    # the entry block's ECA gates folded into per-image weights: the gate multiplies the bf16 WEIGHTS instead of the bf16 activation,
    # a different rounding of the same product -- one entry block (first predicted frame) within bf16 noise of the explicit path
    # (the autoregressive roll-out then amplifies it like any other bf16 perturbation: tests/punet_parity.py)
    eng.debug_pass_out = []
    try:
        run()
        fold = [t.float().clone() for t in eng.debug_pass_out]
        eng.debug_pass_out = []
        run(fold_entry_eca=False)
        expl = [t.float().clone() for t in eng.debug_pass_out]
    finally:
        eng.debug_pass_out = None
    T = inp["images"].shape[1]
    assert len(fold) == len(expl) > T
    for t in range(T):                                      # the past frames do not pass the entry block
        assert torch.equal(fold[t], expl[t]), t
    # (yardstick: ONE train-mode bf16 U-Net pass is 0.04-0.05 rms / 0.3-0.5 max off float64 in the metric |d| / (1 + |ref|),
    #  tests/punet_parity.py:run_punet_per_pass -- two bf16 realisations of the same pass differ by about as much)
    rel = (fold[T] - expl[T]).abs() / (1 + expl[T].abs())
    print("entry-block ECA fold vs explicit path, first predicted frame: rms %.3e max %.3e" % (rel.pow(2).mean().sqrt().item(), rel.max().item()))
    assert rel.pow(2).mean().sqrt().item() <= 0.1 and rel.max().item() <= 1.0, (rel.pow(2).mean().sqrt().item(), rel.max().item())


def test_punet_engine_packs_its_weights_once(tmp_path):
    """The packed bf16 weight layouts are cached on the parameters' version counters: a second step over unchanged weights must not
    launch a single pack kernel (until round 4 the PU-Net engine rebuilt its pointer tables, and with them all 79 packs, every step)."""
    import pmoe_amd.ops as ops
    from pmoe_amd.loss import punet_loss
    g = torch.load(GOLDEN / "p1_punet_b2_64_f2.pt", weights_only=False)
    _, _, model, inp = build_pair(tmp_path, g, torch.bfloat16)
    args = [inp[k].cuda() for k in ("images", "speed", "command")]
    counts = []
    for _ in range(3):
        model.zero_grad(set_to_none=True)
        ops.profile_begin()
        act, sp = model(*args)
        punet_loss(act, sp, inp["control"].cuda(), inp["target_speed"].cuda(), [0.7, 0.3]).backward()
        counts.append(sum(1 for n, _, _ in ops.profile_end() if n.startswith("pack_conv_weights") and "gated" not in n))
    assert counts[0] > 0 and counts[1] == 0 and counts[2] == 0, counts
    # ... and an optimizer step invalidates them
    with torch.no_grad():
        for p_ in model.parameters():
            if p_.requires_grad:
                p_.add_(0.0)
    ops.profile_begin()
    model(*args)
    assert sum(1 for n, _, _ in ops.profile_end() if n.startswith("pack_conv_weights") and "gated" not in n) > 0
