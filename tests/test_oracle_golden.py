"""The CPU oracle (oracle/pmoe_oracle.py) against golden vectors produced by the imported
reference (oracle/make_golden.py).  This is what pins the oracle; tolerance is fp32 round-off of
two CPU implementations of the same graph (bit-equal in practice)."""
import pytest
import torch

from oracle import pmoe_oracle as O
from oracle import weights as W


def _load(golden_dir, name):
    return torch.load(golden_dir / f"{name}.pt", weights_only=False)


def _run(g):
    m = g["meta"]
    cfg = O.stage2_cfg(m["type"], m["n_experts"], dropout=0.0)
    model = O.get_model(cfg)
    W.fill_state_dict(model, seed=m["weight_seed"])
    model.train(m["train"])
    inp = W.make_inputs(m["batch"], m["size"], m["size"], seed=m["input_seed"])
    return cfg, model, inp


CASES = ["g1_moe_e4_b2_128", "g3_moe_e8_b2_128", "g4_moealt_e4_b2_64", "g5_moe_e3_b3_96", "g10_moe_e4_b32_64", "g6_moeshared_k4_b6_96",
         "g8_moeshared_k3_b4_128", "g9_moeshared_k5_b8_64", "g11_moe_e4_b8_128"]


@pytest.mark.parametrize("name", CASES + ["g2_moe_e4_b1_224_eval", "g7_moeshared_k6_b1_224_eval"])
def test_state_dict_layout_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    cfg, model, _ = _run(g)
    sd = model.state_dict()
    assert list(sd.keys()) == g["state_dict_keys"]
    assert [tuple(v.shape) for v in sd.values()] == g["state_dict_shapes"]


@pytest.mark.parametrize("name", CASES)
def test_train_forward_backward_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    cfg, model, inp = _run(g)
    dist, speeds = model(inp["images"], inp["speed"], inp["command"])
    loss = O.moe_loss(dist, speeds, inp["control"], inp["target_speed"], cfg.loss_coefs)
    loss.backward()
    tol = dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dist.mixture_distribution.probs, g["probs"], **tol)
    torch.testing.assert_close(dist.component_distribution.base_dist.loc, g["mean"], **tol)
    torch.testing.assert_close(dist.component_distribution.base_dist.scale, g["std"], **tol)
    torch.testing.assert_close(speeds, g["speeds"], **tol)
    torch.testing.assert_close(loss.detach(), g["loss"], **tol)
    torch.testing.assert_close(dist.log_prob(inp["control"]).detach(), g["log_prob"], **tol)
    named = dict(model.named_parameters())
    for k, n in g["grad_norms"].items():
        assert named[k].grad.norm().item() == pytest.approx(n, rel=1e-4, abs=1e-7), k
    for k, sl in g["grad_slices"].items():
        torch.testing.assert_close(named[k].grad.flatten()[:64], sl, rtol=1e-4, atol=1e-6)
    sd = model.state_dict()
    for k, v in g["bn_after_1"].items():
        torch.testing.assert_close(sd[k], v, rtol=1e-5, atol=1e-6)
    # explicit mixture log-likelihood formula == torch.distributions
    ll = O.mixture_nll_explicit(dist.mixture_distribution.probs, g["mean"], g["std"], inp["control"])
    torch.testing.assert_close(ll.detach(), g["log_prob"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("name", ["g2_moe_e4_b1_224_eval", "g7_moeshared_k6_b1_224_eval"])
def test_eval_forward_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    cfg, model, inp = _run(g)
    with torch.no_grad():
        dist, speeds = model(inp["images"], inp["speed"], inp["command"])
    torch.testing.assert_close(dist.mixture_distribution.probs, g["probs"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(dist.component_distribution.base_dist.loc, g["mean"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(speeds, g["speeds"], rtol=1e-5, atol=1e-6)
    assert model.sample(inp["images"], inp["speed"], inp["command"]).shape == (1, 2)


@pytest.mark.parametrize("name", ["g1_moe_e4_b2_128", "g6_moeshared_k4_b6_96"])
def test_h1_step_trajectory_matches_reference(golden_dir, name):
    """Caller row H1 (train_2.py:149-165): steps of fwd / moe_loss / backward / clip 1.0 / Adam(amsgrad)."""
    g = _load(golden_dir, name)
    cfg, model, inp = _run(g)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4, betas=(0.9, 0.999), eps=1e-8, amsgrad=True)
    for ref in g["h1"]["traj"]:
        dist, speeds = model(inp["images"], inp["speed"], inp["command"])
        loss = O.moe_loss(dist, speeds, inp["control"], inp["target_speed"], cfg.loss_coefs)
        opt.zero_grad()
        loss.backward()
        gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        opt.step()
        assert loss.item() == pytest.approx(ref["loss"], rel=2e-4)
        assert float(gn) == pytest.approx(ref["grad_norm"], rel=2e-3)
    named = dict(model.named_parameters())
    for k, v in g["h1"]["param_l2"].items():
        assert named[k].norm().item() == pytest.approx(v, rel=1e-5)


def test_micro_cases(golden_dir):
    g = _load(golden_dir, "micro")
    for c, k in g["eca_k"].items():
        assert O.eca_kernel_size(c) == k
    for (bn, p, dims), keys in g["mlp_layouts"].items():
        assert list(O.make_mlp(list(dims), "relu", False, bn, p).state_dict().keys()) == keys
    eca = O.EfficientBlock(64)
    with torch.no_grad():
        eca.conv.weight.copy_(torch.tensor([[[0.3, -0.7, 0.5]]]))
    x = torch.randn(2, 64, 5, 7, generator=torch.Generator().manual_seed(g["eca_x_seed"]))
    torch.testing.assert_close(eca(x), g["eca_y"], rtol=1e-6, atol=1e-6)
    lc = g["loss_case"]
    import torch.distributions as D
    dist = D.MixtureSameFamily(D.Categorical(lc["probs"]), D.Independent(D.Normal(lc["mean"], lc["std"]), 1))
    torch.testing.assert_close(O.moe_loss(dist, lc["speeds"], lc["act"], lc["tgt"], [0.7, 0.3]), lc["loss"])
    ls = g["loss_case_shared"]       # [B,1] speed prediction of MixtureOfExpertsShared (loss.py:129-130)
    torch.testing.assert_close(O.moe_loss(dist, ls["speeds"], ls["act"], ls["tgt"], [0.7, 0.3]), ls["loss"])


PUNET_CASES = ["p1_punet_b2_64_f2", "p3_punetinter_b2_64_f2", "p6_punet_b8_96_f2"]


def _punet_oracle(g):
    m = g["meta"]
    cfg = O.stage2_cfg(m["type"], m["n_experts"], dropout=0.0, future_frames=m["future_frames"],
                       exclude_freeze=["lat_weights", "long_weights"] if m["type"] == "pmoe" else [])
    model = O.get_model(cfg)
    W.fill_state_dict(model, seed=m["weight_seed"])
    model.train(m["train"])
    return cfg, model, W.make_inputs(m["batch"], m["size"], m["size"], seed=m["input_seed"])


@pytest.mark.parametrize("name", PUNET_CASES + ["p2_punet_b1_64_f6_eval", "p4_punet_b3_96_f3", "p5_pmoe_e2_b2_64_f2"])
def test_punet_state_dict_layout_matches_reference(golden_dir, name):
    g = _load(golden_dir, name)
    _, model, _ = _punet_oracle(g)
    sd = model.state_dict()
    assert list(sd.keys()) == g["state_dict_keys"]
    assert [tuple(v.shape) for v in sd.values()] == g["state_dict_shapes"]
    assert {k: p.requires_grad for k, p in model.named_parameters()} == g["requires_grad"]


@pytest.mark.parametrize("name", PUNET_CASES)
def test_punet_train_matches_reference(golden_dir, name):
    """PredictiveUnet / PUNetExpert restatement (punet.py:75-120, moe.py:268-323) vs the imported reference."""
    g = _load(golden_dir, name)
    cfg, model, inp = _punet_oracle(g)
    a, s = model(inp["images"], inp["speed"], inp["command"])
    loss = O.punet_loss(a, s, inp["control"], inp["target_speed"], cfg.loss_coefs)
    loss.backward()
    tol = dict(rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(a, g["actions"], **tol)
    torch.testing.assert_close(s, g["speeds"], **tol)
    torch.testing.assert_close(loss.detach(), g["loss"], **tol)
    named = dict(model.named_parameters())
    for k, sl in g["grad_slices"].items():
        torch.testing.assert_close(named[k].grad.flatten()[:64], sl, rtol=1e-4, atol=1e-6)
    assert {k for k, p in named.items() if p.grad is not None} == set(g["grad_norms"])
    sd = model.state_dict()
    for k, v in g["bn_after_1"].items():
        torch.testing.assert_close(sd[k], v, rtol=1e-5, atol=1e-6)


def test_punet_eval_matches_reference(golden_dir):
    g = _load(golden_dir, "p2_punet_b1_64_f6_eval")
    _, model, inp = _punet_oracle(g)
    with torch.no_grad():
        a, s = model(inp["images"], inp["speed"], inp["command"])
        masks = model.punet(inp["images"])[:, :, :, ::8, ::8]
    torch.testing.assert_close(a, g["actions"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(s, g["speeds"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(masks, g["punet_masks_eval"], rtol=1e-4, atol=1e-5)


def test_pmoe_matches_reference(golden_dir):
    """PMoE (moe.py:326-363) under the reference's sampling seed: same draw, same blended actions, same gradients."""
    g = _load(golden_dir, "p5_pmoe_e2_b2_64_f2")
    cfg, model, inp = _punet_oracle(g)
    torch.manual_seed(g["meta"]["sample_seed"])
    actions, dummy = model(inp["images"], inp["speed"], inp["command"])
    assert dummy == -1
    loss = O.pmoe_loss(actions, dummy, inp["control"], inp["target_speed"], cfg.loss_coefs)
    loss.backward()
    torch.testing.assert_close(actions, g["actions"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(loss.detach(), g["loss"], rtol=1e-5, atol=1e-6)
    named = dict(model.named_parameters())
    for k, ref in g["grads_small"].items():
        torch.testing.assert_close(named[k].grad, ref, rtol=1e-4, atol=1e-6)
    assert {k for k, p in named.items() if p.grad is not None} == set(g["grad_norms"])
    ac = _load(golden_dir, "micro")["action_loss_case"]
    torch.testing.assert_close(O.punet_loss(ac["actions"], ac["speeds"], ac["act"], ac["tgt"], [0.7, 0.3]), ac["punet_loss"])
    torch.testing.assert_close(O.pmoe_loss(ac["actions"], -1, ac["act"], ac["tgt"], [0.7, 0.3]), ac["pmoe_loss"])


def _stage1_oracle(g, dtype=torch.float32):
    m = g["meta"]
    model = O.PredictiveUnet(past_frames=4, future_frames=m["future_frames"], model_path=None)
    W.fill_state_dict(model, seed=m["weight_seed"])
    model.train()            # train_1.py:122: model.train() also reaches the frozen ``unet``
    model.to(dtype)
    inp = W.make_inputs(m["batch"], m["size"], m["size"], seed=m["input_seed"])
    tgt = W.make_seg_targets(m["batch"], m["future_frames"], m["size"], m["size"], 23, seed=m["target_seed"])
    return model, inp["images"].to(dtype), tgt


@pytest.mark.parametrize("name", ["s1_stage1_b3_32_f3", "s2_stage1_b8_32_f2"])
def test_stage1_training_step_matches_reference(golden_dir, name):
    """Stage-1 PU-Net training (train_1.py:129-141, punet.py:75-120, loss.py:86-118): restatement vs imported reference."""
    g = _load(golden_dir, name)
    model, images, tgt = _stage1_oracle(g)
    assert list(model.state_dict().keys()) == g["state_dict_keys"]
    assert {k: p.requires_grad for k, p in model.named_parameters()} == g["requires_grad"]
    out = model(images)
    out.retain_grad()
    loss = O.AutoregressiveCriterion(g["meta"]["future_frames"], "tversky")(out, tgt)
    loss.backward()
    torch.testing.assert_close(out.detach()[..., ::4, ::4], g["out_sub"], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(loss.detach(), g["loss"], rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(out.grad[..., ::4, ::4], g["dout_sub"], rtol=1e-3, atol=1e-8)
    named = dict(model.named_parameters())
    assert {k for k, p in named.items() if p.grad is not None} == set(g["grad_norms"])
    for k, sl in g["grad_slices"].items():
        scale = g["grad_norms"][k] / max(1.0, named[k].numel() ** 0.5)
        torch.testing.assert_close(named[k].grad.flatten()[:64], sl, rtol=2e-3, atol=2e-3 * scale + 1e-9)
    sd = model.state_dict()
    for k, v in g["bn_after_1"].items():
        torch.testing.assert_close(sd[k], v, rtol=1e-4, atol=1e-5)


def test_segmentation_criterion_matches_reference(golden_dir):
    """AutoregressiveCriterion (loss.py:86-118; class_dice 6-17, tversky_loss 34-45, weighted CE 48-57)."""
    cases = _load(golden_dir, "s0_segloss")
    for nm, c in cases.items():
        f = c["logits"].shape[1]
        for lt in ("tversky", "l1", "l2"):
            x = c["logits"].clone().requires_grad_(True)
            loss = O.AutoregressiveCriterion(f, lt)(x, c["target"])
            loss.backward()
            torch.testing.assert_close(loss.detach(), c[lt]["loss"], rtol=1e-5, atol=1e-6)
            torch.testing.assert_close(x.grad, c[lt]["dlogits"], rtol=1e-4, atol=1e-9)
    with pytest.raises(ValueError):
        O.AutoregressiveCriterion(1, "huber")


def test_bf16_storage_emulation_variants():
    """oracle/bf16_emulation.py, the yardstick of the bf16 parity tests: each variant changes WHERE values are rounded to bf16, never
    the arithmetic -- all three stay within bf16 noise of the float64 block, the "folded" one (ECA gates multiplied into per-image
    weights, rounded once) reproduces the gate algebra exactly when nothing is rounded, and it leaves the convolution's master
    weights in f32."""
    import copy

    from oracle import bf16_emulation as EM
    from oracle import pmoe_oracle as O
    torch.manual_seed(0)
    blk = O.EfficientConvBlock(23, 3)
    blk.train()
    x = torch.randn(3, 23, 32, 32)
    ref = copy.deepcopy(blk).double()(x.double())
    errs = {}
    for v in ("fused", "all", "folded"):
        b = copy.deepcopy(blk)
        EM.emulate_bf16(b, v)
        with torch.no_grad():
            errs[v] = EM.metric(b(x.to(torch.bfloat16).float()), ref)
    assert all(1e-4 < e < 5e-2 for e in errs.values()), errs           # bf16 noise: neither exact nor broken
    assert max(errs.values()) < 3 * min(errs.values()), errs           # the variants are the same order of magnitude
    # the fold itself is exact algebra: with the rounding switched off it equals the plain block to f32 accuracy
    b = copy.deepcopy(blk)
    orig = EM._round
    EM._round = lambda t: t
    try:
        EM._fold_gates(b)
        with torch.no_grad():
            out = b(x)
    finally:
        EM._round = orig
    assert EM.metric(out, ref) < 1e-5
    b = copy.deepcopy(blk)
    w0 = b.layer1.conv1[0].weight.detach().clone()
    EM.emulate_bf16(b, "folded")
    assert torch.equal(b.layer1.conv1[0].weight, w0)                    # master weights stay f32: W * g[n] is what gets rounded
