"""End-to-end parity (-m gpu): the drop-in MixtureOfExperts on cuda:0 through the C-ABI kernels versus
the golden vectors of the imported reference and the live CPU oracle (tests/parity_util.py)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

from tests.parity_util import GOLDEN, build_pair, rel_err, run_parity_case  # noqa: E402


@pytest.mark.parametrize("name", ["g4_moealt_e4_b2_64", "g5_moe_e3_b3_96", "g1_moe_e4_b2_128", "g3_moe_e8_b2_128",
                                  "g10_moe_e4_b32_64", "g11_moe_e4_b8_128", "g6_moeshared_k4_b6_96", "g8_moeshared_k3_b4_128",
                                  "g9_moeshared_k5_b8_64"])
def test_train_parity_f32(name):
    """Exact-f32 path: forward within north_star's 1e-4 of the reference's golden vectors; EVERY parameter gradient of EVERY
    expert within parity_util.FORCED_GRAD_TOL of the float64 oracle evaluated on the HIP path's own ReLU / max-pool decisions
    (tests/forced_masks.py), every decision that differs from the float64 oracle's own a near-tie (|pre-activation| <=
    FLIP_ZONE), and the reference's golden gradient slices on the experts without any.  Round 2's rule ("half of the experts
    may be off by 15 %: ReLU flips") is gone: flips are taken out of the comparison, not allowed for."""
    r = run_parity_case(name, torch.float32, check_grads=True)
    assert r["grad_worst"][0] <= 1e-3, r


@pytest.mark.parametrize("name", ["g4_moealt_e4_b2_64", "g5_moe_e3_b3_96", "g1_moe_e4_b2_128", "g6_moeshared_k4_b6_96",
                                  "g11_moe_e4_b8_128"])
def test_train_parity_bf16(name):
    """bf16 path end to end: every output within max(1e-2, 1.25 x the bf16-storage-emulating oracle's worst of 14 draws) of
    the float64 oracle (tests/golden/bf16_bounds.pt), the measured error printed beside north_star's flat 1e-2 and the
    emulation's.  g11 (batch of 8, 128x128: every BatchNorm statistic over >= 128 values) is the best conditioned case, and
    even there the EMULATION is 2.1e-2 off on `mean` (5.5e-3 / 1.1e-2 / 1.4e-2 on probs / std / speeds): the flat 1e-2 is
    met by the outputs whose emulated error is below it, not by all four."""
    r = run_parity_case(name, torch.bfloat16, check_grads=True)
    flat = {k[:-4]: ("%.1e" % v[0], "meets 1e-2" if v[0] <= 1e-2 else "beyond 1e-2 (emulation: %.1e)" % v[1])
            for k, v in r.items() if k.endswith("_abs")}
    print(name, "bf16 vs float64, flat 1e-2:", flat)


def test_train_parity_bf16_realistic_batch():
    """B=32 batch statistics (g10): like every bf16 case, each output within 1.25 x the bf16-storage-emulating oracle's own
    distance from the float64 oracle (tests/golden/bf16_bounds.pt; 3.6e-2 on `mean` here)."""
    run_parity_case("g10_moe_e4_b32_64", torch.bfloat16, check_grads=True)


@pytest.mark.parametrize("name", ["g2_moe_e4_b1_224_eval", "g7_moeshared_k6_b1_224_eval"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eval_parity_agent_shape(dtype, name):
    """image_agent.py:158-159: B=1, 224x224, eval mode, model.sample()."""
    # eval mode uses the (synthetic, mismatched) running statistics: activations are not re-normalised and the outputs
    # reach |8|; bf16 is held to 1.25 x the bf16-emulating oracle's measured error like the train-mode cases
    run_parity_case(name, dtype, check_grads=False)
    g = torch.load(GOLDEN / f"{name}.pt", weights_only=False)
    _, _, model, inp = build_pair(g, dtype)
    with torch.no_grad():
        a = model.sample(inp["images"].cuda(), inp["speed"].cuda(), inp["command"].cuda())
    assert a.shape == (1, 2) and torch.isfinite(a).all()


def test_fused_and_unfused_bn_stats_agree():
    g = torch.load(GOLDEN / "g5_moe_e3_b3_96.pt", weights_only=False)
    outs = []
    for fuse in (True, False):
        _, _, model, inp = build_pair(g, torch.float32)
        model._engine().fuse_conv_stats = fuse
        with torch.no_grad():
            outs.append(model.mixture_params(inp["images"].cuda(), inp["speed"].cuda(), inp["command"].cuda()))
    for a, b in zip(*outs):
        assert rel_err(a, b) < 1e-5


def test_bn_fold_cache_follows_the_running_statistics():
    """the eval-mode BatchNorm fold is cached; train-mode forwards with FROZEN weights (PMoE training
    with freeze(moe), the SWA BatchNorm re-estimation loop train_2.py:225-233) update the running statistics through raw
    pointers, and the next eval forward must see them -- like nn.BatchNorm2d, which always reads the live buffers."""
    g = torch.load(GOLDEN / "g5_moe_e3_b3_96.pt", weights_only=False)
    _, oracle, model, inp = build_pair(g, torch.float32)
    for p in model.parameters():
        p.requires_grad_(False)
    dev = [inp[k].cuda() for k in ("images", "speed", "command")]
    model.eval()
    with torch.no_grad():
        before = model.mixture_params(*dev)
        model.train()
        oracle.train()
        for _ in range(3):
            model(*dev)
            oracle(inp["images"], inp["speed"], inp["command"])
        model.eval()
        oracle.eval()
        folded = model.mixture_params(*dev)
        model._engine().fold_bn_eval = False
        unfolded = model.mixture_params(*dev)
        od, _ = oracle(inp["images"], inp["speed"], inp["command"])
    for a, b in zip(folded, unfolded):
        assert rel_err(a, b) < 1e-5
    assert rel_err(folded[1], od.component_distribution.base_dist.loc) < 1e-3        # the oracle after the same 3 updates
    assert rel_err(folded[1], before[1]) > 1e-3                                       # and the statistics did move


@pytest.mark.parametrize("alt", [False, True])
def test_lone_expert_forward(alt):
    """BaseExpert.forward / BaseExpertAlt.forward called on their own (moe.py:74-101, 112-128) -> (alpha, mean, std,
    pred_speed): a group of one on the same engine, alpha WITHOUT the mixture's softmax (post-ReLU for BaseExpert)."""
    import statistics
    from oracle import pmoe_oracle as O, weights as W
    from pmoe_amd.model.moe import BaseExpert, BaseExpertAlt
    from pmoe_amd.utils import stage2_model_cfg
    from tests.parity_util import rel_l2
    kind = "moe_alt" if alt else "moe"
    oracle = (O.BaseExpertAlt if alt else O.BaseExpert)(O.stage2_cfg(kind, 1))
    W.fill_state_dict(oracle, seed=3)
    oracle.train()
    model = (BaseExpertAlt if alt else BaseExpert)(stage2_model_cfg(kind, 1, dropout=0.0))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.cuda()
    model.compute_dtype = torch.float32
    model.train()
    inp = W.make_inputs(5, 64, 64, seed=5)
    got = model(inp["images"].cuda(), inp["speed"].cuda(), inp["command"].cuda())
    ref = oracle(inp["images"], inp["speed"], inp["command"])
    assert [tuple(t.shape) for t in got] == [(5, 1), (5, 2), (5, 2), (5, 1)]
    gen = torch.Generator().manual_seed(1)
    wts = [torch.randn(t.shape, generator=gen) for t in ref]
    for a, b, nm in zip(got, ref, ("alpha", "mean", "std", "pred_speed")):
        assert rel_err(a, b) <= 1e-4, (nm, rel_err(a, b))
    sum((a * w.cuda()).sum() for a, w in zip(got, wts)).backward()
    sum((b * w).sum() for b, w in zip(ref, wts)).backward()
    og = dict(oracle.named_parameters())
    errs = [rel_l2(p.grad, og[k].grad) for k, p in model.named_parameters() if og[k].grad.norm() > 0]
    assert statistics.median(errs) <= 5e-3 and max(errs) <= 0.15, (statistics.median(errs), max(errs))


def test_mlp_activation_choices_tanh_sigmoid():
    """make_mlp's `act` may be relu / tanh / sigmoid / elu (basics.py:23-28).  A mixture whose encoders use sigmoid and whose
    heads use tanh, against the oracle: forward 1e-4, every gradient tensor within 5e-3 (median) / 0.15."""
    import statistics
    from oracle import pmoe_oracle as O, weights as W
    from pmoe_amd.loss import moe_loss
    from pmoe_amd.model.moe import get_model
    from pmoe_amd.utils import stage2_model_cfg
    from tests.parity_util import rel_l2

    def tweak(cfg):
        cfg.speed_encoder.act = cfg.command_encoder.act = "sigmoid"
        cfg.action_head.act = cfg.speed_prediction.act = "tanh"
        return cfg
    ocfg = tweak(O.stage2_cfg("moe", 2))
    oracle = O.get_model(ocfg)
    W.fill_state_dict(oracle, seed=11)
    oracle.train()
    model = get_model(tweak(stage2_model_cfg("moe", 2, dropout=0.0)))
    model.load_state_dict(oracle.state_dict(), strict=True)
    model = model.cuda()
    model.compute_dtype = torch.float32
    model.train()
    inp = W.make_inputs(4, 64, 64, seed=21)
    dev = {k: v.cuda() for k, v in inp.items()}
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    od, os_ = oracle(inp["images"], inp["speed"], inp["command"])
    for a, b in zip(dist.hip_params + (speeds,), (od.mixture_distribution.probs, od.component_distribution.base_dist.loc,
                                                  od.component_distribution.base_dist.scale, os_)):
        assert rel_err(a, b) <= 1e-4, rel_err(a, b)
    moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs).backward()
    O.moe_loss(od, os_, inp["control"], inp["target_speed"], ocfg.loss_coefs).backward()
    og = dict(oracle.named_parameters())
    errs = [rel_l2(p.grad, og[k].grad) for k, p in model.named_parameters() if og[k].grad.norm() > 0]
    assert statistics.median(errs) <= 5e-3 and max(errs) <= 0.15, (statistics.median(errs), max(errs))
    # and in bf16 with dropout: runs, finite, dropout masks reproducible under a fixed seed
    m2 = get_model(tweak(stage2_model_cfg("moe", 2, dropout=0.3))).cuda()
    m2.train()
    torch.manual_seed(5)
    d1, s1 = m2(dev["images"], dev["speed"], dev["command"])
    moe_loss(d1, s1, dev["control"], dev["target_speed"], [0.7, 0.3]).backward()
    assert all(torch.isfinite(p.grad).all() for p in m2.parameters())


def test_mlp_with_batchnorm1d_heads():
    """make_mlp(bn=True) (docs/experiments.md:17-40, conf/stage_3.yaml:77-100): Linear without bias -> BatchNorm1d -> act ->
    Dropout per hidden layer.  state_dict layout of SURVEY appendix B, forward 1e-4, gradients, BatchNorm1d buffers."""
    import statistics
    from oracle import pmoe_oracle as O, weights as W
    from pmoe_amd.loss import moe_loss
    from pmoe_amd.model.moe import get_model
    from pmoe_amd.utils import stage2_model_cfg
    from tests.parity_util import rel_l2

    def tweak(cfg):
        for k in ("speed_encoder", "command_encoder", "action_head", "speed_prediction"):
            cfg[k].bn = True
        cfg.command_encoder.act = "tanh"
        return cfg
    ocfg = tweak(O.stage2_cfg("moe", 2))
    oracle = O.get_model(ocfg)
    W.fill_state_dict(oracle, seed=13)
    oracle.train()
    model = get_model(tweak(stage2_model_cfg("moe", 2, dropout=0.0)))
    sd = oracle.state_dict()
    assert "moe.0.speed_encoder.1.running_mean" in sd and "moe.0.speed_encoder.0.bias" not in sd     # BN at index 1, no bias
    model.load_state_dict(sd, strict=True)
    model = model.cuda()
    model.compute_dtype = torch.float32
    model.train()
    inp = W.make_inputs(6, 64, 64, seed=23)
    dev = {k: v.cuda() for k, v in inp.items()}
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    od, os_ = oracle(inp["images"], inp["speed"], inp["command"])
    for a, b in zip(dist.hip_params + (speeds,), (od.mixture_distribution.probs, od.component_distribution.base_dist.loc,
                                                  od.component_distribution.base_dist.scale, os_)):
        assert rel_err(a, b) <= 1e-4, rel_err(a, b)
    moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs).backward()
    O.moe_loss(od, os_, inp["control"], inp["target_speed"], ocfg.loss_coefs).backward()
    og = dict(oracle.named_parameters())
    errs = [rel_l2(p.grad, og[k].grad) for k, p in model.named_parameters() if og[k].grad.norm() > 0]
    assert statistics.median(errs) <= 5e-3 and max(errs) <= 0.15, (statistics.median(errs), max(errs))
    osd, msd = oracle.state_dict(), model.state_dict()
    for k in osd:
        if k.endswith("num_batches_tracked"):
            assert int(msd[k]) == int(osd[k]), k
        elif "running_" in k and ("encoder" in k or "speed_pred" in k or "action_features" in k):
            assert rel_err(msd[k], osd[k]) <= 1e-4, k
    model.eval()
    oracle.eval()
    with torch.no_grad():
        d2, _ = model(dev["images"], dev["speed"], dev["command"])
        o2, _ = oracle(inp["images"], inp["speed"], inp["command"])
    assert rel_err(d2.hip_params[1], o2.component_distribution.base_dist.loc) <= 1e-3


def test_module_contract():
    """deepcopy (AveragedModel), freeze-by-name, state_dict round trip, frozen parameters get no grads."""
    from pmoe_amd.loss import moe_loss
    from pmoe_amd.utils import freeze
    g = torch.load(GOLDEN / "g4_moealt_e4_b2_64.pt", weights_only=False)
    ocfg, oracle, model, inp = build_pair(g, torch.float32)
    dev = {k: v.cuda() for k, v in inp.items()}
    swa = torch.optim.swa_utils.AveragedModel(model)
    swa.update_parameters(model)
    with torch.no_grad():
        model.eval(); swa.eval()
        d2, s2 = swa(dev["images"], dev["speed"], dev["command"])
        d3, s3 = model(dev["images"], dev["speed"], dev["command"])
    assert rel_err(s2, s3) < 1e-6
    model.train()
    freeze(model, ["alpha", "action_pred"])
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    moe_loss(dist, speeds, dev["control"], dev["target_speed"], [0.7, 0.3]).backward()
    for n, p in model.named_parameters():
        if "alpha" in n or "action_pred" in n:
            assert p.grad is not None and p.requires_grad, n
        else:
            assert p.grad is None, n
    # generic torch.distributions path (what the reference trainer calls) gives the same loss as the fused kernel
    for p in model.parameters():
        p.requires_grad_(True)
    model.eval()
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    fused = moe_loss(dist, speeds, dev["control"], dev["target_speed"], [0.7, 0.3])
    generic = -dist.log_prob(dev["control"]).mean() * 0.7 + 0.3 * torch.nn.functional.mse_loss(
        speeds, dev["target_speed"].unsqueeze(1).expand_as(speeds)) / speeds.shape[1]
    assert abs(fused.item() - generic.item()) < 1e-5
    generic.backward()
    assert model.moe[0].backbone.layer1[0].conv1.weight.grad is not None


def test_dropout_train_mode_runs_and_is_seeded():
    g = torch.load(GOLDEN / "g4_moealt_e4_b2_64.pt", weights_only=False)
    _, _, model, inp = build_pair(g, torch.bfloat16, dropout=0.3)
    dev = {k: v.cuda() for k, v in inp.items()}
    torch.manual_seed(1)
    with torch.no_grad():
        a = model.mixture_params(dev["images"], dev["speed"], dev["command"])
    torch.manual_seed(1)
    with torch.no_grad():
        b = model.mixture_params(dev["images"], dev["speed"], dev["command"])
    with torch.no_grad():
        c = model.mixture_params(dev["images"], dev["speed"], dev["command"])
    # BN running stats do not influence train-mode outputs, so same seed -> same masks -> same outputs
    assert torch.equal(a[3], b[3]) and not torch.equal(a[3], c[3])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_fused_stem_tail_matches_unfused(dtype):
    """csrc/stem_tail.hip (BN+ReLU+BN+ReLU+MaxPool re-derived from z2) against the same chain built from the
    separate BatchNorm / max-pool kernels: forward outputs and every parameter gradient."""
    from pmoe_amd.loss import moe_loss
    g = torch.load(GOLDEN / "g5_moe_e3_b3_96.pt", weights_only=False)
    res = []
    for fuse in (True, False):
        _, _, model, inp = build_pair(g, dtype)
        model._engine().fuse_stem_tail = fuse
        dev = {k: v.cuda() for k, v in inp.items()}
        dist, speeds = model(dev["images"], dev["speed"], dev["command"])
        moe_loss(dist, speeds, dev["control"], dev["target_speed"], [0.7, 0.3]).backward()
        res.append((dist.hip_params, speeds, {k: p.grad.clone() for k, p in model.named_parameters()},
                    {k: v.clone() for k, v in model.state_dict().items() if "running" in k}))
    (pa, sa, ga, ba), (pb, sb, gb, bb) = res
    ftol = 1e-4 if dtype == torch.float32 else 3e-2
    for a, b in zip(pa, pb):
        assert rel_err(a, b) <= ftol
    assert rel_err(sa, sb) <= ftol
    for k in ba:
        assert rel_err(ba[k], bb[k]) <= (1e-4 if dtype == torch.float32 else 1e-2), k
    if dtype == torch.float32:
        # two exact-f32 evaluation orders of a chaotic 20-layer gradient (see parity_util): compare the
        # distribution, not the single worst near-cancelling tensor
        errs = sorted(((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-20)).item() for k in ga if gb[k].norm() > 1e-8)
        assert errs[len(errs) // 2] <= 1e-3, errs[len(errs) // 2]
        assert errs[int(0.95 * len(errs))] <= 2e-2, errs[int(0.95 * len(errs))]


def test_stem_input_fold_matches_dgrad_path():
    """conv1.weight / eca1 gradients from per-image filter gradients (eca_stem_fold) vs the explicit
    data-gradient convolution + ECA backward."""
    from pmoe_amd.loss import moe_loss
    g = torch.load(GOLDEN / "g5_moe_e3_b3_96.pt", weights_only=False)
    grads = []
    for fold in (True, False):
        _, _, model, inp = build_pair(g, torch.float32)
        model._engine().fold_stem_input = fold
        dev = {k: v.cuda() for k, v in inp.items()}
        dist, speeds = model(dev["images"], dev["speed"], dev["command"])
        moe_loss(dist, speeds, dev["control"], dev["target_speed"], [0.7, 0.3]).backward()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if "conv1.layer1" in k})
    for k in grads[0]:
        e = ((grads[0][k] - grads[1][k]).norm() / (grads[1][k].norm() + 1e-20)).item()
        assert e <= 1e-3, (k, e)      # two f32 summation orders (per-image fold vs per-expert atomics)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_eca_gate_fold_matches_explicit_path(dtype):
    """ECA(64) -> conv2 with the gate folded into per-image weights (engine._eca_conv_folded) against the explicit
    gap / scale / conv chain: forward outputs and every gradient of the stem."""
    from pmoe_amd.loss import moe_loss
    g = torch.load(GOLDEN / "g5_moe_e3_b3_96.pt", weights_only=False)
    res = []
    for fold in (True, False):
        _, _, model, inp = build_pair(g, dtype)
        model._engine().fold_eca_gate = fold
        dev = {k: v.cuda() for k, v in inp.items()}
        dist, speeds = model(dev["images"], dev["speed"], dev["command"])
        moe_loss(dist, speeds, dev["control"], dev["target_speed"], [0.7, 0.3]).backward()
        res.append((dist.hip_params, speeds, {k: p.grad.clone() for k, p in model.named_parameters()}))
    (pa, sa, ga), (pb, sb, gb) = res
    ftol = 1e-4 if dtype == torch.float32 else 3e-2
    for a, b in zip(pa, pb):
        assert rel_err(a, b) <= ftol
    assert rel_err(sa, sb) <= ftol
    if dtype == torch.float32:
        stem = [k for k in ga if "backbone.conv1." in k]
        assert len(stem) == 3 * 8
        for k in stem:
            e = ((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-20)).item()
            assert e <= 2e-2, (k, e)           # two f32 summation orders upstream of a chaotic network (see parity_util)
        errs = sorted(((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-20)).item() for k in ga if gb[k].norm() > 1e-8)
        assert errs[len(errs) // 2] <= 1e-3, errs[len(errs) // 2]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_pooled_stem_tail_backward_matches_three_phase_path(dtype):
    """Train-mode stem-tail BatchNorm reductions from ONE pass over the pooled tensors + forward channel moments
    (stem_tail_pooled / stem_tail_combine) against the three full sweeps over z2."""
    from pmoe_amd.loss import moe_loss
    g = torch.load(GOLDEN / "g5_moe_e3_b3_96.pt", weights_only=False)
    res = []
    for pooled in (True, False):
        _, _, model, inp = build_pair(g, dtype)
        model._engine().pooled_stem_bwd = pooled
        dev = {k: v.cuda() for k, v in inp.items()}
        dist, speeds = model(dev["images"], dev["speed"], dev["command"])
        moe_loss(dist, speeds, dev["control"], dev["target_speed"], [0.7, 0.3]).backward()
        res.append({k: p.grad.clone() for k, p in model.named_parameters()})
    ga, gb = res
    errs = sorted((((ga[k] - gb[k]).norm() / (gb[k].norm() + 1e-20)).item(), k) for k in ga
                  if "backbone.conv1." in k or "backbone.bn1." in k)
    assert len(errs) == 3 * 10
    if dtype == torch.float32:
        assert errs[-1][0] <= 2e-4, errs[-1]
    else:       # bf16: xhat is recovered from the bf16-ROUNDED pooled output; the 3-element ECA filters cancel heavily
        assert errs[len(errs) // 2][0] <= 2e-2 and errs[-1][0] <= 0.15, (errs[len(errs) // 2], errs[-1])


def test_graph_captured_inference_matches_eager():
    """pmoe_amd.infer.GraphedMixture: the eval-mode B=1 chain captured into a HIP graph replays
    bit-identically, follows new inputs, and must be refreshed after a weight change."""
    from pmoe_amd.infer import GraphedMixture
    g = torch.load(GOLDEN / "g2_moe_e4_b1_224_eval.pt", weights_only=False)
    _, _, model, inp = build_pair(g, torch.bfloat16)
    dev = {k: v.cuda() for k, v in inp.items()}
    with torch.no_grad():
        ref = [t.clone() for t in model.mixture_params(dev["images"], dev["speed"], dev["command"])]
    gm = GraphedMixture(model, dev["images"], dev["speed"], dev["command"])
    for a, b in zip(ref, gm(dev["images"], dev["speed"], dev["command"])):
        assert torch.equal(a, b)
    img2 = dev["images"].flip(-1).contiguous()
    with torch.no_grad():
        ref2 = [t.clone() for t in model.mixture_params(img2, dev["speed"], dev["command"])]
    for a, b in zip(ref2, gm(img2, dev["speed"], dev["command"])):
        assert torch.equal(a, b)
    assert gm.sample(img2, dev["speed"], dev["command"]).shape == (1, 2)
    with pytest.raises(ValueError, match="captured for input shape"):
        gm(dev["images"][:, :, :, :64], dev["speed"], dev["command"])
    model.train()
    with pytest.raises(RuntimeError, match="eval"):
        GraphedMixture(model, dev["images"], dev["speed"], dev["command"])


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_side_stream_weight_gradients_equal_the_main_stream_ones(dtype):
    """engine.overlap_wgrad (PMOE_OVERLAP_WGRAD=1): weight / bias gradients on a second HIP stream with their own K-split
    scratch, ordered against the main stream by events.  Every reduction is fixed-order, so the result must be bit-identical."""
    from pmoe_amd.loss import moe_loss
    g = torch.load(GOLDEN / "g5_moe_e3_b3_96.pt", weights_only=False)
    res = []
    for overlap in (False, True):
        _, _, model, inp = build_pair(g, dtype)
        model._engine().overlap_wgrad = overlap
        dev = {k: v.cuda() for k, v in inp.items()}
        dist, speeds = model(dev["images"], dev["speed"], dev["command"])
        moe_loss(dist, speeds, dev["control"], dev["target_speed"], [0.7, 0.3]).backward()
        torch.cuda.synchronize()
        res.append({k: p.grad.clone() for k, p in model.named_parameters()})
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k


def test_planned_inference_replays_the_recorded_launches():
    """pmoe_amd.infer.PlannedMixture (host launch cost): the eval-mode B=1 chain recorded as a list of C-ABI
    calls and re-issued without the engine's Python equals the eager path bit for bit, follows new inputs, survives unrelated
    allocations (its buffers live in a private memory pool), and must be refreshed after a weight change."""
    from pmoe_amd.infer import PlannedMixture
    g = torch.load(GOLDEN / "g2_moe_e4_b1_224_eval.pt", weights_only=False)
    _, _, model, inp = build_pair(g, torch.bfloat16)
    dev = {k: v.cuda().float().contiguous() for k, v in inp.items()}
    model.eval()
    with torch.no_grad():
        ref = [t.clone() for t in model.mixture_params(dev["images"], dev["speed"], dev["command"])]
    pm = PlannedMixture(model, dev["images"], dev["speed"], dev["command"])
    assert len(pm.plan.calls) > 20
    for a, b in zip(ref, pm(dev["images"], dev["speed"], dev["command"])):
        assert torch.equal(a, b)
    junk = [torch.full((1 << 20,), float(i), device="cuda") for i in range(8)]       # would land in the plan's buffers if they were free
    img2 = dev["images"].flip(-1).contiguous()
    with torch.no_grad():
        ref2 = [t.clone() for t in model.mixture_params(img2, dev["speed"], dev["command"])]
    for a, b in zip(ref2, pm(img2, dev["speed"], dev["command"])):
        assert torch.equal(a, b)
    del junk
    assert pm.sample(img2, dev["speed"], dev["command"]).shape == (1, 2)
    with pytest.raises(ValueError, match="recorded for input shape"):
        pm(dev["images"][:, :, :, :64].contiguous(), dev["speed"], dev["command"])
    with torch.no_grad():                                  # weight change: stale until refresh()
        for p_ in model.parameters():
            p_.mul_(1.01)
        ref3 = [t.clone() for t in model.mixture_params(img2, dev["speed"], dev["command"])]
    # (no explicit refresh() -- the replay compares the engine's build / pointer / version keys with the recorded
    #  ones and re-records by itself; the eager call above re-packed the weight banks the old plan pointed into)
    for a, b in zip(ref3, pm(img2, dev["speed"], dev["command"])):
        assert torch.equal(a, b)
    # an eager call in ANOTHER compute dtype re-allocates every packed bank (engine._ensure_built): the plan must notice
    model.compute_dtype = torch.float32
    with torch.no_grad():
        model.mixture_params(img2, dev["speed"], dev["command"])
    model.compute_dtype = torch.bfloat16
    for a, b in zip(ref3, pm(img2, dev["speed"], dev["command"])):
        assert torch.equal(a, b)
    model.train()
    with pytest.raises(RuntimeError, match="eval"):
        PlannedMixture(model, dev["images"], dev["speed"], dev["command"])


@pytest.mark.parametrize("shape", [(1, 70, 54), (3, 33, 47), (2, 130, 64)])
def test_ragged_shapes_f32_against_live_oracle(shape):
    """Edge cases the goldens do not carry: batch of ONE in train mode, odd and non-square image sides (pooling /
    stride-2 layers with odd extents, tiles that straddle image borders).  HIP f32 vs the CPU oracle run live."""
    from oracle import pmoe_oracle as O
    from oracle import weights as W
    from pmoe_amd.loss import moe_loss
    from pmoe_amd.model.moe import get_model
    from pmoe_amd.utils import stage2_model_cfg
    Bn, H, Wd = shape
    ocfg = O.stage2_cfg("moe", 2, dropout=0.0)
    oracle = O.get_model(ocfg)
    W.fill_state_dict(oracle, seed=3)
    oracle.train()
    model = get_model(stage2_model_cfg("moe", 2, dropout=0.0))
    model.load_state_dict(oracle.state_dict())
    model = model.cuda().train()
    model.compute_dtype = torch.float32
    inp = W.make_inputs(Bn, H, Wd, seed=99)
    dev = {k: v.cuda() for k, v in inp.items()}
    if Bn == 1:
        # BatchNorm over one sample still has H*W > 1 values per channel everywhere except the 512-d MLPs (no BN there)
        pass
    od, os_ = oracle(inp["images"], inp["speed"], inp["command"])
    ol = O.moe_loss(od, os_, inp["control"], inp["target_speed"], ocfg.loss_coefs)
    ol.backward()
    dist, speeds = model(dev["images"], dev["speed"], dev["command"])
    loss = moe_loss(dist, speeds, dev["control"], dev["target_speed"], ocfg.loss_coefs)
    loss.backward()
    probs, mean, std = dist.hip_params
    assert rel_err(probs, od.mixture_distribution.probs) <= 2e-4
    assert rel_err(mean, od.component_distribution.base_dist.loc) <= 2e-4
    assert rel_err(std, od.component_distribution.base_dist.scale) <= 2e-4
    assert rel_err(speeds, os_) <= 2e-4
    assert abs(loss.item() - ol.item()) <= 2e-4 * max(1.0, abs(ol.item()))
    on = dict(oracle.named_parameters())
    errs = sorted(((p.grad.cpu() - on[k].grad).norm() / (on[k].grad.norm() + 1e-12)).item()
                  for k, p in model.named_parameters() if on[k].grad.norm() > 1e-7)
    assert errs[len(errs) // 2] <= 2e-3, errs[len(errs) // 2]      # median; single tensors can carry a ReLU-mask flip


def test_bn_relu_with_two_consumers_fails_loudly():
    """the PMOE_RES_DBN data gradient (engine._dgrad_with_bn_reduce) masks and reduces the gradient of
    a = relu(BatchNorm(z)) inside the data gradient of a's consumer, assuming that consumer is the only one.  No shipped graph
    fans such an activation out; one that did must raise instead of returning silently wrong dgamma / dbeta / dz."""
    from pmoe_amd.engine import Var
    from pmoe_amd.model.moe import get_model
    from pmoe_amd.utils import stage2_model_cfg
    m = get_model(stage2_model_cfg("moe", 2, dropout=0.0)).cuda()
    m.compute_dtype = torch.bfloat16
    m.train()
    eng = m._engine()
    eng._begin(torch.rand(4, 4, 3, 64, 64, device="cuda"), True, True, torch.bfloat16, 0)
    eng._layout_arena()
    eng._arena = torch.zeros(eng._arena_numel, device="cuda")
    eng._filled, eng._cursor = set(), 0
    x = Var(torch.randn(eng.N, 64, 64, 64, device="cuda").to(torch.bfloat16))
    x.needs_grad = True
    b0, b1 = eng.blocks[0], eng.blocks[1]
    a = eng._conv_bn(x, b0["conv1"], b0["bn1"], relu=True)
    assert a.bn_src is not None
    y1 = eng._conv(a, b0["conv2"], bias=False)
    y2 = eng._conv(a, b1["conv1"], bias=False)            # second consumer of the same BatchNorm+ReLU output
    for y in (y1, y2):
        y.set_grad(torch.randn_like(y.t))
    tape = list(reversed(eng.tape))
    tape[0]()                                             # y2's backward: takes the fused path and leaves the reductions
    assert a.bn_part is not None, "this shape no longer takes the PMOE_RES_DBN data gradient: pick one that does"
    with pytest.raises(RuntimeError, match="exactly one consumer"):
        tape[1]()
