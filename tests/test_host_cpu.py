"""CPU-side host logic: state_dict contract against the reference goldens, freeze-by-name, config helpers,
deepcopy (AveragedModel), loud failure without the device."""
import copy

import pytest
import torch

from pmoe_amd.model import blocks as B
from pmoe_amd.model.moe import MixtureOfExperts, get_model
from pmoe_amd.utils import AttrDict, freeze, stage2_model_cfg


def _g(golden_dir, name):
    return torch.load(golden_dir / f"{name}.pt", weights_only=False)


@pytest.mark.parametrize("name,typ,E", [("g1_moe_e4_b2_128", "moe", 4), ("g3_moe_e8_b2_128", "moe", 8),
                                        ("g4_moealt_e4_b2_64", "moe_alt", 4), ("g5_moe_e3_b3_96", "moe", 3),
                                        ("g6_moeshared_k4_b6_96", "moe_shared", 4),
                                        ("g7_moeshared_k6_b1_224_eval", "moe_shared", 6)])
def test_state_dict_keys_match_reference(golden_dir, name, typ, E):
    g = _g(golden_dir, name)
    m = get_model(stage2_model_cfg(typ, E, dropout=0.0))
    sd = m.state_dict()
    assert list(sd.keys()) == g["state_dict_keys"]
    assert [tuple(v.shape) for v in sd.values()] == g["state_dict_shapes"]


@pytest.mark.parametrize("name", ["p1_punet_b2_64_f2", "p3_punetinter_b2_64_f2", "p5_pmoe_e2_b2_64_f2"])
def test_punet_state_dict_and_freeze_match_reference(golden_dir, tmp_path, name):
    """PUNetExpert / PMoE containers: checkpoint files are read like the reference does (punet.py:40, moe.py:278,335),
    key order, shapes and the requires_grad pattern after the constructor's freeze() calls."""
    from tests.punet_util import build_product
    g = _g(golden_dir, name)
    m = build_product(tmp_path, g["meta"], exclude_freeze=["lat_weights", "long_weights"] if g["meta"]["type"] == "pmoe" else ())
    sd = m.state_dict()
    assert list(sd.keys()) == g["state_dict_keys"]
    assert [tuple(v.shape) for v in sd.values()] == g["state_dict_shapes"]
    assert {k: p.requires_grad for k, p in m.named_parameters()} == g["requires_grad"]
    eng = (m.punet if g["meta"]["type"] == "pmoe" else m)._engine()
    assert {id(p) for p in eng.flat_params} == {id(p) for p in (m.punet if g["meta"]["type"] == "pmoe" else m).parameters()}


def test_mlp_layouts_match_reference(golden_dir):
    g = _g(golden_dir, "micro")
    for (bn, p, dims), keys in g["mlp_layouts"].items():
        assert list(B.make_mlp(list(dims), "relu", False, bn, p).state_dict().keys()) == keys
    for c, k in g["eca_k"].items():
        assert B.eca_kernel_size(c) == k


def test_get_model_errors_like_reference():
    with pytest.raises(ValueError, match="UNKNOWN"):
        get_model(AttrDict(type="nope"))
    with pytest.raises(AssertionError, match="MoE pretrained"):
        get_model(stage2_model_cfg("pmoe", 4))
    with pytest.raises(FileNotFoundError):
        get_model(stage2_model_cfg("punet", 4, unet_path="/nonexistent/unet.pth"))


def test_freeze_semantics():
    m = get_model(stage2_model_cfg("moe", 2, dropout=0.3))
    freeze(m, ["alpha", "lat_weights"])
    for n, p in m.named_parameters():
        assert p.requires_grad == ("alpha" in n), n
    freeze(m, [])
    assert not any(p.requires_grad for p in m.parameters())


def test_deepcopy_and_engine_grouping():
    m = get_model(stage2_model_cfg("moe", 3, dropout=0.0))
    eng = m._engine()
    assert eng.E == 3 and sum(p.numel() for p in eng.flat_params) == sum(p.numel() for p in m.parameters())
    assert len({id(p) for p in eng.flat_params}) == len(list(m.parameters()))
    m2 = copy.deepcopy(m)
    assert "_eng" not in m2.__dict__ and "_eng" in m.__dict__
    assert m2._engine() is not eng
    swa = torch.optim.swa_utils.AveragedModel(m)
    assert isinstance(swa.module, MixtureOfExperts)
    sh = get_model(stage2_model_cfg("moe_shared", 5, dropout=0.0))
    eng = sh._engine()
    assert (eng.E, eng.K, eng.shared) == (1, 5, True) and eng.head.cout == 25
    assert len({id(p) for p in eng.flat_params}) == len(list(sh.parameters()))


def test_cpu_inputs_fail_loudly():
    m = get_model(stage2_model_cfg("moe", 2, dropout=0.0))
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.zeros(1, 4, 3, 32, 32), torch.zeros(1, 1), torch.zeros(1, 6))
    with pytest.raises(RuntimeError, match="parameter container"):
        m.moe[0].speed_encoder(torch.zeros(1, 1))


def test_product_never_imports_the_oracle():
    import pathlib
    repo = pathlib.Path(__file__).resolve().parents[1]
    for f in (repo / "pmoe_amd").rglob("*.py"):
        assert "oracle" not in f.read_text(), f
    # product-side tools (benchmarks, profilers) may not use it either; experiments that do live in tests/experiments/
    for f in (repo / "tools").glob("*.py"):
        text = f.read_text()
        assert "from oracle" not in text and "import oracle" not in text, f


def test_stage1_entry_points_have_no_cpu_path(tmp_path):
    """PredictiveUnet.forward / AutoregressiveCriterion fail loudly on CPU tensors, keep the reference's
    state_dict layout and argument checks, and survive copy.deepcopy (AveragedModel, train_1.py:113)."""
    import copy

    import pytest
    import torch

    from pmoe_amd.loss import AutoregressiveCriterion
    from pmoe_amd.model import blocks as B
    from pmoe_amd.model.punet import PredictiveUnet
    torch.save({"unet": B.UNet().state_dict()}, tmp_path / "unet.pth")
    m = PredictiveUnet(4, 2, model_name="unet", model_path=str(tmp_path / "unet.pth"))
    assert [k.split(".")[0] for k in m.state_dict()][0] == "unet" and any(k.startswith("pred_unet.up_1.") for k in m.state_dict())
    assert not any(p.requires_grad for p in m.unet.parameters()) and all(p.requires_grad for p in m.pred_unet.parameters())
    with pytest.raises(RuntimeError, match="no CPU path"):
        m(torch.rand(1, 4, 3, 32, 32))
    with pytest.raises(AssertionError):
        m(torch.rand(1, 3, 3, 32, 32))                     # punet.py:84-86: number of past frames
    with pytest.raises(RuntimeError, match="no CPU path"):
        AutoregressiveCriterion(2)(torch.randn(1, 2, 23, 8, 8), torch.zeros(1, 2, 8, 8, dtype=torch.long))
    with pytest.raises(ValueError):
        AutoregressiveCriterion(1, "dice")
    m2 = copy.deepcopy(m)
    assert list(m2.state_dict().keys()) == list(m.state_dict().keys()) and "_eng" not in m2.__dict__


def test_dp_bucket_cuts_follow_backward_time():
    """pmoe_amd.engine._bucket_cuts: the arena is in backward order; the LAST bucket (never hidden: it completes when
    backward ends) holds only stem + layer1, the others split the parameter-heavy head of the arena evenly; every cut is a
    multiple of 256 elements so that reduce-scatter slices divide for any world size up to 256."""
    m = get_model(stage2_model_cfg("moe", 4, dropout=0.0))
    eng = m._engine()
    eng._layout_arena()
    cuts = eng._bucket_cuts(6)
    assert len(cuts) == 6 and cuts[-1] == eng._arena_numel and cuts == sorted(cuts)
    assert all(c % 256 == 0 for c in cuts) and eng._arena_numel - eng._arena_used < 256
    total = sum(p.numel() for p in m.parameters())
    assert eng._arena_used == total == 55373360
    tail = eng._arena_numel - cuts[-2]
    assert 0 < tail < 0.06 * total                       # layer1 + stem + the two measurement encoders: ~5 % of the bytes
    first = cuts[0]
    assert all(abs((cuts[i + 1] - cuts[i]) - first) <= 256 for i in range(3))     # even split of the rest
    # the slot of layer1.0.conv1 lies in the last bucket, layer2's in an earlier one
    def off(name):
        key = next(eng._key(k, l) for k, l, _ in eng.params if getattr(l, "name", "") == name and k == "w")
        return eng._slots[key][0]
    assert off("layer1.0.conv1") >= cuts[-2] > off("layer2.0.conv1")


def test_bench_self_launch_refuses_missing_devices():
    """`python bench.py --gpus N` starts its own one-rank-per-GPU job; on a node with fewer devices it
    must say so -- before any GPU call -- instead of printing a launcher hint.  (This container has no GPU: 0 < 2.)"""
    import subprocess
    import sys
    from pathlib import Path
    repo = Path(__file__).resolve().parents[1]
    env = {k: v for k, v in __import__("os").environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "PMOE_BENCH_SHARE_GPU")}
    r = subprocess.run([sys.executable, str(repo / "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=300, env=env)
    if torch.cuda.device_count() >= 2:
        pytest.skip("node has the devices")
    assert r.returncode == 2 and "needs 2 devices" in r.stderr, (r.returncode, r.stderr[-500:])
