#!/usr/bin/env python3
"""Generate tests/golden/*.pt by running the REAL reference (imported from /root/reference).

TEST INFRASTRUCTURE.  Runs only in the build container (the reference cannot travel to the GPU
box); the outputs are small data fixtures (inputs are re-derived from seeds, weights from
``oracle.weights.fill_state_dict``).  Usage:  python oracle/make_golden.py

Stand-ins injected before the import (ordinary ModuleNotFoundError otherwise):
  * ``torchvision.models`` -> oracle/resnet_topology.py (published ResNet topology, restated)
  * ``thop``               -> empty stub (only imported by utils/nn.py:7, never called on this path)
"""
import os
import sys
import types
from pathlib import Path

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import torch

REPO = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/PMoE")
sys.path.insert(0, str(REPO))

from oracle import resnet_topology, weights          # noqa: E402
from oracle.pmoe_oracle import Cfg, stage2_cfg       # noqa: E402


def import_reference():
    tv = types.ModuleType("torchvision")
    tv.models = resnet_topology
    sys.modules["torchvision"] = tv
    sys.modules["torchvision.models"] = resnet_topology
    thop = types.ModuleType("thop")
    thop.profile = thop.clever_format = lambda *a, **k: None
    sys.modules["thop"] = thop
    sys.path.insert(0, str(REF))
    sys.path.insert(0, str(REF / "trainer"))
    from model import moe as ref_moe               # noqa
    from model.blocks import basics as ref_basics  # noqa
    import loss as ref_loss                        # noqa
    return ref_moe, ref_basics, ref_loss


GRAD_SLICES = [
    "moe.0.backbone.conv1.layer1.eca1.conv.weight",
    "moe.0.backbone.conv1.layer1.conv1.0.weight",
    "moe.0.backbone.conv1.layer1.conv1.1.weight",
    "moe.0.backbone.conv1.layer2.eca2.conv.weight",
    "moe.0.backbone.conv1.layer2.conv2.0.weight",
    "moe.0.backbone.bn1.bias",
    "moe.1.backbone.layer1.0.conv1.weight",
    "moe.1.backbone.layer2.0.downsample.0.weight",
    "moe.1.backbone.layer2.0.downsample.1.weight",
    "moe.0.backbone.layer4.1.conv2.weight",
    "moe.0.backbone.layer4.1.bn2.weight",
    "moe.0.speed_encoder.0.weight",
    "moe.0.command_encoder.0.weight",
    "moe.0.speed_pred.0.weight",
    "moe.1.action_features.0.weight",
    "moe.0.action_pred.weight",
    "moe.0.action_pred.bias",
    "moe.1.alpha.weight",
    # moe_shared (no "moe.N." prefix)
    "backbone.conv1.layer1.eca1.conv.weight", "backbone.conv1.layer1.conv1.0.weight", "backbone.conv1.layer2.conv2.0.weight",
    "backbone.bn1.bias", "backbone.layer2.0.downsample.0.weight", "backbone.layer4.1.conv2.weight",
    "backbone.layer4.1.bn2.weight", "speed_encoder.0.weight", "command_encoder.0.weight", "speed_pred.0.weight",
    "action_features.0.weight", "action_pred.weight", "action_pred.bias", "alpha.weight", "alpha.bias",
]


def run_case(ref_moe, ref_loss, name, model_type, n_experts, batch, size, train=True, steps=0):
    torch.manual_seed(0)
    cfg = stage2_cfg(model_type, n_experts, dropout=0.0)
    model = ref_moe.get_model(cfg)
    weights.fill_state_dict(model, seed=0)
    model.train(train)
    inp = weights.make_inputs(batch, size, size, seed=1234)
    out = {"meta": dict(name=name, type=model_type, n_experts=n_experts, batch=batch, size=size,
                        train=train, weight_seed=0, input_seed=1234),
           "state_dict_keys": list(model.state_dict().keys()),
           "state_dict_shapes": [tuple(v.shape) for v in model.state_dict().values()]}
    if train:
        dist, speeds = model(inp["images"], inp["speed"], inp["command"])
        loss = ref_loss.moe_loss(dist, speeds, inp["control"], inp["target_speed"].clone(), cfg.loss_coefs)
        loss.backward()
        out["loss"] = loss.detach().clone()
        named = dict(model.named_parameters())
        out["grad_norms"] = {k: p.grad.norm().item() for k, p in named.items()}
        out["grad_sums"] = {k: p.grad.double().sum().item() for k, p in named.items()}
        sl = {}
        for k in GRAD_SLICES:
            if k in named:
                sl[k] = named[k].grad.flatten()[:64].clone()
        if model_type == "moe_alt":
            for k in ("moe.0.alpha.0.weight", "moe.0.alpha.2.weight", "moe.0.alpha.2.bias"):
                sl[k] = named[k].grad.flatten()[:64].clone()
        out["grad_slices"] = sl
        sd = model.state_dict()
        pre = "backbone." if model_type == "moe_shared" else "moe.0.backbone."
        out["bn_after_1"] = {k: sd[k].clone() for k in sd
                             if k.startswith(pre) and
                             (k.endswith("running_mean") or k.endswith("running_var")
                              or k.endswith("num_batches_tracked"))
                             and ("conv1.layer1" in k or k[len(pre):].startswith("bn1.") or "layer4.1.bn2" in k
                                  or "layer2.0.downsample" in k)}
    else:
        with torch.no_grad():
            dist, speeds = model(inp["images"], inp["speed"], inp["command"])
    out["probs"] = dist.mixture_distribution.probs.detach().clone()
    out["mean"] = dist.component_distribution.base_dist.loc.detach().clone()
    out["std"] = dist.component_distribution.base_dist.scale.detach().clone()
    out["speeds"] = speeds.detach().clone()
    out["log_prob"] = dist.log_prob(inp["control"]).detach().clone()
    # per-expert 1536-d feature slice (first 8 of each 512 block) via the reference's own modules
    with torch.no_grad():
        model.eval()
        e0 = model if model_type == "moe_shared" else model.moe[0]
        x = inp["images"].view(batch, -1, size, size)
        out["feat_eval_e0"] = e0.backbone(x)[:, :16].clone()
        model.train(train)

    if steps:
        # H1: the reference's own step recipe (train_2.py:149-165): fwd, moe_loss, zero_grad, backward,
        # clip_grad_norm_(1.0), grad-norm, Adam(amsgrad) step.
        torch.manual_seed(0)
        model = ref_moe.get_model(cfg)
        weights.fill_state_dict(model, seed=0)
        model.train()
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, model.parameters()),
                               lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=True)
        traj = []
        for _ in range(steps):
            dist, speeds = model(inp["images"], inp["speed"], inp["command"])
            loss = ref_loss.moe_loss(dist, speeds, inp["control"], inp["target_speed"].clone(), cfg.loss_coefs)
            opt.zero_grad()
            loss.backward()
            gn = torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
            opt.step()
            traj.append(dict(loss=loss.item(), grad_norm=float(gn)))
        named = dict(model.named_parameters())
        out["h1"] = dict(traj=traj, param_sums={k: named[k].double().sum().item() for k in GRAD_SLICES if k in named},
                         param_l2={k: named[k].norm().item() for k in GRAD_SLICES if k in named})
    return out


PUNET_SLICES = [
    "backbone.conv1.layer1.eca1.conv.weight", "backbone.conv1.layer1.conv1.0.weight", "backbone.conv1.layer1.conv1.1.weight",
    "backbone.conv1.layer2.conv2.0.weight", "backbone.bn1.bias", "backbone.layer2.0.downsample.0.weight",
    "backbone.layer4.1.conv2.weight", "speed_encoder.0.weight", "command_encoder.0.weight", "speed_pred.0.weight",
    "action_pred.0.0.weight", "action_pred.0.2.weight", "action_pred.1.weight", "action_pred.1.bias",
]
PUNET_BN = ["punet.unet.dwn_1.1", "punet.unet.dwn_5.4", "punet.unet.up_forw_1.1", "punet.unet.up_forw_4.4",
            "punet.entry_block.layer1.conv1.1", "punet.entry_block.layer2.conv2.1", "punet.pred_unet.dwn_1.1",
            "punet.pred_unet.up_forw_4.4", "backbone.conv1.layer1.conv1.1", "backbone.bn1", "backbone.layer4.1.bn2"]


def _checkpoints(tmp, cfg_kw, model_type, n_experts):
    """The reference constructors read three checkpoint files (punet.py:40, moe.py:278, moe.py:335); their CONTENT
    is irrelevant here (every weight is overwritten by fill_state_dict afterwards), only the key sets must fit."""
    from model.blocks.unet import UNet as RefUNet
    from model.punet import PredictiveUnet as RefPU
    tmp.mkdir(parents=True, exist_ok=True)
    unet_path, punet_path, moe_dir = tmp / "unet.pth", tmp / "punet.pth", tmp / "moe.pth"
    torch.save({"unet": RefUNet().state_dict()}, unet_path)
    cfg = stage2_cfg(model_type, n_experts, dropout=0.0, unet_path=str(unet_path), **cfg_kw)
    pu = RefPU(**{**cfg.punet, "inter_repr": model_type == "punet_inter"})
    torch.save({"model": pu.state_dict()}, punet_path)
    return unet_path, punet_path, moe_dir


def run_punet_case(ref_moe, ref_loss, name, model_type, batch, size, future_frames, train=True):
    tmp = REPO / "build" / "golden_tmp"
    kw = dict(future_frames=future_frames)
    unet_path, punet_path, _ = _checkpoints(tmp, kw, model_type, 2)
    torch.manual_seed(0)
    cfg = stage2_cfg(model_type, 2, dropout=0.0, unet_path=str(unet_path), punet_path=str(punet_path), **kw)
    model = ref_moe.get_model(cfg)
    weights.fill_state_dict(model, seed=0)
    model.train(train)
    inp = weights.make_inputs(batch, size, size, seed=1234)
    out = {"meta": dict(name=name, type=model_type, n_experts=2, batch=batch, size=size, train=train, weight_seed=0,
                        input_seed=1234, future_frames=future_frames),
           "state_dict_keys": list(model.state_dict().keys()),
           "state_dict_shapes": [tuple(v.shape) for v in model.state_dict().values()],
           "requires_grad": {k: p.requires_grad for k, p in model.named_parameters()}}
    if train:
        actions, speeds = model(inp["images"], inp["speed"], inp["command"])
        loss = ref_loss.punet_loss(actions, speeds, inp["control"], inp["target_speed"], cfg.loss_coefs)
        loss.backward()
        out["loss"] = loss.detach().clone()
        named = dict(model.named_parameters())
        out["grad_norms"] = {k: p.grad.norm().item() for k, p in named.items() if p.grad is not None}
        out["grad_slices"] = {k: named[k].grad.flatten()[:64].clone() for k in PUNET_SLICES if k in named}
        sd = model.state_dict()
        out["bn_after_1"] = {f"{b}.{leaf}": sd[f"{b}.{leaf}"].clone() for b in PUNET_BN
                             for leaf in ("running_mean", "running_var", "num_batches_tracked") if f"{b}.{leaf}" in sd}
    else:
        with torch.no_grad():
            actions, speeds = model(inp["images"], inp["speed"], inp["command"])
    out["actions"], out["speeds"] = actions.detach().clone(), speeds.detach().clone()
    if model_type == "punet":                       # the frozen PU-Net's own output, sub-sampled, eval mode
        with torch.no_grad():
            model.eval()
            out["punet_masks_eval"] = model.punet(inp["images"])[:, :, :, ::8, ::8].clone()
            model.train(train)
    return out


STAGE1_SLICES = [
    "entry_block.layer1.eca1.conv.weight", "entry_block.layer1.conv1.0.weight", "entry_block.layer1.conv1.1.weight",
    "entry_block.layer2.eca2.conv.weight", "entry_block.layer2.conv2.0.weight", "entry_block.layer2.conv2.1.weight",
    "entry_block.layer2.conv2.1.bias", "pred_unet.dwn_1.0.weight", "pred_unet.dwn_1.1.weight", "pred_unet.dwn_3.3.weight",
    "pred_unet.dwn_5.0.weight", "pred_unet.dwn_5.4.bias", "pred_unet.up_1.weight", "pred_unet.up_1.bias",
    "pred_unet.up_forw_1.0.weight", "pred_unet.up_3.weight", "pred_unet.up_4.bias", "pred_unet.up_forw_4.3.weight",
    "pred_unet.up_forw_4.4.weight", "pred_unet.out.weight", "pred_unet.out.bias",
]
STAGE1_BN = ["unet.dwn_1.1", "unet.up_forw_4.4", "entry_block.layer1.conv1.1", "entry_block.layer2.conv2.1",
             "pred_unet.dwn_1.1", "pred_unet.dwn_5.4", "pred_unet.up_forw_4.4"]


def run_stage1_case(ref_loss, name, batch, size, future_frames, loss_type="tversky"):
    """Stage-1 PU-Net training step (train_1.py:129-141): PredictiveUnet in train mode, AutoregressiveCriterion, backward
    through the autoregressive chain.  ``unet`` is frozen (punet.py:46-48) but follows ``model.train()``."""
    from model.blocks.unet import UNet as RefUNet
    from model.punet import PredictiveUnet as RefPU
    tmp = REPO / "build" / "golden_tmp"
    tmp.mkdir(parents=True, exist_ok=True)
    torch.save({"unet": RefUNet().state_dict()}, tmp / "unet.pth")
    torch.manual_seed(0)
    model = RefPU(past_frames=4, future_frames=future_frames, in_features=3, num_classes=23, gamma=2, b=1,
                  model_name="unet", model_path=str(tmp / "unet.pth"))
    weights.fill_state_dict(model, seed=0)
    model.train()
    inp = weights.make_inputs(batch, size, size, seed=1234)
    target = weights.make_seg_targets(batch, future_frames, size, size, 23, seed=4321)
    crit = ref_loss.AutoregressiveCriterion(future_frames, loss_type)
    out = model(inp["images"])
    out.retain_grad()
    loss = crit(out, target)
    loss.backward()
    named = dict(model.named_parameters())
    sd = model.state_dict()
    res = {"meta": dict(name=name, batch=batch, size=size, future_frames=future_frames, loss_type=loss_type, weight_seed=0,
                        input_seed=1234, target_seed=4321),
           "state_dict_keys": list(sd.keys()),
           "requires_grad": {k: p.requires_grad for k, p in named.items()},
           "out_sub": out.detach()[..., ::4, ::4].clone(), "out_norm": out.detach().norm().item(),
           "loss": loss.detach().clone(),
           "dout_sub": out.grad[..., ::4, ::4].clone(), "dout_norm": out.grad.norm().item(),
           "grad_norms": {k: p.grad.norm().item() for k, p in named.items() if p.grad is not None},
           "grad_slices": {k: named[k].grad.flatten()[:64].clone() for k in STAGE1_SLICES},
           "bn_after_1": {f"{b}.{leaf}": sd[f"{b}.{leaf}"].clone() for b in STAGE1_BN
                          for leaf in ("running_mean", "running_var", "num_batches_tracked")}}
    return res


def segloss_cases(ref_loss):
    """AutoregressiveCriterion (loss.py:86-118) on random logits: loss and d loss / d logits for the three loss types."""
    g = torch.Generator().manual_seed(99)
    out = {}
    for nm, (b, f, h, w, scale) in {"a": (2, 2, 16, 16, 2.0), "b": (3, 1, 8, 24, 0.5)}.items():
        x = (torch.randn(b, f, 23, h, w, generator=g) * scale)
        t = weights.make_seg_targets(b, f, h, w, 23, seed=7 + b, block=2)
        case = {"logits": x.clone(), "target": t}
        for lt in ("tversky", "l1", "l2"):
            xi = x.clone().requires_grad_(True)
            l = ref_loss.AutoregressiveCriterion(f, lt)(xi, t)
            l.backward()
            case[lt] = dict(loss=l.detach().clone(), dlogits=xi.grad.clone())
        out[nm] = case
    return out


def run_pmoe_case(ref_moe, ref_loss, name, batch, size, future_frames, n_experts):
    """type 'pmoe' (moe.py:326-363) with exclude_freeze = [lat_weights, long_weights] (stage_2_pmoe.yaml:81) and no
    pretrained PU-Net action model: the blend weights and the PU-Net expert's heads / backbone train."""
    tmp = REPO / "build" / "golden_tmp"
    kw = dict(future_frames=future_frames)
    unet_path, punet_path, moe_dir = _checkpoints(tmp, kw, "punet", n_experts)
    cfg0 = stage2_cfg("pmoe", n_experts, dropout=0.0)
    torch.save(ref_moe.MixtureOfExperts(cfg0).state_dict(), moe_dir)
    cfg = stage2_cfg("pmoe", n_experts, dropout=0.0, unet_path=str(unet_path), punet_path=str(punet_path),
                     moe_dir=str(moe_dir), exclude_freeze=["lat_weights", "long_weights"], **kw)
    model = ref_moe.get_model(cfg)
    weights.fill_state_dict(model, seed=0)
    model.train()
    inp = weights.make_inputs(batch, size, size, seed=1234)
    out = {"meta": dict(name=name, type="pmoe", n_experts=n_experts, batch=batch, size=size, train=True, weight_seed=0,
                        input_seed=1234, future_frames=future_frames, sample_seed=77),
           "state_dict_keys": list(model.state_dict().keys()),
           "state_dict_shapes": [tuple(v.shape) for v in model.state_dict().values()],
           "requires_grad": {k: p.requires_grad for k, p in model.named_parameters()}}
    # one pass to learn what dists.sample() draws under the seed (BN buffers are restored afterwards)
    sd0 = {k: v.clone() for k, v in model.state_dict().items()}
    torch.manual_seed(77)
    with torch.no_grad():
        pa, _ = model.punet(inp["images"], inp["speed"], inp["command"])
        dists, _ = model.moe(inp["images"], inp["speed"], inp["command"])
        out["moe_actions"] = dists.sample().clone()
        out["punet_actions"] = pa.clone()
        out["probs"] = dists.mixture_distribution.probs.clone()
        out["mean"] = dists.component_distribution.base_dist.loc.clone()
        out["std"] = dists.component_distribution.base_dist.scale.clone()
    model.load_state_dict(sd0)
    torch.manual_seed(77)
    actions, dummy = model(inp["images"], inp["speed"], inp["command"])
    assert dummy == -1
    loss = ref_loss.pmoe_loss(actions, dummy, inp["control"], inp["target_speed"], cfg.loss_coefs)
    loss.backward()
    out["actions"], out["loss"] = actions.detach().clone(), loss.detach().clone()
    named = dict(model.named_parameters())
    out["grad_norms"] = {k: p.grad.norm().item() for k, p in named.items() if p.grad is not None}
    out["grads_small"] = {k: named[k].grad.clone() for k in ("lat_weights.weight", "lat_weights.bias", "long_weights.weight",
                                                             "long_weights.bias", "punet.action_pred.1.weight",
                                                             "punet.action_pred.1.bias")}
    return out


def micro_cases(ref_basics, ref_loss):
    """Layer-level fixtures (G5): ECA kernel sizes, make_mlp layouts, ECA forward, moe_loss values."""
    out = {}
    out["eca_k"] = {c: ref_basics.EfficientBlock(c).conv.kernel_size[0] for c in (12, 64, 92, 138, 512)}
    layouts = {}
    for bn in (False, True):
        for p in (0.0, 0.3):
            for dims in ([6, 512, 512], [1536, 512, 512, 1]):
                m = ref_basics.make_mlp(dims, "relu", False, bn, p)
                layouts[(bn, p, tuple(dims))] = list(m.state_dict().keys())
    out["mlp_layouts"] = layouts
    g = torch.Generator().manual_seed(7)
    eca = ref_basics.EfficientBlock(64)
    with torch.no_grad():
        eca.conv.weight.copy_(torch.tensor([[[0.3, -0.7, 0.5]]]))
    x = torch.randn(2, 64, 5, 7, generator=g)
    out["eca_x_seed"] = 7
    out["eca_y"] = eca(x).detach()
    # moe_loss on explicit mixture parameters
    import torch.distributions as D
    probs = torch.softmax(torch.randn(5, 4, generator=g), -1)
    mean = torch.randn(5, 4, 2, generator=g)
    std = torch.rand(5, 4, 2, generator=g) + 0.3
    speeds = torch.randn(5, 4, 1, generator=g)
    act = torch.rand(5, 2, generator=g) * 2 - 1
    tgt = torch.rand(5, 1, generator=g)
    dist = D.MixtureSameFamily(D.Categorical(probs), D.Independent(D.Normal(mean, std), 1))
    out["loss_case"] = dict(probs=probs, mean=mean, std=std, speeds=speeds, act=act, tgt=tgt,
                            loss=ref_loss.moe_loss(dist, speeds, act, tgt.clone(), [0.7, 0.3]).detach())
    a2 = torch.tanh(torch.randn(5, 2, generator=g))
    sp1 = torch.randn(5, 1, generator=g)
    out["action_loss_case"] = dict(actions=a2, speeds=sp1, act=act, tgt=tgt,
                                   punet_loss=ref_loss.punet_loss(a2, sp1, act, tgt, [0.7, 0.3]).detach(),
                                   pmoe_loss=ref_loss.pmoe_loss(a2, -1, act, tgt, [0.7, 0.3]).detach())
    sp2 = torch.randn(5, 1, generator=g)
    out["loss_case_shared"] = dict(probs=probs, mean=mean, std=std, speeds=sp2, act=act, tgt=tgt,
                                   loss=ref_loss.moe_loss(dist, sp2, act, tgt.clone(), [0.7, 0.3]).detach())
    return out


def main():
    ref_moe, ref_basics, ref_loss = import_reference()
    gold = REPO / "tests" / "golden"
    gold.mkdir(parents=True, exist_ok=True)
    cases = [
        ("g1_moe_e4_b2_128", "moe", 4, 2, 128, True, 5),
        ("g2_moe_e4_b1_224_eval", "moe", 4, 1, 224, False, 0),
        ("g3_moe_e8_b2_128", "moe", 8, 2, 128, True, 0),
        ("g4_moealt_e4_b2_64", "moe_alt", 4, 2, 64, True, 0),
        ("g5_moe_e3_b3_96", "moe", 3, 3, 96, True, 0),
    ]
    only = set(sys.argv[1:])          # optional: regenerate just the named cases
    cases.append(("g6_moeshared_k4_b6_96", "moe_shared", 4, 6, 96, True, 3))
    cases.append(("g7_moeshared_k6_b1_224_eval", "moe_shared", 6, 1, 224, False, 0))
    cases.append(("g10_moe_e4_b32_64", "moe", 4, 32, 64, True, 0))      # realistic batch statistics (bf16 tolerance case)
    cases.append(("g8_moeshared_k3_b4_128", "moe_shared", 3, 4, 128, True, 0))
    cases.append(("g9_moeshared_k5_b8_64", "moe_shared", 5, 8, 64, True, 0))
    # round 3: BASELINE config 1's image size with a batch of 8 -- layer4 statistics over 128 values,
    # every other BatchNorm over >= 512; the best-conditioned bf16 case (oracle/probe_conditioning.py).  Measured (round 4,
    # oracle/make_bounds.py): even the bf16-storage EMULATION is 2.1e-2 off float64 on `mean` here (probs 5.5e-3, std 1.1e-2,
    # speeds 1.4e-2, worst of 14 draws), so north_star's flat 1e-2 is met on `probs` only; tests/test_model_gpu.py prints both
    cases.append(("g11_moe_e4_b8_128", "moe", 4, 8, 128, True, 0))
    for name, t, e, b, s, train, steps in cases:
        if only and name not in only:
            continue
        res = run_case(ref_moe, ref_loss, name, t, e, b, s, train, steps)
        torch.save(res, gold / f"{name}.pt")
        print(name, "loss" in res and float(res["loss"]), res["probs"][0].tolist())
    pcases = [("p1_punet_b2_64_f2", "punet", 2, 64, 2, True), ("p2_punet_b1_64_f6_eval", "punet", 1, 64, 6, False),
              ("p3_punetinter_b2_64_f2", "punet_inter", 2, 64, 2, True), ("p4_punet_b3_96_f3", "punet", 3, 96, 3, True),
              ("p6_punet_b8_96_f2", "punet", 8, 96, 2, True)]         # round 3: batch of 8
    for name, t, b, sz, f, train in pcases:
        if only and name not in only:
            continue
        res = run_punet_case(ref_moe, ref_loss, name, t, b, sz, f, train)
        torch.save(res, gold / f"{name}.pt")
        print(name, "loss" in res and float(res["loss"]), res["actions"].tolist())
    if not only or "p5_pmoe_e2_b2_64_f2" in only:
        res = run_pmoe_case(ref_moe, ref_loss, "p5_pmoe_e2_b2_64_f2", 2, 64, 2, 2)
        torch.save(res, gold / "p5_pmoe_e2_b2_64_f2.pt")
        print("p5_pmoe_e2_b2_64_f2", float(res["loss"]), res["actions"].tolist())
    for name, b, sz, f in [("s1_stage1_b3_32_f3", 3, 32, 3), ("s2_stage1_b8_32_f2", 8, 32, 2)]:
        if only and name not in only:
            continue
        res = run_stage1_case(ref_loss, name, b, sz, f)
        torch.save(res, gold / f"{name}.pt")
        print(name, float(res["loss"]), res["out_norm"], res["dout_norm"])
    if not only or "s0_segloss" in only:
        torch.save(segloss_cases(ref_loss), gold / "s0_segloss.pt")
    if not only or "micro" in only:
        torch.save(micro_cases(ref_basics, ref_loss), gold / "micro.pt")
    import shutil
    shutil.rmtree(REPO / "build" / "golden_tmp", ignore_errors=True)
    print("wrote", sorted(p.name for p in gold.glob("*.pt")))


if __name__ == "__main__":
    main()
