"""Deterministic synthetic weights and inputs shared by golden generation and parity tests.

TEST INFRASTRUCTURE (see oracle/__init__.py).  Weights are never stored in fixtures: both sides
(the imported reference when goldens are made, the oracle / HIP path when they are checked) fill
every state_dict entry from ``fill_state_dict`` with a CPU ``torch.Generator`` seeded by
crc32(name) ^ seed, which is bit-identical across processes and machines with the same torch.
"""
import zlib

import torch


def _gen(name, seed):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(name.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def fill_tensor(name, ref, seed=0):
    """Return a tensor shaped/dtyped like ``ref`` for state_dict key ``name``."""
    g = _gen(name, seed)
    shape = tuple(ref.shape)
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=ref.dtype)
    if leaf == "running_mean":
        return 0.05 * torch.randn(shape, generator=g)
    if leaf == "running_var":
        return 1.0 + 0.1 * torch.rand(shape, generator=g)
    if ref.dim() == 1:
        is_bn_gamma = leaf == "weight"
        base = 1.0 if is_bn_gamma else 0.0
        if ".alpha." in name and leaf == "bias" and shape == (1,):
            base = 1.5                      # keep most gate ReLUs (moe.py:100) active, some dead
        return base + 0.1 * torch.randn(shape, generator=g)
    if ref.dim() == 3:                       # ECA Conv1d(1,1,k)
        return 0.5 * torch.randn(shape, generator=g)
    fan_in = 1
    for d in shape[1:]:
        fan_in *= d
    std = (2.0 / fan_in) ** 0.5
    if shape[0] <= 4:                        # tiny output layers: keep logits O(1)
        std = (1.0 / fan_in) ** 0.5
    return std * torch.randn(shape, generator=g)


@torch.no_grad()
def fill_state_dict(module, seed=0):
    sd = module.state_dict()
    new = {k: fill_tensor(k, v, seed).to(v.dtype) for k, v in sd.items()}
    module.load_state_dict(new, strict=True)
    return module


def make_inputs(batch, height, width, n_frames=4, n_commands=6, seed=1234):
    """Synthetic batch in the contract of ``model/data_loader.py:216-300`` + ``train_2.py:138-145``
    (SURVEY.md section 8a H2 / section 8d): images U[0,1), speed & target_speed U[0,1), one-hot command,
    control U[-1,1)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    images = torch.rand(batch, n_frames, 3, height, width, generator=g)
    speed = torch.rand(batch, 1, generator=g)
    target_speed = torch.rand(batch, 1, generator=g)
    cmd_idx = torch.randint(0, n_commands, (batch,), generator=g)
    command = torch.nn.functional.one_hot(cmd_idx, n_commands).float()
    control = torch.rand(batch, 2, generator=g) * 2 - 1
    return dict(images=images, speed=speed, command=command, control=control, target_speed=target_speed)


def make_seg_targets(batch, frames, height, width, classes=23, seed=4321, block=4):
    """Synthetic future segmentation labels in the contract of ``model/data_loader.py`` (stage 1, ``load_measurements``
    False): int64 class indices [B,F,H,W], piecewise constant over ``block`` x ``block`` pixels."""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    coarse = torch.randint(0, classes, (batch, frames, (height + block - 1) // block, (width + block - 1) // block),
                           generator=g)
    return coarse.repeat_interleave(block, -2).repeat_interleave(block, -1)[..., :height, :width].contiguous()
