"""pmoe_amd: MI355X-native (gfx950) implementation of PMoE's stage-2 policy-network hot path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); every
computation of the path runs in the hand-written HIP kernels of ``libpmoe_hip.so``.
"""
__version__ = "0.1.0"
