"""Helpers shared by the PU-Net / PMoE tests: the reference constructors read checkpoint files (punet.py:40,
moe.py:278,335), so the tests write files with the right key sets first (contents are overwritten by
oracle.weights.fill_state_dict afterwards).  The helpers themselves live in the product (``pmoe_amd.utils``: bench.py
needs them too and must not import test code)."""
from pmoe_amd.utils import build_product, write_checkpoints  # noqa: F401
