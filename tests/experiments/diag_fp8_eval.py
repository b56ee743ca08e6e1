#!/usr/bin/env python3
"""Prints tests/fp8_layerwise.py's table for a golden case (default: the eval-mode golden g2), bf16 and fp8 policy."""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from tests.fp8_layerwise import layerwise      # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "g2_moe_e4_b1_224_eval"
for fp8 in (False, True):
    rows, outs = layerwise(name, fp8)
    print(f"---- {'fp8 policy' if fp8 else 'bf16'}: rel-L2 to float64 (HIP | emulation), HIP vs emulation, elements beyond 28 (HIP | emulation), max")
    for r in rows:
        print(f"{r[0]:18s} {r[1]:.3e} | {r[2]:.3e}   {r[3]:.3e}   {r[4]} | {r[5]}   max {r[6]:.1f}")
    for k, v in outs.items():
        print(f"{k}: HIP err {v[0]:.3e}  emulation err {v[1]:.3e}  HIP vs emulation {v[2]:.3e}")
