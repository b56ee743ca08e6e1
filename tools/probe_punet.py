#!/usr/bin/env python3
"""Print the PU-Net / PMoE parity reports (GPU).  python tools/probe_punet.py [f32|bf16]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

from tests.punet_parity import run_pmoe_case, run_punet_case  # noqa: E402

tmp = Path(__file__).resolve().parents[1] / "build" / "probe_punet"
dts = [torch.float32, torch.bfloat16] if len(sys.argv) < 2 else [dict(f32=torch.float32, bf16=torch.bfloat16)[sys.argv[1]]]
for dt in dts:
    for n in ("p1_punet_b2_64_f2", "p4_punet_b3_96_f3", "p3_punetinter_b2_64_f2", "p2_punet_b1_64_f6_eval"):
        try:
            run_punet_case(tmp, n, dt, fwd_tol_mult=1e9)
        except Exception as e:       # noqa: BLE001
            import traceback; traceback.print_exc()
            print("FAILED", n, dt, repr(e)[:300])
    try:
        run_pmoe_case(tmp, "p5_pmoe_e2_b2_64_f2", dt, fwd_tol_mult=1e9)
    except Exception as e:           # noqa: BLE001
        import traceback; traceback.print_exc()
        print("FAILED p5", dt, repr(e)[:300])
