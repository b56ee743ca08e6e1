#!/bin/bash
cd "$(dirname "$0")/.."
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-stage1 --no-sub-configs --no-kernel-profile"
for r in 1 2; do
  for cfg in "PMOE_OVERLAP_WGRAD=0" "PMOE_OVERLAP_WGRAD=1"; do
    out=$(env $cfg $B 2>/dev/null | tail -1)
    echo "$cfg: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step %.2f  median %.2f h1 %s" % (d["ms_per_step"], d["ms_per_step_median"], d.get("h1_step",{}).get("ms_per_step")))')"
  done
done
