#!/bin/bash
# Whole-step A/B of the round-3 switches on ONE box, back to back, twice (never compare bench lines across gpurun calls).
#   bash tools/ab_step3.sh "PMOE_BN_REDUCE_IN_DGRAD=0" ...
cd "$(dirname "$0")/.."
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-stage1 --no-sub-configs"
for r in 1 2; do
  for cfg in "default" "$@"; do
    if [ "$cfg" = "default" ]; then out=$($B 2>/dev/null | tail -1); else out=$(env $cfg $B 2>/dev/null | tail -1); fi
    echo "$cfg: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d["kernel_ms"]; print("ms/step %.2f (median %.2f)  conv2d %.2f  wgrad %.2f  bn_bwd_reduce %.2f  bn_bwd_apply %.2f  bn_apply %.2f  total kernels %.2f" % (d["ms_per_step"], d["ms_per_step_median"], k["conv2d"], k["conv2d_wgrad"], k.get("bn_bwd_reduce",0), k.get("bn_bwd_apply",0), k.get("bn_apply",0), d["kernel_ms_total"]))')"
  done
done
