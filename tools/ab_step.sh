#!/bin/bash
# Whole-step A/B on ONE box: the default build against the switches that restore round-1 behaviour, back to back, twice
# (boxes of the pool differ by 10 %: never compare bench lines across gpurun calls).   bash tools/ab_step.sh
cd "$(dirname "$0")/.."
B="python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-stage1 --no-sub-configs"
for r in 1 2; do
  for cfg in "default" "PMOE_CONV_DMA=0" "PMOE_RES_DMA=0" "PMOE_WGRAD_DMA=0" "PMOE_STEM_WALK=0" "PMOE_BN_MASK_IN_REDUCE=0" "PMOE_CONV_C16=0" "PMOE_FUSE_BN_GAP=0" "PMOE_CONV_DMA=0 PMOE_RES_DMA=0 PMOE_WGRAD_DMA=0 PMOE_STEM_WALK=0 PMOE_BN_MASK_IN_REDUCE=0 PMOE_CONV_C16=0 PMOE_FUSE_BN_GAP=0"; do
    if [ "$cfg" = "default" ]; then out=$($B 2>/dev/null | tail -1); else out=$(env $cfg $B 2>/dev/null | tail -1); fi
    echo "$cfg: $(echo "$out" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d["kernel_ms"]; print("ms/step %.2f  conv2d %.2f  wgrad %.2f" % (d["ms_per_step"], k["conv2d"], k["conv2d_wgrad"]))')"
  done
done
