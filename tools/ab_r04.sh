#!/bin/bash
# kernel-level interleaved A/B of the round-4 switches (one process per switch, one box): profiles/r04_kernel_ab.log
cd "$(dirname "$0")/.."
python tools/ab_conv.py PMOE_WGRAD_V2 0 1 -- l2 l3 l4 l1 conv2
python tools/ab_conv.py PMOE_DMA_PRODUCER 0 2 -- l2 l3 l4
python tools/ab_conv.py PMOE_WGRAD_AHEAD 5 6 -- l2 l3 l4 l1
python tools/ab_conv.py PMOE_DMA_STREAM 0 1 -- l2 l3 l4
