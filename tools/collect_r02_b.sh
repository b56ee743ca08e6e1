#!/bin/bash
# Round-2 evidence, part B (GPU box): rocprofv3 kernel trace of the bench command, HBM traffic from PMC (separate passes),
# cycle stamps of the LDS-DMA conv kernel, in-process A/B of every kernel switch.
set -o pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r02"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o r02 -- python3 "$R/bench.py" --no-stage1 --no-cpu-baseline --no-sub-configs > "$O/r02_bench_line_under_rocprof.json" 2> "$O/rocprof_stderr.log"
echo "rocprof rc=$?"
find "$O/prof" -name "*kernel_stats.csv" -exec cp {} "$O/r02_bench_kernel_stats.csv" \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o f -- python3 "$R/bench.py" --steps 4 --warmup 1 --only-steps > "$O/pmc_f.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o w -- python3 "$R/bench.py" --steps 4 --warmup 1 --only-steps > "$O/pmc_w.log" 2>&1
python3 "$R/tools/collect_traffic.py" "$O/pmc_fetch" "$O/pmc_write" 5 && cp "$R/profiles/traffic.json" "$O/traffic.json"
rm -rf "$O/pmc_fetch" "$O/pmc_write" "$O/prof"
cd "$R"
python tools/stamp_conv.py l2 l3 l4 > "$O/r02_dma_cycle_stamps.log" 2>&1; tail -6 "$O/r02_dma_cycle_stamps.log"
{ python tools/ab_conv.py PMOE_CONV_DMA 0 1 -- l2 l3 l4; python tools/ab_conv.py PMOE_RES_DMA 0 1 -- l1 conv2; python tools/ab_conv.py PMOE_WGRAD_DMA 0 1 -- l1 l2 l3 l4 conv2; python tools/ab_stem_tail.py; python tools/ab_c16.py; python tools/hbm_write_probe.py; } > "$O/r02_kernel_ab.log" 2>&1
tail -30 "$O/r02_kernel_ab.log"
