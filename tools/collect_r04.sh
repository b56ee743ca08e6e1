#!/bin/bash
# Round-4 evidence (GPU box): rocprofv3 kernel trace of the bench command, HBM traffic from PMC (separate passes), MFMA / LDS
# counters of the conv kernels (program directly after `--`: no env / shell hop under the profiler).
set -o pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r04"; mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o r04 -- python3 "$R/bench.py" --no-stage1 --no-cpu-baseline --no-sub-configs > "$O/r04_bench_line_under_rocprof.json" 2> "$O/rocprof_stderr.log"
echo "rocprof rc=$?"
find "$O/prof" -name "*kernel_stats.csv" -exec cp {} "$O/r04_bench_kernel_stats.csv" \;
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch" -o f -- python3 "$R/bench.py" --steps 4 --warmup 1 --only-steps > "$O/pmc_f.log" 2>&1
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write" -o w -- python3 "$R/bench.py" --steps 4 --warmup 1 --only-steps > "$O/pmc_w.log" 2>&1
echo "write rc=$?"
python3 "$R/tools/collect_traffic.py" "$O/pmc_fetch" "$O/pmc_write" 5 && cp "$R/profiles/traffic.json" "$O/traffic.json"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/pmc_sq" -o s -- python3 "$R/tools/bench_conv.py" conv2 l1 l2s2 l2 l3 l4 > "$O/pmc_sq.log" 2>&1
echo "sq rc=$?"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$O/pmc_f8" -o s -- python3 "$R/tools/bench_conv_f8.py" > "$O/pmc_f8.log" 2>&1
echo "f8 rc=$?"
python3 "$R/tools/collect_pmc.py" "$O/pmc_sq" "$O/pmc_f8" --out=r04_conv_pmc.json && cp "$R/profiles/r04_conv_pmc.json" "$O/r04_conv_pmc.json"
rm -rf "$O/pmc_fetch" "$O/pmc_write" "$O/prof" "$O/pmc_sq" "$O/pmc_f8"
