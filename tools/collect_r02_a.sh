#!/bin/bash
# Round-2 evidence, part A (GPU box): full GPU test suite with the parity report, smoke, the default bench line, whole-step A/B.
set -o pipefail
R="$(cd "$(dirname "$0")/.." && pwd)"; O="$R/gpurun_out/r02"; mkdir -p "$O"; cd "$R"
python -m pytest tests -m gpu -q -s -x > "$O/gpu_tests_full.log" 2>&1; echo "pytest rc=$?" | tee "$O/gpu_tests_rc.log"
tail -3 "$O/gpu_tests_full.log"
grep -E "^(g[0-9]+|p[0-9]|s[0-9])[_ ]|bf16|fp8" "$O/gpu_tests_full.log" | grep -v "^tests/" > "$O/r02_parity_report.log"
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee "$O/smoke.log"
python bench.py > "$O/r02_bench_line.json" 2> "$O/r02_bench_stderr.log"; tail -c 600 "$O/r02_bench_line.json"
bash tools/ab_step.sh > "$O/r02_ab_step.log" 2>&1; cat "$O/r02_ab_step.log"
