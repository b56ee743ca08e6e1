#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes:
separate --pmc passes, counter unit = KiB, FETCH doubled on gfx950.  Writes profiles/traffic.json.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 4 --warmup 1 --only-steps
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 4 --warmup 1 --only-steps
  python3 $R/tools/collect_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write 5      (5 = steps the profiled command ran)
"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path


def read(dirpath, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for f in Path(dirpath).rglob("*counter_collection.csv"):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:80]
                a = acc[name]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    fetch, write = read(fdir, "FETCH_SIZE"), read(wdir, "WRITE_SIZE")
    per, conv_b, conv_n = {}, 0.0, 0
    for name in sorted(set(fetch) | set(write)):
        fb, fn = fetch.get(name, [0.0, 0])
        wb, wn = write.get(name, [0.0, 0])
        n = max(fn, wn, 1)
        f2, w = fb * 1024 * 2 / n, wb * 1024 / n           # KiB units; gfx950: FETCH_SIZE counts half of the bytes
        per[name] = {"launches": n, "fetch_bytes_per_launch_x2": f2, "write_bytes_per_launch": w}
        if "conv_igemm" in name or "conv3x3_" in name or "conv1x1_" in name:
            conv_b += (f2 + w) * n
            conv_n += n
    total = sum((v["fetch_bytes_per_launch_x2"] + v["write_bytes_per_launch"]) * v["launches"] for v in per.values())
    by_group = defaultdict(float)
    for k, v in per.items():
        grp = ("conv fwd/dgrad" if ("conv_igemm" in k or "conv3x3" in k or "conv1x1" in k or "gemm_skinny" in k) else "conv wgrad" if "wgrad" in k
               else "batchnorm" if ("bn_" in k or "colstats" in k or "reduce_partials" in k) else "stem tail" if "stem_tail" in k
               else "other")
        by_group[grp] += (v["fetch_bytes_per_launch_x2"] + v["write_bytes_per_launch"]) * v["launches"]
    out = {"conv_igemm_bytes_per_launch": conv_b / max(conv_n, 1),
           "steps_profiled": steps,
           "hbm_gb_per_step": round(total / steps / 1e9, 2) if steps else None,
           "hbm_gb_per_step_by_group": {k: round(v / steps / 1e9, 2) for k, v in by_group.items()} if steps else None,
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB units x1024, FETCH doubled (gfx950 "
                   "correction, MI355X_MICROARCH.md HBM section); average over conv_igemm_kernel* and conv3x3_res_kernel* "
                   "launches of python bench.py --steps 2 --warmup 1 (tools/collect_traffic.py)",
           "per_kernel": per}
    dst = Path(__file__).resolve().parents[1] / "profiles" / "traffic.json"
    dst.write_text(json.dumps(out, indent=1))
    print("conv bytes/launch:", out["conv_igemm_bytes_per_launch"], "->", dst)


if __name__ == "__main__":
    main()
