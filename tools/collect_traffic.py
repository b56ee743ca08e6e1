#!/usr/bin/env python3
"""HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE), as MI355X_MICROARCH.md prescribes:
separate --pmc passes, counter unit = KiB, FETCH doubled on gfx950.  Writes profiles/traffic.json.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_fetch -o f -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_write -o w -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline
  python3 $R/tools/collect_traffic.py $R/gpurun_out/pmc_fetch $R/gpurun_out/pmc_write
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
                name = row["Kernel_Name"].split("(")[0][:80]
                a = acc[name]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    return acc


def main():
    fdir, wdir = sys.argv[1], sys.argv[2]
    fetch, write = read(fdir, "FETCH_SIZE"), read(wdir, "WRITE_SIZE")
    per, conv_b, conv_n = {}, 0.0, 0
    for name in sorted(set(fetch) | set(write)):
        fb, fn = fetch.get(name, [0.0, 0])
        wb, wn = write.get(name, [0.0, 0])
        n = max(fn, wn, 1)
        f2, w = fb * 1024 * 2 / n, wb * 1024 / n           # KiB units; gfx950: FETCH_SIZE counts half of the bytes
        per[name] = {"launches": n, "fetch_bytes_per_launch_x2": f2, "write_bytes_per_launch": w}
        if "conv_igemm_kernel" in name or "conv3x3_res_kernel" in name:
            conv_b += (f2 + w) * n
            conv_n += n
    out = {"conv_igemm_bytes_per_launch": conv_b / max(conv_n, 1),
           "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB units x1024, FETCH doubled (gfx950 "
                   "correction, MI355X_MICROARCH.md HBM section); average over conv_igemm_kernel* and conv3x3_res_kernel* "
                   "launches of python bench.py --steps 2 --warmup 1 (tools/collect_traffic.py)",
           "per_kernel": per}
    dst = Path(__file__).resolve().parents[1] / "profiles" / "traffic.json"
    dst.write_text(json.dumps(out, indent=1))
    print("conv bytes/launch:", out["conv_igemm_bytes_per_launch"], "->", dst)


if __name__ == "__main__":
    main()
