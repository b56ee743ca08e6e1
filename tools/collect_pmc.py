#!/usr/bin/env python3
"""MFMA / LDS balance of the conv kernels from a rocprofv3 --pmc pass over tools/bench_conv.py:
    SQ_VALU_MFMA_BUSY_CYCLES  cycles with a matrix instruction executing (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32), summed over SIMDs
    GRBM_GUI_ACTIVE           busy cycles of the dispatch, summed over the 8 XCDs  -> / 8 = the kernel's duration in shader cycles
    SQ_LDS_IDX_ACTIVE / SQ_LDS_BANK_CONFLICT   LDS-array cycles / extra cycles from bank conflicts (per CU, summed)
MFMA utilisation = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs).  Writes profiles/r03_conv_pmc.json.
    python3 tools/collect_pmc.py <rocprof output dir> [<second dir: the same counters over tools/bench_conv_f8.py>]"""
import csv
import json
import sys
from collections import defaultdict
from pathlib import Path

SIMDS = 256 * 4


def main():
    acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    args = [x for x in sys.argv[1:] if not x.startswith("--out=")]
    out_name = next((x[6:] for x in sys.argv[1:] if x.startswith("--out=")), "r03_conv_pmc.json")
    for f in (f for d in args for f in Path(d).rglob("*counter_collection.csv")):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:90]
                if not any(k in name for k in ("conv3x3", "conv_wgrad", "conv_igemm", "conv1x1")):
                    continue
                a = acc[name][row["Counter_Name"]]
                a[0] += float(row["Counter_Value"])
                a[1] += 1
    out = {"command": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                      "--output-format csv -- python3 tools/bench_conv.py conv2 l1 l2s2 l2 l3 l4   (E=4, B=64 layer shapes of the "
                      "headline step, random data; forward, data gradient and weight gradient of each, 11 launches per kind); the e4m3 kernel "
                      "conv3x3_dma_f8_kernel from the same counters over python3 tools/bench_conv_f8.py (layer2-4 forward, beside its bf16 partner)",
           "units": "per launch; mfma_util = MFMA_BUSY / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs); lds_active / lds_conflict per CU "
                    "as a share of the kernel's cycles",
           "kernels": {}}
    for name, cs in sorted(acc.items()):
        per = {k: v[0] / max(v[1], 1) for k, v in cs.items()}
        n = max(v[1] for v in cs.values())
        cyc = per.get("GRBM_GUI_ACTIVE", 0.0) / 8
        rec = {"launches": n, "kernel_cycles": round(cyc), **{k: round(v) for k, v in per.items()}}
        if cyc:
            rec["mfma_util"] = round(per.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cyc * SIMDS), 4)
            rec["lds_active_per_cu_share"] = round(per.get("SQ_LDS_IDX_ACTIVE", 0.0) / 256 / cyc, 4)
            rec["lds_conflict_share_of_lds_cycles"] = round(per.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(per.get("SQ_LDS_IDX_ACTIVE", 1.0), 1.0), 4)
        out["kernels"][name] = rec
    dst = Path(__file__).resolve().parents[1] / "profiles" / out_name
    dst.write_text(json.dumps(out, indent=1))
    for k, v in out["kernels"].items():
        print(f"{k[:70]:70s} n={v['launches']:3d} mfma_util {v.get('mfma_util')}  lds {v.get('lds_active_per_cu_share')}  conflicts {v.get('lds_conflict_share_of_lds_cycles')}")


if __name__ == "__main__":
    main()
