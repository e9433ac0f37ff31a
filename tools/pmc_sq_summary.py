#!/usr/bin/env python3
"""Per-kernel matrix-pipe and wave-state counters from one rocprofv3 --pmc pass over bench.py.

SQ_VALU_MFMA_BUSY_CYCLES counts cycles (32 per 32x32x16 MFMA), summed over the SIMDs of the chip;
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_ANY count quad-cycles per wave; GRBM_GUI_ACTIVE is the sum of the
8 XCDs' busy cycles, so kernel cycles = GRBM_GUI_ACTIVE / 8.  MFMA utilisation = MFMA busy cycles /
(kernel cycles x 256 CUs x 4 SIMDs).  Output: JSON, averages per launch for every conv kernel symbol.
"""
import collections
import csv
import glob
import json
import sys

CUS, SIMDS = 256, 4


# the conv kernels of the forward and the three kernels that make up most of a training iteration
KERNELS = ("conv3x3", "dense_block_band", "wgrad3x3")


def main():
    root = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
    for f in glob.glob(f"{root}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            a = acc[row["Kernel_Name"]][row["Counter_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    out = {}
    for name, ctr in acc.items():
        if not any(k in name for k in KERNELS):
            continue
        avg = {c: v[1] / v[0] for c, v in ctr.items()}
        n = max(v[0] for v in ctr.values())
        k = {"launches": n, "per_launch": {c: round(v, 1) for c, v in sorted(avg.items())}}
        if "GRBM_GUI_ACTIVE" in avg and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
            cyc = avg["GRBM_GUI_ACTIVE"] / 8.0
            k["kernel_cycles"] = round(cyc, 1)
            k["mfma_util"] = round(avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * CUS * SIMDS), 4)
        if "SQ_WAVE_CYCLES" in avg:
            w = avg["SQ_WAVE_CYCLES"]
            for c, label in (("SQ_WAIT_ANY", "wave_parked_frac"), ("SQ_WAIT_INST_ANY", "issue_stall_frac"),
                             ("SQ_ACTIVE_INST_ANY", "issuing_frac")):
                if c in avg:
                    k[label] = round(avg[c] / w, 4)
        if "SQ_LDS_BANK_CONFLICT" in avg and avg.get("SQ_LDS_IDX_ACTIVE"):
            k["lds_conflict_frac"] = round(avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"], 4)
        out[name] = k
    print(json.dumps({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY "
                                "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE "
                                "GRBM_GUI_ACTIVE over bench.py --streams 1 --steps 2 --warmup 1 (kernels serialised); "
                                "mfma_util = MFMA busy cycles / (GRBM_GUI_ACTIVE/8 x 256 CUs x 4 SIMDs)",
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
