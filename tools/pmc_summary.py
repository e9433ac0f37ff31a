#!/usr/bin/env python3
"""Per-kernel HBM-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over bench.py.

FETCH_SIZE and WRITE_SIZE are reported in KiB per dispatch.  On gfx950 FETCH_SIZE tallies the 128-byte
requests of wide coalesced reads at 64 bytes, so read bytes = 2 x FETCH_SIZE; WRITE_SIZE is exact for
16-byte-per-lane stores.  Output: JSON with average bytes per launch for every conv kernel symbol.
"""
import collections
import csv
import glob
import json
import sys


def per_kernel(root, counter):
    files = glob.glob(f"{root}/{counter}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: [0, 0.0])
    for f in files:
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            a = acc[row["Kernel_Name"]]
            a[0] += 1
            a[1] += float(row["Counter_Value"])
    return acc


# the conv kernels of the forward and the three kernels that make up most of a training iteration
KERNELS = ("conv3x3", "dense_block_band", "wgrad3x3")


def main():
    root = sys.argv[1]
    fetch, write = per_kernel(root, "FETCH_SIZE"), per_kernel(root, "WRITE_SIZE")
    out = {}
    for name in fetch:
        if not any(k in name for k in KERNELS):
            continue
        n, f = fetch[name]
        nw, w = write.get(name, (0, 0.0))
        rd = 2.0 * f * 1024 / n
        wr = w * 1024 / nw if nw else 0.0
        out[name] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                     "traffic_bytes_per_launch": round(rd + wr)}
    print(json.dumps({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE over bench.py --streams 1 --steps 2 --warmup 1; "
                                "read = 2 x FETCH_SIZE (gfx950 tallies 128-B read requests at 64 B), write = WRITE_SIZE",
                      "kernels": out}, indent=1))


if __name__ == "__main__":
    main()
