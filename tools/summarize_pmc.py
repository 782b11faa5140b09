#!/usr/bin/env python3
"""Summarises two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same
bench command) into per-kernel HBM-side traffic per launch.

    python tools/summarize_pmc.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

FETCH_SIZE and WRITE_SIZE are reported in KiB.  On gfx950 FETCH_SIZE counts 128-byte read requests
as 64 bytes, so fetched bytes = 2 * FETCH_SIZE (the adam_kernel row checks this: it reads 16 bytes
and writes 12 bytes per parameter)."""
import collections
import csv
import json
import os
import sys


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = agg[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for k, (n, v) in fetch.items():
        wn, wv = write.get(k, (0, 0.0))
        rd = 2.0 * v * 1024 / n
        wr = wv * 1024 / wn if wn else 0.0
        out[k] = {"launches": n, "read_bytes_per_launch": round(rd), "write_bytes_per_launch": round(wr),
                  "traffic_bytes_per_launch": round(rd + wr)}
    out = dict(sorted(out.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"] * kv[1]["launches"]))
    # the kernel sources these counters were collected on (bench.py reports the traffic only for this build)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    out["_source_hash"] = bench.source_hash("conv_bf16.hip", "conv3x3_pipe.h", "wgrad_bf16.hip", "bf16_util.h")
    json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
