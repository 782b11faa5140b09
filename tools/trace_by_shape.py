"""Per-(kernel, grid) launch counts and mean durations from a rocprofv3 kernel-trace CSV.
usage: python tools/trace_by_shape.py <kernel_trace.csv> [steps | 0 = last step only] [name filter regex]"""
import collections
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if steps == 0:  # only the last full training step: the launches between the last two optimiser kernels
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "::adam_" in r["Kernel_Name"]]
    rows = rows[idx[-2] + 1:idx[-1] + 1]
    steps = 1
filt = re.compile(sys.argv[3]) if len(sys.argv) > 3 else None
g = collections.defaultdict(list)
for r in rows:
    m = re.search(r"(\w+_kernel(<[^>]*>)?)", r["Kernel_Name"])
    name = m.group(1) if m else r["Kernel_Name"][:40]
    if filt and not filt.search(name):
        continue
    wg = int(r["Workgroup_Size_X"])
    key = (name, int(r["Grid_Size_X"]) // wg, int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    g[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = 0.0
for k, v in sorted(g.items()):
    tot += sum(v) / steps / 1e3
    print(f"{k[0]:44s} grid {k[1]:>6d} {k[2]:>3d} {k[3]:>3d}  n/step {len(v) / steps:6.2f}  avg {sum(v) / len(v):8.1f} us  {sum(v) / steps / 1e3:7.3f} ms/step")
print(f"total {tot:.3f} ms/step")
