#!/usr/bin/env python3
"""rocprofv3 kernel trace grouped by (kernel, grid): which launches of a multi-shape kernel (the planes GEMM on the conv path)
carry the time.   python tools/trace_by_shape.py <kernel_trace.csv> [steps]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::|pl::", "", r["Kernel_Name"]).split("(")[0][:60]
    grid = int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])
    k = (name, grid)
    agg[k][0] += 1
    agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"total {tot / steps / 1e3:.2f} ms/step")
for (name, grid), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"{us / steps / 1e3:7.2f} ms  {n / steps:5.1f} x {us / n:8.1f} us  grid {grid}  {name}")
