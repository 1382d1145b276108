#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV per (kernel, grid): count, median/mean us, share.
    python tools/trace_summary.py gpurun_out/prof/run/123_kernel_trace.csv [steps]"""
import collections
import csv
import sys


def main():
    tr = list(csv.DictReader(open(sys.argv[1])))
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else None
    d = collections.defaultdict(list)
    for r in tr:
        name = r["Kernel_Name"]
        clean = name.replace("(anonymous namespace)::", "").replace("void ", "")
        short = clean.split("(")[0].split("<")[0].split("::")[-1][:30]
        if "gemm_" in clean and "<" in clean:
            short += "<" + clean.split("<")[1].split(">")[0] + ">"
        grid = "x".join(r[k] for k in ("Grid_Size_X", "Grid_Size_Y", "Grid_Size_Z"))
        d[(short, grid)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    tot = sum(sum(v) for v in d.values())
    print(f"{'kernel':46s} {'grid':>14s} {'n':>5s} {'median us':>10s} {'mean us':>9s} {'total ms':>9s} {'share':>6s}")
    for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
        v2 = sorted(v)
        print(f"{k[0]:46s} {k[1]:>14s} {len(v):5d} {v2[len(v2) // 2]:10.1f} {sum(v) / len(v):9.1f} "
              f"{sum(v) / 1e3:9.2f} {100 * sum(v) / tot:5.1f}%")
    print(f"total kernel time {tot / 1e3:.2f} ms" + (f" = {tot / steps:.1f} us/step over {steps} steps" if steps else ""))


if __name__ == "__main__":
    main()
