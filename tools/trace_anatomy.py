#!/usr/bin/env python3
"""Per-kernel anatomy of one train step from a rocprofv3 --kernel-trace database (results.db).

    python tools/trace_anatomy.py <results.db | *_kernel_trace.csv> [first_step] [n_steps]
Steps are delimited by the AdamW launch; prints launches per step, GPU-busy time per step and per-kernel totals."""
import collections
import re
import sqlite3
import sys


def short(n):
    n = n.replace("pl::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("void ", "")
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"plp::", "", n)
    return n[:64]


def main():
    if sys.argv[1].endswith(".csv"):                     # rocprofv3 --output-format csv: *_kernel_trace.csv
        import csv
        with open(sys.argv[1], newline="") as f:
            rows = sorted(((r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in csv.DictReader(f)),
                          key=lambda r: r[1])
    else:
        db = sqlite3.connect(sys.argv[1])
        rows = list(db.execute("select name, start, end from kernels order by start"))
    ad = [i for i, r in enumerate(rows) if "adamw" in r[0]]
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 50
    sel = rows[ad[first - 1] + 1:ad[first - 1 + n] + 1]
    agg = collections.OrderedDict()
    for name, s, e in sel:
        a = agg.setdefault(short(name), [0, 0])
        a[0] += 1
        a[1] += e - s
    tot = sum(v[1] for v in agg.values())
    span = sel[-1][2] - sel[0][1]
    print(f"{n} steps: {span / n / 1e3:.1f} us/step wall, {tot / n / 1e3:.1f} us/step in kernels, "
          f"{len(sel) / n:.1f} launches/step")
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{t / n / 1e3:8.1f} us/step  {c / n:5.1f} x {t / c / 1e3:7.2f} us  {k}")


if __name__ == "__main__":
    main()
