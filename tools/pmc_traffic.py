#!/usr/bin/env python3
"""HBM-side bytes per launch of the GEMM kernels from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE: they do not fit
one pass on gfx950), corrected as MI355X_MICROARCH.md prescribes (counters in KB; FETCH_SIZE reports half the bytes of
a wide coalesced read stream: reads = 2 x FETCH_SIZE; WRITE_SIZE is exact).

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <dtype> [traffic.json]
Merges {dtype: {"dual": .., "forward": ..}} into profiles/traffic.json (bench.py copies it into roofline.traffic)."""
import collections
import csv
import json
import os
import sys


def per_kernel(path, counter):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            if row.get("Counter_Name") != counter:
                continue
            name = row["Kernel_Name"]
            a = acc[name]
            a[0] += float(row["Counter_Value"])
            a[1] += 1
    return {k: v[0] / v[1] for k, v in acc.items() if v[1]}


def pick(d, *subs):
    for k, v in d.items():
        if all(s in k for s in subs):
            return v
    return None


def main():
    fetch, write, dtype = sys.argv[1], sys.argv[2], sys.argv[3]
    out = sys.argv[4] if len(sys.argv) > 4 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                            "profiles", "traffic.json")
    f, w = per_kernel(fetch, "FETCH_SIZE"), per_kernel(write, "WRITE_SIZE")
    if dtype in ("f16x3", "bf16"):
        # (round 3: the forward GEMM is the pair-staged NT kernel; round 2's name as the fallback)
        # (round 3: dX + dW run chained in one workgroup per CU; round 2's names as the fallback)
        names = {"dual": ("planes_gemm_chain_kernel",), "forward": ("planes_gemm_wide_kernel",)}
        if pick(f, "planes_gemm_chain_kernel") is None:
            names["dual"] = ("planes_gemm_dual_kernel",)
        if pick(f, "planes_gemm_wide_kernel") is None:
            names["forward"] = ("planes_gemm_kernel", "Lb0ELb0E")
    else:
        names = {"dual": ("dual_kernel",), "forward": ("gemm_x6_planes_kernel",) if dtype == "bf16x6" else ("gemm_f32_kernel",)}
    rec = {}
    for key, subs in names.items():
        fk, wk = pick(f, *subs), pick(w, *subs)
        if fk is None or wk is None:
            fk, wk = pick(f, subs[0]), pick(w, subs[0])
        if fk is None or wk is None:
            continue
        rec[key] = {"FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1),
                    "hbm_bytes_per_launch": int(round((2 * fk + wk) * 1024))}
    data = json.load(open(out)) if os.path.exists(out) else {}
    data.setdefault("per_dtype", {})[dtype] = rec
    data["_comment_per_dtype"] = ("per_dtype[dtype][dual|forward].hbm_bytes_per_launch = (2 x FETCH_SIZE + WRITE_SIZE) KB from "
                                  "separate rocprofv3 --pmc passes of `bench.py --steps 20 --no-extras` (tools/final_profiles_r03.sh)")
    json.dump(data, open(out, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
