#!/usr/bin/env python3
"""Run-to-run spread of bench.py's ms_per_step on one box, with and without the HIP-event sampling of the GEMM launches."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
for extra in ([], ["--no-prof"]):
    vals = []
    for _ in range(n):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "200", "--warmup", "20", "--no-cpu-baseline",
                              "--no-extras"] + extra, capture_output=True, text=True, timeout=600)
        vals.append(json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])["ms_per_step"])
    print(" ".join(extra) or "default", " ".join(f"{v:.4f}" for v in vals), flush=True)
