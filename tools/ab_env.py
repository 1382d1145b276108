#!/usr/bin/env python3
"""A/B the bench under different environment switches on the SAME box, interleaved (clock / box variation
between gpurun calls is several percent):  python tools/ab_env.py [rounds] -- NAME=ENV1=1,ENV2=1 NAME2= ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds = int(args.pop(0)) if args and args[0].isdigit() else 2
if args and args[0] == "--":
    args.pop(0)
variants = []
for a in args:
    name, _, envs = a.partition("=")
    variants.append((name, dict(e.split("=", 1) for e in envs.split(",") if e)))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "100", "--warmup", "20",
                              "--no-cpu-baseline", "--no-extras"], env={**os.environ, **env}, capture_output=True,
                             text=True, timeout=600)
        line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
        d = json.loads(line)
        pair = (d.get("roofline") or {}).get("avg_launch_us", float("nan"))
        fwd = (d.get("roofline_forward_gemm") or {}).get("avg_launch_us", float("nan"))
        res[name].append((d["ms_per_step"], pair, fwd))
        print(f"round {r} {name:12s} ms/step {d['ms_per_step']:.4f}  pair {pair:.1f} us  "
              f"fwd {fwd:.1f} us  mpjpe {d['mpjpe_mm_eval_fwd_vs_oracle']}", flush=True)
for name, v in res.items():
    print(f"{name:12s} mean ms/step {sum(x[0] for x in v) / len(v):.4f}  pair {sum(x[1] for x in v) / len(v):.1f}  fwd {sum(x[2] for x in v) / len(v):.1f}")
