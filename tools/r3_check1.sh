#!/bin/bash
# round 3, first GPU check of the host-side changes: planes freshness tests, the headline bench, the cycle workload,
# self-launched 2-rank rehearsals over gloo (both ranks on the one GPU), and the refusal of --gpus 8 on a 1-GPU box
set -o pipefail
O=gpurun_out
python -m pytest tests/test_gpu_planes.py -m gpu -q -x > $O/r3_c1_planes.log 2>&1 || { tail -30 $O/r3_c1_planes.log; exit 1; }
tail -2 $O/r3_c1_planes.log
python bench.py --steps 100 --warmup 20 > $O/r3_bench1.json 2> $O/r3_bench1.err || { tail -20 $O/r3_bench1.err; exit 1; }
python bench.py --workload cycle > $O/r3_cycle1.json 2> $O/r3_cycle1.err || { tail -20 $O/r3_cycle1.err; exit 1; }
python bench.py --workload cycle --flip --no-cpu-baseline > $O/r3_cycle1_flip.json 2> $O/r3_cycle1_flip.err || { tail -20 $O/r3_cycle1_flip.err; exit 1; }
POSELIFT_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras > $O/r3_dp2.json 2> $O/r3_dp2.err || { tail -20 $O/r3_dp2.err; exit 1; }
POSELIFT_DIST_BACKEND=gloo python bench.py --gpus 2 --workload cycle --batch 16 --steps 3 --warmup 1 > $O/r3_cycle_dp2.json 2> $O/r3_cycle_dp2.err || { tail -20 $O/r3_cycle_dp2.err; exit 1; }
python bench.py --gpus 8 > $O/r3_gpus8.out 2>&1; echo "bench --gpus 8 on one GPU: rc=$? $(cat $O/r3_gpus8.out | tail -1)"
for f in r3_bench1 r3_cycle1 r3_cycle1_flip r3_dp2 r3_cycle_dp2; do echo "== $f"; python - <<PY
import json
d = json.load(open("$O/$f.json"))
print({k: d[k] for k in ("value", "unit", "n_gpus", "ms_per_step") if k in d}, d.get("roofline") and {k: d["roofline"][k] for k in ("achieved", "frac", "avg_launch_us") if k in d["roofline"]}, d.get("cpu_baseline") and d["cpu_baseline"]["value"])
PY
done
