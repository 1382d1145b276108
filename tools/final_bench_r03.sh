#!/bin/bash
# Round 3, last pass: the default bench line, the PMC traffic passes of the two planes GEMM kernels, and the self-launched 2-rank
# rehearsals over gloo (both ranks on the one GPU) of the lifter and cycle workloads.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r03b
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err; echo bench rc=$?
POSELIFT_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 20 --warmup 5 --no-extras > $O/dp2_gloo.json 2> $O/dp2.err; echo dp2 rc=$?
POSELIFT_DIST_BACKEND=gloo python bench.py --gpus 2 --workload cycle --batch 16 --steps 3 --warmup 1 > $O/cycle_dp2_gloo.json 2> $O/cycle_dp2.err; echo cycle-dp2 rc=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-prof --launch eager > $O/pmc_fetch.json 2> $O/pmc_fetch.err; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-prof --launch eager > $O/pmc_write.json 2> $O/pmc_write.err; echo write rc=$?
cd $R
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv") $(find $O/pmc_write -name "*counter_collection.csv") f16x3 $O/traffic.json > $O/traffic_summary.txt 2>&1; echo traffic rc=$?
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/traffic_summary.txt
python - <<PY
import json
d = json.load(open("$O/bench_1gpu.json"))
print({k: d[k] for k in ("value", "ms_per_step")}, d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline_forward_gemm"]["avg_launch_us"], d["batch_64"]["ms_per_step"])
for k, v in d["other_modes"].items(): print(" ", k, v.get("poses_per_s") or v)
for f in ("dp2_gloo", "cycle_dp2_gloo"):
    d = json.load(open("$O/%s.json" % f)); print(f, d["n_gpus"], d["value"], d["unit"], d["ms_per_step"], d["config"]["workload"][-60:])
PY
