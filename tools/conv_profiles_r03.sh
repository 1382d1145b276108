#!/bin/bash
# Round-3 conv-path measurement pass (from the repo root on the GPU box): Model_3D training step at B = 256 / 32 / 8 beside
# PyTorch-ROCm eager, the phase5 cycle workload of bench.py (with and without Flip), the conv GEMM shapes with and without the
# persistent form, and rocprofv3 kernel stats of four B = 256 steps.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/conv_r03
rm -rf $O; mkdir -p $O
cd $R
python tools/bench_model3d_train.py --B 256 --iters 5 > $O/m3d_b256.txt 2>&1; echo b256 rc=$?
python tools/bench_model3d_train.py --B 32 --iters 10 --graph --skip-eager > $O/m3d_b32.txt 2>&1; echo b32 rc=$?
python tools/bench_model3d_train.py --B 8 --iters 20 --graph > $O/m3d_b8.txt 2>&1; echo b8 rc=$?
python bench.py --workload cycle > $O/cycle_b128.json 2> $O/cycle_b128.err; echo cycle rc=$?
python bench.py --workload cycle --flip --no-cpu-baseline > $O/cycle_b128_flip.json 2> $O/cycle_flip.err; echo cycle-flip rc=$?
python tools/bench_cycle.py --B 128 > $O/cycle_vs_eager_b128.txt 2>&1; echo cycle-eager rc=$?
(echo "persistent form (default)"; python tools/bench_conv_gemm.py 20; echo; echo "one workgroup per tile (POSELIFT_PERSIST=0)"; POSELIFT_PERSIST=0 python tools/bench_conv_gemm.py 20) > $O/conv_gemm_shapes.txt 2>/dev/null; echo shapes rc=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o t -- python3 $R/tools/run_model3d_train.py 256 4 f16x3 > /dev/null 2>&1; echo prof rc=$?
cd $R
cp $(find $O/prof -name "*kernel_stats.csv") $O/m3d_b256_f16x3_kernel_stats.csv
python tools/trace_by_shape.py $(find $O/prof -name "*kernel_trace.csv") 4 > $O/m3d_b256_f16x3_by_shape.txt
find $O/prof -name "*.csv" -delete
grep -h "B=" $O/m3d_b256.txt $O/m3d_b32.txt $O/m3d_b8.txt $O/cycle_vs_eager_b128.txt
python - <<PY
import json
for f in ("cycle_b128", "cycle_b128_flip"):
    d = json.load(open("$O/%s.json" % f))
    print(f, d["value"], d["unit"], d["ms_per_step"], d["roofline"]["achieved"], d.get("cpu_baseline", {}).get("value"))
PY
head -20 $O/m3d_b256_f16x3_by_shape.txt
echo conv-profiles-done
