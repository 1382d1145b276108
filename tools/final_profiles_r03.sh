#!/bin/bash
# Round-3 measurement pass on the GPU box (from the repo root): default bench line, rocprofv3 kernel stats + step anatomy of
# the same command (B = 4096 and B = 64), PMC traffic passes (FETCH_SIZE and WRITE_SIZE separately).  Outputs under
# gpurun_out/final_r03/; the summaries are copied into profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r03
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err; echo bench rc=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python3 $R/bench.py --steps 60 --warmup 10 --no-extras --no-cpu-baseline --launch eager > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo trace rc=$?
rocprofv3 --kernel-trace --output-format csv -d $O/prof_b64 -o b -- python3 $R/bench.py --batch 64 --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-prof --launch eager > $O/bench_b64_under_rocprof.json 2> $O/bench_b64.err; echo trace64 rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-prof --launch eager > $O/pmc_fetch.json 2> $O/pmc_fetch.err; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-prof --launch eager > $O/pmc_write.json 2> $O/pmc_write.err; echo write rc=$?
cd $R
python tools/trace_anatomy.py $(find $O/prof_bench -name "*kernel_trace.csv") 10 50 > $O/anatomy.txt; echo anatomy rc=$?
python tools/trace_anatomy.py $(find $O/prof_b64 -name "*kernel_trace.csv") 10 50 > $O/anatomy_b64.txt; echo anatomy64 rc=$?
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv") $(find $O/pmc_write -name "*counter_collection.csv") f16x3 $O/traffic.json > $O/traffic_summary.txt 2>&1; echo traffic rc=$?
cp $(find $O/prof_bench -name "*kernel_stats.csv") $O/bench_kernel_stats.csv
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
cat $O/anatomy.txt; cat $O/anatomy_b64.txt | head -30; cat $O/traffic_summary.txt
echo final-profiles-done
