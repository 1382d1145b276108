#!/bin/bash
# Round-2 measurement pass on the GPU box (from the repo root): default bench line, rocprofv3 kernel stats + step
# anatomy of the same command, PMC traffic passes (FETCH_SIZE and WRITE_SIZE separately).  Outputs under
# gpurun_out/final_r02/; the summaries are copied into profiles/ afterwards.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final_r02
rm -rf $O; mkdir -p $O
cd $R
python bench.py > $O/bench_1gpu.json 2> $O/bench_1gpu.err; echo bench rc=$?
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_bench -o b -- python3 $R/bench.py --steps 60 --warmup 10 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo trace rc=$?
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-prof > $O/pmc_fetch.json 2> $O/pmc_fetch.err; echo fetch rc=$?
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python3 $R/bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-prof > $O/pmc_write.json 2> $O/pmc_write.err; echo write rc=$?
cd $R
python tools/trace_anatomy.py $(find $O/prof_bench -name "*kernel_trace.csv") 10 50 > $O/anatomy.txt; echo anatomy rc=$?
python tools/pmc_traffic.py $(find $O/pmc_fetch -name "*counter_collection.csv") $(find $O/pmc_write -name "*counter_collection.csv") f16x3 $O/traffic.json > $O/traffic_summary.txt 2>&1; echo traffic rc=$?
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete
build/planes_gemm 1 2 > $O/planes_gemm_ubench_16x16x32.txt 2>&1; echo ubench rc=$?
cat $O/anatomy.txt; cat $O/traffic_summary.txt
echo final-profiles-done
