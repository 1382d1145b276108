# per-kernel anatomy of the bench step on the GPU box: bash tools/anatomy.sh [extra bench.py flags]   (-> gpurun_out/anatomy/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/anatomy
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-prof --launch eager "$@" > $O/bench.json 2> $O/bench.err
cd $R
python tools/trace_anatomy.py $(find $O -name "*kernel_trace.csv") 10 50 > $O/anatomy.txt
cat $O/anatomy.txt
find $O -name "*kernel_trace.csv" -delete
