# timing-only ablations of small_fwd_kernel (POSELIFT_SL_ABL: 1 no contraction, 2 loads only, 4 MFMAs only): kernel averages
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for abl in 0 1 2 4; do
  O=$R/gpurun_out/slabl$abl; rm -rf $O; mkdir -p $O
  export POSELIFT_SL_ABL=$abl
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o b -- python3 $R/bench.py --batch 64 --steps 60 --warmup 10 --no-extras --no-cpu-baseline --no-prof > $O/bench.json 2> $O/bench.err
  echo "ABL=$abl"; grep -E "small_fwd|small_bwd|bn_small_fwd" $O/prof/b_kernel_stats.csv | awk -F, '{print $1, $2, $4}'
  find $O -name "*kernel_trace.csv" -delete
done
