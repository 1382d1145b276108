for cfg in "X=1" "POSELIFT_CHAIN=0" "POSELIFT_WIDE=0" "POSELIFT_CHAIN=0 POSELIFT_WIDE=0"; do
  env $cfg python bench.py --dtype bf16 --steps 100 --warmup 20 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline_forward_gemm']['avg_launch_us'])"
done
