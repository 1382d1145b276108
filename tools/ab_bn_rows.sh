
timeout -k 10 700 python -m pytest tests -q -m gpu > gpurun_out/t5.log 2>&1; echo rc=$?; tail -4 gpurun_out/t5.log
for cfg in "POSELIFT_BN_UNFUSED=1" "POSELIFT_BN_ROWS=64" "POSELIFT_BN_ROWS=32" "POSELIFT_BN_ROWS=128" "POSELIFT_BN_ROWS=16" "POSELIFT_BN_UNFUSED=1" "POSELIFT_BN_ROWS=64"; do
  env $cfg timeout -k 10 120 python bench.py --steps 150 --warmup 30 --no-extras --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'])"
done
