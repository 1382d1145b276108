# same-box A/B of round-2 levers (bench.py headline step, ms per step).  usage: bash tools/ab_round2.sh "CFG1" "CFG2" ...
timeout -k 10 800 python -m pytest tests -q -m gpu -x > gpurun_out/t_ab.log 2>&1; echo rc=$?; tail -4 gpurun_out/t_ab.log
for rep in 1 2; do
for cfg in "$@"; do
  env $cfg timeout -k 10 120 python bench.py --steps 150 --warmup 30 --no-extras --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['ms_per_step'], d['value'], d['batch_64']['ms_per_step'])"
done
done
