# small-batch train step (bench.py --batch B), ms per step; usage: BS="64 100 128" bash tools/ab_small_batch.sh "ENV=.." ...
for cfg in "$@"; do
for B in ${BS:-64 100 192 320 500}; do
  env $cfg timeout -k 10 120 python bench.py --batch $B --steps 100 --warmup 20 --no-extras --no-cpu-baseline --no-prof 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$cfg B=$B', d['ms_per_step'], d['value'])"
done
done
