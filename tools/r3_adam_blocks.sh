# round 3: workgroups of the AdamW launch (POSELIFT_ADAM_BLOCKS), B = 4096 and B = 64 steps, same box
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "small_batch or adamw or ragged" > gpurun_out/t.log 2>&1; grep -E "^FAILED|passed|failed" gpurun_out/t.log
for rep in 1 2; do for c in 256 512 1024 2048; do
  for B in 64 4096; do
    POSELIFT_ADAM_BLOCKS=$c python bench.py --batch $B --steps 200 --warmup 30 --no-extras --no-cpu-baseline --no-prof --launch eager | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('blocks=$c B=$B', d['ms_per_step'])"
  done
done; done
bash tools/anatomy.sh --batch 64 | tail -7
