# round 3: the small-batch layer kernels (small_layer.hip): tests that reach them, then the same-box A/B at B = 64
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_dp.py -q -m gpu -k "ragged or cut_backward or philox or g2 or g3 or g4 or fused or small_batch" > gpurun_out/sl_tests.log 2>&1
grep -E "^FAILED|passed|failed|Max abs" gpurun_out/sl_tests.log
BS="64 32" bash tools/ab_small_batch.sh "POSELIFT_SMALL_LAYER=0" "POSELIFT_SMALL_F16=0" "POSELIFT_SMALL_LAYER=1" "POSELIFT_SMALL_LAYER=0" "POSELIFT_SMALL_F16=0" "POSELIFT_SMALL_LAYER=1" 2>&1 | tee gpurun_out/sl_ab.log
