mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_gpu_general.py tests/test_gpu_synonyms.py tests/test_gpu_bigtable.py -m gpu -x -q > gpurun_out/r04/near8_tests.log 2>&1; echo tests rc $?; tail -5 gpurun_out/r04/near8_tests.log
for w in 8 10; do
  for v in 1 0; do
    FS_SCAN_NEAR8=$v timeout -k 10 300 python tools/step_bench.py --window $w --steps 60 --inflight 1 "FS_LANES=1" 2>&1 | grep variant | cut -c1-220
    FS_SCAN_NEAR8=$v timeout -k 10 300 python tools/step_bench.py --window $w --steps 100 --inflight 4 "FS_LANES=4" 2>&1 | grep variant | cut -c1-220
  done
done
