mkdir -p gpurun_out/r04
timeout -k 10 1000 python -m pytest tests/test_gpu_general.py tests/test_gpu_synonyms.py tests/test_gpu_bigtable.py -m gpu -x -q > gpurun_out/r04/near8_tests.log 2>&1; echo tests rc $?; tail -3 gpurun_out/r04/near8_tests.log
for w in 8 10; do
  for v in 0 1; do
    FS_LSH_FULL_GRID=$v timeout -k 10 300 python tools/step_bench.py --window $w --steps 60 --inflight 1 "FS_LANES=1" 2>&1 | grep variant | cut -c1-220
    FS_LSH_FULL_GRID=$v timeout -k 10 300 python tools/step_bench.py --window $w --steps 100 --inflight 4 "FS_LANES=4" 2>&1 | grep variant | cut -c1-220
  done
done
for v in 0 1; do FS_LSH_FULL_GRID=$v timeout -k 10 300 python tools/lsh_bench.py --works 5000 --reps 3 2>&1 | grep "^{" | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('syn full_grid=$v', d['total_ms'], d['rows'])"; done
