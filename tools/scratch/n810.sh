set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/n810
timeout -k 10 500 python3 -m pytest tests/test_gpu_general.py tests/test_gpu_bigtable.py -x -q -m gpu > gpurun_out/n810/tests.log 2>&1 || { tail -20 gpurun_out/n810/tests.log; exit 1; }
tail -2 gpurun_out/n810/tests.log
timeout -k 10 300 python3 tools/step_bench.py --window 8 --steps 100 --inflight 4 "FS_LSH_WMAP=1" "FS_LSH_WMAP=0" > gpurun_out/n810/n8.log 2>&1
timeout -k 10 300 python3 tools/step_bench.py --window 10 --steps 100 --inflight 4 "FS_LSH_WMAP=1" "FS_LSH_WMAP=2" "FS_LSH_GRAMTAB=0" > gpurun_out/n810/n10.log 2>&1
cat gpurun_out/n810/n8.log gpurun_out/n810/n10.log
