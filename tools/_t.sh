for d in 4 10 6; do FS_LSH_DIAG=$d timeout -k 10 200 python tools/realistic_bench.py --oov 0.0 > gpurun_out/share_d$d.log 2>&1 || echo "variant $d failed"; done
