#!/bin/bash
# Runs on the GPU box (gpurun): the measurements behind DESIGN.md section 8, each into
# gpurun_out/r02/ under the name it keeps in profiles/.  Usage: tools/collect_profiles.sh part1|part2
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r02
mkdir -p $OUT
export TMPDIR=/tmp
part=${1:-part1}

prof_stats() {   # name, args...: rocprofv3 kernel stats of `bench.py args`
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof_$name -- \
      python3 $OLDPWD/bench.py "$@" > $OLDPWD/$OUT/${name}_bench_under_rocprof.json 2>/dev/null )
  cp $(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1) $OUT/${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
}

pmc() {          # name, counter, args...
  local name=$1 ctr=$2; shift 2
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OLDPWD/$OUT/pmc_${name}_$ctr -- \
      python3 $OLDPWD/bench.py "$@" > /dev/null 2>&1 )
  cp $(find $OUT/pmc_${name}_$ctr -name "*counter_collection.csv" | head -1) $OUT/${name}_pmc_$ctr.csv
  rm -rf $OUT/pmc_${name}_$ctr
}

if [ "$part" = part1 ]; then
  echo "== default bench line (N=1, c2, four distinct batches, 4 lanes)"
  python bench.py > $OUT/r02_c2_bench.json 2> $OUT/r02_c2_bench.err; tail -c 400 $OUT/r02_c2_bench.json; echo
  echo "== the driver's form: --steps 20"
  python bench.py --steps 20 --warmup 5 --no-companions --no-cpu-baseline > $OUT/r02_c2_bench_steps20.json 2>/dev/null
  echo "== one lane (searches one after the other) and one resident batch"
  python bench.py --lanes 1 --inflight 2 --no-companions --no-cpu-baseline > $OUT/r02_c2_bench_lanes1.json 2>/dev/null
  python bench.py --rotate 1 --no-companions --no-cpu-baseline > $OUT/r02_c2_bench_resident.json 2>/dev/null
  echo "== rocprofv3 kernel stats of the default command"
  prof_stats r02_c2 --no-companions --no-cpu-baseline
  prof_stats r02_c2_lanes1 --lanes 1 --inflight 2 --no-companions --no-cpu-baseline
  echo "== PMC: HBM traffic of k_scan_rows (separate passes)"
  pmc r02_c2 FETCH_SIZE --steps 8 --warmup 2 --inflight 1 --no-companions --no-cpu-baseline
  pmc r02_c2 WRITE_SIZE --steps 8 --warmup 2 --inflight 1 --no-companions --no-cpu-baseline
  python tools/pmc_to_traffic.py $OUT/r02_c2_pmc_FETCH_SIZE.csv $OUT/r02_c2_pmc_WRITE_SIZE.csv \
      $OUT/r02_c2_bench_steps20.json --out $OUT/scan_traffic.json
  echo "== c3shard: one GPU's share of configs[2] (12.5k works x 5k tokens, 250 MB)"
  python bench.py --workload c3shard --steps 100 --no-companions --no-cpu-baseline > $OUT/r02_c3shard_bench.json 2>/dev/null
  prof_stats r02_c3shard --workload c3shard --steps 60 --no-companions --no-cpu-baseline
  echo "== larger batches: the N = 4 and N = 2 shares of configs[2] (500 MB / 1 GB of ids per batch)"
  python bench.py --workload c3 --works 25000 --steps 40 --warmup 4 --no-companions --no-cpu-baseline > $OUT/r02_c3_quarter_bench.json 2>/dev/null
  python bench.py --workload c3 --works 50000 --steps 40 --warmup 4 --no-companions --no-cpu-baseline > $OUT/r02_c3_half_bench.json 2>/dev/null
  ls -la $OUT
fi

if [ "$part" = part2 ]; then
  echo "== configs[3]: n = 4, 8, 10 on the 10k-work corpus"
  for n in 4 8 10; do
    python bench.py --steps 100 --window $n --no-companions --no-cpu-baseline > $OUT/r02_c4_n${n}_bench.json 2>/dev/null
    prof_stats r02_c4_n$n --steps 40 --window $n --no-companions --no-cpu-baseline
  done
  FS_LSH_PREFILTER=0 python bench.py --steps 20 --window 8 --no-companions --no-cpu-baseline > $OUT/r02_c4_n8_noprefilter_bench.json 2>/dev/null
  echo "== N = 2 rehearsal (both ranks on this GPU, gloo): strong scaling path, gather verified"
  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 \
      bench.py --gpus 2 --steps 30 --warmup 5 --works 6000 --backend gloo > $OUT/r02_c3_gloo2_rehearsal.json 2>/dev/null
  echo "== LSH pipeline on the synonym-rich table"
  python tools/lsh_bench.py > $OUT/r02_lsh_clustered.json 2>/dev/null
  python tools/lsh_bench.py --window 8 --table synthetic --works 2000 > $OUT/r02_lsh_n8_synthetic.json 2>/dev/null
  echo "== streamed corpus (configs[4])"
  python tools/stream_bench.py > $OUT/r02_stream_c5.log 2>&1
  echo "== the reference's command end to end"
  python tools/cli_bench.py > $OUT/r02_cli_bench.json 2>$OUT/r02_cli_bench.err
  echo "== lanes / finish A/B in one process"
  python tools/step_bench.py --inflight 4 "FS_SCAN_ROWS=1" "FS_SCAN_ROWS=1 FS_LANES=2" "FS_SCAN_ROWS=1 FS_LANES=4" \
      "FS_SCAN_ROWS=1 FS_LANES=4 FS_ROWS_BLOCKS_PER_CU=1" "FS_SCAN_ROWS=1 FS_LANES=4 FS_SCAN_SUB=0" \
      "FS_SCAN_ROWS=0" "FS_SCAN_ROWS=0 FS_LANES=4" > $OUT/r02_step_ab.log 2>/dev/null
  echo "== LSH pipeline filters A/B (n = 8)"
  for e in FS_X=1 FS_LSH_WILD=0 FS_LSH_SELFLEV=0 FS_LSH_PREFILTER=0; do
    echo "$e $(env $e python tools/lsh_bench.py --window 8 --table synthetic --works 10000 2>/dev/null | tail -1)"
  done > $OUT/r02_lsh_n8_filters_ab.log
  ls -la $OUT
fi
