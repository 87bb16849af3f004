#!/bin/bash
# Runs on the GPU box (gpurun): the measurements behind DESIGN.md section 8 for round 5, each into
# gpurun_out/r05/ under the name it keeps in profiles/.  Usage: tools/collect_profiles_r05.sh part1|part2|realistic_stats
# (tools/collect_profiles.sh is round 4's; the helpers are the same).  Every artifact carries the
# hash of the sources it was measured on (fandom_search_amd._lib.source_hash).
set -eu -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=r05
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$PWD
part=${1:-part1}
BUILD=$(python3 -c "from fandom_search_amd import _lib; print(_lib.source_hash())")
echo "$BUILD" > $OUT/${R}_build.txt

run() {          # name, program, args...
  local name=$1; shift
  python3 "$@" > $OUT/$name 2> $OUT/$name.err
  test -s $OUT/$name
}
stamp_csv() { echo "{\"build\": \"$BUILD\", \"file\": \"$1\"}" > $OUT/$1.build.json; }
prof_stats() {   # name, program args...: rocprofv3 kernel stats
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $ROOTDIR/$OUT/prof_$name -- \
      python3 "$@" > $ROOTDIR/$OUT/${name}_under_rocprof.json 2> $ROOTDIR/$OUT/${name}_under_rocprof.err )
  local f
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  test -s "$f"
  cp "$f" $OUT/${name}_kernel_stats.csv
  stamp_csv ${name}_kernel_stats.csv
  rm -rf $OUT/prof_$name
}
pmc() {          # name, counter, bench args...
  local name=$1 ctr=$2; shift 2
  ( cd /tmp && rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $ROOTDIR/$OUT/pmc_${name}_$ctr -- \
      python3 $ROOTDIR/bench.py "$@" > /dev/null 2> $ROOTDIR/$OUT/${name}_pmc_$ctr.err )
  local f
  f=$(find $OUT/pmc_${name}_$ctr -name "*counter_collection.csv" | head -1)
  test -s "$f"
  cp "$f" $OUT/${name}_pmc_$ctr.csv
  stamp_csv ${name}_pmc_$ctr.csv
  rm -rf $OUT/pmc_${name}_$ctr
}

B="--no-companions --no-cpu-baseline"
if [ "$part" = part1 ]; then
  echo "== the driver's form with every companion, and 200 steps"
  run ${R}_c2_bench_steps20.json bench.py --steps 20 --warmup 5
  run ${R}_c2_bench.json bench.py $B
  run ${R}_c2_bench_lanes1.json bench.py --lanes 1 --inflight 2 $B
  echo "== rocprofv3 kernel stats: the default command, one lane"
  prof_stats ${R}_c2 $ROOTDIR/bench.py $B
  prof_stats ${R}_c2_lanes1 $ROOTDIR/bench.py --lanes 1 --inflight 2 $B
  echo "== PMC: HBM traffic of k_scan_rows (separate passes), one lane and four"
  pmc ${R}_c2_lanes1 FETCH_SIZE --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  pmc ${R}_c2_lanes1 WRITE_SIZE --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  run ${R}_c2_bench_lanes1_steps8.json bench.py --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  python3 tools/pmc_to_traffic.py $OUT/${R}_c2_lanes1_pmc_FETCH_SIZE.csv $OUT/${R}_c2_lanes1_pmc_WRITE_SIZE.csv \
      $OUT/${R}_c2_bench_lanes1_steps8.json --also k_compact --out $OUT/${R}_scan_traffic_lanes1.json > /dev/null
  pmc ${R}_c2 FETCH_SIZE --steps 8 --warmup 2 --inflight 1 $B
  pmc ${R}_c2 WRITE_SIZE --steps 8 --warmup 2 --inflight 1 $B
  python3 tools/pmc_to_traffic.py $OUT/${R}_c2_pmc_FETCH_SIZE.csv $OUT/${R}_c2_pmc_WRITE_SIZE.csv \
      $OUT/${R}_c2_bench_steps20.json --out $OUT/scan_traffic.json > /dev/null
  echo "== c3shard (one GPU's share of configs[2], 250 MB of ids): bench, kernel stats, PMC traffic"
  run ${R}_c3shard_bench.json bench.py --workload c3shard --steps 100 $B
  prof_stats ${R}_c3shard $ROOTDIR/bench.py --workload c3shard --steps 40 $B
  pmc ${R}_c3shard FETCH_SIZE --workload c3shard --steps 8 --warmup 2 --inflight 1 $B
  pmc ${R}_c3shard WRITE_SIZE --workload c3shard --steps 8 --warmup 2 --inflight 1 $B
  run ${R}_c3shard_bench_steps8.json bench.py --workload c3shard --steps 8 --warmup 2 --inflight 1 $B
  python3 tools/pmc_to_traffic.py $OUT/${R}_c3shard_pmc_FETCH_SIZE.csv $OUT/${R}_c3shard_pmc_WRITE_SIZE.csv \
      $OUT/${R}_c3shard_bench_steps8.json --out $OUT/${R}_c3shard_scan_traffic.json > /dev/null
  ls -la $OUT
fi

if [ "$part" = realistic_stats ]; then
  prof_stats ${R}_realistic $ROOTDIR/tools/realistic_bench.py --works 2000 --no-counts
fi

if [ "$part" = part2 ]; then
  echo "== configs[3]: n = 4, 8, 10 -- bench line, kernel stats, the chain of round 4 beside the fused front end"
  for n in 4 8 10; do
    run ${R}_c4_n${n}_bench.json bench.py --steps 100 --window $n $B
    prof_stats ${R}_c4_n$n $ROOTDIR/bench.py --steps 40 --window $n $B
  done
  run ${R}_near8_ab.log tools/near_bench.py --window 8 "FS_NEAR_FUSED=1" "FS_NEAR_FUSED=0" "FS_LSH_DEFER_MIN=0"
  run ${R}_near10_ab.log tools/near_bench.py --window 10 "FS_NEAR_FUSED=1" "FS_NEAR_FUSED=0" "FS_LSH_EMAP=0" "FS_LSH_BATCH=0"
  echo "== the table with near-synonyms: variants in one process, kernel stats, both UniqueFilter settings"
  run ${R}_clustered_ab.log tools/near_bench.py --window 6 --table clustered "FS_NEAR_FUSED=1" "FS_NEAR_FUSED=0" "FS_LSH_EMAP=0" "FS_LSH_BATCH=0"
  run ${R}_clustered_unique_ab.log tools/near_bench.py --window 6 --table clustered --unique 1 "FS_NEAR_FUSED=1" "FS_LSH_EMAP=0"
  prof_stats ${R}_clustered $ROOTDIR/tools/near_bench.py --window 6 --table clustered --rounds 2 "FS_NEAR_FUSED=1"
  echo "== a table shaped like a real one (unnormalised, three scales, OOV names)"
  run ${R}_realistic.log tools/realistic_bench.py --works 2000
  FS_LSH_SHARE=3 run ${R}_realistic_gate_in_front_of_the_key_scan.log tools/realistic_bench.py --works 2000
  FS_LSH_SHARE=0 run ${R}_realistic_key_scan_only.log tools/realistic_bench.py --works 2000
  for d in 4 10 6 5 7; do FS_LSH_DIAG=$d run ${R}_realistic_share_scan_diag$d.log tools/realistic_bench.py --works 2000 --oov 0.0; done
  prof_stats ${R}_realistic $ROOTDIR/tools/realistic_bench.py --works 2000 --no-counts
  for n in 8 10; do
    run ${R}_realistic_n$n.log tools/realistic_bench.py --works 2000 --window $n
    FS_LSH_SHARE=0 run ${R}_realistic_n${n}_key_scan_only.log tools/realistic_bench.py --works 2000 --window $n
  done
  run ${R}_stress_share.log tools/stress_share.py --cases 60
  echo "== stress cross-checks"
  run ${R}_stress_lsh.log tools/stress_lsh.py --cases 48
  run ${R}_stress_rows.log tools/stress_rows.py
  echo "== N = 2 rehearsal (both ranks on this GPU, gloo)"
  run ${R}_gloo2_rehearsal.json bench.py --gpus 2 --steps 30 --warmup 5 --backend gloo $B
  echo "== the reference's command end to end: native text front end, and without it"
  run ${R}_cli_bench_20000.json tools/cli_bench.py --works 20000
  FANDOM_SEARCH_NATIVE_TEXT=0 run ${R}_cli_bench_20000_python_text.json tools/cli_bench.py --works 20000
  run ${R}_cli_bench_prose_20000.json tools/cli_bench.py --works 20000 --prose
  run ${R}_cli_bench_100000.json tools/cli_bench.py --works 100000
  ls -la $OUT
fi
