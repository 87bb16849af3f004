#!/bin/bash
# Runs on the GPU box (gpurun): the measurements behind DESIGN.md section 8, each into
# gpurun_out/r04/ under the name it keeps in profiles/.  Usage: tools/collect_profiles.sh part1|part2
# Every artifact carries the hash of the sources it was measured on (fandom_search_amd._lib
# .source_hash: git is not available on the box); a step that fails stops the script, its
# stderr stays next to the artifact (.err).
set -eu -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
R=r04
OUT=gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
ROOTDIR=$PWD
part=${1:-part1}
BUILD=$(python3 -c "from fandom_search_amd import _lib; print(_lib.source_hash())")
echo "$BUILD" > $OUT/${R}_build.txt

run() {          # name, program, args...: stdout -> $OUT/name, stderr -> $OUT/name.err, must not be empty
  local name=$1; shift
  python3 "$@" > $OUT/$name 2> $OUT/$name.err
  test -s $OUT/$name
}

stamp_csv() {    # prepend the build id as a comment line is not valid CSV for every reader: a sidecar instead
  echo "{\"build\": \"$BUILD\", \"file\": \"$1\"}" > $OUT/$1.build.json
}

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
  echo "== default bench line (N=1, c2, four distinct batches, 4 lanes), and the driver's form"
  run ${R}_c2_bench.json bench.py
  run ${R}_c2_bench_steps20.json bench.py --steps 20 --warmup 5 $B
  echo "== one lane (searches one after the other) and one resident batch"
  run ${R}_c2_bench_lanes1.json bench.py --lanes 1 --inflight 2 $B
  run ${R}_c2_bench_resident.json bench.py --rotate 1 $B
  echo "== rocprofv3 kernel stats of the default command and of one lane"
  prof_stats ${R}_c2 $ROOTDIR/bench.py $B
  prof_stats ${R}_c2_lanes1 $ROOTDIR/bench.py --lanes 1 --inflight 2 $B
  echo "== PMC: HBM traffic of k_scan_rows (separate passes), one lane = the kernel alone, and 4 lanes"
  pmc ${R}_c2_lanes1 FETCH_SIZE --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  pmc ${R}_c2_lanes1 WRITE_SIZE --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  run ${R}_c2_bench_lanes1_steps8.json bench.py --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  python3 tools/pmc_to_traffic.py $OUT/${R}_c2_lanes1_pmc_FETCH_SIZE.csv $OUT/${R}_c2_lanes1_pmc_WRITE_SIZE.csv \
      $OUT/${R}_c2_bench_lanes1_steps8.json --also k_compact --out $OUT/scan_traffic_lanes1.json > /dev/null
  pmc ${R}_c2 FETCH_SIZE --steps 8 --warmup 2 --inflight 1 $B
  pmc ${R}_c2 WRITE_SIZE --steps 8 --warmup 2 --inflight 1 $B
  python3 tools/pmc_to_traffic.py $OUT/${R}_c2_pmc_FETCH_SIZE.csv $OUT/${R}_c2_pmc_WRITE_SIZE.csv \
      $OUT/${R}_c2_bench_steps20.json --out $OUT/scan_traffic.json > /dev/null
  echo "== SQ counters and the in-kernel timeline of k_scan_rows (one search at a time)"
  tools/collect_pmc.sh $OUT/sq 6 > $OUT/${R}_c2_scan_rows_sq.log 2>&1
  cp $OUT/sq/sq_counters.json $OUT/${R}_c2_scan_rows_pmc_sq.json
  run ${R}_c2_scan_rows_timeline.json tools/scan_timeline.py
  echo "== c3shard: one GPU's share of configs[2] (12.5k works x 5k tokens, 250 MB)"
  run ${R}_c3shard_bench.json bench.py --workload c3shard --steps 100 $B
  echo "== larger batches in one launch: the N = 4 / N = 2 shares of configs[2] (500 MB / 1 GB of ids)"
  run ${R}_c3_quarter_bench.json bench.py --workload c3 --works 25000 --steps 40 --warmup 4 $B
  run ${R}_c3_half_bench.json bench.py --workload c3 --works 50000 --steps 40 --warmup 4 $B
  echo "== lanes A/B in one process"
  echo "== the same from HBM (four distinct batches): what round 4 added, switched off one at a time"
  run ${R}_step_ab.log tools/step_bench.py --rotate 4 --inflight 3 "FS_LANES=1" "FS_LANES=1 FS_DIAG=256" "FS_LANES=1 FS_DIAG=32" \
      "FS_LANES=1 FS_DIAG=128" "FS_LANES=1 FS_DIAG=16" "FS_LANES=2" "FS_LANES=4" "FS_SCAN_ROWS=0" "FS_SCAN_ROWS=0 FS_LANES=4"
  FS_LIB_FILE=libfandomsearch_hip_r03.so run ${R}_step_ab_r03lib.log tools/step_bench.py --rotate 4 --inflight 3 "FS_LANES=1" "FS_LANES=2" "FS_LANES=4"
  run ${R}_step_ab_resident.log tools/step_bench.py --inflight 4 "FS_LANES=1" "FS_LANES=1 FS_DIAG=256" "FS_LANES=2" "FS_LANES=4" "FS_LANES=4 FS_ROWS_BLOCKS_PER_CU=1"
  run ${R}_c2_scan_rows_timeline_phases.json tools/scan_timeline.py --extra FS_DIAG=6
  ls -la $OUT
fi

if [ "$part" = traffic ]; then
  echo "== (again, after a change of the sources) HBM traffic of k_scan_rows, one lane and four, and the kernel stats"
  run ${R}_c2_bench_steps20.json bench.py --steps 20 --warmup 5 $B
  run ${R}_c2_bench_lanes1.json bench.py --lanes 1 --inflight 2 $B
  prof_stats ${R}_c2 $ROOTDIR/bench.py $B
  prof_stats ${R}_c2_lanes1 $ROOTDIR/bench.py --lanes 1 --inflight 2 $B
  pmc ${R}_c2_lanes1 FETCH_SIZE --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  pmc ${R}_c2_lanes1 WRITE_SIZE --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  run ${R}_c2_bench_lanes1_steps8.json bench.py --steps 8 --warmup 2 --lanes 1 --inflight 1 $B
  python3 tools/pmc_to_traffic.py $OUT/${R}_c2_lanes1_pmc_FETCH_SIZE.csv $OUT/${R}_c2_lanes1_pmc_WRITE_SIZE.csv \
      $OUT/${R}_c2_bench_lanes1_steps8.json --also k_compact --out $OUT/scan_traffic_lanes1.json > /dev/null
  pmc ${R}_c2 FETCH_SIZE --steps 8 --warmup 2 --inflight 1 $B
  pmc ${R}_c2 WRITE_SIZE --steps 8 --warmup 2 --inflight 1 $B
  python3 tools/pmc_to_traffic.py $OUT/${R}_c2_pmc_FETCH_SIZE.csv $OUT/${R}_c2_pmc_WRITE_SIZE.csv \
      $OUT/${R}_c2_bench_steps20.json --out $OUT/scan_traffic.json > /dev/null
  tools/collect_pmc.sh $OUT/sq 6 > $OUT/${R}_c2_scan_rows_sq.log 2>&1
  cp $OUT/sq/sq_counters.json $OUT/${R}_c2_scan_rows_pmc_sq.json
  run ${R}_c2_scan_rows_timeline.json tools/scan_timeline.py
  ls -la $OUT
fi

if [ "$part" = part2 ]; then
  echo "== configs[3]: n = 4, 8, 10 on the 10k-work corpus"
  for n in 4 8 10; do
    run ${R}_c4_n${n}_bench.json bench.py --steps 100 --window $n $B
    prof_stats ${R}_c4_n$n $ROOTDIR/bench.py --steps 40 --window $n $B
  done
  echo "== the same without the per-n-gram records and the one-slot map (round 2's k_lsh_verify work)"
  run ${R}_c4_ab.log tools/step_bench.py --window 8 --steps 100 --inflight 4 "FS_LANES=4" "FS_LANES=4 FS_LSH_WMAP=0" "FS_LANES=4 FS_LSH_GRAMTAB=0 FS_LSH_WMAP=0"
  run ${R}_c4_n10_ab.log tools/step_bench.py --window 10 --steps 100 --inflight 4 "FS_LANES=4" "FS_LANES=4 FS_LSH_WMAP=0" "FS_LANES=4 FS_LSH_GRAMTAB=0 FS_LSH_WMAP=0"
  echo "== round 4, second half: the prefilter scan with four tokens per lane, kNB workgroups for k_lsh_sift / k_lsh_verify; one lane, kernel by kernel"
  run ${R}_c4_near_ab.log tools/step_bench.py --window 8 --steps 100 --inflight 4 "FS_LANES=4" "FS_LANES=4 FS_SCAN_NEAR8=0" "FS_LANES=1" "FS_LANES=1 FS_SCAN_NEAR8=0"
  run ${R}_c4_n10_near_ab.log tools/step_bench.py --window 10 --steps 100 --inflight 4 "FS_LANES=4" "FS_LANES=4 FS_SCAN_NEAR8=0" "FS_LANES=1" "FS_LANES=1 FS_SCAN_NEAR8=0"
  FS_LSH_FULL_GRID=1 run ${R}_c4_fullgrid.log tools/step_bench.py --window 8 --steps 100 --inflight 4 "FS_LANES=4" "FS_LANES=1"
  FS_LSH_FULL_GRID=1 run ${R}_c4_n10_fullgrid.log tools/step_bench.py --window 10 --steps 100 --inflight 4 "FS_LANES=4" "FS_LANES=1"
  prof_stats ${R}_c4_n8_lanes1 $ROOTDIR/tools/step_bench.py --window 8 --steps 60 --rounds 2 --inflight 1 "FS_LANES=1"
  prof_stats ${R}_c4_n10_lanes1 $ROOTDIR/tools/step_bench.py --window 10 --steps 60 --rounds 2 --inflight 1 "FS_LANES=1"
  echo "== N = 2 rehearsal: bench.py --gpus 2 typed as is (both ranks on this GPU, gloo)"
  run ${R}_gloo2_rehearsal.json bench.py --gpus 2 --steps 30 --warmup 5 --backend gloo $B
  echo "== LSH pipeline on the synonym-rich table: 5000 works, kernel stats, SQ counters"
  run ${R}_lsh_clustered.json tools/lsh_bench.py --works 5000 --reps 2
  FS_LSH_SYN=0 run ${R}_lsh_clustered_nosyn.json tools/lsh_bench.py --works 5000 --reps 2
  prof_stats ${R}_lsh_clustered $ROOTDIR/tools/lsh_bench.py --works 5000 --reps 4
  FS_LSH_SYN=0 KERNEL=k_lsh_scan tools/collect_pmc.sh $OUT/sq_lsh - tools/lsh_bench.py --works 5000 --reps 2 > $OUT/${R}_lsh_sq.log 2>&1
  cp $OUT/sq_lsh/sq_counters.json $OUT/${R}_lsh_scan_pmc_sq.json
  KERNEL=k_lsh_sift tools/collect_pmc.sh $OUT/sq_sift - tools/lsh_bench.py --works 5000 --reps 2 > $OUT/${R}_lsh_sift_sq.log 2>&1
  cp $OUT/sq_sift/sq_counters.json $OUT/${R}_lsh_sift_pmc_sq.json
  echo "== mixed-case companion by itself, kernel stats"
  run ${R}_tokstr.json tools/tokstr_bench.py 30
  FS_LANES=1 run ${R}_tokstr_lanes1.json tools/tokstr_bench.py 30
  FS_STR_FUSED=0 run ${R}_tokstr_chain.json tools/tokstr_bench.py 30
  FS_STR_FAST=0 run ${R}_tokstr_wave_per_pair.json tools/tokstr_bench.py 30
  prof_stats ${R}_tokstr $ROOTDIR/tools/tokstr_bench.py 30
  FS_STR_FUSED=0 prof_stats ${R}_tokstr_chain $ROOTDIR/tools/tokstr_bench.py 30
  echo "== host rows: wall time of the synchronous call against the GPU time inside it"
  run ${R}_host_rows.log tools/host_rows_time.py
  FS_HOST_ZEROCOPY=0 run ${R}_host_rows_copy.log tools/host_rows_time.py
  echo "== streamed corpus (configs[4])"
  run ${R}_stream_c5.log tools/stream_bench.py
  echo "== the reference's command end to end"
  run ${R}_cli_bench.json tools/cli_bench.py
  run ${R}_cli_bench_20000.json tools/cli_bench.py --works 20000
  echo "== ... on prose (capitals, punctuation, contractions: the tokenizer's rule path, OOV tokens, LSH pipeline)"
  run ${R}_cli_bench_prose_20000.json tools/cli_bench.py --works 20000 --prose
  ls -la $OUT
fi
