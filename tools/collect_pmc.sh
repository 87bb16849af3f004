#!/bin/bash
# SQ counters of k_scan_rows on one c2 batch, one search at a time (runs on the GPU box).
#   tools/collect_pmc.sh OUTDIR [window]                     k_scan_rows (tools/pmc_run.py)
#   KERNEL=k_lsh_scan tools/collect_pmc.sh OUTDIR - tools/lsh_bench.py --works 5000 --reps 2
# Separate passes of at most eight SQ counters each (no trace domains beside --pmc).
set -eu -o pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=${1:-gpurun_out/pmc}; WIN=${2:-6}
KERNEL=${KERNEL:-k_scan_rows}
mkdir -p "$OUT"
export TMPDIR=/tmp
ROOTDIR=$PWD
if [ $# -gt 2 ]; then shift 2; PROG=("$ROOTDIR/$1"); shift; PROG+=("$@"); else PROG=("$ROOTDIR/tools/pmc_run.py" "$WIN" 6); fi
pass() {   # name, counters...
  local name=$1; shift
  ( cd /tmp && rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$ROOTDIR/$OUT/raw_$name" -- \
      python3 "${PROG[@]}" > "$ROOTDIR/$OUT/$name.log" 2> "$ROOTDIR/$OUT/$name.err" )
  local f
  f=$(find "$OUT/raw_$name" -name "*counter_collection.csv" | head -1)
  test -s "$f"
  cp "$f" "$OUT/pmc_$name.csv"
  rm -rf "$OUT/raw_$name"
}
pass a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA
pass b SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU
pass c SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_ACTIVE_INST_MISC
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, json, sys
out, kernel = sys.argv[1], sys.argv[2]
res = {}
for f in sorted(glob.glob(out + "/pmc_*.csv")):
    acc = {}
    for row in csv.DictReader(open(f, newline="")):
        if kernel in row["Kernel_Name"]:
            acc.setdefault(row["Counter_Name"], {}).setdefault(row["Dispatch_Id"], 0.0)
            acc[row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    for k, v in acc.items():
        vals = sorted(v.values())
        res[k] = vals[len(vals) // 2]          # median dispatch
res["kernel"] = kernel
print(json.dumps(res, indent=1))
json.dump(res, open(out + "/sq_counters.json", "w"), indent=1)
PY
