# usage: tools/fetch_probe.sh NAME [ENV=VAL ...]   FETCH_SIZE / WRITE_SIZE per k_scan_rows launch (tools/pmc_run.py, one lane)
name=$1; shift
cd "${GRAFT_REPO_ROOT:-/root/repo}"; ROOTDIR=$PWD; OUT=gpurun_out/r04/fetch_$name; mkdir -p $OUT; export TMPDIR=/tmp
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && env "$@" rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $ROOTDIR/$OUT/$ctr -- python3 $ROOTDIR/tools/pmc_run.py 6 8 > $ROOTDIR/$OUT/$ctr.log 2>&1 )
done
python3 - "$OUT" "$name" <<'P'
import csv, glob, sys
out, name = sys.argv[1], sys.argv[2]
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(out + "/" + ctr + "/**/*counter_collection.csv", recursive=True)[0]
    acc = {}
    for r in csv.DictReader(open(f)):
        if "k_scan_rows" in r["Kernel_Name"] and r["Counter_Name"] == ctr:
            acc[r["Dispatch_Id"]] = acc.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    v = sorted(acc.values()); res[ctr] = v[len(v)//2]
print(name, "FETCH_KB", round(res["FETCH_SIZE"]), "WRITE_KB", round(res["WRITE_SIZE"]), "hbm_MB(convention)", round((2*res["FETCH_SIZE"]+res["WRITE_SIZE"])*1024/1e6, 1))
P
