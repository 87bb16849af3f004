set -e
cd /tmp && export TMPDIR=/tmp
export FS_LSH_WMAP=2
for n in 8 10; do
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/n810/q$n -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --window $n --lanes 1 --inflight 1 --no-companions --no-cpu-baseline > /dev/null 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/n810/q$n <<'PY'
import csv,glob,statistics,re,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
d={}
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_\w+)", r["Kernel_Name"])
    k=m.group(1) if m else r["Kernel_Name"][:30]
    d.setdefault(k,[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:7]:
    print("  %-22s calls %4d median %9.1f min %8.1f max %10.1f"%(k,len(v),statistics.median(v)/1e3,min(v)/1e3,max(v)/1e3))
PY
done
