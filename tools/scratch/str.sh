set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/str
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_fuzz.py tests/test_gpu_general.py -x -q -m gpu > gpurun_out/str/tests.log 2>&1 || { tail -30 gpurun_out/str/tests.log; exit 1; }
tail -2 gpurun_out/str/tests.log
python3 tools/tokstr_bench.py 30 2>/dev/null
FS_STR_FAST=0 python3 tools/tokstr_bench.py 30 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/str/p -- python3 $GRAFT_REPO_ROOT/tools/tokstr_bench.py 30 > /dev/null 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/str/p <<'PY'
import csv,glob,statistics,re,sys
f=glob.glob(sys.argv[1]+"/*/*kernel_trace.csv")[0]
d={}
for r in csv.DictReader(open(f)):
    m=re.search(r"(k_\w+)", r["Kernel_Name"])
    k=m.group(1) if m else r["Kernel_Name"][:30]
    d.setdefault(k,[]).append(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))
for k,v in sorted(d.items(), key=lambda kv:-sum(kv[1]))[:9]:
    print("  %-22s calls %4d median %9.1f min %8.1f max %10.1f"%(k,len(v),statistics.median(v)/1e3,min(v)/1e3,max(v)/1e3))
PY
