# usage: tools/prof_stats.sh <name> <python args...>   (rocprofv3 --kernel-trace --stats of one command; top kernels)
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/r04/prof_$name
mkdir -p $out && cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $out -o $name --output-format csv -- python3 "$@" > $out/run.log 2>&1; echo rc $?
cd $GRAFT_REPO_ROOT
f=$(find $out -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<P
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print(r["Name"][:58].ljust(58), r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"], round(float(r["MinNs"])/1e3,1))
P
