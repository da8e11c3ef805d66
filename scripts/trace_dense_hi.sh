# kernel trace of one large-scan search (fp16 first pass, 64 queries) on the synthetic 10 M x 768 matrix
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/hi_trace
mkdir -p $O
export AB_CASES="${AB_CASES:-64:10}" AB_ONLY_HI=1 AB_STEPS=5
rocprofv3 --kernel-trace --stats -d $O/t --output-format csv -- python3 $R/scripts/ab_dense_hi.py > $O/t.log 2>&1
cd $R
f=$(find $O/t -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(r["Name"][:70], r["Calls"], r["AverageNs"], r["TotalDurationNs"])
P
tail -2 $O/t.log | cut -c1-400
