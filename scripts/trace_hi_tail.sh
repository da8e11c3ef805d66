# kernel trace of one large dense search per case (what follows the scan): gpurun_out/$1/{kt_*.md, timeline_*.md}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r4/tail}
mkdir -p $O
for tail in 1 0; do
  AMDR_DENSE_HI_TAIL=$tail AB_ONLY=1 AB_CASES=${AB_CASES:-64:10} AB_STEPS=6 rocprofv3 --kernel-trace -d $O/kt$tail -o kt --output-format csv -- python3 $R/scripts/ab_hi_tail.py > $O/run$tail.json 2> $O/run$tail.err
  f=$(find $O/kt$tail -name "*kernel_trace.csv" | head -1)
  python3 $R/scripts/summarize_rocprof.py $f > $O/kt_tail$tail.md
  python3 $R/scripts/summarize_rocprof.py --timeline ${TL:-24} $f > $O/timeline_tail$tail.md
done
cat $O/timeline_tail1.md
