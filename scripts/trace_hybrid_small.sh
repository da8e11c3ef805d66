# kernel trace of the serving call, separate launches vs one launch: gpurun_out/r4/hs_kt.md
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4/hs_kt
mkdir -p $O
export AB_SHAPES=${AB_SHAPES:-591:384,1260:768}
rocprofv3 --kernel-trace -d $O/run -o kt --output-format csv -- python3 $R/scripts/ab_hybrid_small.py > $O/ab.json 2> $O/ab.err
python3 $R/scripts/summarize_rocprof.py $(find $O/run -name "*kernel_trace.csv" | head -1) > $R/gpurun_out/r4/hs_kt.md
head -30 $R/gpurun_out/r4/hs_kt.md
