# kernel trace of one side object of the bench: gpurun_out/$2/kt_$1.md   (usage: trace_bench_only.sh ucc_colbert r4/kt)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${2:-r4/kt}
mkdir -p $O
rocprofv3 --kernel-trace -d $O/$1 -o kt --output-format csv -- python3 $R/bench.py --only $1 --steps 5 > $O/$1.json 2> $O/$1.err
python3 $R/scripts/summarize_rocprof.py $(find $O/$1 -name "*kernel_trace.csv" | head -1) > $O/kt_$1.md
head -16 $O/kt_$1.md
