# rocprofv3 passes of the default bench (GPU box): kernel trace + stats, then FETCH_SIZE / WRITE_SIZE in separate --pmc passes
# (kernel trace only beside them).  Output under gpurun_out/$1; summaries are copied into profiles/ by hand.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3prof}
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/fetch -o f --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --windows 1 --no-cpu-baseline --no-hbm-scan > $O/fetch.json 2> $O/fetch.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/write -o w --output-format csv -- python3 $R/bench.py --steps 5 --warmup 1 --windows 1 --no-cpu-baseline --no-hbm-scan > $O/write.json 2> $O/write.err
cd $R
python3 scripts/summarize_rocprof.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/kernel_trace.md
python3 scripts/summarize_rocprof.py --pmc $(find $O/fetch -name "*counter_collection.csv" | head -1) > $O/fetch.md
python3 scripts/summarize_rocprof.py --pmc $(find $O/write -name "*counter_collection.csv" | head -1) > $O/write.md
python3 bench.py --steps 20 --warmup 5 > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err
tail -c 600 $O/bench_unprofiled.json
