# Evidence run for profiles/ (GPU box): the driver's bench command un-profiled, the same under rocprofv3 --kernel-trace --stats,
# and the 2-rank rehearsal of the N > 1 line (both ranks on the one card, gloo in place of RCCL).  Output: gpurun_out/$1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r3final}
mkdir -p $O
cd $R
echo "bench un-profiled" >> $O/progress.log
timeout -k 10 500 python3 bench.py > $O/bench_unprofiled.json 2> $O/bench_unprofiled.err; echo "rc=$?" >> $O/progress.log
echo "bench under rocprofv3" >> $O/progress.log
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_profiled.json 2> $O/bench_profiled.err); echo "rc=$?" >> $O/progress.log
python3 scripts/summarize_rocprof.py $(find $O/kt -name "*kernel_trace.csv" | head -1) > $O/kernel_trace.md 2>> $O/progress.log
echo "2-rank rehearsal" >> $O/progress.log
BENCH_FORCE_DEVICE=0 BENCH_DIST_BACKEND=gloo timeout -k 10 500 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 5 --warmup 2 --synth-rows 2000000 > $O/bench_2rank.out 2> $O/bench_2rank.err; echo "rc=$?" >> $O/progress.log
grep "^{" $O/bench_2rank.out > $O/bench_2rank.json
cat $O/progress.log
tail -c 700 $O/bench_unprofiled.json
