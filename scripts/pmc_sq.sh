# SQ counters of one side object of the bench (two --pmc passes, kernel trace only beside them):
#   bash scripts/pmc_sq.sh ucc_colbert r4/sq maxsim      -> gpurun_out/r4/sq/{sq1,sq2}.md filtered by kernel substring
#   bash scripts/pmc_sq.sh headline r4/sq_bm25 bm25       (the default bench step, --no-extras)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OBJ=${1:-ucc_colbert}
O=$R/gpurun_out/${2:-r4/sq}
FILT=${3:-amdr::}
mkdir -p $O
run() { # name, counters...
  n=$1; shift
  if [ "$OBJ" = headline ]; then ARGS="--steps 3 --warmup 1 --windows 1 --no-cpu-baseline --no-extras"; else ARGS="--only $OBJ --steps 3"; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n --output-format csv -- python3 $R/bench.py $ARGS > $O/$n.log 2>&1
  echo "pass $n rc=$?" >> $O/progress.log
  f=$(find $O/$n -name "*counter_collection.csv" 2>/dev/null | head -1)
  [ -n "$f" ] && python3 $R/scripts/summarize_rocprof.py --pmc --only "$FILT" $f > $O/$n.md
}
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE
run sq2 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES
if [ -n "$EXTRA" ]; then  # latency / back-pressure counters (LEVEL / INSTS = mean cycles in flight per instruction)
run sq3 SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA
run sq4 SQ_VALU_MFMA_COEXEC_CYCLES SQ_LEVEL_WAVES SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT
run sq5 SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_IFETCH_LEVEL SQ_IFETCH SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM
fi
cat $O/progress.log
grep -h "$FILT\|kernel |" $O/sq*.md | head -${LINES:-40}
