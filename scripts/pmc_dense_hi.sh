# HBM requests + SQ counters of the fp16 first pass (dense_hi_tilemax_kernel) on 10 M x 768, 64 queries per scan
# (rocprofv3, separate --pmc passes, kernel trace only; every pass under its own timeout)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r3/hi_pmc
mkdir -p $O
B=${B:-64}
run() { # name, counters...
  n=$1; shift
  echo "pass $n" >> $O/progress.log
  timeout -k 10 150 rocprofv3 --kernel-trace --pmc "$@" -d $O/$n --output-format csv -- python3 $R/scripts/run_dense_once.py 10000000 $B 768 3 > $O/$n.log 2>&1
  echo "pass $n rc=$?" >> $O/progress.log
}
[ -z "$SKIP_EA" ] && run ea_$B TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_sum
[ -z "$SKIP_SQ" ] && run sq1_$B SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
[ -z "$SKIP_SQ" ] && run sq2_$B SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
cd $R
for d in ea_$B sq1_$B sq2_$B; do f=$(find $O/$d -name "*counter_collection.csv" 2>/dev/null | head -1); echo "== $d"; [ -n "$f" ] && python3 scripts/summarize_rocprof.py --pmc $f | grep -i "hi_tilemax\|transpose\|kernel |\|---" ; done
cat $O/progress.log
