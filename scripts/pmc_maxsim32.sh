cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export AB_VARIANTS="1:4:32:0"  # one-pass form
O=$R/gpurun_out/r3/ms_pmc32
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/p1 --output-format csv -- python3 $R/scripts/ab_maxsim.py en > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_VALU_MFMA_COEXEC_CYCLES -d $O/p2 --output-format csv -- python3 $R/scripts/ab_maxsim.py en > $O/p2.log 2>&1
export AMDR_MAXSIM_ABL=3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/p3 --output-format csv -- python3 $R/scripts/ab_maxsim.py en > $O/p3.log 2>&1
cd $R
for d in p1 p2 p3; do echo "== $d"; f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 scripts/summarize_rocprof.py --pmc $f | grep -i "ring32" | grep -v "| 58368 |\|| 3072" ; f=$(find $O/$d -name "*kernel_trace.csv" | head -1); python3 scripts/summarize_rocprof.py $f | grep ring32; done
