# SQ counters of the MaxSim launch (rocprofv3, separate --pmc passes, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export AB_VARIANTS="${AB_VARIANTS:-1:4:32:0}"  # one-pass form (the counters of profiles/r03_pmc.md §2)
O=$R/gpurun_out/r3/ms_pmc
mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/p1 --output-format csv -- python3 $R/scripts/ab_maxsim.py en > $O/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $O/p2 --output-format csv -- python3 $R/scripts/ab_maxsim.py en > $O/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS -d $O/p3 --output-format csv -- python3 $R/scripts/ab_maxsim.py en > $O/p3.log 2>&1
cd $R
for d in p1 p2 p3; do f=$(find $O/$d -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 scripts/summarize_rocprof.py --pmc $f | grep -i "ring" | grep -v "| 58368 |\|| 3072" ; done
tail -3 $O/p3.log
