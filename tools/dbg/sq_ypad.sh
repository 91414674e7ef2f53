cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/prof_r03
mkdir -p $OUT
SQB="SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
export PCL_TUNE_FUSED_STEP=0
export PCL_LIB_OVERRIDE=$R/build/libs/libpyclaw_amd_ypad.so
rocprofv3 --pmc $SQB --kernel-trace --output-format csv -d $OUT/sq_2p_b_ypad -- python3 $R/bench.py --no-cpu-baseline --no-states --math exact --state dense --steps 3 --warmup 1 > $OUT/sq_2p_b_ypad.log 2>&1
tail -2 $OUT/sq_2p_b_ypad.log | cut -c1-200
