#!/bin/bash
# SQ counters of a bench's kernels (gpurun -- bash scripts/gpu_pmc_sq.sh TAG NAME <bench command>): two rocprofv3 --pmc passes with --kernel-trace only,
# mean per dispatch and kernel name -> gpurun_out/TAG_NAME_pmc_sq.txt. SQ_WAVE_CYCLES / SQ_BUSY_CYCLES count quad-cycles per wave / per SE.
cd $GRAFT_REPO_ROOT; TAG=$1; NAME=$2; shift 2; CMD="$*"; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 900 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $R/gpurun_out/${TAG}_${NAME}_sq1 -o p --output-format csv -- $CMD > $R/gpurun_out/${TAG}_${NAME}_sq1.log 2>&1 )
( cd $R && timeout 900 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/${TAG}_${NAME}_sq2 -o p --output-format csv -- $CMD > $R/gpurun_out/${TAG}_${NAME}_sq2.log 2>&1 )
{ echo "# rocprofv3 --kernel-trace --pmc <SQ group 1 | SQ group 2> (one group per run) -- $CMD ; mean per dispatch; MI355X, checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_${NAME}_sq1 $R/gpurun_out/${TAG}_${NAME}_sq2 -name '*counter_collection.csv') | grep "pass<\|==" | cut -c1-600; } > $R/gpurun_out/${TAG}_${NAME}_pmc_sq.txt 2>&1
grep "pass<true, [45]\|pass<2, 4, 4" $R/gpurun_out/${TAG}_${NAME}_pmc_sq.txt | cut -c1-500
rm -rf $R/gpurun_out/${TAG}_${NAME}_sq1 $R/gpurun_out/${TAG}_${NAME}_sq2
