#!/bin/bash
# LBR (config 5): bench at 2^20 hands per seat, rocprofv3 kernel trace + the two SQ counter groups of a shorter run; the two longer fixtures' GPU tests
cd $GRAFT_REPO_ROOT; TAG=${1:-r06}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_lbr.py -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_lbr_pytest.txt 2>&1; tail -2 gpurun_out/${TAG}_lbr_pytest.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "d2_i6 or placement" -p no:cacheprovider > gpurun_out/${TAG}_fixtures_pytest.txt 2>&1; tail -2 gpurun_out/${TAG}_fixtures_pytest.txt
timeout 900 python bench_lbr.py --hands 1048576 > gpurun_out/${TAG}_bench_lbr.json 2> gpurun_out/${TAG}_bench_lbr.err; tail -c 1800 gpurun_out/${TAG}_bench_lbr.json; tail -2 gpurun_out/${TAG}_bench_lbr.err
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_lbr.py --hands 131072 --cpu-hands 0"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_lbrprof -o l -- $B > $R/gpurun_out/${TAG}_lbrprof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_lbr.py --hands 131072 --cpu-hands 0, MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_lbrprof -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_lbr_kernel_stats.txt 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
timeout 600 rocprofv3 --kernel-trace --pmc $SQ1 -d $R/gpurun_out/${TAG}_lbrpmc1 -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_lbrpmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc $SQ2 -d $R/gpurun_out/${TAG}_lbrpmc2 -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_lbrpmc2.log 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc <one SQ group per run> -- python bench_lbr.py --hands 131072 --cpu-hands 0; mean per dispatch; MI355X, checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_lbrpmc1 $R/gpurun_out/${TAG}_lbrpmc2 -name '*counter_collection.csv') | grep "lbr_batch\|==" | cut -c1-500; } > $R/gpurun_out/${TAG}_lbr_pmc_sq.txt 2>&1
head -5 $R/gpurun_out/${TAG}_lbr_kernel_stats.txt | cut -c1-150; cat $R/gpurun_out/${TAG}_lbr_pmc_sq.txt | cut -c1-400
rm -rf $R/gpurun_out/${TAG}_lbrprof $R/gpurun_out/${TAG}_lbrpmc1 $R/gpurun_out/${TAG}_lbrpmc2
