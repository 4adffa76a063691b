#!/bin/bash
# round 4: the small-tree kernel with its state pointers visibly in LDS (bench_leduc.py, Leduc parity tests); LBR kernel SQ counters + kernel trace
cd $GRAFT_REPO_ROOT; TAG=${1:-r15}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "leduc or Leduc or small" > gpurun_out/${TAG}_leduc_pytest.txt 2>&1; tail -2 gpurun_out/${TAG}_leduc_pytest.txt
for g in StandardLeduc DiscretizedNLLeduc BigLeduc; do timeout 300 python bench_leduc.py --game $g > gpurun_out/${TAG}_bench_leduc_$g.json 2> gpurun_out/${TAG}_bench_leduc.err; python -c "
import json;d=json.loads(open('gpurun_out/${TAG}_bench_leduc_$g.json').read().strip().splitlines()[-1]);print('$g', d['value'], d['ms_per_step'], d['many_solves_one_launch']['node_updates_per_s'], d['cpu_baseline']['value'])"; done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_lbr.py --hands 131072 --cpu-hands 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_lbrprof -o l -- $B > $R/gpurun_out/${TAG}_lbrprof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_lbr.py --hands 131072 --cpu-hands 0, MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_lbrprof -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_lbr_kernel_stats.txt 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
i=0
for grp in "$SQ1" "$SQ2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_lbrpmc$i -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_lbrpmc$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <one SQ group per run> -- python bench_lbr.py --hands 131072 --cpu-hands 0; mean per dispatch; MI355X, checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_lbrpmc1 $R/gpurun_out/${TAG}_lbrpmc2 -name '*counter_collection.csv') | grep "lbr_batch\|==" | cut -c1-600; } > $R/gpurun_out/${TAG}_lbr_pmc_sq.txt 2>&1
head -4 $R/gpurun_out/${TAG}_lbr_kernel_stats.txt | cut -c1-160; grep lbr_batch $R/gpurun_out/${TAG}_lbr_pmc_sq.txt | cut -c60-500
rm -rf $R/gpurun_out/${TAG}_lbrprof $R/gpurun_out/${TAG}_lbrpmc?
