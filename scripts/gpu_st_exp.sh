#!/bin/bash
# experiment: kernel stats of bench_multistreet.py with library variants (pokerrl_amd/lib/libpokerrl_hip_<V>.so), + extra SQ counters
cd $GRAFT_REPO_ROOT; TAG=${1:-x}; shift; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_multistreet.py --steps 3 --warmup 1 --no-cpu-baseline"
for V in "$@"; do
  export POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip$V.so
  timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_p$V -o x -- $B > $R/gpurun_out/${TAG}_p$V.log 2>&1
  DB=$(find $R/gpurun_out/${TAG}_p$V -name "*.db" | head -1)
  { echo "# variant '$V'"; python $R/scripts/rocprof_summary.py $DB | head -9; } >> $R/gpurun_out/${TAG}_exp_stats.txt 2>&1
  rm -rf $R/gpurun_out/${TAG}_p$V
done
export POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip.so
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_INST_CYCLES_SMEM SQ_WAIT_INST_LDS -d $R/gpurun_out/${TAG}_c2 -o p2 --output-format csv -- $B > $R/gpurun_out/${TAG}_c2.log 2>&1
python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_c2 -name '*counter_collection.csv') | grep "st_down\|st_pass<true\|==" | cut -c1-600 > $R/gpurun_out/${TAG}_exp_pmc.txt 2>&1
tail -3 $R/gpurun_out/${TAG}_c2.log
rm -rf $R/gpurun_out/${TAG}_c2
cat $R/gpurun_out/${TAG}_exp_stats.txt | cut -c1-180; cat $R/gpurun_out/${TAG}_exp_pmc.txt
