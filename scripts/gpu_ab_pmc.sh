# SQ counters of two library variants side by side (16384 boards): bash scripts/gpu_ab_pmc.sh variant...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  lib=$R/pokerrl_amd/lib/libpokerrl_hip_$v.so; [ "$v" = product ] && lib=$R/pokerrl_amd/lib/libpokerrl_hip.so
  B="python $R/bench.py --boards 16384 --steps 4 --warmup 1 --no-cpu-baseline"
  POKERRL_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $R/gpurun_out/abpmc1_$v -o p1 --output-format csv -- $B > $R/gpurun_out/abpmc1_$v.log 2>&1
  POKERRL_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $R/gpurun_out/abpmc2_$v -o p2 --output-format csv -- $B > $R/gpurun_out/abpmc2_$v.log 2>&1
  POKERRL_AMD_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INSTS_CBRANCH_TAKEN SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_LEVEL_LDS SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/abpmc3_$v -o p3 --output-format csv -- $B > $R/gpurun_out/abpmc3_$v.log 2>&1
  echo "=== $v"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/abpmc1_$v $R/gpurun_out/abpmc2_$v $R/gpurun_out/abpmc3_$v -name '*counter_collection.csv') 2>&1 | grep "fhp_pass\|==" | cut -c1-600
done
