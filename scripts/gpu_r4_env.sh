#!/bin/bash
# round 4: batched env after the coalesced observation stream: GPU parity tests of the env, bench_env.py, kernel trace, PMC traffic + SQ counters of
# prl_k_ebf_random_step (one counter group per run, --kernel-trace only); optionally the two-process bench-size test (VMM=1)
cd $GRAFT_REPO_ROOT; TAG=${1:-r07}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_envbatch.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/${TAG}_env_pytest.txt
timeout 300 python bench_env.py > gpurun_out/${TAG}_bench_env.json 2> gpurun_out/${TAG}_bench_env.err; tail -c 900 gpurun_out/${TAG}_bench_env.json; tail -2 gpurun_out/${TAG}_bench_env.err
if [ -n "$VMM" ]; then timeout 1200 python -m pytest tests/test_sharded.py -m gpu -x -q -k "bench_size_on_shuffled" 2>&1 | tail -15 | tee gpurun_out/${TAG}_vmm_pytest.txt; fi
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_env.py --steps 40 --warmup 5 --no-cpu-baseline"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_envprof -o e -- $B > $R/gpurun_out/${TAG}_envprof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_env.py --steps 40 --warmup 5 --no-cpu-baseline (2^20 envs), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_envprof -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_env_kernel_stats.txt 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_envpmc$i -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_envpmc$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <one counter group per run> -- python bench_env.py --steps 40 --warmup 5 --no-cpu-baseline (2^20 envs); mean per dispatch; checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_envpmc1 $R/gpurun_out/${TAG}_envpmc2 $R/gpurun_out/${TAG}_envpmc3 $R/gpurun_out/${TAG}_envpmc4 -name '*counter_collection.csv') | grep "prl_k_eb\|==" | cut -c1-700; } > $R/gpurun_out/${TAG}_env_pmc.txt 2>&1
head -6 $R/gpurun_out/${TAG}_env_kernel_stats.txt | cut -c1-160; grep "random_step" $R/gpurun_out/${TAG}_env_pmc.txt | cut -c1-400
rm -rf $R/gpurun_out/${TAG}_envprof $R/gpurun_out/${TAG}_envpmc?
