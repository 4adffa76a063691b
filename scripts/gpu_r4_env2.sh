#!/bin/bash
# round 4: batched env without scratch (seat-indexed state through selects) -- env parity tests, bench_env, PMC traffic; LBR phase clocks (timing variant)
cd $GRAFT_REPO_ROOT; TAG=${1:-r08}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_envbatch.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/${TAG}_env_pytest.txt
timeout 300 python bench_env.py > gpurun_out/${TAG}_bench_env.json 2> gpurun_out/${TAG}_bench_env.err; tail -c 1200 gpurun_out/${TAG}_bench_env.json; tail -2 gpurun_out/${TAG}_bench_env.err
POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip_lbrtiming.so timeout 600 python bench_lbr.py --hands 262144 --cpu-hands 0 > gpurun_out/${TAG}_lbr_timing.json 2> gpurun_out/${TAG}_lbr_phases.txt; grep "lbrb phase" gpurun_out/${TAG}_lbr_phases.txt | tail -12
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_env.py --steps 40 --warmup 5 --no-cpu-baseline"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_envpmc$i -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_envpmc$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <one counter per run> -- python bench_env.py --steps 40 --warmup 5 --no-cpu-baseline (2^20 envs); mean per dispatch; checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_envpmc1 $R/gpurun_out/${TAG}_envpmc2 -name '*counter_collection.csv') | grep "prl_k_eb\|==" | cut -c1-300; } > $R/gpurun_out/${TAG}_env_pmc.txt 2>&1
grep "random_step" $R/gpurun_out/${TAG}_env_pmc.txt | cut -c1-300
rm -rf $R/gpurun_out/${TAG}_envpmc?
