#!/bin/bash
# checkpoint r05j (gpurun -- bash scripts/gpu_r05j.sh TAG): streets parity + bench after the spill / instance-table fixes, the 262144-board oracle
# fixture test, kernel trace of the multi-street bench
cd $GRAFT_REPO_ROOT; TAG=${1:-r05j}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streets or bench_size_vs_oracle_fixture" > gpurun_out/${TAG}_tests.txt 2>&1; tail -4 gpurun_out/${TAG}_tests.txt
bash scripts/gpu_ms_bench.sh $TAG
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_multistreet.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_ms_prof -o ${TAG} -- $B > $R/gpurun_out/${TAG}_ms_prof.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_ms_prof -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_multistreet.py --steps 4 --warmup 1 --no-cpu-baseline (LimitHoldem 4x2x2 run-outs, 259330 nodes), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $DB; } > $R/gpurun_out/${TAG}_multistreet_kernel_stats.txt 2>&1
head -16 $R/gpurun_out/${TAG}_multistreet_kernel_stats.txt | cut -c1-160
rm -rf $R/gpurun_out/${TAG}_ms_prof
