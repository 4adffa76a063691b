#!/bin/bash
# round 6: kernel-trace summary of bench_multistreet.py's default tree (LimitHoldem 4 x 2 x 2).  gpurun -- bash scripts/gpu_r6l.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r94}; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
args="--steps 40 --warmup 2 --no-cpu-baseline --placement-candidates 1"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_ms -o ms -- python $R/bench_multistreet.py $args > $R/gpurun_out/${TAG}_prof_ms.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_prof_ms -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_multistreet.py $args, MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $DB; } > $R/gpurun_out/${TAG}_multistreet_kernel_stats.txt 2>&1
head -30 $R/gpurun_out/${TAG}_multistreet_kernel_stats.txt | cut -c1-200
rm -rf $R/gpurun_out/${TAG}_prof_ms
