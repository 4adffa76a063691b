#!/bin/bash
# round 6: where a mixed-streets iteration's time goes -- rocprofv3 kernel-trace summary of bench_multistreet.py on DiscretizedNLHoldem.  gpurun -- bash scripts/gpu_r6e.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r82}; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
args="--game DiscretizedNLHoldem --flops 16 --turns 8 --rivers 8 --steps 40 --warmup 2 --no-cpu-baseline --placement-candidates 1"
timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_nl -o nl -- python $R/bench_multistreet.py $args > $R/gpurun_out/${TAG}_prof_nl.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_prof_nl -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_multistreet.py $args, MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $DB; } > $R/gpurun_out/${TAG}_multistreet_nl_kernel_stats.txt 2>&1
head -40 $R/gpurun_out/${TAG}_multistreet_nl_kernel_stats.txt | cut -c1-200
rm -rf $R/gpurun_out/${TAG}_prof_nl
