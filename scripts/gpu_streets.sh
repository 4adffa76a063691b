#!/bin/bash
# Per-street engine checkpoint on the GPU box (gpurun -- scripts/gpu_streets.sh TAG): its GPU tests, bench_multistreet.py on both engines,
# rocprofv3 kernel trace of the bench command; summaries under gpurun_out/ (copy into profiles/).
cd $GRAFT_REPO_ROOT; TAG=${1:-r05}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -m gpu -x -q -k "streets" > gpurun_out/${TAG}_streets_tests.txt 2>&1; tail -5 gpurun_out/${TAG}_streets_tests.txt
timeout 600 python bench_multistreet.py > gpurun_out/${TAG}_bench_multistreet.json 2> gpurun_out/${TAG}_bench_multistreet.err; tail -c 1500 gpurun_out/${TAG}_bench_multistreet.json; tail -3 gpurun_out/${TAG}_bench_multistreet.err
timeout 600 python bench_multistreet.py --flops 16 --turns 3 --rivers 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_16x3x3.json 2>> gpurun_out/${TAG}_bench_multistreet.err; tail -c 1200 gpurun_out/${TAG}_bench_multistreet_16x3x3.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_multistreet.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_ms_prof -o ${TAG} -- $B > $R/gpurun_out/${TAG}_ms_prof.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_ms_prof -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_multistreet.py --steps 4 --warmup 1 --no-cpu-baseline (LimitHoldem 4x2x2 run-outs, 259330 nodes), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $DB; } > $R/gpurun_out/${TAG}_multistreet_kernel_stats.txt 2>&1
head -14 $R/gpurun_out/${TAG}_multistreet_kernel_stats.txt | cut -c1-200
rm -rf $R/gpurun_out/${TAG}_ms_prof
