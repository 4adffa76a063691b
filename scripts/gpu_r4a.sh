#!/bin/bash
# round 4 checkpoint A: fused-engine GPU parity tests, headline bench, best-response bench, kernel trace of both
cd $GRAFT_REPO_ROOT; TAG=${1:-r4a}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or fhp or checkpoint or bench_size or br" -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -5 gpurun_out/${TAG}_pytest.log
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 1500 gpurun_out/${TAG}_bench.json; tail -3 gpurun_out/${TAG}_bench.err
timeout 600 python bench_br.py --no-cpu-baseline > gpurun_out/${TAG}_bench_br.json 2> gpurun_out/${TAG}_bench_br.err; tail -c 900 gpurun_out/${TAG}_bench_br.json; tail -3 gpurun_out/${TAG}_bench_br.err
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe > $R/gpurun_out/${TAG}_prof.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_prof -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe (262144 boards), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $DB; } > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
head -14 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-170
rm -rf $R/gpurun_out/${TAG}_prof
