#!/bin/bash
# checkpoint (gpurun -- bash scripts/gpu_r05m.sh TAG): multi-street bench at two sizes + kernel trace, headline bench
cd $GRAFT_REPO_ROOT; TAG=${1:-r05m}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
bash scripts/gpu_ms_bench.sh $TAG
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("headline", d["value"] / 1e6, "M", d["ms_per_step"], "ms frac", d["roofline"]["frac"], d["config"].get("placement_probe_ms_per_iteration"))
PY
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_multistreet.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_ms_prof -o ${TAG} -- $B > $R/gpurun_out/${TAG}_ms_prof.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_ms_prof -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_multistreet.py --steps 4 --warmup 1 --no-cpu-baseline (LimitHoldem 4x2x2 run-outs, 259330 nodes), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $DB; } > $R/gpurun_out/${TAG}_multistreet_kernel_stats.txt 2>&1
head -40 $R/gpurun_out/${TAG}_multistreet_kernel_stats.txt | cut -c1-160
rm -rf $R/gpurun_out/${TAG}_ms_prof
