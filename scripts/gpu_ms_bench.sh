#!/bin/bash
# quick: bench_multistreet.py at two sizes (gpurun -- scripts/gpu_ms_bench.sh TAG)
cd $GRAFT_REPO_ROOT; TAG=${1:-q}; mkdir -p gpurun_out
timeout 600 python bench_multistreet.py --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet.json 2> gpurun_out/${TAG}_bench_multistreet.err
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_multistreet.json",):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(d["value"] / 1e6, "M node-updates/s", d["ms_per_step"], "ms; last-street kernel ms/iter", d["roofline"]["kernel_ms_per_iteration"], "frac", d["roofline"]["frac"], d["roofline"]["frac_whole_iteration"])
PY
timeout 600 python bench_multistreet.py --flops 16 --turns 3 --rivers 3 --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_16x3x3.json 2>> gpurun_out/${TAG}_bench_multistreet.err
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_multistreet_16x3x3.json",):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(d["value"] / 1e6, "M node-updates/s", d["ms_per_step"], "ms; last-street kernel ms/iter", d["roofline"]["kernel_ms_per_iteration"], "frac", d["roofline"]["frac"], d["roofline"]["frac_whole_iteration"])
PY
