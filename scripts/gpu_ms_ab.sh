#!/bin/bash
# A/B of library variants on ONE box: bench_multistreet.py, interleaved (gpurun -- scripts/gpu_ms_ab.sh TAG "" _variant ...)
cd $GRAFT_REPO_ROOT; TAG=${1:-ab}; shift; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
for rep in 1 2; do for V in "$@"; do
  export POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip$V.so
  timeout 300 python bench_multistreet.py --no-cpu-baseline --steps 30 > gpurun_out/${TAG}_ab.json 2>> gpurun_out/${TAG}_ab.err
  python -c "
import json
d = json.loads(open('gpurun_out/${TAG}_ab.json').read().strip().splitlines()[-1])
print('variant [$V] rep $rep: %.1f M node-updates/s, %.3f ms/iter, last-street %.3f ms/iter, exploitability %.9g' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['kernel_ms_per_iteration'], d['config']['exploitability_chips']))" | tee -a gpurun_out/${TAG}_ab.txt
done; done
