#!/bin/bash
# round 6: the run-out forest on its own stream beside the last street's passes -- A/B on one box (PRL_ST_CHAIN_STREAM=0: on the solver's stream).  gpurun -- bash scripts/gpu_r6k.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r92}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_sharded.py -m gpu -q -p no:cacheprovider -k "discretized or all_in or mixed" > gpurun_out/${TAG}_gpu_mixed_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_mixed_tests.txt; tail -n 3 gpurun_out/${TAG}_gpu_mixed_tests.txt
for rep in 1 2; do for cs in 1 0; do export PRL_ST_GROUP_STREAMS=$cs;
  for cfg in "--flops 16 --turns 8 --rivers 8" "--stack 20000 --flops 8 --turns 4 --rivers 4"; do
    PRL_ST_CHAIN_STREAM=1 timeout 600 python bench_multistreet.py --game DiscretizedNLHoldem $cfg --steps 30 --no-cpu-baseline --placement-candidates 1 > gpurun_out/${TAG}_ab.json 2>> gpurun_out/${TAG}_ab.err
    python -c "
import json
d = json.loads(open('gpurun_out/${TAG}_ab.json').read().strip().splitlines()[-1])
print('streams $cs rep $rep [$cfg]: %.1f M node-updates/s, %.3f ms/iter (device %.3f), last-street %.3f ms/iter, exploitability %.9g' % (d['value'] / 1e6, d['ms_per_step'], d['config']['device_ms_per_iteration'], d['roofline']['kernel_ms_per_iteration'], d['config']['exploitability_chips']))" | tee -a gpurun_out/${TAG}_group_streams_ab.txt
  done
done; done
