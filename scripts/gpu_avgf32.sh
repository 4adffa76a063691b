#!/bin/bash
# the opt-in float32 running average: its GPU test, then bench.py with and without it on ONE box (gpurun -- scripts/gpu_avgf32.sh TAG)
cd $GRAFT_REPO_ROOT; TAG=${1:-f}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "float32_running or lbr_equity" 2>&1 | tail -n 4
timeout 900 python -m pytest tests/test_lbr.py tests/test_envbatch.py -m gpu -q -x 2>&1 | tail -n 4
for V in "" "--avg-f32"; do
  timeout 900 python bench.py --no-cpu-baseline $V > gpurun_out/${TAG}_bench$V.json 2>> gpurun_out/${TAG}_bench.err
  python -c "
import json
d = json.loads(open('gpurun_out/${TAG}_bench$V.json').read().strip().splitlines()[-1])
print('[$V] %.1f M node-updates/s, %.3f ms/iter, frac %.3f, probe %s, avg eval %.2f ms, check %s' % (d['value'] / 1e6, d['ms_per_step'], d['roofline']['frac'], d['config']['placement_probe_ms_per_iteration'], d['config']['avg_strategy_evaluation_ms'], d['config']['avg_f32_check']))"
done
