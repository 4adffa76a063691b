#!/bin/bash
# quick checkpoint: a few fused GPU parity tests, headline bench, best-response bench, phase shares
cd $GRAFT_REPO_ROOT; TAG=${1:-r4c}; mkdir -p gpurun_out
if [ -n "$PYTEST" ]; then timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_engine_vs_oracle or fused_engine_best or twentyone or nine_node or fused_br" -p no:cacheprovider > gpurun_out/${TAG}_pytest.log 2>&1; tail -3 gpurun_out/${TAG}_pytest.log; fi
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench.json').read().strip().split('\n')[-1])
print('bench: value %.1f M  ms/step %.3f  frac %.4f  kernel_ms %.3f  avg_eval_ms %.3f  probe %s' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], d['roofline']['kernel_ms_per_iteration'], d['config']['avg_strategy_evaluation_ms'], d['config'].get('placement_probe_ms_per_iteration')))
PY
tail -2 gpurun_out/${TAG}_bench.err
timeout 600 python bench_br.py --no-cpu-baseline > gpurun_out/${TAG}_bench_br.json 2> gpurun_out/${TAG}_bench_br.err; python - <<PY
import json
d=json.loads(open('gpurun_out/${TAG}_bench_br.json').read().strip().split('\n')[-1])
print('bench_br: %.1f eval/s  kernel_ms %.3f  frac %.4f' % (d['value'], d['roofline']['kernel_ms_per_evaluation'], d['roofline']['frac']))
PY
tail -2 gpurun_out/${TAG}_bench_br.err
if [ -f pokerrl_amd/lib/libpokerrl_hip_timing.so ]; then
timeout 300 python scripts/phase_timing_br.py 32768 4 > gpurun_out/${TAG}_br_phases.txt 2>&1; grep -v "^barrier\|waiting" gpurun_out/${TAG}_br_phases.txt | head -8
timeout 300 python scripts/phase_timing.py 32768 4 > gpurun_out/${TAG}_cfr_phases.txt 2>&1; grep -v "^barrier\|waiting" gpurun_out/${TAG}_cfr_phases.txt | head -8
fi
