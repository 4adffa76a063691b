#!/bin/bash
# small-tree kernels with the tree's arrays in LDS: parity tests of the Leduc family, bench_leduc.py per game
cd $GRAFT_REPO_ROOT; TAG=${1:-r23}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_plugin_surface.py -m gpu -x -q -p no:cacheprovider -k "leduc or Leduc or small or many or golden or reference" 2>&1 | tail -2 | tee gpurun_out/${TAG}_leduc_pytest.txt
for g in StandardLeduc DiscretizedNLLeduc; do timeout 300 python bench_leduc.py --game $g > gpurun_out/${TAG}_bench_leduc_$g.json 2> gpurun_out/${TAG}_bench_leduc.err; python -c "
import json;d=json.loads(open('gpurun_out/${TAG}_bench_leduc_$g.json').read().strip().splitlines()[-1]);print('$g', d['value'], d['ms_per_step'], d['many_solves_one_launch']['node_updates_per_s'], d['many_solves_one_launch']['ms_per_step'], d['cpu_baseline']['value'])"; done
