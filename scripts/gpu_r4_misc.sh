#!/bin/bash
# LBR phase clocks at 1024 lanes; Leduc benches after the many-solves kernel's pointers became global ones
cd $GRAFT_REPO_ROOT; TAG=${1:-r22}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip_lbrtiming.so timeout 600 python bench_lbr.py --hands 262144 --cpu-hands 0 > gpurun_out/${TAG}_lbr_timing.json 2> gpurun_out/${TAG}_lbr_phases.txt; grep "lbrb phase" gpurun_out/${TAG}_lbr_phases.txt | head -10
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -p no:cacheprovider -k "leduc or Leduc or small or many" 2>&1 | tail -1
for g in StandardLeduc DiscretizedNLLeduc; do timeout 300 python bench_leduc.py --game $g > gpurun_out/${TAG}_bench_leduc_$g.json 2> gpurun_out/${TAG}_bench_leduc.err; python -c "
import json;d=json.loads(open('gpurun_out/${TAG}_bench_leduc_$g.json').read().strip().splitlines()[-1]);print('$g', d['value'], d['ms_per_step'], d['many_solves_one_launch']['node_updates_per_s'], d['many_solves_one_launch']['ms_per_step'])"; done
