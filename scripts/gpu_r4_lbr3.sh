#!/bin/bash
# quick LBR iteration: the batched golden tests only, a shorter bench, phase clocks
cd $GRAFT_REPO_ROOT; TAG=${1:-r14}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_lbr.py -m gpu -x -q -p no:cacheprovider -k "batched_lbr" > gpurun_out/${TAG}_lbr_pytest.txt 2>&1; tail -2 gpurun_out/${TAG}_lbr_pytest.txt
timeout 600 python bench_lbr.py --cpu-hands 0 > gpurun_out/${TAG}_bench_lbr.json 2> gpurun_out/${TAG}_bench_lbr.err; python -c "
import json;d=json.loads(open('gpurun_out/${TAG}_bench_lbr.json').read().strip().splitlines()[-1]);print('hands/s',d['value'],'frac',d['roofline']['frac'])"
POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip_lbrtiming.so timeout 600 python bench_lbr.py --hands 262144 --cpu-hands 0 > gpurun_out/${TAG}_lbr_timing.json 2> gpurun_out/${TAG}_lbr_phases.txt; grep "lbrb phase" gpurun_out/${TAG}_lbr_phases.txt | head -10
