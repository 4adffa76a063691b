#!/bin/bash
# round 4: LBR batch kernel with cooperative pairwise sums: golden tests, bench_lbr.py, phase clocks (timing variant)
cd $GRAFT_REPO_ROOT; TAG=${1:-r09}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_lbr.py -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_lbr_pytest.txt 2>&1; tail -3 gpurun_out/${TAG}_lbr_pytest.txt
timeout 900 python bench_lbr.py ${LBR_ARGS:-} > gpurun_out/${TAG}_bench_lbr.json 2> gpurun_out/${TAG}_bench_lbr.err; tail -c 1500 gpurun_out/${TAG}_bench_lbr.json; tail -2 gpurun_out/${TAG}_bench_lbr.err
POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip_lbrtiming.so timeout 600 python bench_lbr.py --hands 262144 --cpu-hands 0 > gpurun_out/${TAG}_lbr_timing.json 2> gpurun_out/${TAG}_lbr_phases.txt; grep "lbrb phase" gpurun_out/${TAG}_lbr_phases.txt | head -10
