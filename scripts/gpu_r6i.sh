#!/bin/bash
# round 6: the per-street solver tables and the multi-street LBR test on the GPU.  gpurun -- bash scripts/gpu_r6i.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r90}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_lbr.py tests/test_gpu_parity.py tests/test_sharded.py -m gpu -q --durations=12 -p no:cacheprovider -k "per_street or multi_street_solution or discretized_nl_holdem_on_the_street or mixed" > gpurun_out/${TAG}_gpu_new_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_new_tests.txt; tail -n 30 gpurun_out/${TAG}_gpu_new_tests.txt
