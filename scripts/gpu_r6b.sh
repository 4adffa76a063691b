#!/bin/bash
# round 6: selected GPU tests + the default bench line.   gpurun -- bash scripts/gpu_r6b.sh TAG "pytest -k expression"
cd $GRAFT_REPO_ROOT; TAG=${1:-r71}; K=${2:-"before_the_flop_vs_host or device_resident or solver_table"}; mkdir -p gpurun_out
PRL_LBRB_DEBUG=1 timeout 1500 python -m pytest tests -m gpu -q -x -k "$K" -p no:cacheprovider --durations=15 -s > gpurun_out/${TAG}_gpu_tests_selected.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_tests_selected.txt
grep -v "^LBR hand" gpurun_out/${TAG}_gpu_tests_selected.txt | tail -30
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 6000 gpurun_out/${TAG}_bench.json; tail -n 3 gpurun_out/${TAG}_bench.err
