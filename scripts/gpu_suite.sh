#!/bin/bash
# the whole GPU suite + the headline bench line (gpurun -- scripts/gpu_suite.sh TAG)
cd $GRAFT_REPO_ROOT; TAG=${1:-s}; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x --durations=8 > gpurun_out/${TAG}_gpu_tests.txt 2>&1; tail -n 25 gpurun_out/${TAG}_gpu_tests.txt
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 2500 gpurun_out/${TAG}_bench.json; tail -n 3 gpurun_out/${TAG}_bench.err
