#!/bin/bash
# the whole GPU suite on the round's code
cd $GRAFT_REPO_ROOT; TAG=${1:-r06}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > gpurun_out/${TAG}_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_tests.txt
tail -14 gpurun_out/${TAG}_gpu_tests.txt
