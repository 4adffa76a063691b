#!/bin/bash
# round 6: the whole GPU suite with its durations, then the per-street A/B of HEAD against round 4's final library.  gpurun -- bash scripts/gpu_r6c.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r72}; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q --durations=40 -p no:cacheprovider > gpurun_out/${TAG}_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_tests.txt
tail -n 60 gpurun_out/${TAG}_gpu_tests.txt
if [ -f pokerrl_amd/lib/libpokerrl_hip_r4final.so ]; then bash scripts/gpu_ms_ab.sh ${TAG} "" _r4final; fi
