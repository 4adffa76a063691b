#!/bin/bash
# round 6: quick loop on the run-out-chain kernels -- the mixed parity tests, then the kernel-trace summary of the big NL tree.  gpurun -- bash scripts/gpu_r6g.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r84}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "discretized or all_in or mixed" > gpurun_out/${TAG}_gpu_mixed_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_mixed_tests.txt; tail -n 4 gpurun_out/${TAG}_gpu_mixed_tests.txt
bash scripts/gpu_r6e.sh ${TAG} | head -14
