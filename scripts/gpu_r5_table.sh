#!/bin/bash
# tabular agents on the GPU: the parity tests (batched engines = host evaluators playing the same table), LBR as a lower bound of the exact exploitability,
# the example's table.   gpurun -- bash scripts/gpu_r5_table.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r60}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_lbr.py -m gpu -q -k "table_agent or lower_bound" -p no:cacheprovider --durations=8 > gpurun_out/${TAG}_table_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_table_tests.txt
tail -12 gpurun_out/${TAG}_table_tests.txt
timeout 600 python examples/run_lbr_vs_cfrp_leduc.py > gpurun_out/${TAG}_lbr_vs_cfrp_leduc.txt 2>&1; tail -6 gpurun_out/${TAG}_lbr_vs_cfrp_leduc.txt
