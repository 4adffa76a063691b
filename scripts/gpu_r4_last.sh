#!/bin/bash
# after the small-tree kernel changes: smoke + every GPU test that runs the LEVELS engine / small trees (the whole suite last ran on the state before them: r21)
cd $GRAFT_REPO_ROOT; TAG=${1:-r25}; mkdir -p gpurun_out
timeout 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1 | tee gpurun_out/${TAG}_smoke.txt
timeout 700 python -m pytest tests/test_gpu_plugin_surface.py tests/test_f1_agents.py tests/test_cfr_hooks.py tests/test_wrappers.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "not bench_size and not streets_engine and not 262144 and not 65536" --durations=5 > gpurun_out/${TAG}_gpu_tests_levels.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_tests_levels.txt; tail -9 gpurun_out/${TAG}_gpu_tests_levels.txt
