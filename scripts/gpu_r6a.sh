#!/bin/bash
# round 6, first GPU call: the new solver-table tests, the whole suite with its durations, the LBR counters + bench lines.   gpurun -- bash scripts/gpu_r6a.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r70}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -x -k "solver_table or whole_game_flop5holdem or suit_classes_are or symmetrize" -p no:cacheprovider --durations=10 -s > gpurun_out/${TAG}_gpu_tests_new.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_tests_new.txt
tail -25 gpurun_out/${TAG}_gpu_tests_new.txt
timeout 900 python bench_lbr.py --game Flop5Holdem --agent table > gpurun_out/${TAG}_bench_lbr_fhp_table.json 2> gpurun_out/${TAG}_bench_lbr_fhp_table.err; tail -c 3000 gpurun_out/${TAG}_bench_lbr_fhp_table.json; tail -3 gpurun_out/${TAG}_bench_lbr_fhp_table.err
bash scripts/gpu_r6_lbr.sh $TAG
if [ -z "$NO_SUITE" ]; then
timeout 2400 python -m pytest tests -m gpu -q --durations=70 -p no:cacheprovider > gpurun_out/${TAG}_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_tests.txt
tail -n 90 gpurun_out/${TAG}_gpu_tests.txt
fi
