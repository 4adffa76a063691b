#!/bin/bash
# round 6: mixed streets on the per-street engine -- the new parity tests, the streets tests as regression, the whole-game vanilla fixture, then
# bench_multistreet: LimitHoldem default (regression against r74/r76), DiscretizedNLHoldem pot-sized raises on the per-street engine and on LEVELS.
#   gpurun -- bash scripts/gpu_r6d.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r81}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q --durations=25 -p no:cacheprovider -k "streets or multistreet or whole_game or all_in" > gpurun_out/${TAG}_gpu_streets_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_streets_tests.txt; tail -n 45 gpurun_out/${TAG}_gpu_streets_tests.txt
timeout 600 python bench_multistreet.py --steps 20 > gpurun_out/${TAG}_bench_multistreet.json 2> gpurun_out/${TAG}_bench_multistreet.err; tail -c 1500 gpurun_out/${TAG}_bench_multistreet.json
for eng in auto levels; do
  timeout 900 python bench_multistreet.py --game DiscretizedNLHoldem --flops 4 --turns 4 --rivers 4 --steps 20 --engine $eng --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_nl_${eng}.json 2> gpurun_out/${TAG}_bench_multistreet_nl_${eng}.err
  tail -c 1800 gpurun_out/${TAG}_bench_multistreet_nl_${eng}.json; tail -n 3 gpurun_out/${TAG}_bench_multistreet_nl_${eng}.err
done
timeout 900 python bench_multistreet.py --game DiscretizedNLHoldem --flops 16 --turns 8 --rivers 8 --steps 20 --cpu-iters 2 > gpurun_out/${TAG}_bench_multistreet_nl_big.json 2> gpurun_out/${TAG}_bench_multistreet_nl_big.err
tail -c 1800 gpurun_out/${TAG}_bench_multistreet_nl_big.json; tail -n 3 gpurun_out/${TAG}_bench_multistreet_nl_big.err
