#!/bin/bash
# round 6: mixed streets after the fused run-out-chain evaluation -- parity tests, NL benches, kernel-trace summary.  gpurun -- bash scripts/gpu_r6f.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r83}; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "streets or multistreet or all_in" > gpurun_out/${TAG}_gpu_streets_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_streets_tests.txt; tail -n 5 gpurun_out/${TAG}_gpu_streets_tests.txt
timeout 600 python bench_multistreet.py --game DiscretizedNLHoldem --flops 4 --turns 4 --rivers 4 --steps 20 --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_nl_auto.json 2> gpurun_out/${TAG}_bench_multistreet_nl_auto.err
tail -c 1500 gpurun_out/${TAG}_bench_multistreet_nl_auto.json
timeout 900 python bench_multistreet.py --game DiscretizedNLHoldem --flops 16 --turns 8 --rivers 8 --steps 20 --cpu-iters 2 > gpurun_out/${TAG}_bench_multistreet_nl_big.json 2> gpurun_out/${TAG}_bench_multistreet_nl_big.err
tail -c 1500 gpurun_out/${TAG}_bench_multistreet_nl_big.json; tail -n 3 gpurun_out/${TAG}_bench_multistreet_nl_big.err
bash scripts/gpu_r6e.sh ${TAG}
