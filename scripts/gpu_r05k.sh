#!/bin/bash
# checkpoint r05k (gpurun -- bash scripts/gpu_r05k.sh TAG): streets parity + both oracle fixtures at bench size, multi-street bench, PMC HBM traffic of
# the multi-street and best-response benches
cd $GRAFT_REPO_ROOT; TAG=${1:-r05k}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "streets or oracle_fixture" > gpurun_out/${TAG}_tests.txt 2>&1; tail -4 gpurun_out/${TAG}_tests.txt
bash scripts/gpu_ms_bench.sh $TAG
timeout 600 python bench_br.py > gpurun_out/${TAG}_bench_br.json 2> gpurun_out/${TAG}_bench_br.err; cut -c1-900 gpurun_out/${TAG}_bench_br.json
bash scripts/gpu_pmc_traffic.sh $TAG ms python bench_multistreet.py --steps 4 --warmup 1 --no-cpu-baseline
bash scripts/gpu_pmc_traffic.sh $TAG br python bench_br.py --steps 4 --warmup 1 --no-cpu-baseline
