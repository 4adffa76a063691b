#!/bin/bash
# LBR batch kernel: workgroup size A/B (576 = product, 768, 1024 lanes), same box; batched golden tests on each variant
cd $GRAFT_REPO_ROOT; TAG=${1:-r20}; mkdir -p gpurun_out
for v in product lbr768 lbr1024 product lbr768 lbr1024; do
  if [ $v = product ]; then unset POKERRL_AMD_LIB; else export POKERRL_AMD_LIB=$GRAFT_REPO_ROOT/pokerrl_amd/lib/libpokerrl_hip_$v.so; fi
  timeout 600 python bench_lbr.py --hands 524288 --cpu-hands 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.0f hands/s' % d['value'], 'device s %.3f' % d['config']['device_seconds_rank0'])" | tee -a gpurun_out/${TAG}_lbr_threads.txt
done
for v in lbr768 lbr1024; do POKERRL_AMD_LIB=$GRAFT_REPO_ROOT/pokerrl_amd/lib/libpokerrl_hip_$v.so timeout 600 python -m pytest tests/test_lbr.py -m gpu -x -q -p no:cacheprovider -k "batched_lbr" 2>&1 | tail -1 | tee -a gpurun_out/${TAG}_lbr_threads.txt; done
