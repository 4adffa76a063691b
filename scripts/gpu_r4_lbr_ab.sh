#!/bin/bash
# LBR batch kernel: product library (2 workgroups per CU, 96 VGPRs) against the 1-workgroup / 168-VGPR build, same box; golden tests on the product
cd $GRAFT_REPO_ROOT; TAG=${1:-r06ab}; mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_lbr.py -m gpu -x -q -p no:cacheprovider > gpurun_out/${TAG}_lbr_pytest.txt 2>&1; tail -2 gpurun_out/${TAG}_lbr_pytest.txt
for v in product lbrw3 product lbrw3; do
  if [ $v = product ]; then unset POKERRL_AMD_LIB; else export POKERRL_AMD_LIB=$GRAFT_REPO_ROOT/pokerrl_amd/lib/libpokerrl_hip_$v.so; fi
  timeout 600 python bench_lbr.py --hands 524288 --cpu-hands 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', '%.0f hands/s' % d['value'], 'device s %.3f' % d['config']['device_seconds_rank0'], 'frac %.4f' % d['roofline']['frac'])"
done
