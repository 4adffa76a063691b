# A/B of library variants on one box: bash scripts/gpu_ab.sh BOARDS variant... ("" = product library); prints value / ms per iteration / per-pass kernel ms
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
B=$1; shift
for rep in 1; do
for v in "$@"; do
  lib=pokerrl_amd/lib/libpokerrl_hip_$v.so; [ "$v" = product ] && lib=pokerrl_amd/lib/libpokerrl_hip.so
  POKERRL_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout 300 python bench.py --boards $B --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/ab_$v.log 2>&1
  echo "variant '${v}': $(grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"kernel_ms_per_iteration": [0-9.]*' gpurun_out/ab_$v.log | tr '\n' ' ')"
done
done
