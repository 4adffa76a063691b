# timing-only experiments: library variants lib/libpokerrl_hip_<name>.so (wrong numerics allowed); args: names...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "$@"; do
  POKERRL_AMD_LIB=$GRAFT_REPO_ROOT/pokerrl_amd/lib/libpokerrl_hip_$v.so timeout 300 python bench.py --steps 5 --warmup 1 --boards 16384 --no-cpu-baseline > gpurun_out/benchexp_$v.log 2>&1
  echo $v; grep -o '"ms_per_step": [0-9.]*' gpurun_out/benchexp_$v.log; tail -1 gpurun_out/benchexp_$v.log | grep -v '^{'
done
