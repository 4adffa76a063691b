# timing-only experiment: library variant given by $1 (wrong numerics allowed), cfgs 0..3
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for cfg in 0 1 2 3; do
  POKERRL_AMD_LIB=$GRAFT_REPO_ROOT/$1 PRL_FHP_CFG=$cfg timeout 300 python bench.py --steps 5 --warmup 1 --boards 16384 --no-cpu-baseline > gpurun_out/benchexp_cfg$cfg.log 2>&1
  echo cfg$cfg; grep -o '"ms_per_step": [0-9.]*' gpurun_out/benchexp_cfg$cfg.log; tail -1 gpurun_out/benchexp_cfg$cfg.log | grep -v '^{'
done
