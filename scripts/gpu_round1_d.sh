cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_d.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_d.log
for cfg in 0 1 2 3; do
  PRL_FHP_CFG=$cfg timeout 300 python bench.py --steps 5 --warmup 1 --boards 16384 --no-cpu-baseline > gpurun_out/benchd_cfg$cfg.log 2>&1; echo "rc=$?" >> gpurun_out/benchd_cfg$cfg.log
done
cd /tmp && export TMPDIR=/tmp
PRL_FHP_CFG=1 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1d -o r1d -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --boards 16384 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_d.log 2>&1
cd $GRAFT_REPO_ROOT
tail -6 gpurun_out/pytest_d.log
for f in gpurun_out/benchd_cfg*.log; do echo $f; grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.e-]*\|"exploitability_mbb_per_g": [0-9.]*' $f | tr '\n' ' '; echo; tail -2 $f | grep -v '^{' ; done
