cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "fused or fhp" > gpurun_out/pytest_fused.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_fused.log
for cfg in 0 1 2 3; do
  PRL_FHP_CFG=$cfg timeout 300 python bench.py --steps 5 --warmup 1 --boards 16384 --no-cpu-baseline > gpurun_out/bench_cfg$cfg.log 2>&1; echo "rc=$?" >> gpurun_out/bench_cfg$cfg.log
done
PRL_FHP_CFG=0 PRL_FHP_GRID=512 timeout 300 python bench.py --steps 5 --warmup 1 --boards 16384 --no-cpu-baseline > gpurun_out/bench_cfg0_g512.log 2>&1
PRL_FHP_CFG=0 PRL_FHP_GRID=256 timeout 300 python bench.py --steps 5 --warmup 1 --boards 16384 --no-cpu-baseline > gpurun_out/bench_cfg0_g256.log 2>&1
PRL_FHP_CFG=0 timeout 300 python bench.py --steps 5 --warmup 1 --boards 65536 --no-cpu-baseline > gpurun_out/bench_cfg0_b65536.log 2>&1
cd /tmp && export TMPDIR=/tmp
PRL_FHP_CFG=0 timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1b -o r1b -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --boards 16384 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_b.log 2>&1
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/pytest_fused.log
for f in gpurun_out/bench_cfg*.log; do echo $f; grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.e-]*\|"exploitability_mbb_per_g": [0-9.]*' $f | tr '\n' ' '; echo; tail -2 $f | grep -v '^{' ; done
