# quick A/B of the fused engine's launch configurations + fused parity tests; args: tag
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -k "fused or smoke" > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
for cfg in 0 1 2 3; do
  PRL_FHP_CFG=$cfg timeout 300 python bench.py --steps 5 --warmup 1 --boards 16384 --no-cpu-baseline > gpurun_out/bench${TAG}_cfg$cfg.log 2>&1; echo "rc=$?" >> gpurun_out/bench${TAG}_cfg$cfg.log
done
cd /tmp && export TMPDIR=/tmp
PRL_FHP_CFG=${2:-1} timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1$TAG -o r1$TAG -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --boards 16384 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
tail -4 gpurun_out/pytest_$TAG.log
for f in gpurun_out/bench${TAG}_cfg*.log; do echo $f; grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.e-]*' $f | tr '\n' ' '; echo; tail -2 $f | grep -v '^{' ; done
