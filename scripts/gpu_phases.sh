# fused parity tests, phase breakdown (instrumented lib) for the given cfgs, then the A/B of the product lib; args: tag cfgs...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
TAG=${1:-x}; shift
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
tail -4 gpurun_out/pytest_$TAG.log
for cfg in ${@:-2}; do
  echo "== cfg $cfg" >> gpurun_out/phases_$TAG.log
  PRL_FHP_CFG=$cfg timeout 300 python scripts/phase_timing.py 16384 4 >> gpurun_out/phases_$TAG.log 2>&1
done
cat gpurun_out/phases_$TAG.log
for cfg in 0 1; do
  PRL_FHP_CFG=$cfg timeout 300 python bench.py --steps 5 --warmup 1 --boards 16384 --no-cpu-baseline > gpurun_out/bench${TAG}_cfg$cfg.log 2>&1; echo "rc=$?" >> gpurun_out/bench${TAG}_cfg$cfg.log
  echo cfg$cfg; grep -o '"value": [0-9.e+]*\|"ms_per_step": [0-9.]*\|"frac": [0-9.e-]*' gpurun_out/bench${TAG}_cfg$cfg.log | tr '\n' ' '; echo; tail -2 gpurun_out/bench${TAG}_cfg$cfg.log | grep -v '^{'
done
