cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rocm-smi --showmeminfo vram 2>&1 | tail -5 > gpurun_out/smi.log
nproc > gpurun_out/host.log; free -g >> gpurun_out/host.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 1200 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 5 --warmup 1 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1a -o r1a -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof.log 2>&1; echo "prof rc=$?" >> $GRAFT_REPO_ROOT/gpurun_out/prof.log
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/smoke.log; tail -15 gpurun_out/pytest_gpu.log; tail -3 gpurun_out/bench.log; tail -3 gpurun_out/prof.log; ls -R gpurun_out/prof_r1a | head -20
