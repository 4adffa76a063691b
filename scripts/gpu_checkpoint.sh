# checkpoint run on the GPU box (gpurun -- bash scripts/gpu_checkpoint.sh TAG): full GPU test-suite, smoke, bench (+cpu baseline), rocprofv3 kernel trace, PMC passes. args: tag
cd $GRAFT_REPO_ROOT; TAG=${1:-e}; mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_$TAG.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_$TAG.log
tail -5 gpurun_out/pytest_$TAG.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/smoke_$TAG.log 2>&1; tail -2 gpurun_out/smoke_$TAG.log
timeout 900 python bench.py > gpurun_out/bench_$TAG.log 2>&1; echo "rc=$?" >> gpurun_out/bench_$TAG.log; tail -3 gpurun_out/bench_$TAG.log
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1$TAG -o r1$TAG -- $B > $GRAFT_REPO_ROOT/gpurun_out/prof_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $GRAFT_REPO_ROOT/gpurun_out/pmc1_$TAG -o p1 --output-format csv -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc1_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc2_$TAG -o p2 --output-format csv -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc2_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc3_$TAG -o p3 --output-format csv -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc3_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc4_$TAG -o p4 --output-format csv -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc4_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_r1$TAG gpurun_out/pmc1_$TAG | head
