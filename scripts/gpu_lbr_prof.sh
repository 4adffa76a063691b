# LBR: bench at 2^20 hands per seat + rocprofv3 kernel trace of a shorter run. args: tag
cd $GRAFT_REPO_ROOT; TAG=${1:-f}; mkdir -p gpurun_out
timeout 900 python bench_lbr.py --hands 1048576 --cpu-hands 60 > gpurun_out/bench_lbr_$TAG.log 2>&1; tail -1 gpurun_out/bench_lbr_$TAG.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_lbr_$TAG -o lbr$TAG -- python $GRAFT_REPO_ROOT/bench_lbr.py --hands 131072 --cpu-hands 0 > $GRAFT_REPO_ROOT/gpurun_out/prof_lbr_$TAG.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc_lbr_$TAG -o p1 --output-format csv -- python $GRAFT_REPO_ROOT/bench_lbr.py --hands 131072 --cpu-hands 0 > $GRAFT_REPO_ROOT/gpurun_out/pmc_lbr_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT; ls gpurun_out/prof_lbr_$TAG gpurun_out/pmc_lbr_$TAG
