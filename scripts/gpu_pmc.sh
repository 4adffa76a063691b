# PMC passes over the bench (counters in their own runs, --kernel-trace only). args: tag cfg
cd $GRAFT_REPO_ROOT
TAG=${1:-x}; CFG=${2:-1}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --boards 16384 --no-cpu-baseline"
PRL_FHP_CFG=$CFG timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $GRAFT_REPO_ROOT/gpurun_out/pmc1_$TAG -o p1 --output-format csv -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc1_$TAG.log 2>&1
PRL_FHP_CFG=$CFG timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $GRAFT_REPO_ROOT/gpurun_out/pmc2_$TAG -o p2 --output-format csv -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc2_$TAG.log 2>&1
PRL_FHP_CFG=$CFG timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc3_$TAG -o p3 --output-format csv -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc3_$TAG.log 2>&1
PRL_FHP_CFG=$CFG timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $GRAFT_REPO_ROOT/gpurun_out/pmc4_$TAG -o p4 --output-format csv -- $B > $GRAFT_REPO_ROOT/gpurun_out/pmc4_$TAG.log 2>&1
cd $GRAFT_REPO_ROOT
ls -R gpurun_out/pmc1_$TAG | head; tail -3 gpurun_out/pmc1_$TAG.log
