# instruction-cache and issue-stall counters of the board pass (16384 boards) -> gpurun_out/icache.txt
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --boards 16384 --steps 4 --warmup 1 --no-cpu-baseline"
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_INST_CYCLES[A-Z_]*\|SQ_WAIT_IFETCH\|SQ_IFETCH[A-Z_]*\|SQ_WAVE_[A-Z_]*\|SQ_THREAD_CYCLES_VALU\|SQ_INSTS_VALU_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $R/gpurun_out/avail.txt
timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $R/gpurun_out/ic1 -o p --output-format csv -- $B > $R/gpurun_out/ic1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAIT_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $R/gpurun_out/ic2 -o p --output-format csv -- $B > $R/gpurun_out/ic2.log 2>&1
{ cat $R/gpurun_out/avail.txt; echo; python $R/scripts/pmc_summary.py $(find $R/gpurun_out/ic1 $R/gpurun_out/ic2 -name '*counter_collection.csv') | grep "pass<[45]\|==" | cut -c1-400; tail -3 $R/gpurun_out/ic1.log $R/gpurun_out/ic2.log | cut -c1-300; } > $R/gpurun_out/icache.txt 2>&1
cat $R/gpurun_out/icache.txt
