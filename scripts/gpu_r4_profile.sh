#!/bin/bash
# round 4 evidence: headline bench (+cpu baseline), best-response bench, rocprofv3 kernel trace of both, PMC traffic (FETCH_SIZE / WRITE_SIZE, one
# counter per run, --kernel-trace only) and SQ groups of both
cd $GRAFT_REPO_ROOT; TAG=${1:-r06}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 600 gpurun_out/${TAG}_bench.json; tail -2 gpurun_out/${TAG}_bench.err
timeout 600 python bench_br.py > gpurun_out/${TAG}_bench_br.json 2> gpurun_out/${TAG}_bench_br.err; tail -c 500 gpurun_out/${TAG}_bench_br.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe"
BR="python $R/bench_br.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- $B > $R/gpurun_out/${TAG}_prof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe (262144 boards), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_prof -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_profbr -o ${TAG} -- $BR > $R/gpurun_out/${TAG}_profbr.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_br.py --steps 4 --warmup 1 --no-cpu-baseline (65536 boards), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_profbr -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_br_kernel_stats.txt 2>&1
head -8 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-160; head -5 $R/gpurun_out/${TAG}_br_kernel_stats.txt | cut -c1-160
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc$i -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_pmc$i.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmcbr$i -o p --output-format csv -- $BR > $R/gpurun_out/${TAG}_pmcbr$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <one counter group per run> -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe (262144 boards); mean per dispatch; checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmc1 $R/gpurun_out/${TAG}_pmc2 $R/gpurun_out/${TAG}_pmc3 $R/gpurun_out/${TAG}_pmc4 -name '*counter_collection.csv') | grep "fhp_pass\|sum_level\|==" | cut -c1-700; } > $R/gpurun_out/${TAG}_pmc.txt 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc <one counter group per run> -- python bench_br.py --steps 4 --warmup 1 --no-cpu-baseline (65536 boards); mean per dispatch; checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmcbr1 $R/gpurun_out/${TAG}_pmcbr2 $R/gpurun_out/${TAG}_pmcbr3 $R/gpurun_out/${TAG}_pmcbr4 -name '*counter_collection.csv') | grep "fhp_pass\|==" | cut -c1-700; } > $R/gpurun_out/${TAG}_br_pmc.txt 2>&1
grep "pass<[45], 0, 0, 1\|pass<2, 2" $R/gpurun_out/${TAG}_pmc.txt | cut -c1-330; grep "pass<2, 4" $R/gpurun_out/${TAG}_br_pmc.txt | cut -c1-330
rm -rf $R/gpurun_out/${TAG}_prof $R/gpurun_out/${TAG}_profbr $R/gpurun_out/${TAG}_pmc? $R/gpurun_out/${TAG}_pmcbr?
