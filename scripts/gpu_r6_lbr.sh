#!/bin/bash
# round 6: counters of the templated LBR kernel -> gpurun_out/TAG_lbr_counters.json (copied to profiles/lbr_counters.json: bench_lbr.py reads its
# VALU occupancy from there), kernel statistics, and the bench line.   gpurun -- bash scripts/gpu_r6_lbr.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r6}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
H=131072
C="python $R/bench_lbr.py --hands $H --cpu-hands 0 --no-warmup"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_lbr -o p -- $C > $R/gpurun_out/${TAG}_prof_lbr.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $C   (MI355X, checkpoint $TAG)" | sed "s#$R/##g"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_prof_lbr -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_lbr_kernel_stats.txt 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
i=0
for grp in "$SQ1" "$SQ2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc_lbr$i -o p --output-format csv -- $C > $R/gpurun_out/${TAG}_pmc_lbr$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <SQ group 1 | SQ group 2> (one group per run) -- $C ; mean per dispatch; MI355X, checkpoint $TAG" | sed "s#$R/##g"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmc_lbr1 $R/gpurun_out/${TAG}_pmc_lbr2 -name '*counter_collection.csv') | grep "lbr_batch\|==" | cut -c1-700; } > $R/gpurun_out/${TAG}_lbr_pmc_sq.txt 2>&1
python $R/scripts/lbr_counters.py $(find $R/gpurun_out/${TAG}_prof_lbr -name "*.db" | head -1) $(find $R/gpurun_out/${TAG}_pmc_lbr1 $R/gpurun_out/${TAG}_pmc_lbr2 -name '*counter_collection.csv') \
  --hands $H --tag $TAG --cmd "bench_lbr.py --hands $H --cpu-hands 0 --no-warmup" > $R/gpurun_out/${TAG}_lbr_counters.json
cat $R/gpurun_out/${TAG}_lbr_counters.json | head -40
rm -rf $R/gpurun_out/${TAG}_prof_lbr $R/gpurun_out/${TAG}_pmc_lbr1 $R/gpurun_out/${TAG}_pmc_lbr2
cd $R
mkdir -p /tmp/prof_copy && cp profiles/lbr_counters.json /tmp/prof_copy/ 2>/dev/null
cp gpurun_out/${TAG}_lbr_counters.json profiles/lbr_counters.json   # (on the box only: the bench below reads this run's counters)
timeout 900 python bench_lbr.py > gpurun_out/${TAG}_bench_lbr.json 2> gpurun_out/${TAG}_bench_lbr.err; tail -c 1500 gpurun_out/${TAG}_bench_lbr.json
