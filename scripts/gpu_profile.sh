#!/bin/bash
# Round checkpoint on the GPU box (gpurun -- scripts/gpu_profile.sh TAG): smoke, bench (+cpu baseline), rocprofv3 kernel trace of the same
# bench command, PMC passes (SQ, LDS conflicts, FETCH_SIZE and WRITE_SIZE each in their own run); summaries as text under
# gpurun_out/ (copy into profiles/). No tracing domain is combined with --pmc.
cd $GRAFT_REPO_ROOT; TAG=${1:-r02}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${TAG}_smoke.log 2>&1; tail -2 gpurun_out/${TAG}_smoke.log
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 400 gpurun_out/${TAG}_bench.json
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline"
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof -o ${TAG} -- $B > $R/gpurun_out/${TAG}_prof.log 2>&1
DB=$(find $R/gpurun_out/${TAG}_prof -name "*.db" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline (262144 boards), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $DB; } > $R/gpurun_out/${TAG}_kernel_stats.txt 2>&1
head -12 $R/gpurun_out/${TAG}_kernel_stats.txt | cut -c1-170
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $R/gpurun_out/${TAG}_pmc1 -o p1 --output-format csv -- $B > $R/gpurun_out/${TAG}_pmc1.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU -d $R/gpurun_out/${TAG}_pmc2 -o p2 --output-format csv -- $B > $R/gpurun_out/${TAG}_pmc2.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/${TAG}_pmc3 -o p3 --output-format csv -- $B > $R/gpurun_out/${TAG}_pmc3.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/${TAG}_pmc4 -o p4 --output-format csv -- $B > $R/gpurun_out/${TAG}_pmc4.log 2>&1
{ echo "# rocprofv3 --kernel-trace --pmc <one counter group per run> -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline (262144 boards); mean per dispatch; checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmc1 $R/gpurun_out/${TAG}_pmc2 $R/gpurun_out/${TAG}_pmc3 $R/gpurun_out/${TAG}_pmc4 -name '*counter_collection.csv') | grep "fhp_pass\|sum_level\|==" | cut -c1-700; } > $R/gpurun_out/${TAG}_pmc.txt 2>&1
grep "pass<[45]" $R/gpurun_out/${TAG}_pmc.txt | cut -c1-400
rm -rf $R/gpurun_out/${TAG}_prof $R/gpurun_out/${TAG}_pmc1 $R/gpurun_out/${TAG}_pmc2 $R/gpurun_out/${TAG}_pmc3 $R/gpurun_out/${TAG}_pmc4
