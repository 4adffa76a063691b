#!/bin/bash
# round 4, final state: smoke, the whole GPU suite, every bench line, LBR kernel trace + SQ counters
cd $GRAFT_REPO_ROOT; TAG=${1:-r21}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt
timeout 1500 python -m pytest tests -m gpu -q --durations=8 -p no:cacheprovider > gpurun_out/${TAG}_gpu_tests.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_tests.txt
tail -5 gpurun_out/${TAG}_gpu_tests.txt
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; tail -c 400 gpurun_out/${TAG}_bench.json | head -c 400; echo
timeout 300 python bench_br.py > gpurun_out/${TAG}_bench_br.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench_lbr.py > gpurun_out/${TAG}_bench_lbr.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_env.py > gpurun_out/${TAG}_bench_env.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_multistreet.py > gpurun_out/${TAG}_bench_multistreet.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_leduc.py > gpurun_out/${TAG}_bench_leduc.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_h2h.py > gpurun_out/${TAG}_bench_h2h.json 2>> gpurun_out/${TAG}_bench.err
for f in bench bench_br bench_lbr bench_env bench_multistreet bench_leduc bench_h2h; do python -c "
import json,sys
try:
    d=json.loads(open('gpurun_out/${TAG}_$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d.get('unit'), (d.get('roofline') or {}).get('frac'))
except Exception as e: print('$f', 'FAILED', e)"; done
cd /tmp && export TMPDIR=/tmp
B="python $R/bench_lbr.py --hands 131072 --cpu-hands 0"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_lbrprof -o l -- $B > $R/gpurun_out/${TAG}_lbrprof.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_lbr.py --hands 131072 --cpu-hands 0, MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_lbrprof -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_lbr_kernel_stats.txt 2>&1
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
i=0
for grp in "$SQ1" "$SQ2"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_lbrpmc$i -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_lbrpmc$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <one SQ group per run> -- python bench_lbr.py --hands 131072 --cpu-hands 0; mean per dispatch; MI355X, checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_lbrpmc1 $R/gpurun_out/${TAG}_lbrpmc2 -name '*counter_collection.csv') | grep "lbr_batch\|==" | cut -c1-600; } > $R/gpurun_out/${TAG}_lbr_pmc_sq.txt 2>&1
head -3 $R/gpurun_out/${TAG}_lbr_kernel_stats.txt | cut -c1-160; grep lbr_batch $R/gpurun_out/${TAG}_lbr_pmc_sq.txt | cut -c60-420
rm -rf $R/gpurun_out/${TAG}_lbrprof $R/gpurun_out/${TAG}_lbrpmc?
