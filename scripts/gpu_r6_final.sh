#!/bin/bash
# round 6 evidence: smoke, every bench line (the default bench.py line carries config.whole_game: CFR+ and Linear CFR on the whole game), rocprofv3 kernel
# statistics of the headline / best-response runs, PMC traffic (FETCH_SIZE, WRITE_SIZE: one counter per run, --kernel-trace only) and the two SQ groups of
# the headline pass, the LBR counters (scripts/gpu_r6_lbr.sh).     gpurun -- bash scripts/gpu_r6_final.sh TAG        NO_PMC=1: skip the counter passes
cd $GRAFT_REPO_ROOT; TAG=${1:-r75}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt
timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --variant linear --no-cpu-baseline --no-whole-game-lines > gpurun_out/${TAG}_bench_linear.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --avg-f32 --no-cpu-baseline --no-whole-game-lines > gpurun_out/${TAG}_bench_avg_f32.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_br.py > gpurun_out/${TAG}_bench_br.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench_lbr.py > gpurun_out/${TAG}_bench_lbr.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench_lbr.py --game Flop5Holdem --agent table > gpurun_out/${TAG}_bench_lbr_fhp_table.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_env.py > gpurun_out/${TAG}_bench_env.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_multistreet.py > gpurun_out/${TAG}_bench_multistreet.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_multistreet.py --avg-f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_avg_f32.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench_multistreet.py --game DiscretizedNLHoldem --flops 16 --turns 8 --rivers 8 --cpu-iters 2 > gpurun_out/${TAG}_bench_multistreet_nl.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench_multistreet.py --game DiscretizedNLHoldem --stack 20000 --flops 8 --turns 4 --rivers 4 --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_nl200bb.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench_multistreet.py --game DiscretizedNLHoldem --stack 20000 --flops 8 --turns 4 --rivers 4 --no-cpu-baseline --engine levels > gpurun_out/${TAG}_bench_multistreet_nl200bb_levels.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_leduc.py > gpurun_out/${TAG}_bench_leduc.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_h2h.py > gpurun_out/${TAG}_bench_h2h.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_handeval.py > gpurun_out/${TAG}_bench_handeval.json 2>> gpurun_out/${TAG}_bench.err
for f in bench bench_linear bench_avg_f32 bench_br bench_lbr bench_lbr_fhp_table bench_env bench_multistreet bench_multistreet_avg_f32 bench_multistreet_nl bench_multistreet_nl200bb bench_multistreet_nl200bb_levels bench_leduc bench_h2h bench_handeval; do python -c "
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_$f.json').read().strip().splitlines()[-1]); print('$f', '%.5g' % d['value'], d.get('unit'), 'ms/step %.4g' % d['ms_per_step'], 'frac', (d.get('roofline') or {}).get('frac'), 'with-eval', (d.get('roofline_with_avg_evaluation') or {}).get('frac'))
    w=(d.get('config') or {}).get('whole_game')
    if w: print('   whole game: CFR+ %.2f ms (%.3f), Linear %.2f ms (%.3f)' % (w['cfr_plus']['ms_per_iteration'], w['cfr_plus']['roofline_frac'], w['linear_cfr']['ms_per_iteration'], w['linear_cfr']['roofline_frac']))
except Exception as e: print('$f', 'FAILED', e)"; done | tee gpurun_out/${TAG}_summary.txt
pushd /tmp > /dev/null; export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe --fixed-check-boards 0 --no-whole-game-lines"
BR="python $R/bench_br.py --steps 4 --warmup 1 --no-cpu-baseline"
for nv in "bench=$B" "br=$BR"; do
  n=${nv%%=*}; c=${nv#*=}
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$n -o p -- $c > $R/gpurun_out/${TAG}_prof_$n.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- $c   (MI355X, checkpoint $TAG)" | sed "s#$R/##g"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_prof_$n -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_${n}_kernel_stats.txt 2>&1
  head -7 $R/gpurun_out/${TAG}_${n}_kernel_stats.txt | cut -c1-170
  rm -rf $R/gpurun_out/${TAG}_prof_$n
done
popd > /dev/null
if [ -n "$NO_PMC" ]; then exit 0; fi
cd /tmp && export TMPDIR=/tmp
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
n=bench; c=$B; i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc_${n}$i -o p --output-format csv -- $c > $R/gpurun_out/${TAG}_pmc_${n}$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <one counter group per run> -- $c ; mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB as printed (FETCH_SIZE to be doubled: MI355X_MICROARCH.md); checkpoint $TAG" | sed "s#$R/##g"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmc_${n}1 $R/gpurun_out/${TAG}_pmc_${n}2 $R/gpurun_out/${TAG}_pmc_${n}3 $R/gpurun_out/${TAG}_pmc_${n}4 -name '*counter_collection.csv') | grep "fhp_pass\|sum_level\|==" | cut -c1-700; } > $R/gpurun_out/${TAG}_${n}_pmc.txt 2>&1
grep "pass<[45], 0, 0, 1\|pass<2, 2, 2" $R/gpurun_out/${TAG}_${n}_pmc.txt | cut -c1-330
rm -rf $R/gpurun_out/${TAG}_pmc_${n}?
cd $R; bash scripts/gpu_r6_lbr.sh $TAG
cd $R; bash scripts/gpu_r6_ms_traffic.sh $TAG
