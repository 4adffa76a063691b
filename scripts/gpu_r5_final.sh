#!/bin/bash
# round 5 evidence: smoke, every bench line, rocprofv3 kernel statistics of the headline / best-response / whole-game runs, PMC traffic (FETCH_SIZE,
# WRITE_SIZE: one counter per run, --kernel-trace only) and the two SQ groups of the headline and the best-response pass.
#   gpurun -- bash scripts/gpu_r5_final.sh TAG
#   NO_PMC=1: skip the counter passes (kernel statistics only); PYTEST_K="expr": run those GPU tests first
cd $GRAFT_REPO_ROOT; TAG=${1:-r50}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
if [ -n "$PYTEST_K" ]; then
  timeout 900 python -m pytest tests -m gpu -q -k "$PYTEST_K" -p no:cacheprovider --durations=5 > gpurun_out/${TAG}_gpu_tests_selected.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_tests_selected.txt
  tail -8 gpurun_out/${TAG}_gpu_tests_selected.txt
fi
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2 | tee gpurun_out/${TAG}_smoke.txt
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --variant linear --no-cpu-baseline > gpurun_out/${TAG}_bench_linear.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --avg-f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_avg_f32.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --whole-game --no-cpu-baseline > gpurun_out/${TAG}_bench_whole_game.json 2>> gpurun_out/${TAG}_bench.err
pushd /tmp > /dev/null; export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe --fixed-check-boards 0 --no-whole-game-lines"
BR="python $R/bench_br.py --steps 4 --warmup 1 --no-cpu-baseline"
WG="python $R/bench.py --whole-game --steps 4 --warmup 1 --no-cpu-baseline --fixed-check-boards 0"
for nv in "bench=$B" "br=$BR" "whole_game=$WG"; do
  n=${nv%%=*}; c=${nv#*=}
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$n -o p -- $c > $R/gpurun_out/${TAG}_prof_$n.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- $c   (MI355X, checkpoint $TAG)" | sed "s#$R/##g"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_prof_$n -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_${n}_kernel_stats.txt 2>&1
  head -7 $R/gpurun_out/${TAG}_${n}_kernel_stats.txt | cut -c1-170
  rm -rf $R/gpurun_out/${TAG}_prof_$n
done
popd > /dev/null
timeout 600 python bench.py --whole-game --variant linear --no-cpu-baseline > gpurun_out/${TAG}_bench_whole_game_linear.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --whole-game --avg-f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_whole_game_avg_f32.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_multistreet.py --avg-f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_avg_f32.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_br.py > gpurun_out/${TAG}_bench_br.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench_lbr.py > gpurun_out/${TAG}_bench_lbr.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_env.py > gpurun_out/${TAG}_bench_env.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_multistreet.py > gpurun_out/${TAG}_bench_multistreet.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_leduc.py > gpurun_out/${TAG}_bench_leduc.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_h2h.py > gpurun_out/${TAG}_bench_h2h.json 2>> gpurun_out/${TAG}_bench.err
timeout 300 python bench_handeval.py > gpurun_out/${TAG}_bench_handeval.json 2>> gpurun_out/${TAG}_bench.err
for f in bench bench_linear bench_avg_f32 bench_whole_game bench_whole_game_linear bench_whole_game_avg_f32 bench_multistreet_avg_f32 bench_br bench_lbr bench_env bench_multistreet bench_leduc bench_h2h bench_handeval; do python -c "
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_$f.json').read().strip().splitlines()[-1]); print('$f', '%.5g' % d['value'], d.get('unit'), 'ms/step %.4g' % d['ms_per_step'], 'frac', (d.get('roofline') or {}).get('frac'), 'with-eval', (d.get('roofline_with_avg_evaluation') or {}).get('frac'))
except Exception as e: print('$f', 'FAILED', e)"; done
cd /tmp && export TMPDIR=/tmp
if [ -n "$NO_PMC" ]; then exit 0; fi
SQ1="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"
SQ2="SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"
for nv in "bench=$B" "br=$BR"; do
  n=${nv%%=*}; c=${nv#*=}; i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "$SQ1" "$SQ2"; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc_${n}$i -o p --output-format csv -- $c > $R/gpurun_out/${TAG}_pmc_${n}$i.log 2>&1
  done
  { echo "# rocprofv3 --kernel-trace --pmc <one counter group per run> -- $c ; mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB as printed (FETCH_SIZE to be doubled: MI355X_MICROARCH.md); checkpoint $TAG" | sed "s#$R/##g"
    python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmc_${n}1 $R/gpurun_out/${TAG}_pmc_${n}2 $R/gpurun_out/${TAG}_pmc_${n}3 $R/gpurun_out/${TAG}_pmc_${n}4 -name '*counter_collection.csv') | grep "fhp_pass\|sum_level\|==" | cut -c1-700; } > $R/gpurun_out/${TAG}_${n}_pmc.txt 2>&1
  grep "pass<[45], 0, 0, 1\|pass<2, 4, 4" $R/gpurun_out/${TAG}_${n}_pmc.txt | cut -c1-330
  rm -rf $R/gpurun_out/${TAG}_pmc_${n}?
done
# the 7-card evaluator: is it integer-issue bound? SQ counters of prl_k_hand_rank_boards (VALU instructions x 4 clocks / (1024 SIMDs x kernel clocks))
HE="python $R/bench_handeval.py"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_he -o p -- $HE > $R/gpurun_out/${TAG}_prof_he.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python bench_handeval.py (131072 boards x 1326 hands per call)   (MI355X, checkpoint $TAG)"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_prof_he -name "*.db" | head -1) | head -6; } > $R/gpurun_out/${TAG}_handeval_kernel_stats.txt 2>&1
i=0
for grp in "$SQ1" "$SQ2" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc_he$i -o p --output-format csv -- $HE > $R/gpurun_out/${TAG}_pmc_he$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <one counter group per run> -- python bench_handeval.py ; mean per dispatch; checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmc_he1 $R/gpurun_out/${TAG}_pmc_he2 $R/gpurun_out/${TAG}_pmc_he3 $R/gpurun_out/${TAG}_pmc_he4 -name '*counter_collection.csv') | grep "hand_rank\|==" | cut -c1-700; } > $R/gpurun_out/${TAG}_handeval_pmc.txt 2>&1
cat $R/gpurun_out/${TAG}_handeval_kernel_stats.txt | cut -c1-160; grep hand_rank $R/gpurun_out/${TAG}_handeval_pmc.txt | cut -c1-400
rm -rf $R/gpurun_out/${TAG}_prof_he $R/gpurun_out/${TAG}_pmc_he?
