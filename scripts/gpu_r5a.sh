#!/bin/bash
# round 5, call A: BASELINE config 3 (Linear CFR) and the opt-in float32 average re-measured on the sorted storage -- bench lines, rocprofv3 kernel
# statistics, PMC traffic (FETCH_SIZE / WRITE_SIZE each in its own --kernel-trace-only run) -- and the default line with the steady first-allocation figure
cd $GRAFT_REPO_ROOT; TAG=${1:-r30}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
timeout 600 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --variant linear > gpurun_out/${TAG}_bench_linear.json 2>> gpurun_out/${TAG}_bench.err
timeout 600 python bench.py --variant vanilla --no-cpu-baseline > gpurun_out/${TAG}_bench_vanilla.json 2>> gpurun_out/${TAG}_bench.err
timeout 900 python bench.py --avg-f32 --no-cpu-baseline > gpurun_out/${TAG}_bench_avg_f32.json 2>> gpurun_out/${TAG}_bench.err
for f in bench bench_linear bench_vanilla bench_avg_f32; do python -c "
import json
try:
    d=json.loads(open('gpurun_out/${TAG}_$f.json').read().strip().splitlines()[-1]); c=d['config']; print('$f %.1f M, %.2f ms, frac %.3f, probe %s chosen %s, avg eval %.2f ms' % (d['value']/1e6, d['ms_per_step'], d['roofline']['frac'], c['placement_probe_ms_per_iteration'], c['placement_chosen'], c['avg_strategy_evaluation_ms']))
except Exception as e: print('$f', 'FAILED', e)"; done
cd /tmp && export TMPDIR=/tmp
for V in linear avg_f32; do
  if [ $V = linear ]; then F="--variant linear"; else F="--avg-f32"; fi
  B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe $F"
  export PRL_BENCH_SKIP_AVG_CHECK=1
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$V -o p -- $B > $R/gpurun_out/${TAG}_prof_$V.log 2>&1
  { echo "# rocprofv3 --kernel-trace --stats -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe $F (262144 boards), MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $(find $R/gpurun_out/${TAG}_prof_$V -name "*.db" | head -1); } > $R/gpurun_out/${TAG}_${V}_kernel_stats.txt 2>&1
  i=0
  for grp in FETCH_SIZE WRITE_SIZE; do
    i=$((i+1))
    timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc_${V}$i -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_pmc_${V}$i.log 2>&1
  done
  { echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one counter per run) -- python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe $F (262144 boards); mean per dispatch, KB as printed (FETCH_SIZE to be doubled: MI355X_MICROARCH.md); checkpoint $TAG"
    python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmc_${V}1 $R/gpurun_out/${TAG}_pmc_${V}2 -name '*counter_collection.csv') | grep "fhp_pass\|sum_level\|==" | cut -c1-300; } > $R/gpurun_out/${TAG}_${V}_pmc_traffic.txt 2>&1
  head -6 $R/gpurun_out/${TAG}_${V}_kernel_stats.txt | cut -c1-160; grep "fhp_pass" $R/gpurun_out/${TAG}_${V}_pmc_traffic.txt | cut -c1-200
  rm -rf $R/gpurun_out/${TAG}_prof_$V $R/gpurun_out/${TAG}_pmc_${V}?
done
