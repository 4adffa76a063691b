#!/bin/bash
# Round 6, batched PokerEnv.step evidence on one box (gpurun -- bash scripts/gpu_r6_env.sh TAG): env + LBR GPU tests, the bench line, kernel statistics,
# PMC counters (traffic, SQ groups; each group its own run with --kernel-trace only), the per-workgroup timeline of an instrumented variant build
# (pokerrl_amd/lib/libpokerrl_hip_ebtl.so: PRL_VARIANT_ONLY=prl_envbatch.hip python -m pokerrl_amd.build --variant ebtl PRL_EB_TIMELINE).
TAG=$1; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$TAG; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_envbatch.py tests/test_lbr.py -m gpu -x -q 2>&1 | tail -4 > $O/env_lbr_gpu_tests.txt; cat $O/env_lbr_gpu_tests.txt
B="python bench_env.py --steps 40 --warmup 5 --no-cpu-baseline"
cd /tmp && export TMPDIR=/tmp
( cd $R && timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- $B > $O/kt.log 2>&1 )
python $R/scripts/rocprof_summary.py $(find $O/kt -name '*.db' | head -1) > $O/env_kernel_stats.txt 2>&1 || cat $(find $O/kt -name '*kernel_stats.csv' | head -1) > $O/env_kernel_stats.txt
grep -i "ebf_random_step\|kernel" $O/env_kernel_stats.txt | head -3
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU"; do
  i=$((i+1))
  ( cd $R && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $O/pmc$i -o p --output-format csv -- $B > $O/pmc$i.log 2>&1 )
done
python $R/scripts/env_counters.py $TAG $O/env_counters.json $(find $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 -name '*counter_collection.csv') > /dev/null
{ echo "# rocprofv3 --kernel-trace --pmc <one group per run> -- $B ; mean per dispatch; MI355X, checkpoint $TAG"; python $R/scripts/pmc_summary.py $(find $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4 -name '*counter_collection.csv') | grep "ebf\|==" | cut -c1-700; } > $O/env_pmc.txt
rm -rf $O/kt $O/pmc1 $O/pmc2 $O/pmc3 $O/pmc4
cd $R
cp $O/env_counters.json profiles/env_counters.json   # the bench line below reads it
python bench_env.py > $O/bench_env.json 2> $O/bench_env.err; python -c "
import json; d=json.loads(open('$O/bench_env.json').read().strip().splitlines()[-1]); print('bench_env %.4f ms  %.3e env-steps/s  frac %.3f  traffic %s' % (d['ms_per_step'], d['value'], d['roofline']['frac'], d['roofline']['traffic']))"
if [ -f pokerrl_amd/lib/libpokerrl_hip_ebtl.so ]; then POKERRL_AMD_LIB=$R/pokerrl_amd/lib/libpokerrl_hip_ebtl.so python scripts/r6_env_timeline.py > $O/env_timeline.txt 2>&1; head -4 $O/env_timeline.txt; fi
