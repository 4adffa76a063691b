#!/bin/bash
# rocprofv3 kernel-trace summaries of the secondary benches (gpurun -- scripts/gpu_profile_secondary.sh TAG); text summaries under gpurun_out/
cd $GRAFT_REPO_ROOT; TAG=${1:-r04}; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for b in br env lbr; do
  args=""; [ $b = lbr ] && args="--hands 131072 --cpu-hands 0"; [ $b = br ] && args="--no-cpu-baseline"
  timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_prof_$b -o $b -- python $R/bench_$b.py $args > $R/gpurun_out/${TAG}_prof_$b.log 2>&1
  DB=$(find $R/gpurun_out/${TAG}_prof_$b -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python bench_$b.py $args, MI355X, checkpoint $TAG"; python $R/scripts/rocprof_summary.py $DB; } > $R/gpurun_out/${TAG}_${b}_kernel_stats.txt 2>&1
  head -8 $R/gpurun_out/${TAG}_${b}_kernel_stats.txt | cut -c1-180
  rm -rf $R/gpurun_out/${TAG}_prof_$b
done
