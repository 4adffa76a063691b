#!/bin/bash
# HBM traffic of the headline pass on the build as it is: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per run, --kernel-trace only.
#   gpurun -- bash scripts/gpu_r5_traffic.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r64}; mkdir -p gpurun_out; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-placement-probe --fixed-check-boards 0"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_pmc_bench$i -o p --output-format csv -- $B > $R/gpurun_out/${TAG}_pmc_bench$i.log 2>&1
done
{ echo "# rocprofv3 --kernel-trace --pmc <one counter per run> -- $B ; mean per dispatch; FETCH_SIZE / WRITE_SIZE in KB as printed (FETCH_SIZE to be doubled: MI355X_MICROARCH.md); checkpoint $TAG" | sed "s#$R/##g"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_pmc_bench1 $R/gpurun_out/${TAG}_pmc_bench2 -name '*counter_collection.csv') | grep "fhp_pass\|==" | cut -c1-400; } > $R/gpurun_out/${TAG}_bench_pmc_traffic.txt 2>&1
cat $R/gpurun_out/${TAG}_bench_pmc_traffic.txt | cut -c1-300
rm -rf $R/gpurun_out/${TAG}_pmc_bench?
