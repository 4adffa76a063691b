#!/bin/bash
# HBM traffic of a bench's kernels from the PMC counters (gpurun -- bash scripts/gpu_pmc_traffic.sh TAG NAME "<bench command relative to the repo>"):
# FETCH_SIZE and WRITE_SIZE each in their own rocprofv3 run with --kernel-trace only (no other tracing domain), mean per dispatch and kernel
# name -> gpurun_out/TAG_NAME_pmc_traffic.txt. Units as rocprofv3 prints them (KB); on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes:
# double it for wide streaming reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is taken as printed.
cd $GRAFT_REPO_ROOT; TAG=$1; NAME=$2; shift 2; CMD="$*"; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd $R && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_${NAME}_$c -o p --output-format csv -- $CMD > $R/gpurun_out/${TAG}_${NAME}_$c.log 2>&1 )
done
{ echo "# rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (one counter per run) -- $CMD ; mean per dispatch, KB as printed; MI355X, checkpoint $TAG"
  python $R/scripts/pmc_summary.py $(find $R/gpurun_out/${TAG}_${NAME}_FETCH_SIZE $R/gpurun_out/${TAG}_${NAME}_WRITE_SIZE -name '*counter_collection.csv') | grep "pass<\|==" | cut -c1-300; } > $R/gpurun_out/${TAG}_${NAME}_pmc_traffic.txt 2>&1
cat $R/gpurun_out/${TAG}_${NAME}_pmc_traffic.txt | cut -c1-200
rm -rf $R/gpurun_out/${TAG}_${NAME}_FETCH_SIZE $R/gpurun_out/${TAG}_${NAME}_WRITE_SIZE
