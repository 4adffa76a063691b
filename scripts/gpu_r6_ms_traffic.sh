#!/bin/bash
# round 6: HBM traffic of the per-street engine's last-street passes from the PMC counters, LimitHoldem default tree and DiscretizedNLHoldem 16 x 8 x 8
# -> gpurun_out/TAG_multistreet_counters.json (copied to profiles/multistreet_counters.json: bench_multistreet.py reads it).  gpurun -- bash scripts/gpu_r6_ms_traffic.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r88}; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
OUT=$R/gpurun_out/${TAG}_multistreet_counters.json; rm -f $OUT
cd /tmp && export TMPDIR=/tmp
run() {  # KEY args...
  KEY=$1; shift; ARGS="$* --steps 12 --warmup 2 --no-cpu-baseline --placement-candidates 1"
  for c in FETCH_SIZE WRITE_SIZE; do
    ( cd $R && timeout 900 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_${KEY}_$c -o p --output-format csv -- python bench_multistreet.py $ARGS > $R/gpurun_out/${TAG}_${KEY}_$c.log 2>&1 )
  done
  F=$(find $R/gpurun_out/${TAG}_${KEY}_FETCH_SIZE -name '*counter_collection.csv' | head -1); W=$(find $R/gpurun_out/${TAG}_${KEY}_WRITE_SIZE -name '*counter_collection.csv' | head -1)
  python $R/scripts/multistreet_counters.py $KEY $R/gpurun_out/${TAG}_${KEY}_FETCH_SIZE.log $F $W --cmd "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench_multistreet.py $ARGS" --tag $TAG --into $OUT > $OUT.tmp && mv $OUT.tmp $OUT
  rm -rf $R/gpurun_out/${TAG}_${KEY}_FETCH_SIZE $R/gpurun_out/${TAG}_${KEY}_WRITE_SIZE
}
run LimitHoldem_4x2x2
run DiscretizedNLHoldem_16x8x8 --game DiscretizedNLHoldem --flops 16 --turns 8 --rivers 8
python - <<PY
import json
d = json.load(open("$OUT"))
for k, e in d.items():
    print(k, "%.3f GB per iteration, %.2f x algorithmic, %d kernel names" % (e["hbm_bytes_per_iteration_last_street"] / 1e9, e["traffic_over_algorithmic"], len(e["kernels"])))
PY
