#!/bin/bash
# same-box A/B of board-pass variants: every lib given (name=path) runs bench.py twice, interleaved; prints ms per iteration
# usage: scripts/gpu_ab.sh <boards> name=lib.so [name=lib.so ...]
B=$1; shift
mkdir -p gpurun_out
for rep in 1 2; do
  for nv in "$@"; do
    n=${nv%%=*}; l=${nv#*=}
    POKERRL_AMD_LIB=$PWD/$l python bench.py --steps 30 --warmup 5 --boards $B --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_${n}_$rep.json
    python - <<PY
import json
j=json.loads(open("gpurun_out/ab_${n}_$rep.json").read())
print("%-12s rep $rep  %.3f ms/iter  kernel %.3f ms  frac %.4f  %.1f M/s" % ("$n", j["ms_per_step"], j["roofline"]["kernel_ms_per_iteration"], j["roofline"]["frac"], j["value"]/1e6))
PY
  done
done
