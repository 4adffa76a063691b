#!/bin/bash
# knock-out timing of board-pass variants (results of the variants are WRONG by construction; only their clocks are read):
#   gpurun -- bash scripts/gpu_ko.sh TAG name=lib.so ...
cd $GRAFT_REPO_ROOT; TAG=$1; shift; mkdir -p gpurun_out
for rep in 1 2; do
for nv in "$@"; do
  n=${nv%%=*}; l=${nv#*=}
  POKERRL_AMD_LIB=$PWD/$l python bench.py --steps 10 --warmup 2 --boards ${BOARDS:-65536} --no-cpu-baseline --no-placement-probe --fixed-check-boards 0 2>/dev/null | tail -1 > gpurun_out/${TAG}_${n}_bench.json
  POKERRL_AMD_LIB=$PWD/$l python bench_br.py --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/${TAG}_${n}_br.json
  python - <<PY
import json
a=json.loads(open("gpurun_out/${TAG}_${n}_bench.json").read()); b=json.loads(open("gpurun_out/${TAG}_${n}_br.json").read())
print("%-10s rep $rep  CFR+ kernel %.3f ms/iter (${BOARDS:-65536} boards)  avg-eval %.3f ms   BR kernel %.3f ms" % ("$n", a["roofline"]["kernel_ms_per_iteration"], a["config"]["avg_strategy_evaluation_ms"], b["roofline"].get("kernel_ms_per_evaluation", 0) or b["ms_per_step"]))
PY
done
done 2>&1 | tee gpurun_out/${TAG}_ko.txt
