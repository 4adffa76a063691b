#!/bin/bash
# same-box A/B of board-pass variants: every lib given (name=path) runs bench.py REPS times, interleaved; prints ms per iteration and,
# because the part is power-managed (a 1.3 kW kernel: runs land at 2.0-2.3 GHz depending on the box and the moment), the shader clock
# sampled beside the run and the board pass in Mcycles per iteration (ms x MHz) -- compare THAT between variants.
# usage: scripts/gpu_ab.sh <boards> name=lib.so [name=lib.so ...]      (REPS=n, default 2; ENV_<name>="VAR=1" adds an env var to a variant)
B=$1; shift
mkdir -p gpurun_out
for rep in $(seq 1 ${REPS:-2}); do
  for nv in "$@"; do
    n=${nv%%=*}; l=${nv#*=}
    ( while true; do rocm-smi --showclocks 2>/dev/null | grep sclk | sed 's/.*(\([0-9]*\)Mhz).*/\1/'; sleep 0.1; done ) > gpurun_out/ab_${n}_$rep.clk 2>/dev/null &
    W=$!
    ev=ENV_$n
    env ${!ev} POKERRL_AMD_LIB=$PWD/$l python bench.py --steps 30 --warmup 5 --boards $B --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/ab_${n}_$rep.json
    kill $W; wait $W 2>/dev/null
    python - <<PY
import json
j=json.loads(open("gpurun_out/ab_${n}_$rep.json").read())
clk=[int(x) for x in open("gpurun_out/ab_${n}_$rep.clk").read().split() if x.isdigit() and int(x) > 1500]
# the last samples of a run are the timed iterations (set-up and warm-up come first)
c=sum(clk[-3:])/max(len(clk[-3:]),1) if clk else 0.0
k=j["roofline"]["kernel_ms_per_iteration"]
print("%-12s rep $rep  %.3f ms/iter  kernel %.3f ms  frac %.4f  sclk %4.0f MHz  kernel %.1f Mcycles" % ("$n", j["ms_per_step"], k, j["roofline"]["frac"], c, k*c/1e3))
PY
  done
done
