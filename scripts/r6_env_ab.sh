#!/bin/bash
# Same-box A/B of the batched PokerEnv.step (round 6): bench_env.py on variant libraries built by
#   PRL_VARIANT_ONLY=prl_envbatch.hip python -m pokerrl_amd.build --variant <name> <defines>
# Usage (GPU box): bash scripts/r6_env_ab.sh <tag> <variant> ...        (knock-out variants compute garbage: timing only)
T=$1; shift
O=gpurun_out/$T
mkdir -p $O
L=pokerrl_amd/lib
for rep in 1 2; do
  for v in "$@"; do
    POKERRL_AMD_LIB=$PWD/$L/libpokerrl_hip_$v.so python bench_env.py --no-cpu-baseline > $O/bench_env_${v}_$rep.json 2> $O/bench_env_${v}_$rep.err
    python - "$O/bench_env_${v}_$rep.json" "$v" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[2], "%.4f ms" % d["ms_per_step"], "%.3e env-steps/s" % d["value"], "frac %.3f" % d["roofline"]["frac"])
PY
  done
done | tee $O/ab.txt
