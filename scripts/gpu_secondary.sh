#!/bin/bash
# the secondary bench lines of a round (gpurun -- bash scripts/gpu_secondary.sh TAG): one JSON line each under gpurun_out/
cd $GRAFT_REPO_ROOT; TAG=${1:-r}; mkdir -p gpurun_out
run() { n=$1; shift; timeout 900 python "$@" > gpurun_out/${TAG}_$n.json 2> gpurun_out/${TAG}_$n.err; tail -c 300 gpurun_out/${TAG}_$n.json | head -c 300; echo; }
run bench_env bench_env.py
run bench_lbr bench_lbr.py
run bench_h2h bench_h2h.py
run bench_handeval bench_handeval.py
run bench_leduc bench_leduc.py
run bench_linear bench.py --variant linear --no-cpu-baseline --no-placement-probe
run bench_avg_f32 bench.py --avg-f32 --no-cpu-baseline
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-40s %14.4g %-22s frac %s" % (f.split("/")[-1], d["value"], d["unit"], d.get("roofline", {}).get("frac")))
    except Exception as e:
        print(f, "unreadable:", e)
PY
