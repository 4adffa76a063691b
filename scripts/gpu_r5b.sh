#!/bin/bash
# round 5 iteration script: parity of the single-deal fused engine (oracle comparisons + fixtures incl. bench size), then the three numbers the board
# pass is judged on: bench.py (CFR+ headline + its average-strategy evaluation) and bench_br.py (best-response-only pass)
#   gpurun -- bash scripts/gpu_r5b.sh TAG [full]
cd $GRAFT_REPO_ROOT; TAG=${1:-r31}; mkdir -p gpurun_out
K="fused and not streets"; [ "$2" = quick ] && K="fused and not streets and not bench_size and not properties"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "$K" -p no:cacheprovider > gpurun_out/${TAG}_pytest_fused.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/${TAG}_pytest_fused.txt
tail -3 gpurun_out/${TAG}_pytest_fused.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
timeout 300 python bench_br.py --no-cpu-baseline > gpurun_out/${TAG}_bench_br.json 2>> gpurun_out/${TAG}_bench.err
python - <<PY
import json
for f in ("bench", "bench_br"):
    try:
        d = json.loads(open("gpurun_out/${TAG}_%s.json" % f).read().strip().splitlines()[-1]); c = d["config"]; r = d["roofline"]
        print(f, "%.4g %s, %.3f ms/step, frac %.3f, kernel ms %.3f" % (d["value"], d["unit"], d["ms_per_step"], r["frac"], r.get("kernel_ms_per_iteration") or r.get("kernel_ms_per_evaluation") or 0),
              "probe", c.get("placement_probe_ms_per_iteration"), "avg eval ms", c.get("avg_strategy_evaluation_ms"), "with avg frac", (d.get("roofline_with_avg_evaluation") or {}).get("frac"))
    except Exception as e:
        print(f, "FAILED", e)
PY
tail -3 gpurun_out/${TAG}_bench.err
