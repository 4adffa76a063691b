#!/bin/bash
# round 6: after PRL_FHP_MAX_NODES 32 -> 40 and the 33-node street shape: streets tests, the default bench line (regression check of the board pass), the
# multi-street lines incl. DiscretizedNLHoldem at its 200-big-blind default on both engines.  gpurun -- bash scripts/gpu_r6j.sh TAG
cd $GRAFT_REPO_ROOT; TAG=${1:-r91}; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "streets or multistreet or all_in" > gpurun_out/${TAG}_gpu_streets_tests.txt 2>&1
echo "pytest rc=$?" >> gpurun_out/${TAG}_gpu_streets_tests.txt; tail -n 4 gpurun_out/${TAG}_gpu_streets_tests.txt
timeout 900 python bench.py --no-whole-game-lines > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
print("bench: %.1f M node-updates/s, %.2f ms/step, frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"]))
PY
timeout 600 python bench_multistreet.py --steps 20 --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet.json 2> gpurun_out/${TAG}_bench_multistreet.err
for eng in auto levels; do
  timeout 900 python bench_multistreet.py --game DiscretizedNLHoldem --stack 20000 --flops 8 --turns 4 --rivers 4 --steps 20 --engine $eng --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_nl200bb_${eng}.json 2> gpurun_out/${TAG}_bench_multistreet_nl200bb_${eng}.err
done
timeout 900 python bench_multistreet.py --game DiscretizedNLHoldem --flops 16 --turns 8 --rivers 8 --steps 20 --no-cpu-baseline > gpurun_out/${TAG}_bench_multistreet_nl_big.json 2> gpurun_out/${TAG}_bench_multistreet_nl_big.err
python - <<PY
import json
for n in ("bench_multistreet", "bench_multistreet_nl200bb_auto", "bench_multistreet_nl200bb_levels", "bench_multistreet_nl_big"):
    try:
        d = json.loads(open("gpurun_out/${TAG}_%s.json" % n).read().strip().splitlines()[-1])
        print("%s: %s, %d nodes: %.1f M node-updates/s, %.3f ms/iter, last-street frac %.3f, whole %.3f, expl %.6g" % (n, d["config"]["engine"], d["config"]["nodes"], d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["frac_whole_iteration"], d["config"]["exploitability_chips"]))
    except Exception as e:
        print(n, "failed", e)
PY
