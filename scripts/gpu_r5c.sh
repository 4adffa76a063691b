#!/bin/bash
# round 5: weighted boards / suit isomorphism on the GPU: its parity tests, then the whole Flop5Holdem game on one GPU (bench.py --whole-game)
cd $GRAFT_REPO_ROOT; TAG=${1:-r41}; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "weighted or isomorphism" -p no:cacheprovider 2>&1 | tail -4 | tee gpurun_out/${TAG}_pytest_iso.txt
timeout 900 python bench.py --whole-game --no-cpu-baseline --steps 20 --warmup 3 > gpurun_out/${TAG}_bench_whole_game.json 2> gpurun_out/${TAG}_bench_whole_game.err
tail -3 gpurun_out/${TAG}_bench_whole_game.err
python - <<PY
import json
d = json.loads(open("gpurun_out/${TAG}_bench_whole_game.json").read().strip().splitlines()[-1]); c = d["config"]
print("whole game: %.1f M class-node-updates/s, %.3f ms per whole-game iteration (%.1f it/s), frac %.3f, equivalent full-tree %.2f G node-updates/s, HBM %.1f GB, expl %.4f mbb/g, avg-strategy expl %.4f, avg eval %.2f ms, fixed check %s" % (
    d["value"] / 1e6, d["ms_per_step"], c["whole_game_iterations_per_s"], d["roofline"]["frac"], c["equivalent_full_tree_node_updates_per_s"] / 1e9, c["hbm_bytes_allocated"] / 1e9,
    c["exploitability_mbb_per_g"], c["avg_strategy_exploitability_mbb_per_g"], c["avg_strategy_evaluation_ms"], (c["fixed_problem_check"] or {}).get("exploitability_f32_hex")))
PY
