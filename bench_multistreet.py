"""
bench_multistreet.py -- CFR+ on a multi-street public tree (secondary to bench.py): LimitHoldem (3 + 1 + 1 board
cards, full betting tree) over F flops x T turns x R rivers of seeded run-outs, 1326-hand ranges, on the level-synchronous (LEVELS)
engine -- the engine multi-street trees run on today; the number is the baseline the per-street fused pass of DESIGN.md section 8 has to beat.

    python bench_multistreet.py [--flops F] [--turns T] [--rivers R] [--steps K] [--warmup W]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


def runouts(n_flops, n_turns, n_rivers, seed=9):
    """deal-ordered run-outs: n_flops flops, below each n_turns turn cards, below each n_rivers river cards; no card twice in a row"""
    rng = np.random.RandomState(seed)
    rows = []
    for _ in range(n_flops):
        flop = sorted(int(c) for c in rng.choice(52, 3, replace=False))
        for t in [int(c) for c in rng.permutation(52) if c not in flop][:n_turns]:
            for r in [int(c) for c in rng.permutation(52) if c not in flop and c != t][:n_rivers]:
                rows.append(flop + [t, r])
    return np.array(rows, np.int8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flops", type=int, default=4)
    ap.add_argument("--turns", type=int, default=2)
    ap.add_argument("--rivers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    args = ap.parse_args()
    import torch
    torch.cuda.set_device(0)
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    _native.require_device()
    t0 = time.perf_counter()
    tree = _native.NativeTree.for_game(G.LimitHoldem, 48, None, runouts(args.flops, args.turns, args.rivers))
    t_tree = time.perf_counter() - t0
    s = _native.NativeSolver(tree, "plus", 0, engine="auto")
    s.iterations(args.warmup)
    s.sync()
    t0 = time.perf_counter()
    dev_ms = s.time_iterations(args.steps)
    s.sync()
    dt = time.perf_counter() - t0
    expl = s.exploitability()
    print(json.dumps({
        "metric": "CFR+ node-updates/sec on a multi-street LimitHoldem public tree", "value": tree.n_nodes * args.steps / dt, "unit": "node-updates/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "CFR+ (delay 0) on LimitHoldem, %d flops x %d turns x %d rivers of seeded run-outs, 1326-hand ranges" % (args.flops, args.turns, args.rivers),
                   "engine": s.engine, "nodes": tree.n_nodes, "action_columns": tree.n_cols, "board_rows": int(tree.n_boards), "tree_build_s": t_tree,
                   "device_ms_per_iteration": dev_ms / args.steps, "exploitability_chips": float(np.mean(expl)),
                   "hbm_bytes_allocated": int(s.get("bytes_allocated")[0])}}), flush=True)


if __name__ == "__main__":
    main()
