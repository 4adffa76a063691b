"""
bench_br.py -- exact best response of an explicit strategy on the Flop5Holdem public tree (BASELINE.json config 4; secondary to
bench.py). One evaluation = PublicTree.fill_with_agent_policy's result already on the device (prl_solver_set_strategy is outside the
timed region) -> update_reach_probs -> compute_ev -> root exploitability (LocalBRMaster.py:67-80). N > 1 GPUs: boards sharded as in
bench.py (torch.distributed.run), one all-gather per evaluation.

    python bench_br.py [--boards B] [--reps K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=65536)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    from pokerrl_amd import _native as _nat
    _nat.set_device(int(os.environ.get("LOCAL_RANK", "0")))  # the library allocates on this process's GPU
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    import bench
    import parity_cases as pc
    from helpers import native_tree
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G

    boards = bench.seeded_boards(args.boards, 0, offset=rank * args.boards)
    tree = native_tree(G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards)
    if world > 1:
        from pokerrl_amd.dist import TorchExchange
        s = _native.NativeSolver(tree, "plus", 0, shard=(world, rank, TorchExchange("cuda")))
    else:
        s = _native.NativeSolver(tree, "plus", 0, engine="fused")
    nt = tree.n_cols - args.boards * 14
    full = pc.seeded_strategy_for_sharding(nt, args.boards, tree.range_size, 1 + rank)  # any valid strategy; float32 columns
    s.set_strategy(full)
    s.compute_ev()
    s.sync()
    t0 = time.perf_counter()
    for _ in range(args.reps):
        s.update_reach()
        s.compute_ev()
    expl = s.exploitability()  # synchronises
    dt = (time.perf_counter() - t0) / args.reps
    n_nodes = (tree.n_nodes - args.boards * 15) + args.boards * 15 * world
    if rank == 0:
        print(json.dumps({"metric": "exact best-response evaluations/s on the FHP public tree", "value": 1.0 / dt, "unit": "evaluations/s",
                          "node_visits_per_s": n_nodes / dt, "ms_per_evaluation": dt * 1e3, "n_gpus": world, "boards_per_gpu": args.boards,
                          "nodes_whole_tree": n_nodes, "exploitability_mbb_per_g": float(np.mean(expl) * 10.0), "data": "synthetic"}), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
