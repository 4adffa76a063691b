"""
bench_leduc.py -- CFR node-updates/s on the Leduc-family public trees (BASELINE.json configs 1-2; secondary to bench.py).
These trees fit in L2 (StandardLeduc: 465 nodes, R = 6), so the run is launch / latency bound: no roofline claim (SURVEY.md
section 8d), just node-updates/s and the CPU oracle beside it. N > 1 GPUs: independent replicas (the
tree does not shard) -- launch with torch.distributed.run, value = sum over replicas.

    python bench_leduc.py [--game StandardLeduc|DiscretizedNLLeduc|BigLeduc] [--variant plus|vanilla|linear] [--steps K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench_ref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--game", default="StandardLeduc")
    ap.add_argument("--variant", default="plus")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cpu-iters", type=int, default=20)
    ap.add_argument("--solves", type=int, default=256,
                    help="independent solves advanced by ONE launch, one workgroup (CU) each (prl_solver_iterations_many); 1 = a single tree")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    import oracle
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G

    game_cls = getattr(G, args.game)
    bets = bet_sets.POT_ONLY if args.game == "DiscretizedNLLeduc" else None
    stack = {"StandardLeduc": 13, "DiscretizedNLLeduc": 20000, "BigLeduc": 100}[args.game]
    boards = np.arange(game_cls.RULES.N_CARDS_IN_DECK, dtype=np.int8).reshape(-1, 1)  # one board per card, ascending
    tree = _native.NativeTree.for_game(game_cls, stack, bets, boards)
    s = _native.NativeSolver(tree, args.variant, 0, engine="levels")
    s.iterations(args.warmup)
    s.sync()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    dev_ms = s.time_iterations(args.steps)
    s.sync()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        dist.barrier()
    many = None
    if args.solves > 1 and s.engine == "levels":
        # a Leduc-sized tree occupies ONE CU: the GPU is filled by solving many trees at once -- here the same game at
        # args.solves different stack sizes (different trees), one workgroup each, all advanced by one launch per call
        stacks = [stack + i for i in range(args.solves)]
        trees = [_native.NativeTree.for_game(game_cls, st, bets, boards) for st in stacks]
        solvers = [_native.NativeSolver(t, args.variant, 0, engine="levels") for t in trees]
        try:
            _native.NativeSolver.iterations_many(solvers, args.warmup)
            for x in solvers:
                x.sync()
            t2 = time.perf_counter()
            _native.NativeSolver.iterations_many(solvers, args.steps)
            for x in solvers:
                x.sync()
            dt_many = time.perf_counter() - t2
            nodes_many = sum(t.n_nodes for t in trees)
            many = {"solves": args.solves, "stack_sizes": [stacks[0], stacks[-1]], "nodes_total": nodes_many, "seconds": dt_many,
                    "node_updates_per_s": nodes_many * args.steps / dt_many, "ms_per_step": dt_many * 1e3 / args.steps,
                    "mean_exploitability": float(np.mean([np.mean(x.exploitability()) for x in solvers]) * game_cls.EV_NORMALIZER)}
        except _native.NativeError as e:  # trees too large for the single-workgroup kernel (BigLeduc at large stacks)
            many = {"error": str(e)}
    out = {"metric": "%s CFR node-updates/s on the %s public tree (replicas)" % (args.variant, args.game),
           "value": tree.n_nodes * args.steps * world / dt, "unit": "node-updates/s", "n_gpus": world, "steps": args.steps,
           "ms_per_step": dt * 1e3 / args.steps, "device_ms_per_step": dev_ms / args.steps, "nodes": tree.n_nodes, "range_size": tree.range_size,
           "exploitability_mA_or_mbb_per_g": float(np.mean(s.exploitability()) * game_cls.EV_NORMALIZER), "engine": s.engine, "graph_replay": s.graph_replay,
           "note": "tree state fits in L2: launch / latency bound, no roofline claim (SURVEY.md 8d)", "data": "synthetic",
           "many_solves_one_launch": many}
    if rank == 0:
        r = game_cls.native_rules()
        o = oracle.Oracle({k: tree.field(k) for k in oracle.Oracle.FIELDS}, boards, r.n_hole_cards, r.n_cards, r.n_suits, r.rank_rule)
        o.cfr_reset({"vanilla": 0, "plus": 1, "linear": 2}[args.variant], 0)
        t1 = time.perf_counter()
        for _ in range(args.cpu_iters):
            o.cfr_iteration()
        out["cpu_baseline"] = {"value": tree.n_nodes * args.cpu_iters / (time.perf_counter() - t1), "unit": "node-updates/s", "cores": 1,
                               "kind": "port", "sample": "oracle/prl_oracle.c, %d iterations of the same tree" % args.cpu_iters,
                               # the reference's own cfr.iteration() on this game (it runs the Leduc family), timed by scripts/time_reference.py
                               "reference_python_node_updates_per_s": bench_ref.figure("cfr_iteration", "%s/%s" % (
                                   "DiscretizedNLLeduc_POT_ONLY" if args.game == "DiscretizedNLLeduc" else args.game,
                                   {"plus": "CFRPlus", "vanilla": "VanillaCFR", "linear": "LinearCFR"}[args.variant]), "node_updates_per_s"),
                               "reference_timing_source": bench_ref.SOURCE, "reference_timing_host": bench_ref.host()}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
