"""
bench_multistreet.py -- CFR+ on a multi-street public tree (secondary to bench.py): LimitHoldem (3 + 1 + 1 board
cards, full betting tree) over F flops x T turns x R rivers of seeded run-outs, 1326-hand ranges. engine=auto runs it on the per-street
fused engine (csrc/prl_st.h); --engine levels is the level-synchronous engine these trees ran on before (10.3 M node-updates/s on the default
tree, profiles/r04p_bench_multistreet.json).

    python bench_multistreet.py [--gpus N] [--flops F] [--turns T] [--rivers R] [--steps K] [--warmup W] [--engine auto|levels] [--no-cpu-baseline]
                                [--game DiscretizedNLHoldem [--stack S]]

--game DiscretizedNLHoldem (round 6): pot-sized raises with finite stacks -- the streets' subtrees differ in shape and all-in calls are dealt out as
run-out chains ("mixed streets", csrc/prl_st.h).

N > 1 GPUs: the flops (first-deal outcomes) are sharded, F per GPU (weak scaling: N x F flops in all), the betting before the flop replicated,
one all-gather of the first street's root rows per EV pass (inside the library over RCCL; `--gpus N` starts the ranks itself as bench.py does).

roofline: algorithmic bytes per iteration 20 R sum(A) + 8 R N_rows of the LAST street (the dominant kernel: 94 % of the
nodes) over that kernel's summed launch time (HIP events on the solver's stream inside the timed region); the whole tree over the whole
iteration beside it. cpu_baseline: the oracle (1 thread) on one flop x one turn x one river of the same game.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
def pmc_traffic(game, flops, turns, rivers):
    """HBM bytes per last-street action column and CFR+ iteration from profiles/multistreet_counters.json (written by scripts/gpu_r6_ms_traffic.sh: rocprofv3
    --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md) -- (bytes per column, source text), or (None, why). The
    LimitHoldem figure scales to other run-out counts of the same game (one shape per street); a mixed-streets tree only has its own measurement."""
    path = os.path.join(ROOT, "profiles", "multistreet_counters.json")
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, "profiles/multistreet_counters.json not found"
    key = "%s_%dx%dx%d" % (game, flops, turns, rivers)
    if key not in d and game == "LimitHoldem":
        key = next((k for k in d if k.startswith("LimitHoldem_")), key)
    if key not in d:
        return None, "not measured on this tree (profiles/multistreet_counters.json has %s)" % ", ".join(sorted(d))
    e = d[key]
    return e["hbm_bytes_per_last_street_column"], "profiles/multistreet_counters.json [%s], checkpoint %s: %s" % (key, e.get("checkpoint"), e.get("command"))


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


def cpu_baseline(n_iters, game_cls, stack, bets):
    """CFR+ on the same game with the CPU oracle (1 thread): one flop x one turn x one river, full betting"""
    import oracle
    from pokerrl_amd import _native
    oracle.set_threads(1)
    t = _native.NativeTree.for_game(game_cls, stack, bets, runouts(1, 1, 1))
    r = game_cls.RULES
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, r.N_HOLE_CARDS, r.N_CARDS_IN_DECK, r.N_SUITS, r._RANK_RULE)
    o.cfr_reset(1, 0)
    t0 = time.perf_counter()
    for _ in range(n_iters):
        o.cfr_iteration()
    dt = time.perf_counter() - t0
    return {"value": t.n_nodes * n_iters / dt, "unit": "node-updates/s", "cores": 1, "kind": "port",
            "sample": "CFR+ delay 0, %s 1 flop x 1 turn x 1 river (%d nodes), %d iterations, oracle/prl_oracle.c, 1 thread, %.1f s" % (game_cls.__name__, t.n_nodes, n_iters, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--flops", type=int, default=4)
    ap.add_argument("--turns", type=int, default=2)
    ap.add_argument("--rivers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--engine", default="auto")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--avg-f32", action="store_true", help="OPT-IN, not the reference's numerics: the street columns' running average stored as float32 "
                    "(PRL_SOLVER_AVG_F32, on this engine since round 5); a second line, config.avg_dtype says so")
    ap.add_argument("--cpu-iters", type=int, default=3)
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--placement-candidates", type=int, default=3, help="one GPU: solver objects built and timed before the run, the fastest is kept (1 = no probe)")
    ap.add_argument("--game", default="LimitHoldem", choices=["LimitHoldem", "DiscretizedNLHoldem"], help="DiscretizedNLHoldem: pot-sized raises (bet_sets.POT_ONLY) -- mixed street "
                    "shapes and all-in run-out chains (csrc/prl_st.h MIXED STREETS)")
    ap.add_argument("--stack", type=int, default=None, help="chips per seat (default: 48 for LimitHoldem, 2500 for DiscretizedNLHoldem)")
    ap.add_argument("--max-raises", default=None, help="raises per betting round, e.g. 1,1,1,1 (smaller street subtrees: the CPU test-suite's emulator runs); default: the game's 4")
    args = ap.parse_args()
    import bench
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(bench.free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    emu_lib = os.environ.get("PRL_BENCH_EMU_LIB")  # CPU test-suite only: the ranks drive the emulator build of the library over gloo
    import torch
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    lib = _native.bind(emu_lib) if emu_lib else None
    if not emu_lib:
        torch.cuda.set_device(local_rank)
        _native.require_device()
        _native.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if emu_lib:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    t0 = time.perf_counter()
    per_flop = args.turns * args.rivers
    all_runouts = runouts(world * args.flops, args.turns, args.rivers)
    mine = all_runouts[rank * args.flops * per_flop:(rank + 1) * args.flops * per_flop]  # this rank's block of the flops, with their run-outs
    from pokerrl_amd.game import bet_sets
    game_cls = getattr(G, args.game)
    nl = args.game == "DiscretizedNLHoldem"
    stack = args.stack if args.stack is not None else (2500 if nl else 48)
    bets = bet_sets.POT_ONLY if nl else None
    if args.max_raises:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import env_args
        game = game_cls.native_game(env_args(game_cls, stack, bets))
        for i, v in enumerate(int(x) for x in args.max_raises.split(",")):
            game.max_raises[i] = v
        tree = _native.NativeTree(game, game_cls.native_rules(), mine, _lib=lib)
    else:
        tree = _native.NativeTree.for_game(game_cls, stack, bets, mine, _lib=lib)
    t_tree = time.perf_counter() - t0
    exchange = None
    placement = None
    force = bool(os.environ.get("PRL_BENCH_FORCE_EXCHANGE")) and world == 1 and not emu_lib  # the all-gather path of a multi-GPU run on one GPU (tests)
    if force:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if world > 1 or force:
        if emu_lib:
            from pokerrl_amd.dist import TorchExchange
            exchange = TorchExchange("cpu")
            s = _native.NativeSolver(tree, "plus", 0, shard=(world, rank, exchange), _lib=lib)
        else:
            from pokerrl_amd.dist import rccl_shard
            s = _native.NativeSolver(tree, "plus", 0, shard=rccl_shard(world, rank))
    else:
        avg_dtype = "f32" if args.avg_f32 else "f64"
        s = _native.NativeSolver(tree, "plus", 0, engine=args.engine, _lib=lib, avg_dtype=avg_dtype)
        if args.placement_candidates > 1 and not emu_lib and s.engine == "fused":
            # as bench.py's placement probe: solver objects of one process differ by ~7 % on this tree, repeatably per object, with where their
            # arrays land physically (scripts/ms_placement_probe.py: 2.38 .. 2.54 ms per iteration). 4 GB each: build a few, keep the fastest,
            # report all (config.placement_probe_ms_per_iteration); --placement-candidates 1 measures the first allocation as it is.
            def probe(sv):
                sv.iterations(3)
                sv.sync()
                return sv.time_iterations(6) / 6.0
            cands = [(probe(s), s)]
            try:
                for _ in range(args.placement_candidates - 1):
                    c = _native.NativeSolver(tree, "plus", 0, engine=args.engine, _lib=lib, avg_dtype=avg_dtype)
                    cands.append((probe(c), c))
            except _native.NativeError as e:
                sys.stderr.write("bench_multistreet.py: placement probe cut short (%s)\n" % e)
            placement = [ms for ms, _ in cands]
            s = min(cands, key=lambda x: x[0])[1]
            cands = None
            s.reset()

    def barrier():
        if not emu_lib:
            torch.cuda.synchronize()
        s.sync()
        if dist is not None:
            dist.barrier()

    s.iterations(args.warmup)
    barrier()
    t0 = time.perf_counter()
    dev_ms, pass_ms, n_pass = s.time_iterations_ex(args.steps)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cpu" if emu_lib else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    expl = s.exploitability()
    R = tree.range_size
    kind, rnd, nch = tree.field("kind"), tree.field("round"), tree.field("n_children")
    last = int(rnd.max())
    cols_last = int(np.sum(nch[(kind == 0) & (rnd == last)]))
    n_rows_last = int(np.sum(np.sum(tree.board_rows >= 0, axis=1) == tree.board_len))
    bytes_iter = 20.0 * R * tree.n_cols + 8.0 * R * int(tree.n_boards)
    bytes_last = 20.0 * R * cols_last + 8.0 * R * n_rows_last
    fused = s.engine == "fused" and n_pass > 0
    n_trunk = int(np.sum(rnd == int(rnd.min())))  # the betting before the first deal: replicated on every rank, counted once
    n_nodes_job = n_trunk + world * (tree.n_nodes - n_trunk)
    achieved = (bytes_last * args.steps / (pass_ms * 1e-3) if fused else bytes_iter * args.steps / (dev_ms * 1e-3)) / 1e9
    per_col, traffic_src = pmc_traffic(args.game, args.flops, args.turns, args.rivers)
    out = {
        "metric": "CFR+ node-updates/sec on a multi-street %s public tree" % args.game, "value": n_nodes_job * args.steps / dt, "unit": "node-updates/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic", "build_flavor": (lib or _native.lib()).prl_build_flavor().decode(),
        "config": {"avg_dtype": "f32 (opt-in: PRL_SOLVER_AVG_F32; the reference's average is float64)" if args.avg_f32 else "f64",
                   "workload": "CFR+ (delay 0) on %s (%d-chip stacks%s), %d flops per GPU x %d turns x %d rivers of seeded run-outs, 1326-hand ranges"
                               % (args.game, stack, ", pot-sized raises" if nl else "", args.flops, args.turns, args.rivers),
                   "engine": s.engine + (" (per-street)" if s.engine == "fused" else ""), "nodes": tree.n_nodes, "nodes_whole_job": n_nodes_job,
                   "placement_probe_ms_per_iteration": placement, "flops_per_gpu": args.flops, "exchanges": int(s.get("exchanges")[0]) if (world > 1 or force) else 0, "action_columns": tree.n_cols,
                   "action_columns_last_street": cols_last, "board_rows": int(tree.n_boards), "tree_build_s": t_tree,
                   "device_ms_per_iteration": dev_ms / args.steps, "exploitability_chips": float(np.mean(expl)), "iterations_done": s.iter,
                   "hbm_bytes_allocated": int(s.get("bytes_allocated")[0])},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": per_col * cols_last if fused and per_col and not args.avg_f32 and not args.max_raises else None, "traffic_source": traffic_src,
                     "kernel": "prl_k_st_pass<last street>" if fused else "all kernels of the iteration",
                     "launches_per_iteration": n_pass / float(args.steps) if fused else None,
                     "kernel_ms_per_iteration": (pass_ms if fused else dev_ms) / args.steps,
                     "bytes_per_iteration_algorithmic": bytes_last if fused else bytes_iter,
                     "bytes_per_iteration_algorithmic_whole_tree": bytes_iter,
                     "achieved_whole_iteration": bytes_iter * args.steps / (dev_ms * 1e-3) / 1e9,
                     "frac_whole_iteration": bytes_iter * args.steps / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_iters, game_cls, stack, bets)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
