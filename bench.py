"""
bench.py -- CFR+ node-updates/s on a synthetic Flop5Holdem public tree (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--boards B] [--engine fused|levels] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
`python bench.py --gpus N` without a launcher (WORLD_SIZE unset) starts the N ranks itself through torch.distributed.run on
127.0.0.1 and relays rank 0's JSON line.

A "step" is one CFRBase.iteration() (reference semantics, PokerRL/cfr/_CFRBase.py:122-134: both seats updated, EVs
recomputed, current-strategy exploitability available) of CFR+ (delay 0) on the Flop5Holdem betting tree (blinds 50/100,
stacks 20000, pot-size raises; PokerRL/game/games.py:222-254) x B seeded boards per GPU (default 262144, a size SURVEY.md 8d config 3 names; 71 GB of HBM), 1326-hand ranges, float32 state
with the reference's float64 average strategy. Inputs are resident in HBM before the timed region. Prints ONE JSON line.

node-updates/s = (tree nodes incl. root) x iterations / s  (SURVEY.md section 8d).
roofline: the dominant kernel is the fused engine's board pass (prl_k_fhp_pass, >90% of the device time). Algorithmic bytes
per iteration = 20*R*sum(A) + 8*R*N_boards, all of them moved by that kernel; "achieved" = those
bytes over the summed duration of its launches, each bracketed by HIP events on the solver's stream inside the timed
region (prl_solver_time_iterations_ex). "traffic" = HBM bytes per iteration from the rocprofv3 PMC pass in profiles/.
N > 1: boards sharded over the ranks (one process per GPU), the trunk replicated, one all-gather of the chance node's
partial sums per EV pass over RCCL (pokerrl_amd/dist.py); weak scaling, value = nodes of the whole tree x iterations / s.
cpu_baseline: the CPU oracle (plain-C restatement of the reference, 1 thread) on a bounded sample of the same workload --
the reference itself cannot run 2-hole-card public trees.
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
# HBM bytes per CFR+ iteration of the board-pass kernels, from the PMC passes (FETCH_SIZE / WRITE_SIZE in their own rocprofv3 runs,
# FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for wide coalesced reads). Measured at 262144 boards with the sorted board
# storage of round 4 (profiles/r06_pmc.txt): UPDATE0_BR 2 * 17.69 GB read + 24.35 GB written, UPDATE1_EVAL1 2 * 17.71 + 24.45 = 119.6 GB per
# iteration (round 3, hand-order columns of 1326: 143.8 GB); every board subtree moves the same bytes (456 KB per board and iteration: 14
# regret columns of 1088 + the plan in, 7 regret columns out, 7 float64 average columns in and out -- the reference's float64 average is
# 53 % of it -- and the block rows of root vectors), so other sizes scale linearly.
PMC_TRAFFIC_BYTES_PER_BOARD_ITERATION = 119.6e9 / 262144
# the opt-in float32 running average on the sorted storage (round 5, profiles/r30_avg_f32_pmc_traffic.txt): UPDATE0_BR 2 * 13.76 GB read +
# 16.97 GB written, UPDATE1_EVAL1 2 * 13.75 + 16.89 = 88.9 GB per iteration = 0.89x the algorithmic bytes (which count 1326 entries per column
# where the storage holds 1088); round 3, hand-order columns: 106.1 GB
PMC_TRAFFIC_BYTES_PER_BOARD_ITERATION_AVG_F32 = 88.9e9 / 262144
# Linear CFR (BASELINE config 3; float32 avg_sum instead of the float64 running average), profiles/r30_linear_pmc_traffic.txt:
# 2 * 13.74 + 16.30 and 2 * 13.74 + 16.15 = 87.4 GB per iteration
PMC_TRAFFIC_BYTES_PER_BOARD_ITERATION_LINEAR = 87.4e9 / 262144
# the two-seat evaluation pass over the float64 averages (prl_k_fhp_pass<EVAL, AVG, AVG>): 2 * 18.16 GB read + 0.17 GB written (profiles/r50_bench_pmc.txt;
# round 4: 2 * 17.65 + 0.17)
PMC_TRAFFIC_BYTES_PER_BOARD_AVG_EVALUATION = 36.49e9 / 262144
PMC_TRAFFIC_SOURCE = ("profiles/r06_pmc.txt, re-measured on round 5's pass in r50_bench_pmc.txt and on round 6's final build in r76_bench_pmc.txt: 2 * 17.72 GB read + 24.2-24.3 GB written per launch = 119.4 GB (CFR+), "
                      "r30_avg_f32_pmc_traffic.txt (--avg-f32), r30_linear_pmc_traffic.txt (--variant linear): "
                      "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md")


def seeded_boards(n, seed, offset=0):
    """Boards offset .. offset+n-1 of a seeded shuffle of ALL C(52,5) = 2 598 960 five-card boards (distinct
    sorted boards from numpy.random.RandomState(seed)): rank r of a sharded run takes offset = r * n, so the shards are disjoint by
    construction and every rank is ready in about a second whatever its offset. int8 [n, 5], cards ascending. The tree builder's own
    enumeration (pokerrl_amd.game.board_enum: what PublicTree(Flop5Holdem) builds from when it is not handed boards)."""
    from pokerrl_amd.game import board_enum
    from pokerrl_amd.game import games as G
    return board_enum.single_deal_boards(G.Flop5Holdem, n_boards=n, seed=seed, offset=offset)


def fhp_tree(boards, lib=None):
    """The Flop5Holdem public tree of bench.py's workload (blinds 50/100, stacks 20000, pot-size raises) over `boards`."""
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    return _native.NativeTree.for_game(G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards, _lib=lib)


def cpu_baseline(n_boards, n_iters):
    """CFR+ on the same kind of tree with the CPU oracle (1 thread): node-updates/s on a bounded sample."""
    import oracle
    oracle.set_threads(1)
    boards = seeded_boards(n_boards, 0)
    t = fhp_tree(boards)  # host-side tree builder only
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, boards, 2, 52, 4, 2)
    o.cfr_reset(1, 0)
    t0 = time.perf_counter()
    for _ in range(n_iters):
        o.cfr_iteration()
    dt = time.perf_counter() - t0
    return {"value": t.n_nodes * n_iters / dt, "unit": "node-updates/s", "cores": 1, "kind": "port",
            "sample": "CFR+ delay 0, Flop5Holdem tree x %d boards (%d nodes), %d iterations, oracle/prl_oracle.c, 1 thread, %.1f s"
                      % (n_boards, t.n_nodes, n_iters, dt),
            "final_exploitability_mbb_per_g": float(np.mean(o.exploitability) * 10.0)}


def free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def shard_geometry(total, world, rank):
    """(boards of every rank before the last, boards of this rank) for ONE list of `total` boards: shards of whole canonical summation
    units -- 1024-board groups if that leaves the last rank something to do, else 32-board blocks, else single boards
    (prl_solver_create_sharded_ragged)."""
    for unit in (1024, 32, 1):
        per = -(-(-(-total // unit)) // world) * unit
        if (world - 1) * per < total:
            break
    mine = per if rank < world - 1 else total - (world - 1) * per
    return per, mine


def fixed_problem_check(total, iters, world, rank, how, lib, emu_lib):
    """The SAME small problem whatever the world size: ONE seeded list of `total` boards split over the ranks (ragged shards of whole canonical
    units, as --total-boards), `iters` CFR+ iterations through the run's own exchange path, exploitability of both seats as float32 bit
    patterns. The sharded solve is bit-identical to the one-GPU solve by construction (canonical chance sum): the driver's N = 1, 2, 4, 8
    lines carry this object, so the claim is checked by the scaling run itself -- equal hex strings in every line."""
    from pokerrl_amd import _native

    def all_ok(ok, what):
        """the ranks agree before they enter the next collective phase: one rank that failed locally (out of memory, a tree that does not build) must
        not leave the others waiting inside an all-gather -- every rank raises instead (round 5's advisor finding)"""
        if world > 1:
            import torch
            import torch.distributed as dist
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if emu_lib else "cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = ok and bool(int(flag.item()))
        if not ok:
            raise RuntimeError("fixed_problem_check: %s failed on this or another rank" % what)

    per, mine = shard_geometry(total, world, rank)
    tree, solver, err = None, None, None
    try:
        tree = fhp_tree(seeded_boards(mine, 0, offset=rank * per), lib)
    except Exception as e:  # noqa: BLE001
        err = e
    all_ok(tree is not None, "the tree build (%s)" % err)
    try:  # (creating a sharded solver is itself collective -- ncclCommInitRank, rccl_shard's own pre-flight agreement: every rank gets here)
        if world == 1:
            solver = _native.NativeSolver(tree, "plus", 0, engine="fused", _lib=lib)
        elif how == "rccl":
            from pokerrl_amd.dist import rccl_shard
            solver = _native.NativeSolver(tree, "plus", 0, shard=rccl_shard(world, rank, per, total, lib=lib), _lib=lib)
        else:
            from pokerrl_amd.dist import TorchExchange
            solver = _native.NativeSolver(tree, "plus", 0, shard=(world, rank, TorchExchange("cpu" if emu_lib else "cuda"), per, total), _lib=lib)
    except Exception as e:  # noqa: BLE001
        err = e
    all_ok(solver is not None, "the solver (%s)" % err)
    solver.iterations(iters)
    e = np.asarray(solver.exploitability(), np.float32)
    a = np.asarray(solver.eval_avg(), np.float32)
    return {"total_boards": total, "iterations": iters, "world": world, "exchange": how if world > 1 else None,
            "exploitability_f32_hex": [x.tobytes().hex() for x in e], "avg_strategy_exploitability_f32_hex": [x.tobytes().hex() for x in a],
            "exploitability_mbb_per_g": float(np.mean(e) * 10.0)}


def whole_game_lines(lib, steps, warmup, candidates=1):
    """BASELINE config 3 exactly as stated -- "LinearCFR on Flop Hold'em Poker (1326-combo ranges, full public tree), 1 MI355X" -- and CFR+ on the same tree,
    as two short HIP-event-timed regions inside the default line (round 5's verdict: the driver should see them): ALL 2 598 960 boards of Flop5Holdem
    through their 134 459 suit classes on this one GPU (prl_solver_create_weighted, ~30 GB). Per variant: ms per iteration (device, HIP events on the
    solver's stream), the board pass's share, whole-game iterations/s, the roofline fraction on the class tree's algorithmic bytes, the node-updates/s
    of the class tree and of the full tree it stands for; the average strategy's exploitability after the timed iterations (bit-exact to the oracle's
    chunked run of the same classes for the first four: tests/golden/fhp_whole_game_*_chunked.npz)."""
    from pokerrl_amd import _native
    from pokerrl_amd.game import board_enum
    from pokerrl_amd.game import games as G
    t0 = time.perf_counter()
    reps, mult = board_enum.single_deal_board_classes(G.Flop5Holdem)
    tree = fhp_tree(reps, lib)
    setup_s = time.perf_counter() - t0
    R, sum_a, n_classes = tree.range_size, tree.n_cols, len(reps)
    bytes_iter = 20.0 * R * sum_a + 8.0 * R * n_classes
    full_nodes = int(mult.sum()) * 15 + (tree.n_nodes - n_classes * 15)
    out = {"workload": "full-width iterations on the WHOLE Flop5Holdem public tree (blinds 50/100, stacks 20000, pot-size raises, 1326-hand ranges): all %d boards "
                       "through their %d suit classes, one GPU" % (int(mult.sum()), n_classes),
           "suit_classes": n_classes, "boards_represented": int(mult.sum()), "class_tree_nodes": tree.n_nodes, "full_tree_nodes_represented": full_nodes,
           "class_enumeration_and_tree_build_s": setup_s, "bytes_per_iteration_algorithmic": bytes_iter, "steps": steps, "warmup": warmup}
    for variant in ("plus", "linear"):
        # the same placement selection as the headline's (the pass's speed depends on where the arrays land: DESIGN.md section 4 "Spread"): candidates
        # built side by side by the library, each timed on `steps` steady iterations, the fastest kept; every candidate's figure is reported
        s = _native.NativeSolver(tree, variant, 0, _lib=lib, board_mult=mult, symmetrize=True, place=candidates if candidates > 1 else None, probe_iters=max(4, steps))
        s.iterations(warmup)
        s.sync()
        t1 = time.perf_counter()
        dev_ms, pass_ms, n_pass = s.time_iterations_ex(steps)
        s.sync()
        wall = time.perf_counter() - t1
        t2 = time.perf_counter()
        avg = s.eval_avg()
        eval_ms = (time.perf_counter() - t2) * 1e3
        k_ms = pass_ms if n_pass else dev_ms
        out[{"plus": "cfr_plus", "linear": "linear_cfr"}[variant]] = {
            "ms_per_iteration": wall * 1e3 / steps, "device_ms_per_iteration": dev_ms / steps, "board_pass_kernel_ms_per_iteration": k_ms / steps,
            "whole_game_iterations_per_s": steps / wall, "class_tree_node_updates_per_s": tree.n_nodes * steps / wall,
            "equivalent_full_tree_node_updates_per_s": full_nodes * steps / wall,
            "roofline_frac": bytes_iter * steps / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "achieved_gbps": bytes_iter * steps / (k_ms * 1e-3) / 1e9,
            "iterations_done": s.iter, "exploitability_mbb_per_g": float(np.mean(s.exploitability()) * 10.0),
            "avg_strategy_exploitability_mbb_per_g": float(np.mean(avg) * 10.0), "avg_strategy_evaluation_ms": eval_ms,
            "hbm_bytes_allocated": int(s.get("bytes_allocated")[0]),
            "placement_probe_ms_per_iteration": [x for x in s.placement_ms if x > 0.0] if s.placement_ms else None, "placement_chosen": s.placement_chosen}
        del s
    return out


def launch_ranks(n, argv):
    """`python bench.py --gpus N` outside a launcher: start the N ranks (one process per GPU) the way the driver would."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # the host driver only supports dmabuf IPC (RCCL across processes)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--boards", type=int, default=int(os.environ.get("PRL_BENCH_BOARDS", "262144")), help="boards per GPU")
    ap.add_argument("--total-boards", type=int, default=0,
                    help="solve ONE board list of this many boards split over the GPUs (strong scaling, ragged shards: every rank but the last "
                         "holds the same whole number of canonical summation units, the last one the rest); overrides --boards")
    ap.add_argument("--all-boards", action="store_true", help="--total-boards 2598960: every flop of Flop5Holdem, the whole game (about 88 GB "
                    "of HBM per GPU on 8 GPUs; does not fit fewer than 4)")
    ap.add_argument("--whole-game", action="store_true",
                    help="ONE GPU: the whole Flop5Holdem game -- all 2 598 960 boards through their 134 459 suit classes (multiplicities 4 / 12 / 24, "
                         "chance values averaged over every hand's suit orbit: prl_solver_create_weighted, an algorithmic extension the reference lacks; "
                         "~28 GB of HBM). A second bench line: config.workload says so; value counts the class nodes actually updated")
    ap.add_argument("--engine", default=os.environ.get("PRL_BENCH_ENGINE", "auto"))
    ap.add_argument("--variant", default="plus", choices=["plus", "linear", "vanilla"],
                    help="CFR variant (the metric is quoted on CFR+; BASELINE config 3 also names LinearCFR)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rccl", "torch"],
                    help="the all-gather of a sharded run: rccl = ncclAllGather inside the library on the solver's stream (default on GPUs), "
                         "torch = torch.distributed through the C ABI's callback")
    ap.add_argument("--avg-f32", action="store_true",
                    help="OPT-IN, not the reference's numerics: CFR+'s running average stored as float32 (PRL_SOLVER_AVG_F32; the float64 one is "
                         "54 %% of the board pass's HBM traffic). A SECOND bench line beside the default one: config.avg_dtype says so, and "
                         "config.avg_f32_check compares the average-strategy exploitability of both dtypes over 100 iterations on 4096 boards")
    ap.add_argument("--placement-candidates", type=int, default=3, help="one GPU: solver objects built and timed one after the other before the run, the fastest is kept")
    ap.add_argument("--no-placement-probe", dest="placement_probe", action="store_false",
                    help="one GPU: do not build a second set of arrays to keep the faster-placed one (DESIGN.md section 4)")
    ap.add_argument("--fixed-check-boards", type=int, default=-1,
                    help="after the timed region: one list of this many boards solved by all ranks together for 3 iterations, exploitability bits in "
                         "config.fixed_problem_check -- the same in the N = 1, 2, 4, 8 lines (default 8192; 0 = off; the emulator runs of the CPU suite: off unless given)")
    ap.add_argument("--no-whole-game-lines", dest="whole_game_lines", action="store_false",
                    help="N = 1: skip config.whole_game (CFR+ and Linear CFR on the whole game through its suit classes, two short timed regions after the headline)")
    ap.add_argument("--whole-game-steps", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-boards", type=int, default=384)
    ap.add_argument("--cpu-iters", type=int, default=24)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # PRL_BENCH_EMU_LIB (CPU test-suite only, tests/test_sharded.py): the ranks drive the emulator build of the library over
    # gloo, so that the launch / sharding / exchange / JSON plumbing of an N-rank run is exercised in the GPU-less container
    emu_lib = os.environ.get("PRL_BENCH_EMU_LIB")
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        if emu_lib:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    elif not emu_lib:
        torch.cuda.set_device(0)

    from pokerrl_amd import _native

    lib = _native.bind(emu_lib) if emu_lib else None
    if not emu_lib:
        _native.require_device()
        _native.set_device(local_rank if world > 1 else 0)  # the library allocates on this process's GPU, like torch above
    # every rank owns a contiguous block of the global board list (weak scaling: fixed boards per GPU)
    total = 2598960 if args.all_boards else args.total_boards
    shard_boards = args.boards
    if total:
        shard_boards, args.boards = shard_geometry(total, world, rank)
    board_mult = None
    if args.whole_game:
        assert world == 1 and not total, "--whole-game: one GPU (the classes of all boards fit one)"
        from pokerrl_amd.game import board_enum
        from pokerrl_amd.game import games as G
        boards, board_mult = board_enum.single_deal_board_classes(G.Flop5Holdem)
        args.boards = shard_boards = len(boards)
    else:
        boards = seeded_boards(args.boards, 0, offset=rank * shard_boards)
    tree = fhp_tree(boards, lib)
    exchange = None
    sharded = world > 1 or bool(os.environ.get("PRL_BENCH_FORCE_EXCHANGE"))  # the env knob runs the all-gather path on one GPU (tests)
    # the exchange: "rccl" = inside the library (ncclAllGather on the solver's stream, no Python in the iteration loop; the default on
    # GPUs), "torch" = torch.distributed through the C ABI's callback (the CPU test-suite's gloo runs; a fallback)
    how = args.exchange if args.exchange != "auto" else ("torch" if emu_lib else "rccl")
    if sharded and dist is None:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    if sharded and how == "rccl":
        from pokerrl_amd.dist import rccl_shard
        solver, err = None, None
        try:
            solver = _native.NativeSolver(tree, args.variant, 0, shard=rccl_shard(world, rank, shard_boards if total else None, total or None, lib=lib), _lib=lib)
        except (_native.NativeError, RuntimeError) as e:  # no librccl.so to bind (every rank raises: rccl_shard), or the communicator could not be set up: the callback path still works
            err = e
        ok = torch.tensor([0 if solver is None else 1], dtype=torch.int32, device="cpu" if emu_lib else "cuda")
        if world > 1:
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)  # the ranks take the same path: all inside the library, or all through the callback
        if int(ok.item()) == 0:
            sys.stderr.write("bench.py[rank %d]: the library's RCCL exchange is not available (%s); using torch.distributed\n" % (rank, err if err else "another rank failed"))
            solver = None
            how = "torch"
    if sharded and how == "torch":
        from pokerrl_amd.dist import TorchExchange
        exchange = TorchExchange("cpu" if emu_lib else "cuda")
        solver = _native.NativeSolver(tree, args.variant, 0, shard=(world, rank, exchange, shard_boards, total) if total else (world, rank, exchange), _lib=lib)
    placement, placement_chosen = None, None
    rccl_path = None
    if sharded and not emu_lib:
        import ctypes
        buf = ctypes.create_string_buffer(512)
        (lib or _native.lib()).prl_rccl_info(buf, 512)
        rccl_path = buf.value.decode("utf-8", "replace")
    if not sharded:
        avg_dtype = "f32" if args.avg_f32 else "f64"
        # The board pass streams within ~15 % of what HBM sustains and its speed depends on WHERE its arrays land physically: solver
        # objects of one process differ by up to 15 %, alternating between two levels ("Spread"). The LIBRARY handles it
        # (prl_solver_create_placed, NativeSolver(place=k)): it builds k solvers side by side (3 x 54 GB fit one GPU at this size), times each,
        # keeps the fastest. All timings are reported (config.placement_probe_ms_per_iteration; the first entry is what a plain create gets);
        # --no-placement-probe measures the first allocation as is.
        probe = args.placement_probe and not emu_lib and args.engine != "levels" and board_mult is None
        if board_mult is not None:
            solver = _native.NativeSolver(tree, args.variant, 0, _lib=lib, avg_dtype=avg_dtype, board_mult=board_mult, symmetrize=True)
        else:
            solver = _native.NativeSolver(tree, args.variant, 0, engine=args.engine, _lib=lib, avg_dtype=avg_dtype,
                                          place=args.placement_candidates if probe else None,
                                          # every candidate is timed the way the headline is: 3 untimed iterations, then as many steady ones as the
                                          # timed region has (HIP events on the solver's stream), so the first allocation's figure below is what a
                                          # user who does not probe gets, on the headline's own clock
                                          probe_iters=max(4, args.steps))
        if solver.placement_ms is not None and solver.engine == "fused":
            placement = [x for x in solver.placement_ms if x > 0.0]
            placement_chosen = solver.placement_chosen
    solver.sync()

    def barrier():
        if not emu_lib:
            torch.cuda.synchronize()
        solver.sync()
        if dist is not None:
            dist.barrier()

    solver.iterations(args.warmup)
    barrier()
    t0 = time.perf_counter()
    dev_ms, pass_ms, n_pass = solver.time_iterations_ex(args.steps)  # the K iterations, HIP events on the solver's stream
    barrier()
    dt = time.perf_counter() - t0
    per_rank = None
    if dist is not None:
        # every rank's own numbers travel to rank 0 (the SCALE line diagnoses itself: which rank is the slow one, how much of the step is the
        # board pass, what the exchange moves), the step time is the MAX over ranks
        mine = torch.tensor([dt, dev_ms, pass_ms], device="cpu" if emu_lib else "cuda", dtype=torch.float64)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[float(x) for x in t.tolist()] for t in allr]
        dt = max(r[0] for r in per_rank)

    # the reference's iteration() also evaluates the AVERAGE strategy every time (_CFRBase.py:134,218-262); timed separately
    # (outside the K steps) as one best-response-style evaluation per iteration
    t1 = time.perf_counter()
    n_avg = 3
    for _ in range(n_avg):
        avg_expl = solver.eval_avg()
    barrier()
    avg_eval_ms = (time.perf_counter() - t1) * 1e3 / n_avg

    avg_check = None
    if args.avg_f32 and rank == 0 and not os.environ.get("PRL_BENCH_SKIP_AVG_CHECK"):  # what the float32 storage costs in accuracy (the env knob: profiling runs of the passes alone): both dtypes, the same 4096 boards, 100 iterations
        small = fhp_tree(seeded_boards(4096, 0), lib)
        ev = {}
        for dt_ in ("f64", "f32"):
            sv = _native.NativeSolver(small, "plus", 0, engine="fused", _lib=lib, avg_dtype=dt_)
            sv.iterations(100)
            ev[dt_] = float(np.mean(sv.eval_avg()) * 10.0)
            cur = float(np.mean(sv.exploitability()) * 10.0)
            del sv
        avg_check = {"boards": 4096, "iterations": 100, "avg_strategy_exploitability_mbb_per_g_f64": ev["f64"], "avg_strategy_exploitability_mbb_per_g_f32": ev["f32"],
                     "relative_difference": abs(ev["f32"] - ev["f64"]) / ev["f64"], "current_strategy_exploitability_mbb_per_g_both": cur}
    fixed = None
    n_fixed = args.fixed_check_boards if args.fixed_check_boards >= 0 else (0 if emu_lib else 8192)
    if n_fixed > 0 and n_fixed >= world:
        # (the check's arrays are small beside the run's: 8192 boards = 1.8 GB over all ranks)
        try:  # (an extra after the timed region: whatever goes wrong here must not cost the line)
            fixed = fixed_problem_check(n_fixed, 3, world, rank, how if world > 1 else None, lib, emu_lib)
        except Exception as e:  # noqa: BLE001
            fixed = {"error": "%s: %s" % (type(e).__name__, e)}
        barrier()
    whole = None
    hbm_allocated, iterations_done, engine_name, n_exchanges = int(solver.get("bytes_allocated")[0]), solver.iter, solver.engine, int(solver.get("exchanges")[0])
    expl = solver.exploitability()
    if args.whole_game_lines and not os.environ.get("PRL_BENCH_NO_WHOLE_GAME") and world == 1 and not emu_lib and not args.whole_game and not sharded and not total and args.engine != "levels":
        solver = None  # the headline's 58 GB go back before the whole game's 30 GB come (both would fit; the pass's speed depends on placement)
        try:  # (after the timed region: it must not cost the line)
            whole = whole_game_lines(lib, args.whole_game_steps, 2, args.placement_candidates if args.placement_probe else 1)
        except Exception as e:  # noqa: BLE001
            whole = {"error": "%s: %s" % (type(e).__name__, e)}
    n_board_nodes = args.boards * 15
    n_nodes_total = (tree.n_nodes - n_board_nodes) + (total * 15 if total else n_board_nodes * world)  # one trunk + every rank's board subtrees
    value = n_nodes_total * args.steps / dt
    R, sum_a = tree.range_size, tree.n_cols
    bytes_iter = 20.0 * R * sum_a + 8.0 * R * args.boards  # per GPU
    kernel_ms = pass_ms if n_pass else dev_ms
    achieved = bytes_iter * args.steps / (kernel_ms * 1e-3) / 1e9
    pmc_ok = engine_name == "fused" and args.variant in ("plus", "linear")
    pmc_per_board = None
    if pmc_ok:
        pmc_per_board = (PMC_TRAFFIC_BYTES_PER_BOARD_ITERATION_LINEAR if args.variant == "linear" else
                         PMC_TRAFFIC_BYTES_PER_BOARD_ITERATION_AVG_F32 if args.avg_f32 else PMC_TRAFFIC_BYTES_PER_BOARD_ITERATION)
    out = {
        "metric": "CFR+ node-updates/sec on FHP public tree" if args.variant == "plus" else "%s CFR node-updates/sec on FHP public tree" % args.variant,
        "value": value, "unit": "node-updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong" if total else "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        # "hip-gfx950" = the product library; anything else (the SIMT emulator of the CPU test-suite, PRL_BENCH_EMU_LIB) is not a measurement
        "build_flavor": (lib or _native.lib()).prl_build_flavor().decode(),
        "config": {
            "workload": {"plus": "CFR+ (delay 0)", "linear": "Linear CFR", "vanilla": "vanilla CFR"}[args.variant] +
                        " full-width iterations on the Flop5Holdem public tree (blinds 50/100, stacks 20000, "
                        "pot-size raises), %s, 1326-hand ranges" % (
                            "THE WHOLE GAME: all 2598960 boards through their %d suit classes (multiplicities 4 / 12 / 24; chance values averaged over "
                            "the hands' suit orbits: prl_solver_create_weighted)" % args.boards if board_mult is not None else
                            "%d boards in all (%d on rank 0)" % (total, args.boards) if total else "%d seeded boards per GPU" % args.boards),
            # --whole-game: what the class solve stands for -- the full tree's nodes and the rate at which WHOLE-GAME iterations are done
            "boards_represented": int(board_mult.sum()) if board_mult is not None else None,
            "whole_game_iterations_per_s": (args.steps / dt) if board_mult is not None else None,
            "full_tree_nodes_represented": (int(board_mult.sum()) * 15 + (tree.n_nodes - n_board_nodes)) if board_mult is not None else None,
            "equivalent_full_tree_node_updates_per_s": ((int(board_mult.sum()) * 15 + (tree.n_nodes - n_board_nodes)) * args.steps / dt) if board_mult is not None else None,
            "boards_per_gpu": args.boards, "nodes_per_gpu": tree.n_nodes, "action_columns_per_gpu": sum_a,
            "engine": engine_name, "avg_dtype": "f32 (opt-in: PRL_SOLVER_AVG_F32; the reference's average is float64)" if args.avg_f32 else "f64", "avg_f32_check": avg_check,
            "parallelism": "boards sharded over %d GPU(s), trunk replicated, 1 all-gather of chance-node partial sums per EV pass" % world,
            "nodes_whole_tree": n_nodes_total, "exchanges": n_exchanges, "exchange": (how if sharded else None),
            "exchange_ms_mean": (exchange.seconds * 1e3 / max(exchange.calls, 1)) if exchange else None,
            # per rank: wall ms per step, device ms per iteration (HIP events on the solver's stream), board-pass kernel ms per iteration
            "per_rank_ms_per_step": [r[0] * 1e3 / args.steps for r in per_rank] if per_rank else None,
            "per_rank_device_ms_per_iteration": [r[1] / args.steps for r in per_rank] if per_rank else None,
            "per_rank_pass_ms_per_iteration": [r[2] / args.steps for r in per_rank] if per_rank else None,
            # what one all-gather moves: every rank contributes its canonical units (blocks / groups of boards) of <= 3 root vectors
            "exchange_bytes_per_pass_per_rank_upper": (int(-(-args.boards // 1024)) if args.boards % 1024 == 0 else int(-(-args.boards // 32)) if args.boards % 32 == 0 else args.boards) * 3 * tree.range_size * 4 if sharded else None,
            "rccl_library": rccl_path,
            "fixed_problem_check": fixed,
            # BASELINE config 3 as stated (Linear CFR, full public tree, one GPU) and CFR+ on the same whole game: whole_game_lines above
            "whole_game": whole,
            "iterations_done": iterations_done, "exploitability_mbb_per_g": float(np.mean(expl) * 10.0),
            "exploitability_pinned_to": "oracle/ (C restatement of the reference with an explicit float32 summation order; the reference cannot build 2-hole-card trees)",
            "avg_strategy_exploitability_mbb_per_g": float(np.mean(avg_expl) * 10.0), "avg_strategy_evaluation_ms": avg_eval_ms,
            "ms_per_step_with_avg_strategy_evaluation": dt * 1e3 / args.steps + avg_eval_ms,
            "hbm_bytes_allocated": hbm_allocated,
            "placement_probe_ms_per_iteration": placement,  # one entry per candidate allocation, built by the library (prl_solver_create_placed): the fastest was kept (None: not probed)
            "placement_chosen": placement_chosen, "first_allocation_probe_ms_per_iteration": placement[0] if placement else None,
            # the same number under the name it deserves since round 5: device ms per STEADY iteration (3 warm-up iterations, then `steps` timed ones) of
            # the first allocation, i.e. an unprobed NativeSolver(tree, ...) in this process on this box; and the node-updates/s it corresponds to
            "first_allocation_steady_ms_per_iteration": placement[0] if placement else None,
            "first_allocation_node_updates_per_s": (n_nodes_total / (placement[0] * 1e-3)) if placement else None,
        },
        # the reference's iteration() also evaluates the AVERAGE strategy every time (_CFRBase.py:134,218-262): the same figure for the
        # iteration WITH that evaluation pass -- algorithmic bytes of the evaluation: the float64 average once + one rank vector per board
        # (8 R sum(A) + 4 R N_boards), over kernel time of the iteration + the evaluation's wall time
        "roofline_with_avg_evaluation": {
            "bound": "hbm", "peak": HBM_PEAK_GBPS, "unit": "GB/s",
            "achieved": (bytes_iter + 8.0 * R * sum_a + 4.0 * R * args.boards) / ((kernel_ms / args.steps + avg_eval_ms) * 1e-3) / 1e9,
            "frac": (bytes_iter + 8.0 * R * sum_a + 4.0 * R * args.boards) / ((kernel_ms / args.steps + avg_eval_ms) * 1e-3) / 1e9 / HBM_PEAK_GBPS,
            "ms_per_iteration": kernel_ms / args.steps + avg_eval_ms, "bytes_per_iteration_algorithmic": bytes_iter + 8.0 * R * sum_a + 4.0 * R * args.boards,
            "kernel": "prl_k_fhp_pass: the two update passes + the two-seat evaluation pass over the float64 averages",
            "traffic": ((pmc_per_board + PMC_TRAFFIC_BYTES_PER_BOARD_AVG_EVALUATION) * args.boards
                        if (pmc_per_board is not None and not args.avg_f32 and args.variant == "plus") else None)},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                     "traffic": (pmc_per_board * args.boards if pmc_per_board is not None else None),
                     "traffic_source": PMC_TRAFFIC_SOURCE,
                     "kernel": "prl_k_fhp_pass" if n_pass else "all kernels of the iteration",
                     "launches_per_iteration": n_pass / float(args.steps) if n_pass else None,
                     "kernel_ms_per_iteration": kernel_ms / args.steps, "device_ms_per_iteration": dev_ms / args.steps,
                     "bytes_per_iteration_algorithmic": bytes_iter,
                     # the PMC-measured bytes over the same kernel time: what the kernel actually pulls through HBM (float64 averages
                     # included), against the ~6.3 TB/s MI355X_MICROARCH.md gives as sustained
                     "traffic_rate_gbps": (pmc_per_board * args.boards * args.steps / (kernel_ms * 1e-3) / 1e9
                                           if pmc_per_board is not None else None),
                     "sustained_hbm_gbps": 6300.0,
                     "achieved_whole_iteration": bytes_iter * args.steps / (dev_ms * 1e-3) / 1e9},
    }
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N=1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_boards, args.cpu_iters)
        import ctypes
        ctypes.CDLL(None).fflush(None)  # RCCL's start-up banner sits in the C stdio buffer: push it out before the JSON line
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
