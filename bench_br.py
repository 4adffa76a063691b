"""
bench_br.py -- exact best response of an explicit strategy on the Flop5Holdem public tree (BASELINE.json config 4; secondary to
bench.py). One evaluation = the work of LocalBRMaster.evaluate after the agent query (LocalBRMaster.py:67-80): the float32
strategy is already on the device (prl_solver_set_strategy = PublicTree.fill_with_agent_policy's result, outside the timed
region) -> update_reach_probs -> compute_ev -> root exploitability of both seats.

    python bench_br.py [--gpus N] [--boards B] [--steps K] [--warmup W] [--no-cpu-baseline]

The board pass of an evaluation is the best-response-only mode of the fused engine: both seats' walks, the strategy streamed
through the LDS prefetch like regrets and played as is -- no regret matching, no regret / average traffic.
roofline: algorithmic bytes per evaluation = 4*R*sum(A) + 4*R*N_boards (the strategy once + one rank vector per
board), over the summed duration of the board-pass launches (HIP events on the solver's stream, prl_solver_time_evaluations).
N > 1 GPUs: boards sharded as in bench.py (`--gpus N` starts the ranks itself), one all-gather of the chance node's partial sums
per evaluation over RCCL; weak scaling. cpu_baseline: the CPU oracle (1 thread) evaluating the same kind of strategy on a
bounded sample of boards -- the reference itself cannot run 2-hole-card public trees.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench  # noqa: E402

HBM_PEAK_GBPS = 8000.0
# HBM bytes of the best-response-only pass per board, from the PMC counters at 65536 boards with the sorted board storage of round 4:
# 2 x 2.441e6 + 4.24e4 KB per launch = 4.92 GB (round 3, hand-order columns: 5.71 GB). Below the algorithmic bytes, which count all 1326
# hands of a column: the storage holds the 1081 live ones.
PMC_TRAFFIC_BYTES_PER_BOARD = 4.924e9 / 65536
PMC_TRAFFIC_SOURCE = "profiles/r06_br_pmc.txt, re-measured in r50_br_pmc.txt: 2 * 2.442 GB read + 0.042 GB written (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate runs, FETCH_SIZE doubled per MI355X_MICROARCH.md)"


def seeded_strategy(n_trunk_cols, n_boards, R, seed):
    """float32 [n_cols, R] strategy of a Flop5Holdem tree (trunk: 2 + 2 actions; per board the 6 decision nodes with 2,2,3,2,3,2
    actions), random and normalised per node and hand (fill_random_random semantics, StrategyFiller.py:67-86)."""
    rng = np.random.RandomState(seed)
    sizes = [2] * (n_trunk_cols // 2) + [2, 2, 3, 2, 3, 2] * n_boards
    out = np.empty((sum(sizes), R), np.float32)
    at = 0
    for a in sizes:
        x = rng.random_sample((a, R)).astype(np.float32)
        out[at:at + a] = x / x.sum(axis=0, keepdims=True)
        at += a
    return out


def cpu_baseline(n_boards, reps):
    import oracle
    oracle.set_threads(1)
    boards = bench.seeded_boards(n_boards, 0)
    t = bench.fhp_tree(boards)
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, boards, 2, 52, 4, 2)
    strat = seeded_strategy(t.n_cols - 14 * n_boards, n_boards, t.range_size, 1)
    o.set_strategy(strat.astype(np.float64), False)  # includes the reach push-down
    o.compute_ev()                                   # builds the showdown plans (one-time)
    t0 = time.perf_counter()
    for _ in range(reps):
        o.update_reach()
        o.compute_ev()
    dt = (time.perf_counter() - t0) / reps
    return {"value": 1.0 / dt, "unit": "evaluations/s", "cores": 1, "kind": "port", "node_visits_per_s": t.n_nodes / dt,
            "sample": "exact BR of a seeded float32 strategy, Flop5Holdem tree x %d boards (%d nodes), %d evaluations, oracle/prl_oracle.c, "
                      "1 thread, %.1f s; scales linearly with the boards" % (n_boards, t.n_nodes, reps, dt * reps),
            "evaluations_per_s_at_bench_boards": None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--boards", type=int, default=65536, help="boards per GPU")
    ap.add_argument("--steps", "--reps", dest="steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-boards", type=int, default=512)
    ap.add_argument("--cpu-reps", type=int, default=6)
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import subprocess
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(bench.free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # PRL_BENCH_EMU_LIB (CPU test-suite only, tests/test_sharded.py): the ranks drive the emulator build of the library over gloo
    emu_lib = os.environ.get("PRL_BENCH_EMU_LIB")
    import torch
    from pokerrl_amd import _native
    lib = _native.bind(emu_lib) if emu_lib else None
    if not emu_lib:
        torch.cuda.set_device(local_rank)
        _native.require_device()
        _native.set_device(local_rank)  # the library allocates on this process's GPU
    dist = None
    if world > 1:
        import torch.distributed as dist
        if emu_lib:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    boards = bench.seeded_boards(args.boards, 0, offset=rank * args.boards)
    tree = bench.fhp_tree(boards, lib)
    exchange = None
    if world > 1:
        from pokerrl_amd.dist import TorchExchange
        exchange = TorchExchange("cpu" if emu_lib else "cuda")
        s = _native.NativeSolver(tree, "plus", 0, shard=(world, rank, exchange), _lib=lib)
    else:
        s = _native.NativeSolver(tree, "plus", 0, engine="fused", _lib=lib)
    nt = tree.n_cols - args.boards * 14
    s.set_strategy(seeded_strategy(nt, args.boards, tree.range_size, 1 + rank))  # any valid strategy; float32 columns
    s.time_evaluations(args.warmup)

    def barrier():
        if not emu_lib:
            torch.cuda.synchronize()
        s.sync()
        if dist is not None:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    dev_ms, pass_ms, n_pass = s.time_evaluations(args.steps)
    expl = s.exploitability()  # synchronises
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cpu" if emu_lib else "cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_nodes = (tree.n_nodes - args.boards * 15) + args.boards * 15 * world
    R = tree.range_size
    bytes_br = 4.0 * R * tree.n_cols + 4.0 * R * args.boards  # per GPU and evaluation
    kernel_ms = pass_ms if n_pass else dev_ms
    achieved = bytes_br * args.steps / (kernel_ms * 1e-3) / 1e9
    out = {"metric": "exact best-response evaluations/s on the FHP public tree", "value": args.steps / dt, "unit": "evaluations/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "exact best response (both seats) of a seeded float32 strategy on the Flop5Holdem public tree (blinds 50/100, "
                                  "stacks 20000, pot-size raises), %d seeded boards per GPU, 1326-hand ranges" % args.boards,
                      "boards_per_gpu": args.boards, "nodes_whole_tree": n_nodes, "node_visits_per_s": n_nodes * args.steps / dt,
                      "engine": s.engine, "exchanges": exchange.calls if exchange else 0,
                      "exploitability_mbb_per_g": float(np.mean(expl) * 10.0),
                      "exploitability_pinned_to": "oracle/ (C restatement of the reference with an explicit float32 summation order; the reference cannot build 2-hole-card trees)"},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                        "traffic": PMC_TRAFFIC_BYTES_PER_BOARD * args.boards if s.engine == "fused" else None, "traffic_source": PMC_TRAFFIC_SOURCE,
                        "kernel": "prl_k_fhp_pass<EVAL, STRAT32, STRAT32>", "launches_per_evaluation": n_pass / float(args.steps) if n_pass else None,
                        "kernel_ms_per_evaluation": kernel_ms / args.steps, "device_ms_per_evaluation": dev_ms / args.steps,
                        "bytes_per_evaluation_algorithmic": bytes_br}}
    if rank == 0:
        if not args.no_cpu_baseline and world == 1:
            cb = cpu_baseline(args.cpu_boards, args.cpu_reps)
            cb["evaluations_per_s_at_bench_boards"] = cb["node_visits_per_s"] / tree.n_nodes
            out["cpu_baseline"] = cb
        import ctypes
        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
