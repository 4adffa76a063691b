"""
bench_lbr.py -- LBR hands/s with the device-resident batched engine (BASELINE.json config 5; secondary metric, bench.py is the
driver's contract). DiscretizedNLHoldem (blinds 50/100, stacks 20000), agent bet set B_5, LBR bet set OFF_TREE_11,
lbr_check_to_round = TURN (LBRArgs.py:16-18,31-35), 2^20 hands per agent seat, synthetic hash agent, counter-based decks keyed
by (seed, hand id).

    python bench_lbr.py [--gpus N] [--hands H] [--agent hash|uniform] [--cpu-hands M]
    python bench_lbr.py --game Flop5Holdem --agent table [--solver-iters 100]      LBR against the WHOLE-GAME solution of Flop5Holdem: CFR+ over all
        2 598 960 boards (134 459 suit classes, bench.py --whole-game), its average strategy as a suit-canonical policy table built on the device
        (PolicyTable.from_solver, 12.8 GB), LBR acting from the flop; the line carries the solver's exact exploitability beside LBR's winnings

N > 1 GPUs (`--gpus N` starts the ranks itself; or torch.distributed.run): hands are independent (LocalLBRMaster.py:53-69 splits
them over workers the same way); rank r plays hands [r * H/N, (r + 1) * H/N) of the same deck / agent-draw streams and
(sum, sum of squares, n) are all-reduced over RCCL (BatchedLBR.run_sharded) -- strong scaling of a fixed evaluation.
cpu_baseline: this package's host LocalLBRWorker (the Python episode loop of the reference, pinned to the reference's per-hand
winnings by tests/golden/lbr_*.npz; its equity queries run on the GPU) timed in the same run on a bounded number of hands.
The reference's own worker measured 30 hands/s on one core in this configuration; it cannot run on the GPU box.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench_ref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--hands", type=int, default=1 << 20, help="hands per agent seat (whole job)")
    ap.add_argument("--agent", default="hash", help="hash | uniform | table (--game Flop5Holdem: the whole-game CFR+ average strategy)")
    ap.add_argument("--game", default="DiscretizedNLHoldem", help="DiscretizedNLHoldem (config 5) | Flop5Holdem")
    ap.add_argument("--solver-iters", type=int, default=100, help="--agent table: CFR+ iterations of the whole-game solve the table is made of")
    ap.add_argument("--no-warmup", action="store_true", help="profiling runs: exactly one launch per seat")
    ap.add_argument("--lbr-from-the-start", action="store_true",
                    help="lbr_check_to_round = None (the reference's default): LBR also decides BEFORE the flop -- C(50, 5) run-outs per candidate range, cached per "
                         "(public history, LBR hand) and computed on request between replay rounds (DESIGN.md section 6); the line then times rounds AND requests")
    ap.add_argument("--cpu-hands", type=int, default=200, help="hands of the host LocalLBRWorker timed as the baseline (0 = skip)")
    args = ap.parse_args()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import subprocess

        import bench
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(bench.free_port()), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))))
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    torch.cuda.set_device(local_rank)
    from pokerrl_amd import _native
    _native.require_device()
    _native.set_device(local_rank)  # the library allocates on this process's GPU
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from pokerrl_amd.eval.lbr import BatchedLBR, LBRArgs, LocalLBRWorker
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game.games import DiscretizedNLHoldem, Flop5Holdem
    from pokerrl_amd.game.Poker import Poker
    from pokerrl_amd.game.wrappers import HistoryEnvBuilder
    from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase

    fhp = args.game == "Flop5Holdem"
    assert fhp or args.game == "DiscretizedNLHoldem"
    assert args.agent != "table" or fhp, "--agent table: the whole-game solution of Flop5Holdem (--game Flop5Holdem)"
    if fhp:  # the fixed-limit env with pot-size raises (games.py:222-254); LBR acts from the flop on: the board is complete there
        env_args = Flop5Holdem.ARGS_CLS(n_seats=2)
        lbr_args = LBRArgs(n_lbr_hands_per_seat=args.hands, lbr_check_to_round=None if args.lbr_from_the_start else Poker.FLOP)
    else:
        env_args = DiscretizedNLHoldem.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bet_sets.B_5)
        lbr_args = LBRArgs(lbr_bet_set=bet_sets.OFF_TREE_11, n_lbr_hands_per_seat=args.hands, lbr_check_to_round=None if args.lbr_from_the_start else Poker.TURN)
    game_cls = Flop5Holdem if fhp else DiscretizedNLHoldem
    t_prof = TrainingProfileBase(
        name="lbr_bench", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9,
        game_cls=game_cls, env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH",), eval_stack_sizes=None,
        module_args={"env": env_args, "lbr": lbr_args}, path_data=tempfile.mkdtemp())
    table, solve = None, None
    if args.agent == "table":
        from pokerrl_amd.game import board_enum
        from pokerrl_amd.rl.tabular_agent import PolicyTable
        t0 = time.perf_counter()
        reps, mult = board_enum.single_deal_board_classes(Flop5Holdem)
        tree = _native.NativeTree.for_game(Flop5Holdem, 20000, bet_sets.POT_ONLY, reps)
        solver = _native.NativeSolver(tree, "plus", 0, engine="fused", board_mult=mult, symmetrize=True)
        t1 = time.perf_counter()
        solver.iterations(args.solver_iters)
        solver.sync()
        t2 = time.perf_counter()
        expl = [float(x) * float(Flop5Holdem.EV_NORMALIZER) for x in solver.eval_avg()]
        t3 = time.perf_counter()
        table = PolicyTable.from_solver(solver)
        t4 = time.perf_counter()
        solve = {"cfr_plus_iterations": args.solver_iters, "suit_classes": int(len(reps)), "boards_represented": int(mult.sum()),
                 "tree_and_solver_build_s": t1 - t0, "iterations_s": t2 - t1, "average_strategy_exploitability_mbb_per_g": expl,
                 "average_strategy_exploitability_mean_mbb_per_g": sum(expl) / 2.0, "evaluation_s": t3 - t2, "policy_table_build_s": t4 - t3,
                 "policy_table_rows": table.n_rows, "policy_table_gb": table.n_rows * table.n_actions * table.range_size * 4 / 1e9,
                 "solver_gb": float(solver.get("bytes_allocated")[0]) / 1e9}
        del solver  # the table is a copy: the solver's 30 GB are not needed any more
    b = BatchedLBR(t_prof, agent_kind=args.agent, agent_seed=7, table=table)
    if not args.no_warmup:
        b.run(agent_seat_id=0, n_hands=4096, deck_seed=99)  # warm-up

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    barrier()
    t0 = time.perf_counter()
    res, stats = [], []
    for seat in (0, 1):
        res.append(b.run_sharded(seat, args.hands, deck_seed=seat))
        stats.append(dict(b.last_stats))
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n = sum(r[2] for r in res)
    # both seats pooled (LocalLBRMaster.py:61-69 concatenates the seats' scores before _get_95confidence)
    mean = sum(r[0] * r[2] for r in res) / n
    dev_s = sum(s["device_ms"] for s in stats) * 1e-3
    out = {"metric": "LBR hands/s (%s, batched rollouts on the GPU)" % args.game, "value": n / dt, "unit": "hands/s", "n_gpus": world,
           "steps": 1, "warmup": 1, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32",
           "data": "synthetic",
           "config": {"workload": ("LBR vs the WHOLE-GAME CFR+ average strategy of Flop5Holdem (policy table in HBM, suit-canonical look-ups; LBR acts from the flop on), "
                                   if args.agent == "table" else "LBR vs a synthetic tabular agent on Flop5Holdem (LBR acts from the flop on), " if fhp else
                                   "LBR vs a synthetic tabular agent on DiscretizedNLHoldem (agent bets B_5, LBR bets OFF_TREE_11, LBR acts from the turn on), ") +
                                  "%d hands per agent seat over %d GPU(s), every hand played start to finish by one workgroup" % (args.hands, world),
                      "hands_total": int(n), "device_seconds_rank0": dev_s, "agent": args.agent,
                      "lbr_check_to_round": "None (LBR decides before the flop too: equity cache + request / replay rounds)" if args.lbr_from_the_start else ("FLOP" if fhp else "TURN"),
                      "env_steps_per_s_rank0": sum(s["env_steps"] for s in stats) / dev_s,
                      "lbr_lookaheads_per_s_rank0": sum(s["lbr_lookaheads"] for s in stats) / dev_s,
                      "hand_evals_per_s_rank0": sum(s["range_board_equities"] for s in stats) * 1326 / dev_s,
                      "lbr_winnings_mbb_per_g": mean, "lbr_winnings_per_agent_seat_mbb_per_g": [r[0] for r in res], "conf95_per_seat": [r[1] for r in res],
                      "whole_game_solve": solve}}
    # The batch kernel keeps a hand's whole state in LDS / registers: HBM moves the decks in and the winnings out (bytes per hand in the
    # tens), and nothing in it is a contraction: neither the HBM nor the MFMA roofline says anything about it. It is bound by VECTOR
    # INSTRUCTION ISSUE, integer and float32 alike, and the model counts what the kernel executes for the dominant step, the (range, board)
    # equities of LBR's look-aheads (LocalLBRWorker.py:427-512; prl_lbr_board_equity_lists), per equity:
    #   normalising sum over the R = 1326 hands: hole-card look-up, blocker test (two 64-bit shifts = 4 ops, or, and, compare), select, add      ~10 ops
    #   sums over the n_big + n_eq <= R hands LBR beats / ties with: list read, the same blocker test, select, a CORRECTLY ROUNDED float32
    #   division (10 ops: NumPy divides every element of the range by the normalising sum before it sums), add                             ~24 ops
    # = 10 R + 24 * 0.93 R ~ 43 000 lane operations (0.93: the share of the range LBR beats or ties on an average turn, from the kernel's own
    # class counts), against the peak of one operation per lane and clock: 256 CUs x 4 SIMDs x 32 lanes x 2.4 GHz = 78.6 T lane-ops/s (the
    # FP32 vector peak of MI355X_MICROARCH.md, 157.3 TFLOP/s, counts an FMA as two). Hand evaluations: the reference's rollout manager applies
    # the win / tie lists of the FIRST enumerated board to every board (LocalLBRWorker.py:470,509-510), so a look-ahead ranks the 1326 hands
    # ONCE: rank evaluations = look-aheads x (R + 1), not equities x R.
    R_ = 1326
    ops_eq = 10.0 * R_ + 24.0 * 0.93 * R_
    n_eq_tot = sum(s["range_board_equities"] for s in stats)
    achieved = n_eq_tot * ops_eq / dev_s / 1e12
    peak = 256 * 4 * 32 * 2.4e9 / 1e12
    out["config"]["hand_rank_evaluations_per_s_rank0"] = sum(s["lbr_lookaheads"] for s in stats) * (R_ + 1) / dev_s
    del out["config"]["hand_evals_per_s_rank0"]
    # HEADLINE of this object since round 5: the MEASURED vector-issue occupancy of the kernel (SQ counters) -- `frac` is that; the operation-count
    # model of rounds 3-4 (a yardstick chosen by the author, which the kernel beats by executing fewer operations) stays beside it as `model_*`.
    # (MI355X_MICROARCH.md: a wave64 VALU instruction issues over 2 cycles on CDNA4's 32-lane SIMDs -- `peak` above is exactly that rate.)
    # The occupancy comes from a FILE written by a profiling run of this bench (scripts/gpu_r6_lbr.sh -> scripts/lbr_counters.py -> profiles/lbr_counters.json:
    # rocprofv3 --pmc SQ_INSTS_VALU ... and the --stats launch duration of the same command), not from a literal: no file, no figure.
    ctr = None
    try:
        with open(os.path.join(ROOT, "profiles", "lbr_counters.json")) as f:
            ctr = json.load(f)
        valu, k_us = float(ctr["counters"]["SQ_INSTS_VALU"]), float(ctr["kernel_us_mean"])
        VALU_BUSY = valu * 2.0 / (1024 * k_us * 1e-6 * 2.4e9)
    except (OSError, ValueError, KeyError, TypeError):
        ctr, VALU_BUSY = None, None
    cnt = (ctr or {}).get("counters", {})
    ratio = lambda a, b: (cnt[a] / cnt[b]) if (a in cnt and b in cnt and cnt[b]) else None  # noqa: E731
    applies = ctr is not None and not fhp and args.agent != "table"  # the counters are those of config 5's kernel and workload
    out["roofline"] = {"bound": "valu-issue (int + fp32)", "achieved": VALU_BUSY * peak if (applies and VALU_BUSY) else None, "peak": peak, "unit": "T lane-ops/s",
                       "frac": VALU_BUSY if applies else None, "traffic": None,
                       "frac_is": "SQ_INSTS_VALU x 2 clocks (wave64 on a 32-lane SIMD) / (1024 SIMDs x kernel clocks): the share of the vector ALUs' issue slots the "
                                  "kernel fills, MEASURED by the profiling run named in `counters_source` -- the kernel is bound by dependent chains (LDS gathers -> "
                                  "compare -> divide -> add) and barriers, not by issue" + ("" if applies else "; not quoted for this workload (the counters are config 5's)"),
                       "counters_source": None if ctr is None else "profiles/lbr_counters.json (checkpoint %s: rocprofv3 --pmc / --stats -- %s; %d hands per launch, %.2f ms per launch)"
                                          % (ctr.get("checkpoint"), ctr.get("command"), ctr.get("hands_per_launch") or 0, float(ctr.get("kernel_us_mean", 0.0)) / 1e3),
                       "counters_kernel": (ctr or {}).get("kernel"),
                       "sq_wait_any_over_wave_cycles": ratio("SQ_WAIT_ANY", "SQ_WAVE_CYCLES"),
                       "lds_bank_conflict_over_idx_active": ratio("SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
                       "model_achieved": achieved, "model_frac": achieved / peak,
                       "kernel": "prl_k_lbr_batch<%s>" % ("true" if args.agent == "table" else "false"), "lane_ops_per_range_board_equity": ops_eq,
                       "range_board_equities_per_hand": n_eq_tot / max(float(n), 1.0), "range_board_equities_per_s_rank0": n_eq_tot / dev_s,
                       "note": "not an HBM- or MFMA-bound kernel: state on chip, no contraction. Modelled: the (range, board) equities only (the betting "
                               "engine, the agent's draws and the range updates are not counted: the model is a lower bound of the work done). The "
                               "operation counts are ALGORITHMIC ones, read off round 3's kernel (generic float32 division, 64-bit blocker test) and "
                               "kept as the yardstick; the present kernel executes fewer per equity (shared-reciprocal division, byte compares), so "
                               "'model_frac' is modelled work delivered per peak; 'frac' is what the vector ALUs were actually busy with"}
    if rank == 0 and args.cpu_hands > 0 and world == 1 and not fhp:
        # cpu_baseline: the SAME episode loop (pokerrl_amd.eval.lbr.LocalLBRWorker = the reference's LocalLBRWorker.run) with the check-down
        # equity of every decision computed ON THE HOST by the NumPy restatement of the reference's rollout manager (oracle/lbr.py, pinned to
        # the reference bit for bit) -- no device call in the timed region. (Round 3 timed the host worker with its equity queries on the GPU
        # and called that a CPU baseline; that number is reported beside it as what it is.)
        import oracle
        from oracle.lbr import checkdown_equity
        from pokerrl_amd.rl import hash_agent as fx
        from pokerrl_amd.rl.base_cls.EvalAgentBase import EvalAgentBase
        from pokerrl_amd.game.Poker import Poker as _P

        def rank_fn(board):
            return oracle.rank_boards(np.array([board], dtype=np.int8))[0]

        class HostOnlyWorker(LocalLBRWorker):
            def _checkdown_equity(self, lbr_hand_2d, ranges):
                lut = self._eval_env_bldr.lut_holder
                board_1d = np.asarray(lut.get_1d_cards(self._env.board))
                dealt = [int(c) for c in board_1d if c != _P.CARD_NOT_DEALT_TOKEN_1D]
                hand_1d = [int(c) for c in lut.get_1d_cards(cards_2d=lbr_hand_2d)]
                return np.array([checkdown_equity(rank_fn, 2, 52, 5, dealt, hand_1d, r) for r in np.asarray(ranges, np.float32)], np.float32)

        agent_cls = fx.make_agent_cls(EvalAgentBase, seed=7)
        w = HostOnlyWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=agent_cls)
        np.random.seed(0)
        t1 = time.perf_counter()
        n_host = max(8, args.cpu_hands // 4)
        w.run(agent_seat_id=0, n_iterations=n_host, mode="HASH", stack_size=[20000, 20000])
        dth = time.perf_counter() - t1
        wd = LocalLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=agent_cls)
        np.random.seed(0)
        t1 = time.perf_counter()
        wd.run(agent_seat_id=0, n_iterations=args.cpu_hands, mode="HASH", stack_size=[20000, 20000])
        dtc = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": n_host / dth, "unit": "hands/s", "cores": 1, "kind": "port",
                               "sample": "the reference's LBR episode loop (pokerrl_amd.eval.lbr.LocalLBRWorker) with every check-down equity on the host "
                                         "(oracle/lbr.py, NumPy, the reference's rollout manager restated), %d hands, same agent and bet sets, %.1f s" % (n_host, dth),
                               "host_worker_with_device_equity_hands_per_s": args.cpu_hands / dtc,
                               "host_worker_with_device_equity_sample": "the same loop with prl_lbr_checkdown_equity (one GPU call per LBR decision), %d hands, %.1f s" % (args.cpu_hands, dtc),
                               # the reference's own worker needs /root/reference, which does not travel to the GPU box: timed by scripts/time_reference.py
                               "reference_python_hands_per_s": bench_ref.figure("local_lbr_worker_run", "DiscretizedNLHoldem_TURN_OFF_TREE_11", "hands_per_s"),
                               "reference_timing_source": bench_ref.SOURCE, "reference_timing_host": bench_ref.host()}
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
