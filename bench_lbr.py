"""
bench_lbr.py -- LBR hands/s with the device-resident batched engine (BASELINE.json config 5; secondary metric, bench.py is the
driver's contract). DiscretizedNLHoldem (blinds 50/100, stacks 20000), agent bet set B_5, LBR bet set OFF_TREE_11,
lbr_check_to_round = TURN, synthetic hash agent, counter-based decks keyed by (seed, hand id).

    python bench_lbr.py [--hands N] [--agent hash|uniform] [--cpu-hands M]
N > 1 GPUs: hands are independent; rank r plays hands [r * N, (r + 1) * N) of the same counter-based deck stream and the
(sum, sum of squares, n) are all-reduced (torch.distributed, nccl) -- launch with torch.distributed.run.
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hands", type=int, default=1 << 18, help="hands per seat per GPU")
    ap.add_argument("--agent", default="hash")
    ap.add_argument("--cpu-hands", type=int, default=40, help="hands of the host LocalLBRWorker timed as the baseline (0 = skip)")
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")))
        from pokerrl_amd import _native as _nat
        _nat.set_device(int(os.environ.get("LOCAL_RANK", "0")))  # the library allocates on this process's GPU
        dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0"))))
    from pokerrl_amd.eval.lbr import BatchedLBR, LBRArgs, LocalLBRWorker
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game.games import DiscretizedNLHoldem
    from pokerrl_amd.game.Poker import Poker
    from pokerrl_amd.game.wrappers import HistoryEnvBuilder
    from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase

    t_prof = TrainingProfileBase(
        name="lbr_bench", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9,
        game_cls=DiscretizedNLHoldem, env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH",), eval_stack_sizes=None,
        module_args={"env": DiscretizedNLHoldem.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bet_sets.B_5),
                     "lbr": LBRArgs(lbr_bet_set=bet_sets.OFF_TREE_11, n_lbr_hands_per_seat=args.hands, lbr_check_to_round=Poker.TURN)},
        path_data=tempfile.mkdtemp())
    b = BatchedLBR(t_prof, agent_kind=args.agent, agent_seed=7)
    b.run(agent_seat_id=0, n_hands=min(args.hands, 4096), deck_seed=1)  # warm-up
    t0 = time.perf_counter()
    scores, stats = [], []
    for seat in (0, 1):
        scores.append(b.run(agent_seat_id=seat, n_hands=args.hands, deck_seed=seat, first_hand=rank * args.hands))
        stats.append(dict(b.last_stats))
    dt = time.perf_counter() - t0
    x = np.concatenate(scores).astype(np.float64)
    agg = np.array([x.sum(), (x * x).sum(), x.shape[0], dt], dtype=np.float64)
    if dist is not None:
        t = torch.tensor(agg[:3], device="cuda")
        dist.all_reduce(t)
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        agg[:3], dt = t.cpu().numpy(), float(tt.item())
    n = agg[2]
    mean = agg[0] / n
    sd = np.sqrt(max(agg[1] / n - mean * mean, 0.0))
    dev_s = sum(s["device_ms"] for s in stats) * 1e-3
    out = {"metric": "LBR hands/s (DiscretizedNLHoldem, batched rollouts on the GPU)", "value": n / dt, "unit": "hands/s", "n_gpus": world,
           "hands_total": int(n), "seconds": dt, "device_seconds_rank0": dev_s,
           "env_steps_per_s": sum(s["env_steps"] for s in stats) / dev_s, "lbr_lookaheads_per_s": sum(s["lbr_lookaheads"] for s in stats) / dev_s,
           "hand_evals_per_s": sum(s["range_board_equities"] for s in stats) * 1326 / dev_s,
           "lbr_winnings_mbb_per_g": mean, "conf95": 1.96 * sd / np.sqrt(n), "agent": args.agent, "data": "synthetic"}
    if rank == 0 and args.cpu_hands > 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import lbr_fixture_agent as fx
        from pokerrl_amd.rl.base_cls.EvalAgentBase import EvalAgentBase
        w = LocalLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=7))
        np.random.seed(0)
        t1 = time.perf_counter()
        w.run(agent_seat_id=0, n_iterations=args.cpu_hands, mode="HASH", stack_size=[20000, 20000])
        out["host_worker_hands_per_s"] = args.cpu_hands / (time.perf_counter() - t1)
        out["host_worker_note"] = "LocalLBRWorker drop-in (Python episode loop, one GPU equity call per LBR decision), %d hands" % args.cpu_hands
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
