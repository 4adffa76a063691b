"""Secondary benchmark: head-to-head hands per second between two synthetic agents on
DiscretizedNLHoldem, every hand played on the GPU by one lane (pokerrl_amd.eval.head_to_head.BatchedHead2Head); the host
drop-in evaluator (LocalHead2HeadMaster, Python episode loop) on a small sample next to it.
    python bench_h2h.py [--hands 1048576]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hands", type=int, default=1 << 20, help="hands per seat assignment (the run plays 2x this)")
    ap.add_argument("--host-hands", type=int, default=100)
    args = ap.parse_args()
    from pokerrl_amd.rl import hash_agent as fx
    from pokerrl_amd import _native
    from pokerrl_amd.eval.head_to_head import BatchedHead2Head, H2HArgs, LocalHead2HeadMaster
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game.games import DiscretizedNLHoldem
    from pokerrl_amd.game.wrappers import HistoryEnvBuilder
    from pokerrl_amd.rl.base_cls.EvalAgentBase import EvalAgentBase
    from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase
    _native.require_device()
    t_prof = TrainingProfileBase(
        name="h2h", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=DiscretizedNLHoldem,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH", "HASH2"), eval_stack_sizes=None,
        module_args={"env": DiscretizedNLHoldem.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bet_sets.B_5), "h2h": H2HArgs(n_hands=args.host_hands)},
        path_data=tempfile.mkdtemp(prefix="prl_h2h_"))
    b = BatchedHead2Head(t_prof, kinds=("hash", "hash"), seeds=(11, 12))
    b.play(n_hands=4096)  # warm-up
    t0 = time.perf_counter()
    w = b.play(n_hands=args.hands, deck_seed=1)
    dt = time.perf_counter() - t0
    dev_ms = b.last_stats["device_ms"]
    steps = b.last_stats["env_steps"]

    class Chief:
        def create_experiment(self, name):
            return name

        def add_scalar(self, *a):
            pass

    m = LocalHead2HeadMaster(t_prof=t_prof, chief_handle=Chief(), eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=11))
    m.set_modes(["HASH", "HASH2"])
    np.random.seed(0)
    t1 = time.perf_counter()
    m.play(stack_size=t_prof.eval_stack_sizes[0])
    host_dt = time.perf_counter() - t1
    mean, d = float(np.mean(w)), float(1.96 * np.std(w) / np.sqrt(w.shape[0]))
    print(json.dumps({
        "metric": "head-to-head hands/s (DiscretizedNLHoldem, one GPU lane per hand)", "value": 2 * args.hands / dt, "unit": "hands/s", "n_gpus": 1,
        "hands_total": 2 * args.hands, "seconds": dt, "device_ms_last_half": dev_ms, "device_hands_per_s_last_half": args.hands / (dev_ms * 1e-3),
        "env_steps_last_half": steps, "winnings_mbb_per_g": mean, "conf95": d, "agents": "hash(11) vs hash(12)", "data": "synthetic",
        "steps": 1, "warmup": 0, "ms_per_step": dt * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "int32",
        "config": {"workload": "%d hands per seat assignment between two synthetic tabular agents, dealing / betting / showdown / payout per lane" % args.hands},
        "roofline": None, "roofline_note": "one lane plays one hand start to finish out of registers: ~17 bytes per hand reach HBM (deck seed in, winnings out); "
                                            "bound by divergent integer control flow and the two 7-card evaluations of a showdown, no byte or flop roofline applies",
        "cpu_baseline": {"value": 2 * args.host_hands / host_dt, "unit": "hands/s", "cores": 1, "kind": "port",
                         "sample": "LocalHead2HeadMaster drop-in (Python episode loop on the native-backed env), %d hands" % (2 * args.host_hands)},
        "host_evaluator_hands_per_s": 2 * args.host_hands / host_dt,
        "host_evaluator_note": "LocalHead2HeadMaster drop-in (Python episode loop on the native-backed env), %d hands" % (2 * args.host_hands)}))


if __name__ == "__main__":
    main()
