"""
bench_env.py -- batched PokerEnv.step on the device (BASELINE.json config 5's env part; secondary to bench.py): 2^20 heads-up
DiscretizedNLHoldem envs (blinds 50/100, stacks 20000, bet set B_5), uniform-random legal play, finished hands restart.

    python bench_env.py [--envs N] [--steps K] [--warmup W] [--game DiscretizedNLHoldem] [--no-cpu-baseline]

A "step" of the JSON line is one launch of the step kernel over the whole batch = N env steps with the state in HBM between the
launches (include/pokerrl_hip.h section 4b: 13 int32 words in, 13 out per env and step): value = env-steps/s.
roofline: HBM, algorithmic bytes = 104 B per env step. `config.rollout_env_steps_per_s` is the same play with the state held in
registers for 64 steps per launch (no HBM round trip per step: the integer ALU figure).
cpu_baseline: the same hands (same counter-based draws) played by the same C++ engine on ONE host core (kind "port"; the
reference's Python PokerEnv.step runs 16.1 k steps/s on one core).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--game", default="DiscretizedNLHoldem")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    _native.require_device()
    cls = getattr(G, args.game)
    stack = {"StandardLeduc": 13, "BigLeduc": 100}.get(args.game, 20000)
    ea = cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bet_sets.B_5)
    game = cls.native_game(ea)
    b = _native.NativeEnvBatch(game, args.envs)
    b.random_steps(args.warmup, 1)
    t0 = time.perf_counter()
    steps, hands, pots, ms = b.random_steps(args.steps, 2)
    dt = time.perf_counter() - t0
    r_steps, r_hands, _p, r_ms = b.random_rollout(64, 3)
    bytes_step = 104.0
    achieved = steps * bytes_step / (ms * 1e-3) / 1e9
    out = {"metric": "batched PokerEnv.step env-steps/s", "value": steps / dt, "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32", "data": "synthetic",
           "config": {"workload": "%d heads-up %s envs (stacks %d, bet set B_5), uniform-random legal play, one step per env and launch, "
                                  "state as struct-of-arrays in HBM" % (args.envs, args.game, stack),
                      "hands_finished": hands, "mean_pot": pots / max(hands, 1), "rollout_env_steps_per_s": r_steps / (r_ms * 1e-3),
                      "rollout_hands_per_s": r_hands / (r_ms * 1e-3)},
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                        "kernel": "prl_k_eb_random_step", "kernel_ms_per_launch": ms / args.steps, "bytes_per_env_step_algorithmic": bytes_step}}
    if not args.no_cpu_baseline:
        n_cpu, k_cpu = 4096, 512
        t0 = time.perf_counter()
        s3 = _native.env_random_rollout_host(game, n_cpu, k_cpu, 2)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": s3[0] / dtc, "unit": "env-steps/s", "cores": 1, "kind": "port",
                               "sample": "%d envs x %d steps, the same engine (csrc/prl_env.h) and draws on one host core, %.1f s; the reference's "
                                         "Python PokerEnv.step: 16.1 k steps/s per core (BASELINE.md)" % (n_cpu, k_cpu, dtc)}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
