"""
bench_env.py -- batched PokerEnv.step on the device (BASELINE.json config 5's env part; secondary to bench.py): 2^20 heads-up
DiscretizedNLHoldem envs (blinds 50/100, stacks 20000, bet set B_5), uniform-random legal play, finished hands are dealt again.

    python bench_env.py [--envs N] [--steps K] [--warmup W] [--game DiscretizedNLHoldem] [--betting-only] [--no-cpu-baseline]

A "step" of the JSON line is one launch of the WHOLE PokerEnv.step over the batch (include/pokerrl_hip.h section 4b: cards, betting, showdown
ranks, payout, rewards, the observation vector) = N env steps with everything in HBM between the launches -- the shape an agent-driven
rollout has: per env and step 13 state words in and out, the observation vector (109 floats for hold'em), two float64 rewards, the done flag
and the env's cards. value = env-steps/s. roofline: HBM, algorithmic bytes per env step = 104 + 4 obs_dim + 16 + 1 + n_cards_of_the_env.
--betting-only: the public betting state machine alone (104 B per env step; round 2's line).
config.rollout_*: the same play with the state held in registers (64 steps per launch, no HBM round trip per step; whole hands incl. dealing,
showdown ranks and payouts): the integer-ALU figure.
cpu_baseline: the same hands (same counter-based draws and decks) played by the same C++ engine on ONE host core (kind "port"); the
reference's own Python PokerEnv.step loop is timed by scripts/time_reference.py (profiles/reference_cpu.json, quoted in the line) -- it does not travel to
the GPU box, so it cannot be timed in the same run.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import bench_ref  # noqa: E402

HBM_PEAK_GBPS = 8000.0
# HBM bytes per env step of prl_k_ebf_random_step from the PMC counters: profiles/env_counters.json, written by scripts/env_counters.py from
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this bench (separate runs; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).
# Round 4's figure (profiles/r08_env_pmc.txt) was 583.7 MB per launch of 2^20 envs against 593.5 MB algorithmic.
PMC_COUNTERS = os.path.join(ROOT, "profiles", "env_counters.json")


def pmc_traffic_per_env_step():
    try:
        d = json.load(open(PMC_COUNTERS))
        return float(d["hbm_bytes_per_env_step"]), "profiles/env_counters.json (%s)" % d["source"]
    except (OSError, KeyError, ValueError):
        return None, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=1 << 20)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--game", default="DiscretizedNLHoldem")
    ap.add_argument("--betting-only", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    _native.require_device()
    cls = getattr(G, args.game)
    stack = {"StandardLeduc": 13, "BigLeduc": 100}.get(args.game, 20000)
    ea = cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bet_sets.B_5)
    game, rules = cls.native_game(ea), cls.native_rules()
    full = not args.betting_only
    b = _native.NativeEnvBatch.with_cards(game, rules, args.envs, deck_seed=11) if full else _native.NativeEnvBatch(game, args.envs)
    run = b.random_steps_full if full else b.random_steps
    run(args.warmup, 1)
    t0 = time.perf_counter()
    steps, hands, pots, ms = run(args.steps, 2)
    dt = time.perf_counter() - t0
    plain = _native.NativeEnvBatch(game, args.envs) if full else b
    r_steps, r_hands, _p, r_ms = plain.random_rollout(64, 3)
    traffic_step, traffic_src = pmc_traffic_per_env_step()
    bytes_step = (104.0 + 4.0 * b.obs_dim + 16.0 + 1.0 + b.n_deal) if full else 104.0
    achieved = steps * bytes_step / (ms * 1e-3) / 1e9
    cfg = {"workload": "%d heads-up %s envs (stacks %d, bet set B_5), uniform-random legal play, one %s per env and launch, everything in HBM between "
                       "the launches" % (args.envs, args.game, stack, "whole PokerEnv.step (cards, payouts, rewards, observation vector)" if full else "betting step"),
           "hands_finished": hands, "mean_pot": pots / max(hands, 1), "rollout_betting_only_env_steps_per_s": r_steps / (r_ms * 1e-3),
           "rollout_betting_only_hands_per_s": r_hands / (r_ms * 1e-3)}
    if full:
        f_steps, f_hands, f_show, _c, f_ms = b.random_rollout_full(64, 3)
        cfg.update({"obs_dim": b.obs_dim, "rollout_whole_hands_env_steps_per_s": f_steps / (f_ms * 1e-3), "rollout_whole_hands_per_s": f_hands / (f_ms * 1e-3),
                    "rollout_showdowns_per_s": f_show / (f_ms * 1e-3)})
    out = {"metric": "batched PokerEnv.step env-steps/s", "value": steps / dt, "unit": "env-steps/s", "n_gpus": 1, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "int32", "data": "synthetic", "build_flavor": _native.build_flavor(), "config": cfg,
           "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                        "traffic": traffic_step * args.envs if (traffic_step is not None and full and args.game == "DiscretizedNLHoldem") else None,
                        "traffic_source": traffic_src,
                        "kernel": "prl_k_ebf_random_step" if full else "prl_k_eb_random_step", "kernel_ms_per_launch": ms / args.steps,
                        "bytes_per_env_step_algorithmic": bytes_step}}
    if not args.no_cpu_baseline:
        n_cpu, k_cpu = 65536, 2048
        t0 = time.perf_counter()
        s3 = _native.env_random_rollout_full_host(game, rules, n_cpu, k_cpu, 2, deck_seed=11) if full else _native.env_random_rollout_host(game, n_cpu, k_cpu, 2)
        dtc = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": s3[0] / dtc, "unit": "env-steps/s", "cores": 1, "kind": "port",
                               "sample": "%d envs x %d steps, the same engine (csrc/prl_env.h%s) and draws on one host core, %.1f s; no observation vectors on "
                                         "the host leg" % (n_cpu, k_cpu, " + dealing, showdown ranks, payouts" if full else "", dtc),
                               # the reference's own env.step loop (random play, the same game), timed by scripts/time_reference.py
                               "reference_python_steps_per_s": bench_ref.figure("env_step_random_play", "DiscretizedNLHoldem_B_5" if args.game == "DiscretizedNLHoldem" else args.game, "steps_per_s"),
                               "reference_timing_source": bench_ref.SOURCE, "reference_timing_host": bench_ref.host()}
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
