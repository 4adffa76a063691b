"""
Times the REFERENCE's own Python path (imported from /root/reference through tests/golden/ref_harness.py with its test-only shims) on THIS
container's host cores and writes profiles/reference_cpu.json -- the file every bench*.py quotes its `reference_python_*` figures from (no
hard-coded reference number in any bench). The reference does not travel to the GPU box, so this runs here, single process, single thread
(OMP_NUM_THREADS=1: the reference has no intra-op parallelism on this path; SURVEY.md section 8d "CPU baseline timing").

    python scripts/time_reference.py [--quick]          (about 4 minutes; --quick: about one)

Timed (reference file:line):
  cfr.iteration()               StandardLeduc / DiscretizedNLLeduc+POT_ONLY / BigLeduc     PokerRL/cfr/_CFRBase.py:122-134
  LocalBRMaster.evaluate        StandardLeduc, DiscretizedNLLeduc (hash fixture agent)       PokerRL/eval/br/LocalBRMaster.py:41-80
  LocalLBRWorker.run            StandardLeduc, DiscretizedNLHoldem (check_to_round = TURN)   PokerRL/eval/lbr/LocalLBRWorker.py:35-308
  env.step, random play         StandardLeduc, DiscretizedNLHoldem                           PokerRL/game/_/rl_env/base/PokerEnv.py:681-789
  batched 7-card evaluator      20000 boards x 1326 hands (lib_hand_eval.so)                 PokerRL/game/_/cpp_wrappers/CppHandeval.py:45-65
"""
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import ref_harness  # noqa: E402

np = ref_harness.setup()
QUICK = "--quick" in sys.argv


def timed(fn, n, warm=1):
    for _ in range(warm):
        fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


def cfr_rows():
    from PokerRL.cfr.CFRPlus import CFRPlus
    from PokerRL.cfr.LinearCFR import LinearCFR
    from PokerRL.cfr.VanillaCFR import VanillaCFR
    from PokerRL.game import bet_sets
    from PokerRL.game.games import BigLeduc, DiscretizedNLLeduc, StandardLeduc
    from PokerRL.rl.base_cls.workers.ChiefBase import ChiefBase
    out = {}
    cases = [("StandardLeduc", StandardLeduc, 13, bet_sets.POT_ONLY, (("CFRPlus", CFRPlus), ("VanillaCFR", VanillaCFR), ("LinearCFR", LinearCFR)), 3 if QUICK else 10),
             ("DiscretizedNLLeduc_POT_ONLY", DiscretizedNLLeduc, 20000, bet_sets.POT_ONLY, (("CFRPlus", CFRPlus),), 2 if QUICK else 6)]
    if not QUICK:
        cases.append(("BigLeduc", BigLeduc, 100, bet_sets.POT_ONLY, (("CFRPlus", CFRPlus),), 2))
    for gname, cls, stack, bets, algos, n in cases:
        for aname, algo in algos:
            kw = dict(name="t", game_cls=cls, agent_bet_set=bets, chief_handle=ChiefBase(t_prof=None), starting_stack_sizes=[stack])
            if aname == "CFRPlus":
                kw["delay"] = 0
            t0 = time.perf_counter()
            cfr = algo(**kw)
            build_s = time.perf_counter() - t0
            n_nodes = cfr._trees[0].n_nodes + 1  # the reference's counter leaves the root out (PublicTree.py:60,163)
            s = timed(cfr.iteration, n, warm=0 if gname == "BigLeduc" else 1)
            out["%s/%s" % (gname, aname)] = {"seconds_per_iteration": s, "iterations_per_s": 1.0 / s, "nodes": n_nodes, "node_updates_per_s": n_nodes / s,
                                            "construct_and_reset_s": build_s, "iterations_timed": n}
            print("cfr", gname, aname, "%.3f s/it  %.0f node-updates/s" % (s, n_nodes / s), flush=True)
    return out


def t_prof_for(game_cls, bets, lbr_args=None):
    from PokerRL.game.wrappers import HistoryEnvBuilder
    from PokerRL.rl.base_cls.TrainingProfileBase import TrainingProfileBase
    margs = {"env": game_cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bets) if bets is not None else game_cls.ARGS_CLS(n_seats=2)}
    if lbr_args is not None:
        margs["lbr"] = lbr_args
    return TrainingProfileBase(name="t", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=game_cls,
                               env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH",), eval_stack_sizes=None, module_args=margs,
                               path_data=os.environ["HOME"])


def br_rows():
    from PokerRL.eval.br.LocalBRMaster import LocalBRMaster
    from PokerRL.game import bet_sets
    from PokerRL.game.games import DiscretizedNLLeduc, StandardLeduc
    from PokerRL.rl.base_cls.EvalAgentBase import EvalAgentBase
    from PokerRL.rl.base_cls.workers.ChiefBase import ChiefBase
    from pokerrl_amd.rl import hash_agent

    class Chief(ChiefBase):
        def pull_current_eval_strategy(self, last):
            return None, last

    out = {}
    for gname, cls, bets in (("StandardLeduc", StandardLeduc, None), ("DiscretizedNLLeduc_POT_ONLY", DiscretizedNLLeduc, bet_sets.POT_ONLY)):
        prof = t_prof_for(cls, bets)
        br = LocalBRMaster(t_prof=prof, chief_handle=Chief(prof), eval_agent_cls=hash_agent.make_agent_cls(EvalAgentBase, seed=7))
        br.update_weights()
        s = timed(lambda: br.evaluate(iter_nr=0), 2 if QUICK else 5)
        out[gname] = {"seconds_per_evaluation": s, "evaluations_per_s": 1.0 / s}
        print("br", gname, "%.3f s per evaluation" % s, flush=True)
    return out


def lbr_rows():
    from PokerRL.eval.lbr.LBRArgs import LBRArgs
    from PokerRL.eval.lbr.LocalLBRWorker import LocalLBRWorker
    from PokerRL.game import Poker, bet_sets
    from PokerRL.game.games import DiscretizedNLHoldem, StandardLeduc
    from PokerRL.rl.base_cls.EvalAgentBase import EvalAgentBase
    from pokerrl_amd.rl import hash_agent
    out = {}
    cases = (("StandardLeduc", StandardLeduc, None, LBRArgs(n_lbr_hands_per_seat=300, lbr_check_to_round=None), 100 if QUICK else 300),
             ("DiscretizedNLHoldem_TURN_OFF_TREE_11", DiscretizedNLHoldem, bet_sets.B_5,
              LBRArgs(lbr_bet_set=bet_sets.OFF_TREE_11, n_lbr_hands_per_seat=150, lbr_check_to_round=Poker.TURN), 8 if QUICK else 32))
    for gname, cls, bets, args, n in cases:
        w = LocalLBRWorker(t_prof=t_prof_for(cls, bets, args), chief_handle=None, eval_agent_cls=hash_agent.make_agent_cls(EvalAgentBase, seed=7))
        np.random.seed(1)
        t0 = time.perf_counter()
        for seat in (0, 1):
            w.run(agent_seat_id=seat, n_iterations=n, mode="HASH", stack_size=[cls.DEFAULT_STACK_SIZE] * 2)
        s = time.perf_counter() - t0
        out[gname] = {"hands_per_s": 2 * n / s, "hands_timed": 2 * n, "agent": "hash fixture agent (pokerrl_amd/rl/hash_agent.py bound to the reference's EvalAgentBase)"}
        print("lbr", gname, "%.1f hands/s" % (2 * n / s), flush=True)
    return out


def env_rows():
    from PokerRL.game import bet_sets
    from PokerRL.game.games import DiscretizedNLHoldem, StandardLeduc
    out = {}
    for gname, cls, bets in (("StandardLeduc", StandardLeduc, None), ("DiscretizedNLHoldem_B_5", DiscretizedNLHoldem, bet_sets.B_5)):
        args = cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bets) if bets is not None else cls.ARGS_CLS(n_seats=2)
        from PokerRL.game._.look_up_table import LutHolderHoldem, LutHolderLeduc  # noqa: F401
        env = cls(env_args=args, lut_holder=cls.get_lut_holder(), is_evaluating=True)
        rng = np.random.RandomState(0)
        n_steps, n_eps, target = 0, 0, (5000 if QUICK else 40000)
        t0 = time.perf_counter()
        while n_steps < target:
            env.reset()
            done = False
            while not done:
                legal = env.get_legal_actions()
                _o, _r, done, _i = env.step(legal[rng.randint(len(legal))])
                n_steps += 1
            n_eps += 1
        s = time.perf_counter() - t0
        out[gname] = {"steps_per_s": n_steps / s, "episodes_per_s": n_eps / s, "steps_timed": n_steps}
        print("env", gname, "%.0f steps/s" % (n_steps / s), flush=True)
    return out


def handeval_rows():
    from PokerRL.game._.cpp_wrappers.CppHandeval import CppHandeval
    ev = CppHandeval()
    rng = np.random.RandomState(0)
    n = 4000 if QUICK else 20000
    boards_1d = np.stack([rng.choice(52, 5, replace=False) for _ in range(n)]).astype(np.int8)
    from PokerRL.game.games import DiscretizedNLHoldem
    lut = DiscretizedNLHoldem.get_lut_holder()
    s = timed(lambda: ev.get_hand_rank_all_hands_on_given_boards_52_holdem(boards_1d=boards_1d, lut_holder=lut), 1 if QUICK else 2)
    out = {"batched": {"evals_per_s": n * 1326 / s, "boards": n, "seconds": s}}
    hand_2d, board_2d = lut.get_2d_cards(np.array([0, 13], np.int8)), lut.get_2d_cards(boards_1d[0])
    m = 2000 if QUICK else 20000
    t0 = time.perf_counter()
    for _ in range(m):
        ev.get_hand_rank_52_holdem(hand_2d=hand_2d, board_2d=board_2d)
    out["single_call"] = {"evals_per_s": m / (time.perf_counter() - t0)}
    print("handeval batched %.3g evals/s, single call %.3g" % (out["batched"]["evals_per_s"], out["single_call"]["evals_per_s"]), flush=True)
    return out


if __name__ == "__main__":
    res = {
        "what": "the reference's own Python path (/root/reference, imported with the shims of tests/golden/ref_harness.py), one process, one thread",
        "generator": "scripts/time_reference.py" + (" --quick" if QUICK else ""),
        "host": {"cpu_count": os.cpu_count(), "machine": platform.machine(), "python": platform.python_version(), "numpy": np.__version__,
                 "omp_num_threads": os.environ.get("OMP_NUM_THREADS"), "note": "this container's host cores (no GPU here); the GPU box has other cores"},
        "timestamp": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()),
        "cfr_iteration": cfr_rows(), "local_br_master_evaluate": br_rows(), "local_lbr_worker_run": lbr_rows(), "env_step_random_play": env_rows(),
        "hand_evaluator": handeval_rows(),
    }
    path = os.path.join(ROOT, "profiles", "reference_cpu.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1)
    print("wrote", path)
