"""Golden vectors for head-to-head evaluation: the REFERENCE's LocalHead2HeadMaster with the two
modes of the fixture agent (tests/lbr_fixture_agent.py) against each other -> per-hand winnings of _run_eval's loop and the
scalars evaluate() logs. Usage: python tests/golden/make_h2h_golden.py -> tests/golden/h2h_*.npz"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.eval.head_to_head.H2HArgs import H2HArgs  # noqa: E402
from PokerRL.eval.head_to_head.LocalHead2HeadMaster import LocalHead2HeadMaster  # noqa: E402
from PokerRL.game import bet_sets  # noqa: E402
from PokerRL.game.games import DiscretizedNLHoldem, DiscretizedNLLeduc, StandardLeduc  # noqa: E402
from PokerRL.game.wrappers import HistoryEnvBuilder  # noqa: E402
from PokerRL.rl.base_cls.EvalAgentBase import EvalAgentBase  # noqa: E402
from PokerRL.rl.base_cls.TrainingProfileBase import TrainingProfileBase  # noqa: E402

import lbr_fixture_agent as fx  # noqa: E402


class Chief:
    def __init__(self):
        self.names, self.log = [], []

    def create_experiment(self, name):
        self.names.append(name)
        return name

    def add_scalar(self, exp, graph, step, value):
        self.log.append([exp, graph, int(step), float(value)])


def run(tag, game_cls, bets, n_hands, np_seed):
    t_prof = TrainingProfileBase(
        name="h2h", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9,
        game_cls=game_cls, env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH", "HASH2"), eval_stack_sizes=None,
        module_args={"env": game_cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bets) if bets is not None
                     else game_cls.ARGS_CLS(n_seats=2), "h2h": H2HArgs(n_hands=n_hands)}, path_data=os.environ["HOME"])
    chief = Chief()
    m = LocalHead2HeadMaster(t_prof=t_prof, chief_handle=chief, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=11))
    m.set_modes(["HASH", "HASH2"])
    captured = []
    orig = m._get_95confidence
    m._get_95confidence = lambda scores: (captured.append(np.array(scores, copy=True)), orig(scores))[1]
    np.random.seed(np_seed)
    m.evaluate(iter_nr=3)
    out = {"winnings": captured[0], "log": np.array(json.dumps(chief.log)), "experiments": np.array(json.dumps(chief.names)),
           "n_hands": np.int64(n_hands), "np_seed": np.int64(np_seed)}
    np.savez_compressed(os.path.join(HERE, "h2h_%s.npz" % tag), **out)
    print(tag, captured[0].shape, float(np.mean(captured[0])), chief.log[0])


if __name__ == "__main__":
    run("StandardLeduc", StandardLeduc, None, 400, 500)
    run("DiscretizedNLLeduc", DiscretizedNLLeduc, bet_sets.B_3, 300, 501)
    run("DiscretizedNLHoldem", DiscretizedNLHoldem, bet_sets.B_5, 60, 502)
