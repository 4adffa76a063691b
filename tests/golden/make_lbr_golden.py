"""Golden vectors for LBR: runs the REFERENCE's LocalLBRWorker (imported from
/root/reference with the shims of ref_harness) against the fixture agent of tests/lbr_fixture_agent.py and stores the per-hand
winnings plus every episode's deck. Usage: python tests/golden/make_lbr_golden.py  -> tests/golden/lbr_*.npz"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.eval.lbr.LBRArgs import LBRArgs  # noqa: E402
from PokerRL.eval.lbr.LocalLBRWorker import LocalLBRWorker  # noqa: E402
from PokerRL.game import Poker, bet_sets  # noqa: E402
from PokerRL.game.games import DiscretizedNLHoldem, DiscretizedNLLeduc, StandardLeduc  # noqa: E402
from PokerRL.game.wrappers import HistoryEnvBuilder  # noqa: E402
from PokerRL.rl.base_cls.EvalAgentBase import EvalAgentBase  # noqa: E402
from PokerRL.rl.base_cls.TrainingProfileBase import TrainingProfileBase  # noqa: E402

import lbr_fixture_agent as fx  # noqa: E402


def run(tag, game_cls, agent_bets, lbr_args, n_hands, np_seed):
    record = []
    t_prof = TrainingProfileBase(
        name="lbr", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9,
        game_cls=game_cls, env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH",), eval_stack_sizes=None,
        module_args={"env": game_cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=agent_bets) if agent_bets is not None
                     else game_cls.ARGS_CLS(n_seats=2), "lbr": lbr_args}, path_data=os.environ["HOME"])
    w = LocalLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=7, record=record))
    out = {}
    for seat in (0, 1):
        np.random.seed(np_seed + seat)
        n0 = len(record)
        out["winnings_agent_seat%d" % seat] = w.run(agent_seat_id=seat, n_iterations=n_hands, mode="HASH",
                                                    stack_size=[game_cls.DEFAULT_STACK_SIZE] * 2)
        decks = record[n0:]
        out["hands_agent_seat%d" % seat] = np.stack([np.stack(d["hand"]) for d in decks]).astype(np.int8)
        out["board_agent_seat%d" % seat] = np.stack([np.asarray(d["board"]) for d in decks]).astype(np.int8)
    out["n_hands"], out["np_seed"] = np.int64(n_hands), np.int64(np_seed)
    np.savez_compressed(os.path.join(HERE, "lbr_%s.npz" % tag), **out)
    print(tag, {k: (v.shape, float(np.mean(v))) for k, v in out.items() if k.startswith("winnings")})


if __name__ == "__main__":
    which = sys.argv[1:] or ["leduc", "nlleduc", "holdem", "holdem_flop"]
    if "leduc" in which:
        run("StandardLeduc", StandardLeduc, None, LBRArgs(n_lbr_hands_per_seat=300, lbr_check_to_round=None), 300, 100)
    if "nlleduc" in which:
        run("DiscretizedNLLeduc", DiscretizedNLLeduc, bet_sets.B_3, LBRArgs(lbr_bet_set=bet_sets.B_5, n_lbr_hands_per_seat=200,
                                                                            lbr_check_to_round=None), 200, 200)
    if "holdem" in which:
        run("DiscretizedNLHoldem", DiscretizedNLHoldem, bet_sets.B_5,
            LBRArgs(lbr_bet_set=bet_sets.OFF_TREE_11, n_lbr_hands_per_seat=150, lbr_check_to_round=Poker.TURN), 150, 300)
    if "holdem_flop" in which:  # LBR also acts on the flop: C(45, 2) = 990 boards per equity
        run("DiscretizedNLHoldem_flop", DiscretizedNLHoldem, bet_sets.B_3,
            LBRArgs(lbr_bet_set=bet_sets.B_5, n_lbr_hands_per_seat=8, lbr_check_to_round=Poker.FLOP), 8, 400)
