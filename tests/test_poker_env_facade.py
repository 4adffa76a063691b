"""
CPU test of the PokerEnv facade (pokerrl_amd/game/poker_env.py) against full episodes recorded from the reference env
(tests/golden/env_obs.npz): same seeded deck order, same observation vectors after every step, same terminal rewards,
same board / hole cards. Betting itself is native (test_host_golden.py::test_env_matches_reference_fuzz).
"""
import numpy as np
import pytest

from helpers import golden
from pokerrl_amd.game.wrappers import HistoryEnvBuilder
from test_host_golden import ENV_FUZZ

NAMES = ["StandardLeduc", "BigLeduc_short", "DiscretizedNLLeduc_B5_short", "LimitHoldem", "DiscretizedNLHoldem_B5",
         "DiscretizedNLHoldem_OT11_short", "Flop5Holdem"]


@pytest.mark.parametrize("name", NAMES)
def test_env_facade_episodes_match_reference(name):
    g = golden("env_obs.npz")
    obs, meta = g[name + "_obs"], g[name + "_meta"]
    cls, stack, bets = ENV_FUZZ[name]
    args = cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bets)
    env = HistoryEnvBuilder(env_cls=cls, env_args=args).get_new_env(is_evaluating=True)
    row = 0
    while row < len(meta):
        ep = int(meta[row, 0])
        np.random.seed(1000 + ep)
        o, r, done, _ = env.reset()
        assert o.dtype == np.float32 and np.array_equal(o, obs[row]), (name, ep, "reset")
        row += 1
        while not done:
            act = int(meta[row, 1])
            assert act in env.get_legal_actions()
            o, r, done, _ = env.step(act)
            assert np.array_equal(o, obs[row]), (name, ep, row)
            assert done == bool(meta[row, 2])
            assert float(r[0]) == meta[row, 3] and float(r[1]) == meta[row, 4], (name, ep, r, meta[row])
            row += 1
        cards = [int(c) for c in env.lut_holder.get_1d_cards(env.board)] + \
                [int(c) for p in range(2) for c in env.lut_holder.get_1d_cards(env.seats[p].hand)]
        assert cards == [int(c) for c in meta[row - 1, 5:5 + len(cards)]], (name, ep)


def test_env_state_dict_round_trip():
    """test/game/test_pokerEnv.py:128-156: a state restored from state_dict continues identically"""
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game.games import DiscretizedNLHoldem
    args = DiscretizedNLHoldem.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bet_sets.B_5)
    bldr = HistoryEnvBuilder(env_cls=DiscretizedNLHoldem, env_args=args)
    a, b = bldr.get_new_env(is_evaluating=True), bldr.get_new_env(is_evaluating=True)
    rng = np.random.RandomState(3)
    for ep in range(30):
        np.random.seed(ep)
        a.reset()
        done, k = False, 0
        while not done:
            if k == 1:
                b.load_state_dict(a.state_dict())
            legal = a.get_legal_actions()
            act = legal[rng.randint(len(legal))]
            if k >= 1:
                assert b.get_legal_actions() == legal
                ob, rb, db, _ = b.step(act)
            oa, ra, done, _ = a.step(act)
            if k >= 1:
                assert np.array_equal(oa, ob) and list(ra) == list(rb) and done == db
            k += 1
        # chip conservation (test_pokerEnv.py:61-102)
        assert a.seats[0].stack + a.seats[1].stack == 2 * 20000
