"""Golden vectors for the LBR check-down equity alone (LocalLBRWorker.py:379-512): the REFERENCE's _LBRRolloutManager on
seeded random agent ranges, at every stage LBR can act in. -> tests/golden/lbr_equity.npz"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.eval.lbr.LocalLBRWorker import _LBRRolloutManager  # noqa: E402
from PokerRL.game import Poker, bet_sets  # noqa: E402
from PokerRL.game.PokerRange import PokerRange  # noqa: E402
from PokerRL.game.games import DiscretizedNLHoldem, StandardLeduc  # (BigLeduc: the reference's all-hands ranker is sized for Leduc, game_rules.py:124)  # noqa: E402
from PokerRL.game.wrappers import HistoryEnvBuilder  # noqa: E402


class _TP:
    DEBUGGING = False


def cases(game_cls, args, n, seed, max_to_deal):
    bldr = HistoryEnvBuilder(env_cls=game_cls, env_args=args)
    env = bldr.get_new_env(is_evaluating=True)
    rng = np.random.RandomState(seed)
    out = []
    trial = 0
    while len(out) < n:
        trial += 1
        np.random.seed(seed * 1000 + trial)
        env.reset()
        done = False
        for _ in range(rng.randint(0, 8)):
            if env.current_round == max(env.ALL_ROUNDS_LIST):
                break
            _o, _r, done, _i = env.step(1)
            if done:
                break
        if done:
            continue
        seat = int(rng.randint(0, 2))
        lbr_hand = env.get_hole_cards_of_player(seat)
        ar = PokerRange(env_bldr=bldr)
        kind = rng.randint(0, 4)
        if kind == 0:
            ar._range = (rng.random_sample(bldr.rules.RANGE_SIZE) ** 3).astype(np.float32)
        elif kind == 1:  # sparse
            ar._range = (rng.random_sample(bldr.rules.RANGE_SIZE) * (rng.random_sample(bldr.rules.RANGE_SIZE) < 0.05)).astype(np.float32)
        elif kind == 2:  # wide dynamic range
            ar._range = np.exp(rng.uniform(-30, 0, bldr.rules.RANGE_SIZE)).astype(np.float32)
        ar.set_cards_to_zero_prob(lbr_hand)
        dealt2d = np.array([c for c in env.board if c[0] != Poker.CARD_NOT_DEALT_TOKEN_1D]).reshape(-1, 2)
        if dealt2d.shape[0]:
            ar.set_cards_to_zero_prob(dealt2d)
        n_to_deal = bldr.lut_holder.DICT_LUT_N_CARDS_OUT[env.ALL_ROUNDS_LIST[-1]] - bldr.lut_holder.DICT_LUT_N_CARDS_OUT[env.current_round]
        if n_to_deal > max_to_deal:
            continue
        m = _LBRRolloutManager(t_prof=_TP, env_bldr=bldr, env=env, lbr_hand_2d=lbr_hand)
        wp = np.float32(m.get_lbr_checkdown_equity(agent_range=ar))
        dealt = bldr.lut_holder.get_1d_cards(dealt2d) if dealt2d.shape[0] else np.zeros(0, np.int8)
        board = np.full(5, -1, np.int8)
        board[:dealt.shape[0]] = dealt
        hand = np.full(2, -1, np.int8)
        h1 = bldr.lut_holder.get_1d_cards(lbr_hand)
        hand[:h1.shape[0]] = h1
        out.append((board, np.int8(dealt.shape[0]), hand, ar.range.copy(), wp))
    return out


if __name__ == "__main__":
    res = {}
    for tag, game_cls, args, n, mx in (("StandardLeduc", StandardLeduc, StandardLeduc.ARGS_CLS(n_seats=2), 60, 1),
                                       ("DiscretizedNLHoldem", DiscretizedNLHoldem,
                                        DiscretizedNLHoldem.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bet_sets.B_3), 40, 2)):
        cs = cases(game_cls, args, n, 11, mx)
        res[tag + "_board"] = np.stack([c[0] for c in cs])
        res[tag + "_n_dealt"] = np.array([c[1] for c in cs], np.int8)
        res[tag + "_hand"] = np.stack([c[2] for c in cs])
        res[tag + "_range"] = np.stack([c[3] for c in cs])
        res[tag + "_wp"] = np.array([c[4] for c in cs], np.float32)
        print(tag, len(cs), "to deal:", np.bincount((5 if "Holdem" in tag else 1) - res[tag + "_n_dealt"]))
    np.savez_compressed(os.path.join(HERE, "lbr_equity.npz"), **res)
