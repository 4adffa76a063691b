"""Golden vectors for LBR's check-down equity BEFORE THE FLOP (five board cards to come, all C(48,5) = 1 712 304 run-outs): the REFERENCE's
_LBRRolloutManager (LocalLBRWorker.py:379-512) on DiscretizedNLHoldem at the first decision, seeded agent ranges.
-> tests/golden/lbr_equity_preflop.npz. Needs ~20 GB of RAM (the reference keeps two index lists per run-out) and ~15 minutes."""
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.eval.lbr.LocalLBRWorker import _LBRRolloutManager  # noqa: E402
from PokerRL.game import bet_sets  # noqa: E402
from PokerRL.game.PokerRange import PokerRange  # noqa: E402
from PokerRL.game.games import DiscretizedNLHoldem  # noqa: E402
from PokerRL.game.wrappers import HistoryEnvBuilder  # noqa: E402


class _TP:
    DEBUGGING = False


if __name__ == "__main__":
    n_ranges = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    args = DiscretizedNLHoldem.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bet_sets.B_3)
    bldr = HistoryEnvBuilder(env_cls=DiscretizedNLHoldem, env_args=args)
    env = bldr.get_new_env(is_evaluating=True)
    np.random.seed(4242)
    env.reset()
    seat = 1
    lbr_hand = env.get_hole_cards_of_player(seat)
    t0 = time.time()
    m = _LBRRolloutManager(t_prof=_TP, env_bldr=bldr, env=env, lbr_hand_2d=lbr_hand)
    print("manager built: %d run-outs, %.0f s" % (len(m._bigger_idxs), time.time() - t0), flush=True)
    rng = np.random.RandomState(17)
    ranges, wps = [], []
    for k in range(n_ranges):
        ar = PokerRange(env_bldr=bldr)
        if k == 1:
            ar._range = (rng.random_sample(bldr.rules.RANGE_SIZE) ** 3).astype(np.float32)
        elif k == 2:  # sparse
            ar._range = (rng.random_sample(bldr.rules.RANGE_SIZE) * (rng.random_sample(bldr.rules.RANGE_SIZE) < 0.05)).astype(np.float32)
        ar.set_cards_to_zero_prob(lbr_hand)
        ranges.append(ar.range.copy())
        t0 = time.time()
        wps.append(np.float32(m.get_lbr_checkdown_equity(agent_range=ar)))
        print("range %d: wp %r, %.0f s" % (k, wps[-1], time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "lbr_equity_preflop.npz"), hand=bldr.lut_holder.get_1d_cards(lbr_hand).astype(np.int8),
                        range=np.stack(ranges), wp=np.array(wps, np.float32), n_runouts=len(m._bigger_idxs), numpy=np.__version__)
    print("wrote lbr_equity_preflop.npz")
