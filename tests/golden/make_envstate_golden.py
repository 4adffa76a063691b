"""
Generates tests/golden/env_states.npz by RUNNING THE REFERENCE (/root/reference) in this container:
  tree_<game>       [n_nodes, 24] public part of node.env_state of every node of the reference's PublicTree, DFS pre-order
                    (PublicTree.py:205-293: decision nodes = state after the action; terminal and chance-pending nodes = state
                    after the action BEFORE the money moves; chance outcomes = post-transition state with the board dealt)
  step_<game>       random play with RETURN_PRE_TRANSITION_STATE_IN_INFO: one row per step,
                    [episode, action, amount, done, chance_acts, has_pre, pre-state (24)]  (PokerEnv.py:737-787)
State row = make_golden._pub_state's 22 fields + first board card (1d, -1 = not dealt) + number of cards left in the deck.

    python tests/golden/make_envstate_golden.py
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.game import bet_sets  # noqa: E402
from PokerRL.game.Poker import Poker  # noqa: E402
from PokerRL.game.PokerEnvStateDictEnums import EnvDictIdxs  # noqa: E402
from PokerRL.game.games import DiscretizedNLHoldem, DiscretizedNLLeduc, Flop5Holdem, LimitHoldem, NoLimitHoldem, StandardLeduc  # noqa: E402
from PokerRL.game.wrappers import HistoryEnvBuilder  # noqa: E402

TREES = {"StandardLeduc": (StandardLeduc, 13, bet_sets.POT_ONLY), "DiscretizedNLLeduc_POT": (DiscretizedNLLeduc, 20000, bet_sets.POT_ONLY),
         "DiscretizedNLLeduc_B3_short": (DiscretizedNLLeduc, 1500, bet_sets.B_3)}
STEPS = {"StandardLeduc": (StandardLeduc, 13, [0.0]), "DiscretizedNLLeduc_B3": (DiscretizedNLLeduc, 20000, bet_sets.B_3),
         "LimitHoldem": (LimitHoldem, 48, [0.0]), "DiscretizedNLHoldem_B5": (DiscretizedNLHoldem, 20000, bet_sets.B_5),
         "NoLimitHoldem_short": (NoLimitHoldem, 1700, [0.0]), "Flop5Holdem_short": (Flop5Holdem, 1100, [0.0])}


def bldr_of(cls, stack, bets):
    args = cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bets)
    return HistoryEnvBuilder(env_cls=cls, env_args=args), args


def row_of(s, lut_holder):
    seats = s[EnvDictIdxs.seats]
    cr, la = s[EnvDictIdxs.capped_raise], s[EnvDictIdxs.last_action]
    none = lambda v: -1 if v is None else int(v)  # noqa: E731
    b1d = lut_holder.get_1d_cards(s[EnvDictIdxs.board_2d])
    deck = s[EnvDictIdxs.deck]["deck_remaining"]
    return [int(s[EnvDictIdxs.current_round]), int(s[EnvDictIdxs.main_pot]), int(seats[0]["current_bet"]), int(seats[1]["current_bet"]),
            int(round(float(seats[0]["stack"]))), int(round(float(seats[1]["stack"]))), int(seats[0]["is_allin"]), int(seats[1]["is_allin"]),
            int(seats[0]["folded_this_episode"]), int(seats[1]["folded_this_episode"]), int(seats[0]["has_acted_this_round"]),
            int(seats[1]["has_acted_this_round"]), none(s[EnvDictIdxs.current_player]), none(s[EnvDictIdxs.last_raiser]),
            0 if cr is None else 1, -1 if cr is None else none(cr[0]), -1 if cr is None else none(cr[1]),
            int(s[EnvDictIdxs.n_actions_this_episode]), int(s.get(EnvDictIdxs.n_raises_this_round, 0) or 0), none(la[0]), none(la[1]), none(la[2]),
            int(b1d[0]) if b1d[0] != Poker.CARD_NOT_DEALT_TOKEN_1D else -1, int(len(deck))]


def main():
    from PokerRL.game._.tree.PublicTree import PublicTree
    out = {}
    for name, (cls, stack, bets) in TREES.items():
        bldr, args = bldr_of(cls, stack, bets)
        tree = PublicTree(env_bldr=bldr, stack_size=args.starting_stack_sizes_list, stop_at_street=None)
        tree.build_tree()
        rows = []

        def visit(n):
            rows.append(row_of(n.env_state, bldr.lut_holder))
            for c in n.children:
                visit(c)

        visit(tree.root)
        out["tree_" + name] = np.array(rows, np.int64)
        print("tree", name, out["tree_" + name].shape)
    for name, (cls, stack, bets) in STEPS.items():
        bldr, args = bldr_of(cls, stack, bets)
        env = bldr.get_new_env(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
        a = env.get_args()
        a.RETURN_PRE_TRANSITION_STATE_IN_INFO = True
        env.set_args(a)
        rng = np.random.RandomState(sum(map(ord, name)) + 7)
        np.random.seed(sum(map(ord, name)) + 7)
        is_nl = cls is NoLimitHoldem
        rows = []
        for ep in range(60):
            env.reset()
            done = False
            while not done:
                legal = env.get_legal_actions()
                act = int(legal[rng.randint(len(legal))]) if rng.rand() < 0.8 else int(rng.randint(env.N_ACTIONS))
                amount = -1
                if is_nl:
                    amount = int(rng.randint(0, 2 * stack + 2))
                    _, _, done, info = env.step((act, amount))
                else:
                    _, _, done, info = env.step(act)
                pre = info["state_dict_before_money_move"]
                rows.append([ep, act, amount, int(done), int(bool(info["chance_acts"])), int(pre is not None)] +
                            (row_of(pre, bldr.lut_holder) if pre is not None else [0] * 24))
        out["step_" + name] = np.array(rows, np.int64)
        out["step_" + name + "_cfg"] = np.array([stack] + [int(round(b * 1000)) for b in sorted(bets)], np.int64)
        print("step", name, out["step_" + name].shape)
    np.savez_compressed(os.path.join(HERE, "env_states.npz"), numpy=np.__version__, **out)
    print("wrote env_states.npz")


if __name__ == "__main__":
    main()
