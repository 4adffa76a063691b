"""Shared helpers of the test-suite (golden loading, tree conversion, game table)."""
import hashlib
import os

import numpy as np

from pokerrl_amd.game import bet_sets
from pokerrl_amd.game import games as G

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

# fixture key -> (game class, stack, bet set); mirrors tests/golden/make_golden.py:GAMES
GAMES = {
    "StandardLeduc": (G.StandardLeduc, 13, bet_sets.POT_ONLY),
    "BigLeduc": (G.BigLeduc, 100, bet_sets.POT_ONLY),
    "DiscretizedNLLeduc_POT": (G.DiscretizedNLLeduc, 20000, bet_sets.POT_ONLY),
    "DiscretizedNLLeduc_B3_short": (G.DiscretizedNLLeduc, 1500, bet_sets.B_3),
}


def golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def h32(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256((a + a.dtype.type(0)).tobytes()).hexdigest()  # "+ 0" folds -0.0 into +0.0


def env_args(game_cls, stack, bets):
    return game_cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bets)


def golden_tree_as_flat(name):
    """tests/golden/tree_<name>.npz (captured from the reference's PublicTree) -> dict with CSR children + board ids."""
    t = golden("tree_%s.npz" % name)
    n = len(t["kind"])
    cs = np.zeros(n + 1, np.int32)
    cs[1:] = np.cumsum(t["n_children"])
    cl = np.full(max(n - 1, 0), -1, np.int32)
    for i in range(1, n):
        cl[cs[t["parent"][i]] + t["child_idx"][i]] = i
    t["child_start"], t["child_list"] = cs, cl
    if "board_card" in t:
        t["board_id"] = t["board_card"].astype(np.int32)  # Leduc family: the board table is "all cards ascending"
    return t


def native_tree(game_cls, stack, bets, boards):
    from pokerrl_amd import _native
    args = env_args(game_cls, stack, bets)
    return _native.NativeTree(game_cls.native_game(args), game_cls.native_rules(), boards)


def all_single_card_boards(game_cls):
    return np.arange(game_cls.RULES.N_CARDS_IN_DECK, dtype=np.int8).reshape(-1, 1)
