"""
CPU tests of the host-side C-ABI entry points against fixtures captured from the reference (tests/golden/*.npz):
index LUTs (lib_luts.so), the scalar 7-card evaluator (lib_hand_eval.so), the heads-up betting engine (PokerEnv) and the
public-tree builder (PublicTree.build_tree).
"""
import ctypes

import numpy as np
import pytest

from helpers import GAMES, all_single_card_boards, env_args, golden, native_tree
from pokerrl_amd import _native
from pokerrl_amd.game import bet_sets
from pokerrl_amd.game import games as G


# ---------------------------------------------------------------------------------------------------------------------
# LUTs: test/game/test_look_up_table.py:14-167 pins the layouts; here the full tables are compared
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("game", ["StandardLeduc", "BigLeduc", "DiscretizedNLHoldem", "Flop5Holdem"])
def test_luts_match_reference(game):
    g = golden("luts.npz")
    lh = getattr(G, game).get_lut_holder()
    for k in ("IDX_2_HOLE_CARDS", "HOLE_CARDS_2_IDX", "CARD_IN_WHAT_RANGE_IDXS", "1DCARD_2_2DCARD", "2DCARD_2_1DCARD"):
        ref = g["%s_%s" % (game, k)]
        mine = getattr(lh, "LUT_" + k)
        assert mine.shape == ref.shape, k
        assert mine.dtype == ref.dtype, (k, mine.dtype, ref.dtype)
        assert np.array_equal(mine, ref), k


def test_lut_invariants():
    lh = G.DiscretizedNLHoldem.get_lut_holder()
    n = 0
    for c1 in range(52):
        for c2 in range(c1 + 1, 52):  # lexicographic counter (test_look_up_table.py:136-143)
            assert lh.LUT_HOLE_CARDS_2_IDX[c1, c2] == n
            assert tuple(lh.LUT_IDX_2_HOLE_CARDS[n]) == (c1, c2)
            n += 1
    assert n == 1326
    counts = np.bincount(lh.LUT_CARD_IN_WHAT_RANGE_IDXS.ravel(), minlength=1326)
    assert np.all(counts == 2)  # every hand appears in exactly two card rows (test_look_up_table.py:44-56)
    assert lh.get_1d_card(np.array([-127, -127])) == -127
    assert np.array_equal(lh.get_1d_cards(lh.get_2d_cards(np.array([0, 51, -127, 17]))), [0, 51, -127, 17])


def test_legacy_lut_symbols_row_pointer_convention():
    """The reference's ctypes wrappers pass 2-D arrays as vectors of row pointers (CppWrapper.py:24-27)."""
    L = _native.lib()

    def rows(a):
        return (a.__array_interface__["data"][0] + np.arange(a.shape[0]) * a.strides[0]).astype(np.intp)

    idx2hc = np.full((1326, 2), -2, np.int8)
    L.get_idx_2_hole_card_lut(rows(idx2hc).ctypes.data_as(ctypes.c_void_p))
    hc2idx = np.full((52, 52), -2, np.int16)
    L.get_hole_card_2_idx_lut(rows(hc2idx).ctypes.data_as(ctypes.c_void_p))
    g = golden("luts.npz")
    assert np.array_equal(idx2hc, g["DiscretizedNLHoldem_IDX_2_HOLE_CARDS"])
    assert np.array_equal(hc2idx, g["DiscretizedNLHoldem_HOLE_CARDS_2_IDX"])
    L.get_1d_card.restype = ctypes.c_int8
    c2 = np.array([7, 3], np.int8)
    assert L.get_1d_card(c2.ctypes.data_as(ctypes.c_void_p)) == 31
    out = np.zeros(2, np.int8)
    L.get_2d_card(ctypes.c_int8(31), out.ctypes.data_as(ctypes.c_void_p))
    assert tuple(out) == (7, 3)


# ---------------------------------------------------------------------------------------------------------------------
# scalar hand evaluator vs the reference binary (values captured by tests/golden/make_golden.py:make_handrank)
# ---------------------------------------------------------------------------------------------------------------------
def test_hand_rank_known_answers():
    g = golden("handrank.npz")
    L = _native.lib()
    L.get_hand_rank_52_holdem.restype = ctypes.c_int32

    def rows(a):
        return (a.__array_interface__["data"][0] + np.arange(a.shape[0]) * a.strides[0]).astype(np.intp)

    for hand, board, rank in zip(g["known_hands"], g["known_boards"], g["known_ranks"]):
        assert _native.hand_rank_7(board, hand[0], hand[1]) == rank
        h2 = np.stack([hand // 4, hand % 4], axis=1).astype(np.int8)
        b2 = np.stack([board // 4, board % 4], axis=1).astype(np.int8)
        got = L.get_hand_rank_52_holdem(rows(h2).ctypes.data_as(ctypes.c_void_p), rows(b2).ctypes.data_as(ctypes.c_void_p))
        assert got == rank
    # survey's table: the quads kicker is the sorted neighbour, not the best card
    assert g["known_ranks"][-1] == 1240704


def test_hand_rank_scalar_matches_reference_on_64_boards():
    g = golden("handrank.npz")
    lut = G.DiscretizedNLHoldem.get_lut_holder().LUT_IDX_2_HOLE_CARDS
    for b, ref in zip(g["boards"], g["ranks"]):
        for h in range(0, 1326):
            c1, c2 = lut[h]
            if c1 in b or c2 in b:
                assert ref[h] == -1
            else:
                assert _native.hand_rank_7(b, c1, c2) == ref[h], (b, h)


# ---------------------------------------------------------------------------------------------------------------------
# heads-up betting engine vs the reference PokerEnv on random (legal and illegal) action sequences
# ---------------------------------------------------------------------------------------------------------------------
ENV_FUZZ = {
    "StandardLeduc": (G.StandardLeduc, 13, [0.0]),
    "BigLeduc": (G.BigLeduc, 100, [0.0]),
    "BigLeduc_short": (G.BigLeduc, 9, [0.0]),
    "NoLimitLeduc_short": (G.NoLimitLeduc, 700, [0.0]),
    "DiscretizedNLLeduc_B3": (G.DiscretizedNLLeduc, 20000, bet_sets.B_3),
    "DiscretizedNLLeduc_B5_short": (G.DiscretizedNLLeduc, 900, bet_sets.B_5),
    "LimitHoldem": (G.LimitHoldem, 48, [0.0]),
    "LimitHoldem_short": (G.LimitHoldem, 11, [0.0]),
    "DiscretizedNLHoldem_B5": (G.DiscretizedNLHoldem, 20000, bet_sets.B_5),
    "DiscretizedNLHoldem_OT11_short": (G.DiscretizedNLHoldem, 2300, bet_sets.OFF_TREE_11),
    "NoLimitHoldem_short": (G.NoLimitHoldem, 1700, [0.0]),
    "Flop5Holdem": (G.Flop5Holdem, 20000, [0.0]),
    "Flop5Holdem_short": (G.Flop5Holdem, 1100, [0.0]),
}


def _state_row(s):
    return [s.round, s.main_pot, s.bet[0], s.bet[1], s.stack[0], s.stack[1], s.allin[0], s.allin[1], s.folded[0],
            s.folded[1], s.acted[0], s.acted[1], s.cur, s.last_raiser, s.capped_happened, s.capped_raiser,
            s.capped_cant_reopen, s.n_actions_ep, s.n_raises_round, s.last_action[0], s.last_action[1], s.last_action[2]]


@pytest.mark.parametrize("name", sorted(ENV_FUZZ))
def test_env_matches_reference_fuzz(name):
    cls, stack, bets = ENV_FUZZ[name]
    rows = golden("env_fuzz.npz")[name]
    game = cls.native_game(env_args(cls, stack, bets))
    L = _native.lib()
    st, info = _native.PrlEnvState(), _native.PrlStepInfo()
    legal = np.zeros(128, np.int32)
    n_legal = ctypes.c_int32()
    is_limit = cls.IS_FIXED_LIMIT_GAME
    is_nl = cls._GAME_TYPE == G.GAME_NOLIMIT
    MAXL = 16
    n_steps = 0
    for r in rows:
        kind, act, amount, nl = int(r[1]), int(r[2]), int(r[3]), int(r[4])
        ref_legal = [int(x) for x in r[5:5 + min(nl, MAXL)]]
        term, chance, pot_before = int(r[5 + MAXL]), int(r[6 + MAXL]), int(r[7 + MAXL])
        ref_state = [int(x) for x in r[8 + MAXL:]]
        if kind == 0:
            _native.check(L.prl_env_reset_host(ctypes.byref(game), ctypes.byref(st)))
        else:
            if is_nl:
                _native.check(L.prl_env_step_processed_host(ctypes.byref(game), ctypes.byref(st), act, amount, ctypes.byref(info)))
            else:
                _native.check(L.prl_env_step_host(ctypes.byref(game), ctypes.byref(st), act, ctypes.byref(info)))
            n_steps += 1
            assert info.is_terminal == term
            assert info.chance_acts == chance
            if term:
                assert info.pot_before_payout == pot_before
                continue
        _native.check(L.prl_env_legal_actions_host(ctypes.byref(game), ctypes.byref(st), legal.ctypes.data_as(ctypes.c_void_p),
                                                   ctypes.byref(n_legal)))
        assert n_legal.value == nl
        assert list(legal[:min(nl, MAXL)]) == ref_legal
        mine = _state_row(st)
        if not is_limit:
            ref_state[18] = mine[18]  # n_raises_this_round exists only in fixed-limit games (PokerEnv.py:1196-1197)
        assert mine == ref_state, (n_steps, mine, ref_state)
    assert n_steps > 200


# ---------------------------------------------------------------------------------------------------------------------
# public-tree structure vs the reference's PublicTree (and vs a walk of the reference env for Flop5Holdem)
# ---------------------------------------------------------------------------------------------------------------------
TREE_FIELDS = ("kind", "actor", "parent", "child_idx", "action", "acted_last", "round", "main_pot", "depth", "n_children",
               "first_col", "col_action")


@pytest.mark.parametrize("name", sorted(GAMES))
def test_tree_matches_reference(name):
    cls, stack, bets = GAMES[name]
    ref = golden("tree_%s.npz" % name)
    t = native_tree(cls, stack, bets, all_single_card_boards(cls))
    assert t.n_nodes == len(ref["kind"])
    for f in TREE_FIELDS:
        assert np.array_equal(t.field(f), ref[f]), f
    assert np.array_equal(t.field("board_id"), ref["board_card"])  # board table = all cards ascending
    # reference counters (PublicTree.n_nodes excludes the root: PublicTree.py:60,163)
    assert t.n_decision + t.n_terminal + int(np.sum(t.field("kind") == 1)) == t.n_nodes


def test_tree_limit_holdem_multistreet_structure_like_reference_env():
    """LimitHoldem, one run-out: the multi-street flat tree (three chance levels, 17 221 nodes) node for node against the betting tree walked
    through the REFERENCE env street by street (tests/golden/make_golden.py tree_lh) -- kinds, actors, actions, pots, rounds, depths,
    child counts, action columns; and the per-street engine's view of it: 7 / 63 / 567 street instances of the registered 27-node shape"""
    ref = golden("tree_LimitHoldem_1runout.npz")
    t = native_tree(G.LimitHoldem, 48, None, np.array([[3, 17, 40, 8, 51]], np.int8))
    assert t.n_nodes == len(ref["kind"]) == 17221
    for f in ("kind", "actor", "parent", "child_idx", "action", "acted_last", "round", "main_pot", "depth", "n_children", "first_col", "col_action"):
        assert np.array_equal(t.field(f), ref[f]), f
    kind, par, rnd = ref["kind"], ref["parent"], ref["round"]
    roots = [i for i in range(len(kind)) if par[i] >= 0 and kind[par[i]] == 1]
    assert np.bincount(rnd[roots]).tolist() == [0, 7, 63, 567]
    sizes = {int(t.field("subtree_size")[r]) for r in roots if rnd[r] == 3}
    assert sizes == {27}


def test_tree_limit_holdem_several_run_outs_like_reference_env():
    """LimitHoldem under 2 flops x 2 turns x 2 rivers (129 676 nodes): the multi-run-out flat tree node for node against the tree walked through
    the REFERENCE env with its deck set per outcome (tests/golden/make_golden.py tree_lh_runouts) -- the protocol fields, the chance nodes'
    child counts and order, and per node the BOARD the reference env holds there against the product's prefix row (board_id -> board_rows):
    the wiring of chance children to prefix rows, which the one-run-out fixture cannot see"""
    ref = golden("tree_LimitHoldem_2x2x2.npz")
    ro = pc_runouts = ref["runouts"]
    import parity_cases as pc
    assert np.array_equal(ro, pc.multistreet_runouts(2, 2, 2))
    t = native_tree(G.LimitHoldem, 48, None, ro)
    assert t.n_nodes == len(ref["kind"]) == 129676
    for f in ("kind", "actor", "parent", "child_idx", "action", "acted_last", "round", "main_pot", "depth", "n_children", "first_col", "col_action"):
        assert np.array_equal(t.field(f), ref[f]), f
    rows = np.asarray(t.board_rows)
    bid = t.field("board_id")
    mine = np.where(bid[:, None] >= 0, rows[np.maximum(bid, 0)], -1)
    assert np.array_equal(mine, ref["board"].astype(mine.dtype))
    # prefix rows: 2 flops + 4 flop-turn prefixes + 8 complete run-outs, every chance node fans out into two children
    assert t.n_boards == 2 + 4 + 8 and set(ref["n_children"][ref["kind"] == 1].tolist()) == {2}


def test_tree_flop5holdem_structure():
    ref = golden("tree_Flop5Holdem_1board.npz")
    boards = np.array([[0, 5, 10, 15, 20], [1, 2, 3, 50, 51], [7, 8, 9, 30, 44]], np.int8)
    t = native_tree(G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards)
    # expand the 1-board reference walk to 3 boards: trunk nodes + 3 copies of the board subtree
    kind = ref["kind"]
    ch = int(np.where(kind == 1)[0][0])
    sub = int(ref["n_children"].shape[0]) - (ch + 1)  # board subtree is the tail of the DFS order ...
    first_board_child = ch + 1
    # ... unless trunk nodes follow it; verify through parents
    size = 1
    stack_ = [first_board_child]
    members = set()
    while stack_:
        x = stack_.pop()
        members.add(x)
        stack_.extend(int(i) for i in np.where(ref["parent"] == x)[0])
    sub_ids = sorted(members)
    assert sub_ids == list(range(first_board_child, first_board_child + len(sub_ids)))
    T = len(sub_ids)
    assert t.n_nodes == len(kind) + 2 * T
    for f in ("kind", "actor", "action", "acted_last", "round", "main_pot", "depth", "n_children"):
        mine = t.field(f)
        assert np.array_equal(mine[:first_board_child + T], ref[f][:first_board_child + T] if f != "n_children" else
                              np.where(np.arange(first_board_child + T) == ch, 3, ref[f][:first_board_child + T])), f
        for b in (1, 2):
            lo = first_board_child + b * T
            assert np.array_equal(mine[lo:lo + T], ref[f][first_board_child:first_board_child + T]), (f, b)
        tail_ref = ref[f][first_board_child + T:]
        assert np.array_equal(mine[first_board_child + 3 * T:], tail_ref), f
    # per board 6 decision + 4 fold + 5 showdown nodes, sum of actions 14; 2 pre-flop decision nodes
    k = t.field("kind")[first_board_child:first_board_child + T]
    assert (int(np.sum(k == 0)), int(np.sum(k == 2)), int(np.sum(k == 3))) == (6, 4, 5)
    assert int(np.sum(t.field("n_children")[first_board_child:first_board_child + T])) == 14
    bid = t.field("board_id")
    assert np.all(bid[:first_board_child] == -1)
    assert np.all(bid[first_board_child + T:first_board_child + 2 * T] == 1)


def test_tree_rejects_runouts_shorter_than_the_deal():
    with pytest.raises(_native.NativeError):
        native_tree(G.LimitHoldem, 48, [0.0], np.array([[0, 1, 2]], np.int8))


def test_tree_rejects_runouts_with_repeated_or_foreign_cards():
    """the board-pass kernels count on C(47, 2) live hands per 5-card board: a card twice (or outside the deck) is refused at build"""
    for bad in ([3, 3, 10, 20, 30], [0, 1, 2, 3, 52], [0, 1, 2, 3, -1]):
        with pytest.raises(_native.NativeError, match="card"):
            native_tree(G.Flop5Holdem, 20000, [1.0], np.array([[4, 9, 14, 19, 24], bad], np.int8))
    with pytest.raises(_native.NativeError, match="card"):
        native_tree(G.StandardLeduc, 13, None, np.array([[0], [6]], np.int8))   # 6-card deck: cards 0..5


def test_tree_deals_out_an_all_in_before_the_deal_on_two_card_ranges():
    """ValueFiller.py:160-175 averages an all-in-before-the-deal showdown over every run-out, for 1-card ranges only. On a 2-card tree
    the builder deals the hand out instead: a chance node without decisions whose children are showdown leaves on the listed boards (the
    same expectation under the same chance weights; round 1 valued such a terminal at 0). A 250-chip stack puts the pot-sized
    pre-flop raise all-in."""
    boards = np.array([[0, 5, 10, 15, 20], [1, 2, 3, 50, 51]], np.int8)
    t = native_tree(G.Flop5Holdem, 250, bet_sets.POT_ONLY, boards)
    k, bid, par, pot = t.field("kind"), t.field("board_id"), t.field("parent"), t.field("main_pot")
    assert not np.any((k == 3) & (bid < 0))                      # no showdown without a board
    run = [i for i in np.where(k == 1)[0] if all(k[c] == 3 for c in np.where(par == i)[0])]
    assert len(run) == 1 and sorted(bid[np.where(par == run[0])[0]]) == [0, 1] and np.all(pot[np.where(par == run[0])[0]] == 500)
    t = native_tree(G.Flop5Holdem, 2000, bet_sets.POT_ONLY, boards)  # deep enough: every showdown is after the deal
    assert int(np.sum(t.field("kind") == 1)) == 1
    # a 1-card game keeps the reference's run-out TERMINAL (V6, pinned by the B3_short goldens)
    t = native_tree(G.DiscretizedNLLeduc, 1500, bet_sets.B_3, np.arange(6, dtype=np.int8).reshape(-1, 1))
    assert np.any((t.field("kind") == 3) & (t.field("board_id") < 0))


def test_multi_street_tree_structure():
    """LimitHoldem deals 3 + 1 + 1 (games.py:134-167). The caller lists run-outs; every street gets a chance level whose
    children are the distinct prefixes. [probe] of the survey: 8 / 70 / 630 / 5670 decision nodes per street for one run-out."""
    runouts = np.array([[0, 1, 2, 3, 4], [0, 1, 2, 3, 5], [0, 1, 2, 6, 7], [8, 9, 10, 11, 12]], np.int8)  # 2 flops; 2 turns below the first
    t = native_tree(G.LimitHoldem, 48, None, runouts)
    k, bid, rnd, par, nch = t.field("kind"), t.field("board_id"), t.field("round"), t.field("parent"), t.field("n_children")
    rows = t.board_rows
    assert rows.shape == (2 + 3 + 4, 5)                           # prefix rows: 2 flops, 3 (flop, turn), 4 run-outs
    assert rows[:2].tolist() == [[0, 1, 2, -1, -1], [8, 9, 10, -1, -1]]
    n_dealt = np.where(bid >= 0, np.sum(rows[np.maximum(bid, 0)] >= 0, axis=1), 0)
    assert np.array_equal(n_dealt[k != 1], np.array([0, 3, 4, 5])[rnd[k != 1]])   # a node of round r sees the board of round r
    ch = np.where(k == 1)[0]
    assert set(nch[ch][rnd[ch] == 0].tolist()) == {2}             # pre-flop chance nodes: two flops
    assert set(nch[ch][rnd[ch] == 1].tolist()) == {1, 2}          # flop chance nodes: two turns below flop 0, one below flop 1
    one = native_tree(G.LimitHoldem, 48, None, runouts[:1])
    dec = one.field("kind") == 0
    assert [int(np.sum(dec & (one.field("round") == r))) for r in range(4)] == [8, 70, 630, 5670]
    with pytest.raises(_native.NativeError, match="run-outs"):    # 3-card rows for a game that deals 5
        native_tree(G.LimitHoldem, 48, None, runouts[:, :3])


def test_c_abi_exports_every_declared_symbol():
    """include/pokerrl_hip.h is the contract: every prototype it declares must be exported by the shared library."""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "pokerrl_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b([a-z][a-z0-9_]*)\s*\(", hdr))
    names = {n for n in names if n.startswith(("prl_", "get_"))}
    assert len(names) > 20
    L = _native.lib()
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing
    assert _native.build_flavor() == "hip-gfx950"


def test_solver_field_ids_of_the_python_host_match_the_header():
    """the PRL_SF_* ids of include/pokerrl_hip.h are what pokerrl_amd._native passes to prl_solver_get / prl_solver_get_cols"""
    import os
    import re
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "pokerrl_hip.h")).read()
    ids = {m.group(1).lower(): int(m.group(2)) for m in re.finditer(r"\bPRL_SF_([A-Z0-9_]+)\s*=\s*(\d+)", hdr)}
    assert len(ids) >= 19 and len(set(ids.values())) == len(ids)
    assert ids == dict(_native.SF), (sorted(set(ids.items()) ^ set(_native.SF.items())))


@pytest.mark.parametrize("tag", ["StandardLeduc", "DiscretizedNLLeduc", "DiscretizedNLHoldem"])
def test_head_to_head_vs_reference(tag, tmp_path):
    """LocalHead2HeadMaster on the native-backed env -- per-hand winnings and logged scalars of the reference's
    evaluator for the two modes of the fixture agent (tests/golden/make_h2h_golden.py). Host-side functions of the library only."""
    import json
    import lbr_fixture_agent as fx
    from pokerrl_amd.eval.head_to_head import H2HArgs, LocalHead2HeadMaster
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    from pokerrl_amd.game.wrappers import HistoryEnvBuilder
    from pokerrl_amd.rl.base_cls.EvalAgentBase import EvalAgentBase
    from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase
    g = golden("h2h_%s.npz" % tag)
    game_cls, bets = {"StandardLeduc": (G.StandardLeduc, None), "DiscretizedNLLeduc": (G.DiscretizedNLLeduc, bet_sets.B_3),
                      "DiscretizedNLHoldem": (G.DiscretizedNLHoldem, bet_sets.B_5)}[tag]

    class Chief:
        def __init__(self):
            self.names, self.log = [], []

        def create_experiment(self, name):
            self.names.append(name)
            return name

        def add_scalar(self, exp, graph, step, value):
            self.log.append([exp, graph, int(step), float(value)])

    t_prof = TrainingProfileBase(
        name="h2h", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9,
        game_cls=game_cls, env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH", "HASH2"), eval_stack_sizes=None,
        module_args={"env": game_cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bets) if bets is not None
                     else game_cls.ARGS_CLS(n_seats=2), "h2h": H2HArgs(n_hands=int(g["n_hands"]))}, path_data=str(tmp_path))
    chief = Chief()
    m = LocalHead2HeadMaster(t_prof=t_prof, chief_handle=chief, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=11))
    m.set_modes(["HASH", "HASH2"])
    np.random.seed(int(g["np_seed"]))
    w = m.play(stack_size=t_prof.eval_stack_sizes[0])
    assert w.dtype == np.float32 and np.array_equal(w, g["winnings"])
    np.random.seed(int(g["np_seed"]))
    m.set_modes(["HASH", "HASH2"])  # restarts the fixture agents' draw counters
    m.evaluate(iter_nr=3)
    assert chief.names == json.loads(str(g["experiments"]))
    assert chief.log == json.loads(str(g["log"]))


@pytest.mark.parametrize("tag", ["StandardLeduc", "DiscretizedNLLeduc", "DiscretizedNLHoldem"])
def test_agent_tournament_vs_reference(tag, tmp_path, capsys):
    """AgentTournament on the native-backed env -- the reference's (mean, upper, lower) and per-hand winnings for the two
    modes of the fixture agent and the same np.random seed (tests/golden/make_tournament_golden.py). Also: a game class is callable
    like the reference's (game_cls(env_args=..., lut_holder=..., is_evaluating=...) builds an env)."""
    import lbr_fixture_agent as fx
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    from pokerrl_amd.game.AgentTournament import AgentTournament
    from pokerrl_amd.game.poker_env import PokerEnv
    from pokerrl_amd.game.wrappers import HistoryEnvBuilder
    from pokerrl_amd.rl.base_cls.EvalAgentBase import EvalAgentBase
    from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase
    g = golden("tournament_%s.npz" % tag)
    game_cls, bets = {"StandardLeduc": (G.StandardLeduc, None), "DiscretizedNLLeduc": (G.DiscretizedNLLeduc, bet_sets.B_3),
                      "DiscretizedNLHoldem": (G.DiscretizedNLHoldem, bet_sets.B_5)}[tag]
    env_args = game_cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bets) if bets is not None else game_cls.ARGS_CLS(n_seats=2)
    t_prof = TrainingProfileBase(
        name="tournament", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9,
        game_cls=game_cls, env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH", "HASH2"), eval_stack_sizes=None,
        module_args={"env": env_args}, path_data=str(tmp_path))
    cls = fx.make_agent_cls(EvalAgentBase, seed=11)
    n = int(g["n_games_per_seat"])
    t = AgentTournament(env_cls=game_cls, env_args=env_args, eval_agent_1=cls(t_prof=t_prof, mode="HASH"), eval_agent_2=cls(t_prof=t_prof, mode="HASH2"))
    np.random.seed(int(g["np_seed"]))
    w = t.play(n)
    assert w.dtype == np.float32 and np.array_equal(w, g["winnings"])
    t = AgentTournament(env_cls=game_cls, env_args=env_args, eval_agent_1=cls(t_prof=t_prof, mode="HASH"), eval_agent_2=cls(t_prof=t_prof, mode="HASH2"))
    np.random.seed(int(g["np_seed"]))
    assert t.run(n) == tuple(float(x) for x in g["result"])
    assert "Played %d hands of poker." % (2 * n) in capsys.readouterr().out
    env = game_cls(env_args=env_args, lut_holder=game_cls.get_lut_holder(), is_evaluating=True)
    assert isinstance(env, PokerEnv) and env.N_SEATS == 2


def test_file_util_formats(tmp_path):
    """pokerrl_amd.util.file_util: the reference's names and formats (file_util.py:13-56); files of one package load in the other"""
    import json
    import os
    import pickle
    from pokerrl_amd.util import file_util as fu
    d = {"a": [1, 2.5, "x"], "b": {"c": None}}
    sub = os.path.join(str(tmp_path), "new", "dir")
    fu.write_dict_to_file_json(sub, "tree", d)
    fu.write_dict_to_file_js(sub, 7, d)
    fu.do_pickle(d, sub, "state")
    assert json.load(open(os.path.join(sub, "tree.json"))) == d
    assert open(os.path.join(sub, "7.js")).read() == "const data=" + json.dumps(d)
    assert pickle.load(open(os.path.join(sub, "state.pkl"), "rb")) == d
    assert fu.load_pickle(sub, "state") == d and fu.load_pickle(os.path.join(sub, "state.pkl")) == d
    assert sorted(fu.get_all_files_in_dir(sub)) == ["7.js", "state.pkl", "tree.json"]
    assert fu.get_all_dirs_in_dir(os.path.join(str(tmp_path), "new")) == ["dir"]
    assert fu.get_file_name_without_ending_and_path_from_path(os.path.join(sub, "state.pkl")) == "state"
    fu.create_dir_if_not_exist(sub)  # exists already: no error


def test_product_fails_loudly_without_a_device():
    """No CPU fallback: in a GPU-less container every device entry point of the PRODUCT library reports PRL_ERR_NO_DEVICE (or
    a bad-argument error first) with a message, and the Python host raises instead of computing anything on the CPU."""
    if _native.device_available():
        pytest.skip("a HIP device is present")
    L = _native.lib()
    assert L.prl_set_device(0) < 0 and b"HIP device" in L.prl_last_error()
    with pytest.raises(_native.NativeError):
        _native.require_device()
    out = np.zeros((4, 9), np.int8)
    assert L.prl_deal_decks(4, 52, 9, 0, 0, out.ctypes.data_as(ctypes.c_void_p)) < 0
    from pokerrl_amd.eval.lbr.BatchedLBR import deal_decks
    with pytest.raises(_native.NativeError):
        deal_decks(4, 52, 9, 0)
    from pokerrl_amd.game import bet_sets as bs
    t = native_tree(G.StandardLeduc, 13, None, all_single_card_boards(G.StandardLeduc))  # trees are host objects
    with pytest.raises(_native.NativeError):
        _native.NativeSolver(t, "plus", 0)  # solver state lives in HBM
    del bs


def test_c_abi_rejects_bad_arguments():
    """error behaviour of the flat ABI: negative status + prl_last_error(), never a crash, for NULL / out-of-range arguments"""
    L = _native.lib()
    out = np.zeros((4, 9), np.int8)
    for args in [(0, 52, 9), (4, 52, 0), (4, 52, 17), (4, 5, 9), (4, 200, 9)]:
        assert L.prl_deal_decks(args[0], args[1], args[2], 0, 0, out.ctypes.data_as(ctypes.c_void_p)) < 0
        assert L.prl_last_error()
    assert L.prl_deal_decks(4, 52, 9, 0, 0, None) < 0
    assert L.prl_set_device(-1) < 0
    assert L.prl_lbr_batch_run(None, None, None, 1, 0, -1, 1, 7, 0, 1.0, 1.0, None, None, None, None) < 0
    assert L.prl_h2h_batch_run(None, None, 1, 0, 1, 1, 1, 2, 0, 1.0, 1.0, None, None, None, None) < 0


def test_bench_board_sets_are_disjoint_shards():
    """bench.seeded_boards: sorted distinct 5-card boards; different offsets (= ranks of a sharded run) never share a board"""
    import bench
    a, b = bench.seeded_boards(4096, 0), bench.seeded_boards(4096, 0, offset=7 * 4096)
    both = np.concatenate([a, b]).astype(np.int64)
    assert both.dtype == np.int64 and a.dtype == np.int8 and a.shape == (4096, 5)
    assert np.all(np.diff(both, axis=1) > 0) and both.min() >= 0 and both.max() <= 51
    keys = (both * np.array([52 ** 4, 52 ** 3, 52 ** 2, 52, 1])).sum(axis=1)
    assert len(np.unique(keys)) == len(keys)
    assert np.array_equal(bench.seeded_boards(100, 0, offset=50), bench.seeded_boards(150, 0)[50:])
    assert not np.array_equal(bench.seeded_boards(100, 1), bench.seeded_boards(100, 0))


# ---------------------------------------------------------------------------------------------------------------------
# partial trees: PublicTree(stop_at_street=...) (PublicTree.py:72,173,185) -- host-side structure, no device needed
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["StandardLeduc", "DiscretizedNLLeduc_POT"])
@pytest.mark.parametrize("stop", [0, 1])
def test_partial_tree_matches_reference(name, stop):
    """the reference's tree with stop_at_street (tests/golden/make_partial_tree_golden.py): same nodes in the same DFS order, leaves
    where the round reaches the limit, the reference's n_nodes / n_nonterm counters; the solver refuses such a tree"""
    from pokerrl_amd.game.PublicTree import PublicTree
    from pokerrl_amd.game.wrappers import HistoryEnvBuilder
    cls, stack, bets = GAMES[name]
    args = env_args(cls, stack, bets)
    g = golden("tree_partial.npz")
    key = "%s_stop%d_" % (name, stop)
    tree = PublicTree(env_bldr=HistoryEnvBuilder(env_cls=cls, env_args=args), stack_size=args.starting_stack_sizes_list, stop_at_street=stop)
    tree.build_tree()
    t = tree.native_tree
    kind = t.field("kind")
    assert np.array_equal(kind, g[key + "kind"])
    assert np.array_equal(t.field("parent"), g[key + "parent"]) and np.array_equal(t.field("n_children"), g[key + "n_children"])
    assert np.array_equal(t.field("round"), g[key + "round"]) and np.array_equal(t.field("depth"), g[key + "depth"])
    assert np.array_equal(np.where(kind == 0, t.field("actor"), -1), g[key + "actor"])
    assert [tree.n_nodes, tree.n_nonterm] == g[key + "counters"].tolist()
    leaves = [n for n in tree.nodes() if not n.is_terminal and not n.children]
    assert leaves and all(n.env_state["current_round"] >= stop for n in leaves if n.p_id_acting_next != tree.CHANCE_ID)
    assert all(len(n.allowed_actions) >= 2 for n in leaves if n.p_id_acting_next != tree.CHANCE_ID)  # legal actions of an unexpanded node
    with pytest.raises(RuntimeError, match="partial tree"):
        tree.compute_ev()
    with pytest.raises(_native.NativeError):  # "partial tree" on a GPU box, "no usable HIP device" here
        _native.NativeSolver(t, "plus", 0)


def test_board_enumeration_like_the_tree_builder_deals():
    """pokerrl_amd/game/board_enum.py (PublicTree.py:188-210 for games that deal several cards): Leduc = the deck, cards ascending (the
    reference's order); Flop5Holdem = all C(52,5) boards in combinatorial order / a seeded subset; LimitHoldem = run-outs street by street
    with caps; the host tree builder accepts all of them and PublicTree builds from them when it is not handed boards"""
    from math import comb
    from pokerrl_amd.game import board_enum
    from pokerrl_amd.game import games as G
    assert board_enum.single_deal_boards(G.StandardLeduc).ravel().tolist() == [0, 1, 2, 3, 4, 5]
    assert board_enum.single_deal_boards(G.BigLeduc).shape == (24, 1)
    b = board_enum.single_deal_boards(G.Flop5Holdem, n_boards=5000, offset=comb(52, 5) - 5000)
    assert b[-1].tolist() == [47, 48, 49, 50, 51] and np.all(np.diff(b.astype(int), axis=1) > 0) and len({tuple(x) for x in b.tolist()}) == 5000
    s1, s2 = board_enum.single_deal_boards(G.Flop5Holdem, n_boards=300, seed=3), board_enum.single_deal_boards(G.Flop5Holdem, n_boards=300, seed=3, offset=300)
    assert not ({tuple(x) for x in s1.tolist()} & {tuple(x) for x in s2.tolist()})
    r = board_enum.runouts(G.LimitHoldem, (2, 3, 2))
    assert r.shape == (12, 5) and r[0].tolist() == [0, 1, 2, 3, 4] and r[-1].tolist() == [0, 1, 3, 5, 4]
    assert all(len(set(x)) == 5 for x in r.tolist())
    rs = board_enum.runouts(G.LimitHoldem, (3, 2, 2), seed=1)
    assert rs.shape == (12, 5) and np.array_equal(rs, board_enum.runouts(G.LimitHoldem, (3, 2, 2), seed=1)) and all(len(set(x)) == 5 for x in rs.tolist())
    with pytest.raises(ValueError, match="cap the chance outcomes"):
        board_enum.runouts(G.LimitHoldem)
    # the tree builder takes them: 2 flops x 3 turns x 2 rivers -> 2 + 6 + 12 prefix rows, one chance level per street
    t = _native.NativeTree.for_game(G.LimitHoldem, 48, None, r)
    assert t.n_boards == 2 + 6 + 12
    # PublicTree without boards: the builder's own enumeration (partial tree: structure only, no device needed)
    from pokerrl_amd.game.PublicTree import PublicTree
    from pokerrl_amd.game.wrappers import HistoryEnvBuilder
    args = G.Flop5Holdem.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[20000, 20000])
    tree = PublicTree(env_bldr=HistoryEnvBuilder(env_cls=G.Flop5Holdem, env_args=args), stack_size=[20000, 20000], stop_at_street=1, n_boards=40, board_seed=0)
    tree.build_tree()
    import bench
    assert np.array_equal(tree._boards, bench.seeded_boards(40, 0))
    # a copy made BEFORE build_tree deals the same capped / seeded boards (the caps travel with it)
    tree2 = PublicTree(env_bldr=HistoryEnvBuilder(env_cls=G.Flop5Holdem, env_args=args), stack_size=[20000, 20000], stop_at_street=1, n_boards=40, board_seed=0)
    assert tree2.copy()._board_caps == (40, None, 0)
    # every board of Flop5Holdem is more than one GPU holds: said so at once, not as a late out-of-memory
    with pytest.raises(ValueError, match="pass n_boards"):
        board_enum.single_deal_boards(G.Flop5Holdem)
    assert board_enum.single_deal_boards(G.Flop5Holdem, n_boards=7).shape == (7, 5)


def test_board_classes_and_the_default_board_choice():
    """board_enum: Flop5Holdem's 2 598 960 boards fall into 134 459 suit classes with orbit sizes 4 / 12 / 24 (the literature's count); asking for every
    board of the game picks the classes, a capped / seeded list stays literal, Leduc is untouched; 169 suit classes of hands"""
    from pokerrl_amd.game import board_enum
    reps, mult = board_enum.single_deal_board_classes(G.Flop5Holdem)
    assert reps.shape == (134459, 5) and int(mult.sum()) == 2598960 and sorted(set(mult.tolist())) == [4, 12, 24]
    assert np.all(np.diff(reps.astype(np.int64), axis=1) > 0)                       # cards ascending
    assert tuple(reps[0]) == (0, 1, 2, 3, 4)                                        # the smallest member of the first class
    keys = reps.astype(np.int64) @ (52 ** np.arange(4, -1, -1, dtype=np.int64))
    assert np.all(np.diff(keys) > 0)                                                # classes in ascending order of their smallest member
    b, m = board_enum.default_boards_or_classes(G.Flop5Holdem)
    assert m is not None and np.array_equal(b, reps)
    b, m = board_enum.default_boards_or_classes(G.Flop5Holdem, n_boards=100, seed=3)
    assert m is None and b.shape == (100, 5)
    b, m = board_enum.default_boards_or_classes(G.StandardLeduc)
    assert m is None and b.shape == (6, 1)
    with pytest.raises(ValueError):
        board_enum.default_boards_or_classes(G.Flop5Holdem, n_boards=100, suit_isomorphism=True)
    cls = board_enum.hand_suit_classes(G.Flop5Holdem)
    assert cls.shape == (1326,) and int(cls.max()) == 168 and sorted(np.bincount(np.bincount(cls)).nonzero()[0].tolist()) == [4, 6, 12]
    # the whole-game fixtures (oracle's chunked run over exactly these classes, make_fhp_golden_chunked.py --whole-game) still describe this enumeration
    import os
    from helpers import GOLDEN, h32
    for variant in ("plus", "linear"):
        path = os.path.join(GOLDEN, "fhp_whole_game_%s_chunked.npz" % variant)
        if os.path.isfile(path):
            g = np.load(path)
            assert h32(reps) == str(g["boards_sha256"]) and h32(mult.astype(np.int32)) == str(g["mult_sha256"]) and int(g["n_iters"]) >= 4
            assert g["expl_history"].shape == (int(g["n_iters"]) + 1, 2) and np.all(np.isfinite(g["expl_history"]))


def test_policy_table_host_logic():
    """pokerrl_amd.rl.tabular_agent.PolicyTable without a device: the open-addressed key table holds every row where the device's probe sequence finds it,
    unknown histories fall back to uniform play over the legal actions, two rows under one key are refused"""
    from pokerrl_amd.rl.tabular_agent import PolicyTable, _first_slot, _key64
    rng = np.random.RandomState(0)
    keys = [(int(a), int(b)) for a, b in rng.randint(0, 2 ** 32, size=(300, 2), dtype=np.uint64)]
    keys[5] = (0, 0)  # the all-zero key is stored as 1 (0 marks an empty slot)
    probs = rng.random_sample((300, 4, 6)).astype(np.float32)
    t = PolicyTable(keys, probs)
    assert t.capacity == 1024 and (t.keys != 0).sum() == 300
    for r, hk in enumerate(keys):
        i = _first_slot(hk, t.capacity - 1)
        while t.keys[i] != _key64(hk):  # the device's probe: linear from the first slot, never across an empty one
            assert t.keys[i] != 0
            i = (i + 1) & (t.capacity - 1)
        assert t.rows[i] == r and t.row_of(hk) == r
        assert np.array_equal(t.policy(hk, [0, 1]), probs[r].T)
    p = t.policy((123, 456), [1, 3])
    assert t.row_of((123, 456)) == -1 and np.array_equal(p[:, [1, 3]], np.full((6, 2), np.float32(0.5))) and not p[:, [0, 2]].any()
    with pytest.raises(AssertionError):
        PolicyTable([(1, 2), (1, 2)], probs[:2])
