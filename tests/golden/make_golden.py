"""
Generates the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (/root/reference, read-only) in this
container. The fixtures travel to the GPU box; the reference does not.

    python tests/golden/make_golden.py [luts] [handrank] [handrank_exhaustive] [tree] [tree_lh] [env] [cfr] [br]

Everything is deterministic (fixed seeds); numpy version is recorded in every file because the reference's float32
results depend on NumPy-2 promotion rules (SURVEY.md section 8a "dtype ledger").
"""
import itertools
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

np = ref_harness.setup()

from PokerRL.game import bet_sets  # noqa: E402
from PokerRL.game.Poker import Poker  # noqa: E402
from PokerRL.game.PokerEnvStateDictEnums import EnvDictIdxs  # noqa: E402
from PokerRL.game.games import (BigLeduc, DiscretizedNLHoldem, DiscretizedNLLeduc, Flop5Holdem, LimitHoldem,  # noqa: E402
                                NoLimitHoldem, NoLimitLeduc, StandardLeduc)
from PokerRL.game.wrappers import HistoryEnvBuilder  # noqa: E402

META = {"numpy": np.__version__, "generator": "tests/golden/make_golden.py"}


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, meta=json.dumps(META), **arrays)
    print("wrote", path, os.path.getsize(path), "bytes")


# ------------------------------------------------------------------------------------------------------------------
# LUTs (look_up_table.py:191-220)
# ------------------------------------------------------------------------------------------------------------------
def make_luts():
    out = {}
    for cls in (StandardLeduc, BigLeduc, DiscretizedNLHoldem, Flop5Holdem):
        lh = cls.get_lut_holder()
        n = cls.__name__
        out[n + "_IDX_2_HOLE_CARDS"] = lh.LUT_IDX_2_HOLE_CARDS
        out[n + "_HOLE_CARDS_2_IDX"] = lh.LUT_HOLE_CARDS_2_IDX
        out[n + "_CARD_IN_WHAT_RANGE_IDXS"] = lh.LUT_CARD_IN_WHAT_RANGE_IDXS
        out[n + "_1DCARD_2_2DCARD"] = lh.LUT_1DCARD_2_2DCARD
        out[n + "_2DCARD_2_1DCARD"] = lh.LUT_2DCARD_2_1DCARD
    save("luts.npz", **out)


# ------------------------------------------------------------------------------------------------------------------
# hand ranks from the binary evaluator (CppHandeval.py:34-65)
# ------------------------------------------------------------------------------------------------------------------
def _structured_boards():
    c = lambda r, s: r * 4 + s  # noqa: E731
    b = [
        [c(6, 0), c(6, 1), c(6, 2), c(10, 3), c(8, 1)],   # trips on board (survey's quads-quirk example)
        [c(6, 0), c(6, 1), c(6, 2), c(6, 3), c(8, 1)],    # quads on board
        [c(12, 0), c(12, 1), c(12, 2), c(12, 3), c(0, 1)],  # top quads on board
        [c(0, 0), c(0, 1), c(0, 2), c(0, 3), c(12, 1)],   # bottom quads on board
        [c(5, 0), c(5, 1), c(9, 2), c(9, 3), c(9, 1)],    # full house on board
        [c(5, 0), c(5, 1), c(9, 2), c(9, 3), c(2, 1)],    # two pair on board
        [c(0, 0), c(1, 0), c(2, 0), c(3, 0), c(12, 0)],   # wheel straight flush on board
        [c(8, 2), c(9, 2), c(10, 2), c(11, 2), c(12, 2)],  # royal flush on board
        [c(0, 0), c(1, 0), c(2, 0), c(3, 0), c(7, 1)],    # 4 to a straight flush
        [c(1, 1), c(3, 1), c(5, 1), c(7, 1), c(9, 1)],    # 5-flush on board
        [c(1, 1), c(3, 1), c(5, 1), c(7, 1), c(9, 2)],    # 4-flush
        [c(1, 1), c(3, 1), c(5, 1), c(7, 2), c(9, 2)],    # 3-flush
        [c(4, 0), c(5, 1), c(6, 2), c(7, 3), c(8, 0)],    # straight on board
        [c(0, 0), c(1, 1), c(2, 2), c(3, 3), c(12, 0)],   # wheel on board
        [c(9, 0), c(10, 1), c(11, 2), c(12, 3), c(0, 0)],  # broadway draw + deuce
        [c(0, 0), c(2, 1), c(5, 2), c(8, 3), c(11, 0)],   # dry rainbow
        [c(12, 0), c(12, 1), c(11, 2), c(11, 3), c(10, 0)],  # AAKKQ
        [c(3, 0), c(3, 1), c(3, 2), c(7, 3), c(7, 0)],    # 555 99
    ]
    return np.array(b, dtype=np.int8)


def make_handrank():
    lh = DiscretizedNLHoldem.get_lut_holder()
    rules = DiscretizedNLHoldem.RULES()
    rng = np.random.RandomState(1234)
    rnd = np.array([rng.choice(52, 5, replace=False) for _ in range(46)], dtype=np.int8)  # unsorted on purpose
    boards = np.concatenate([_structured_boards(), rnd], axis=0)
    ranks = rules.get_hand_rank_all_hands_on_given_boards(boards_1d=boards, lut_holder=lh)
    assert ranks.dtype == np.int32

    # the survey's hash tripwire: RandomState(0), 20000 boards
    rng0 = np.random.RandomState(0)
    big = np.array([rng0.choice(52, 5, replace=False) for _ in range(20000)], dtype=np.int8)
    import hashlib
    big_ranks = rules.get_hand_rank_all_hands_on_given_boards(boards_1d=big, lut_holder=lh)
    sha = hashlib.sha256(big_ranks.tobytes()).hexdigest()
    print("sha256 of 20000x1326 ranks:", sha)

    # known answers through the single-hand entry point
    def hand(cards):
        return lh.get_2d_cards(np.array(cards, dtype=np.int8))

    c = lambda r, s: r * 4 + s  # noqa: E731
    known = [
        ([c(12, 0), c(11, 1)], [c(10, 2), c(9, 3), c(7, 0), c(3, 1), c(0, 2)]),   # AKQJ9 high
        ([c(12, 0), c(12, 1)], [c(11, 2), c(10, 3), c(9, 0), c(3, 1), c(0, 2)]),  # AA KQJ
        ([c(12, 0), c(12, 1)], [c(11, 2), c(11, 3), c(10, 0), c(3, 1), c(0, 2)]),  # AAKK Q
        ([c(12, 0), c(12, 1)], [c(12, 2), c(11, 3), c(10, 0), c(3, 1), c(0, 2)]),  # AAA KQ
        ([c(12, 0), c(11, 1)], [c(10, 2), c(9, 3), c(8, 0), c(3, 1), c(0, 2)]),   # A-high straight
        ([c(12, 0), c(0, 1)], [c(1, 2), c(2, 3), c(3, 0), c(7, 1), c(9, 2)]),     # wheel
        ([c(12, 0), c(7, 0)], [c(5, 0), c(2, 0), c(0, 0), c(3, 1), c(9, 2)]),     # A9742 flush
        ([c(12, 0), c(12, 1)], [c(12, 2), c(11, 3), c(11, 0), c(3, 1), c(0, 2)]),  # AAA KK
        ([c(12, 0), c(12, 1)], [c(12, 2), c(12, 3), c(11, 0), c(3, 1), c(0, 2)]),  # AAAA K
        ([c(12, 0), c(11, 0)], [c(10, 0), c(9, 0), c(8, 0), c(3, 1), c(0, 2)]),   # royal
        ([c(0, 0), c(1, 0)], [c(2, 0), c(3, 0), c(12, 0), c(3, 1), c(9, 2)]),     # 5-high straight flush
        ([c(3, 1), c(6, 3)], [c(6, 0), c(6, 1), c(6, 2), c(10, 3), c(8, 1)]),     # 8888 quirk: kicker T not Q
    ]
    kh = np.array([k[0] for k in known], dtype=np.int8)
    kb = np.array([k[1] for k in known], dtype=np.int8)
    kr = np.array([rules.get_hand_rank(hand_2d=hand(k[0]), board_2d=hand(k[1])) for k in known], dtype=np.int32)
    save("handrank.npz", boards=boards, ranks=ranks, known_hands=kh, known_boards=kb, known_ranks=kr,
         sha256_rs0_20000=np.array(sha))


def _board_checksum(ranks):
    """u64 checksum of one chunk of ranks [n, 1326] (order-sensitive across hands and boards)."""
    n = ranks.shape[0]
    w = (np.arange(1326, dtype=np.uint64) * np.uint64(2654435761) + np.uint64(12345))
    per_board = ((ranks.astype(np.int64) + 2).astype(np.uint64) * w[None, :]).sum(axis=1, dtype=np.uint64)
    k = (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(1))
    return (per_board * k).sum(dtype=np.uint64)


def _exh_worker(args):
    lo, hi = args
    ref_harness.setup()
    lh = DiscretizedNLHoldem.get_lut_holder()
    rules = DiscretizedNLHoldem.RULES()
    it = itertools.islice(itertools.combinations(range(52), 5), lo, hi)
    boards = np.array(list(it), dtype=np.int8)
    out = []
    for s in range(0, boards.shape[0], 256):
        r = rules.get_hand_rank_all_hands_on_given_boards(boards_1d=boards[s:s + 256], lut_holder=lh)
        out.append(_board_checksum(r))
    return lo, np.array(out, dtype=np.uint64)


def make_handrank_exhaustive():
    """All C(52,5) boards x 1326 hands through the reference binary, reduced to one u64 per chunk of 256 boards."""
    import multiprocessing as mp
    n = 2598960
    per = 256 * 508  # chunk-aligned work items
    jobs = [(lo, min(lo + per, n)) for lo in range(0, n, per)]
    t0 = time.time()
    with mp.Pool(8) as pool:
        res = pool.map(_exh_worker, jobs)
    res.sort(key=lambda x: x[0])
    sums = np.concatenate([r[1] for r in res])
    print("exhaustive sweep", n, "boards in", time.time() - t0, "s;", sums.shape[0], "chunks")
    save("handrank_exhaustive.npz", chunk_boards=np.array(256), checksums=sums)


# ------------------------------------------------------------------------------------------------------------------
# public-tree structure (PublicTree.py:111-293) and env fuzz (PokerEnv.py)
# ------------------------------------------------------------------------------------------------------------------
GAMES = {
    "StandardLeduc": (StandardLeduc, dict(stack=13, bets=bet_sets.POT_ONLY)),
    "BigLeduc": (BigLeduc, dict(stack=100, bets=bet_sets.POT_ONLY)),
    "DiscretizedNLLeduc_POT": (DiscretizedNLLeduc, dict(stack=20000, bets=bet_sets.POT_ONLY)),
    "DiscretizedNLLeduc_B3_short": (DiscretizedNLLeduc, dict(stack=1500, bets=bet_sets.B_3)),
}


def make_bldr(game_cls, stack, bets):
    args = game_cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bets)
    return HistoryEnvBuilder(env_cls=game_cls, env_args=args), args


def flatten_reference_tree(tree):
    """DFS pre-order flattening of the reference's object tree into the arrays of pokerrl_amd's flat tree."""
    lh = tree.env_bldr.lut_holder
    rec = dict(kind=[], actor=[], parent=[], child_idx=[], action=[], acted_last=[], round=[], board_card=[],
               main_pot=[], depth=[], n_children=[], first_col=[])
    col_action = []
    nodes = []

    def visit(node, parent_id, child_idx):
        my = len(nodes)
        nodes.append(node)
        if node.is_terminal:
            kind = 2 if node.action == Poker.FOLD else 3
        elif node.p_id_acting_next == tree.CHANCE_ID:
            kind = 1
        else:
            kind = 0
        rec["kind"].append(kind)
        rec["actor"].append(node.p_id_acting_next if kind == 0 else -1)
        rec["parent"].append(parent_id)
        rec["child_idx"].append(child_idx)
        rec["action"].append(-1 if (node.action == "CHANCE" or node.action is None) else int(node.action))
        al = node.p_id_acted_last
        rec["acted_last"].append(-1 if al is None else (-2 if al == tree.CHANCE_ID else int(al)))
        rec["round"].append(int(node.env_state[EnvDictIdxs.current_round]))
        b1d = lh.get_1d_cards(node.env_state[EnvDictIdxs.board_2d])
        rec["board_card"].append(int(b1d[0]) if b1d[0] != Poker.CARD_NOT_DEALT_TOKEN_1D else -1)
        rec["main_pot"].append(int(node.env_state[EnvDictIdxs.main_pot]))
        rec["depth"].append(int(node.depth))
        rec["n_children"].append(len(node.children))
        if kind == 0:
            rec["first_col"].append(len(col_action))
            col_action.extend(int(a) for a in node.allowed_actions)
            assert len(node.allowed_actions) == len(node.children)
        else:
            rec["first_col"].append(-1)
        for i, c in enumerate(node.children):
            visit(c, my, i)

    visit(tree.root, -1, 0)
    out = {k: np.array(v, dtype=np.int32) for k, v in rec.items()}
    out["col_action"] = np.array(col_action, dtype=np.int32)
    return out, nodes


def make_tree():
    from PokerRL.game._.tree.PublicTree import PublicTree
    for name, (cls, kw) in GAMES.items():
        bldr, args = make_bldr(cls, kw["stack"], kw["bets"])
        tree = PublicTree(env_bldr=bldr, stack_size=args.starting_stack_sizes_list, stop_at_street=None)
        tree.build_tree()
        flat, _ = flatten_reference_tree(tree)
        print(name, "nodes", len(flat["kind"]), "cols", len(flat["col_action"]))
        save("tree_%s.npz" % name, **flat)

    # Flop5Holdem: the reference's board enumeration cannot deal 5 cards, so the betting
    # structure is captured by walking the reference ENV itself (cards are irrelevant to betting).
    flat = walk_env_tree(Flop5Holdem, stack=20000, bets=bet_sets.POT_ONLY)
    print("Flop5Holdem betting structure: nodes", len(flat["kind"]))
    save("tree_Flop5Holdem_1board.npz", **flat)
    make_tree_limit_holdem()


def make_tree_limit_holdem():
    """LimitHoldem (pre-flop, flop, turn, river; full betting structure, 48-chip stacks): the betting tree of ONE run-out, walked through the
    reference ENV street by street (the reference's PublicTree cannot deal several cards) -- what pins the
    multi-street tree structure (csrc/prl_tree.cpp, the per-street engine's street instances) to the reference. 17 221 nodes."""
    sys.setrecursionlimit(20000)
    flat = walk_env_tree(LimitHoldem, stack=48, bets=None)
    print("LimitHoldem betting structure (one run-out): nodes", len(flat["kind"]), "per round", np.bincount(flat["round"]).tolist())
    save("tree_LimitHoldem_1runout.npz", **flat)


def walk_env_tree(cls, stack, bets):
    """Same DFS as PublicTree._build_tree but with ONE chance outcome (whatever the env's deck deals)."""
    bldr, args = make_bldr(cls, stack, bets)
    env = bldr.get_new_env(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
    a = env.get_args()
    a.RETURN_PRE_TRANSITION_STATE_IN_INFO = True
    env.set_args(a)
    np.random.seed(0)
    env.reset()
    rec = dict(kind=[], actor=[], parent=[], child_idx=[], action=[], acted_last=[], round=[], main_pot=[], depth=[],
               n_children=[], first_col=[])
    col_action = []

    def add(kind, actor, parent, child_idx, action, acted_last, rnd, pot, depth):
        for k, v in zip(("kind", "actor", "parent", "child_idx", "action", "acted_last", "round", "main_pot", "depth"),
                        (kind, actor, parent, child_idx, action, acted_last, rnd, pot, depth)):
            rec[k].append(v)
        rec["n_children"].append(0)
        rec["first_col"].append(-1)
        return len(rec["kind"]) - 1

    def expand(my, state, depth):
        env.load_state_dict(state)
        legal = env.get_legal_actions()
        rec["n_children"][my] = len(legal)
        rec["first_col"][my] = len(col_action)
        col_action.extend(int(x) for x in legal)
        actor = state[EnvDictIdxs.current_player]
        rnd = state[EnvDictIdxs.current_round]
        for i, act in enumerate(legal):
            env.load_state_dict(state)
            _, _, term, info = env.step(act)
            if term:
                pre = info["state_dict_before_money_move"]
                add(2 if act == Poker.FOLD else 3, -1, my, i, int(act), actor, rnd, int(pre[EnvDictIdxs.main_pot]), depth + 1)
            elif info["chance_acts"]:
                pre = info["state_dict_before_money_move"]
                ch = add(1, -1, my, i, int(act), actor, rnd, int(pre[EnvDictIdxs.main_pot]), depth + 1)
                rec["n_children"][ch] = 1
                st = env.state_dict()
                c = add(0, st[EnvDictIdxs.current_player], ch, 0, -1, -2, st[EnvDictIdxs.current_round],
                        int(st[EnvDictIdxs.main_pot]), depth + 2)
                expand(c, st, depth + 2)
            else:
                st = env.state_dict()
                c = add(0, st[EnvDictIdxs.current_player], my, i, int(act), actor, st[EnvDictIdxs.current_round],
                        int(st[EnvDictIdxs.main_pot]), depth + 1)
                expand(c, st, depth + 1)

    st0 = env.state_dict()
    root = add(0, st0[EnvDictIdxs.current_player], -1, 0, -1, -1, st0[EnvDictIdxs.current_round],
               int(st0[EnvDictIdxs.main_pot]), 0)
    expand(root, st0, 0)
    out = {k: np.array(v, dtype=np.int32) for k, v in rec.items()}
    out["col_action"] = np.array(col_action, dtype=np.int32)
    return out


def walk_env_tree_runouts(cls, stack, bets, runouts):
    """The betting tree below SEVERAL run-outs, walked through the REFERENCE env: the DFS of walk_env_tree, but at every deal the env is
    reloaded once per distinct outcome among the listed run-outs that continue the cards already out (row order), its deck set so that it
    deals exactly those cards -- the board of every node is what the env then holds. What pins the product's multi-run-out wiring (chance
    children per prefix, board rows, parents) to the reference, not only the betting tree of one run-out."""
    bldr, args = make_bldr(cls, stack, bets)
    env = bldr.get_new_env(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
    a = env.get_args()
    a.RETURN_PRE_TRANSITION_STATE_IN_INFO = True
    env.set_args(a)
    np.random.seed(0)
    env.reset()
    runouts = [[int(c) for c in r] for r in np.asarray(runouts)]
    n_board = len(runouts[0])
    rec = dict(kind=[], actor=[], parent=[], child_idx=[], action=[], acted_last=[], round=[], main_pot=[], depth=[], n_children=[], first_col=[])
    boards = []
    col_action = []
    lh = env.lut_holder

    def board_now():
        b = [int(c) for c in lh.get_1d_cards(env.board)]
        return [c if c >= 0 else -1 for c in b] + [-1] * (n_board - len(b))

    def add(kind, actor, parent, child_idx, action, acted_last, rnd, pot, depth, board):
        for k, v in zip(("kind", "actor", "parent", "child_idx", "action", "acted_last", "round", "main_pot", "depth"),
                        (kind, actor, parent, child_idx, action, acted_last, rnd, pot, depth)):
            rec[k].append(v)
        rec["n_children"].append(0)
        rec["first_col"].append(-1)
        boards.append(list(board))
        return len(rec["kind"]) - 1

    def expand(my, state, depth, board):
        env.load_state_dict(state)
        legal = env.get_legal_actions()
        rec["n_children"][my] = len(legal)
        rec["first_col"][my] = len(col_action)
        col_action.extend(int(x) for x in legal)
        actor = state[EnvDictIdxs.current_player]
        rnd = state[EnvDictIdxs.current_round]
        n_out = sum(c >= 0 for c in board)
        for i, act in enumerate(legal):
            env.load_state_dict(state)
            _, _, term, info = env.step(act)
            if term:
                pre = info["state_dict_before_money_move"]
                add(2 if act == Poker.FOLD else 3, -1, my, i, int(act), actor, rnd, int(pre[EnvDictIdxs.main_pot]), depth + 1, board)
            elif info["chance_acts"]:
                pre = info["state_dict_before_money_move"]
                n_dealt = sum(c >= 0 for c in board_now()) - n_out  # cards this street deals
                outs = []
                for r in runouts:  # distinct continuations of the cards already out, in row order
                    if r[:n_out] == [c for c in board if c >= 0] and r[n_out:n_out + n_dealt] not in outs:
                        outs.append(r[n_out:n_out + n_dealt])
                ch = add(1, -1, my, i, int(act), actor, rnd, int(pre[EnvDictIdxs.main_pot]), depth + 1, board)
                rec["n_children"][ch] = len(outs)
                for k, cards in enumerate(outs):
                    env.load_state_dict(state)
                    # the env deals from the top of its deck: put this outcome's cards there
                    top = lh.get_2d_cards(np.array(cards, np.int32))
                    rest = np.array([c for c in env.deck.deck_remaining if not any((c == t).all() for t in top)], np.int8).reshape(-1, 2)
                    env.deck.deck_remaining = np.concatenate([top.astype(np.int8), rest], axis=0)
                    env.step(act)
                    st = env.state_dict()
                    nb = board_now()
                    assert [c for c in nb if c >= 0] == [c for c in board if c >= 0] + cards, (nb, board, cards)
                    c = add(0, st[EnvDictIdxs.current_player], ch, k, -1, -2, st[EnvDictIdxs.current_round], int(st[EnvDictIdxs.main_pot]), depth + 2, nb)
                    expand(c, st, depth + 2, nb)
            else:
                st = env.state_dict()
                c = add(0, st[EnvDictIdxs.current_player], my, i, int(act), actor, st[EnvDictIdxs.current_round], int(st[EnvDictIdxs.main_pot]), depth + 1, board)
                expand(c, st, depth + 1, board)

    st0 = env.state_dict()
    b0 = [-1] * n_board
    root = add(0, st0[EnvDictIdxs.current_player], -1, 0, -1, -1, st0[EnvDictIdxs.current_round], int(st0[EnvDictIdxs.main_pot]), 0, b0)
    expand(root, st0, 0, b0)
    out = {k: np.array(v, dtype=np.int32) for k, v in rec.items()}
    out["col_action"] = np.array(col_action, dtype=np.int32)
    out["board"] = np.array(boards, dtype=np.int8)
    return out


def make_tree_limit_holdem_runouts():
    """LimitHoldem under 2 flops x 2 turns x 2 rivers (the run-outs of tests/parity_cases.multistreet_runouts(2, 2, 2)): every node's protocol
    fields AND the board the reference env holds there"""
    sys.path.insert(0, os.path.dirname(HERE))                     # tests/
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))    # the repository root (oracle/, pokerrl_amd/)
    import parity_cases as pc
    sys.setrecursionlimit(20000)
    ro = pc.multistreet_runouts(2, 2, 2)
    flat = walk_env_tree_runouts(LimitHoldem, stack=48, bets=None, runouts=ro)
    print("LimitHoldem 2x2x2 run-outs: nodes", len(flat["kind"]), "per round", np.bincount(flat["round"]).tolist())
    save("tree_LimitHoldem_2x2x2.npz", runouts=ro, **flat)


ENV_FUZZ = {
    "StandardLeduc": (StandardLeduc, 13, [0.0]),
    "BigLeduc": (BigLeduc, 100, [0.0]),
    "BigLeduc_short": (BigLeduc, 9, [0.0]),
    "NoLimitLeduc_short": (NoLimitLeduc, 700, [0.0]),
    "DiscretizedNLLeduc_B3": (DiscretizedNLLeduc, 20000, bet_sets.B_3),
    "DiscretizedNLLeduc_B5_short": (DiscretizedNLLeduc, 900, bet_sets.B_5),
    "LimitHoldem": (LimitHoldem, 48, [0.0]),
    "LimitHoldem_short": (LimitHoldem, 11, [0.0]),
    "DiscretizedNLHoldem_B5": (DiscretizedNLHoldem, 20000, bet_sets.B_5),
    "DiscretizedNLHoldem_OT11_short": (DiscretizedNLHoldem, 2300, bet_sets.OFF_TREE_11),
    "NoLimitHoldem_short": (NoLimitHoldem, 1700, [0.0]),
    "Flop5Holdem": (Flop5Holdem, 20000, [0.0]),
    "Flop5Holdem_short": (Flop5Holdem, 1100, [0.0]),
}


def _pub_state(env):
    s = env.state_dict()
    seats = s[EnvDictIdxs.seats]
    cr = s[EnvDictIdxs.capped_raise]
    la = s[EnvDictIdxs.last_action]
    none = lambda v: -1 if v is None else int(v)  # noqa: E731
    return [int(s[EnvDictIdxs.current_round]), int(s[EnvDictIdxs.main_pot]),
            int(seats[0]["current_bet"]), int(seats[1]["current_bet"]),
            int(round(float(seats[0]["stack"]))), int(round(float(seats[1]["stack"]))),
            int(seats[0]["is_allin"]), int(seats[1]["is_allin"]),
            int(seats[0]["folded_this_episode"]), int(seats[1]["folded_this_episode"]),
            int(seats[0]["has_acted_this_round"]), int(seats[1]["has_acted_this_round"]),
            none(s[EnvDictIdxs.current_player]), none(s[EnvDictIdxs.last_raiser]),
            0 if cr is None else 1, -1 if cr is None else none(cr[0]), -1 if cr is None else none(cr[1]),
            int(s[EnvDictIdxs.n_actions_this_episode]),
            int(s.get(EnvDictIdxs.n_raises_this_round, 0) or 0),
            none(la[0]), none(la[1]), none(la[2])]


def make_env():
    """Random legal + illegal action sequences through the reference env; records public state after every step."""
    out = {}
    for name, (cls, stack, bets) in ENV_FUZZ.items():
        bldr, args = make_bldr(cls, stack, bets)
        env = bldr.get_new_env(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
        a = env.get_args()
        a.RETURN_PRE_TRANSITION_STATE_IN_INFO = True
        env.set_args(a)
        rng = np.random.RandomState(abs(hash(name)) % (2 ** 31) if False else sum(map(ord, name)))
        np.random.seed(sum(map(ord, name)))
        is_nl = cls in (NoLimitLeduc, NoLimitHoldem)
        # one row per event: [episode, kind(0 reset / 1 step), action, amount, n_legal, legal..., is_term, chance, pot_before, state...]
        rows = []
        n_act = env.N_ACTIONS
        MAXL = 16
        for ep in range(150):
            env.reset()
            legal = env.get_legal_actions()
            rows.append([ep, 0, -1, -1, len(legal)] + list(legal)[:MAXL] + [-1] * (MAXL - min(len(legal), MAXL)) + [0, 0, 0]
                        + _pub_state(env))
            done = False
            while not done:
                if rng.rand() < 0.7:
                    act = int(legal[rng.randint(len(legal))])
                else:
                    act = int(rng.randint(n_act))  # possibly illegal -> exercises _get_fixed_action
                amount = -1
                if is_nl:  # NoLimit envs take (type, chips) tuples (PokerEnv.step -> _step)
                    amount = int(rng.randint(0, 2 * stack + 2))
                    _, _, done, info = env.step((act, amount))
                else:
                    _, _, done, info = env.step(act)
                pre = info["state_dict_before_money_move"]
                pot_before = int(pre[EnvDictIdxs.main_pot]) if (pre is not None and done) else 0
                if done:
                    legal = []
                    st = [0] * 22
                else:
                    legal = env.get_legal_actions()
                    st = _pub_state(env)
                rows.append([ep, 1, act, amount, len(legal)] + list(legal)[:MAXL] + [-1] * (MAXL - min(len(legal), MAXL))
                            + [int(done), int(bool(info["chance_acts"])), pot_before] + st)
        arr = np.array(rows, dtype=np.int64)
        out[name] = arr
        out[name + "_cfg"] = np.array([stack] + [int(round(b * 1000)) for b in sorted(bets)], dtype=np.int64)
        print(name, arr.shape)
    save("env_fuzz.npz", **out)


# ------------------------------------------------------------------------------------------------------------------
# CFR per-node dumps + exploitability series (PokerRL/cfr/*.py), BR of a uniform agent (eval/br/LocalBRMaster.py)
# ------------------------------------------------------------------------------------------------------------------
def _dfs_nodes(root):
    out = []

    def visit(n):
        out.append(n)
        for c in n.children:
            visit(c)

    visit(root)
    return out


def _h32(a):
    import hashlib
    a = np.ascontiguousarray(a)
    return hashlib.sha256((a + a.dtype.type(0)).tobytes()).hexdigest()  # "+ 0" folds -0.0 into +0.0


def snapshot(tree, with_cfr_data):
    """Column-major [n_cols, R] / node-major [n_nodes, 2, R] arrays in DFS pre-order, as pokerrl_amd lays them out."""
    nodes = _dfs_nodes(tree.root)
    R = tree.env_bldr.rules.RANGE_SIZE
    n = len(nodes)
    reach = np.zeros((n, 2, R), np.float32)
    ev = np.zeros((n, 2, R), np.float32)
    ev_br = np.zeros((n, 2, R), np.float32)
    strat, strat_f64, regret, avg, avg_f64 = [], [], [], [], []
    for i, nd in enumerate(nodes):
        reach[i] = nd.reach_probs
        ev[i] = nd.ev
        ev_br[i] = nd.ev_br
        assert nd.reach_probs.dtype == np.float32 and nd.ev.dtype == np.float32 and nd.ev_br.dtype == np.float32
        is_dec = (not nd.is_terminal) and nd.p_id_acting_next != tree.CHANCE_ID
        if is_dec:
            assert nd.strategy.dtype in (np.float32, np.float64)
            strat.append(nd.strategy.T.astype(np.float64))
            strat_f64.append(int(nd.strategy.dtype == np.float64))
            if with_cfr_data:
                rg = nd.data["regret"]
                regret.append(np.zeros_like(nd.strategy.T, dtype=np.float32) if rg is None else rg.T)
                if rg is not None:
                    assert rg.dtype == np.float32, rg.dtype
                av = nd.data["avg_strat"]
                avg.append(np.zeros_like(nd.strategy.T, dtype=np.float64) if av is None else av.T.astype(np.float64))
                avg_f64.append(0 if av is None else int(av.dtype == np.float64))
        else:
            strat_f64.append(0)
            avg_f64.append(0)
    out = dict(reach=reach, ev=ev, ev_br=ev_br, strategy=np.concatenate(strat, axis=0),
               strat_f64=np.array(strat_f64, np.uint8),
               exploitability=np.array(tree.root.exploitability, np.float32))
    if with_cfr_data:
        out.update(regret=np.concatenate(regret, axis=0), avg=np.concatenate(avg, axis=0),
                   avg_f64=np.array(avg_f64, np.uint8))
    return out


CFR_RUNS = [
    # (fixture name, game key, algo, n_iters, snapshot iterations, store full arrays?)
    ("StandardLeduc_CFRPlus", "StandardLeduc", "CFRPlus", 10, (1, 2, 3, 10), True),
    ("StandardLeduc_VanillaCFR", "StandardLeduc", "VanillaCFR", 10, (1, 2, 10), True),
    ("StandardLeduc_LinearCFR", "StandardLeduc", "LinearCFR", 10, (1, 2, 10), True),
    ("DiscretizedNLLeduc_POT_CFRPlus", "DiscretizedNLLeduc_POT", "CFRPlus", 10, (1, 2, 10), True),
    ("DiscretizedNLLeduc_POT_LinearCFR", "DiscretizedNLLeduc_POT", "LinearCFR", 5, (1, 5), False),
    ("DiscretizedNLLeduc_B3_short_VanillaCFR", "DiscretizedNLLeduc_B3_short", "VanillaCFR", 4, (1, 4), False),
    ("BigLeduc_CFRPlus", "BigLeduc", "CFRPlus", 2, (1, 2), False),
]


def make_cfr(only=None):
    from PokerRL.cfr.CFRPlus import CFRPlus
    from PokerRL.cfr.LinearCFR import LinearCFR
    from PokerRL.cfr.VanillaCFR import VanillaCFR
    from PokerRL.rl.base_cls.workers.ChiefBase import ChiefBase
    algos = dict(CFRPlus=CFRPlus, LinearCFR=LinearCFR, VanillaCFR=VanillaCFR)
    for fx, gkey, algo, n_iters, snaps, full in CFR_RUNS:
        if only and fx not in only:
            continue
        cls, kw = GAMES[gkey]
        chief = ChiefBase(t_prof=None)
        kwargs = dict(name="g", game_cls=cls, agent_bet_set=kw["bets"], chief_handle=chief,
                      starting_stack_sizes=[kw["stack"]])
        if algo == "CFRPlus":
            kwargs["delay"] = 0
        t0 = time.time()
        cfr = algos[algo](**kwargs)
        out = {}

        def put(prefix, snap):
            for k, v in snap.items():
                if full or k in ("exploitability", "strat_f64", "avg_f64"):
                    out[prefix + k] = v
                else:
                    out[prefix + k + "_sha256"] = np.array(_h32(v))

        put("it0_", snapshot(cfr._trees[0], with_cfr_data=False))
        for it in range(1, n_iters + 1):
            cfr.iteration()
            if it in snaps:
                put("it%d_" % it, snapshot(cfr._trees[0], with_cfr_data=True))
        vals = chief.get_new_values()[0]
        name_curr = [k for k in vals if "_Curr_S" in k][0]
        name_avg = [k for k in vals if "_Avg_total_S" in k][0]
        graph = list(vals[name_curr].keys())[0]
        out["curr_series"] = np.array(vals[name_curr][graph], dtype=np.float64)   # [[iter, value], ...]
        out["avg_series"] = np.array(vals[name_avg][graph], dtype=np.float64)
        out["ev_normalizer"] = np.array(cls.EV_NORMALIZER, dtype=np.float64)
        print(fx, "done in %.1fs" % (time.time() - t0), "curr", out["curr_series"][-1], "avg", out["avg_series"][-1])
        save("cfr_%s.npz" % fx, **out)


def make_env_obs():
    """Full env episodes (cards, observation vectors, rewards) for seeded decks: pins the PokerEnv facade's dealing
    order (_Deck.py), observation layout (PokerEnv.py:199-261,1253-1271), payouts (:468-481) and rewards (:1069-1072)."""
    out = {}
    for name in ("StandardLeduc", "BigLeduc_short", "DiscretizedNLLeduc_B5_short", "LimitHoldem", "DiscretizedNLHoldem_B5",
                 "DiscretizedNLHoldem_OT11_short", "Flop5Holdem"):
        cls, stack, bets = ENV_FUZZ[name]
        bldr, args = make_bldr(cls, stack, bets)
        env = bldr.get_new_env(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
        rng = np.random.RandomState(sum(map(ord, name)) + 7)
        obs_rows, meta_rows = [], []
        for ep in range(40):
            np.random.seed(1000 + ep)
            o, r, done, _ = env.reset()
            obs_rows.append(o)
            meta_rows.append([ep, -1, 0, 0.0, 0.0])
            while not done:
                legal = env.get_legal_actions()
                act = int(legal[rng.randint(len(legal))])
                o, r, done, _ = env.step(act)
                obs_rows.append(o)
                meta_rows.append([ep, act, int(done), float(r[0]), float(r[1])])
            cards = [int(c) for c in env.lut_holder.get_1d_cards(env.board)] + \
                    [int(c) for p in range(2) for c in env.lut_holder.get_1d_cards(env.seats[p].hand)]
            meta_rows[-1] = meta_rows[-1] + cards
        width = max(len(m) for m in meta_rows)
        meta = np.array([m + [-999] * (width - len(m)) for m in meta_rows], dtype=np.float64)
        out[name + "_obs"] = np.array(obs_rows, dtype=np.float32)
        out[name + "_meta"] = meta
        print(name, out[name + "_obs"].shape)
    save("env_obs.npz", **out)


if __name__ == "__main__":
    what = sys.argv[1:] or ["luts", "handrank", "tree", "env", "cfr"]
    fns = {"luts": make_luts, "handrank": make_handrank, "handrank_exhaustive": make_handrank_exhaustive,
           "tree": make_tree, "tree_lh": make_tree_limit_holdem, "tree_lh_runouts": make_tree_limit_holdem_runouts, "env": make_env, "cfr": make_cfr, "env_obs": make_env_obs}
    i = 0
    while i < len(what):
        w = what[i]
        if w == "cfr" and i + 1 < len(what) and what[i + 1] not in fns:
            make_cfr(only=what[i + 1].split(","))
            i += 2
            continue
        fns[w]()
        i += 1
