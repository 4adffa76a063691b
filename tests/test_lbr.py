"""Local Best Response: pokerrl_amd.eval.lbr.LocalLBRWorker against the per-hand winnings
the REFERENCE's LocalLBRWorker produced with the same fixture agent, decks (np.random seed) and action draws
(tests/golden/make_lbr_golden.py). Bit-exact float32. CPU: the worker's equity calls go to the emulator build of the
library (same kernel sources); GPU: the product library."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import lbr_fixture_agent as fx  # noqa: E402
from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.eval.lbr import BatchedLBR, LBRArgs, LocalLBRMaster, LocalLBRWorker  # noqa: E402
from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game.games import DiscretizedNLHoldem, DiscretizedNLLeduc, StandardLeduc  # noqa: E402
from pokerrl_amd.game.Poker import Poker  # noqa: E402
from pokerrl_amd.game.wrappers import HistoryEnvBuilder  # noqa: E402
from pokerrl_amd.rl.base_cls.EvalAgentBase import EvalAgentBase  # noqa: E402
from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase  # noqa: E402
from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase  # noqa: E402

CASES = {
    "StandardLeduc": (StandardLeduc, None, dict(lbr_check_to_round=None)),
    "DiscretizedNLLeduc": (DiscretizedNLLeduc, bet_sets.B_3, dict(lbr_bet_set=bet_sets.B_5, lbr_check_to_round=None)),
    "DiscretizedNLHoldem": (DiscretizedNLHoldem, bet_sets.B_5, dict(lbr_bet_set=bet_sets.OFF_TREE_11, lbr_check_to_round=Poker.TURN)),
    "DiscretizedNLHoldem_flop": (DiscretizedNLHoldem, bet_sets.B_3, dict(lbr_bet_set=bet_sets.B_5, lbr_check_to_round=Poker.FLOP)),
}


def make_t_prof(game_cls, agent_bets, lbr_kwargs, n_hands, path):
    env_args = game_cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=agent_bets) if agent_bets is not None else game_cls.ARGS_CLS(n_seats=2)
    return TrainingProfileBase(
        name="lbr", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=game_cls,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH",), eval_stack_sizes=None,
        module_args={"env": env_args, "lbr": LBRArgs(n_lbr_hands_per_seat=n_hands, **lbr_kwargs)}, path_data=str(path))


def check_case(tag, tmp_path, max_hands=None):
    game_cls, agent_bets, lbr_kwargs = CASES[tag]
    g = np.load(os.path.join(HERE, "golden", "lbr_%s.npz" % tag))
    n = int(g["n_hands"]) if max_hands is None else min(int(g["n_hands"]), max_hands)
    t_prof = make_t_prof(game_cls, agent_bets, lbr_kwargs, n, tmp_path)
    record = []
    w = LocalLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=7, record=record))
    for seat in (0, 1):
        np.random.seed(int(g["np_seed"]) + seat)
        n0 = len(record)
        got = w.run(agent_seat_id=seat, n_iterations=n, mode="HASH", stack_size=[game_cls.DEFAULT_STACK_SIZE] * 2)
        hands = np.stack([np.stack(d["hand"]) for d in record[n0:]]).astype(np.int8)
        assert np.array_equal(hands, g["hands_agent_seat%d" % seat][:n]), "decks diverged"
        want = g["winnings_agent_seat%d" % seat][:n]
        assert got.dtype == np.float32
        assert np.array_equal(got, want), "%s seat %d: %d of %d hands differ" % (tag, seat, int(np.sum(got != want)), n)
    assert w.n_equity_calls > 0
    return w


def _rank_fn(tag):
    from oracle import rank_boards
    if tag == "StandardLeduc":
        return lambda board: np.array([100 + c // 2 if c // 2 == board[0] // 2 else c // 2 for c in range(6)], dtype=np.int32)
    return lambda board: rank_boards(np.array([board], dtype=np.int8))[0]


def _game_dims(tag):
    return (1, 6, 1, StandardLeduc) if tag == "StandardLeduc" else (2, 52, 5, DiscretizedNLHoldem)


def check_equity_oracle_vs_golden(tag, max_to_deal=2):
    """the NumPy restatement (oracle/lbr.py) reproduces the reference's rollout manager bit for bit"""
    from oracle.lbr import checkdown_equity
    g = np.load(os.path.join(HERE, "golden", "lbr_equity.npz"))
    n_hole, n_cards, n_board, _ = _game_dims(tag)
    n = 0
    for board, nd, hand, rng, wp in zip(g[tag + "_board"], g[tag + "_n_dealt"], g[tag + "_hand"], g[tag + "_range"], g[tag + "_wp"]):
        if n_board - int(nd) > max_to_deal:
            continue
        got = checkdown_equity(_rank_fn(tag), n_hole, n_cards, n_board, board[:nd], hand[:n_hole], rng)
        assert got == wp or (np.isnan(got) and np.isnan(wp)), (tag, board, nd, got, wp)
        n += 1
    assert n > 10


def check_equity_kernel(L, tag, extra_random=0, max_to_deal=2):
    """prl_lbr_checkdown_equity against the reference's outputs (golden) and, on fresh seeded ranges, against the oracle"""
    import ctypes
    from oracle.lbr import checkdown_equity
    g = np.load(os.path.join(HERE, "golden", "lbr_equity.npz"))
    n_hole, n_cards, n_board, game_cls = _game_dims(tag)
    rules = game_cls.native_rules()

    def native(board_dealt, hand, ranges):
        b = np.ascontiguousarray(board_dealt, dtype=np.int8)
        h = np.ascontiguousarray(hand, dtype=np.int8)
        r = np.ascontiguousarray(ranges, dtype=np.float32)
        out = np.zeros(r.shape[0], np.float32)
        assert L.prl_lbr_checkdown_equity(ctypes.byref(rules), b.ctypes.data_as(ctypes.c_void_p), int(b.shape[0]), h.ctypes.data_as(ctypes.c_void_p),
                                          r.ctypes.data_as(ctypes.c_void_p), int(r.shape[0]), out.ctypes.data_as(ctypes.c_void_p)) == 0
        return out

    n = 0
    for board, nd, hand, rng, wp in zip(g[tag + "_board"], g[tag + "_n_dealt"], g[tag + "_hand"], g[tag + "_range"], g[tag + "_wp"]):
        if n_board - int(nd) > max_to_deal:
            continue
        got = native(board[:nd], hand[:n_hole], rng[None, :])[0]
        assert got == wp or (np.isnan(got) and np.isnan(wp)), (tag, board, nd, got, wp)
        n += 1
    assert n > 5
    rs = np.random.RandomState(3)
    for _ in range(extra_random):  # several candidate ranges per call, incl. an all-zero one (-> uniform)
        cards = rs.choice(n_cards, n_hole + n_board, replace=False)
        nd = int(rs.randint(max(0, n_board - max_to_deal), n_board + 1))
        hand, board = np.sort(cards[:n_hole]), cards[n_hole:n_hole + nd]
        R = 6 if n_hole == 1 else 1326
        ranges = (rs.random_sample((4, R)) ** 4).astype(np.float32)
        ranges[3] = 0
        got = native(board, hand, ranges)
        for q in range(4):
            want = checkdown_equity(_rank_fn(tag), n_hole, n_cards, n_board, board, hand, ranges[q])
            assert got[q] == want or (np.isnan(got[q]) and np.isnan(want)), (tag, q, got[q], want)


def check_equity_many_cards_to_come(L, n_to_deal, n_ranges, seed=4):
    """three to five board cards to come (hold'em before the flop = 5: all C(50, 5) run-outs, LocalLBRWorker.py:388-425): the equity
    kernels' deep reduction against the oracle's restatement of the reference's recursion, on properly prepared agent ranges (LBR's cards
    and the board removed, normalised -- what LocalLBRWorker hands over)"""
    import ctypes
    from oracle.lbr import checkdown_equity
    from pokerrl_amd.game import games as G
    rules = G.DiscretizedNLHoldem.native_rules()
    rs = np.random.RandomState(seed + n_to_deal)
    cards = rs.choice(52, 2 + 5 - n_to_deal, replace=False)
    hand, board = np.sort(cards[:2]).astype(np.int8), cards[2:].astype(np.int8)
    ranges = (rs.random_sample((n_ranges, 1326)) ** 3).astype(np.float32)
    c1, c2 = np.triu_indices(52, 1)
    dead = np.isin(c1, list(cards)) | np.isin(c2, list(cards))
    ranges[:, dead] = 0
    ranges /= ranges.sum(axis=1, keepdims=True)
    out = np.zeros(n_ranges, np.float32)
    assert L.prl_lbr_checkdown_equity(ctypes.byref(rules), board.ctypes.data_as(ctypes.c_void_p), int(board.shape[0]), hand.ctypes.data_as(ctypes.c_void_p),
                                      ranges.ctypes.data_as(ctypes.c_void_p), n_ranges, out.ctypes.data_as(ctypes.c_void_p)) == 0
    for q in range(n_ranges):
        want = checkdown_equity(_rank_fn("DiscretizedNLHoldem"), 2, 52, 5, board, hand, ranges[q])
        assert out[q] == want and 0.0 < float(out[q]) < 1.0, (n_to_deal, q, out[q], want)
    return out


def test_emu_lbr_equity_three_cards_to_come(emu_lib):
    check_equity_many_cards_to_come(emu_lib, 3, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("n_to_deal,n_ranges", [(3, 3), (4, 2)])
def test_gpu_lbr_equity_before_the_flop(n_to_deal, n_ranges):
    """three and four board cards to come on the device against the oracle (five -- a hold'em decision before the flop, 2 118 760 run-outs per
    range -- is checked against the REFERENCE's own numbers below)"""
    _native.require_device()
    check_equity_many_cards_to_come(_native.lib(), n_to_deal, n_ranges)


def _preflop_fixture():
    path = os.path.join(HERE, "golden", "lbr_equity_preflop.npz")
    if not os.path.isfile(path):
        pytest.skip("fixture not generated (tests/golden/make_lbr_equity_preflop_golden.py: ~20 GB of RAM, the reference's rollout manager before the flop)")
    return np.load(path)


@pytest.mark.gpu
def test_gpu_lbr_equity_before_the_flop_vs_reference():
    """the REFERENCE's _LBRRolloutManager at hold'em's first decision (five cards to come, 1 712 304 run-outs; lbr_equity_preflop.npz):
    prl_lbr_checkdown_equity returns its numbers bit for bit -- and so does the oracle's restatement of the recursion for the first range"""
    import ctypes
    from oracle.lbr import checkdown_equity
    from pokerrl_amd.game import games as G
    _native.require_device()
    g = _preflop_fixture()
    L = _native.lib()
    rules = G.DiscretizedNLHoldem.native_rules()
    hand = np.ascontiguousarray(g["hand"], dtype=np.int8)
    ranges = np.ascontiguousarray(g["range"], dtype=np.float32)
    board = np.zeros(0, np.int8)
    out = np.zeros(ranges.shape[0], np.float32)
    assert L.prl_lbr_checkdown_equity(ctypes.byref(rules), board.ctypes.data_as(ctypes.c_void_p), 0, hand.ctypes.data_as(ctypes.c_void_p),
                                      ranges.ctypes.data_as(ctypes.c_void_p), int(ranges.shape[0]), out.ctypes.data_as(ctypes.c_void_p)) == 0
    assert np.array_equal(out, g["wp"]), (out, g["wp"])
    want = checkdown_equity(_rank_fn("DiscretizedNLHoldem"), 2, 52, 5, board, hand, ranges[0])
    assert want == g["wp"][0], (want, g["wp"][0])


def test_lbr_equity_oracle_vs_reference_before_the_flop():
    """oracle/lbr.py's recursion for any number of cards to come = the reference's rollout manager at hold'em's first decision (one of the
    fixture's ranges here, ~1 minute of NumPy; the GPU suite checks the kernel against all of them)"""
    from oracle.lbr import checkdown_equity
    g = _preflop_fixture()
    assert int(g["n_runouts"]) == 2118760  # C(50, 5): LBR's own cards are out, the agent's are not known
    want = checkdown_equity(_rank_fn("DiscretizedNLHoldem"), 2, 52, 5, np.zeros(0, np.int8), g["hand"], g["range"][2])
    assert want == g["wp"][2], (want, g["wp"][2])


def test_lbr_equity_oracle_vs_reference_golden():
    check_equity_oracle_vs_golden("StandardLeduc")
    check_equity_oracle_vs_golden("DiscretizedNLHoldem", max_to_deal=1)  # the flop cases (990 boards) run in the GPU suite


def test_lbr_equity_kernel_emu():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    L = _native.bind(build_emu.build())
    check_equity_kernel(L, "StandardLeduc", extra_random=20)
    check_equity_kernel(L, "DiscretizedNLHoldem", extra_random=2, max_to_deal=1)


@pytest.mark.gpu
def test_gpu_lbr_equity_kernel_vs_reference_and_oracle():
    check_equity_oracle_vs_golden("DiscretizedNLHoldem")
    check_equity_kernel(_native.lib(), "StandardLeduc", extra_random=40)
    check_equity_kernel(_native.lib(), "DiscretizedNLHoldem", extra_random=6)


def decks_from_record(record, lut, n_board):
    """cards of every recorded episode in the batched engine's layout: seat 0's hole cards, seat 1's, the board in deal order"""
    out = []
    for d in record:
        hole = [lut.get_1d_cards(np.asarray(h)) for h in d["hand"]]
        board = lut.get_1d_cards(np.asarray(d["deck"]["deck_remaining"])[:n_board])
        out.append(np.concatenate([hole[0], hole[1], board]).astype(np.int8))
    return np.stack(out)


def check_batched_vs_golden(tag, tmp_path, max_hands=None):
    """the device-resident engine plays the golden hands (same decks, same agent draws): per-hand winnings bit-identical to
    the reference's LocalLBRWorker"""
    game_cls, agent_bets, lbr_kwargs = CASES[tag]
    g = np.load(os.path.join(HERE, "golden", "lbr_%s.npz" % tag))
    n = int(g["n_hands"]) if max_hands is None else min(int(g["n_hands"]), max_hands)
    t_prof = make_t_prof(game_cls, agent_bets, lbr_kwargs, n, tmp_path)
    b = BatchedLBR(t_prof, agent_kind="hash", agent_seed=7)
    lut = game_cls.get_lut_holder()
    # the decks of the golden run: replay the reference's shuffles with the facade env (same np.random consumption)
    record = []
    w = LocalLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=7, record=record))
    w._lbr_action = lambda **kw: 1  # decks only: LBR just calls, no equity work
    for seat in (0, 1):
        np.random.seed(int(g["np_seed"]) + seat)
        n0 = len(record)
        w.run(agent_seat_id=seat, n_iterations=n, mode="HASH", stack_size=[game_cls.DEFAULT_STACK_SIZE] * 2)
        decks = decks_from_record(record[n0:], lut, game_cls.RULES.N_TOTAL_BOARD_CARDS if hasattr(game_cls.RULES, "N_TOTAL_BOARD_CARDS") else b.n_deal - 2 * b._rules.n_hole_cards)
        got = b.run(agent_seat_id=seat, n_hands=n, decks=decks)
        want = g["winnings_agent_seat%d" % seat][:n]
        assert np.array_equal(got, want), "%s seat %d: %d of %d hands differ (first at %s)" % (
            tag, seat, int(np.sum(got != want)), n, np.flatnonzero(got != want)[:5])
        assert b.last_stats["env_steps"] > n and b.last_stats["agent_actions"] > 0


def test_batched_lbr_standard_leduc_vs_reference_emu(emu_lib, tmp_path):
    check_batched_vs_golden("StandardLeduc", tmp_path, max_hands=60)


def test_batched_lbr_nl_leduc_vs_reference_emu(emu_lib, tmp_path):
    check_batched_vs_golden("DiscretizedNLLeduc", tmp_path, max_hands=40)


@pytest.mark.parametrize("tag", ["StandardLeduc", "DiscretizedNLLeduc"])
def test_batched_lbr_equity_cache_and_replay_rounds_emu(emu_lib, monkeypatch, tag, tmp_path):
    """the machinery of LBR decisions with many board cards to come (hold'em before the flop: the equities cached per (public history, LBR hand), hands that
    miss stop, file a request and are played again) walked by the Leduc games: PRL_LBRB_PF_MIN=1 sends their one-card-to-come decisions through it.
    Same per-hand winnings as the reference's worker (tests/golden/lbr_*.npz)."""
    monkeypatch.setenv("PRL_LBRB_PF_MIN", "1")
    check_batched_vs_golden(tag, tmp_path, max_hands=60)


def check_batched_before_the_flop_vs_host(tmp_path, n_hands):
    """lbr_check_to_round = None on DiscretizedNLHoldem (the reference's default, LBRArgs.py:18): LBR decides before the flop too, C(50, 5) run-outs per
    candidate range. BatchedLBR (equity cache + request / replay rounds) against the host LocalLBRWorker (one prl_lbr_checkdown_equity call per such
    decision), the same decks and agent draws, hand for hand"""
    game_cls, agent_bets, lbr_kwargs = DiscretizedNLHoldem, bet_sets.B_2, dict(lbr_bet_set=bet_sets.B_2, lbr_check_to_round=None)
    t_prof = make_t_prof(game_cls, agent_bets, lbr_kwargs, n_hands, tmp_path)
    record = []
    w = LocalLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=7, record=record))
    b = BatchedLBR(t_prof, agent_kind="hash", agent_seed=7)
    lut = game_cls.get_lut_holder()
    for seat in (0, 1):
        np.random.seed(300 + seat)
        n0 = len(record)
        host = w.run(agent_seat_id=seat, n_iterations=n_hands, mode="HASH", stack_size=[game_cls.DEFAULT_STACK_SIZE] * 2)
        decks = decks_from_record(record[n0:], lut, b.n_deal - 2 * b._rules.n_hole_cards)
        got = b.run(agent_seat_id=seat, n_hands=n_hands, decks=decks)
        assert np.array_equal(got, host), "seat %d: %d of %d hands differ (first at %s): %s vs %s" % (seat, int(np.sum(got != host)), n_hands, np.flatnonzero(got != host)[:5], got[:6], host[:6])
        assert b.last_stats["lbr_lookaheads"] > 0


@pytest.mark.gpu
def test_gpu_batched_lbr_before_the_flop_vs_host_worker(tmp_path):
    # (r71: 16 hands per seat took 67 s with the one-lane walk of the deal tree; r73, terms in parallel: 2.5 s)
    check_batched_before_the_flop_vs_host(tmp_path, int(os.environ.get("PRL_LBR_PREFLOP_HANDS", "64")))


def test_batched_lbr_holdem_vs_reference_emu(emu_lib, tmp_path):
    """1326-hand ranges through the emulator: every inter-lane hand-off of the kernel is exercised with the fibers run strictly
    one after another (this is how a missing barrier before lane 0's env step was found)."""
    check_batched_vs_golden("DiscretizedNLHoldem", tmp_path, max_hands=2)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["StandardLeduc", "DiscretizedNLLeduc", "DiscretizedNLHoldem", "DiscretizedNLHoldem_flop"])
def test_gpu_batched_lbr_vs_reference(tag, tmp_path):
    check_batched_vs_golden(tag, tmp_path)


# ---- tabular agents (agent kind "table": a CFR solver's average strategy in HBM) --------------------------------------------------------------------
TABLE_CASES = {
    "StandardLeduc": (StandardLeduc, None, dict(lbr_check_to_round=None), 12),
    # LBR with the agent's own bet set: every hand stays in the agent's tree
    "DiscretizedNLLeduc": (DiscretizedNLLeduc, bet_sets.POT_ONLY, dict(lbr_bet_set=bet_sets.POT_ONLY, lbr_check_to_round=None), 6),
    # LBR with more bet sizes than the agent: raises outside the agent's tree meet the uniform fall-back, on both sides
    "DiscretizedNLLeduc_off_tree": (DiscretizedNLLeduc, bet_sets.POT_ONLY, dict(lbr_bet_set=bet_sets.B_3, lbr_check_to_round=None), 6),
}


class _Chief:
    def create_experiment(self, name):
        return name

    def add_scalar(self, *a):
        pass


def solved_table(game_cls, agent_bets, n_iters):
    """CFR+ on the agent's game, its average strategy as a PolicyTable (and the tree it came from)"""
    from pokerrl_amd.cfr.CFRPlus import CFRPlus
    from pokerrl_amd.rl.tabular_agent import PolicyTable
    cfr = CFRPlus(name="tab", chief_handle=_Chief(), game_cls=game_cls, agent_bet_set=agent_bets, delay=0)
    cfr.reset()
    for _ in range(n_iters):
        cfr.iteration()
    return PolicyTable.from_cfr(cfr), cfr


def check_table_on_device(table):
    """the table where it lives: every node's history key finds its row through the device's open-addressed look-up and the device reads the host's
    float32 probabilities; keys the table does not hold miss"""
    rng = np.random.RandomState(3)
    keys = [hk for hk in table.node_keys.values() if table.row_of(hk) >= 0]
    keys = [keys[i] for i in rng.permutation(len(keys))[:512]]
    a, h = rng.randint(0, table.n_actions, len(keys)), rng.randint(0, table.range_size, len(keys))
    rows, probs = table.probe(keys, a, h)
    want_rows = np.array([table.row_of(k) for k in keys])
    assert np.array_equal(rows, want_rows), (rows[:8], want_rows[:8])
    assert np.array_equal(probs, table.probs[want_rows, a, h])
    strangers = [(k[0] ^ 0x55, k[1]) for k in keys[:64]]
    rows, probs = table.probe(strangers, a[:64], h[:64])
    assert np.all(rows == -1) and np.all(probs == 0)


def check_batched_table_vs_host(tag, tmp_path, n_hands):
    """BatchedLBR against a tabular agent = the host LocalLBRWorker playing the same table as an EvalAgent, hand by hand (float32, bit for bit);
    the table is the average strategy CFR+ left in the agent's public tree"""
    from pokerrl_amd.rl.tabular_agent import make_table_agent_cls
    game_cls, agent_bets, lbr_kwargs, n_iters = TABLE_CASES[tag]
    table, cfr = solved_table(game_cls, agent_bets, n_iters)
    # the table is the tree's average strategy, node by node: filling a tree from the agent gives the columns back
    tree = cfr._trees[0]
    t_prof = make_t_prof(game_cls, agent_bets, lbr_kwargs, n_hands, tmp_path)
    agent_cls = make_table_agent_cls(EvalAgentBase, table, seed=7)
    tree.fill_with_agent_policy(agent_cls(t_prof=t_prof, mode="TABLE"))
    assert np.array_equal(tree.solver.get("strategy").astype(np.float32), cfr.average_strategy().astype(np.float32))
    check_table_on_device(table)
    record = []
    w = LocalLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=make_table_agent_cls(EvalAgentBase, table, seed=7, record=record))
    b = BatchedLBR(t_prof, agent_kind="table", agent_seed=7, table=table)
    lut = game_cls.get_lut_holder()
    n_off = 0
    for seat in (0, 1):
        np.random.seed(4242 + seat)
        n0 = len(record)
        agent = w.agent.cpu_agent
        misses = []
        lookup = agent.TABLE.row_of
        agent.TABLE.row_of = lambda hk, _f=lookup: (misses.append(1) if _f(hk) < 0 else None, _f(hk))[1]
        want = w.run(agent_seat_id=seat, n_iterations=n_hands, mode="TABLE", stack_size=[game_cls.DEFAULT_STACK_SIZE] * 2)
        del agent.TABLE.row_of
        n_off += len(misses)
        decks = decks_from_record(record[n0:], lut, b.n_deal - 2 * b._rules.n_hole_cards)
        got = b.run(agent_seat_id=seat, n_hands=n_hands, decks=decks)
        assert np.array_equal(got, want), "%s seat %d: %d of %d hands differ (first at %s)" % (tag, seat, int(np.sum(got != want)), n_hands, np.flatnonzero(got != want)[:5])
        assert b.last_stats["agent_actions"] > 0
    assert (n_off > 0) == tag.endswith("off_tree"), n_off  # the off-tree case does leave the tree, the others never do
    # a solved agent is not the uniform one: the table is in use (LBR wins less against it)
    if not tag.endswith("off_tree"):
        uni = BatchedLBR(t_prof, agent_kind="uniform").run(agent_seat_id=0, n_hands=n_hands, decks=decks)
        assert not np.array_equal(uni, got)
    table.close()


@pytest.mark.parametrize("tag", ["StandardLeduc", "DiscretizedNLLeduc", "DiscretizedNLLeduc_off_tree"])
def test_batched_lbr_table_agent_vs_host_worker_emu(emu_lib, tag, tmp_path):
    check_batched_table_vs_host(tag, tmp_path, 60)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["StandardLeduc", "DiscretizedNLLeduc", "DiscretizedNLLeduc_off_tree"])
def test_gpu_batched_lbr_table_agent_vs_host_worker(tag, tmp_path):
    check_batched_table_vs_host(tag, tmp_path, 400)


@pytest.mark.gpu
def test_gpu_lbr_against_the_solvers_average_strategy_is_a_lower_bound_of_its_exploitability(tmp_path):
    """the two evaluators on ONE tabular agent: LBR's winnings against the average strategy of CFR+ (policy table in HBM, 2^17 hands per seat) stay
    below the exact best response the solver computes for the same strategy (LBR is a lower bound: Lisy & Bowling 2017), and both fall as CFR+ runs"""
    game_cls, n = StandardLeduc, 1 << 17
    t_prof = make_t_prof(game_cls, None, dict(lbr_check_to_round=None), n, tmp_path)
    seen = []
    for n_iters in (2, 300):
        table, cfr = solved_table(game_cls, None, n_iters)
        expl = cfr._scaled(0, cfr._trees[0].solver.eval_avg())
        b = BatchedLBR(t_prof, agent_kind="table", agent_seed=7, table=table)
        w = np.concatenate([b.run(agent_seat_id=s, n_hands=n, deck_seed=5, first_hand=s * n, episode_base=s * n) for s in (0, 1)]).astype(np.float64)
        half = 1.96 * w.std() / np.sqrt(w.size)
        assert w.mean() - half <= expl, (n_iters, w.mean(), half, expl)
        seen.append((expl, w.mean(), half))
        table.close()
    assert seen[1][0] < 0.2 * seen[0][0] and seen[1][1] + seen[1][2] < seen[0][1] - seen[0][2], seen  # exploitability and LBR winnings both fell


# ---- hold'em-sized tables straight from the fused solver (PolicyTable.from_solver), suit-canonical look-ups -----------------------------------------------
import contextlib


@contextlib.contextmanager
def forced_deck(deck):
    """inside: a 52-card deck that is shuffled comes out with `deck` (1d cards) on top, in that order"""
    deck, shuffle = [int(c) for c in deck], np.random.shuffle
    order = np.asarray(deck + [c for c in range(52) if c not in deck])

    def forced(arr):
        if getattr(arr, "shape", None) == (52, 2):
            arr[:] = np.stack([order // 4, order % 4], axis=1)
        else:
            shuffle(arr)

    np.random.shuffle = forced
    try:
        yield
    finally:
        np.random.shuffle = shuffle


class _DealtLBRWorker(LocalLBRWorker):
    """the host worker on GIVEN decks (seat 0's hole cards, seat 1's, the board in deal order): the env's shuffle of hand k yields deck k"""
    DECKS, _k = None, 0

    def _reset_episode(self):
        with forced_deck(self.DECKS[self._k]):
            ret = self._env.reset()
        self._k += 1
        self.agent.reset(deck_state_dict=self._env.cards_state_dict())
        self.agent_range.reset()
        return ret


def fhp_class_solver(L, n_classes, n_iters, variant="plus"):
    """CFR on Flop5Holdem over a few suit classes (the whole-game solve in small: representatives x orbit sizes, orbit-mean chance values)"""
    import parity_cases as pc
    reps, mult = pc.iso_classes(n_classes)
    t = pc.fhp_tree_of(L, reps)
    s = _native.NativeSolver(t, variant, 0, _lib=L, board_mult=mult, symmetrize="subset")
    s.iterations(n_iters)
    return s, reps, mult


def decks_on_classes(reps, n_hands, seed, stranger_every=0):
    """hands whose boards are random MEMBERS of the given suit classes (any relabelling, any deal order); every `stranger_every`-th board is from no listed class"""
    import parity_cases as pc
    rng = np.random.RandomState(seed)
    orbits = [pc.suit_orbit([int(c) for c in r]) for r in reps]
    listed = {b for o in orbits for b in o}
    decks = []
    for i in range(n_hands):
        if stranger_every and i % stranger_every == stranger_every - 1:
            while True:
                board = tuple(sorted(int(c) for c in rng.choice(52, 5, replace=False)))
                if board not in listed:
                    break
        else:
            o = orbits[rng.randint(len(orbits))]
            board = o[rng.randint(len(o))]
        board = [board[j] for j in rng.permutation(5)]
        rest = [c for c in range(52) if c not in board]
        hole = [rest[j] for j in rng.choice(len(rest), 4, replace=False)]
        decks.append(hole + board)
    return np.asarray(decks, np.int8)


def check_canon_agrees_with_the_library(L):
    """the host twin's canonicalisation (rl/tabular_agent.py: suit_canon, hand_perm) was written on its own: equal to the library's (prl_suit_canon /
    prl_suit_perm_hand through the table look-ups below) on random boards, boards with stabilisers among them; representatives are fixed points"""
    import ctypes
    from pokerrl_amd.game import board_enum
    from pokerrl_amd.rl.tabular_agent import suit_canon
    rng = np.random.RandomState(1)
    boards = np.stack([np.sort(rng.choice(52, 5, replace=False)) for _ in range(3000)] + [[0, 4, 8, 12, 16], [0, 1, 2, 3, 7], [3, 7, 11, 50, 51], [1, 5, 9, 13, 18]]).astype(np.int8)
    out, perm = np.zeros_like(boards), np.zeros(len(boards), np.int32)
    L.prl_suit_canon_boards.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.prl_suit_canon_boards.restype = ctypes.c_int32
    _native.check(L.prl_suit_canon_boards(boards.ctypes.data_as(ctypes.c_void_p), len(boards), 5, 4, out.ctypes.data_as(ctypes.c_void_p), perm.ctypes.data_as(ctypes.c_void_p)), L)
    for b, cb, k in zip(boards, out, perm):
        want, wk = suit_canon(b)
        assert np.array_equal(cb, want) and k == wk, (b, cb, k, want, wk)
    import parity_cases as pc
    reps, _ = pc.iso_classes(40)
    r2, k2 = np.zeros_like(reps), np.zeros(len(reps), np.int32)
    _native.check(L.prl_suit_canon_boards(np.ascontiguousarray(reps).ctypes.data_as(ctypes.c_void_p), len(reps), 5, 4, r2.ctypes.data_as(ctypes.c_void_p), k2.ctypes.data_as(ctypes.c_void_p)), L)
    assert np.array_equal(r2, reps) and np.all(k2 == 0)  # a representative is its own canonical form under the identity
    assert board_enum is not None


def check_solver_table_vs_host(L, tmp_path, n_classes, n_iters, n_hands, variant="plus"):
    """PolicyTable.from_solver on a suit-class solve of Flop5Holdem = the solver's average strategy (row for row against prl_solver_get), and BatchedLBR
    against it = the host LocalLBRWorker playing the same table, hand for hand, on boards that are arbitrary members of the classes (relabelled, shuffled
    deal order: the device and the host twin each canonicalise on their own) and on boards of no listed class (uniform play on both sides)"""
    from pokerrl_amd.game.games import Flop5Holdem
    from pokerrl_amd.rl.tabular_agent import PolicyTable, make_table_agent_cls
    s, reps, mult = fhp_class_solver(L, n_classes, n_iters, variant)
    table = PolicyTable.from_solver(s)
    assert table.suit_canon and table.n_actions == 3 and table.range_size == 1326
    t = s.tree
    kind, first_col, nch, col_action = t.field("kind"), t.field("first_col"), t.field("n_children"), t.field("col_action")
    dec = [i for i in range(t.n_nodes) if kind[i] == 0]
    assert table.n_rows == len(dec) == 2 + 6 * len(reps)
    avg = s.get("avg")
    for r in (0, 1, 2, 5, 8, table.n_rows - 1):  # rows are the decision nodes in node order: the table holds float32(average), 0 for actions a node lacks
        want = np.zeros((3, 1326), np.float32)
        for j in range(nch[dec[r]]):
            want[col_action[first_col[dec[r]] + j]] = avg[first_col[dec[r]] + j].astype(np.float32)
        assert np.array_equal(table.row_probs(r), want), r
    t_prof = make_t_prof(Flop5Holdem, None, dict(lbr_check_to_round=Poker.FLOP), n_hands, tmp_path)
    b = BatchedLBR(t_prof, agent_kind="table", agent_seed=7, table=table)
    total_misses = 0
    for seat in (0, 1):
        decks = decks_on_classes(reps, n_hands, seed=11 + seat, stranger_every=5)
        w = _DealtLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=make_table_agent_cls(EvalAgentBase, table, seed=7))
        w.DECKS = decks
        misses, lookup = [], table.row_of
        table.row_of = lambda hk, _f=lookup: (misses.append(1) if _f(hk) < 0 else None, _f(hk))[1]
        try:
            want = w.run(agent_seat_id=seat, n_iterations=n_hands, mode="TABLE", stack_size=[Flop5Holdem.DEFAULT_STACK_SIZE] * 2)
        finally:
            del table.row_of
        got = b.run(agent_seat_id=seat, n_hands=n_hands, decks=decks)
        assert np.array_equal(got, want), "seat %d: %d of %d hands differ (first at %s): %s vs %s" % (seat, int(np.sum(got != want)), n_hands, np.flatnonzero(got != want)[:5], got[:8], want[:8])
        total_misses += len(misses)
        if seat == 0:  # the agent (seat 0) raises or folds before the flop; LBR (the big blind) calls and plays the flop: the table is in use there
            assert len(np.unique(got)) > 1, got  # (with the agent in seat 1, LBR's forced pre-flop "call" is a fold of the small blind: -500 every hand)
            uni = BatchedLBR(t_prof, agent_kind="uniform").run(agent_seat_id=seat, n_hands=n_hands, decks=decks)
            assert not np.array_equal(uni, got)
            assert b.last_stats["lbr_lookaheads"] > 0
    assert total_misses > 0  # the strangers' boards were met (and played uniformly on both sides)
    table.close()
    return s


def check_solver_table_h2h_vs_host(L, tmp_path, n_classes, n_iters, n_hands):
    """head-to-head on Flop5Holdem: the class solve's average strategy (mode "TABLE", a suit-canonical table built on the device) against the hash agent
    -- BatchedHead2Head = the host LocalHead2HeadMaster on the same decks, hand for hand"""
    from pokerrl_amd.eval.head_to_head import BatchedHead2Head, H2HArgs, LocalHead2HeadMaster
    from pokerrl_amd.game.games import Flop5Holdem
    from pokerrl_amd.rl.tabular_agent import PolicyTable, make_table_agent_cls
    s, reps, _mult = fhp_class_solver(L, n_classes, n_iters)
    table = PolicyTable.from_solver(s)
    t_prof = TrainingProfileBase(
        name="h2h_fhp", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=Flop5Holdem,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("TABLE", "HASH2"), eval_stack_sizes=None,
        module_args={"env": Flop5Holdem.ARGS_CLS(n_seats=2), "h2h": H2HArgs(n_hands=n_hands)}, path_data=str(tmp_path))
    decks = decks_on_classes(reps, 2 * n_hands, seed=23, stranger_every=7)
    m = LocalHead2HeadMaster(t_prof=t_prof, chief_handle=_Chief(), eval_agent_cls=make_table_agent_cls(EvalAgentBase, table, seed=11))
    m.set_modes(["TABLE", "HASH2"])
    get_env, dealt = m._eval_env_bldr.get_new_env, [0]

    def get_dealt_env(**kw):  # the master's own env deals deck k for hand k (the agents' envs take their cards from it)
        env = get_env(**kw)
        reset = env.reset

        def dealt_reset(deck_state_dict=None):
            with forced_deck(decks[dealt[0]]):
                out = reset(deck_state_dict=deck_state_dict)
            dealt[0] += 1
            return out

        env.reset = dealt_reset
        return env

    m._eval_env_bldr.get_new_env = get_dealt_env
    want = m.play(stack_size=t_prof.eval_stack_sizes[0])
    assert dealt[0] == 2 * n_hands
    b = BatchedHead2Head(t_prof, kinds=("table", "hash"), seeds=(11, 12), tables=(table, None))
    got = b.play(n_hands=n_hands, decks=decks)
    assert np.array_equal(got, want), "%d of %d hands differ (first at %s)" % (int(np.sum(got != want)), 2 * n_hands, np.flatnonzero(got != want)[:5])
    assert len(np.unique(got)) > 4
    table.close()


def check_solver_table_equals_tree_table(game_cls, agent_bets, n_iters, variant="plus", expect_twins=False, expect_engine=None, **cfr_kw):
    """prl_policy_table_from_solver on the LEVELS engine (columns already in hand order) = the host path PolicyTable.from_cfr: the same rows in the same
    order under the same keys, the same float32 probabilities; node keys by path = node keys by index; and on the device the look-ups hit.
    expect_twins: a bet set whose sizes the env turns into ONE amount (a size below the minimum raise is raised to it, the next size IS the minimum raise):
    such siblings repeat each other node for node under one history key -- both builders keep the first and drop the rest (round 5's advisor finding)"""
    from pokerrl_amd.cfr.CFRPlus import CFRPlus
    from pokerrl_amd.cfr.LinearCFR import LinearCFR
    from pokerrl_amd.cfr.VanillaCFR import VanillaCFR
    from pokerrl_amd.rl.tabular_agent import PolicyTable
    cls = {"plus": CFRPlus, "linear": LinearCFR, "vanilla": VanillaCFR}[variant]
    kw = dict(delay=0) if variant == "plus" else {}
    cfr = cls(name="tab", chief_handle=_Chief(), game_cls=game_cls, agent_bet_set=agent_bets, **kw, **cfr_kw)
    assert expect_engine is None or cfr._trees[0].solver.engine == expect_engine
    cfr.reset()
    for _ in range(n_iters):
        cfr.iteration()
    host, dev = PolicyTable.from_cfr(cfr), PolicyTable.from_solver(cfr)
    n_dec = int(np.sum(cfr._trees[0]._kind == 0))
    assert (host.n_rows < n_dec) == expect_twins, (host.n_rows, n_dec)
    assert (dev.n_rows, dev.n_actions, dev.range_size, dev.suit_canon) == (host.n_rows, host.n_actions, host.range_size, False)
    assert np.array_equal(dev.keys, host.keys) and np.array_equal(dev.rows, host.rows)  # the same open-addressed table, slot for slot
    for r in range(host.n_rows):
        assert np.array_equal(dev.row_probs(r), host.probs[r]), r
    tree = cfr._trees[0]
    for i in np.flatnonzero(tree._kind == 0)[::7]:
        assert dev.key_of_node(tree.node(int(i))) == host.node_keys[int(i)]
    keys = [hk for hk in host.node_keys.values() if host.row_of(hk) >= 0][:64]
    rng = np.random.RandomState(0)
    a, h = rng.randint(0, host.n_actions, len(keys)), rng.randint(0, host.range_size, len(keys))
    rows, probs = dev.probe(keys, a, h)
    want = np.array([host.row_of(k) for k in keys])
    assert np.array_equal(rows, want) and np.array_equal(probs, host.probs[want, a, h])
    host.close(), dev.close()


@pytest.mark.parametrize("game,bets,variant", [("StandardLeduc", None, "plus"), ("StandardLeduc", None, "linear"), ("DiscretizedNLLeduc", "B_3", "vanilla")])
def test_solver_table_on_the_levels_engine_equals_the_tree_table_emu(emu_lib, game, bets, variant):
    check_solver_table_equals_tree_table({"StandardLeduc": StandardLeduc, "DiscretizedNLLeduc": DiscretizedNLLeduc}[game], getattr(bet_sets, bets) if bets else None, 4, variant)


@pytest.mark.parametrize("game,bets,stack,outcomes,variant", [("DiscretizedNLHoldem", "POT_ONLY", 600, (1, 1, 1), "plus"), ("LimitHoldem", None, 6, (1, 2, 1), "linear")])
def test_solver_table_on_the_per_street_engine_equals_the_tree_table_emu(emu_lib, game, bets, stack, outcomes, variant):
    """prl_policy_table_from_solver on the per-street fused engine (round 6): columns gathered from the engine's internal order -- multi-street trees with mixed
    street shapes and all-in run-out chains (no decision below an all-in call: the chains have no rows) = the host path, slot for slot, float for float"""
    from pokerrl_amd.game.games import DiscretizedNLHoldem, LimitHoldem
    check_solver_table_equals_tree_table({"DiscretizedNLHoldem": DiscretizedNLHoldem, "LimitHoldem": LimitHoldem}[game], getattr(bet_sets, bets) if bets else None, 2, variant,
                                         starting_stack_sizes=[stack], max_outcomes=outcomes, expect_engine="fused")


def check_street_table_float32_average(L, to_rows=None):
    """the table of a per-street solver that keeps its street columns' average in float32 (PRL_SOLVER_AVG_F32): the trunk's rows from the float64 array, the
    streets' rows from the float32 one -- every row = float32(prl_solver_get(AVG)) of its node's columns"""
    import parity_cases as pc
    from pokerrl_amd import _native
    from pokerrl_amd.game.games import DiscretizedNLHoldem
    from pokerrl_amd.rl.tabular_agent import PolicyTable
    t, s, _o = pc.make_streets_pair(L, DiscretizedNLHoldem, 600, pc.multistreet_runouts(1, 2, 1), "plus", 0, bets=bet_sets.POT_ONLY, tape=_NoOracle(), avg_dtype="f32")
    s.iterations(3)
    tab = PolicyTable.from_solver(s)
    avg, kind, fc, nch, ca = s.get("avg"), t.field("kind"), t.field("first_col"), t.field("n_children"), t.field("col_action")
    dec = np.flatnonzero(kind == 0)
    assert tab.n_rows == len(dec)
    for r, n in enumerate(dec):
        got = tab.row_probs(r)
        want = np.zeros_like(got)
        for j in range(nch[n]):
            want[ca[fc[n] + j]] = avg[fc[n] + j].astype(np.float32)
        assert np.array_equal(got, want), (r, n)
    tab.close()


class _NoOracle:
    """make_streets_pair without the oracle (a 'replayed tape' that is never asked)"""
    recording, live = False, False


def test_solver_table_of_a_float32_average_on_the_per_street_engine_emu(emu_lib):
    check_street_table_float32_average(emu_lib)


@pytest.mark.gpu
def test_gpu_solver_table_on_the_per_street_engine():
    from pokerrl_amd.game.games import DiscretizedNLHoldem, LimitHoldem
    L = _native.lib()
    check_solver_table_equals_tree_table(DiscretizedNLHoldem, bet_sets.POT_ONLY, 4, "plus", starting_stack_sizes=[2500], max_outcomes=(2, 2, 2), expect_engine="fused")
    check_solver_table_equals_tree_table(LimitHoldem, None, 3, "vanilla", max_outcomes=(2, 1, 2), expect_engine="fused")
    check_street_table_float32_average(L)


def check_street_solver_table_lbr_vs_host(L, tmp_path, stack, runouts, n_iters, n_hands, stranger_every=4):
    """the evaluators pointed at a MULTI-STREET solution (round 6): CFR+ on DiscretizedNLHoldem with pot-sized raises over a few run-outs on the per-street
    engine (mixed street shapes, run-out chains), its average strategy as a policy table built on the device, and BatchedLBR (LBR raises 0.5 / 1 / 2 pots:
    sizes the agent's tree has and sizes it does not have) against it = the host LocalLBRWorker playing the same table, hand for hand -- on decks whose boards
    are the tree's run-outs (flop in the tree's order: the history key hashes the cards as dealt) and on boards the tree never dealt (uniform play on both sides)"""
    import parity_cases as pc
    from pokerrl_amd.game.games import DiscretizedNLHoldem
    from pokerrl_amd.rl.tabular_agent import PolicyTable, make_table_agent_cls
    ro = pc.multistreet_runouts(*runouts)
    t, s, _o = pc.make_streets_pair(L, DiscretizedNLHoldem, stack, ro, "plus", 0, bets=bet_sets.POT_ONLY, tape=_NoOracle())
    s.iterations(n_iters)
    table = PolicyTable.from_solver(s)
    assert not table.suit_canon and table.n_actions == 3 and table.n_rows == int(np.sum(t.field("kind") == 0))
    t_prof = make_t_prof(DiscretizedNLHoldem, bet_sets.POT_ONLY, dict(lbr_bet_set=[0.5, 1.0, 2.0], lbr_check_to_round=Poker.FLOP), n_hands, tmp_path)
    b = BatchedLBR(t_prof, agent_kind="table", agent_seed=5, table=table)
    b.set_stack_size([stack, stack])
    rng = np.random.RandomState(3)
    total_misses = 0
    for seat in (0, 1):
        decks = []
        for i in range(n_hands):
            if stranger_every and i % stranger_every == stranger_every - 1:
                board = [int(c) for c in rng.choice(52, 5, replace=False)]
            else:
                board = [int(c) for c in ro[rng.randint(len(ro))]]
            rest = [c for c in range(52) if c not in board]
            decks.append([rest[j] for j in rng.choice(len(rest), 4, replace=False)] + board)
        decks = np.asarray(decks, np.int8)
        w = _DealtLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=make_table_agent_cls(EvalAgentBase, table, seed=5))
        w.DECKS = decks
        misses, lookup = [], table.row_of
        table.row_of = lambda hk, _f=lookup: (misses.append(1) if _f(hk) < 0 else None, _f(hk))[1]
        try:
            want = w.run(agent_seat_id=seat, n_iterations=n_hands, mode="TABLE", stack_size=[stack, stack])
        finally:
            del table.row_of
        got = b.run(agent_seat_id=seat, n_hands=n_hands, decks=decks)
        assert np.array_equal(got, want), "seat %d: %d of %d hands differ (first at %s): %s vs %s" % (seat, int(np.sum(got != want)), n_hands, np.flatnonzero(got != want)[:5], got[:8], want[:8])
        total_misses += len(misses)
        if seat == 0:
            uni = BatchedLBR(t_prof, agent_kind="uniform")
            uni.set_stack_size([stack, stack])
            assert not np.array_equal(uni.run(agent_seat_id=seat, n_hands=n_hands, decks=decks), got)
    assert total_misses > 0 and b.last_stats["lbr_lookaheads"] > 0
    table.close()


def test_batched_lbr_against_a_multi_street_solution_vs_host_worker_emu(emu_lib, tmp_path):
    check_street_solver_table_lbr_vs_host(emu_lib, tmp_path, 600, (1, 2, 1), 2, 8)


@pytest.mark.gpu
def test_gpu_batched_lbr_against_a_multi_street_solution_vs_host_worker(tmp_path):
    check_street_solver_table_lbr_vs_host(_native.lib(), tmp_path, 2500, (3, 3, 3), 20, 96)


def test_solver_table_merges_children_the_env_turns_into_one_state_emu(emu_lib):
    check_solver_table_equals_tree_table(DiscretizedNLLeduc, [0.3, 0.34, 1.0], 2, "plus", expect_twins=True, starting_stack_sizes=[1000])


@pytest.mark.gpu
def test_gpu_solver_table_on_the_levels_engine_equals_the_tree_table():
    check_solver_table_equals_tree_table(DiscretizedNLLeduc, bet_sets.B_3, 6, "plus")


def test_host_canonicalisation_equals_the_librarys(emu_lib):
    check_canon_agrees_with_the_library(emu_lib)


def test_batched_lbr_solver_table_flop5holdem_vs_host_worker_emu(emu_lib, tmp_path):
    check_solver_table_vs_host(emu_lib, tmp_path, 3, 2, 10)


def test_batched_h2h_solver_table_flop5holdem_vs_host_master_emu(emu_lib, tmp_path):
    check_solver_table_h2h_vs_host(emu_lib, tmp_path, 3, 2, 14)


@pytest.mark.gpu
def test_gpu_batched_h2h_solver_table_flop5holdem_vs_host_master(tmp_path):
    check_solver_table_h2h_vs_host(_native.lib(), tmp_path, 12, 4, 300)


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["plus", "linear"])
def test_gpu_batched_lbr_solver_table_flop5holdem_vs_host_worker(variant, tmp_path):
    check_canon_agrees_with_the_library(_native.lib())
    check_solver_table_vs_host(_native.lib(), tmp_path, 12, 4, 160, variant)


@pytest.mark.gpu
def test_gpu_lbr_and_self_play_on_the_whole_game_flop5holdem_solution(tmp_path):
    """The evaluators pointed at the headline solver's own output, the WHOLE Flop5Holdem game (2 598 960 boards as 134 459 suit classes, 30 GB), with no
    oracle in the loop: after 10 and after 100 CFR+ iterations the average strategy goes into a suit-canonical policy table on the device (806 758 rows,
    12.8 GB); batched LBR plays 2^20 hands per seat against it (LBR acts from the flop on: as the big blind it calls the raise and plays the flop; as the
    small blind its forced pre-flop call IS a fold in this game -- FIRST_ACTION_NO_CALL -- and loses the 50-chip blind every hand, on the host worker
    alike); batched self-play of the table against itself gives the game value of the big blind. Three engines have to agree:
      * LBR is a strategy, so the big blind's LBR winnings stay below its exact best-response value = the solver's own exploitability[seat 1] + the
        game value of seat 1 (self-play), and LBR's mean over both seats below the mean exploitability (Lisy & Bowling 2017);
      * exploitability and LBR's edge over the game value both fall from 10 to 100 iterations."""
    import parity_cases as pc
    from pokerrl_amd.eval.head_to_head import BatchedHead2Head, H2HArgs
    from pokerrl_amd.game import bet_sets, board_enum
    from pokerrl_amd.game.games import Flop5Holdem
    from pokerrl_amd.rl.tabular_agent import PolicyTable
    n = 1 << 20
    reps, mult = board_enum.single_deal_board_classes(Flop5Holdem)
    t = _native.NativeTree.for_game(Flop5Holdem, 20000, bet_sets.POT_ONLY, reps)
    s = _native.NativeSolver(t, "plus", 0, engine="fused", board_mult=mult, symmetrize=True)
    t_prof = TrainingProfileBase(
        name="whole", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=Flop5Holdem,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("TABLE",), eval_stack_sizes=None,
        module_args={"env": Flop5Holdem.ARGS_CLS(n_seats=2), "lbr": LBRArgs(n_lbr_hands_per_seat=n, lbr_check_to_round=Poker.FLOP), "h2h": H2HArgs(n_hands=n)},
        path_data=str(tmp_path))
    seen = []
    for n_iters in (10, 100):
        s.iterations(n_iters - s.iter)
        expl = s.eval_avg().astype(np.float64) * float(Flop5Holdem.EV_NORMALIZER)  # per seat, mbb per game
        table = PolicyTable.from_solver(s)
        assert table.suit_canon and table.n_rows == 2 + 6 * len(reps)
        b = BatchedLBR(t_prof, agent_kind="table", agent_seed=7, table=table)
        w_bb = b.run(agent_seat_id=0, n_hands=n, deck_seed=5).astype(np.float64)                        # LBR in the big blind
        w_sb = b.run(agent_seat_id=1, n_hands=n, deck_seed=5, first_hand=n, episode_base=n).astype(np.float64)
        assert np.all(w_sb == -500.0)  # 50 chips = half a big blind, every hand (see above)
        h = BatchedHead2Head(t_prof, kinds=("table", "table"), seeds=(11, 12), tables=(table, table))
        v = h.play(n_hands=n, deck_seed=9).astype(np.float64)  # [2n]: the reference copy's winnings in seat 0, then in seat 1
        v1 = 0.5 * (v[n:].mean() - v[:n].mean())               # game value of seat 1 under the average strategy (zero-sum: V1 = -V0)
        ci = lambda x: 1.96 * x.std() / np.sqrt(x.size)        # noqa: E731
        ci_v = 0.5 * np.hypot(ci(v[:n]), ci(v[n:]))
        edge, ci_e = w_bb.mean() - v1, np.hypot(ci(w_bb), ci_v)
        # LBR's winnings in the big blind <= the exact best-response value of seat 1 = exploitability[1] + V1
        assert edge - ci_e <= expl[1], (n_iters, w_bb.mean(), v1, expl, ci_e)
        assert 0.5 * (w_bb.mean() + w_sb.mean()) - 0.5 * ci(w_bb) <= expl.mean(), (n_iters, w_bb.mean(), expl)
        assert abs(v[:n].mean() + v[n:].mean()) <= 2.0 * np.hypot(ci(v[:n]), ci(v[n:])) + 1e-9  # self-play is zero-sum up to the two halves' noise
        seen.append((expl.mean(), edge, ci_e, expl[1]))
        print("whole game, %d CFR+ iterations: exploitability %s mbb/g, LBR (big blind) %.2f, game value of the big blind %.2f +- %.2f, LBR's edge %.2f +- %.2f of "
              "the exact %.2f" % (n_iters, expl, w_bb.mean(), v1, ci_v, edge, ci_e, expl[1]))
        table.close()
    assert seen[1][0] < 0.5 * seen[0][0], seen                      # the exact exploitability fell
    assert seen[1][1] + seen[1][2] < seen[0][1] - seen[0][2], seen  # and so did what LBR finds
    assert pc is not None


def check_master_drives_batched_worker(tmp_path, n_hands):
    """LocalLBRMaster with a BatchedLBRWorker: the chief hands the solver's table over through update_weights, the master logs mean and confidence of
    exactly the hands BatchedLBR plays for those deck / episode numbers"""
    from pokerrl_amd.eval.lbr import BatchedLBRWorker
    game_cls = StandardLeduc
    table, _cfr = solved_table(game_cls, None, 8)
    t_prof = make_t_prof(game_cls, None, dict(lbr_check_to_round=None), n_hands, tmp_path)

    class Chief(ChiefBase):
        def pull_current_eval_strategy(self, last):
            return table, last

    chief = Chief(t_prof)
    m = LocalLBRMaster(t_prof=t_prof, chief_handle=chief)
    worker = BatchedLBRWorker(t_prof, chief_handle=chief, deck_seed=3)
    assert worker.run(0, 4, "HASH", [game_cls.DEFAULT_STACK_SIZE] * 2) is None  # no table yet
    m.set_worker_handles(worker)
    m.update_weights()
    m.evaluate(iter_nr=0)
    vals, _ = chief.get_new_values()
    total = [g for n, g in vals.items() if n.endswith("LBR Total")]
    assert len(total) == 1
    logged = list(total[0].values())[0][-1][1]
    b = BatchedLBR(t_prof, agent_kind="table", agent_seed=7, table=table)
    n = int(t_prof.module_args["lbr"].n_lbr_hands / t_prof.module_args["lbr"].n_workers)
    want = np.concatenate([b.run(s, n, deck_seed=3, first_hand=s * n, episode_base=s * n) for s in (0, 1)])
    assert logged == float(np.mean(want)), (logged, float(np.mean(want)))
    m.evaluate(iter_nr=1)  # the next evaluation plays the next hands
    vals, _ = chief.get_new_values()
    logged2 = [list(g.values())[0][-1][1] for nme, g in vals.items() if nme.endswith("LBR Total")][0]
    want2 = np.concatenate([b.run(s, n, deck_seed=3, first_hand=(2 + s) * n, episode_base=(2 + s) * n) for s in (0, 1)])
    assert logged2 == float(np.mean(want2)) and logged2 != logged
    table.close()


def test_lbr_master_drives_the_batched_worker_emu(emu_lib, tmp_path):
    check_master_drives_batched_worker(tmp_path, 40)


@pytest.mark.gpu
def test_gpu_lbr_master_drives_the_batched_worker(tmp_path):
    check_master_drives_batched_worker(tmp_path, 4096)


def check_batched_h2h_table_vs_host(tmp_path, n_hands):
    """head-to-head: the solver's average strategy (mode "TABLE") against the hash agent (mode "HASH2") -- BatchedHead2Head with kinds ("table", "hash")
    = the host LocalHead2HeadMaster with the two modes of the table agent class, hand by hand"""
    from pokerrl_amd.eval.head_to_head import BatchedHead2Head, H2HArgs, LocalHead2HeadMaster
    from pokerrl_amd.rl.tabular_agent import make_table_agent_cls
    table, _cfr = solved_table(StandardLeduc, None, 10)
    t_prof = TrainingProfileBase(
        name="h2h_tab", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=StandardLeduc,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("TABLE", "HASH2"), eval_stack_sizes=None,
        module_args={"env": StandardLeduc.ARGS_CLS(n_seats=2), "h2h": H2HArgs(n_hands=n_hands)}, path_data=str(tmp_path))
    record = []
    m = LocalHead2HeadMaster(t_prof=t_prof, chief_handle=_Chief(), eval_agent_cls=make_table_agent_cls(EvalAgentBase, table, seed=11, record=record))
    m.set_modes(["TABLE", "HASH2"])
    np.random.seed(99)
    want = m.play(stack_size=t_prof.eval_stack_sizes[0])
    b = BatchedHead2Head(t_prof, kinds=("table", "hash"), seeds=(11, 12), tables=(table, None))
    decks = decks_from_record(record[::2], StandardLeduc.get_lut_holder(), b.n_deal - 2 * b._rules.n_hole_cards)
    got = b.play(n_hands=n_hands, decks=decks)
    assert np.array_equal(got, want), "%d of %d hands differ (first at %s)" % (int(np.sum(got != want)), 2 * n_hands, np.flatnonzero(got != want)[:5])
    table.close()


def check_h2h_master_on_the_batched_engine(tmp_path, n_hands):
    """BatchedHead2HeadMaster: the reference's master protocol (set_modes / update_weights / evaluate) with the hands on the GPU; the chief hands a table
    per mode over; the logged mean is the mean of the hands BatchedHead2Head plays for those deck numbers, the next evaluation plays the next hands"""
    from pokerrl_amd.eval.head_to_head import BatchedHead2Head, BatchedHead2HeadMaster, H2HArgs
    table_a, _ = solved_table(StandardLeduc, None, 10)
    table_b, _ = solved_table(StandardLeduc, None, 2)
    t_prof = TrainingProfileBase(
        name="h2h_m", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=StandardLeduc,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("A", "B"), eval_stack_sizes=None,
        module_args={"env": StandardLeduc.ARGS_CLS(n_seats=2), "h2h": H2HArgs(n_hands=n_hands)}, path_data=str(tmp_path))

    class Chief(ChiefBase):
        def pull_current_eval_strategy(self, last):
            return {"A": table_a, "B": table_b}, last

    chief = Chief(t_prof)
    m = BatchedHead2HeadMaster(t_prof=t_prof, chief_handle=chief, deck_seed=21)
    m.set_modes(["A", "B"])
    m.evaluate(iter_nr=0)  # no tables yet: nothing can be played, nothing is logged
    assert not any(n.endswith("Head2Head_Winnings Total") and list(g.values())[0] for n, g in chief.get_new_values()[0].items())
    m.update_weights()
    b = BatchedHead2Head(t_prof, kinds=("table", "table"), seeds=(11, 12), tables=(table_a, table_b))
    for it in range(2):
        m.evaluate(iter_nr=it + 1)
        vals, _ = chief.get_new_values()
        logged = [list(g.values())[0][-1][1] for n, g in vals.items() if n.endswith("Head2Head_Winnings Total") and n.startswith("h2h_m A")]
        want = b.play(n_hands, deck_seed=21, first_hand=it * 2 * n_hands)
        assert logged == [float(np.mean(want))], (logged, float(np.mean(want)))
    # ten CFR+ iterations beat two (mean over many hands; the emulator run is too short to assert the sign)
    if n_hands >= 1 << 15:
        assert float(np.mean(want)) > 0
    table_a.close(), table_b.close()


def test_h2h_master_on_the_batched_engine_emu(emu_lib, tmp_path):
    check_h2h_master_on_the_batched_engine(tmp_path, 100)


@pytest.mark.gpu
def test_gpu_h2h_master_on_the_batched_engine(tmp_path):
    check_h2h_master_on_the_batched_engine(tmp_path, 1 << 16)


def test_batched_h2h_table_agent_vs_host_master_emu(emu_lib, tmp_path):
    check_batched_h2h_table_vs_host(tmp_path, 150)


@pytest.mark.gpu
def test_gpu_batched_h2h_table_agent_vs_host_master(tmp_path):
    check_batched_h2h_table_vs_host(tmp_path, 2000)


H2H_CASES = {"StandardLeduc": (StandardLeduc, None), "DiscretizedNLLeduc": (DiscretizedNLLeduc, bet_sets.B_3),
             "DiscretizedNLHoldem": (DiscretizedNLHoldem, bet_sets.B_5)}


def check_batched_h2h_vs_golden(tag, tmp_path):
    """SURVEY 8f-3 on the batched env: the one-lane-per-hand engine plays the golden head-to-head hands (same decks, same agent
    draws) -- per-hand winnings bit-identical to the reference's LocalHead2HeadMaster (tests/golden/h2h_*.npz)"""
    from pokerrl_amd.eval.head_to_head import BatchedHead2Head, H2HArgs, LocalHead2HeadMaster
    game_cls, bets = H2H_CASES[tag]
    g = np.load(os.path.join(HERE, "golden", "h2h_%s.npz" % tag))
    n = int(g["n_hands"])
    env_args = game_cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bets) if bets is not None else game_cls.ARGS_CLS(n_seats=2)
    t_prof = TrainingProfileBase(
        name="h2h", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=game_cls,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH", "HASH2"), eval_stack_sizes=None,
        module_args={"env": env_args, "h2h": H2HArgs(n_hands=n)}, path_data=str(tmp_path))

    class Chief:
        def create_experiment(self, name):
            return name

        def add_scalar(self, *a):
            pass

    # the decks of the golden run: replay the reference's shuffles with the host evaluator (same np.random consumption)
    record = []
    m = LocalHead2HeadMaster(t_prof=t_prof, chief_handle=Chief(), eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=11, record=record))
    m.set_modes(["HASH", "HASH2"])
    np.random.seed(int(g["np_seed"]))
    host = m.play(stack_size=t_prof.eval_stack_sizes[0])
    assert np.array_equal(host, g["winnings"])
    lut = game_cls.get_lut_holder()
    b = BatchedHead2Head(t_prof, kinds=("hash", "hash"), seeds=(11, 12))
    decks = decks_from_record(record[::2], lut, b.n_deal - 2 * b._rules.n_hole_cards)  # both agents record every episode
    assert decks.shape[0] == 2 * n
    got = b.play(n_hands=n, decks=decks)
    want = g["winnings"]
    assert np.array_equal(got, want), "%s: %d of %d hands differ (first at %s)" % (tag, int(np.sum(got != want)), 2 * n, np.flatnonzero(got != want)[:5])
    assert b.last_stats["env_steps"] >= n


@pytest.mark.parametrize("tag", ["StandardLeduc", "DiscretizedNLLeduc", "DiscretizedNLHoldem"])
def test_batched_h2h_vs_reference_emu(emu_lib, tag, tmp_path):
    check_batched_h2h_vs_golden(tag, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["StandardLeduc", "DiscretizedNLLeduc", "DiscretizedNLHoldem"])
def test_gpu_batched_h2h_vs_reference(tag, tmp_path):
    check_batched_h2h_vs_golden(tag, tmp_path)


@pytest.mark.parametrize("agent", ["hash", "table"])
def test_lbr_hand_split_two_ranks_equals_one_rank_gloo_emu(tmp_path, agent):
    """SURVEY 8e, LBR row (LocalLBRMaster.py:53-69: hands split evenly over the workers): BatchedLBR.run_sharded with world_size 2
    over gloo -- every rank plays its half of the same counter-based deck / agent-draw streams -- must give the per-hand winnings
    of the one-rank run, concatenated in rank order, and the same all-reduced (mean, confidence, n). "table": against a tabular agent (every rank
    holds its own copy of the solver's average strategy in its device memory)."""
    import subprocess
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    from test_sharded import _free_port
    lib = build_emu.build()
    n_total, seed = 96, 5

    def run(world, d):
        port = _free_port()
        procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "lbr_split_worker.py"), lib, str(d), str(n_total), str(seed), agent],
                                  env=dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
                 for r in range(world)]
        for p in procs:
            assert p.wait(timeout=900) == 0
        return [dict(np.load(os.path.join(str(d), "rank%d.npz" % r))) for r in range(world)]

    d1, d2 = tmp_path / "w1", tmp_path / "w2"
    d1.mkdir(); d2.mkdir()
    one, two = run(1, d1)[0], run(2, d2)
    for seat in (0, 1):
        k = "x%d" % seat
        assert np.array_equal(np.concatenate([two[0][k], two[1][k]]), one[k]), seat
        for r in (0, 1):
            assert int(two[r]["n%d" % seat]) == n_total
            assert float(two[r]["mean%d" % seat]) == pytest.approx(float(one["mean%d" % seat]), rel=1e-12)  # float64 sums, two addends swapped
            assert float(two[r]["conf%d" % seat]) == pytest.approx(float(one["conf%d" % seat]), rel=1e-9)
    assert len(set(one["x0"].tolist())) > 5  # not a degenerate run


@pytest.mark.gpu
def test_gpu_batched_lbr_equals_host_worker_at_scale(tmp_path):
    """5 000 hold'em hands (2 500 per agent seat; PRL_LBR_SCALE_HANDS=5000: round 5's 10 000 -- the host worker is the slow side and the GPU suite has a
    time limit), LBR acting from the flop on, 990-board look-aheads included: the device-resident
    engine against this package's host LocalLBRWorker -- the drop-in that is itself pinned to the reference's per-hand winnings
    (tests/golden/lbr_*.npz) -- on the SAME decks and agent draws. Every one of the winnings must be bit-identical."""
    game_cls, agent_bets, lbr_kwargs = CASES["DiscretizedNLHoldem_flop"]
    n = int(os.environ.get("PRL_LBR_SCALE_HANDS", "2500"))
    t_prof = make_t_prof(game_cls, agent_bets, lbr_kwargs, n, tmp_path)
    record = []
    w = LocalLBRWorker(t_prof=t_prof, chief_handle=None, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=7, record=record))
    b = BatchedLBR(t_prof, agent_kind="hash", agent_seed=7)
    lut = game_cls.get_lut_holder()
    n_flop_decisions = 0
    for seat in (0, 1):
        np.random.seed(100 + seat)
        n0 = len(record)
        host = w.run(agent_seat_id=seat, n_iterations=n, mode="HASH", stack_size=[game_cls.DEFAULT_STACK_SIZE] * 2)
        decks = decks_from_record(record[n0:], lut, b.n_deal - 2 * b._rules.n_hole_cards)
        got = b.run(agent_seat_id=seat, n_hands=n, decks=decks)
        assert np.array_equal(got, host), "seat %d: %d of %d hands differ (first at %s)" % (seat, int(np.sum(got != host)), n, np.flatnonzero(got != host)[:5])
        n_flop_decisions += b.last_stats["lbr_lookaheads"]
    assert n_flop_decisions > n  # LBR really looked ahead


def check_deal_decks():
    """prl_deal_decks (one lane per hand) against the NumPy statement of the same counter-based shuffle"""
    from pokerrl_amd.eval.lbr.BatchedLBR import deal_decks, deal_decks_host
    for n, nc, nd, seed, first in [(1000, 52, 9, 0, 0), (777, 52, 9, 12345678901234567, 5000), (300, 6, 3, 3, 1 << 40), (64, 24, 3, 9, 7)]:
        a, b = deal_decks(n, nc, nd, seed, first), deal_decks_host(n, nc, nd, seed, first)
        assert a.dtype == np.int8 and np.array_equal(a, b)
        assert all(len(set(row)) == nd for row in a.tolist()) and a.min() >= 0 and a.max() < nc
    assert np.array_equal(deal_decks(10, 52, 9, 1, 90), deal_decks(100, 52, 9, 1, 0)[90:])  # a hand's cards do not depend on the split


def test_deal_decks_emu(emu_lib):
    check_deal_decks()


@pytest.mark.gpu
def test_gpu_deal_decks():
    check_deal_decks()


@pytest.fixture()
def emu_lib(monkeypatch):
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    L = _native.bind(build_emu.build())
    monkeypatch.setattr(_native, "lib", lambda: L)
    monkeypatch.setattr(_native, "require_device", lambda: None)
    return L


def test_lbr_standard_leduc_vs_reference_emu(emu_lib, tmp_path):
    check_case("StandardLeduc", tmp_path, max_hands=120)


def test_lbr_nl_leduc_vs_reference_emu(emu_lib, tmp_path):
    check_case("DiscretizedNLLeduc", tmp_path, max_hands=80)


def test_lbr_holdem_vs_reference_emu(emu_lib, tmp_path):
    check_case("DiscretizedNLHoldem", tmp_path, max_hands=4)


def test_poker_range_matches_reference_semantics():
    """R1: card probabilities sum to N_HOLE_CARDS, blockers are removed, an impossible range resets to uniform."""
    from pokerrl_amd.game.PokerRange import PokerRange
    bldr = HistoryEnvBuilder(env_cls=DiscretizedNLHoldem, env_args=DiscretizedNLHoldem.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bet_sets.B_3))
    r = PokerRange(bldr)
    assert r.range.dtype == np.float32 and r.range.shape == (1326,)
    r.set_cards_to_zero_prob(np.array([[12, 3], [0, 0]], dtype=np.int8))
    cp = r.get_card_probs()
    assert cp[12 * 4 + 3] == 0 and cp[0] == 0 and abs(float(cp.sum()) - 2.0) < 1e-4
    assert PokerRange.get_possible_range_idxs(bldr.rules, bldr.lut_holder, np.array([[12, 3], [0, 0]], dtype=np.int8)).shape[0] == 1326 - 101
    r.mul_and_norm(np.zeros(1326, dtype=np.float32))
    assert np.all(r.range == np.float32(1.0 / 1326))
    assert PokerRange.get_range_size(2, 52) == 1326


@pytest.mark.gpu
@pytest.mark.parametrize("tag", list(CASES))
def test_gpu_lbr_vs_reference(tag, tmp_path):
    check_case(tag, tmp_path)


@pytest.mark.gpu
def test_gpu_lbr_master_logs_mean_and_confidence(tmp_path):
    game_cls, agent_bets, lbr_kwargs = CASES["StandardLeduc"]
    t_prof = make_t_prof(game_cls, agent_bets, lbr_kwargs, 60, tmp_path)

    class Chief(ChiefBase):
        def pull_current_eval_strategy(self, last):
            return None, last

    chief = Chief(t_prof)
    m = LocalLBRMaster(t_prof=t_prof, chief_handle=chief)
    m.set_worker_handles(LocalLBRWorker(t_prof=t_prof, chief_handle=chief, eval_agent_cls=fx.make_agent_cls(EvalAgentBase, seed=7)))
    m.update_weights()
    np.random.seed(5)
    m.evaluate(iter_nr=0)
    vals, _ = chief.get_new_values()
    names = sorted(vals)
    assert any("LBR" in n and "Conf_lower95" in n for n in names) and any(n.endswith("LBR") or "LBR" in n for n in names)
