"""
Parity checks shared by the GPU suite (product library, `-m gpu`) and the CPU emulator suite (same kernel SOURCES built
with -DPRL_EMU, see tests/emu/prl_emu.h). Every check compares the library's results with the CPU oracle (oracle/) on the
same inputs, bit for bit, and -- where fixtures exist -- with values captured from the reference itself (tests/golden).
"""
import ctypes

import numpy as np
import pytest

import oracle
from helpers import GAMES, all_single_card_boards, env_args, golden
from pokerrl_amd import _native
from pokerrl_amd.game import bet_sets
from pokerrl_amd.game import games as G

STATE_FIELDS = ("reach", "ev", "ev_br", "strategy", "strat_f64", "regret", "avg", "avg_f64", "br_idx")
RANK_RULE = {"StandardLeduc": 0, "BigLeduc": 1, "DiscretizedNLLeduc_POT": 0, "DiscretizedNLLeduc_B3_short": 0}


def make_pair(L, game_cls, stack, bets, boards, variant, delay=0, max_raises=None):
    """(NativeTree, NativeSolver, Oracle) on the same flat tree (the product's builder feeds the oracle)."""
    args = env_args(game_cls, stack, bets)
    game = game_cls.native_game(args)
    if max_raises is not None:  # a smaller betting tree than MAX_N_RAISES_PER_ROUND gives
        for i, v in enumerate(max_raises):
            game.max_raises[i] = v
    t = _native.NativeTree(game, game_cls.native_rules(), boards, _lib=L)
    s = _native.NativeSolver(t, variant, delay, engine="levels", _lib=L)  # every per-node vector is compared below
    r = game_cls.RULES
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, r.N_HOLE_CARDS, r.N_CARDS_IN_DECK, r.N_SUITS,
                      r._RANK_RULE)
    o.cfr_reset(_native.VARIANTS[variant], delay)
    c = s.get("constants")
    assert c[0] == o.chance_prob and c[1] == o.eq_const
    return t, s, o


def assert_state_equal(s, o, tag, fields=STATE_FIELDS):
    for k in fields:
        a, b = s.get(k), np.asarray(getattr(o, k))
        assert np.array_equal(a, b), "%s: %s differs in %d entries, first %s" % (tag, k, int(np.sum(a != b)), np.argwhere(a != b)[:3].tolist())
    assert np.array_equal(s.exploitability(), o.exploitability), (tag, s.exploitability(), o.exploitability)


def check_cfr_vs_oracle(L, gkey, variant, n_iters, delay=0, check_every=1):
    cls, stack, bets = GAMES[gkey]
    t, s, o = make_pair(L, cls, stack, bets, all_single_card_boards(cls), variant, delay)
    assert_state_equal(s, o, "%s/%s it0" % (gkey, variant))
    for it in range(1, n_iters + 1):
        s.iteration()
        o.cfr_iteration()
        if it % check_every == 0 or it == n_iters:
            assert_state_equal(s, o, "%s/%s it%d" % (gkey, variant, it))
        if variant != "plus" or it > delay:
            assert np.array_equal(s.eval_avg(), o.eval_avg()), (gkey, variant, it)
    hist = s.get("expl_history")
    assert hist.shape == (n_iters + 1, 2)
    assert np.array_equal(hist[-1], o.exploitability)


def check_cfr_vs_reference_series(L, fixture, gkey, variant):
    """Logged exploitability (mean of seats x EV_NORMALIZER, _CFRBase.py:198-216,257-262) vs the reference's own log."""
    g = golden("cfr_%s.npz" % fixture)
    cls, stack, bets = GAMES[gkey]
    args = env_args(cls, stack, bets)
    t = _native.NativeTree(cls.native_game(args), cls.native_rules(), all_single_card_boards(cls), _lib=L)
    s = _native.NativeSolver(t, variant, 0, _lib=L)
    evn = float(g["ev_normalizer"])

    def logged(e):
        return (float(e[0]) * evn + float(e[1]) * evn) / 2

    curr, avg = g["curr_series"], g["avg_series"]
    assert logged(s.exploitability()) == curr[0, 1]
    for it in range(1, int(curr[-1, 0]) + 1):
        s.iteration()
        assert logged(s.exploitability()) == curr[it, 1], (fixture, it)
        row = avg[avg[:, 0] == it]
        assert len(row) == 1 and logged(s.eval_avg()) == row[0, 1], (fixture, it)
    # per-node arrays of the last snapshot, straight from the reference's tree
    it = int(curr[-1, 0])
    for k in ("reach", "ev", "ev_br", "regret", "strategy", "avg"):
        key = "it%d_%s" % (it, k)
        if key in g:
            assert np.array_equal(s.get(k), g[key]), (fixture, k)
        elif key + "_sha256" in g:
            from helpers import h32
            assert h32(s.get(k)) == str(g[key + "_sha256"]), (fixture, k)


def fhp_boards(n, seed=5, with_special=True):
    rng = np.random.RandomState(seed)
    seen, out = set(), []
    if with_special and n >= 3:
        for b in ([24, 25, 26, 27, 33], [32, 36, 40, 44, 48], [0, 5, 10, 15, 51]):  # quads, royal flush, dry board
            out.append(b)
            seen.add(tuple(b))
    while len(out) < n:
        b = tuple(sorted(int(x) for x in rng.choice(52, 5, replace=False)))
        if b not in seen:
            seen.add(b)
            out.append(list(b))
    return np.array(out[:n], dtype=np.int8)


def check_fhp_vs_oracle(L, n_boards, variant, n_iters, check_fields=STATE_FIELDS):
    boards = fhp_boards(n_boards)
    t, s, o = make_pair(L, G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards, variant)
    assert t.n_nodes == 5 + 15 * n_boards
    assert_state_equal(s, o, "FHP/%s it0" % variant, check_fields)
    for it in range(1, n_iters + 1):
        s.iteration()
        o.cfr_iteration()
        assert_state_equal(s, o, "FHP/%s it%d" % (variant, it), check_fields)
        assert np.array_equal(s.eval_avg(), o.eval_avg())
    return s, o


def multistreet_runouts(n_flops, n_turns, n_rivers, seed=9):
    """run-outs of a hold'em game that deals 3 + 1 + 1: n_flops flops, below each n_turns turn cards, below each n_rivers river cards
    (deal order = row order, no card twice in a row)"""
    rng = np.random.RandomState(seed)
    rows = []
    for _ in range(n_flops):
        flop = rng.choice(52, 3, replace=False)
        rest = [c for c in rng.permutation(52) if c not in flop]
        for t in rest[:n_turns]:
            rest2 = [c for c in rng.permutation(52) if c not in flop and c != t]
            for r in rest2[:n_rivers]:
                rows.append(list(flop) + [t, r])
    return np.array(rows, np.int8)


def check_multistreet_vs_oracle(L, game_cls, stack, bets, runouts, variant, n_iters, expect_runout_chain=False, max_raises=None):
    """public trees of games that deal on several streets (one chance level per street, children = the distinct prefixes
    of the listed run-outs), incl. all-in run-outs dealt as chance chains: every per-node vector, regrets, averages and the
    exploitability of the level-synchronous engine against the oracle, bit for bit, plus the structural invariants."""
    from oracle_tape import digest, same
    tape = _tape("multistreet", L, game_cls.__name__, stack, bets, np.asarray(runouts), variant, n_iters, max_raises)
    rec, live = tape is not None and tape.recording, tape is None or tape.live
    take = (lambda tag, fn: fn()) if tape is None else tape.take
    big = (lambda a: a) if (tape is None or (tape.live and not tape.recording)) else digest
    if rec or not live:  # the oracle alone (recording, no GPU) / the solver alone (replay)
        args = env_args(game_cls, stack, bets)
        game = game_cls.native_game(args)
        if max_raises is not None:
            for i, v in enumerate(max_raises):
                game.max_raises[i] = v
        t = _native.NativeTree(game, game_cls.native_rules(), runouts, _lib=L)
        s = o = None
        if rec:
            o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, game_cls.RULES.N_HOLE_CARDS, game_cls.RULES.N_CARDS_IN_DECK,
                              game_cls.RULES.N_SUITS, game_cls.RULES._RANK_RULE)
            o.cfr_reset(_native.VARIANTS[variant], 0)
        else:
            s = _native.NativeSolver(t, variant, 0, engine="levels", _lib=L)
    else:
        t, s, o = make_pair(L, game_cls, stack, bets, runouts, variant, max_raises=max_raises)

    def state(tag):
        for k in STATE_FIELDS:
            want = take("%s/%s" % (tag, k), lambda: big(np.asarray(getattr(o, k))))
            if s is not None:
                same(s.get(k), want, "%s: %s" % (tag, k))
        want = take(tag + "/expl", lambda: np.array(o.exploitability, np.float32))
        if s is not None:
            assert np.array_equal(s.exploitability(), want), (tag, s.exploitability(), want)
    kind, bid, rnd, par = t.field("kind"), t.field("board_id"), t.field("round"), t.field("parent")
    n_chance_levels = len({int(np.sum(t.board_rows[bid[c]] >= 0)) for c in np.where(par >= 0)[0] if kind[par[c]] == 1})
    assert n_chance_levels == sum(1 for k in game_cls.native_rules().board_cards_in_round[1:game_cls.native_rules().n_rounds] if k > 0)
    assert t.n_boards > len(runouts) or n_chance_levels == 1          # prefix rows of every street
    sd = np.where(kind == 3)[0]
    assert np.all(np.sum(t.board_rows[bid[sd]] >= 0, axis=1) == t.board_len)  # every showdown sits on a complete board
    if expect_runout_chain:
        assert np.any((kind == 1) & (kind[np.maximum(par, 0)] == 1))      # a chance node below a chance node: the all-in run-out
    assert s is None or s.engine == "levels"
    state("multistreet it0")
    for it in range(1, n_iters + 1):
        if s is not None:
            s.iteration()
        if live:
            o.cfr_iteration()
        state("multistreet it%d" % it)
        want = take("it%d/eval_avg" % it, lambda: np.array(o.eval_avg(), np.float32))
        if s is not None:
            assert np.array_equal(s.eval_avg(), want)
    if tape is not None:
        tape.close()
    if s is None:
        return t, s, o
    # ValueFiller.py:98: zero-sum at every node (float64 accumulation of the float32 products)
    ev, reach = s.get("ev"), s.get("reach")
    zs = np.sum(ev.astype(np.float64) * reach.astype(np.float64), axis=(1, 2))
    assert np.max(np.abs(zs)) < 1e-3
    e = s.exploitability()
    assert np.all(e >= -1e-3)
    return t, s, o


def check_br_of_given_strategy(L, gkey, seed, f64):
    """LocalBRMaster semantics (LocalBRMaster.py:67-80): fill an arbitrary strategy, reach, EV + best response."""
    cls, stack, bets = GAMES[gkey]
    t, s, o = make_pair(L, cls, stack, bets, all_single_card_boards(cls), "vanilla")
    rng = np.random.RandomState(seed)
    strat = np.zeros((t.n_cols, t.range_size), np.float64 if f64 else np.float32)
    first_col, n_ch, kind = t.field("first_col"), t.field("n_children"), t.field("kind")
    for n in np.where(kind == 0)[0]:
        a = n_ch[n]
        x = rng.random_sample((a, t.range_size))  # fill_random_random (StrategyFiller.py:67-86)
        x /= x.sum(axis=0, keepdims=True)
        strat[first_col[n]:first_col[n] + a] = x
    s.set_strategy(strat)
    o.set_strategy(strat.astype(np.float64), f64)
    s.compute_ev()
    o.compute_ev()
    assert_state_equal(s, o, "BR %s" % gkey, ("reach", "ev", "ev_br", "br_idx"))
    e = s.exploitability()
    assert np.all(e >= -1e-3)  # a best response never does worse than the strategy itself


def check_hand_rank_golden(L):
    g = golden("handrank.npz")
    b = np.ascontiguousarray(g["boards"])
    out = np.empty((b.shape[0], 1326), np.int32)
    _native.check(L.prl_hand_rank_boards(b.ctypes.data_as(ctypes.c_void_p), b.shape[0], out.ctypes.data_as(ctypes.c_void_p)), L)
    assert np.array_equal(out, g["ranks"])
    assert np.array_equal(out, oracle.rank_boards(b))
    assert int(np.sum(out[0] == -1)) == 245  # blocked hands per 5-card board


def check_hand_rank_checksums(L, n_chunks=None):
    import itertools
    g = golden("handrank_exhaustive.npz")
    chunk = int(g["chunk_boards"])
    ref = g["checksums"]
    n_boards = 2598960 if n_chunks is None else n_chunks * chunk
    it = itertools.chain.from_iterable(itertools.islice(itertools.combinations(range(52), 5), n_boards))
    boards = np.fromiter(it, dtype=np.int8, count=5 * n_boards).reshape(n_boards, 5)
    L.prl_hand_rank_checksums.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
    L.prl_hand_rank_checksums.restype = ctypes.c_int32
    out = np.zeros((n_boards + chunk - 1) // chunk, np.uint64)
    _native.check(L.prl_hand_rank_checksums(boards.ctypes.data_as(ctypes.c_void_p), n_boards, chunk, out.ctypes.data_as(ctypes.c_void_p)), L)
    assert np.array_equal(out, ref[:out.shape[0]])
    return out.shape[0]


FUSED_FIELDS = ("regret", "avg")


VARIANT_ID = {"vanilla": 0, "plus": 1, "linear": 2}


def fhp_game(stack=20000, flop_raises=None):
    """Flop5Holdem's PrlGame; flop_raises overrides MAX_N_RAISES_PER_ROUND[FLOP] (games.py:239: 2)"""
    g = G.Flop5Holdem.native_game(env_args(G.Flop5Holdem, stack, bet_sets.POT_ONLY))
    if flop_raises is not None:
        g.max_raises[1] = flop_raises
    return g


def check_fused_vs_oracle(L, n_boards, n_iters, delay=0, variant="plus", stack=20000, flop_raises=None, nodes_per_board=15):
    """Fused board-block engine (per-node vectors on chip) against the oracle: every regret / average column of every
    board, the strategy implied by the regrets, current- and average-strategy exploitability, after every iteration.
    stack / flop_raises select other betting structures (other registered board-subtree shapes, csrc/prl_fhp.h)."""
    boards = fhp_boards(n_boards)
    t = _native.NativeTree(fhp_game(stack, flop_raises), G.Flop5Holdem.native_rules(), boards, _lib=L)
    assert t.n_nodes == 5 + nodes_per_board * n_boards
    s = _native.NativeSolver(t, variant, delay, engine="auto", _lib=L)  # AUTO must pick the fused engine for a registered shape
    assert s.engine == "fused"
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, 2, 52, 4, 2)
    o.cfr_reset(VARIANT_ID[variant], delay)
    assert np.array_equal(s.exploitability(), o.exploitability)
    fields = FUSED_FIELDS + ("strategy",) + (() if variant == "plus" else ("avg_sum",))
    for it in range(1, n_iters + 1):
        s.iteration()
        o.cfr_iteration()
        for k in fields:
            a, b = s.get(k), np.asarray(getattr(o, k))
            assert np.array_equal(a, b), "fused it%d: %s differs in %d entries" % (it, k, int(np.sum(a != b)))
        assert np.array_equal(s.exploitability(), o.exploitability), it
        if it > delay:
            assert np.array_equal(s.eval_avg(), o.eval_avg()), it
    return s, o


def check_fused_vs_fixture(L, name):
    """Fused engine against an oracle-generated fixture (tests/golden/make_fhp_golden.py): history + array hashes."""
    from helpers import h32
    g = golden("%s.npz" % name)
    boards = fhp_boards(int(g["n_boards"]), seed=int(g["seed"]))
    assert h32(boards) == str(g["boards_sha256"])
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards, _lib=L)
    variant = str(g["variant"])
    s = _native.NativeSolver(t, variant, int(g["delay"]) if "delay" in g else 0, engine="fused", _lib=L)
    s.iterations(int(g["n_iters"]))
    assert np.array_equal(s.get("expl_history"), g["expl_history"])
    assert np.array_equal(s.eval_avg(), g["eval_avg"])
    assert h32(s.get("regret")) == str(g["regret_sha256"])
    assert h32(s.get("avg")) == str(g["avg_sha256"])


def check_fused_br_vs_oracle(L, n_boards, seed=3):
    """Exact best response of an explicit strategy on the fused engine (LocalBRMaster.py:67-80: fill, reach, EV + BR, root
    exploitability) against the oracle, float32 (the best-response-only pass: strategy streamed like regrets) and float64."""
    boards = fhp_boards(n_boards)
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards, _lib=L)
    s = _native.NativeSolver(t, "plus", 0, engine="fused", _lib=L)
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, 2, 52, 4, 2)
    o.cfr_reset(1, 0)
    nt = t.n_cols - 14 * n_boards
    for f64 in (False, True):
        strat = seeded_strategy_for_sharding(nt, n_boards, t.range_size, seed + int(f64))
        if f64:
            rng = np.random.RandomState(seed)
            strat = strat.astype(np.float64) * (1.0 + 1e-9 * rng.random_sample(strat.shape))  # not float32-representable
        s.set_strategy(strat)
        o.set_strategy(strat.astype(np.float64), f64)
        s.compute_ev()
        o.compute_ev()
        assert np.array_equal(s.exploitability(), o.exploitability), (f64, s.exploitability(), o.exploitability)
        assert np.array_equal(s.get("strategy"), strat.astype(np.float64))
    s.reset()  # and the solver iterates again afterwards
    o.cfr_reset(1, 0)
    s.iterations(2)
    o.cfr_iteration(); o.cfr_iteration()
    assert np.array_equal(s.exploitability(), o.exploitability)


def check_fused_batched_vs_oracle(L, n_boards, n_iters, delay=0, variant="plus"):
    """prl_solver_iterations(n) on the fused engine folds every closing evaluation into the next iteration's first board
    pass; the exploitability history and the final state must equal the oracle's (= n single iteration() calls)."""
    boards = fhp_boards(n_boards)
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards, _lib=L)
    s = _native.NativeSolver(t, variant, delay, engine="fused", _lib=L)
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, 2, 52, 4, 2)
    o.cfr_reset(VARIANT_ID[variant], delay)
    want = [np.array(o.exploitability, np.float32)]
    for _ in range(n_iters):
        o.cfr_iteration()
        want.append(np.array(o.exploitability, np.float32))
    s.iterations(n_iters - 1)
    s.iterations(1)  # a batch of one closes with its own evaluation pass
    assert np.array_equal(s.get("expl_history"), np.stack(want))
    for k in FUSED_FIELDS + ("strategy",) + (() if variant == "plus" else ("avg_sum",)):
        assert np.array_equal(s.get(k), np.asarray(getattr(o, k))), k
    assert np.array_equal(s.eval_avg(), o.eval_avg())


def check_fused_vs_levels(L, n_boards, n_iters, seed=11, variant="plus"):
    """Same tree solved by both engines of the library: bit-identical regrets, averages and exploitability history."""
    boards = fhp_boards(n_boards, seed=seed)
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards, _lib=L)
    a = _native.NativeSolver(t, variant, 0, engine="fused", _lib=L)
    b = _native.NativeSolver(t, variant, 0, engine="levels", _lib=L)
    assert (a.engine, b.engine) == ("fused", "levels")
    a.iterations(n_iters)
    b.iterations(n_iters)
    assert np.array_equal(a.get("expl_history"), b.get("expl_history"))
    for k in FUSED_FIELDS + (() if variant == "plus" else ("avg_sum",)):
        assert np.array_equal(a.get(k), b.get(k)), k
    assert np.array_equal(a.eval_avg(), b.eval_avg())
    # exact best response of an explicit strategy (LocalBRMaster semantics) through both engines
    rng = np.random.RandomState(seed)
    strat = np.zeros((t.n_cols, t.range_size), np.float32)
    first_col, n_ch, kind = t.field("first_col"), t.field("n_children"), t.field("kind")
    for n in np.where(kind == 0)[0]:
        x = rng.random_sample((n_ch[n], t.range_size)).astype(np.float32)
        strat[first_col[n]:first_col[n] + n_ch[n]] = x / x.sum(axis=0, keepdims=True)
    for sol in (a, b):
        sol.set_strategy(strat)
        sol.compute_ev()
    assert np.array_equal(a.exploitability(), b.exploitability())
    return a, b


def check_checkpoint_resume(L, fused, variant="plus", n_before=3, n_after=2):
    """save_state after n_before iterations, load into a fresh solver, run n_after more: identical to n_before + n_after
    straight (regrets, averages, exploitability history, average-strategy exploitability)."""
    if fused:
        boards = fhp_boards(3)
        args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
        t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards, _lib=L)
        mk = lambda: _native.NativeSolver(t, variant, 1, engine="fused", _lib=L)  # noqa: E731
    else:
        cls, stack, bets = GAMES["StandardLeduc"]
        t = _native.NativeTree(cls.native_game(env_args(cls, stack, bets)), cls.native_rules(), all_single_card_boards(cls), _lib=L)
        mk = lambda: _native.NativeSolver(t, variant, 1, engine="levels", _lib=L)  # noqa: E731
    a = mk()
    a.iterations(n_before + n_after)
    b = mk()
    b.iterations(n_before)
    blob = b.save_state()
    c = mk()
    c.iterations(1)  # some other state that the load must overwrite
    c.load_state(blob)
    assert c.iter == n_before
    c.iterations(n_after)
    for k in ("regret", "avg", "expl_history") + (() if variant == "plus" else ("avg_sum",)):
        assert np.array_equal(a.get(k), c.get(k)), k
    assert np.array_equal(a.eval_avg(), c.eval_avg())
    with pytest.raises(Exception):
        mk2 = _native.NativeSolver(t, variant, 0, engine="fused" if fused else "levels", _lib=L)  # different delay
        mk2.load_state(blob)


def seeded_strategy_for_sharding(n_trunk_cols, n_boards, R, seed):
    """float32 [n_cols, R] column-major strategy of a Flop5Holdem tree (trunk: SB {fold, raise}, BB {fold, call}; per board the 6
    decision nodes with 2,2,3,2,3,2 actions), normalised per node and hand; the same array whatever the sharding."""
    rng = np.random.RandomState(seed)
    sizes = [2] * (n_trunk_cols // 2) + [2, 2, 3, 2, 3, 2] * n_boards
    cols = []
    for a in sizes:
        x = rng.random_sample((a, R)).astype(np.float32)
        cols.append(x / x.sum(axis=0, keepdims=True))
    return np.concatenate(cols).astype(np.float32)


def check_iterations_many(L, n_iters=5):
    """prl_solver_iterations_many: several DIFFERENT small trees (games, stack sizes, variants) advanced by one launch, one
    workgroup each -- every solver must end in exactly the state (all arrays, history) the C oracle reaches on its tree, and a
    second batch must continue from there."""
    from pokerrl_amd.game import games as G
    cases = [(G.StandardLeduc, 13, None, "plus", 0), (G.StandardLeduc, 7, None, "vanilla", 0), (G.StandardLeduc, 20, None, "linear", 0),
             (G.StandardLeduc, 13, None, "plus", 2), (G.StandardLeduc, 5, None, "plus", 0)]
    trios = [make_pair(L, cls, stack, bets, all_single_card_boards(cls), variant, delay) for cls, stack, bets, variant, delay in cases]
    solvers = [s for _t, s, _o in trios]
    _native.NativeSolver.iterations_many(solvers, 3)
    _native.NativeSolver.iterations_many(solvers, n_iters - 3)
    for i, (_t, s, o) in enumerate(trios):
        for _ in range(n_iters):
            o.cfr_iteration()
        assert s.iter == n_iters
        assert_state_equal(s, o, "many[%d]" % i)
        hist = s.get("expl_history")
        assert hist.shape == (n_iters + 1, 2) and np.array_equal(hist[-1], o.exploitability)
    # a solver advanced alone afterwards keeps going from the batched state
    _t, s, o = trios[0]
    s.iterations(2)
    o.cfr_iteration(); o.cfr_iteration()
    assert_state_equal(s, o, "many[0] + 2")


def make_streets_pair(L, game_cls, stack, runouts, variant, delay=0, max_raises=None, tape=None, bets=None, **solver_kw):
    """(tree, fused solver on the per-street engine, oracle) on one multi-street flat tree (csrc/prl_st.h). With an oracle tape (tests/oracle_tape.py):
    no oracle when the tape is replayed, no solver when it is being recorded (the generator runs without a GPU)."""
    args = env_args(game_cls, stack, bets)
    game = game_cls.native_game(args)
    if max_raises is not None:
        for i, v in enumerate(max_raises):
            game.max_raises[i] = v
    t = _native.NativeTree(game, game_cls.native_rules(), runouts, _lib=L)
    s = None
    if tape is None or not tape.recording:
        s = _native.NativeSolver(t, variant, delay, engine="auto", _lib=L, **solver_kw)
        assert s.engine == "fused", "engine=auto must take the per-street fused engine for a multi-street tree of registered street shapes"
    o = None
    if tape is None or tape.live:
        r = game_cls.RULES
        o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, r.N_HOLE_CARDS, r.N_CARDS_IN_DECK, r.N_SUITS, r._RANK_RULE)
        o.cfr_reset(_native.VARIANTS[variant], delay)
    return t, s, o


def _tape(kind, L, *parts):
    """an oracle tape for the GPU suite's checks (the emulator suite of the container keeps the oracle live: its problems are small)"""
    import oracle_tape
    if L is not None and L is not _native_product_lib():
        return None
    return oracle_tape.Tape(kind, *parts)


def _native_product_lib():
    try:
        return _native.lib()
    except Exception:  # noqa: BLE001
        return None


def check_set_strategy_device(L, t, make_solver, to_device=None, seed=9):
    """prl_solver_set_strategy_device (the agent's probabilities [n_decision_nodes][R][A] float32 already in device memory, scattered into the solver's
    columns on the device) = prl_solver_set_strategy with the same numbers gathered into [n_cols][R] on the host: strategy read back, exploitability.
    to_device: array -> (object that keeps the device copy alive, its address); None: the emulator, where host memory IS device memory"""
    kind, nch, fc, col_action = t.field("kind"), t.field("n_children"), t.field("first_col"), t.field("col_action")
    dec = np.flatnonzero(kind == 0)
    n_act = int(col_action.max()) + 1
    rng = np.random.RandomState(seed)
    probs = rng.random_sample((len(dec), t.range_size, n_act)).astype(np.float32)
    probs /= probs.sum(axis=2, keepdims=True)
    cols = np.zeros((t.n_cols, t.range_size), np.float32)
    for k, n in enumerate(dec):
        for j in range(nch[n]):
            cols[fc[n] + j] = probs[k, :, col_action[fc[n] + j]]
    a, b = make_solver(), make_solver()
    a.set_strategy(cols)
    keep, ptr = (probs, probs.ctypes.data) if to_device is None else to_device(probs)
    b.set_strategy_device(ptr, n_act)
    assert np.array_equal(a.get("strategy"), b.get("strategy"))
    a.compute_ev(); b.compute_ev()
    assert np.array_equal(a.exploitability(), b.exploitability()) and np.all(a.exploitability() > 0)
    del keep
    return a.engine


def check_streets_br_vs_oracle(L, game_cls, stack, runouts, max_raises=None, seed=5, bets=None):
    """Exact best response of an explicit strategy on a multi-street tree, per-street engine (LocalBRMaster.py:67-80): a seeded strategy given as
    float32 (played as float32 from the engine's internal column order) and as float64 columns in the flat tree's DFS order; exploitability
    against the oracle, the strategy read back unchanged; iterating again after reset()"""
    tape = _tape("streets_br", L, game_cls.__name__, stack, np.asarray(runouts), max_raises, seed, *(() if bets is None else (tuple(bets),)))
    t, s, o = make_streets_pair(L, game_cls, stack, runouts, "plus", 0, max_raises, tape=tape, bets=bets)
    live = tape is None or tape.live
    take = (lambda tag, fn: fn()) if tape is None else tape.take
    kind, nch, fc = t.field("kind"), t.field("n_children"), t.field("first_col")
    rng = np.random.RandomState(seed)
    strat = np.empty((t.n_cols, t.range_size), np.float32)
    for n in np.where(kind == 0)[0]:
        x = rng.random_sample((nch[n], t.range_size)).astype(np.float32)
        strat[fc[n]:fc[n] + nch[n]] = x / x.sum(axis=0, keepdims=True)
    for f64 in (False, True):
        st = strat.astype(np.float64) * (1.0 + 1e-9 * rng.random_sample(strat.shape)) if f64 else strat
        if live:
            o.set_strategy(np.asarray(st, dtype=np.float64), f64)
            o.compute_ev()
        want = take("br/f64=%d" % f64, lambda: np.array(o.exploitability, np.float32))
        if s is not None:
            s.set_strategy(st)
            s.compute_ev()
            assert np.array_equal(s.exploitability(), want), (f64, s.exploitability(), want)
            assert np.array_equal(s.get("strategy"), np.asarray(st, dtype=np.float64))
    if live:
        o.cfr_reset(1, 0)
        o.cfr_iteration(); o.cfr_iteration()
    want = take("after reset/expl", lambda: np.array(o.exploitability, np.float32))
    if s is not None:
        s.reset()
        s.iterations(2)
        assert np.array_equal(s.exploitability(), want)
    if tape is not None:
        tape.close()
    return t


def check_streets_vs_oracle(L, game_cls, stack, runouts, variant, n_iters, delay=0, max_raises=None, batched=False, bets=None):
    """SURVEY 8f-4 on the per-street fused engine (csrc/prl_st.h): regrets, averages, the strategy implied by the regrets, current- and
    average-strategy exploitability after every iteration (batched: the exploitability history of prl_solver_iterations(n) and the
    final state), bit for bit against the oracle -- in the flat tree's DFS column order, which the engine does not use internally.
    On the GPU box the oracle's side comes from a tape (tests/oracle_tape.py) when one was recorded for exactly this problem."""
    from oracle_tape import digest, same
    tape = _tape("streets", L, game_cls.__name__, stack, np.asarray(runouts), variant, n_iters, delay, max_raises, batched, *(() if bets is None else (tuple(bets),)))
    t, s, o = make_streets_pair(L, game_cls, stack, runouts, variant, delay, max_raises, tape=tape, bets=bets)
    live = tape is None or tape.live
    take = (lambda tag, fn: fn()) if tape is None else tape.take
    big = (lambda a: a) if (tape is None or (tape.live and not tape.recording)) else digest  # arrays entry by entry when the oracle is here, digests on tape

    def expl(tag):
        want = take(tag, lambda: np.array(o.exploitability, np.float32))
        if s is not None:
            assert np.array_equal(s.exploitability(), want), (tag, s.exploitability(), want)

    def eval_avg(tag):
        want = take(tag, lambda: np.array(o.eval_avg(), np.float32))
        if s is not None:
            assert np.array_equal(s.eval_avg(), want), (tag, s.eval_avg(), want)

    expl("reset/expl")
    fields = FUSED_FIELDS + ("strategy",) + (() if variant == "plus" else ("avg_sum",))

    def same_state(tag):
        for k in fields:
            want = take("%s/%s" % (tag, k), lambda: big(np.asarray(getattr(o, k))))
            if s is not None:
                same(s.get(k), want, "streets %s: %s" % (tag, k))
    if batched:
        def run():
            want = [np.array(o.exploitability, np.float32)]
            for _ in range(n_iters):
                o.cfr_iteration()
                want.append(np.array(o.exploitability, np.float32))
            return np.stack(want)
        want = take("batched/expl_history", run)
        if s is not None:
            s.iterations(n_iters - 1)
            s.iterations(1)
            assert np.array_equal(s.get("expl_history"), want), (s.get("expl_history"), want)
        same_state("batched")
        eval_avg("batched/eval_avg")
    else:
        for it in range(1, n_iters + 1):
            if s is not None:
                s.iteration()
            if live:
                o.cfr_iteration()
            same_state("it%d" % it)
            expl("it%d/expl" % it)
            if it > delay:
                eval_avg("it%d/eval_avg" % it)
    if tape is not None:
        tape.close()
    if s is None:
        return t, s, o
    # prl_solver_get_cols on this engine: any window of flat-tree columns, gathered from the internal order
    nc = t.n_cols
    for name in ("regret", "avg") + (() if variant == "plus" else ("avg_sum",)):
        full = s.get(name)
        for c0, n in ((0, nc), (nc // 3, min(37, nc - nc // 3)), (nc - 1, 1)):
            assert np.array_equal(s.get_cols(name, c0, n), full[c0:c0 + n]), (name, c0, n)
    return t, s, o


def check_fused_avg_f32(L, n_boards, n_iters):
    """PRL_SOLVER_AVG_F32 (opt-in): the board columns' running average stored as float32. Everything else -- regrets, strategies, the
    current-strategy exploitability history -- stays bit-exact to the oracle; the average equals the reference's recurrence with one rounding
    per iteration (restated here from the oracle's strategies: avg32 <- float32(m_old * float64(avg32) + m_new * strategy), CFRPlus.py:65-87);
    the trunk's columns stay float64 = the oracle's; the average-strategy exploitability stays within 1e-5 relative of the float64 one."""
    boards = fhp_boards(n_boards)
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards, _lib=L)
    s = _native.NativeSolver(t, "plus", 0, engine="fused", _lib=L, avg_dtype="f32")
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, 2, 52, 4, 2)
    o.cfr_reset(1, 0)
    nt = t.n_cols - 14 * n_boards
    a32 = np.zeros((t.n_cols - nt, t.range_size), np.float32)
    for it in range(n_iters):
        s.iteration()
        o.cfr_iteration()
        strat = np.asarray(o.strategy)[nt:]
        if it == 0:
            a32 = strat.astype(np.float32)
        else:
            cw, nw = sum(range(1, it + 1)), it + 1
            a32 = (cw / (cw + nw) * a32.astype(np.float64) + nw / (cw + nw) * strat).astype(np.float32)
        assert np.array_equal(s.get("regret"), np.asarray(o.regret)), it
        assert np.array_equal(s.exploitability(), o.exploitability), it
        avg = s.get("avg")
        assert np.array_equal(avg[:nt], np.asarray(o.avg)[:nt]), (it, "trunk average")
        assert np.array_equal(avg[nt:], a32.astype(np.float64)), (it, "board average: float32 recurrence")
        e32, e64 = s.eval_avg(), o.eval_avg()
        assert np.allclose(e32, e64, rtol=1e-5, atol=0), (it, e32, e64)
    return s.eval_avg(), o.eval_avg()


# ---- weighted boards / suit isomorphism (prl_solver_create_weighted; include/pokerrl_hip.h) ---------------------------------------------
def suit_orbit(board, n_suits=4):
    """the distinct boards a board maps to under the suit permutations (sorted cards, lexicographic order)"""
    from itertools import permutations
    out = set()
    for perm in permutations(range(n_suits)):
        out.add(tuple(sorted((c // n_suits) * n_suits + perm[c % n_suits] for c in board)))
    return sorted(out)


_ALL_CLASSES = []


def iso_classes(n_classes, seed=3):
    """a few suit classes of Flop5Holdem boards (representatives + orbit sizes), incl. the three orbit sizes 4 / 12 / 24"""
    from pokerrl_amd.game import board_enum
    if not _ALL_CLASSES:  # (8 s of enumeration: once per process)
        _ALL_CLASSES.append(board_enum.single_deal_board_classes(G.Flop5Holdem))
    reps, mult = _ALL_CLASSES[0]
    rng = np.random.RandomState(seed)
    pick = [int(np.where(mult == m)[0][rng.randint(np.sum(mult == m))]) for m in (4, 12, 24)]
    while len(pick) < n_classes:
        i = int(rng.randint(len(reps)))
        if i not in pick:
            pick.append(i)
    pick = sorted(pick[:n_classes])
    return reps[pick], mult[pick]


def fhp_tree_of(L, boards):
    return _native.NativeTree(G.Flop5Holdem.native_game(env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)), G.Flop5Holdem.native_rules(),
                              np.ascontiguousarray(boards, np.int8), _lib=L)


def classes_of(boards):
    """(representatives, orbit sizes) of the suit classes a few boards belong to, without enumerating the game (smoke test)"""
    orbits = sorted({tuple(suit_orbit([int(c) for c in b])[0]): len(suit_orbit([int(c) for c in b])) for b in boards}.items())
    return np.array([o[0] for o in orbits], np.int8), np.array([o[1] for o in orbits], np.int32)


def check_weighted_vs_oracle(L, n_classes, n_iters, variant="plus", symmetrize=True, classes=None):
    """the fused engine on suit-class representatives with multiplicities (+ orbit-mean chance values) against the oracle's restatement: every regret /
    average column, current- and average-strategy exploitability, bit for bit"""
    from pokerrl_amd.game import board_enum
    reps, mult = classes if classes is not None else iso_classes(n_classes)
    t = fhp_tree_of(L, reps)
    s = _native.NativeSolver(t, variant, 0, _lib=L, board_mult=mult, symmetrize="subset" if symmetrize else False)  # a few classes, not the game
    assert s.engine == "fused"
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, t.board_rows, 2, 52, 4, 2)
    w = o.set_board_weights(mult)
    if symmetrize:
        o.set_symmetrize(board_enum.hand_suit_classes(G.Flop5Holdem))
    o.cfr_reset(VARIANT_ID[variant], 0)
    assert s.get("constants")[0] == oracle.chance_prob_f32(int(mult.sum()), 52, 2, 5) and w[0] == s.get("constants")[0] * np.float32(mult[0])
    assert np.array_equal(s.exploitability(), o.exploitability)
    for it in range(1, n_iters + 1):
        s.iteration()
        o.cfr_iteration()
        for k in FUSED_FIELDS + (() if variant == "plus" else ("avg_sum",)):
            assert np.array_equal(s.get(k), np.asarray(getattr(o, k))), "weighted it%d: %s" % (it, k)
        assert np.array_equal(s.exploitability(), o.exploitability), it
        assert np.array_equal(s.eval_avg(), o.eval_avg()), it
    return s, o


def iso_vs_full(make_iso, make_full, n_classes, n_iters, rtol=2e-5):
    """suit isomorphism is EXACT in exact arithmetic: the class solve (representatives x multiplicities, orbit-mean chance values) against the
    solve of the full suit-closed board list the classes stand for (every board listed, weight 1): exploitability of the current and the
    average strategy after every iteration agree to float32 accumulation noise. make_*(boards, mult or None) -> object with iteration() /
    exploitability() / eval_avg()."""
    reps, mult = iso_classes(n_classes)
    full = [b for r in reps for b in suit_orbit([int(c) for c in r])]
    assert len(full) == int(mult.sum()) and len(set(full)) == len(full)
    a, b = make_iso(reps, mult), make_full(np.array(full, np.int8), None)
    ea, eb = np.asarray(a.exploitability(), np.float64), np.asarray(b.exploitability(), np.float64)
    assert np.all(np.abs(ea - eb) <= rtol * np.abs(eb)), (ea, eb)
    for it in range(n_iters):
        a.iteration(); b.iteration()
        ea, eb = np.asarray(a.exploitability(), np.float64), np.asarray(b.exploitability(), np.float64)
        assert np.all(np.abs(ea - eb) <= rtol * np.abs(eb)), (it, ea, eb)
        va, vb = np.asarray(a.eval_avg(), np.float64), np.asarray(b.eval_avg(), np.float64)
        assert np.all(np.abs(va - vb) <= rtol * np.abs(vb)), (it, va, vb)
    return a, b


def check_weighted_checkpoint(L):
    """save_state / load_state of a weighted (suit-class) solver: a resumed run continues bit-identically; a blob does not load into a solver whose
    multiplicities differ (they are part of the fingerprint)"""
    reps, mult = iso_classes(4)
    t = fhp_tree_of(L, reps)
    mk = lambda m: _native.NativeSolver(t, "plus", 0, _lib=L, board_mult=m, symmetrize="subset")  # noqa: E731
    a = mk(mult)
    a.iterations(3)
    b = mk(mult)
    b.iterations(2)
    blob = b.save_state()
    c = mk(mult)
    c.load_state(blob)
    c.iterations(1)
    for k in ("regret", "avg", "expl_history"):
        assert np.array_equal(a.get(k), c.get(k)), k
    assert np.array_equal(a.eval_avg(), c.eval_avg())
    other = mult.copy()
    other[0] = 12 if other[0] != 12 else 24
    with pytest.raises(_native.NativeError, match="orbit"):  # suit classes are checked, not trusted: a multiplicity that is not the orbit's size
        mk(other)
    with pytest.raises(_native.NativeError):  # ... and as plain weighted boards (no orbit means) the problem is another one: the blob does not load
        _native.NativeSolver(t, "plus", 0, _lib=L, board_mult=other, symmetrize=False).load_state(blob)


def check_symmetrize_is_validated(L):
    """prl_solver_create_weighted with `symmetrize` checks its premise (round 5's advisor finding): every listed board the representative of its suit class,
    its multiplicity the orbit's size, and -- unless a subset is declared -- the whole game covered. Weighted boards that are not suit classes take symmetrize=False."""
    reps, mult = iso_classes(4)
    t = fhp_tree_of(L, reps)
    mk = lambda tree, m, sym: _native.NativeSolver(tree, "plus", 0, _lib=L, board_mult=m, symmetrize=sym)  # noqa: E731
    mk(t, mult, "subset")
    with pytest.raises(_native.NativeError, match="do not cover the game"):
        mk(t, mult, True)
    member = np.array(suit_orbit([int(c) for c in reps[1]])[-1], np.int8)  # another member of class 1: not its representative
    assert not np.array_equal(member, reps[1])
    boards = reps.copy()
    boards[1] = member
    with pytest.raises(_native.NativeError, match="not the representative"):
        mk(fhp_tree_of(L, boards), mult, "subset")
    mk(fhp_tree_of(L, boards), mult, False)  # importance-style weights on arbitrary boards: fine without the orbit means
    with pytest.raises(ValueError):
        _native.NativeSolver(t, "plus", 0, _lib=L, board_mult=mult, symmetrize="subset", shard=(2, 0, None))


def check_streets_avg_f32(L, game_cls, stack, runouts, n_iters, max_raises=None, batched=False, bets=None):
    """PRL_SOLVER_AVG_F32 on the per-street fused engine (round 5: feature parity with the single-deal board pass): the street columns' running
    average stored as float32 -- regrets, current-strategy exploitability history: bit-exact to the oracle; the average = the reference's
    recurrence with one float32 rounding per iteration, restated here from the oracle's strategies; the trunk's columns stay float64 = the
    oracle's; average-strategy exploitability within 1e-5 relative of the float64 one."""
    from oracle_tape import digest, same
    tape = _tape("streets_avg_f32", L, game_cls.__name__, stack, np.asarray(runouts), n_iters, max_raises, batched, *(() if bets is None else (tuple(bets),)))
    t, s, o = make_streets_pair(L, game_cls, stack, runouts, "plus", 0, max_raises, tape=tape, bets=bets, avg_dtype="f32")
    live = tape is None or tape.live
    take = (lambda tag, fn: fn()) if tape is None else tape.take
    big = (lambda a: a) if (tape is None or (tape.live and not tape.recording)) else digest
    kind, rnd, first_col, n_ch = t.field("kind"), t.field("round"), t.field("first_col"), t.field("n_children")
    trunk = np.zeros(t.n_cols, bool)
    for n in np.where(kind == 0)[0]:
        if rnd[n] == 0:  # the betting before the first deal: the trunk (LEVELS kernels, float64 average as the reference)
            trunk[first_col[n]:first_col[n] + n_ch[n]] = True
    a32 = None
    hist = [take("reset/expl", lambda: np.array(o.exploitability, np.float32))]
    for it in range(n_iters):
        if live:
            o.cfr_iteration()
            strat = np.asarray(o.strategy)
            if it == 0:
                a32 = strat.astype(np.float32)
            else:
                cw, nw = sum(range(1, it + 1)), it + 1
                a32 = (cw / (cw + nw) * a32.astype(np.float64) + nw / (cw + nw) * strat).astype(np.float32)
        hist.append(take("it%d/expl" % it, lambda: np.array(o.exploitability, np.float32)))
        if not batched:
            want = take("it%d/regret" % it, lambda: big(np.asarray(o.regret)))
            if s is not None:
                s.iteration()
                same(s.get("regret"), want, "regret it%d" % it)
                assert np.array_equal(s.exploitability(), hist[-1]), it
    if batched:
        want = take("batched/regret", lambda: big(np.asarray(o.regret)))
        if s is not None:
            s.iterations(n_iters)
            same(s.get("regret"), want, "regret")
    want_trunk = take("avg/trunk", lambda: big(np.asarray(o.avg)[trunk]))
    want_rest = take("avg/streets", lambda: big(a32.astype(np.float64)[~trunk]))
    e64 = take("eval_avg", lambda: np.array(o.eval_avg(), np.float32))
    if tape is not None:
        tape.close()
    if s is None:
        return
    assert np.array_equal(s.get("expl_history"), np.stack(hist))
    avg = s.get("avg")
    same(avg[trunk], want_trunk, "trunk average")
    same(avg[~trunk], want_rest, "street columns: the float32 recurrence")
    e32 = s.eval_avg()
    assert np.allclose(e32, e64, rtol=1e-5, atol=0), (e32, e64)
    with pytest.raises(Exception):
        s.get_cols("avg", 0, 1)  # float32 storage: prl_solver_get translates, get_cols does not
    return s, o
