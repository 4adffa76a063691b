"""
GPU suite (`pytest -m gpu`, real MI355X): the hipcc-built libpokerrl_hip.so, called through its C ABI, against the CPU
oracle and the reference's golden vectors. Bit-exact for ranks / indices / every float32 array of the Leduc-family CFR
runs; float tolerance appears nowhere in this file except for the size-independent properties at bench scale.
"""
import numpy as np
import pytest

import parity_cases as pc
from helpers import golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from pokerrl_amd import _native
    _native.require_device()
    lib = _native.lib()
    assert lib.prl_build_flavor() == b"hip-gfx950"
    return lib


# ---- hand evaluator -----------------------------------------------------------------------------------------------------
def test_gpu_hand_rank_64_boards_vs_reference_binary(L):
    pc.check_hand_rank_golden(L)


def test_gpu_hand_rank_20000_boards_sha256(L):
    """SHA-256 of the 20000 x 1326 result for RandomState(0) boards starts 928d0aa3b1ae9432."""
    import hashlib

    from pokerrl_amd import _native
    g = golden("handrank.npz")
    rng = np.random.RandomState(0)
    boards = np.array([rng.choice(52, 5, replace=False) for _ in range(20000)], dtype=np.int8)
    ranks = _native.hand_rank_boards(boards)
    sha = hashlib.sha256(ranks.tobytes()).hexdigest()
    assert sha == str(g["sha256_rs0_20000"])
    assert sha.startswith("928d0aa3b1ae9432")
    # board-order invariance (SURVEY.md 2.2 pinned property)
    perm = boards[:, ::-1].copy()
    assert np.array_equal(_native.hand_rank_boards(perm), ranks)


def test_gpu_hand_rank_exhaustive_checksums(L):
    """All C(52,5) x 1326 evaluations against per-256-board checksums of the reference binary's output."""
    assert pc.check_hand_rank_checksums(L) == 10153


def test_gpu_legacy_batched_symbol_row_pointers(L):
    """Drop-in signature of lib_hand_eval.so: CppHandeval.py:45-65 passes row-pointer vectors."""
    import ctypes
    g = golden("handrank.npz")
    boards = np.ascontiguousarray(g["boards"][:7])
    out = np.full((7, 1326), -1, np.int32)

    def rows(a):
        return (a.__array_interface__["data"][0] + np.arange(a.shape[0]) * a.strides[0]).astype(np.intp)

    lut1 = np.zeros((1326, 2), np.int8)
    lut2 = np.zeros((52, 2), np.int8)
    L.get_hand_rank_all_hands_on_given_boards_52_holdem(
        rows(out).ctypes.data_as(ctypes.c_void_p), rows(boards).ctypes.data_as(ctypes.c_void_p), ctypes.c_int32(7),
        rows(lut1).ctypes.data_as(ctypes.c_void_p), rows(lut2).ctypes.data_as(ctypes.c_void_p))
    assert np.array_equal(out, g["ranks"][:7])


# ---- CFR on the Leduc family: oracle (all arrays, every iteration) and the reference's own logs ---------------------------
@pytest.mark.parametrize("variant", ["vanilla", "plus", "linear"])
def test_gpu_standard_leduc_vs_oracle(L, variant):
    pc.check_cfr_vs_oracle(L, "StandardLeduc", variant, 12)


def test_gpu_iterations_many(L):
    pc.check_iterations_many(L)


def test_gpu_cfrplus_delay(L):
    pc.check_cfr_vs_oracle(L, "StandardLeduc", "plus", 6, delay=3)


@pytest.mark.parametrize("gkey,variant", [("DiscretizedNLLeduc_POT", "plus"), ("DiscretizedNLLeduc_POT", "linear"),
                                          ("DiscretizedNLLeduc_B3_short", "vanilla")])
def test_gpu_nl_leduc_vs_oracle(L, gkey, variant):
    pc.check_cfr_vs_oracle(L, gkey, variant, 5)


def test_gpu_big_leduc_vs_oracle(L):
    pc.check_cfr_vs_oracle(L, "BigLeduc", "plus", 3)


@pytest.mark.parametrize("fixture,gkey,variant", [
    ("StandardLeduc_CFRPlus", "StandardLeduc", "plus"),
    ("StandardLeduc_VanillaCFR", "StandardLeduc", "vanilla"),
    ("StandardLeduc_LinearCFR", "StandardLeduc", "linear"),
    ("DiscretizedNLLeduc_POT_CFRPlus", "DiscretizedNLLeduc_POT", "plus"),
    ("DiscretizedNLLeduc_POT_LinearCFR", "DiscretizedNLLeduc_POT", "linear"),
    ("DiscretizedNLLeduc_B3_short_VanillaCFR", "DiscretizedNLLeduc_B3_short", "vanilla"),
    ("BigLeduc_CFRPlus", "BigLeduc", "plus"),
])
def test_gpu_reference_exploitability_series(L, fixture, gkey, variant):
    pc.check_cfr_vs_reference_series(L, fixture, gkey, variant)


def test_gpu_long_run_150_iterations_matches_oracle(L):
    """examples/run_cfrp_example.py runs 150 iterations; arrays compared every 50."""
    pc.check_cfr_vs_oracle(L, "StandardLeduc", "plus", 150, check_every=50)


@pytest.mark.parametrize("gkey", ["StandardLeduc", "DiscretizedNLLeduc_POT"])
def test_gpu_best_response_of_random_strategy(L, gkey):
    pc.check_br_of_given_strategy(L, gkey, 0, f64=True)
    pc.check_br_of_given_strategy(L, gkey, 1, f64=False)


# ---- Flop5Holdem (1326-hand ranges): oracle on small board sets, properties at bench scale --------------------------------
@pytest.mark.parametrize("variant", ["plus", "linear", "vanilla"])
def test_gpu_fhp_40_boards_vs_oracle(L, variant):
    pc.check_fhp_vs_oracle(L, 40, variant, 3)  # 40 boards > one canonical chance block of 32


# ---- fused board-block engine (the benchmark path) -------------------------------------------------------------------------
def test_gpu_fused_40_boards_vs_oracle(L):
    pc.check_fused_vs_oracle(L, 40, 4)


def test_gpu_fused_cfrplus_delay_vs_oracle(L):
    pc.check_fused_vs_oracle(L, 33, 4, delay=2)


@pytest.mark.parametrize("variant", ["vanilla", "linear"])
def test_gpu_fused_vanilla_linear_vs_oracle(L, variant):
    pc.check_fused_vs_oracle(L, 40, 4, variant=variant)


def test_gpu_fused_linear_batched_vs_oracle_and_levels(L):
    pc.check_fused_batched_vs_oracle(L, 35, 5, variant="linear")
    pc.check_fused_batched_vs_oracle(L, 35, 4, variant="vanilla")
    pc.check_fused_vs_levels(L, 2048, 5, variant="linear")


@pytest.mark.parametrize("n_boards,variant", [(2048, "plus"), (2048, "linear"), (2048, "vanilla"),
                                              (4096, "plus"), (4096, "linear"), (4096, "vanilla")])
def test_gpu_fused_many_boards_per_cu_vs_oracle(L, n_boards, variant):
    """The benchmark path against the ORACLE with more boards than CUs: every persistent workgroup walks 8 (16) boards, all but
    its first from the LDS prefetch area. 2048 boards = 2 canonical chance groups of 1024, 4096 = 4 (third summation level).
    Regrets, averages, implied strategy, current- and average-strategy exploitability, bit for bit, after every iteration."""
    pc.check_fused_vs_oracle(L, n_boards, 3, variant=variant)


@pytest.mark.parametrize("name", ["fhp_16384_plus", "fhp_16384_plus_d2_i6"])
def test_gpu_fused_16384_boards_fixture(L, name):
    """Oracle-generated fixtures (tests/golden/make_fhp_golden.py): exploitability history and SHA-256 of the regrets / averages on 16384 seeded
    boards (64 boards per CU) -- after 3 CFR+ iterations; after 6 with an averaging delay of 2 (the blend weights of four averaging steps,
    CFRPlus.py:65-87, and the scalar recurrence that stands for the averages of the hands a board blocks: sorted storage)."""
    import os
    from helpers import GOLDEN
    if not os.path.isfile(os.path.join(GOLDEN, name + ".npz")):
        pytest.skip("fixture not generated (tests/golden/make_fhp_golden.py)")
    pc.check_fused_vs_fixture(L, name)


@pytest.mark.parametrize("stack,flop_raises,nodes,n_boards,variant", [
    (700, None, 9, 40, "plus"), (700, None, 9, 1024, "linear"), (20000, 1, 9, 300, "vanilla"),
    (20000, 3, 21, 40, "plus"), (20000, 3, 21, 600, "linear")])
def test_gpu_fused_other_registered_shapes_vs_oracle(L, stack, flop_raises, nodes, n_boards, variant):
    """engine=auto picks the fused engine for every registered board-subtree shape (csrc/prl_fhp.h): FHP9 (short stacks / one
    raise per round) and FHP21 (three post-flop raises), each bit-exact against the oracle with several boards per CU"""
    pc.check_fused_vs_oracle(L, n_boards, 3, variant=variant, stack=stack, flop_raises=flop_raises, nodes_per_board=nodes)


# ---- trees that deal on several streets, all-in run-outs -------------------------------------------------------
def test_gpu_multistreet_limit_holdem_full_betting_vs_oracle(L):
    """LimitHoldem with its full betting structure (8 / 70 / 630 / 5670 decision nodes per street) over 2 flops x 2 turns x 1 river:
    ~52 k nodes, three chance levels; every per-node vector against the oracle"""
    from pokerrl_amd.game import games as G
    t, s, o = pc.check_multistreet_vs_oracle(L, G.LimitHoldem, 48, None, pc.multistreet_runouts(2, 2, 1), "plus", 2)
    assert t.n_nodes > 40000


@pytest.mark.parametrize("variant,batched", [("plus", False), ("plus", True), ("vanilla", True), ("linear", True)])
def test_gpu_streets_engine_limit_holdem_full_betting_vs_oracle(L, variant, batched):
    """the per-street fused engine (csrc/prl_st.h) on LimitHoldem with its full betting structure (27-node street subtrees, 7 / 14 / 252
    / 2268 street instances over 2 flops x 2 turns x 1 river, ~68 k nodes): regrets, averages, strategies, exploitability history and
    average-strategy exploitability against the oracle, bit for bit -- single iterations and the batched steady state, all variants"""
    from pokerrl_amd.game import games as G
    t, s, o = pc.check_streets_vs_oracle(L, G.LimitHoldem, 48, pc.multistreet_runouts(2, 2, 1), variant, 4 if batched else 2, batched=batched)
    assert t.n_nodes > 60000


@pytest.mark.parametrize("variant,max_raises", [("plus", (1, 1, 1, 1)), ("linear", (2, 2, 2, 2)), ("plus", (3, 3, 3, 3)), ("vanilla", (1, 2, 3, 4))])
def test_gpu_streets_engine_other_street_shapes_vs_oracle(L, variant, max_raises):
    """the per-street engine's other registered street subtrees (9 / 15 / 21 nodes: one, two, three raises per round) and a game whose streets
    differ in shape (1, 2, 3, 4 raises: 9-, 15-, 21- and 27-node streets in one tree), 2 flops x 2 turns x 2 rivers, batched iterations"""
    from pokerrl_amd.game import games as G
    pc.check_streets_vs_oracle(L, G.LimitHoldem, 48, pc.multistreet_runouts(2, 2, 2), variant, 3, max_raises=max_raises, batched=True)


@pytest.mark.parametrize("variant,runouts", [("plus", (1, 34, 1)), ("linear", (1, 5, 2)), ("plus", (2, 33, 2))])
def test_gpu_streets_engine_many_outcomes_per_deal_vs_oracle(L, variant, runouts):
    """more outcomes below a chance node than one canonical block of 32 (34 turn cards: two blocks) and than one batch of child rows of the leaf
    sums (5 > 4): the non-final streets' chance sums against the oracle, 9-node street subtrees, batched iterations"""
    from pokerrl_amd.game import games as G
    pc.check_streets_vs_oracle(L, G.LimitHoldem, 48, pc.multistreet_runouts(*runouts), variant, 3, max_raises=(1, 1, 1, 1), batched=True)


@pytest.mark.parametrize("batched", [False, True])
def test_gpu_streets_engine_cfr_plus_with_averaging_delay_vs_oracle(L, batched):
    """CFR+ with a linear-averaging delay of 2 (CFRPlus.py:65-87: no average before iteration 2, a copy at 2, blends after) on the per-street engine,
    LimitHoldem with two raises per round, 2 flops x 2 turns x 2 rivers"""
    from pokerrl_amd.game import games as G
    pc.check_streets_vs_oracle(L, G.LimitHoldem, 48, pc.multistreet_runouts(2, 2, 2), "plus", 5, delay=2, max_raises=(2, 2, 2, 2), batched=batched)


@pytest.mark.parametrize("batched,max_raises", [(False, (1, 2, 1, 1)), (True, None)])
def test_gpu_streets_engine_float32_running_average_opt_in(L, batched, max_raises):
    """PRL_SOLVER_AVG_F32 on the per-street engine (full betting = the 27-node shape when max_raises is None)"""
    from pokerrl_amd.game import games as G
    pc.check_streets_avg_f32(L, G.LimitHoldem, 48, pc.multistreet_runouts(2, 2, 2), 4, max_raises=max_raises, batched=batched)


def test_gpu_streets_engine_best_response_of_an_explicit_strategy_vs_oracle(L):
    """LocalBRMaster's evaluation (LocalBRMaster.py:67-80) on LimitHoldem with its full betting, 2 flops x 2 turns x 1 river: explicit float32 /
    float64 strategies on the per-street engine against the oracle; iterating again after reset()"""
    from pokerrl_amd.game import games as G
    t = pc.check_streets_br_vs_oracle(L, G.LimitHoldem, 48, pc.multistreet_runouts(2, 2, 1))
    assert t.n_nodes > 60000


@pytest.mark.parametrize("variant,stack,runouts,batched", [("plus", 1200, (2, 2, 1), False), ("vanilla", 2500, (2, 2, 2), True), ("linear", 600, (2, 3, 2), True),
                                                           ("plus", 1200, (1, 34, 2), True), ("linear", 20000, (2, 2, 2), True)])
def test_gpu_streets_engine_discretized_nl_holdem_vs_oracle(L, variant, stack, runouts, batched):
    """MIXED STREETS (csrc/prl_st.h): DiscretizedNLHoldem (games.py:114-131) with pot-sized raises -- the subtrees of one street differ with the stacks
    behind (9-, 15-, 21-node shapes side by side: 6 to 9 (street, shape) groups) and every raise sequence that runs out of chips ends in an all-in call
    whose hand is dealt out as a chain of chance nodes (the decision-free run-out forest); 34 turn cards = two canonical blocks; 20000 chips = the game's
    200-big-blind default, where five raises fit on a street (the 33-node shape)"""
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    pc.check_streets_vs_oracle(L, G.DiscretizedNLHoldem, stack, pc.multistreet_runouts(*runouts), variant, 4 if batched else 2, batched=batched, bets=bet_sets.POT_ONLY)


@pytest.mark.parametrize("variant,stack,runouts,delay", [("plus", 4, (2, 2, 1), 0), ("linear", 10, (2, 2, 2), 0), ("plus", 20, (2, 1, 2), 2)])
def test_gpu_streets_engine_all_in_run_outs_vs_oracle(L, variant, stack, runouts, delay):
    """LimitHoldem with its full betting and stacks that run out (4, 10, 20 chips): capped raise sequences (mixed street shapes) and all-in run-out
    chains on every street, on the per-street engine; CFR+ with an averaging delay"""
    from pokerrl_amd.game import games as G
    pc.check_streets_vs_oracle(L, G.LimitHoldem, stack, pc.multistreet_runouts(*runouts), variant, 4, delay=delay, batched=True)


def test_gpu_streets_engine_mixed_best_response_and_f32_average_vs_oracle(L):
    """explicit strategies (LocalBRMaster.py:67-80) and PRL_SOLVER_AVG_F32 on a tree with mixed street shapes and run-out chains"""
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    pc.check_streets_br_vs_oracle(L, G.DiscretizedNLHoldem, 1200, pc.multistreet_runouts(2, 2, 1), bets=bet_sets.POT_ONLY)
    pc.check_streets_avg_f32(L, G.DiscretizedNLHoldem, 1200, pc.multistreet_runouts(2, 2, 2), 3, batched=True, bets=bet_sets.POT_ONLY)


@pytest.mark.parametrize("variant", ["plus", "linear", "vanilla", "plus_d2_i6"])
def test_gpu_streets_engine_bench_tree_vs_oracle_fixture(L, variant):
    """bench_multistreet.py's tree (4 flops x 2 turns x 2 rivers, 259 330 nodes) on the per-street engine against the ORACLE's own run of it
    (tests/golden/make_streets_golden.py): exploitability history, average-strategy exploitability, SHA-256 of the regrets / averages"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_multistreet
    from helpers import GOLDEN, h32
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    path = os.path.join(GOLDEN, "lh_4x2x2_%s.npz" % variant)
    if not os.path.isfile(path):
        pytest.skip("fixture not generated (tests/golden/make_streets_golden.py)")
    g = np.load(path)
    ro = bench_multistreet.runouts(int(g["flops"]), int(g["turns"]), int(g["rivers"]))
    assert h32(ro) == str(g["runouts_sha256"])
    t = _native.NativeTree.for_game(G.LimitHoldem, 48, None, ro, _lib=L)
    assert t.n_nodes == int(g["n_nodes"])
    s = _native.NativeSolver(t, str(g["variant"]), int(g["delay"]) if "delay" in g else 0, engine="auto", _lib=L)
    assert s.engine == "fused"
    s.iterations(int(g["n_iters"]))
    assert np.array_equal(s.get("expl_history"), g["expl_history"]), (s.get("expl_history"), g["expl_history"])
    assert np.array_equal(s.eval_avg(), g["eval_avg"])
    assert h32(s.get("regret")) == str(g["regret_sha256"])
    assert h32(s.get("avg")) == str(g["avg_sha256"])


@pytest.mark.parametrize("name", ["nl2500_16x8x8_plus", "nl20000_8x4x4_linear"])
def test_gpu_streets_engine_nl_bench_trees_vs_oracle_fixture(L, name):
    """bench_multistreet.py --game DiscretizedNLHoldem at its two benched sizes (16 x 8 x 8 run-outs at 50 big blinds: 340 742 nodes; 8 x 4 x 4 at the 200-big-
    blind default: 257 754 nodes, five-raise streets) on the per-street engine -- mixed street shapes, ~39 k showdowns in run-out chains -- against the ORACLE's
    own run (tests/golden/make_streets_golden.py nl): exploitability history, average-strategy exploitability, SHA-256 of the regrets / averages"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_multistreet
    from helpers import GOLDEN, h32
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    path = os.path.join(GOLDEN, name + ".npz")
    if not os.path.isfile(path):
        pytest.skip("fixture not generated (tests/golden/make_streets_golden.py nl)")
    g = np.load(path)
    ro = bench_multistreet.runouts(int(g["flops"]), int(g["turns"]), int(g["rivers"]))
    assert h32(ro) == str(g["runouts_sha256"])
    t = _native.NativeTree.for_game(G.DiscretizedNLHoldem, int(g["stack"]), bet_sets.POT_ONLY, ro, _lib=L)
    assert t.n_nodes == int(g["n_nodes"])
    s = _native.NativeSolver(t, str(g["variant"]), 0, engine="auto", _lib=L)
    assert s.engine == "fused"
    s.iterations(int(g["n_iters"]))
    assert np.array_equal(s.get("expl_history"), g["expl_history"]), (s.get("expl_history"), g["expl_history"])
    assert np.array_equal(s.eval_avg(), g["eval_avg"])
    assert h32(s.get("regret")) == str(g["regret_sha256"])
    assert h32(s.get("avg")) == str(g["avg_sha256"])


def test_gpu_streets_engine_bench_tree_best_response_vs_oracle_fixture(L):
    """exact best response of a seeded float32 strategy on bench_multistreet.py's tree: the per-street engine's evaluation pass against the
    oracle's exploitability of the same strategy (tests/golden/make_streets_golden.py br)"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import bench_multistreet
    from helpers import GOLDEN, h32
    from make_streets_golden import seeded_strategy
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    path = os.path.join(GOLDEN, "lh_4x2x2_br.npz")
    if not os.path.isfile(path):
        pytest.skip("fixture not generated (tests/golden/make_streets_golden.py br)")
    g = np.load(path)
    ro = bench_multistreet.runouts(int(g["flops"]), int(g["turns"]), int(g["rivers"]))
    assert h32(ro) == str(g["runouts_sha256"])
    t = _native.NativeTree.for_game(G.LimitHoldem, 48, None, ro, _lib=L)
    strat = seeded_strategy(t, int(g["seed"]))
    assert h32(strat) == str(g["strategy_sha256"])
    s = _native.NativeSolver(t, "plus", 0, engine="auto", _lib=L)
    assert s.engine == "fused"
    s.set_strategy(strat)
    s.compute_ev()
    assert np.array_equal(s.exploitability(), g["exploitability"]), (s.exploitability(), g["exploitability"])


def test_gpu_streets_engine_vs_levels_engine_bench_tree(L):
    """bench_multistreet.py's tree (4 flops x 2 turns x 2 rivers, 259 330 nodes) on both engines of the library: the same exploitability
    history, regrets and averages; the per-street engine in < 1/3 of the level-synchronous engine's HBM"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench_multistreet
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    t = _native.NativeTree.for_game(G.LimitHoldem, 48, None, bench_multistreet.runouts(4, 2, 2), _lib=L)
    a = _native.NativeSolver(t, "plus", 0, engine="auto", _lib=L)
    assert a.engine == "fused"
    a.iterations(3)
    hist, regret, avg, ev_avg, mem_a = a.get("expl_history"), a.get("regret"), a.get("avg"), a.eval_avg(), int(a.get("bytes_allocated")[0])
    del a
    b = _native.NativeSolver(t, "plus", 0, engine="levels", _lib=L)
    b.iterations(3)
    assert np.array_equal(hist, b.get("expl_history"))
    assert np.array_equal(regret, b.get("regret")) and np.array_equal(avg, b.get("avg"))
    assert np.array_equal(ev_avg, b.eval_avg())
    assert mem_a * 3 < int(b.get("bytes_allocated")[0])


@pytest.mark.parametrize("variant", ["vanilla", "linear"])
def test_gpu_multistreet_short_stack_run_outs_vs_oracle(L, variant):
    from pokerrl_amd.game import games as G
    pc.check_multistreet_vs_oracle(L, G.LimitHoldem, 6, None, pc.multistreet_runouts(3, 2, 2), variant, 3, expect_runout_chain=True)


def test_gpu_all_in_before_the_deal_run_out_vs_oracle(L):
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    pc.check_multistreet_vs_oracle(L, G.Flop5Holdem, 250, bet_sets.POT_ONLY, pc.fhp_boards(300), "plus", 4)


def test_gpu_cfr_plus_on_limit_holdem_builder_dealt_run_outs_on_the_street_engine(L):
    """CFRPlus(game_cls=LimitHoldem, max_outcomes=(2, 2, 1)): the builder deals the run-outs from the deck itself (board_enum.py), engine=auto
    takes the per-street fused engine; the logged exploitability series equals the level-synchronous engine's on the same tree"""
    from pokerrl_amd.cfr.CFRPlus import CFRPlus
    from pokerrl_amd.game import games as G
    from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase
    series = []
    for engine in ("auto", "levels"):
        chief = ChiefBase(t_prof=None)
        cfr = CFRPlus(name="lh_" + engine, chief_handle=chief, game_cls=G.LimitHoldem, agent_bet_set=None, delay=0, max_outcomes=(2, 2, 1), engine=engine)
        assert cfr._trees[0].solver.engine == ("fused" if engine == "auto" else "levels")
        cfr.iterations(4)
        vals, _ = chief.get_new_values()
        series.append([v for k, v in vals.items() if "_Curr_S" in k][0]["Evaluation/" + G.LimitHoldem.WIN_METRIC])
    assert series[0] == series[1] and len(series[0]) == 5 and series[0][-1][1] < series[0][0][1]


def test_gpu_cfr_plus_on_discretized_nl_holdem_on_the_street_engine(L):
    """CFRPlus(game_cls=DiscretizedNLHoldem, agent_bet_set=POT_ONLY, starting_stack_sizes=[600, 2500], max_outcomes=(2, 2, 2)) through the reference's
    class (CFRBase.py:13-75: one public tree per stack size): mixed street shapes and all-in run-out chains -- engine=auto takes the per-street fused
    engine for both trees; the logged exploitability series of each stack equals the level-synchronous engine's on the same tree"""
    from pokerrl_amd.cfr.CFRPlus import CFRPlus
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase
    series = []
    for engine in ("auto", "levels"):
        chief = ChiefBase(t_prof=None)
        cfr = CFRPlus(name="nl_" + engine, chief_handle=chief, game_cls=G.DiscretizedNLHoldem, agent_bet_set=bet_sets.POT_ONLY, starting_stack_sizes=[600, 2500],
                      delay=0, max_outcomes=(2, 2, 2), engine=engine)
        assert [t.solver.engine for t in cfr._trees] == ["fused" if engine == "auto" else "levels"] * 2
        cfr.iterations(4)
        vals, _ = chief.get_new_values()
        series.append([[v for k, v in vals.items() if "_Curr_S%d" % st in k][0]["Evaluation/" + G.DiscretizedNLHoldem.WIN_METRIC] for st in (600, 2500)])
    assert series[0] == series[1]
    for one in series[0]:
        assert len(one) == 5 and one[-1][1] < one[0][1]


def test_gpu_bench_total_boards_one_gpu_smoke():
    """bench.py --total-boards T on one GPU: ONE board list (strong-scaling mode; the shape of --all-boards, which needs 4-8 GPUs), the
    builder's own seeded enumeration; the JSON says so"""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--total-boards", "3000", "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                          "--no-placement-probe"], capture_output=True, text=True, timeout=600, check=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    assert d["scaling"] == "strong" and d["config"]["nodes_whole_tree"] == 5 + 3000 * 15 and d["config"]["boards_per_gpu"] == 3000
    assert d["build_flavor"] == "hip-gfx950" and d["config"]["exploitability_mbb_per_g"] > 0


def test_gpu_cfr_plus_on_limit_holdem_through_the_python_surface(L):
    """CFRPlus(game_cls=LimitHoldem, boards=run-outs): the reference's class, a game its tree code cannot build; exploitability falls"""
    from pokerrl_amd.cfr.CFRPlus import CFRPlus
    from pokerrl_amd.game import games as G
    from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase
    chief = ChiefBase(t_prof=None)
    cfr = CFRPlus(name="lh", chief_handle=chief, game_cls=G.LimitHoldem, agent_bet_set=None, delay=0, boards=pc.multistreet_runouts(2, 1, 1),
                  starting_stack_sizes=[8])
    cfr.iterations(8)
    vals, _ = chief.get_new_values()
    curr = [v for k, v in vals.items() if "_Curr_S" in k][0]["Evaluation/" + G.LimitHoldem.WIN_METRIC]
    assert len(curr) == 9 and curr[-1][1] < curr[0][1] and curr[-1][1] >= 0
    root = cfr._trees[0].root
    flop_chance = [n for n in cfr._trees[0].nodes() if n.p_id_acting_next == "Ch"][0]
    cs = flop_chance.strategy
    assert cs.shape == (1326, 2) and set(np.unique(cs).tolist()) <= {0.0, float(cs.max())}
    st = flop_chance.children[0].env_state
    assert st["current_round"] == 1 and int(np.sum(st["board_2d"][:, 0] >= 0)) == 3
    assert root.exploitability.shape == (2,)


@pytest.mark.parametrize("n_boards", [40, 2048])
def test_gpu_fused_best_response_only_pass_vs_oracle(L, n_boards):
    """BASELINE config 4: exact best response of an explicit strategy; float32 strategies go through the best-response-only pass"""
    pc.check_fused_br_vs_oracle(L, n_boards)


@pytest.mark.parametrize("fused,variant", [(False, "vanilla"), (True, "plus"), (True, "linear")])
def test_gpu_checkpoint_resume(L, fused, variant):
    pc.check_checkpoint_resume(L, fused, variant, n_before=4, n_after=3)


def test_gpu_fused_batched_iterations_vs_oracle(L):
    pc.check_fused_batched_vs_oracle(L, 40, 5, delay=1)


@pytest.mark.gpu
def test_gpu_fused_per_board_rows_vs_oracle(L, monkeypatch):
    """PRL_FHP_NO_BLOCK_SUM: the pass writes one row per board and the sum kernels do level 0 (the path of shards that are not whole blocks)"""
    monkeypatch.setenv("PRL_FHP_NO_BLOCK_SUM", "1")
    pc.check_fused_batched_vs_oracle(L, 70, 3, delay=0)


@pytest.mark.gpu
@pytest.mark.parametrize("no_steady", [False, True])
def test_gpu_fused_steady_state_specialisation_vs_oracle(L, monkeypatch, no_steady):
    """the CFR+ steady-state instantiation of the update passes and the generic one (PRL_FHP_NO_STEADY) against the oracle;
    block sums forced on (at 300 boards the default keeps per-board rows, which test_gpu_fused_batched_iterations_vs_oracle covers)"""
    monkeypatch.setenv("PRL_FHP_BLOCK_SUM", "1")
    if no_steady:
        monkeypatch.setenv("PRL_FHP_NO_STEADY", "1")
    pc.check_fused_batched_vs_oracle(L, 300, 5, delay=0)


def test_gpu_fused_float32_running_average_opt_in(L):
    """PRL_SOLVER_AVG_F32 (opt-in, bench.py --avg-f32): regrets / current exploitability bit-exact to the oracle, the average = the float32
    recurrence, its exploitability within 1e-5 of the float64 one; 600 boards, 6 iterations (several boards per CU, the steady-state kernels)"""
    pc.check_fused_avg_f32(L, 600, 6)


@pytest.mark.gpu
@pytest.mark.parametrize("variant,symmetrize,n_classes", [("plus", True, 40), ("linear", True, 6), ("vanilla", True, 6), ("plus", False, 6)])
def test_gpu_fused_weighted_boards_suit_classes_vs_oracle(L, variant, symmetrize, n_classes):
    """prl_solver_create_weighted (suit isomorphism): class representatives with multiplicities, chance values averaged over the hands' suit
    orbits -- bit for bit against the oracle's restatement (40 classes: two 32-board blocks of the canonical sum)"""
    pc.check_weighted_vs_oracle(L, n_classes, 3, variant, symmetrize)


@pytest.mark.gpu
def test_gpu_fused_weighted_boards_checkpoint_resume(L):
    pc.check_weighted_checkpoint(L)


@pytest.mark.gpu
def test_gpu_fused_suit_isomorphism_equals_the_full_board_list(L):
    """the class solve against the fused solve of the FULL suit-closed board list the classes stand for (every board listed, weight 1): the
    isomorphism is exact in exact arithmetic; float32 runs agree to 2e-5 over 5 CFR+ iterations (6 classes = 100-odd boards)"""
    class S:
        def __init__(self, boards, mult):
            from pokerrl_amd import _native
            self.s = _native.NativeSolver(pc.fhp_tree_of(L, boards), "plus", 0, engine="fused", _lib=L, board_mult=mult, symmetrize="subset" if mult is not None else False)

        def iteration(self):
            self.s.iteration()

        def exploitability(self):
            return self.s.exploitability()

        def eval_avg(self):
            return self.s.eval_avg()

    pc.iso_vs_full(S, S, 6, 5)


@pytest.mark.gpu
def test_gpu_fused_vs_levels_2048_boards(L):
    """2048 boards = 64 canonical chance blocks = 2 groups: both engines of the library must agree bit for bit."""
    pc.check_fused_vs_levels(L, 2048, 6)


def test_gpu_fhp_properties_at_scale(L):
    """1024 boards: zero-sum, BR >= EV, exploitability >= 0 and decreasing on average, strategies row-stochastic."""
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    from helpers import native_tree
    boards = pc.fhp_boards(1024, seed=11)
    t = native_tree(G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards)
    s = _native.NativeSolver(t, "plus", 0, engine="levels")
    e0 = s.exploitability()
    s.iterations(10)
    e10 = s.exploitability()
    assert np.all(e0 > 0) and np.all(e10 >= 0)
    assert e10.mean() < e0.mean()
    ev, ev_br, reach = s.get("ev"), s.get("ev_br"), s.get("reach")
    zs = np.sum(ev.astype(np.float64) * reach.astype(np.float64), axis=(1, 2))  # ValueFiller.py:98 per node
    assert np.max(np.abs(zs)) < 1e-3
    assert np.all(ev_br[0] >= ev[0] - 1e-3)
    strat = s.get("strategy")
    fc, nc, kind = t.field("first_col"), t.field("n_children"), t.field("kind")
    for n in np.where(kind == 0)[0][:200]:
        assert np.allclose(strat[fc[n]:fc[n] + nc[n]].sum(axis=0), 1, atol=1e-5)


_BIG_TREES = {}


def _big_tree(key, make_boards):
    """the bench-size / whole-game trees are built once per test session (three variants each walk the same one)"""
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    from helpers import native_tree
    if key not in _BIG_TREES:
        boards = make_boards()
        _BIG_TREES[key] = (boards, native_tree(G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards[0] if isinstance(boards, tuple) else boards))
    return _BIG_TREES[key]


def _check_state_hashes(s, g, variant):
    """SHA-256 of every regret / average (/ average-sum) column against the fixture; the arrays are hashed side by side while they stream off the GPU"""
    names = ("regret",) + (() if variant == "plus" else ("avg_sum",)) + ("avg",)  # (avg after avg_sum: the last average update rode on the closing evaluation)
    got = s.sha256_of_many(names)
    for n in names:
        assert got[n] == str(g[n + "_sha256"]), n


@pytest.mark.parametrize("variant", ["plus", "linear", "vanilla"])
def test_gpu_fused_bench_size_vs_oracle_fixture(L, variant):
    """bench.py's workload at full size (262144 boards, the board list of rank 0) against the ORACLE: two CFR+ (Linear CFR: BASELINE config 3)
    iterations, exploitability history and SHA-256 of all 3.67 M regret / average (/ average-sum) columns. The oracle cannot hold that tree in
    one piece; the fixture comes from its chunked run (tests/golden/make_fhp_golden_chunked.py: 16 chunks of 16384 boards, the trunk's chance
    node fed the canonical sum of all chunks; checked against the one-piece oracle at 2048 boards, every variant). The arrays are streamed
    through prl_solver_get_cols."""
    import os
    import bench
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    from helpers import GOLDEN, h32, native_tree
    path = os.path.join(GOLDEN, "fhp_262144_%s_chunked.npz" % variant)
    if not os.path.isfile(path):
        pytest.skip("fixture not generated (an hour of oracle time: tests/golden/make_fhp_golden_chunked.py)")
    g = np.load(path)
    assert str(g["variant"]) == variant
    boards, t = _big_tree(("bench", int(g["n_boards"]), int(g["seed"])), lambda: bench.seeded_boards(int(g["n_boards"]), int(g["seed"])))
    assert h32(boards) == str(g["boards_sha256"])
    s = _native.NativeSolver(t, variant, 0, engine="fused")
    s.iterations(int(g["n_iters"]))
    assert np.array_equal(s.get("expl_history"), g["expl_history"]), (s.get("expl_history"), g["expl_history"])
    _check_state_hashes(s, g, variant)
    assert bet_sets is not None and G is not None and native_tree is not None


@pytest.mark.parametrize("variant", ["plus", "linear", "vanilla"])
def test_gpu_whole_game_vs_oracle_fixture(L, variant):
    """The WHOLE Flop5Holdem game on the GPU -- all 2 598 960 boards through their 134 459 suit classes (prl_solver_create_weighted: multiplicities in
    the chance weights, orbit-mean chance values) -- against the ORACLE: four iterations of CFR+ and of Linear CFR (BASELINE config 3: "LinearCFR, full
    public tree, 1 MI355X"), exploitability history and SHA-256 of all 1.88 M regret / average (/ average-sum) columns. The fixtures come from the
    oracle's chunked run (make_fhp_golden_chunked.py --whole-game: 8 chunks of 16384 classes + one of 3387, the trunk's chance node fed the canonical
    weighted sum and symmetrised; equal to the one-piece oracle on 2363 classes in ragged chunks, `--selftest <dir> weighted [variant]`).
    Round 6: all three variants, and the AVERAGE strategy's exploitability at full size (the two-seat evaluation pass over the float64 averages with the
    ragged last chunk and the orbit means: `eval_avg` of the chunked oracle)."""
    import os
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets, board_enum
    from pokerrl_amd.game import games as G
    from helpers import GOLDEN, h32, native_tree
    path = os.path.join(GOLDEN, "fhp_whole_game_%s_chunked.npz" % variant)
    if not os.path.isfile(path):
        pytest.skip("fixture not generated (an hour of oracle time: tests/golden/make_fhp_golden_chunked.py --whole-game)")
    g = np.load(path)
    assert str(g["variant"]) == variant
    (reps, mult), t = _big_tree("whole", lambda: board_enum.single_deal_board_classes(G.Flop5Holdem))
    assert len(reps) == int(g["n_classes"]) and int(mult.sum()) == int(g["n_boards"]) == 2598960
    assert h32(reps) == str(g["boards_sha256"]) and h32(mult.astype(np.int32)) == str(g["mult_sha256"])
    s = _native.NativeSolver(t, variant, 0, engine="fused", board_mult=mult, symmetrize=True)
    s.iterations(int(g["n_iters"]))
    assert np.array_equal(s.get("expl_history"), g["expl_history"]), (s.get("expl_history"), g["expl_history"])
    _check_state_hashes(s, g, variant)
    if "eval_avg" in g.files:  # (fixtures regenerated in round 6 carry it)
        assert np.array_equal(s.eval_avg(), g["eval_avg"]), (s.eval_avg(), g["eval_avg"])
    assert bet_sets is not None and native_tree is not None


def test_gpu_fused_br_bench_size_vs_oracle_fixture(L):
    """bench_br.py's workload at full size (65536 boards, rank 0's board list and seeded float32 strategy) against the ORACLE's chunked
    evaluation (tests/golden/make_fhp_br_golden_chunked.py; equals the one-piece oracle at 2048 boards): both seats' exploitability."""
    import os
    import bench
    import bench_br
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    from helpers import GOLDEN, h32, native_tree
    path = os.path.join(GOLDEN, "fhp_br_65536_chunked.npz")
    if not os.path.isfile(path):
        pytest.skip("fixture not generated (tests/golden/make_fhp_br_golden_chunked.py)")
    g = np.load(path)
    n_boards = int(g["n_boards"])
    boards = bench.seeded_boards(n_boards, int(g["board_seed"]))
    assert h32(boards) == str(g["boards_sha256"])
    t = native_tree(G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards)
    strat = bench_br.seeded_strategy(t.n_cols - 14 * n_boards, n_boards, t.range_size, int(g["strategy_seed"]))
    assert h32(strat) == str(g["strategy_sha256"])
    s = _native.NativeSolver(t, "plus", 0, engine="fused")
    s.set_strategy(strat)
    s.compute_ev()
    assert np.array_equal(s.exploitability(), g["exploitability"]), (s.exploitability(), g["exploitability"])
    s.time_evaluations(2)  # the timed entry point of bench_br.py runs the same pass
    assert np.array_equal(s.exploitability(), g["exploitability"])


def test_gpu_fused_properties_at_bench_size(L):
    """bench.py's workload at full size (262144 boards, 71 GB) on the fused engine: the size-independent properties the engine
    exposes at that size -- exploitability of the current and of the average strategy positive and falling, the closing
    evaluation equal to the last history entry -- and run-to-run determinism of every one of those numbers (bits), which a race
    anywhere in the 262144-board passes would break."""
    import bench
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    from helpers import native_tree
    boards = bench.seeded_boards(262144, 0)
    runs = []
    for _ in range(2):
        t = native_tree(G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards)
        s = _native.NativeSolver(t, "plus", 0, engine="fused")
        s.iterations(3)
        avg3 = s.eval_avg().copy()
        s.iterations(3)
        hist = s.get("expl_history").copy()
        avg6 = s.eval_avg().copy()
        cur = s.exploitability().copy()
        runs.append((hist, avg3, avg6, cur))
        assert hist.shape == (7, 2) and np.all(hist > 0) and hist[6].mean() < hist[0].mean()
        assert np.array_equal(cur, hist[6])
        assert np.all(avg6 > 0) and avg6.mean() < avg3.mean() < hist[0].mean()
        del s, t
    for a, b in zip(runs[0], runs[1]):
        assert np.array_equal(a, b)


def test_gpu_placement_selection_inside_the_library():
    """prl_solver_create_placed (NativeSolver(place=k)): k solvers built side by side, each timed, the fastest kept and reset -- the state handed
    back is that of a plain create: same exploitability history and regrets after the same iterations"""
    from helpers import env_args
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    boards = pc.fhp_boards(512)
    t = _native.NativeTree(G.Flop5Holdem.native_game(env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)), G.Flop5Holdem.native_rules(), boards)
    a = _native.NativeSolver(t, "plus", 0, engine="fused", place=3, probe_iters=2)
    assert a.engine == "fused" and len(a.placement_ms) == 3 and all(x > 0 for x in a.placement_ms) and 0 <= a.placement_chosen < 3
    assert a.iter == 0
    b = _native.NativeSolver(t, "plus", 0, engine="fused")
    a.iterations(3)
    b.iterations(3)
    assert np.array_equal(a.get("expl_history"), b.get("expl_history")) and np.array_equal(a.get("regret"), b.get("regret"))
    # engines with nothing to choose: one solver, no timings
    c = _native.NativeSolver(t, "plus", 0, engine="levels", place=2)
    assert c.engine == "levels" and c.placement_ms == [0.0, 0.0]
