"""
CPU suite: the HIP kernel SOURCES (pokerrl_amd/csrc/*.hip) compiled with -DPRL_EMU and executed by the fiber-based SIMT
emulator of tests/emu/ -- compared bit for bit with the CPU oracle and the reference's golden vectors. This catches
indexing / barrier / scan-order mistakes in the GPU-less container; the same cases run against the real hipcc build on an
MI355X in test_gpu_parity.py. (The emulator is test infrastructure; the package never loads it.)
"""
import os
import sys

import pytest

import parity_cases as pc

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "emu"))


@pytest.fixture(scope="module")
def L():
    import build_emu
    from pokerrl_amd import _native
    lib = _native.bind(build_emu.build())
    assert lib.prl_build_flavor().startswith(b"emu")
    return lib


@pytest.mark.parametrize("variant", ["vanilla", "plus", "linear"])
def test_emu_standard_leduc_vs_oracle(L, variant):
    pc.check_cfr_vs_oracle(L, "StandardLeduc", variant, 4)


def test_emu_iterations_many(L):
    pc.check_iterations_many(L)


def test_emu_cfrplus_delay(L):
    pc.check_cfr_vs_oracle(L, "StandardLeduc", "plus", 4, delay=2)


def test_emu_nl_leduc_all_in_preflop_terminals(L):
    pc.check_cfr_vs_oracle(L, "DiscretizedNLLeduc_POT", "linear", 2)


def test_emu_reference_series_standard_leduc(L):
    pc.check_cfr_vs_reference_series(L, "StandardLeduc_CFRPlus", "StandardLeduc", "plus")


@pytest.mark.slow
def test_emu_big_leduc_pairwise_sums(L):
    pc.check_cfr_vs_oracle(L, "BigLeduc", "plus", 1)


def test_emu_fhp_three_boards(L):
    pc.check_fhp_vs_oracle(L, 3, "plus", 1)


def test_emu_fused_engine_three_boards(L):
    pc.check_fused_vs_oracle(L, 3, 2)


def test_emu_fused_engine_several_boards_per_workgroup(L, monkeypatch):
    """PRL_FHP_GRID=2: two persistent workgroups walk 7 boards, so each starts its 2nd..4th board from the LDS prefetch area
    that was filled while the previous board was being walked (on the GPU that is every board but the first of a CU)."""
    monkeypatch.setenv("PRL_FHP_GRID", "2")
    pc.check_fused_vs_oracle(L, 7, 2)


def test_emu_fused_engine_best_response_of_explicit_strategy(L):
    pc.check_fused_br_vs_oracle(L, 3)


@pytest.mark.parametrize("stack,flop_raises", [(700, None), (20000, 1)])
def test_emu_fused_engine_nine_node_board_subtree(L, stack, flop_raises):
    """the second registered shape (csrc/prl_fhp.h: FHP9): short stacks make the first post-flop bet all-in, or one raise per round"""
    pc.check_fused_vs_oracle(L, 3, 2, stack=stack, flop_raises=flop_raises, nodes_per_board=9)
    pc.check_fused_vs_oracle(L, 3, 2, variant="linear", stack=stack, flop_raises=flop_raises, nodes_per_board=9)


def test_emu_fused_engine_twentyone_node_board_subtree(L):
    """the third registered shape (FHP21: three post-flop raises, 20 action columns): its regret block does not fit the LDS
    prefetch area, so this instantiation reads the regrets from HBM at the board start"""
    pc.check_fused_vs_oracle(L, 3, 2, flop_raises=3, nodes_per_board=21)


def test_emu_unregistered_board_subtree_falls_back_to_levels(L):
    """four and five post-flop raises = the 27- and 33-node shapes, which only the per-street engine instantiates: a single-deal tree is one street to it;
    six raises (with stacks that allow them) match no registered shape: level-synchronous engine, and asking for the fused one is an error"""
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    pc.check_fused_vs_oracle(L, 3, 2, flop_raises=4, nodes_per_board=27)
    pc.check_fused_vs_oracle(L, 2, 1, stack=200000, flop_raises=5, nodes_per_board=33)
    t = _native.NativeTree(pc.fhp_game(2000000, flop_raises=6), G.Flop5Holdem.native_rules(), pc.fhp_boards(3), _lib=L)  # 39 nodes per board
    assert t.n_nodes == 5 + 39 * 3
    assert _native.NativeSolver(t, "plus", 0, engine="auto", _lib=L).engine == "levels"
    with pytest.raises(_native.NativeError):
        _native.NativeSolver(t, "plus", 0, engine="fused", _lib=L)


def test_emu_fused_engine_cfrplus_delay(L):
    pc.check_fused_vs_oracle(L, 3, 3, delay=1)


@pytest.mark.parametrize("variant", ["vanilla", "linear"])
def test_emu_fused_engine_vanilla_linear(L, variant):
    pc.check_fused_vs_oracle(L, 3, 3, variant=variant)


@pytest.mark.parametrize("variant,no_steady", [("linear", False), ("linear", True), ("vanilla", False)])
def test_emu_fused_engine_linear_batched_iterations(L, monkeypatch, variant, no_steady):
    """Linear / vanilla CFR, batched: iterations 2.. run their steady-state instantiation of the update passes (PRL_FHP_NO_STEADY: the generic one)"""
    if no_steady:
        monkeypatch.setenv("PRL_FHP_NO_STEADY", "1")
    pc.check_fused_batched_vs_oracle(L, 3, 3, variant=variant)


def test_emu_fused_engine_batched_iterations(L):
    pc.check_fused_batched_vs_oracle(L, 3, 4, delay=1)


def test_emu_fused_engine_block_sums_across_a_block_boundary(L, monkeypatch):
    """33 boards on two workgroups: the pass sums its root vectors per 32-board block (one full block + a block of one), the sum
    kernels start a level higher -- against the oracle's board-by-board canonical sum"""
    monkeypatch.setenv("PRL_FHP_GRID", "2")
    monkeypatch.setenv("PRL_FHP_BLOCK_SUM", "1")  # (the default keeps per-board rows below ~32 boards per workgroup)
    pc.check_fused_batched_vs_oracle(L, 33, 2, delay=0)


@pytest.mark.parametrize("variant,symmetrize", [("plus", True), ("linear", True), ("plus", False)])
def test_emu_fused_engine_weighted_boards_suit_classes(L, variant, symmetrize):
    """prl_solver_create_weighted: suit-class representatives with multiplicities (orbit sizes 4 / 12 / 24 among them), chance values averaged over
    the hands' suit orbits -- the fused engine against the oracle's restatement, bit for bit"""
    pc.check_weighted_vs_oracle(L, 4, 2, variant, symmetrize)


def test_emu_fused_engine_weighted_boards_checkpoint_resume(L):
    pc.check_weighted_checkpoint(L)


def test_emu_set_strategy_device_on_every_engine(L):
    """the device-side scatter of an agent's probabilities into the solver's columns: LEVELS (hand-order columns), the single-deal fused engine (sorted
    board storage, through the staging buffer) and the per-street engine (internal column order)"""
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    t = pc.fhp_tree_of(L, pc.fhp_boards(3))
    assert pc.check_set_strategy_device(L, t, lambda: _native.NativeSolver(t, "plus", 0, engine="levels", _lib=L)) == "levels"
    assert pc.check_set_strategy_device(L, t, lambda: _native.NativeSolver(t, "plus", 0, engine="fused", _lib=L)) == "fused"
    game = G.LimitHoldem.native_game(pc.env_args(G.LimitHoldem, 48, None))
    for i, v in enumerate((1, 2, 1, 1)):
        game.max_raises[i] = v
    t2 = _native.NativeTree(game, G.LimitHoldem.native_rules(), pc.multistreet_runouts(1, 2, 1), _lib=L)
    assert pc.check_set_strategy_device(L, t2, lambda: _native.NativeSolver(t2, "plus", 0, engine="auto", _lib=L)) == "fused"


def test_emu_suit_classes_are_checked_not_trusted(L):
    pc.check_symmetrize_is_validated(L)


def test_emu_fused_engine_float32_running_average_opt_in(L):
    """PRL_SOLVER_AVG_F32: generic and steady-state instantiations of the update passes with the average stored as float32"""
    pc.check_fused_avg_f32(L, 3, 4)


@pytest.mark.parametrize("no_steady", [False, True])
def test_emu_fused_engine_steady_state_specialisation(L, monkeypatch, no_steady):
    """CFR+ delay 0, batched: iterations 2.. run the steady-state instantiation of the two update passes (prl_fhp_pass.inc, FhpCtxT);
    PRL_FHP_NO_STEADY keeps the generic one -- both must equal the oracle bit for bit"""
    if no_steady:
        monkeypatch.setenv("PRL_FHP_NO_STEADY", "1")
    pc.check_fused_batched_vs_oracle(L, 3, 4, delay=0)


@pytest.mark.parametrize("fused,variant", [(False, "plus"), (False, "linear"), (True, "plus"), (True, "vanilla")])
def test_emu_checkpoint_resume(L, fused, variant):
    pc.check_checkpoint_resume(L, fused, variant)


def test_emu_br_of_random_strategy(L):
    pc.check_br_of_given_strategy(L, "StandardLeduc", 0, f64=True)
    pc.check_br_of_given_strategy(L, "StandardLeduc", 1, f64=False)


def test_emu_multistreet_limit_holdem_tree(L):
    """LimitHoldem (pre-flop, flop, turn, river; games.py:134-167) over 2 flops x 2 turns x 1 river: three chance levels"""
    from pokerrl_amd.game import games as G
    pc.check_multistreet_vs_oracle(L, G.LimitHoldem, 48, None, pc.multistreet_runouts(2, 2, 2), "plus", 1, max_raises=(1, 1, 0, 0))  # 130 nodes


@pytest.mark.parametrize("variant,runouts,max_raises,batched", [("plus", (2, 2, 1), (1, 1, 1, 1), False), ("vanilla", (1, 2, 2), (1, 1, 1, 1), True),
                                                                ("linear", (1, 1, 2), (1, 2, 1, 1), True), ("plus", (1, 1, 1), (1, 2, 3, 1), True),
                                                                ("plus", (1, 1, 1), (1, 5, 1, 1), False), ("linear", (1, 1, 1), (1, 1, 1, 5), True)])  # (five raises: the 33-node street, not last / last)
def test_emu_streets_engine_limit_holdem(L, variant, runouts, max_raises, batched):
    """the per-street fused engine (csrc/prl_st.h) on LimitHoldem trees with three dealing streets: engine=auto takes it, regrets /
    averages / exploitability history / average-strategy exploitability equal the oracle's bit for bit -- single iterations and the
    batched steady state, 9-, 15-, 21- and 33-node street subtrees, several outcomes per chance node"""
    from pokerrl_amd.game import games as G
    pc.check_streets_vs_oracle(L, G.LimitHoldem, 48, pc.multistreet_runouts(*runouts), variant, 3 if batched else 2, max_raises=max_raises, batched=batched)


@pytest.mark.parametrize("batched", [False, True])
def test_emu_streets_engine_float32_running_average_opt_in(L, batched):
    from pokerrl_amd.game import games as G
    pc.check_streets_avg_f32(L, G.LimitHoldem, 48, pc.multistreet_runouts(1, 2, 1), 3, max_raises=(1, 1, 1, 1), batched=batched)


def test_emu_streets_engine_best_response_of_an_explicit_strategy(L):
    """LocalBRMaster's evaluation on a multi-street tree: explicit float32 / float64 strategies on the per-street engine against the oracle"""
    from pokerrl_amd.game import games as G
    pc.check_streets_br_vs_oracle(L, G.LimitHoldem, 48, pc.multistreet_runouts(1, 2, 1), max_raises=(1, 2, 1, 1))


def test_emu_streets_engine_checkpoint_resume(L):
    """prl_solver_save_state / load_state on the per-street engine: a resumed solve continues bit for bit"""
    import numpy as np
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    t, s, _o = pc.make_streets_pair(L, G.LimitHoldem, 48, pc.multistreet_runouts(1, 2, 1), "plus", 0, (1, 1, 1, 1))
    s.iterations(2)
    blob = s.save_state()
    s2 = _native.NativeSolver(t, "plus", 0, engine="auto", _lib=L)
    s2.load_state(blob)
    s.iterations(2)
    s2.iterations(2)
    for k in ("regret", "avg", "expl_history"):
        assert np.array_equal(s.get(k), s2.get(k)), k
    assert np.array_equal(s.eval_avg(), s2.eval_avg())


@pytest.mark.parametrize("variant,stack,runouts,batched", [("plus", 4, (1, 1, 1), False), ("linear", 6, (1, 2, 1), True)])
def test_emu_streets_engine_all_in_run_outs(L, variant, stack, runouts, batched):
    """MIXED STREETS (csrc/prl_st.h): 4- and 6-chip stacks put all-in calls on every street, each dealt out as a chain of chance nodes down to showdown
    leaves (a decision-free forest on the level kernels, next to the street instances) -- engine=auto takes the per-street engine, results equal the
    oracle's bit for bit"""
    from pokerrl_amd.game import games as G
    pc.check_streets_vs_oracle(L, G.LimitHoldem, stack, pc.multistreet_runouts(*runouts), variant, 3 if batched else 2, batched=batched)


@pytest.mark.parametrize("variant,stack,runouts,batched,bundle", [("plus", 600, (1, 1, 1), False, 3), ("vanilla", 1200, (1, 1, 1), True, 0), ("linear", 600, (1, 2, 1), True, 2)])
def test_emu_streets_engine_discretized_nl_holdem(L, monkeypatch, variant, stack, runouts, batched, bundle):
    """MIXED STREETS: DiscretizedNLHoldem (games.py:114-131) with pot-sized raises -- a street's subtrees differ with the stacks behind (9-, 15- and
    21-node shapes side by side on one street, 6 to 9 (street, shape) groups) and every raise sequence that runs out of chips ends in a run-out chain;
    bundle: showdowns of the run-out forest per workgroup (PRL_ST_CHAIN_BUNDLE; trees this small would form none by themselves)"""
    from pokerrl_amd.game import bet_sets
    if bundle:
        monkeypatch.setenv("PRL_ST_CHAIN_BUNDLE", str(bundle))
    from pokerrl_amd.game import games as G
    pc.check_streets_vs_oracle(L, G.DiscretizedNLHoldem, stack, pc.multistreet_runouts(*runouts), variant, 3 if batched else 2, batched=batched, bets=bet_sets.POT_ONLY)


def test_emu_streets_engine_mixed_best_response_of_an_explicit_strategy(L):
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    pc.check_streets_br_vs_oracle(L, G.DiscretizedNLHoldem, 600, pc.multistreet_runouts(1, 2, 1), bets=bet_sets.POT_ONLY)


def test_emu_streets_engine_mixed_checkpoint_and_device_fill(L):
    """save / load and prl_solver_set_strategy_device on a tree with mixed street shapes and run-out chains"""
    import numpy as np
    from pokerrl_amd import _native
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    t, s, _o = pc.make_streets_pair(L, G.DiscretizedNLHoldem, 600, pc.multistreet_runouts(1, 1, 1), "linear", 0, bets=bet_sets.POT_ONLY)
    s.iterations(2)
    blob = s.save_state()
    s2 = _native.NativeSolver(t, "linear", 0, engine="auto", _lib=L)
    s2.load_state(blob)
    s.iterations(2)
    s2.iterations(2)
    for k in ("regret", "avg", "avg_sum", "expl_history"):
        assert np.array_equal(s.get(k), s2.get(k)), k
    assert np.array_equal(s.eval_avg(), s2.eval_avg())
    pc.check_set_strategy_device(L, t, lambda: _native.NativeSolver(t, "plus", 0, engine="auto", _lib=L))


def test_emu_streets_engine_refuses_what_it_cannot_walk(L):
    """two raise sizes (half pot, pot): decisions with four actions, street subtrees that are none of the registered shapes -- engine=auto falls back to
    the level-synchronous engine, engine=fused says why"""
    from helpers import env_args
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    game = G.DiscretizedNLHoldem.native_game(env_args(G.DiscretizedNLHoldem, 600, [0.5, 1.0]))
    t = _native.NativeTree(game, G.DiscretizedNLHoldem.native_rules(), pc.multistreet_runouts(1, 1, 1), _lib=L)
    assert _native.NativeSolver(t, "plus", 0, engine="auto", _lib=L).engine == "levels"
    with pytest.raises(_native.NativeError, match="street subtree"):
        _native.NativeSolver(t, "plus", 0, engine="fused", _lib=L)


def test_emu_streets_engine_refuses_a_first_deal_of_fewer_than_three_cards(L):
    """custom rules that deal 2 + 2 + 1 board cards: a street with 2 cards out leaves 1225 live hands and 49-entry card lists, beyond what the
    street pass is laid out for (<= 1209 / 48) -- engine=auto must take the level-synchronous engine, engine=fused must say why"""
    import copy
    import numpy as np
    from helpers import env_args
    from pokerrl_amd import _native
    from pokerrl_amd.game import games as G
    rules = copy.copy(G.LimitHoldem.native_rules())
    rules.board_cards_in_round[1], rules.board_cards_in_round[2], rules.board_cards_in_round[3] = 2, 2, 1
    rows = np.array([[0, 5, 10, 15, 20], [0, 5, 10, 15, 21]], np.int8)
    game = G.LimitHoldem.native_game(env_args(G.LimitHoldem, 48, None))
    for i in range(4):
        game.max_raises[i] = 1  # (a small betting tree: the level-synchronous engine runs on the emulator here)
    t = _native.NativeTree(game, rules, rows, _lib=L)
    assert _native.NativeSolver(t, "plus", 0, engine="auto", _lib=L).engine == "levels"
    with pytest.raises(_native.NativeError, match="fewer than 3 board cards"):
        _native.NativeSolver(t, "plus", 0, engine="fused", _lib=L)


def test_emu_multistreet_short_stack_run_outs(L):
    """4-chip stacks: all-ins on every street, each dealt out as a chain of chance nodes down to showdown leaves (41 chance nodes)"""
    from pokerrl_amd.game import games as G
    pc.check_multistreet_vs_oracle(L, G.LimitHoldem, 4, None, pc.multistreet_runouts(2, 2, 1), "vanilla", 1, expect_runout_chain=True)


def test_emu_all_in_before_the_deal_is_dealt_out_as_a_chance_chain(L):
    """Flop5Holdem with 250-chip stacks: the pot-sized pre-flop raise is all-in, the hand is dealt out (ValueFiller.py:160-175's
    case for 2-card ranges): a run-out chance node with showdown leaves next to the ordinary one"""
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    pc.check_multistreet_vs_oracle(L, G.Flop5Holdem, 250, bet_sets.POT_ONLY, pc.fhp_boards(3), "linear", 2, expect_runout_chain=False)


def test_emu_hand_rank_kernel(L):
    pc.check_hand_rank_golden(L)
    assert pc.check_hand_rank_checksums(L, n_chunks=2) == 2
