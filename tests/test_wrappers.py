"""Observation wrappers / env builders (pokerrl_amd/game/wrappers.py) against the reference's (tests/golden/wrappers.npz, captured
from PokerRL by tests/golden/make_wrapper_golden.py): builder attributes, the wrapped observation after every step of seeded
play, the history a wrapper rebuilds when it is positioned on a public-tree node (the thing that lets a recurrent agent be
queried by StrategyFiller / LocalBRMaster), and the batched `history_of_nodes`."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from helpers import golden  # noqa: E402
from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game import wrappers as W  # noqa: E402
from pokerrl_amd.game.games import DiscretizedNLLeduc, StandardLeduc  # noqa: E402

CASES = {
    "History_StandardLeduc": (W.HistoryEnvBuilder, {}, StandardLeduc, 13, None),
    "HistoryInv_StandardLeduc": (W.HistoryEnvBuilder, {"invert_history_order": True}, StandardLeduc, 13, None),
    "Flat_StandardLeduc": (W.FlatLimitPokerEnvBuilder, {}, StandardLeduc, 13, None),
    "History_DiscretizedNLLeduc": (W.HistoryEnvBuilder, {}, DiscretizedNLLeduc, 1500, bet_sets.B_3),
    "Vanilla_StandardLeduc": (W.VanillaEnvBuilder, {}, StandardLeduc, 13, None),
}


def make(name):
    bcls, kw, game, stack, bets = CASES[name]
    args = game.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bets) if bets is not None \
        else game.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack])
    return bcls(env_cls=game, env_args=args, **kw), args


def unpack(g, key):
    rows, off = g[key + "_obs"], g[key + "_off"]
    return [rows[off[i]:off[i + 1]] for i in range(len(off) - 1)]


@pytest.mark.parametrize("name", sorted(CASES))
def test_builder_attributes_and_play_observations_match_reference(name):
    g = golden("wrappers.npz")
    bldr, args = make(name)
    attrs = [bldr.pub_obs_size, bldr.priv_obs_size, bldr.complete_obs_size, bldr.obs_size_board, bldr.obs_size_player_info_each,
             bldr.obs_size_table_state, getattr(bldr, "action_vector_size", 0)]
    assert attrs == g[name + "_attrs"].tolist()
    assert sorted(bldr.obs_table_state_idxs + sum(bldr.obs_players_idxs, []) + bldr.obs_board_idxs) == list(range(27))
    w = bldr.get_new_wrapper(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
    want, acts = unpack(g, name + "_play"), g[name + "_play_act"]
    np.random.seed(5)
    for i, a in enumerate(acts):
        o = w.reset()[0] if a < 0 else w.step(int(a))[0]
        assert o.dtype == np.float32
        assert np.array_equal(np.atleast_2d(o), want[i]), (name, i)
    sd = w.state_dict()  # round trip of the wrapper state (history included)
    w2 = bldr.get_new_wrapper(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
    w2.load_state_dict(sd)
    assert np.array_equal(w2.get_current_obs(), w.get_current_obs())


@pytest.fixture()
def emu_lib(monkeypatch):
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    L = _native.bind(build_emu.build())
    monkeypatch.setattr(_native, "lib", lambda: L)
    monkeypatch.setattr(_native, "require_device", lambda: None)
    return L


def check_node_histories(name):
    from pokerrl_amd.game.PublicTree import PublicTree
    g = golden("wrappers.npz")
    bldr, args = make(name)
    tree = PublicTree(env_bldr=bldr, stack_size=args.starting_stack_sizes_list, stop_at_street=None)
    tree.build_tree()
    nodes = [n for n in tree.nodes() if (not n.is_terminal) and n.p_id_acting_next != tree.CHANCE_ID]
    want = unpack(g, name + "_node")
    assert len(nodes) == len(want)
    w = bldr.get_new_wrapper(is_evaluating=True, stack_size=args.starting_stack_sizes_list)
    for i in list(range(0, len(nodes), 7)) + [len(nodes) - 1]:  # the per-node replay (reference protocol) on a sample
        w.set_to_public_tree_node_state(nodes[i])
        assert np.array_equal(np.atleast_2d(w.get_current_obs()), want[i]), (name, i)
        assert w.env.current_player.seat_id == nodes[i].p_id_acting_next
    batched = W.history_of_nodes(bldr, nodes, stack_size=args.starting_stack_sizes_list)  # all nodes, one tree walk
    for i, (a, b) in enumerate(zip(batched, want)):
        assert np.array_equal(np.atleast_2d(a), b), (name, i)


@pytest.mark.parametrize("name", ["History_StandardLeduc", "HistoryInv_StandardLeduc", "Flat_StandardLeduc"])
def test_node_histories_match_reference_emu(emu_lib, name):
    check_node_histories(name)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["History_StandardLeduc", "Flat_StandardLeduc", "History_DiscretizedNLLeduc"])
def test_gpu_node_histories_match_reference(name):
    check_node_histories(name)
