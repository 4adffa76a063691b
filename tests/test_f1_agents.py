"""SURVEY.md section 8f-1 (batched agent querying) and 8f-2 (EvalAgent pickle): a PyTorch policy agent evaluated by LocalBRMaster
through the batched protocol (one forward per history length) against the reference protocol (one positioned query per node);
store_to_disk / load_from_disk round trips; and -- where the reference is present -- a PokerRL-side EvalAgentBase subclass
evaluated by pokerrl_amd's LocalBRMaster with the reference LocalBRMaster's own result."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game.games import DiscretizedNLLeduc, StandardLeduc  # noqa: E402
from pokerrl_amd.game.wrappers import FlatLimitPokerEnvBuilder, HistoryEnvBuilder  # noqa: E402
from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase  # noqa: E402
from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase  # noqa: E402

REF = os.environ.get("POKERRL_REFERENCE", "/root/reference")


@pytest.fixture()
def emu_lib(monkeypatch):
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    L = _native.bind(build_emu.build())
    monkeypatch.setattr(_native, "lib", lambda: L)
    monkeypatch.setattr(_native, "require_device", lambda: None)
    return L


class Chief(ChiefBase):
    def pull_current_eval_strategy(self, last):
        return None, last


def t_prof_of(game_cls, bldr_cls, path, bets=None, device="cpu"):
    return TrainingProfileBase(
        name="f1", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=game_cls,
        env_bldr_cls=bldr_cls, start_chips=None, eval_modes_of_algo=("POLICY",), eval_stack_sizes=None,
        module_args={"env": game_cls.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=bets) if bets is not None else game_cls.ARGS_CLS(n_seats=2)},
        path_data=str(path), device_inference=device)


def br_of(t_prof, agent_cls, **tree_kw):
    from pokerrl_amd.eval.br.LocalBRMaster import LocalBRMaster
    chief = Chief(t_prof)
    br = LocalBRMaster(t_prof=t_prof, chief_handle=chief, eval_agent_cls=agent_cls, **tree_kw)
    br.update_weights()
    br.evaluate(iter_nr=0)
    vals, _ = chief.get_new_values()
    (_exp, graphs), = vals.items()
    (_g, series), = graphs.items()
    return series[0][1], br


def check_batched_vs_per_node(game_cls, bldr_cls, tmp_path, bets=None, device="cpu"):
    from pokerrl_amd.rl.neural import TorchPolicyAgent

    class PerNode(TorchPolicyAgent):  # the same network, reference protocol only
        get_a_probs_for_each_hand_in_nodes = None
        get_a_probs_for_each_hand_in_nodes_device = None

    t_prof = t_prof_of(game_cls, bldr_cls, tmp_path, bets, device)
    e_b, br_b = br_of(t_prof, TorchPolicyAgent)
    e_p, br_p = br_of(t_prof, PerNode)
    tree_b, tree_p = br_b._game_trees[0], br_p._game_trees[0]
    n_dec = int(np.sum(tree_b._kind == 0))
    assert br_p._eval_agent.n_forwards == n_dec                      # one forward per decision node ...
    lengths = {n.depth for n in tree_b.nodes() if not n.is_terminal and n.p_id_acting_next != tree_b.CHANCE_ID}
    assert br_b._eval_agent.n_forwards <= len(lengths) < n_dec / 4   # ... against one per history length
    sb, sp = tree_b.solver.get("strategy"), tree_p.solver.get("strategy")
    assert np.allclose(sb, sp, rtol=0, atol=2e-6)                    # same numbers up to the GEMM's batch-size dependent rounding
    assert np.allclose(sb.reshape(-1, sb.shape[-1])[:4].sum(), sp.reshape(-1, sp.shape[-1])[:4].sum(), rtol=1e-5)
    assert e_b > 0 and abs(e_b - e_p) <= 1e-4 * e_p
    return e_b


def check_device_fill_equals_host_fill(game_cls, bldr_cls, tmp_path, bets=None, device="cpu", **tree_kw):
    """fill_with_agent_policy with the probabilities LEFT IN HBM (TorchPolicyAgent.get_a_probs_for_each_hand_in_nodes_device -> prl_solver_set_strategy_device:
    the network's output tensor scattered into the solver's columns on the GPU) against the host path (tensor -> NumPy -> [n_cols, R] -> upload): the
    same strategy columns and the same best-response values, bit for bit"""
    from pokerrl_amd.rl.neural import TorchPolicyAgent

    class HostFill(TorchPolicyAgent):
        DEVICE_RESIDENT_FILL = False

    calls = []

    class DeviceFill(TorchPolicyAgent):
        def get_a_probs_for_each_hand_in_nodes(self, nodes):  # the device path must not fall back to this
            raise AssertionError("host path taken")

        def get_a_probs_for_each_hand_in_nodes_device(self, nodes):
            out = super().get_a_probs_for_each_hand_in_nodes_device(nodes)
            calls.append(tuple(out.shape))
            return out

    t_prof = t_prof_of(game_cls, bldr_cls, tmp_path, bets, device)
    e_h, br_h = br_of(t_prof, HostFill, **tree_kw)
    e_d, br_d = br_of(t_prof, DeviceFill, **tree_kw)
    tree_h, tree_d = br_h._game_trees[0], br_d._game_trees[0]
    assert calls == [(int(np.sum(tree_d._kind == 0)), tree_d.native_tree.range_size, tree_d.env_bldr.N_ACTIONS)]
    assert tree_h.solver.engine == tree_d.solver.engine
    sh, sd = tree_h.solver.get("strategy"), tree_d.solver.get("strategy")
    assert sh.shape == sd.shape and np.array_equal(sh, sd)
    assert np.array_equal(tree_h.solver.exploitability(), tree_d.solver.exploitability())
    assert e_h == e_d and e_d > 0
    return e_d, tree_d.solver.engine


def fhp_boards_for_fill(n):
    sys.path.insert(0, os.path.dirname(HERE))
    import bench
    return bench.seeded_boards(n, 0)


def test_device_resident_agent_fill_equals_host_fill_emu(emu_lib, tmp_path):
    from pokerrl_amd.game.games import Flop5Holdem
    _e, eng = check_device_fill_equals_host_fill(StandardLeduc, HistoryEnvBuilder, tmp_path)
    assert eng == "levels"
    _e, eng = check_device_fill_equals_host_fill(Flop5Holdem, HistoryEnvBuilder, tmp_path, boards=fhp_boards_for_fill(3), engine="fused")
    assert eng == "fused"  # the sorted board columns: scattered through the staging buffer and compacted on the "device"


@pytest.mark.gpu
def test_gpu_device_resident_agent_fill_on_the_per_street_engine(tmp_path):
    """a multi-street tree (LimitHoldem, one run-out, full betting: 17 221 nodes) on the per-street fused engine, whose columns live in an internal order
    (trunk, then street by street, instance by instance): the device-side scatter goes through that order's column map"""
    import parity_cases as pc
    from pokerrl_amd.game.games import LimitHoldem
    _e, eng = check_device_fill_equals_host_fill(LimitHoldem, HistoryEnvBuilder, tmp_path, device="cuda", boards=pc.multistreet_runouts(1, 1, 1), engine="auto")
    assert eng == "fused"


@pytest.mark.gpu
def test_gpu_device_resident_agent_fill_on_a_4096_board_tree(tmp_path):
    """best response against a neural agent on Flop5Holdem x 4096 boards (24 578 decision nodes, a 391 MB probability tensor): the strategy never visits
    the host -- and equals the host path bit for bit"""
    from pokerrl_amd.game.games import Flop5Holdem
    _e, eng = check_device_fill_equals_host_fill(Flop5Holdem, HistoryEnvBuilder, tmp_path, device="cuda", boards=fhp_boards_for_fill(4096), engine="fused")
    assert eng == "fused"
    check_device_fill_equals_host_fill(DiscretizedNLLeduc, HistoryEnvBuilder, tmp_path, bets=bet_sets.B_3, device="cuda")


@pytest.mark.parametrize("bldr_cls", [HistoryEnvBuilder, FlatLimitPokerEnvBuilder])
def test_batched_forward_equals_per_node_queries_emu(emu_lib, tmp_path, bldr_cls):
    check_batched_vs_per_node(StandardLeduc, bldr_cls, tmp_path)


@pytest.mark.gpu
@pytest.mark.parametrize("game_cls,bldr_cls,bets", [(StandardLeduc, HistoryEnvBuilder, None), (StandardLeduc, FlatLimitPokerEnvBuilder, None),
                                                    (DiscretizedNLLeduc, HistoryEnvBuilder, bet_sets.B_2)])
def test_gpu_batched_forward_equals_per_node_queries(tmp_path, game_cls, bldr_cls, bets):
    """the network runs on the GPU too (PyTorch-ROCm): device_inference = cuda"""
    check_batched_vs_per_node(game_cls, bldr_cls, tmp_path, bets, device="cuda")


def test_eval_agent_store_and_load_round_trip(tmp_path):
    """EvalAgentBase.store_to_disk / load_from_disk (EvalAgentBase.py:75-95,160-170): weights, mode and the env wrapper's state
    (observation history included) survive the pickle"""
    from pokerrl_amd.rl.neural import TorchPolicyAgent
    t_prof = t_prof_of(StandardLeduc, HistoryEnvBuilder, tmp_path)
    a = TorchPolicyAgent(t_prof=t_prof, mode="POLICY")
    with __import__("torch").no_grad():
        for p in a._net.parameters():
            p.add_(0.25)  # not the constructor's weights
    np.random.seed(3)
    a.reset()
    a.get_action(step_env=True)
    want = a.get_a_probs_for_each_hand()
    a.store_to_disk(path=str(tmp_path / "agents"), file_name="agent7")
    b = TorchPolicyAgent.load_from_disk(path_to_eval_agent=str(tmp_path / "agents" / "agent7.pkl"))
    assert b.get_mode() == "POLICY"
    assert np.array_equal(b._internal_env_wrapper.get_current_obs(), a._internal_env_wrapper.get_current_obs())
    assert np.array_equal(b.get_a_probs_for_each_hand(), want)
    for (k1, v1), (k2, v2) in zip(a._net.state_dict().items(), b._net.state_dict().items()):
        assert k1 == k2 and np.array_equal(v1.cpu().numpy(), v2.cpu().numpy())
    # the hash agent (no weights) round-trips too
    from pokerrl_amd.rl import hash_agent
    from pokerrl_amd.rl.base_cls.EvalAgentBase import EvalAgentBase
    H = hash_agent.make_agent_cls(EvalAgentBase, seed=7)
    h = H(t_prof=t_prof_of(StandardLeduc, HistoryEnvBuilder, tmp_path), mode="HASH")
    h.store_to_disk(path=str(tmp_path / "agents"), file_name="hash")
    assert H.load_from_disk(str(tmp_path / "agents" / "hash.pkl")).get_mode() == "HASH"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "PokerRL")), reason="the reference does not travel to the GPU box")
@pytest.mark.parametrize("game_name,bets_name", [("StandardLeduc", None), ("DiscretizedNLLeduc", "POT_ONLY")])
def test_reference_side_agent_evaluated_by_pokerrl_amd_br_master(emu_lib, tmp_path, game_name, bets_name):
    """Interop at the plugin surface: the SAME agent class -- a subclass of the REFERENCE's PokerRL.rl.base_cls.EvalAgentBase, with
    the reference's env wrapper and env inside -- is evaluated by the reference's LocalBRMaster and by pokerrl_amd's LocalBRMaster
    (which positions it on pokerrl_amd tree nodes through node.parent / node.env_state / node.tree.CHANCE_ID). Equal results."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import ref_harness
    ref_harness.setup()
    import PokerRL.game.bet_sets as ref_bets
    from PokerRL.eval.br.LocalBRMaster import LocalBRMaster as RefBR
    from PokerRL.game import games as ref_games
    from PokerRL.game.wrappers import HistoryEnvBuilder as RefHistoryEnvBuilder
    from PokerRL.rl.base_cls.EvalAgentBase import EvalAgentBase as RefEvalAgentBase
    from PokerRL.rl.base_cls.TrainingProfileBase import TrainingProfileBase as RefTProf
    from PokerRL.rl.base_cls.workers.ChiefBase import ChiefBase as RefChiefBase
    from pokerrl_amd.rl import hash_agent

    Agent = hash_agent.make_agent_cls(RefEvalAgentBase, seed=7)  # the hash policy: different at every node, hand and action
    ref_game = getattr(ref_games, game_name)
    ref_args = ref_game.ARGS_CLS(n_seats=2, bet_sizes_list_as_frac_of_pot=getattr(ref_bets, bets_name)) if bets_name else ref_game.ARGS_CLS(n_seats=2)
    ref_prof = RefTProf(name="x", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9,
                        game_cls=ref_game, env_bldr_cls=RefHistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("HASH",),
                        eval_stack_sizes=None, module_args={"env": ref_args}, path_data=str(tmp_path / "ref"))

    class RefChief(RefChiefBase):
        def pull_current_eval_strategy(self, last):
            return None, last

    rc = RefChief(ref_prof)
    rbr = RefBR(t_prof=ref_prof, chief_handle=rc, eval_agent_cls=Agent)
    rbr.update_weights()
    rbr.evaluate(iter_nr=0)
    (_e, graphs), = rc.get_new_values()[0].items()
    (_g, series), = graphs.items()
    want = series[0][1]

    # pokerrl_amd's master, the reference-side agent (it builds its own -- reference -- env from the profile's class names)
    from pokerrl_amd.eval.br.LocalBRMaster import LocalBRMaster
    chief = Chief(ref_prof)
    br = LocalBRMaster(t_prof=ref_prof, chief_handle=chief, eval_agent_cls=Agent)
    assert isinstance(br._eval_agent, RefEvalAgentBase)
    br.update_weights()
    br.evaluate(iter_nr=0)
    (_e, graphs), = chief.get_new_values()[0].items()
    (_g, series), = graphs.items()
    assert series[0][1] == want, (series[0][1], want)
