"""CFR variants written against the reference's protected hook methods (_CFRBase.py:140-144,187-196): a subclass of
pokerrl_amd.cfr._CFRBase.CFRBase that leaves _VARIANT = None and implements the four hooks in NumPy -- here plain regret matching
with reach-weighted averaging (the formulas of VanillaCFR.py:26-77) and a discounted variant nobody compiled into the kernels --
runs the reference's iteration loop with the tree passes on the device. The plain one must log the REFERENCE's own exploitability
series (tests/golden/cfr_StandardLeduc_VanillaCFR.npz) to the last digit."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

from helpers import golden  # noqa: E402
from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.cfr._CFRBase import CFRBase  # noqa: E402
from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game.games import StandardLeduc  # noqa: E402
from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase  # noqa: E402


class HookedRegretMatching(CFRBase):
    """user-side variant: cumulative regrets, regret matching on the positive part, reach-weighted average"""
    DISCOUNT = None  # None: plain; else regrets are multiplied by t / (t + DISCOUNT) before the new ones are added

    def __init__(self, name, chief_handle, game_cls, agent_bet_set, starting_stack_sizes=None):
        super().__init__(name=name, chief_handle=chief_handle, game_cls=game_cls, starting_stack_sizes=starting_stack_sizes,
                         agent_bet_set=agent_bet_set, algo_name="HookedRM")
        self.reset()

    def _regret_formula_first_it(self, ev_all_actions, strat_ev):
        return ev_all_actions - strat_ev

    def _regret_formula_after_first_it(self, ev_all_actions, strat_ev, last_regrets):
        if self.DISCOUNT is not None:
            t = np.float32(self._iter_counter)
            last_regrets = last_regrets * (t / (t + np.float32(self.DISCOUNT)))
        return ev_all_actions - strat_ev + last_regrets

    def _compute_new_strategy(self, p_id):
        for t_idx, tree in enumerate(self._trees):
            R = self._env_bldrs[t_idx].rules.RANGE_SIZE
            for node in tree.nodes():
                if node.p_id_acting_next == p_id and not node.is_terminal:
                    n = len(node.children)
                    pos = np.maximum(node.data["regret"], 0)
                    s = np.expand_dims(np.sum(pos, axis=1), axis=1).repeat(n, axis=1)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        node.strategy = np.where(s > 0.0, pos / s, np.full(shape=(R, n), fill_value=1.0 / n, dtype=np.float32))

    def _add_strategy_to_average(self, p_id):
        for tree in self._trees:
            for node in tree.nodes():
                if node.p_id_acting_next == p_id and not node.is_terminal:
                    contrib = node.strategy * np.expand_dims(node.reach_probs[p_id], axis=1)
                    node.data["avg_strat_sum"] = node.data["avg_strat_sum"] + contrib if self._iter_counter > 0 else contrib
                    s = np.expand_dims(np.sum(node.data["avg_strat_sum"], axis=1), axis=1)
                    n = len(node.allowed_actions)
                    with np.errstate(divide="ignore", invalid="ignore"):
                        node.data["avg_strat"] = np.where(s == 0, np.full(shape=n, fill_value=1.0 / n), node.data["avg_strat_sum"] / s)


class Chief(ChiefBase):
    pass


def run(cls, n_iters):
    chief = Chief(t_prof=None)
    algo = cls(name="hooks", chief_handle=chief, game_cls=StandardLeduc, agent_bet_set=bet_sets.POT_ONLY)
    for _ in range(n_iters):
        algo.iteration()
    vals, _ = chief.get_new_values()
    curr = [v for k, v in vals.items() if "_Curr_S" in k][0]["Evaluation/" + StandardLeduc.WIN_METRIC]
    avg = [v for k, v in vals.items() if "_Avg_total_S" in k][0]["Evaluation/" + StandardLeduc.WIN_METRIC]
    return np.array(curr, np.float64), np.array(avg, np.float64)


def check_hooked_variants():
    g = golden("cfr_StandardLeduc_VanillaCFR.npz")
    n = 6
    curr, avg = run(HookedRegretMatching, n)
    assert np.array_equal(curr[:, 1], g["curr_series"][:n + 1, 1]), (curr[:, 1], g["curr_series"][:n + 1, 1])
    assert np.array_equal(avg[:, 1], g["avg_series"][:n, 1])

    class Discounted(HookedRegretMatching):
        DISCOUNT = 1.5

    c2, a2 = run(Discounted, n)
    assert c2[0, 1] == curr[0, 1] and c2[1, 1] == curr[1, 1]       # the discount starts with the second update
    assert not np.array_equal(c2[:, 1], curr[:, 1]) and a2[-1, 1] < a2[0, 1]  # a different, still converging, run


def test_hooked_variant_reproduces_reference_vanilla_series_emu(monkeypatch):
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    L = _native.bind(build_emu.build())
    monkeypatch.setattr(_native, "lib", lambda: L)
    monkeypatch.setattr(_native, "require_device", lambda: None)
    check_hooked_variants()


@pytest.mark.gpu
def test_gpu_hooked_variant_reproduces_reference_vanilla_series():
    check_hooked_variants()
