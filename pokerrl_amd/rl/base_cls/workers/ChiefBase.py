"""
Logging sink of the tabular solvers and evaluators. Same interface as the reference's ChiefBase (PokerRL/rl/base_cls/
workers/ChiefBase.py:44-69): `create_experiment`, `add_scalar`, `get_new_values` -- that is all the hot path touches
(`_CFRBase.py:78-94,198-216,257-262`, `EvaluatorMasterBase.py:134-156`). Actor orchestration (ray) is out of scope.
"""
from collections import defaultdict


class ChiefBase:
    def __init__(self, t_prof=None):
        self._t_prof = t_prof
        self._all = {}                       # experiment -> graph -> [[step, value], ...]
        self._fresh = defaultdict(lambda: defaultdict(list))

    # ---- hooks that algorithm-specific chiefs override -------------------------------------------------------------
    def pull_current_eval_strategy(self, last_iteration_receiver_has):
        raise NotImplementedError

    def export_agent(self, step):
        raise NotImplementedError

    # ---- log buffer --------------------------------------------------------------------------------------------------
    def create_experiment(self, name):
        self._all.setdefault(name, {})
        return name

    def add_scalar(self, exp_name, graph_name, step, value):
        if exp_name not in self._all:
            raise AttributeError("Should create experiment before adding to it")
        self._all[exp_name].setdefault(graph_name, []).append([step, value])
        self._fresh[exp_name][graph_name].append([step, value])

    def get_new_values(self):
        """({experiment: {graph: [[step, value], ...]}} since the last call, [all experiment names])"""
        fresh = {e: dict(g) for e, g in self._fresh.items()}
        self._fresh = defaultdict(lambda: defaultdict(list))
        return fresh, list(self._all.keys())
