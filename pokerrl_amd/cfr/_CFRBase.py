"""
CFRBase: full-width tabular CFR on a public tree -- the reference's class (PokerRL/cfr/_CFRBase.py:13-278) with the
arithmetic on an MI355X. Same constructor, `reset()`, `iteration()`, `name`, `algo_name`, `iter_counter`, same experiment
names and logged scalars (exploitability of the current and of the average strategy, mean over seats x EV_NORMALIZER).

One `iteration()` = both seats updated (EV -> regrets -> regret matching -> reach -> average), EVs recomputed, current
strategy exploitability logged, average strategy evaluated (_CFRBase.py:122-134) -- all as HIP kernels behind
`pokerrl_amd._native.NativeSolver`; nothing is computed in Python. The regret / averaging formulas of the three variants
are compiled into the kernels (VanillaCFR.py, CFRPlus.py, LinearCFR.py), so the reference's protected hooks
(`_regret_formula_*`, `_compute_new_strategy`, `_add_strategy_to_average`) are not overridable here.

Extra keyword arguments: `boards` (chance outcomes for 2-hole-card games, which the reference cannot build at all) and
`engine` ("auto" | "levels" | "fused").
"""
import copy

import numpy as np

from pokerrl_amd.game.PublicTree import PublicTree
from pokerrl_amd.game.wrappers import HistoryEnvBuilder
from pokerrl_amd import _native
from pokerrl_amd.rl.rl_util import get_env_cls_from_str


class CFRBase:
    _VARIANT = None

    def __init__(self, name, chief_handle, game_cls, agent_bet_set, algo_name, starting_stack_sizes=None, delay=0, boards=None,
                 engine="auto"):
        self._name = name
        self._n_seats = 2
        self._chief_handle = chief_handle
        self._starting_stack_sizes = [game_cls.DEFAULT_STACK_SIZE] if starting_stack_sizes is None else copy.deepcopy(starting_stack_sizes)
        self._game_cls_str = game_cls.__name__
        self._env_args = [game_cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[s, s], bet_sizes_list_as_frac_of_pot=agent_bet_set)
                          for s in self._starting_stack_sizes]
        self._env_bldrs = [HistoryEnvBuilder(env_cls=get_env_cls_from_str(self._game_cls_str), env_args=a) for a in self._env_args]
        self._trees = [PublicTree(env_bldr=b, stack_size=a.starting_stack_sizes_list, stop_at_street=None, boards=boards, engine=engine)
                       for b, a in zip(self._env_bldrs, self._env_args)]
        for tree in self._trees:
            tree.build_tree(variant=self._VARIANT, delay=delay)
            print("Tree with stack size", tree.stack_size, "has", tree.n_nodes, "nodes out of which", tree.n_nonterm, "are non-terminal.")
        self._algo_name = algo_name
        c = chief_handle.create_experiment
        self._exps_curr_total = [c(name + "_Curr_S" + str(s) + "_total_" + algo_name) for s in self._starting_stack_sizes]
        self._exps_avg_total = [c(name + "_Avg_total_S" + str(s) + "_" + algo_name) for s in self._starting_stack_sizes]
        self._exp_all_averaged_curr_total = c(name + "_Curr_total_averaged_" + algo_name)
        self._exp_all_averaged_avg_total = c(name + "_Avg_total_averaged_" + algo_name)
        self._iter_counter = None

    name = property(lambda s: s._name)
    algo_name = property(lambda s: s._algo_name)
    iter_counter = property(lambda s: s._iter_counter)

    def reset(self):  # _CFRBase.py:110-120
        self._iter_counter = 0
        for t in self._trees:
            t.solver.reset()
            t._invalidate()
        self._log_curr_strat_expl()

    def _advance(self, n):
        """n iterations of every tree (one tree per starting stack size): Leduc-sized trees advance together in one launch, one
        workgroup (CU) each (prl_solver_iterations_many); otherwise one tree after the other"""
        solvers = [t.solver for t in self._trees]
        batched = False
        if len(solvers) > 1:
            try:
                _native.NativeSolver.iterations_many(solvers, n)
                batched = True
            except _native.NativeError as e:
                if e.status != _native.ERR_UNSUPPORTED:  # a HIP failure part-way through must surface, not be re-run
                    raise
                batched = False  # not all of them are small 1-hole-card trees (refused before anything was launched)
        for t in self._trees:
            if not batched:
                t.solver.iterations(n)
            t._invalidate()

    def iteration(self):  # _CFRBase.py:122-134
        self._advance(1)
        self._iter_counter += 1
        self._log_curr_strat_expl()
        self._evaluate_avg_strats()

    def iterations(self, n, log=True):
        """n iterations back to back on the GPU without a host round trip per iteration; logs afterwards from the
        device-side exploitability history (average-strategy evaluation only after the last one)."""
        start = self._iter_counter
        self._advance(n)
        self._iter_counter += n
        if log:
            hists = [t.solver.get("expl_history") for t in self._trees]
            for k in range(start + 1, self._iter_counter + 1):
                self._log_curr(k, [h[k] for h in hists])
            self._evaluate_avg_strats()

    def _scaled(self, t_idx, expl2):
        n = self._env_bldrs[t_idx].env_cls.EV_NORMALIZER
        return sum(float(expl2[p]) * n for p in range(self._n_seats)) / self._n_seats

    def _log_curr(self, step, expls):
        totals = []
        for t_idx, e in enumerate(expls):
            metric = self._env_bldrs[t_idx].env_cls.WIN_METRIC
            totals.append(self._scaled(t_idx, e))
            self._chief_handle.add_scalar(self._exps_curr_total[t_idx], "Evaluation/" + metric, step, totals[-1])
        self._chief_handle.add_scalar(self._exp_all_averaged_curr_total, "Evaluation/" + metric, step, sum(totals) / float(len(totals)))

    def _log_curr_strat_expl(self):  # _CFRBase.py:198-216
        self._log_curr(self._iter_counter, [t.solver.exploitability() for t in self._trees])

    def _evaluate_avg_strats(self):  # _CFRBase.py:218-262
        totals = []
        for t_idx, t in enumerate(self._trees):
            metric = self._env_bldrs[t_idx].env_cls.WIN_METRIC
            totals.append(self._scaled(t_idx, t.solver.eval_avg()))
            self._chief_handle.add_scalar(self._exps_avg_total[t_idx], "Evaluation/" + metric, self._iter_counter, totals[-1])
        self._chief_handle.add_scalar(self._exp_all_averaged_avg_total, "Evaluation/" + metric, self._iter_counter,
                                      sum(totals) / float(len(totals)))

    # ---- checkpoint / resume (the reference's CFR has none; prl_solver_save_state / load_state) ----------------------------
    def state_dict(self):
        return {"iter_counter": self._iter_counter, "solvers": [t.solver.save_state() for t in self._trees]}

    def load_state_dict(self, state):
        """Into an instance constructed with the same arguments: the run continues bit-identically."""
        assert len(state["solvers"]) == len(self._trees)
        for t, blob in zip(self._trees, state["solvers"]):
            t.solver.load_state(blob)
            t._invalidate()
        self._iter_counter = state["iter_counter"]

    # ---- access to the tabular state (column-major [n_cols, R]; column first_col[node] + a == node.strategy[:, a]) ------
    def regrets(self, t_idx=0):
        return self._trees[t_idx].solver.get("regret")

    def average_strategy(self, t_idx=0):
        return self._trees[t_idx].solver.get("avg")
