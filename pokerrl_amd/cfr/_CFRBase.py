"""
CFRBase: full-width tabular CFR on a public tree -- the reference's class (PokerRL/cfr/_CFRBase.py:13-278) with the
arithmetic on an MI355X. Same constructor, `reset()`, `iteration()`, `name`, `algo_name`, `iter_counter`, same experiment
names and logged scalars (exploitability of the current and of the average strategy, mean over seats x EV_NORMALIZER).

One `iteration()` = both seats updated (EV -> regrets -> regret matching -> reach -> average), EVs recomputed, current
strategy exploitability logged, average strategy evaluated (_CFRBase.py:122-134) -- all as HIP kernels behind
`pokerrl_amd._native.NativeSolver`; nothing is computed in Python. The regret / averaging formulas of the three built-in variants
are compiled into the kernels (VanillaCFR.py, CFRPlus.py, LinearCFR.py).

Other variants: a subclass that leaves `_VARIANT = None` and implements the reference's protected hooks `_regret_formula_first_it`,
`_regret_formula_after_first_it`, `_compute_new_strategy`, `_add_strategy_to_average` (_CFRBase.py:140-144,187-196) exactly as it
would against the reference -- NumPy on `node.data["regret" | "avg_strat" | "avg_strat_sum"]`, `node.strategy`, `node.ev`,
`node.reach_probs` -- runs the reference's iteration loop with the tree passes (reach, EV / best response, exploitability) on the
GPU and only those four formulas on the host (level-synchronous engine: every per-node vector is observable).

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
                 engine="auto", n_boards=None, max_outcomes=None, board_seed=None, suit_isomorphism=None):
        self._name = name
        self._n_seats = 2
        self._chief_handle = chief_handle
        self._starting_stack_sizes = [game_cls.DEFAULT_STACK_SIZE] if starting_stack_sizes is None else copy.deepcopy(starting_stack_sizes)
        self._game_cls_str = game_cls.__name__
        self._env_args = [game_cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[s, s], bet_sizes_list_as_frac_of_pot=agent_bet_set)
                          for s in self._starting_stack_sizes]
        self._env_bldrs = [HistoryEnvBuilder(env_cls=get_env_cls_from_str(self._game_cls_str), env_args=a) for a in self._env_args]
        self._host_hooks = self._VARIANT is None  # a variant written against the reference's hook methods
        if self._host_hooks:
            engine = "levels"
        if boards is None:  # the builder deals the chance outcomes from the deck (all of them, or the capped / seeded subset): board_enum.py
            from pokerrl_amd.game import board_enum
            # (Flop5Holdem with the reference's default arguments -- every board -- is solved through its 134 459 suit classes: board_enum)
            boards, board_mult = board_enum.default_boards_or_classes(get_env_cls_from_str(self._game_cls_str), n_boards=n_boards, max_outcomes=max_outcomes,
                                                                      seed=board_seed, suit_isomorphism=suit_isomorphism)
        else:
            board_mult = None
        if board_mult is not None and self._host_hooks:
            raise ValueError("a CFR variant written against the reference's hook methods runs on per-node vectors (LEVELS engine): give it boards= / "
                             "n_boards=; the whole game through its suit classes needs a built-in variant (CFRPlus / LinearCFR / VanillaCFR)")
        self._boards, self._engine = boards, engine
        self._trees = [PublicTree(env_bldr=b, stack_size=a.starting_stack_sizes_list, stop_at_street=None, boards=boards, engine=engine, board_mult=board_mult,
                                  suit_isomorphism=True if board_mult is not None else None)  # (the classes of the whole game, as enumerated above)
                       for b, a in zip(self._env_bldrs, self._env_args)]
        self._eval_trees = None
        for tree in self._trees:
            tree.build_tree(variant="vanilla" if self._host_hooks else self._VARIANT, delay=delay)
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
        if self._host_hooks:
            for t in self._trees:
                for n in t.nodes():
                    n.data = {"regret": None, "avg_strat": None, "avg_strat_sum": None}
                t.fill_uniform_random()
            self._compute_cfv()
            self._log_curr_strat_expl()
            return
        for t in self._trees:
            t.solver.reset()
            t._invalidate()
        self._log_curr_strat_expl()

    # ---- the reference's hook protocol (_CFRBase.py:136-196): used when a subclass leaves _VARIANT = None ---------------------
    def _regret_formula_first_it(self, ev_all_actions, strat_ev):
        raise NotImplementedError

    def _regret_formula_after_first_it(self, ev_all_actions, strat_ev, last_regrets):
        raise NotImplementedError

    def _compute_new_strategy(self, p_id):
        raise NotImplementedError

    def _add_strategy_to_average(self, p_id):
        raise NotImplementedError

    def _compute_cfv(self):
        for t in self._trees:
            t.compute_ev()

    def _update_reach_probs(self):
        for t in self._trees:
            t.update_reach_probs()

    def _compute_regrets(self, p_id):  # _CFRBase.py:146-185
        for t_idx, tree in enumerate(self._trees):
            R = self._env_bldrs[t_idx].rules.RANGE_SIZE
            ev = tree._vec("ev")  # one device -> host copy for the whole tree instead of one per node
            for node in tree.nodes():
                if node.p_id_acting_next != p_id or node.is_terminal:
                    continue
                n_act = len(node.children)
                ev_all = np.zeros(shape=(R, n_act), dtype=np.float32)
                for i, child in enumerate(node.children):
                    ev_all[:, i] = ev[child._i][p_id]
                strat_ev = np.expand_dims(ev[node._i][p_id], axis=-1).repeat(n_act, axis=-1)
                if self._iter_counter == 0:
                    node.data["regret"] = self._regret_formula_first_it(ev_all_actions=ev_all, strat_ev=strat_ev)
                else:
                    node.data["regret"] = self._regret_formula_after_first_it(ev_all_actions=ev_all, strat_ev=strat_ev,
                                                                              last_regrets=node.data["regret"])

    def _iteration_with_hooks(self):  # _CFRBase.py:122-134
        for p in range(self._n_seats):
            self._compute_cfv()
            self._compute_regrets(p_id=p)
            self._compute_new_strategy(p_id=p)
            self._update_reach_probs()
            self._add_strategy_to_average(p_id=p)
        self._iter_counter += 1
        self._compute_cfv()
        self._log_curr_strat_expl()
        self._evaluate_avg_strats()

    def _eval_avg_with_hooks(self, t_idx):
        """_CFRBase.py:218-262: a second tree filled with node.data["avg_strat"], reach, EV + best response, root exploitability"""
        if self._eval_trees is None:
            self._eval_trees = []
            for b, a in zip(self._env_bldrs, self._env_args):
                et = PublicTree(env_bldr=b, stack_size=a.starting_stack_sizes_list, stop_at_street=None, boards=self._boards, engine="levels")
                et.build_tree()
                self._eval_trees.append(et)
        et, train = self._eval_trees[t_idx], self._trees[t_idx]
        et.fill_uniform_random()
        for n_eval, n_train in zip(et.nodes(), train.nodes()):
            if n_eval.p_id_acting_next != et.CHANCE_ID and not n_eval.is_terminal:
                n_eval.strategy = np.copy(n_train.data["avg_strat"])
        et.update_reach_probs()
        et.compute_ev()
        return et.solver.exploitability()

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
        if self._host_hooks:
            return self._iteration_with_hooks()
        self._advance(1)
        self._iter_counter += 1
        self._log_curr_strat_expl()
        self._evaluate_avg_strats()

    def iterations(self, n, log=True):
        """n iterations back to back on the GPU without a host round trip per iteration; logs afterwards from the
        device-side exploitability history (average-strategy evaluation only after the last one)."""
        if self._host_hooks:
            for _ in range(n):
                self._iteration_with_hooks()
            return
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
            totals.append(self._scaled(t_idx, self._eval_avg_with_hooks(t_idx) if self._host_hooks else t.solver.eval_avg()))
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
