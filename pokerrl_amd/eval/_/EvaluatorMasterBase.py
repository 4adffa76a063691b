"""
Evaluator base: experiment naming, 95 % confidence interval and result logging to the chief, with the reference's
experiment names (PokerRL/eval/_/EvaluatorMasterBase.py:83-156) so that dashboards keep working. The ray indirection of
the reference collapses to direct calls (its local mode does exactly that, PokerRL/rl/MaybeRay.py:47-65).
"""
import numpy as np


class EvaluatorMasterBase:
    def __init__(self, t_prof, eval_env_bldr, chief_handle, eval_type, log_conf_interval=False):
        self._t_prof = t_prof
        self._eval_env_bldr = eval_env_bldr
        self._chief_handle = chief_handle
        self._chief_info = [None for _ in range(t_prof.n_seats)]
        self._is_multi_stack = len(t_prof.eval_stack_sizes) > 1
        self._log_conf_interval = log_conf_interval
        c = chief_handle.create_experiment
        pre = t_prof.name + " "
        self._exp_names_conf = None
        if log_conf_interval:  # registered before the totals, as in the reference (:88-107): chiefs number experiments in this order
            self._exp_names_conf = {
                m: [[c(pre + m + "_stack_" + str(s[0]) + ": " + eval_type + " Conf_" + b) for b in ("lower95", "upper95")]
                    for s in t_prof.eval_stack_sizes]
                for m in t_prof.eval_modes_of_algo}
        self._exp_name_total = {
            m: [c(pre + m + "_stack_" + str(s[0]) + ": " + eval_type + " Total") for s in t_prof.eval_stack_sizes]
            for m in t_prof.eval_modes_of_algo}
        if self._is_multi_stack:
            self._exp_name_multi_stack = {m: c(pre + m + "Multi_Stack" + ": " + eval_type + " Averaged Total")
                                          for m in t_prof.eval_modes_of_algo}
            if log_conf_interval:
                self._exp_names_multi_stack_conf = {
                    m: [c(pre + m + ": " + eval_type + " Conf_" + b) for b in ("lower95", "upper95")] for m in t_prof.eval_modes_of_algo}

    @property
    def is_multi_stack(self):
        return self._is_multi_stack

    def evaluate(self, iter_nr):
        raise NotImplementedError

    def update_weights(self):
        raise NotImplementedError

    def pull_current_strat_from_chief(self):
        w, self._chief_info = self._chief_handle.pull_current_eval_strategy(self._chief_info)
        return w

    def _get_95confidence(self, scores):
        scores = np.asarray(scores)
        mean, std = np.mean(scores).item(), np.std(scores).item()
        return float(mean), float(1.96 * std / np.sqrt(scores.shape[0]))

    def _graph(self):
        return "Evaluation/" + self._eval_env_bldr.env_cls.WIN_METRIC

    def _log_results(self, agent_mode, stack_size_idx, iter_nr, score, upper_conf95=None, lower_conf95=None):
        self._chief_handle.add_scalar(self._exp_name_total[agent_mode][stack_size_idx], self._graph(), iter_nr, score)
        if self._log_conf_interval:
            self._chief_handle.add_scalar(self._exp_names_conf[agent_mode][stack_size_idx][0], self._graph(), iter_nr, lower_conf95)
            self._chief_handle.add_scalar(self._exp_names_conf[agent_mode][stack_size_idx][1], self._graph(), iter_nr, upper_conf95)

    def _log_multi_stack(self, agent_mode, iter_nr, score_total, upper_conf95=None, lower_conf95=None):
        self._chief_handle.add_scalar(self._exp_name_multi_stack[agent_mode], self._graph(), iter_nr, score_total)
        if self._log_conf_interval:
            self._chief_handle.add_scalar(self._exp_names_multi_stack_conf[agent_mode][0], self._graph(), iter_nr, lower_conf95)
            self._chief_handle.add_scalar(self._exp_names_multi_stack_conf[agent_mode][1], self._graph(), iter_nr, upper_conf95)
