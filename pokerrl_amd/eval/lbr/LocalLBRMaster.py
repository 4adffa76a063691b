"""LocalLBRMaster (PokerRL/eval/lbr/LocalLBRMaster.py:13-90): runs LBR for every seat / mode / stack size through its
workers (plain objects here: ray orchestration is out of scope) and logs mean +- 95% confidence."""
import numpy as np

from pokerrl_amd.eval._.EvaluatorMasterBase import EvaluatorMasterBase
from pokerrl_amd.eval.lbr import _util


class LocalLBRMaster(EvaluatorMasterBase):
    def __init__(self, t_prof, chief_handle):
        assert t_prof.n_seats == 2
        EvaluatorMasterBase.__init__(self, t_prof=t_prof, eval_env_bldr=_util.get_env_builder_lbr(t_prof=t_prof),
                                     chief_handle=chief_handle, eval_type="LBR", log_conf_interval=True)
        self.lbr_args = t_prof.module_args["lbr"]
        self.weights_for_eval_agent = None
        self.alive_worker_handles = None

    def set_worker_handles(self, *worker_handles):
        self.alive_worker_handles = list(worker_handles)

    def evaluate(self, iter_nr):
        for worker in self.alive_worker_handles:
            worker.update_weights(self.weights_for_eval_agent)
        n_per_worker = int(self.lbr_args.n_lbr_hands / self.lbr_args.n_workers)
        for mode in self._t_prof.eval_modes_of_algo:
            totals = []
            for stack_size_idx, stack_size in enumerate(self._t_prof.eval_stack_sizes):
                scores = []
                for p_id in range(self._t_prof.n_seats):
                    scores += [worker.run(p_id, n_per_worker, mode, stack_size) for worker in self.alive_worker_handles]
                scores = [s for s in scores if s is not None]
                if not scores:
                    continue
                scores = np.concatenate(scores, axis=0)
                if len(scores) > 0:
                    mean, d = self._get_95confidence(scores)
                    self._log_results(iter_nr=iter_nr, agent_mode=mode, stack_size_idx=stack_size_idx, score=mean,
                                      upper_conf95=mean + d, lower_conf95=mean - d)
                    totals.append((mean, d))
            if self.is_multi_stack and totals:
                m = sum(t[0] for t in totals) / float(len(totals))
                d = sum(t[1] for t in totals) / float(len(totals))
                self._log_multi_stack(agent_mode=mode, iter_nr=iter_nr, score_total=m, upper_conf95=m + d, lower_conf95=m - d)

    def update_weights(self):
        self.weights_for_eval_agent = self.pull_current_strat_from_chief()
