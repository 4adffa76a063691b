"""Shared body of the CFR examples (the reference's examples/run_cfr_example.py, run_cfrp_example.py, run_lcfr_example.py: 150
iterations on DiscretizedNLLeduc with FOLD / CHECK-CALL / POT-SIZE-RAISE). Needs an MI355X: the tree passes run on the GPU and there
is no CPU fallback. The reference plots through its Crayon wrapper; here the logged scalars stay in the chief's buffer
(ChiefBase.get_new_values) and the exploitability of the current and average strategy is printed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game.games import DiscretizedNLLeduc  # noqa: E402
from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase  # noqa: E402


def run(cfr_cls, name, n_iterations=150, **kwargs):
    chief = ChiefBase(t_prof=None)  # only used to log, as in the reference: this CFR is not distributed
    cfr = cfr_cls(name=name, game_cls=DiscretizedNLLeduc, agent_bet_set=bet_sets.POT_ONLY, chief_handle=chief, **kwargs)
    metric = "Evaluation/" + DiscretizedNLLeduc.WIN_METRIC
    for iter_id in range(n_iterations):
        cfr.iteration()  # the reference's iteration(): both seats' updates + the evaluation of the current and the average strategy
        new, _ = chief.get_new_values()
        line = ["Iteration: %3d" % iter_id]
        for tag in ("_Curr_total_averaged_", "_Avg_total_averaged_"):  # the reference's experiment names (_CFRBase.py:60-101)
            for exp_name, graphs in new.items():
                if tag in exp_name and graphs.get(metric):
                    line.append("%s %.3f %s" % ("current" if "Curr" in tag else "average", graphs[metric][-1][1], DiscretizedNLLeduc.WIN_METRIC))
        print("  ".join(line))
    return cfr
