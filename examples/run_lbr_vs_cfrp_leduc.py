"""LBR and the exact best response against the SAME tabular agent: CFR+ on StandardLeduc, its average strategy moved into HBM as a policy table
(pokerrl_amd.rl.tabular_agent.PolicyTable), 2^18 LBR hands per seat against it on the GPU (BatchedLBR, agent kind "table"). LBR is a lower bound of
the exploitability the solver reports; the two columns show how tight it is as the strategy improves. (The reference has no tabular EvalAgent: its
LBR runs against neural agents only, eval/lbr/LocalLBRWorker.py:61-308.)"""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pokerrl_amd.cfr.CFRPlus import CFRPlus  # noqa: E402
from pokerrl_amd.eval.lbr import BatchedLBR, LBRArgs  # noqa: E402
from pokerrl_amd.game.games import StandardLeduc  # noqa: E402
from pokerrl_amd.game.wrappers import HistoryEnvBuilder  # noqa: E402
from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase  # noqa: E402
from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase  # noqa: E402
from pokerrl_amd.rl.tabular_agent import PolicyTable  # noqa: E402

if __name__ == "__main__":
    n_hands = int(os.environ.get("N_HANDS", 1 << 18))
    t_prof = TrainingProfileBase(
        name="lbr_vs_cfrp", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=StandardLeduc,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("TABLE",), eval_stack_sizes=None,
        module_args={"env": StandardLeduc.ARGS_CLS(n_seats=2), "lbr": LBRArgs(n_lbr_hands_per_seat=n_hands, lbr_check_to_round=None)},
        path_data=tempfile.mkdtemp())
    cfr = CFRPlus(name="lbr_vs_cfrp", chief_handle=ChiefBase(t_prof=t_prof), game_cls=StandardLeduc, agent_bet_set=None, delay=0)
    cfr.reset()
    done = 0
    print("%10s %32s %34s %20s" % ("iterations", "exploitability (exact BR) mA/g", "LBR winnings mA/g (+- 95 %)", "LBR M hands/s (GPU)"))
    for upto in [int(x) for x in os.environ.get("ITERATIONS", "1,10,100,1000").split(",")]:
        cfr.iterations(upto - done, log=False)
        done = upto
        expl = cfr._scaled(0, cfr._trees[0].solver.eval_avg())
        table = PolicyTable.from_cfr(cfr)
        lbr = BatchedLBR(t_prof, agent_kind="table", agent_seed=7, table=table)
        ws, ms = [], 0.0
        for seat in (0, 1):
            ws.append(lbr.run(agent_seat_id=seat, n_hands=n_hands, deck_seed=upto, first_hand=seat * n_hands, episode_base=seat * n_hands))
            ms += lbr.last_stats["device_ms"]
        w = np.concatenate(ws).astype(np.float64)
        table.close()
        print("%10d %32.2f %24.2f +- %-7.2f %20.2f" % (upto, expl, w.mean(), 1.96 * w.std() / np.sqrt(w.size), 2 * n_hands / ms / 1e3))
