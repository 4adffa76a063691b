"""The evaluators pointed at the solver's own output at hold'em size: CFR+ on the WHOLE Flop5Holdem game (the reference's default arguments: every board;
all 2 598 960 of them through their 134 459 suit classes, 30 GB on one MI355X), its average strategy moved into a policy table in HBM without touching the
host (PolicyTable.from_solver: 806 756 rows, 12.8 GB, suit-canonical look-ups), then on the GPU: LBR against it (BatchedLBR; LBR_FROM_THE_START=1:
lbr_check_to_round = None, pre-flop equities cached per public history and LBR hand -- about a minute more) and its self-play value (BatchedHead2Head),
beside the exact exploitability the solver computes. What the reference's LocalLBRMaster / LocalHead2HeadMaster / LocalBRMaster do for a trained agent
(eval/lbr/LocalLBRWorker.py:61-308, eval/head_to_head/LocalHead2HeadMaster.py:82-126, eval/br/LocalBRMaster.py:67-80), on a game they cannot hold."""
import os
import sys
import tempfile

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pokerrl_amd.cfr.CFRPlus import CFRPlus  # noqa: E402
from pokerrl_amd.eval.head_to_head import BatchedHead2Head, H2HArgs  # noqa: E402
from pokerrl_amd.eval.lbr import BatchedLBR, LBRArgs  # noqa: E402
from pokerrl_amd.game.games import Flop5Holdem  # noqa: E402
from pokerrl_amd.game.Poker import Poker  # noqa: E402
from pokerrl_amd.game.wrappers import HistoryEnvBuilder  # noqa: E402
from pokerrl_amd.rl.base_cls.TrainingProfileBase import TrainingProfileBase  # noqa: E402
from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase  # noqa: E402
from pokerrl_amd.rl.tabular_agent import PolicyTable  # noqa: E402

if __name__ == "__main__":
    n = int(os.environ.get("N_HANDS", 1 << 19))
    from_start = bool(int(os.environ.get("LBR_FROM_THE_START", "0")))
    t_prof = TrainingProfileBase(
        name="whole_fhp", log_verbose=False, log_export_freq=1, checkpoint_freq=10 ** 9, eval_agent_export_freq=10 ** 9, game_cls=Flop5Holdem,
        env_bldr_cls=HistoryEnvBuilder, start_chips=None, eval_modes_of_algo=("TABLE",), eval_stack_sizes=None,
        module_args={"env": Flop5Holdem.ARGS_CLS(n_seats=2), "lbr": LBRArgs(n_lbr_hands_per_seat=n, lbr_check_to_round=None if from_start else Poker.FLOP),
                     "h2h": H2HArgs(n_hands=n)}, path_data=tempfile.mkdtemp())
    cfr = CFRPlus(name="whole_fhp", chief_handle=ChiefBase(t_prof=t_prof), game_cls=Flop5Holdem, agent_bet_set=None, delay=0)  # every board: the suit classes
    cfr.reset()
    done = 0
    print("%10s %30s %34s %30s" % ("iterations", "exploitability mbb/g (seats)", "LBR winnings mbb/g per LBR seat", "value of seat 0 (self-play)"))
    for upto in [int(x) for x in os.environ.get("ITERATIONS", "10,100,300").split(",")]:
        cfr.iterations(upto - done, log=False)
        done = upto
        expl = cfr._trees[0].solver.eval_avg().astype(np.float64) * float(Flop5Holdem.EV_NORMALIZER)
        table = PolicyTable.from_solver(cfr)
        lbr = BatchedLBR(t_prof, agent_kind="table", agent_seed=7, table=table)
        w = [lbr.run(agent_seat_id=s, n_hands=n, deck_seed=upto, first_hand=s * n, episode_base=s * n).astype(np.float64) for s in (0, 1)]  # w[s]: LBR sits in seat 1 - s
        v = BatchedHead2Head(t_prof, kinds=("table", "table"), seeds=(11, 12), tables=(table, table)).play(n_hands=n, deck_seed=upto + 1).astype(np.float64)
        v0 = 0.5 * (v[:n].mean() - v[n:].mean())
        ci = lambda x: 1.96 * x.std() / np.sqrt(x.size)  # noqa: E731
        print("%10d %14.2f %14.2f   seat 0: %8.2f +- %-6.2f seat 1: %8.2f +- %-6.2f %14.2f +- %.2f"
              % (upto, expl[0], expl[1], w[1].mean(), ci(w[1]), w[0].mean(), ci(w[0]), v0, 0.5 * np.hypot(ci(v[:n]), ci(v[n:]))))
        table.close()
