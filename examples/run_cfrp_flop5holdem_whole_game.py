"""CFR+ on the WHOLE Flop5Holdem game on one MI355X: the reference's own call -- CFRPlus(name, chief_handle, game_cls, agent_bet_set), no board list --
which its 1-hole-card tree code cannot serve. Here the builder deals all 2 598 960 boards as their 134 459 suit classes ("weighted
boards"; ~30 GB of HBM) and an iteration of the whole game takes ~12 ms.

    python examples/run_cfrp_flop5holdem_whole_game.py [iterations, default 200]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from pokerrl_amd.cfr.CFRPlus import CFRPlus  # noqa: E402
from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game.games import Flop5Holdem  # noqa: E402
from pokerrl_amd.rl.base_cls.workers.ChiefBase import ChiefBase  # noqa: E402

if __name__ == "__main__":
    n_iterations = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    chief = ChiefBase(t_prof=None)
    t0 = time.time()
    cfr = CFRPlus(name="FHP_WHOLE_GAME", game_cls=Flop5Holdem, agent_bet_set=bet_sets.POT_ONLY, chief_handle=chief, delay=0)
    print("built in %.1f s" % (time.time() - t0))
    t0 = time.time()
    for i in range(n_iterations):
        cfr.iteration()
        if (i + 1) % 25 == 0 or i == 0:
            vals, _ = chief.get_new_values()
            for name, graphs in sorted(vals.items()):
                if "_Avg_total_S" in name or "_Curr_S" in name:
                    for g, series in graphs.items():
                        print("iteration %4d  %-44s %s = %.4f" % (i + 1, name, g, series[-1][1]))
    print("%d whole-game iterations (each with the reference's average-strategy evaluation) in %.1f s" % (n_iterations, time.time() - t0))
