"""CFR+ with linear averaging delay (reference: PokerRL/cfr/CFRPlus.py:9-87)."""
from pokerrl_amd.cfr._CFRBase import CFRBase as _CFRBase


class CFRPlus(_CFRBase):
    _VARIANT = "plus"

    def __init__(self, name, chief_handle, game_cls, agent_bet_set, starting_stack_sizes=None, delay=0, **kw):
        super().__init__(name=name, chief_handle=chief_handle, game_cls=game_cls, starting_stack_sizes=starting_stack_sizes,
                         agent_bet_set=agent_bet_set, algo_name="CFRp_delay" + str(delay), delay=delay, **kw)
        self.delay = delay
        self.reset()

    def _evaluate_avg_strats(self):  # CFRPlus.py:33-35
        if self._iter_counter > self.delay:
            return super()._evaluate_avg_strats()
