"""Linear CFR (reference: PokerRL/cfr/LinearCFR.py:9-76)."""
from pokerrl_amd.cfr._CFRBase import CFRBase as _CFRBase


class LinearCFR(_CFRBase):
    _VARIANT = "linear"

    def __init__(self, name, chief_handle, game_cls, agent_bet_set, starting_stack_sizes=None, **kw):
        super().__init__(name=name, chief_handle=chief_handle, game_cls=game_cls, starting_stack_sizes=starting_stack_sizes,
                         agent_bet_set=agent_bet_set, algo_name="LinCFR", **kw)
        self.reset()
