"""Vanilla CFR (reference: PokerRL/cfr/VanillaCFR.py:9-77)."""
from pokerrl_amd.cfr._CFRBase import CFRBase as _CFRBase


class VanillaCFR(_CFRBase):
    _VARIANT = "vanilla"

    def __init__(self, name, chief_handle, game_cls, agent_bet_set, starting_stack_sizes=None, **kw):
        super().__init__(name=name, chief_handle=chief_handle, game_cls=game_cls, starting_stack_sizes=starting_stack_sizes,
                         agent_bet_set=agent_bet_set, algo_name="CFR", **kw)
        self.reset()
