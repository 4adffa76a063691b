"""
Argument objects for the poker games; constructor signatures and attribute names follow the reference
(PokerRL/game/poker_env_args.py:4-131) so that `game_cls.ARGS_CLS(n_seats=..., starting_stack_sizes_list=...,
bet_sizes_list_as_frac_of_pot=...)` (PokerRL/cfr/_CFRBase.py:50-56) keeps working.
"""


class _PokerEnvArgs:
    N_ACTIONS = 3

    def __init__(self, n_seats, starting_stack_sizes_list=None, stack_randomization_range=(0, 0), scale_rewards=True,
                 use_simplified_headsup_obs=True, return_pre_transition_state_in_info=False, *args, **kwargs):
        self.n_seats = n_seats
        self.starting_stack_sizes_list = ([None] * n_seats if starting_stack_sizes_list is None
                                          else starting_stack_sizes_list)
        self.stack_randomization_range = stack_randomization_range
        self.scale_rewards = scale_rewards
        self.use_simplified_headsup_obs = use_simplified_headsup_obs
        self.RETURN_PRE_TRANSITION_STATE_IN_INFO = return_pre_transition_state_in_info


class NoLimitPokerEnvArgs(_PokerEnvArgs):
    def __init__(self, n_seats, *args, **kwargs):
        super().__init__(n_seats, *args, **kwargs)
        self.N_ACTIONS = 3


class LimitPokerEnvArgs(_PokerEnvArgs):
    def __init__(self, n_seats, *args, **kwargs):
        super().__init__(n_seats, *args, **kwargs)
        self.N_ACTIONS = 3


class DiscretizedPokerEnvArgs(_PokerEnvArgs):
    def __init__(self, n_seats, bet_sizes_list_as_frac_of_pot, starting_stack_sizes_list=None,
                 stack_randomization_range=(0, 0), uniform_action_interpolation=False, use_simplified_headsup_obs=True,
                 scale_rewards=True, return_pre_transition_state_in_info=False, *args, **kwargs):
        super().__init__(n_seats, starting_stack_sizes_list=starting_stack_sizes_list,
                         stack_randomization_range=stack_randomization_range, scale_rewards=scale_rewards,
                         use_simplified_headsup_obs=use_simplified_headsup_obs,
                         return_pre_transition_state_in_info=return_pre_transition_state_in_info, *args, **kwargs)
        self.bet_sizes_list_as_frac_of_pot = bet_sizes_list_as_frac_of_pot
        self.uniform_action_interpolation = uniform_action_interpolation
        self.N_ACTIONS = len(bet_sizes_list_as_frac_of_pot) + 2  # + FOLD, CHECK/CALL
