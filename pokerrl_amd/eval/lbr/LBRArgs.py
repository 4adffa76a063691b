"""LBR arguments (PokerRL/eval/lbr/LBRArgs.py:11-75): same constructor, same attributes."""
import copy

from pokerrl_amd.game import bet_sets
from pokerrl_amd.game.poker_env_args import DiscretizedPokerEnvArgs, LimitPokerEnvArgs, NoLimitPokerEnvArgs


class LBRArgs:
    def __init__(self, lbr_bet_set=bet_sets.OFF_TREE_11, n_lbr_hands_per_seat=30000, lbr_check_to_round=None,
                 n_parallel_lbr_workers=10, use_gpu_for_batch_eval=True, DISTRIBUTED=False):
        self.lbr_bet_set = lbr_bet_set
        self.n_lbr_hands = n_lbr_hands_per_seat
        self.lbr_check_to_round = lbr_check_to_round  # Poker.TURN is recommended for 4-round games
        self.n_workers = n_parallel_lbr_workers if DISTRIBUTED else 1
        self.use_gpu_for_batch_eval = use_gpu_for_batch_eval
        self.DISTRIBUTED = DISTRIBUTED

    def get_lbr_env_args(self, agents_env_args):
        """LBR plays in its own env: the agent's game with LBR's bet set, no stack randomisation (LBRArgs.py:49-75)."""
        cls = type(agents_env_args)
        common = dict(n_seats=agents_env_args.n_seats, starting_stack_sizes_list=copy.deepcopy(agents_env_args.starting_stack_sizes_list),
                      stack_randomization_range=(0, 0), use_simplified_headsup_obs=agents_env_args.use_simplified_headsup_obs,
                      uniform_action_interpolation=False)
        if cls is DiscretizedPokerEnvArgs:
            return DiscretizedPokerEnvArgs(bet_sizes_list_as_frac_of_pot=copy.deepcopy(self.lbr_bet_set), **common)
        if cls is LimitPokerEnvArgs:
            return LimitPokerEnvArgs(**common)
        if cls is NoLimitPokerEnvArgs:
            raise NotImplementedError("Currently not supported")
        raise TypeError(cls)
