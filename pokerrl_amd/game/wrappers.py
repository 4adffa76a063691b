"""
Env builders: the object through which CFR / BR / agents obtain a game's rules, LUTs and argument set
(reference: PokerRL/game/_/EnvWrapperBuilderBase.py:7-111, PokerRL/game/wrappers.py:18-68). Only what the tabular hot
path reads is kept: `env_cls`, `env_args`, `rules`, `lut_holder`, `N_SEATS`, `N_ACTIONS`, `get_new_env`.
Observation-history tensors are a neural-agent concern and out of scope.
"""
import copy


class EnvWrapperBuilderBase:
    WRAPPER_CLS = None

    def __init__(self, env_cls, env_args):
        self.env_cls = env_cls
        self.env_args = env_args
        self.rules = env_cls.RULES
        self.lut_holder = env_cls.get_lut_holder()
        self.N_SEATS = env_args.n_seats
        self.N_ACTIONS = env_args.N_ACTIONS

    def get_new_env(self, is_evaluating, stack_size=None):
        from pokerrl_amd.game.poker_env import PokerEnv
        args = copy.deepcopy(self.env_args)
        if stack_size is not None:
            args.starting_stack_sizes_list = copy.deepcopy(stack_size)
        return PokerEnv(env_cls=self.env_cls, env_args=args, lut_holder=self.lut_holder, is_evaluating=is_evaluating)

    def get_new_wrapper(self, is_evaluating, init_from_env=None, stack_size=None):
        env = init_from_env if init_from_env is not None else self.get_new_env(is_evaluating, stack_size)
        return EnvWrapper(env=env, env_bldr=self)


class EnvWrapper:
    """Minimal wrapper around one env: what EvalAgentBase needs to be positioned on a public-tree node."""

    def __init__(self, env, env_bldr):
        self.env = env
        self.env_bldr = env_bldr

    def reset(self, deck_state_dict=None):
        return self.env.reset(deck_state_dict=deck_state_dict)

    def step(self, action):
        return self.env.step(action)

    def state_dict(self):
        return {"env": self.env.state_dict()}

    def load_state_dict(self, state_dict):
        self.env.load_state_dict(state_dict["env"])

    def set_to_public_tree_node_state(self, node):
        self.env.load_state_dict(node.env_state, blank_private_info=True)


class VanillaEnvBuilder(EnvWrapperBuilderBase):
    pass


class HistoryEnvBuilder(EnvWrapperBuilderBase):
    def __init__(self, env_cls, env_args, invert_history_order=False):
        super().__init__(env_cls=env_cls, env_args=env_args)
        self.invert_history_order = invert_history_order


class FlatLimitPokerEnvBuilder(EnvWrapperBuilderBase):
    pass


ALL_BUILDERS = [HistoryEnvBuilder, FlatLimitPokerEnvBuilder, VanillaEnvBuilder]
