"""String -> class registries the hot path imports (reference: PokerRL/rl/rl_util.py:75-91)."""
from pokerrl_amd.game.games import ALL_ENVS
from pokerrl_amd.game.wrappers import ALL_BUILDERS


def get_env_cls_from_str(env_str):
    for e in ALL_ENVS:
        if env_str == e.__name__:
            return e
    raise ValueError(env_str, "is not registered or does not exist.")


def get_builder_from_str(wrapper_str):
    for b in ALL_BUILDERS:
        if wrapper_str == b.__name__:
            return b
    raise ValueError(wrapper_str, "is not registered or does not exist.")


def get_env_builder(t_prof):
    builder = get_builder_from_str(t_prof.env_builder_cls_str)
    return builder(env_cls=get_env_cls_from_str(t_prof.game_cls_str), env_args=t_prof.module_args["env"])
