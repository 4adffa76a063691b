"""LBR plays in its own copy of the game: same rules, LBR's bet sizes (the reference's eval/lbr/_util.py)."""
from pokerrl_amd.rl import rl_util


def get_env_builder_lbr(t_prof):
    lbr_args = t_prof.module_args["lbr"]
    lbr_env_args = lbr_args.get_lbr_env_args(agents_env_args=t_prof.module_args["env"])
    builder = rl_util.get_builder_from_str(t_prof.env_builder_cls_str)
    game = rl_util.get_env_cls_from_str(t_prof.game_cls_str)
    return builder(env_cls=game, env_args=lbr_env_args)
