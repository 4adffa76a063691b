"""Keys of PokerEnv.state_dict(): the key strings are part of the reference's on-disk / plugin format
(PokerRL/game/PokerEnvStateDictEnums.py:9-36), so the names are fixed; every attribute is its own name except `deck`."""


def _keys(cls_name, names, renamed=()):
    table = {n: n for n in names.split()}
    table.update(dict(renamed))
    return type(cls_name, (), table)


# n_raises_this_round exists for fixed-limit games only
EnvDictIdxs = _keys(
    "EnvDictIdxs",
    "is_evaluating current_round main_pot side_pots board_2d seats current_player last_action last_raiser capped_raise "
    "n_actions_this_episode n_raises_this_round",
    renamed=[("deck", "deck_remaining")])

PlayerDictIdxs = _keys(
    "PlayerDictIdxs", "seat_id hand hand_rank stack current_bet is_allin folded_this_episode has_acted_this_round side_pot_rank")
