"""
Rule sets (deck, rounds, hand ranking) of the supported games. Attribute names follow the reference
(PokerRL/game/_/rl_env/game_rules.py:15-312) because callers read them directly (`env_bldr.rules.RANGE_SIZE`,
`rules.N_CARDS_IN_DECK`, ... e.g. ValueFiller.py:19, StrategyFiller.py:166). Hand ranking goes through the native
library; `to_native()` produces the POD the C ABI takes.
"""
from math import comb

import numpy as np

from pokerrl_amd import _native
from pokerrl_amd.game.Poker import Poker

RANK_RULE_LEDUC, RANK_RULE_BIG_LEDUC, RANK_RULE_HOLDEM52 = 0, 1, 2


class _RulesBase:
    N_HOLE_CARDS = None
    N_RANKS = None
    N_SUITS = None
    BTN_IS_FIRST_POSTFLOP = None
    N_FLOP_CARDS = N_TURN_CARDS = N_RIVER_CARDS = 0
    ALL_ROUNDS_LIST = None
    SUITS_MATTER = None
    STRING = None
    _RANK_RULE = None

    @classmethod
    def to_native(cls):
        r = _native.PrlRules()
        r.n_hole_cards, r.n_ranks, r.n_suits = cls.N_HOLE_CARDS, cls.N_RANKS, cls.N_SUITS
        r.n_cards = cls.N_CARDS_IN_DECK
        r.range_size = cls.RANGE_SIZE
        r.n_rounds = len(cls.ALL_ROUNDS_LIST)
        per_round = [0, cls.N_FLOP_CARDS, cls.N_TURN_CARDS, cls.N_RIVER_CARDS]
        for i in range(4):
            r.board_cards_in_round[i] = per_round[i]
        r.n_board_cards = cls.N_TOTAL_BOARD_CARDS
        r.btn_first_postflop = int(cls.BTN_IS_FIRST_POSTFLOP)
        r.rank_rule = cls._RANK_RULE
        return r

    @classmethod
    def get_lut_holder(cls):
        from pokerrl_amd.game.look_up_table import LutHolder
        return LutHolder(cls)


def _finish(cls):
    cls.N_CARDS_IN_DECK = cls.N_RANKS * cls.N_SUITS
    cls.RANGE_SIZE = comb(cls.N_CARDS_IN_DECK, cls.N_HOLE_CARDS)
    cls.N_TOTAL_BOARD_CARDS = cls.N_FLOP_CARDS + cls.N_TURN_CARDS + cls.N_RIVER_CARDS
    return cls


_TWO_ROUNDS_BEFORE = {Poker.PREFLOP: Poker.PREFLOP, Poker.FLOP: Poker.PREFLOP}
_TWO_ROUNDS_AFTER = {Poker.PREFLOP: Poker.FLOP, Poker.FLOP: None}
_HOLDEM_RANKS = {Poker.CARD_NOT_DEALT_TOKEN_1D: "", **{i: "23456789TJQKA"[i] for i in range(13)}}
_HOLDEM_SUITS = {Poker.CARD_NOT_DEALT_TOKEN_1D: "", 0: "h", 1: "d", 2: "s", 3: "c"}


class _OneCardRules(_RulesBase):
    """Leduc family: one hole card, one board card; a pair with the board beats everything (game_rules.py:68-75)."""
    N_HOLE_CARDS = 1
    N_SUITS = 2
    BTN_IS_FIRST_POSTFLOP = True
    N_FLOP_CARDS = 1
    ALL_ROUNDS_LIST = [Poker.PREFLOP, Poker.FLOP]
    SUITS_MATTER = False
    ROUND_BEFORE = _TWO_ROUNDS_BEFORE
    ROUND_AFTER = _TWO_ROUNDS_AFTER
    _PAIR_BONUS = None

    def get_hand_rank(self, hand_2d, board_2d):
        hr = int(hand_2d[0, 0])
        return self._PAIR_BONUS + hr if int(board_2d[0, 0]) == hr else hr

    def get_hand_rank_all_hands_on_given_boards(self, boards_1d, lut_holder):
        ranks = np.arange(self.RANGE_SIZE, dtype=np.int32)[None, :] // self.N_SUITS
        board_ranks = np.asarray(boards_1d, dtype=np.int32)[:, :1] // self.N_SUITS
        out = np.where(ranks == board_ranks, self._PAIR_BONUS + ranks, ranks).astype(np.int32)
        return out


@_finish
class LeducRules(_OneCardRules):
    N_RANKS = 3
    STRING = "LEDUC_RULES"
    _RANK_RULE = RANK_RULE_LEDUC
    _PAIR_BONUS = 100
    RANK_DICT = {i: str(i + 2) for i in range(N_RANKS)}
    SUIT_DICT = {0: "a", 1: "b"}


@_finish
class BigLeducRules(_OneCardRules):
    N_RANKS = 12
    STRING = "BIG_LEDUC_RULES"
    _RANK_RULE = RANK_RULE_BIG_LEDUC
    _PAIR_BONUS = 10000
    RANK_DICT = {i: str(i + 2) for i in range(N_RANKS)}
    SUIT_DICT = {0: "a", 1: "b"}


class _Deck52Rules(_RulesBase):
    N_HOLE_CARDS = 2
    N_RANKS = 13
    N_SUITS = 4
    BTN_IS_FIRST_POSTFLOP = False
    SUITS_MATTER = True
    RANK_DICT = _HOLDEM_RANKS
    SUIT_DICT = _HOLDEM_SUITS
    _RANK_RULE = RANK_RULE_HOLDEM52

    def get_hand_rank(self, hand_2d, board_2d):
        """best-5-of-7 strength, higher is better (replaces CppHandeval.get_hand_rank_52_holdem)"""
        board_1d = (np.asarray(board_2d, dtype=np.int32)[:, 0] * 4 + np.asarray(board_2d, dtype=np.int32)[:, 1])
        h = np.asarray(hand_2d, dtype=np.int32)
        return _native.hand_rank_7(board_1d.astype(np.int8), h[0, 0] * 4 + h[0, 1], h[1, 0] * 4 + h[1, 1])

    def get_hand_rank_all_hands_on_given_boards(self, boards_1d, lut_holder):
        """[N, 5] boards -> int32 [N, 1326], -1 for blocked hands; GPU (replaces CppHandeval.py:45-65)"""
        return _native.hand_rank_boards(boards_1d)


@_finish
class HoldemRules(_Deck52Rules):
    N_FLOP_CARDS, N_TURN_CARDS, N_RIVER_CARDS = 3, 1, 1
    ALL_ROUNDS_LIST = [Poker.PREFLOP, Poker.FLOP, Poker.TURN, Poker.RIVER]
    ROUND_BEFORE = {Poker.PREFLOP: Poker.PREFLOP, Poker.FLOP: Poker.PREFLOP, Poker.TURN: Poker.FLOP,
                    Poker.RIVER: Poker.TURN}
    ROUND_AFTER = {Poker.PREFLOP: Poker.FLOP, Poker.FLOP: Poker.TURN, Poker.TURN: Poker.RIVER, Poker.RIVER: None}
    STRING = "HOLDEM_RULES"


@_finish
class FlopHoldemRules(_Deck52Rules):
    """Two rounds; all five board cards arrive on the "flop" (game_rules.py:232-312, N_FLOP_CARDS = 5)."""
    N_FLOP_CARDS = 5
    ALL_ROUNDS_LIST = [Poker.PREFLOP, Poker.FLOP]
    ROUND_BEFORE = {Poker.PREFLOP: Poker.PREFLOP, Poker.FLOP: Poker.PREFLOP, Poker.TURN: None, Poker.RIVER: None}
    ROUND_AFTER = {Poker.PREFLOP: Poker.FLOP, Poker.FLOP: None, Poker.TURN: None, Poker.RIVER: None}
    STRING = "FLOP_HOLDEM_RULES"
