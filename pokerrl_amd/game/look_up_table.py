"""
Card / hand index look-up tables. Same attribute and method names as the reference's LutHolder classes
(PokerRL/game/_/look_up_table.py:191-322); the index tables come from the native library (replacement of lib_luts.so),
layouts pinned by tests/golden/luts.npz.
"""
from math import comb

import numpy as np

from pokerrl_amd import _native
from pokerrl_amd.game.Poker import Poker


class LutHolder:
    def __init__(self, env_or_rules_cls):
        rules = getattr(env_or_rules_cls, "RULES", None) or env_or_rules_cls
        self.rules = rules
        nr = rules.to_native()
        n, s = rules.N_CARDS_IN_DECK, rules.N_SUITS

        cards = np.arange(n)
        self.LUT_1DCARD_2_2DCARD = np.stack([cards // s, cards % s], axis=1).astype(np.int8)   # [c] -> (rank, suit)
        self.LUT_2DCARD_2_1DCARD = cards.reshape(rules.N_RANKS, s).astype(np.int8)              # [rank, suit] -> c
        self.LUT_IDX_2_HOLE_CARDS = _native.lut_idx_2_hole_cards(nr)
        self.LUT_HOLE_CARDS_2_IDX = _native.lut_hole_cards_2_idx(nr)
        self.LUT_CARD_IN_WHAT_RANGE_IDXS = _native.lut_card_in_what_range_idxs(nr)
        if rules.N_HOLE_CARDS == 1:  # the reference builds these with np.arange (int64), look_up_table.py:149-158
            self.LUT_IDX_2_HOLE_CARDS = self.LUT_IDX_2_HOLE_CARDS.astype(np.int64)
            self.LUT_HOLE_CARDS_2_IDX = self.LUT_HOLE_CARDS_2_IDX.astype(np.int64)
            self.LUT_CARD_IN_WHAT_RANGE_IDXS = self.LUT_CARD_IN_WHAT_RANGE_IDXS.astype(np.int64)
        self.LUT_RANGE_IDX_TO_PRIVATE_OBS = self._private_obs_lut()

        dealt = {Poker.PREFLOP: 0, Poker.FLOP: rules.N_FLOP_CARDS, Poker.TURN: rules.N_TURN_CARDS,
                 Poker.RIVER: rules.N_RIVER_CARDS}
        self.DICT_LUT_CARDS_DEALT_IN_TRANSITION_TO = dealt
        self.DICT_LUT_N_CARDS_OUT = {
            Poker.PREFLOP: 0, Poker.FLOP: rules.N_FLOP_CARDS, Poker.TURN: rules.N_FLOP_CARDS + rules.N_TURN_CARDS,
            Poker.RIVER: rules.N_FLOP_CARDS + rules.N_TURN_CARDS + rules.N_RIVER_CARDS}
        # number of boards counted by the cards dealt IN the transition (reference quirk, look_up_table.py:55-60)
        self.DICT_LUT_N_BOARDS = {r: comb(n, dealt[r]) for r in rules.ALL_ROUNDS_LIST}
        self.DICT_LUT_N_BOARD_BRANCHES = {Poker.PREFLOP: 0}
        for r in rules.ALL_ROUNDS_LIST:
            if r != Poker.PREFLOP:
                left = n - self.DICT_LUT_N_CARDS_OUT[rules.ROUND_BEFORE[r]] - rules.N_HOLE_CARDS
                self.DICT_LUT_N_BOARD_BRANCHES[r] = comb(left, dealt[r])

    def _private_obs_lut(self):
        r = self.rules
        d = r.N_SUITS + r.N_RANKS
        lut = np.zeros((r.RANGE_SIZE, d * r.N_HOLE_CARDS), dtype=np.float32)
        for k in range(r.N_HOLE_CARDS):
            c2d = self.LUT_1DCARD_2_2DCARD[self.LUT_IDX_2_HOLE_CARDS[:, k]]
            lut[np.arange(r.RANGE_SIZE), d * k + c2d[:, 0]] = 1
            if r.SUITS_MATTER:
                lut[np.arange(r.RANGE_SIZE), d * k + r.N_RANKS + c2d[:, 1]] = 1
        return lut

    # ---- card conversions (not-dealt token aware) --------------------------------------------------------------------
    def get_1d_card(self, card_2d):
        if card_2d[0] == Poker.CARD_NOT_DEALT_TOKEN_1D:
            return Poker.CARD_NOT_DEALT_TOKEN_1D
        return self.LUT_2DCARD_2_1DCARD[card_2d[0], card_2d[1]]

    def get_1d_cards(self, cards_2d):
        cards_2d = np.asarray(cards_2d)
        if cards_2d.ndim == 0 or cards_2d.shape[0] == 0:
            return np.array([], dtype=np.int8)
        dealt = cards_2d[:, 0] != Poker.CARD_NOT_DEALT_TOKEN_1D
        safe = np.where(dealt[:, None], cards_2d, 0)
        return np.where(dealt, self.LUT_2DCARD_2_1DCARD[safe[:, 0], safe[:, 1]], Poker.CARD_NOT_DEALT_TOKEN_1D)

    def get_2d_cards(self, cards_1d):
        cards_1d = np.asarray(cards_1d)
        if cards_1d.ndim == 0 or cards_1d.shape[0] == 0:
            return np.array([], dtype=np.int8)
        dealt = cards_1d != Poker.CARD_NOT_DEALT_TOKEN_1D
        out = self.LUT_1DCARD_2_2DCARD[np.where(dealt, cards_1d, 0)].reshape(-1, 2).copy()
        out[~dealt] = Poker.CARD_NOT_DEALT_TOKEN_1D
        return out

    # ---- range index <-> hole cards ----------------------------------------------------------------------------------
    def get_range_idx_from_hole_cards(self, hole_cards_2d):
        c = sorted(int(self.LUT_2DCARD_2_1DCARD[h[0], h[1]]) for h in hole_cards_2d)
        if self.rules.N_HOLE_CARDS == 1:
            return self.LUT_HOLE_CARDS_2_IDX[c[0], 0]
        return self.LUT_HOLE_CARDS_2_IDX[c[0], c[1]]

    def get_2d_hole_cards_from_range_idx(self, range_idx):
        return self.LUT_1DCARD_2_2DCARD[self.LUT_IDX_2_HOLE_CARDS[range_idx]].astype(np.int8).reshape(-1, 2)

    def get_1d_hole_cards_from_range_idx(self, range_idx):
        return np.copy(self.LUT_IDX_2_HOLE_CARDS[range_idx])


# the reference exposes two holder classes; both are the same object here
LutHolderLeduc = LutHolder
LutHolderHoldem = LutHolder
