"""
PokerRange: a float32 belief over a player's RANGE_SIZE hands (API of PokerRL/game/PokerRange.py:9-160).

Host-side NumPy bookkeeping of one [R] vector per evaluator (every operation is element-wise float32 or a NumPy sum, so
its values are the reference's by construction); the O(boards x R) consumer of ranges, LBR's check-down equity, runs on the
GPU (pokerrl_amd/eval/lbr, csrc/prl_lbr_kernels.hip) and restates the same arithmetic there.
"""
import numpy as np

from pokerrl_amd.game.Poker import Poker


class PokerRange:
    def __init__(self, env_bldr):
        if env_bldr.rules.N_HOLE_CARDS > 2:
            raise NotImplementedError("at most 2 hole cards")
        self._env_bldr = env_bldr
        self._range = None
        self.reset()

    @property
    def range(self):
        return self._range

    def get_range(self):
        return np.copy(self._range)

    def get_card_probs(self):
        """P(card is in the hand) per deck card (PokerRange.py:26-38)."""
        rules, lut = self._env_bldr.rules, self._env_bldr.lut_holder
        if rules.N_HOLE_CARDS == 1:
            return np.copy(self._range)
        out = np.zeros(rules.N_CARDS_IN_DECK, dtype=np.float32)
        for c in range(rules.N_CARDS_IN_DECK):
            out[c] = np.sum(self._range[lut.LUT_CARD_IN_WHAT_RANGE_IDXS[c]])
        return out

    def normalize(self):
        total = np.sum(self._range, axis=-1)
        if total == 0:
            self._reset_range()  # an impossible history leaves a uniform belief (PokerRange.py:45-50)
        else:
            self._range = self._range / total

    def mul_and_norm(self, mul_vector):
        self._range *= mul_vector
        self.normalize()

    def update_after_action(self, action, all_a_probs_for_all_hands):
        self._range *= all_a_probs_for_all_hands[:, action]
        self.normalize()

    def update_after_new_round(self, new_round, board_now_2d):
        self.set_cards_to_zero_prob(cards_2d=self._get_new_blockers_2d(game_round=new_round, board_2d=board_now_2d))

    def reset(self):
        self._reset_range()

    def set_cards_to_zero_prob(self, cards_2d):
        rules, lut = self._env_bldr.rules, self._env_bldr.lut_holder
        cards_1d = lut.get_1d_cards(cards_2d=cards_2d)
        if rules.N_HOLE_CARDS == 1:
            self._range[cards_1d] = 0
        else:
            n = rules.N_CARDS_IN_DECK
            for c in cards_1d:
                self._range[lut.LUT_HOLE_CARDS_2_IDX[0:c, c]] = 0
                self._range[lut.LUT_HOLE_CARDS_2_IDX[c, c + 1:n]] = 0
        self.normalize()

    @staticmethod
    def get_possible_range_idxs(rules, lut_holder, board_2d):
        idxs = np.arange(rules.RANGE_SIZE)
        if board_2d.shape[0] == 0:
            return idxs
        blocked = [int(c) for c in lut_holder.get_1d_cards(cards_2d=board_2d) if c != Poker.CARD_NOT_DEALT_TOKEN_1D]
        if rules.N_HOLE_CARDS == 1:
            return np.delete(idxs, np.array(blocked, dtype=np.int64))
        if rules.N_HOLE_CARDS == 2:
            n = rules.N_CARDS_IN_DECK
            gone = set()
            for c in blocked:
                gone.update(int(i) for i in lut_holder.LUT_HOLE_CARDS_2_IDX[0:c, c])
                gone.update(int(i) for i in lut_holder.LUT_HOLE_CARDS_2_IDX[c, c + 1:n])
            return np.delete(idxs, np.array(sorted(gone), dtype=np.int64))
        raise NotImplementedError("N_HOLE_CARDS > 2")

    @staticmethod
    def get_range_size(n_hole_cards, n_cards_in_deck):
        size = 1
        for i in range(n_hole_cards):
            size *= n_cards_in_deck - i
        return int(size / np.prod(np.arange(1, n_hole_cards + 1)))

    def load_state_dict(self, state):
        self._range = np.copy(state["range"])

    def state_dict(self):
        return {"range": np.copy(self._range)}

    def _get_new_blockers_1d(self, game_round, board_2d):
        return self._env_bldr.lut_holder.get_1d_cards(self._get_new_blockers_2d(game_round=game_round, board_2d=board_2d))

    def _get_new_blockers_2d(self, game_round, board_2d):
        lut, rules = self._env_bldr.lut_holder, self._env_bldr.rules
        hi = lut.DICT_LUT_N_CARDS_OUT[game_round]
        lo = lut.DICT_LUT_N_CARDS_OUT[rules.ROUND_BEFORE[game_round]]
        return board_2d[lo:hi].reshape(-1, 2)

    def _reset_range(self):
        r = self._env_bldr.rules.RANGE_SIZE
        self._range = np.full(shape=r, fill_value=1.0 / r, dtype=np.float32)
