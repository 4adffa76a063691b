"""
TEST INFRASTRUCTURE ONLY (never imported by pokerrl_amd): NumPy restatement of LBR's check-down equity.

Follows PokerRL/eval/lbr/LocalLBRWorker.py:379-512 (_LBRRolloutManager.__init__ / _build_eq_vecs :381-425,
get_lbr_checkdown_equity :427-468, _calc_eq :470-512) and the PokerRange operations it uses (PokerRange.py:26-84), written
against plain arrays instead of env / range objects. Pinned to the reference by tests/golden/lbr_equity.npz (outputs of the
reference's own rollout manager at every stage from the flop on), tests/golden/lbr_equity_preflop.npz (the same before the flop: five cards to
come, 2 118 760 run-outs, three ranges) and, end to end, by tests/golden/lbr_*.npz.
"""
import numpy as np


def _hands(n_hole, n_cards):
    if n_hole == 1:
        return [(c,) for c in range(n_cards)]
    return [(c1, c2) for c1 in range(n_cards) for c2 in range(c1 + 1, n_cards)]


def _zero_cards_and_normalize(rng, hands, cards):
    """PokerRange.set_cards_to_zero_prob + normalize (PokerRange.py:45-50, :67-84)"""
    r = rng.copy()
    for i, h in enumerate(hands):
        if any(c in h for c in cards):
            r[i] = 0
    s = np.sum(r, axis=-1)
    if s == 0:
        return np.full(r.shape[0], 1.0 / r.shape[0], dtype=np.float32)
    return r / s


def _holds_table(hands, n_cards):
    """[n_cards][R] bool: hand h holds card c (what set_cards_to_zero_prob looks up in LUT_CARD_IN_WHAT_RANGE_IDXS)"""
    t = np.zeros((n_cards, len(hands)), bool)
    for i, h in enumerate(hands):
        for c in h:
            t[c, i] = True
    return t


def checkdown_equity(rank_fn, n_hole, n_cards, n_board_total, board_dealt, lbr_hand, agent_range):
    """rank_fn(full_board) -> int32 [R] hand ranks (hold'em: -1 for hands sharing a card with the board).
    Returns the float32 scalar the reference's get_lbr_checkdown_equity returns."""
    hands = _hands(n_hole, n_cards)
    agent_range = np.asarray(agent_range, dtype=np.float32)
    lbr_hand = tuple(sorted(int(c) for c in lbr_hand))
    lbr_idx = hands.index(lbr_hand)
    dealt = [int(c) for c in board_dealt]
    n_to_deal = n_board_total - len(dealt)
    possible = [c for c in range(n_cards) if c not in dealt and c not in lbr_hand]

    # each board once, the cards to come ascending (:408-417: nested loops over the remaining cards = lexicographic combinations)
    import itertools
    first_board = next(itertools.combinations(possible, n_to_deal))
    # QUIRK (:470, :509-510): the board counter is never advanced -> the index lists of the first board serve every board
    ranks0 = np.asarray(rank_fn(dealt + list(first_board)))
    bigger = np.argwhere(ranks0 < ranks0[lbr_idx])
    equal = np.argwhere(ranks0 == ranks0[lbr_idx])

    # get_card_probs (PokerRange.py:26-38) -> 1 - p, LBR's cards and the board zeroed, normalised if positive (:432-449)
    if n_hole == 1:
        acp = agent_range.copy()
    else:
        acp = np.zeros(n_cards, dtype=np.float32)
        for c in range(n_cards):
            acp[c] = np.sum(agent_range[[i for i, h in enumerate(hands) if c in h]])
    cp = np.subtract(1, acp)
    cp[list(lbr_hand)] = 0.0
    if dealt:
        cp[dealt] = 0.0
    if np.sum(cp) > 0:
        cp /= np.sum(cp)

    win = [0.0]
    holds = _holds_table(hands, n_cards)
    base_blocked = np.zeros(len(hands), bool)
    for c in dealt:
        base_blocked |= holds[c]

    def zero_and_normalize(prefix):
        """_zero_cards_and_normalize(agent_range, hands, dealt + prefix) with the membership test as a table look-up (the same array, so the
        same NumPy sums): what makes the 2.1 M run-outs of a hold'em pre-flop decision a matter of minutes"""
        blocked = base_blocked.copy()
        for c in prefix:
            blocked |= holds[c]
        r = np.where(blocked, np.float32(0), agent_range)
        s_ = np.sum(r, axis=-1)
        if s_ == 0:
            return np.full(r.shape[0], 1.0 / r.shape[0], dtype=np.float32)
        return r / s_

    def rec(prefix, left, probs, poss, reach):
        if left > 0:
            for i in range(len(poss) - (left - 1)):
                nxt = np.copy(probs)
                nxt[poss[i]] = 0.0
                with np.errstate(all="ignore"):
                    nxt /= np.sum(nxt)
                rec(prefix + [poss[i]], left - 1, nxt, poss[i + 1:], reach * probs[poss[i]])
        else:
            r = zero_and_normalize(prefix) if n_to_deal > 2 else _zero_cards_and_normalize(agent_range, hands, dealt + prefix)
            eq = np.sum(r[bigger])
            eq += np.sum(r[equal]) / 2.0
            win[0] += eq * reach

    rec([], n_to_deal, cp, possible, 1.0)
    fact = 1
    for m in range(1, n_to_deal + 1):
        fact *= m
    return np.float32(win[0] * fact)
