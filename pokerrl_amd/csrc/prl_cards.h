// Card / hand index arithmetic shared by host and device code.
//
// Replaces the reference's binary-only lib_luts.so (call sites PokerRL/game/_/cpp_wrappers/CppLUT.py:38-47,73-94;
// layouts pinned by test/game/test_look_up_table.py:110-167):
//   card_1d = rank * n_suits + suit                      (test_look_up_table.py:110-116)
//   range_idx(c1 < c2) = lexicographic counter over all pairs (test_look_up_table.py:136-143)
#pragma once
#include "prl_defs.h"

PRL_HD PRL_INLINE int prl_card_1d(int rank, int suit, int n_suits) { return rank * n_suits + suit; }
PRL_HD PRL_INLINE int prl_card_rank(int c, int n_suits) { return c / n_suits; }
PRL_HD PRL_INLINE int prl_card_suit(int c, int n_suits) { return c % n_suits; }

// number of 2-card hands that precede (c1, *) in lexicographic order for a deck of n cards: sum_{i<c1} (n-1-i)
PRL_HD PRL_INLINE int prl_pair_row_offset(int c1, int n) { return c1 * (2 * n - c1 - 1) / 2; }

// (c1 < c2) -> range idx ; inverse of prl_hole_cards_2()
PRL_HD PRL_INLINE int prl_range_idx_2(int c1, int c2, int n) { return prl_pair_row_offset(c1, n) + (c2 - c1 - 1); }

// range idx -> (c1 < c2), closed form free: walks rows (n <= 52, <= 51 iterations)
PRL_HD PRL_INLINE void prl_hole_cards_2(int idx, int n, int* c1, int* c2) {
    int a = 0;
    int row = n - 1;
    while (idx >= row) {
        idx -= row;
        row--;
        a++;
    }
    *c1 = a;
    *c2 = a + 1 + idx;
}

PRL_HD PRL_INLINE long long prl_comb(int n, int k) {
    if (k < 0 || k > n) return 0;
    long long r = 1;
    for (int i = 1; i <= k; ++i) r = r * (n - k + i) / i;  // exact at every step
    return r;
}

// hands of a range index as 1d cards (c2 = -1 for 1-card games)
PRL_HD PRL_INLINE void prl_hand_cards(const PrlRules& r, int idx, int* c1, int* c2) {
    if (r.n_hole_cards == 1) {
        *c1 = idx;
        *c2 = -1;
    } else {
        prl_hole_cards_2(idx, r.n_cards, c1, c2);
    }
}

// Counter-based decks for the batched engines (no reference counterpart: the reference shuffles with np.random, one hand at a time): the
// first n_deal cards of hand `hand_id` depend on (seed, hand_id) only. A partial Fisher-Yates shuffle of 0 .. n_cards-1 driven by a
// SplitMix-style hash; the deck as a sparse permutation (only the touched positions are remembered). n_deal <= 16.
PRL_HD PRL_INLINE void prl_deal_hand(int n_cards, int n_deal, unsigned long long seed, unsigned long long hand_id, int8_t* out) {
    const unsigned long long idx = (hand_id + 1ull) * 0x9E3779B97F4A7C15ull + seed;
    int pos[16], val[16];
    int n_touched = 0;
    for (int d = 0; d < n_deal; ++d) {
        unsigned long long x = idx + (unsigned long long)(d + 1) * 0xBF58476D1CE4E5B9ull;
        x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull;
        x ^= x >> 27; x *= 0x94D049BB133111EBull;
        x ^= x >> 31;
        const int j = d + (int)(x % (unsigned long long)(n_cards - d));
        int a = d, b = j, ib = -1;
        for (int k = 0; k < n_touched; ++k) { if (pos[k] == d) a = val[k]; if (pos[k] == j) { b = val[k]; ib = k; } }
        // position d takes b (final: no later draw looks below d + 1, so it is not remembered), position j takes a: at most one new entry
        // per card dealt -- 16 entries cover n_deal <= 16 (remembering position d as well overran the arrays from the 9th card on)
        if (j != d) { if (ib >= 0) val[ib] = a; else { pos[n_touched] = j; val[n_touched] = a; ++n_touched; } }
        out[d] = (int8_t)b;
    }
}
