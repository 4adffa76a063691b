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
