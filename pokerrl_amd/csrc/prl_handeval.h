// 7-card (5 board + 2 hole) hand-strength evaluator for the 52-card deck -- bit-exact replacement of the reference's
// binary-only evaluator lib_hand_eval.so (call sites PokerRL/game/_/cpp_wrappers/CppHandeval.py:34-65, used by
// game_rules.py:213-223,296-306). No source of that library exists; the encoding below was recovered by the survey
//  from known answers and is pinned by tests/golden/handrank_*.npz captured from the binary:
//
//   rank = BASE[category] + sum_i k_i * 13^(n-1-i)     (card ranks 2..A -> 0..12, higher rank value = better hand)
//
// including the binary's one quirk: the quads kicker is the NEIGHBOUR of the four-of-a-kind in the rank-sorted seven
// cards (lowest rank above the quads if any card outranks them, otherwise the best remaining card), not the best kicker.
//
// Formulation here (MI355X-first, branch-light integer ALU): four 13-bit per-suit rank masks, per-rank multiplicities by
// a carry-save add of the masks, categories by popcount / leading-bit arithmetic. No tables, no memory traffic beyond
// the seven cards -- one evaluation is ~100 integer ops, so a (board, hand) batch is bound by the 4-byte result store.
#pragma once
#include "prl_defs.h"

#define PRL_HR_BASE_PAIR 576011
#define PRL_HR_BASE_TWO_PAIR 658508
#define PRL_HR_BASE_TRIPS 661446
#define PRL_HR_BASE_STRAIGHT 664384
#define PRL_HR_BASE_FLUSH 664398
#define PRL_HR_BASE_FULL_HOUSE 1240409
#define PRL_HR_BASE_QUADS 1240618
#define PRL_HR_BASE_STRAIGHT_FLUSH 1240827

PRL_HD PRL_INLINE int prl_popc(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __popc(x);
#else
    return __builtin_popcount(x);
#endif
}
// index of highest set bit (x != 0)
PRL_HD PRL_INLINE int prl_msb(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return 31 - __clz((int)x);
#else
    return 31 - __builtin_clz(x);
#endif
}
PRL_HD PRL_INLINE int prl_lsb(uint32_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __ffs((int)x) - 1;
#else
    return __builtin_ctz(x);
#endif
}

// top card (rank index) of the best straight in a 13-bit rank mask, or -1. Wheel (A2345) has top card 3.
PRL_HD PRL_INLINE int prl_straight_top(uint32_t m) {
    uint32_t m2 = (m << 1) | ((m >> 12) & 1u);  // bit 0 = ace-low, bit r+1 = rank r
    uint32_t s = m2 & (m2 >> 1) & (m2 >> 2) & (m2 >> 3) & (m2 >> 4);
    if (!s) return -1;
    return prl_msb(s) + 3;  // low end index in m2 coordinates + 4 cards up - 1 (m2 shift)
}

// pack the n highest set bits of m, high to low, base 13
PRL_HD PRL_INLINE int prl_pack_top(uint32_t m, int n) {
    int v = 0;
    for (int i = 0; i < n; ++i) {
        int b = prl_msb(m);
        v = v * 13 + b;
        m &= ~(1u << b);
    }
    return v;
}

// s0..s3: 13-bit rank masks of the seven cards, one per suit.
PRL_HD PRL_INLINE int32_t prl_rank7_masks(uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3) {
    // ---- flush family ------------------------------------------------------------------------------------------------
    uint32_t fm = 0;
    if (prl_popc(s0) >= 5) fm = s0;
    else if (prl_popc(s1) >= 5) fm = s1;
    else if (prl_popc(s2) >= 5) fm = s2;
    else if (prl_popc(s3) >= 5) fm = s3;
    if (fm) {
        int t = prl_straight_top(fm);
        if (t >= 0) return PRL_HR_BASE_STRAIGHT_FLUSH + t;
    }
    // ---- multiplicities: carry-save add of the four masks --------------------------------------------------------------
    uint32_t x0 = s0 ^ s1, c0 = s0 & s1;
    uint32_t x1 = s2 ^ s3, c1 = s2 & s3;
    uint32_t lo = x0 ^ x1;                  // bit 0 of the per-rank count
    uint32_t tw = c0 ^ c1 ^ (x0 & x1);      // bit 1
    uint32_t quads = c0 & c1;               // count == 4
    uint32_t trips = tw & lo;               // count == 3
    uint32_t pairs = tw & ~lo;              // count == 2
    uint32_t all = s0 | s1 | s2 | s3;

    if (quads) {
        int q = prl_msb(quads);
        uint32_t others = all & ~(1u << q);
        uint32_t higher = others & ~((2u << q) - 1u);
        int k = higher ? prl_lsb(higher) : prl_msb(others);  // sorted-neighbour quirk of the reference binary
        return PRL_HR_BASE_QUADS + 13 * q + k;
    }
    if (trips && (pairs || (trips & (trips - 1)))) {
        int t = prl_msb(trips);
        uint32_t rest = (trips & ~(1u << t)) | pairs;
        return PRL_HR_BASE_FULL_HOUSE + 13 * t + prl_msb(rest);
    }
    if (fm) return PRL_HR_BASE_FLUSH + prl_pack_top(fm, 5);
    {
        int t = prl_straight_top(all);
        if (t >= 0) return PRL_HR_BASE_STRAIGHT + t;
    }
    if (trips) {
        int t = prl_msb(trips);
        return PRL_HR_BASE_TRIPS + 169 * t + prl_pack_top(all & ~(1u << t), 2);
    }
    if (pairs & (pairs - 1)) {  // >= 2 pairs
        int hi = prl_msb(pairs);
        int lo2 = prl_msb(pairs & ~(1u << hi));
        int k = prl_msb(all & ~(1u << hi) & ~(1u << lo2));
        return PRL_HR_BASE_TWO_PAIR + 169 * hi + 13 * lo2 + k;
    }
    if (pairs) {
        int p = prl_msb(pairs);
        return PRL_HR_BASE_PAIR + 2197 * p + prl_pack_top(all & ~(1u << p), 3);
    }
    return prl_pack_top(all, 5);
}

// cards are 52-deck 1d cards (rank = c >> 2, suit = c & 3), see prl_cards.h
PRL_HD PRL_INLINE void prl_add_card_52(int c, uint32_t s[4]) { s[c & 3] |= 1u << (c >> 2); }

PRL_HD PRL_INLINE int32_t prl_rank7_cards_52(const int8_t board[5], int h1, int h2) {
    uint32_t s[4] = {0, 0, 0, 0};
    for (int i = 0; i < 5; ++i) prl_add_card_52(board[i], s);
    prl_add_card_52(h1, s);
    prl_add_card_52(h2, s);
    return prl_rank7_masks(s[0], s[1], s[2], s[3]);
}

// 1-card-game rules (reference: game_rules.py:68-75 LeducRules.get_hand_rank, :133-140 BigLeducRules.get_hand_rank):
// pair with the board card -> bonus + rank, otherwise the rank of the hole card.
PRL_HD PRL_INLINE int32_t prl_rank_leduc(int hand_card, int board_card, int n_suits, int pair_bonus) {
    int hr = hand_card / n_suits, br = board_card / n_suits;
    return (hr == br) ? pair_bonus + hr : hr;
}
