// Local best response (LBR, arXiv:1612.07547) building blocks, host + device.
// reference: PokerRL/eval/lbr/LocalLBRWorker.py:379-512 (_LBRRolloutManager), PokerRL/game/PokerRange.py:26-84.
// Everything is float32 in NumPy's order: pairwise sums (8 accumulators, blocks of <= 128, recursive halving) over the
// elements in ascending index order, element-wise divisions, running float32 accumulation over the boards.
#pragma once
#include "prl_cards.h"
#include "prl_defs.h"
#include "prl_handeval.h"

#define PRL_LBR_MAX_CARDS 52
#define PRL_LBR_MAX_DEAL 2   // cards still to come when LBR evaluates (flop: 2, turn: 1, river / Leduc flop: 0, Leduc pre-flop: 1)

// NumPy's pairwise sum (numpy/_core/src/umath/loops_utils.h.src) over a stream: `next()` yields the elements in order. The
// algorithm visits a[0], a[1], ... exactly once and in order, so no random access is needed. D bounds the halving depth
// (n <= 128 * 2^D).
template <int D, class Next>
PRL_HD PRL_INLINE float prl_np_sum_stream(int n, Next& next) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = res + next();
        return res;
    }
    if (n <= 128 || D == 0) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = next();
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = r[j] + next();
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res = res + next();
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    const float a = prl_np_sum_stream<(D > 0 ? D - 1 : 0)>(n2, next);
    const float b = prl_np_sum_stream<(D > 0 ? D - 1 : 0)>(n - n2, next);
    return a + b;
}

struct PrlLbrGame {
    int32_t n_hole, n_cards, n_suits, rank_rule, R;
    int32_t n_board_total;   // board cards of the last round
    int32_t n_dealt;         // board cards on the table now
    int32_t n_to_deal;       // n_board_total - n_dealt
    int8_t board[5];         // dealt cards first, in deal order
    int8_t lbr_hand[2];
    int8_t pad;
};

PRL_HD PRL_INLINE bool prl_lbr_hand_has(const PrlLbrGame& g, int h, int card) {
    if (g.n_hole == 1) return h == card;
    int c1, c2;
    prl_hole_cards_2(h, g.n_cards, &c1, &c2);
    return c1 == card || c2 == card;
}

// rank of hand h on a complete board; hold'em: -1 for hands that share a card with the board (lib_hand_eval semantics)
PRL_HD PRL_INLINE int32_t prl_lbr_rank(const PrlLbrGame& g, int h, const int8_t* full_board) {
    if (g.n_hole == 1) return prl_rank_leduc(h, full_board[0], g.n_suits, g.rank_rule == 1 ? 10000 : 100);
    int c1, c2;
    prl_hole_cards_2(h, g.n_cards, &c1, &c2);
    for (int i = 0; i < 5; ++i)
        if (full_board[i] == c1 || full_board[i] == c2) return -1;
    return prl_rank7_cards_52(full_board, c1, c2);
}
