// Local best response (LBR, arXiv:1612.07547) building blocks, host + device.
// reference: PokerRL/eval/lbr/LocalLBRWorker.py:379-512 (_LBRRolloutManager), PokerRL/game/PokerRange.py:26-84.
// Everything is float32 in NumPy's order: pairwise sums (8 accumulators, blocks of <= 128, recursive halving) over the
// elements in ascending index order, element-wise divisions, running float32 accumulation over the boards.
#pragma once
#include "prl_cards.h"
#include "prl_defs.h"
#include "prl_handeval.h"

#define PRL_LBR_MAX_CARDS 52
#define PRL_LBR_MAX_DEAL 5   // cards still to come when LBR evaluates (hold'em pre-flop: 5, flop: 2, turn: 1, river / Leduc flop: 0, Leduc pre-flop: 1)

// NumPy's pairwise sum (numpy/_core/src/umath/loops_utils.h.src) over a stream: `next()` yields the elements in order. The
// algorithm visits a[0], a[1], ... exactly once and in order, so no random access is needed. The recursion (halve until a
// block has <= 128 elements, 8 accumulators per block) is walked with an explicit frame stack: one instance of the block
// code, whatever n is. The template argument is kept for call-site compatibility (0 promises n <= 128).
// the next eight elements of a stream: streams with a member eight(float (&)[8]) deliver them as a group
template <class Next>
PRL_HD PRL_INLINE auto prl_nps_eight_impl(Next& next, float (&o)[8], int) -> decltype(next.eight(o), void()) { next.eight(o); }
template <class Next>
PRL_HD PRL_INLINE void prl_nps_eight_impl(Next& next, float (&o)[8], long) {
    for (int j = 0; j < 8; ++j) o[j] = next();
}
template <class Next>
PRL_HD PRL_INLINE void prl_nps_eight(Next& next, float (&o)[8]) { prl_nps_eight_impl(next, o, 0); }

// The frames of the walk live in REGISTERS: five levels (n <= 128 * 2^5: ranges have at most 1326 entries) as individual variables picked by
// compare-and-select. As an array indexed by the stack pointer they sit in private (scratch) memory, and every push / pop / look at the top
// frame is a vector-memory round trip that the lane waits for -- the batched LBR kernel spent 77 % of its wave cycles waiting and issued
// 1.7e9 scratch reads per launch (profiles/r06_lbr_pmc_sq.txt).
template <int D, class Next>
PRL_HD PRL_INLINE float prl_np_sum_stream(int n, Next& next) {
    constexpr int MAXD = 6;
    int fn[MAXD], fstage[MAXD];
    float fleft[MAXD];
#define PRL_NPS_GET(arr, i, out) do { out = arr[0]; for (int d_ = 1; d_ < MAXD; ++d_) out = (i) == d_ ? arr[d_] : out; } while (0)
#define PRL_NPS_SET(arr, i, val) do { for (int d_ = 0; d_ < MAXD; ++d_) arr[d_] = (i) == d_ ? (val) : arr[d_]; } while (0)
#pragma unroll
    for (int d = 0; d < MAXD; ++d) { fn[d] = 0; fstage[d] = 0; fleft[d] = 0.f; }
    int sp = 0;
    fn[0] = n;
    float ret = 0.f;
    while (sp >= 0) {
        int cn, cstage;
        PRL_NPS_GET(fn, sp, cn);
        PRL_NPS_GET(fstage, sp, cstage);
        if (cstage == 0) {
            if (cn < 8) {
                float res = 0.f;
                for (int i = 0; i < cn; ++i) res = res + next();
                ret = res;
                --sp;
            } else if (cn <= 128 || D == 0 || sp == MAXD - 1) {
                // eight accumulators; the next eight elements are FETCHED TOGETHER (prl_nps_eight: a stream that can hands them over as a
                // group -- eight independent gathers / divisions for the scheduler to overlap -- else eight calls)
                float r[8], t[8];
                prl_nps_eight(next, r);
                int i = 8;
                for (; i < cn - (cn % 8); i += 8) {
                    prl_nps_eight(next, t);
#pragma unroll
                    for (int j = 0; j < 8; ++j) r[j] = r[j] + t[j];
                }
                float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < cn; ++i) res = res + next();
                ret = res;
                --sp;
            } else {
                int n2 = cn / 2;
                n2 -= n2 % 8;
                PRL_NPS_SET(fstage, sp, 1);
                ++sp;
                PRL_NPS_SET(fn, sp, n2);
                PRL_NPS_SET(fstage, sp, 0);
            }
        } else if (cstage == 1) {  // the left half returned
            int n2 = cn / 2;
            n2 -= n2 % 8;
            PRL_NPS_SET(fleft, sp, ret);
            PRL_NPS_SET(fstage, sp, 2);
            ++sp;
            PRL_NPS_SET(fn, sp, cn - n2);
            PRL_NPS_SET(fstage, sp, 0);
        } else {  // the right half returned
            float l;
            PRL_NPS_GET(fleft, sp, l);
            ret = l + ret;
            --sp;
        }
    }
#undef PRL_NPS_GET
#undef PRL_NPS_SET
    return ret;
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

// ---------------------------------------------------------------------------------------------------------------------
// per-(range, board) and per-range pieces of the check-down equity (LocalLBRWorker.py:427-512); used by the stand-alone
// equity kernels (one call per LBR decision of the host worker) and by the device-resident batched LBR engine
// ---------------------------------------------------------------------------------------------------------------------
PRL_HD PRL_INLINE bool prl_lbr_blocked(const PrlLbrGame& g, int h, const int8_t* full_board) {
    for (int i = 0; i < g.n_board_total; ++i)
        if (prl_lbr_hand_has(g, h, full_board[i])) return true;
    return false;
}

// cls[h]: 1 where LBR's hand beats hand h on `full_board`, 2 where it ties (np.argwhere(handranks < / == lbr_rank), :424-425)
PRL_HD PRL_INLINE uint8_t prl_lbr_classify_hand(const PrlLbrGame& g, int lbr_idx, int h, const int8_t* full_board) {
    const int32_t rl = prl_lbr_rank(g, lbr_idx, full_board);
    const int32_t rh = prl_lbr_rank(g, h, full_board);
    return rh < rl ? 1 : (rh == rl ? 2 : 0);
}

// bit c set for every card c of the hand / of the board: "hand shares a card with the board" is one AND
PRL_HD PRL_INLINE unsigned long long prl_lbr_hand_mask(const PrlLbrGame& g, int h, const uint16_t* hole_lut /* c1 | c2 << 8, or NULL */) {
    if (g.n_hole == 1) return 1ull << h;
    if (hole_lut) { const unsigned v = hole_lut[h]; return (1ull << (v & 0xFFu)) | (1ull << (v >> 8)); }
    int c1, c2;
    prl_hole_cards_2(h, g.n_cards, &c1, &c2);
    return (1ull << c1) | (1ull << c2);
}

// PokerRange.set_cards_to_zero_prob(board) -> normalize (PokerRange.py:45-50, :67-84; an all-zero range becomes uniform),
// then the sums over the hands LBR beats (+ half the ties) (:509-510). `cl` is the classification of the FIRST board.
// n_big / n_eq: how many hands are in class 1 / 2 (the same for every board and range of a look-ahead: count once), or -1.
PRL_HD PRL_INLINE float prl_lbr_board_equity(const PrlLbrGame& g, const int8_t* full_board, const uint8_t* cl, const float* rg,
                                             const uint16_t* hole_lut = nullptr, int n_big = -1, int n_eq = -1) {
    unsigned long long bmask = 0ull;
    for (int i = 0; i < g.n_board_total; ++i) bmask |= 1ull << full_board[i];
    // with the hole-card table: two shifts of the board mask instead of building the hand's own 64-bit mask
    auto blocked = [&](int h) {
        if (g.n_hole == 2 && hole_lut) { const unsigned v = hole_lut[h]; return (((bmask >> (v & 0xFFu)) | (bmask >> (v >> 8))) & 1ull) != 0ull; }
        return (prl_lbr_hand_mask(g, h, hole_lut) & bmask) != 0ull;
    };
    int h0 = 0;
    auto nx = [&]() { const int h = h0++; return blocked(h) ? 0.f : rg[h]; };
    const float norm = prl_np_sum_stream<4>(g.R, nx);
    const float unif = (float)(1.0 / (double)g.R);
    auto value = [&](int h) { return norm == 0.f ? unif : (blocked(h) ? 0.f : rg[h]) / norm; };
    if (n_big < 0) {
        n_big = 0; n_eq = 0;
        for (int h = 0; h < g.R; ++h) { n_big += cl[h] == 1; n_eq += cl[h] == 2; }
    }
    int hb = 0, he = 0;
    auto next_big = [&]() { while (cl[hb] != 1) ++hb; return value(hb++); };
    auto next_eq = [&]() { while (cl[he] != 2) ++he; return value(he++); };
    const float s_big = prl_np_sum_stream<4>(n_big, next_big);
    const float s_eq = prl_np_sum_stream<4>(n_eq, next_eq);
    return s_big + s_eq / 2.0f;
}

// The same with the two classes given as ascending index LISTS (cls_list: the n_big hands LBR beats, then the n_eq hands it ties with).
// With the class bytes every "next element" has to scan forward from where the previous one stopped -- a loop-carried chain through an
// LDS read per step, which serialises the eight accumulators of NumPy's pairwise sum; with the lists element i is a pure function of i, the
// eight accumulator chains (each a gather, a blocker test and a correctly rounded division) run side by side. Same values, same order.
PRL_HD PRL_INLINE float prl_lbr_board_equity_lists(const PrlLbrGame& g, const int8_t* full_board, const uint16_t* cls_list, int n_big, int n_eq,
                                                   const float* rg, const uint16_t* hole_lut) {
    unsigned long long bmask = 0ull;
    for (int i = 0; i < g.n_board_total; ++i) bmask |= 1ull << full_board[i];
    struct Ctx {
        const PrlLbrGame& g; unsigned long long bmask; const float* rg; const uint16_t* hole_lut;
        PRL_HD PRL_INLINE bool blocked(int h) const {
            if (g.n_hole == 2 && hole_lut) { const unsigned v = hole_lut[h]; return (((bmask >> (v & 0xFFu)) | (bmask >> (v >> 8))) & 1ull) != 0ull; }
            return (prl_lbr_hand_mask(g, h, hole_lut) & bmask) != 0ull;
        }
        PRL_HD PRL_INLINE float live(int h) const { return blocked(h) ? 0.f : rg[h]; }
    };
    const Ctx cx = {g, bmask, rg, hole_lut};
    struct NormStream {  // the range with the hands the board blocks zeroed, ascending
        const Ctx& c; int h;
        PRL_HD PRL_INLINE float operator()() { return c.live(h++); }
        PRL_HD PRL_INLINE void eight(float (&o)[8]) {
            for (int j = 0; j < 8; ++j) o[j] = c.live(h + j);
            h += 8;
        }
    };
    NormStream ns = {cx, 0};
    const float norm = prl_np_sum_stream<4>(g.R, ns);
    const float unif = (float)(1.0 / (double)g.R);
    struct ClassStream {  // the normalised range over one class, by its index list
        const Ctx& c; const uint16_t* list; int i; float norm, unif;
        PRL_HD PRL_INLINE float value(int h) const { return norm == 0.f ? unif : c.live(h) / norm; }
        PRL_HD PRL_INLINE float operator()() { return value((int)list[i++]); }
        PRL_HD PRL_INLINE void eight(float (&o)[8]) {
            int hh[8];
            for (int j = 0; j < 8; ++j) hh[j] = (int)list[i + j];
            for (int j = 0; j < 8; ++j) o[j] = value(hh[j]);
            i += 8;
        }
    };
    ClassStream sb = {cx, cls_list, 0, norm, unif}, se = {cx, cls_list, n_big, norm, unif};
    const float s_big = prl_np_sum_stream<4>(n_big, sb);
    const float s_eq = prl_np_sum_stream<4>(n_eq, se);
    return s_big + s_eq / 2.0f;
}
// the cards that can still come, ascending (LocalLBRWorker.py:392-396)
PRL_HD PRL_INLINE int prl_lbr_possible_cards(const PrlLbrGame& g, int8_t* pc) {
    int n = 0;
    for (int c = 0; c < g.n_cards; ++c) {
        bool used = false;
        for (int i = 0; i < g.n_hole; ++i) used |= g.lbr_hand[i] == c;
        for (int i = 0; i < g.n_dealt; ++i) used |= g.board[i] == c;
        if (!used) pc[n++] = (int8_t)c;
    }
    return n;
}
PRL_HD PRL_INLINE int prl_lbr_n_boards(const PrlLbrGame& g) {
    const int n = g.n_cards - g.n_hole - g.n_dealt;
    return (int)prl_comb(n, g.n_to_deal);  // 1, n, n (n - 1) / 2 ...; hold'em pre-flop: C(50, 5) = 2 118 760
}
// b-th complete board in the reference's enumeration order (:408-417)
PRL_HD PRL_INLINE void prl_lbr_board_at(const PrlLbrGame& g, const int8_t* pc, int n_pc, int b, int8_t* fb) {
    int8_t c0 = 0, c1 = 0;  // the card(s) still to come; placed by selects: a write at a run-time position would push the caller's board into private memory
    if (g.n_to_deal == 1) c0 = pc[b];
    else if (g.n_to_deal == 2) {
        int i = 0, left = b;
        while (left >= n_pc - 1 - i) { left -= n_pc - 1 - i; ++i; }
        c0 = pc[i];
        c1 = pc[i + 1 + left];
    }
    for (int i = 0; i < 5; ++i) fb[i] = i < g.n_dealt ? g.board[i] : (i == g.n_dealt ? c0 : (i == g.n_dealt + 1 ? c1 : (int8_t)0));
}

// card-removal-aware board probabilities and the running float32 sum over the boards (:432-468, :470-497).
// cp / cp2: work arrays of n_cards floats each (cp2 only when two cards are to come); pc: the possible cards, ascending.
// The batched engine passes LDS for them: per-lane private arrays of this size would cap the waves the runtime keeps in flight.
// 1 - P(the agent holds card c) under range rg (:432-447): for 2-card hands the NumPy sum over the 51 hands holding c
PRL_HD PRL_INLINE float prl_lbr_card_not_held(const PrlLbrGame& g, const float* rg, int c) {
    float p;
    if (g.n_hole == 1) p = rg[c];
    else {
        int k = 0;  // the 51 hands holding c, ascending range index (= LUT_CARD_IN_WHAT_RANGE_IDXS[c])
        auto nx = [&]() {
            const int o = k < c ? k : k + 1;
            ++k;
            return rg[o < c ? prl_range_idx_2(o, c, g.n_cards) : prl_range_idx_2(c, o, g.n_cards)];
        };
        p = prl_np_sum_stream<0>(g.n_cards - 1, nx);
    }
    return 1.f - p;
}
// the rest of the reduction; cp[c] = prl_lbr_card_not_held(g, rg, c) on entry (the batched engine fills it with one lane per card)
PRL_HD PRL_INLINE float prl_lbr_reduce_range_cp(const PrlLbrGame& g, const float* e /* [n_boards] */, float* cp, float* cp2, const int8_t* pc, int n_pc) {
    for (int i = 0; i < g.n_hole; ++i) cp[g.lbr_hand[i]] = 0.f;
    for (int i = 0; i < g.n_dealt; ++i) cp[g.board[i]] = 0.f;
    {
        int k = 0;
        auto nx = [&]() { return cp[k++]; };
        const float s = prl_np_sum_stream<0>(g.n_cards, nx);
        if (s > 0.f)
            for (int c = 0; c < g.n_cards; ++c) cp[c] = cp[c] / s;
    }
    float win = 0.f;
    bool first = true;
    auto add = [&](float x) { win = first ? x : win + x; first = false; };  // 0.0 (Python float) + float32 -> float32
    int b = 0;
    if (g.n_to_deal == 0) add(e[b++] * 1.0f);
    else if (g.n_to_deal == 1) {
        for (int i = 0; i < n_pc; ++i) add(e[b++] * cp[pc[i]]);
    } else {
        for (int i = 0; i + 1 < n_pc; ++i) {
            for (int c = 0; c < g.n_cards; ++c) cp2[c] = cp[c];
            cp2[pc[i]] = 0.f;
            int k = 0;
            auto nx = [&]() { return cp2[k++]; };
            const float s = prl_np_sum_stream<0>(g.n_cards, nx);
            for (int c = 0; c < g.n_cards; ++c) cp2[c] = cp2[c] / s;
            const float r1 = cp[pc[i]];  // 1.0 * card_probs[c]
            for (int j = i + 1; j < n_pc; ++j) add(e[b++] * (r1 * cp2[pc[j]]));
        }
    }
    float fact = 1.f;
    for (int m = 2; m <= g.n_to_deal; ++m) fact = fact * (float)m;
    return win * fact;  // :463-468
}
// the same for three to five cards to come (hold'em before the flop): _calc_eq's recursion (:470-512) as an odometer -- at depth l the
// card probabilities of depth l - 1 with the card just dealt zeroed and re-normalised (NumPy sum over the deck, element-wise division),
// the reach probability the float32 product of the dealt cards' probabilities in deal order, the boards in lexicographic order of the
// ascending cards to come. cps: [n_to_deal][n_cards] work floats (level 0 = cp on entry, as prl_lbr_reduce_range_cp).
PRL_HD PRL_INLINE float prl_lbr_reduce_range_deep(const PrlLbrGame& g, const float* e, float* cps, const int8_t* pc, int n_pc) {
    const int k = g.n_to_deal, nc = g.n_cards;
    float* cp = cps;
    for (int i = 0; i < g.n_hole; ++i) cp[g.lbr_hand[i]] = 0.f;
    for (int i = 0; i < g.n_dealt; ++i) cp[g.board[i]] = 0.f;
    {
        int j = 0;
        auto nx = [&]() { return cp[j++]; };
        const float s = prl_np_sum_stream<0>(nc, nx);
        if (s > 0.f)
            for (int c = 0; c < nc; ++c) cp[c] = cp[c] / s;
    }
    float win = 0.f, reach[PRL_LBR_MAX_DEAL + 1];
    bool first = true;
    int idx[PRL_LBR_MAX_DEAL], l = 0, b = 0;
    idx[0] = 0;
    reach[0] = 1.f;
    while (l >= 0) {
        if (idx[l] > n_pc - (k - l)) {  // this depth has dealt its last card: back up
            if (--l >= 0) ++idx[l];
            continue;
        }
        const float* cur = cps + (size_t)l * nc;
        const int card = pc[idx[l]];
        const float r = l == 0 ? cur[card] : reach[l] * cur[card];  // 1.0 * p at depth 0
        if (l == k - 1) {
            const float x = e[b++] * r;
            win = first ? x : win + x;
            first = false;
            ++idx[l];
            continue;
        }
        float* nxt = cps + (size_t)(l + 1) * nc;
        for (int c = 0; c < nc; ++c) nxt[c] = cur[c];
        nxt[card] = 0.f;
        int j = 0;
        auto nx = [&]() { return nxt[j++]; };
        const float s = prl_np_sum_stream<0>(nc, nx);
        for (int c = 0; c < nc; ++c) nxt[c] = nxt[c] / s;
        reach[l + 1] = r;
        idx[l + 1] = idx[l] + 1;
        ++l;
    }
    float fact = 1.f;
    for (int m = 2; m <= k; ++m) fact = fact * (float)m;
    return win * fact;
}
PRL_HD PRL_INLINE float prl_lbr_reduce_range_w(const PrlLbrGame& g, const float* rg, const float* e /* [n_boards] */, float* cp, float* cp2,
                                               const int8_t* pc, int n_pc) {
    for (int c = 0; c < g.n_cards; ++c) cp[c] = prl_lbr_card_not_held(g, rg, c);
    return prl_lbr_reduce_range_cp(g, e, cp, cp2, pc, n_pc);
}
PRL_HD PRL_INLINE float prl_lbr_reduce_range(const PrlLbrGame& g, const float* rg, const float* e /* [n_boards] */) {
    int8_t pc[PRL_LBR_MAX_CARDS];
    const int n_pc = prl_lbr_possible_cards(g, pc);
    if (g.n_to_deal > 2) {
        float cps[PRL_LBR_MAX_DEAL * PRL_LBR_MAX_CARDS];
        for (int c = 0; c < g.n_cards; ++c) cps[c] = prl_lbr_card_not_held(g, rg, c);
        return prl_lbr_reduce_range_deep(g, e, cps, pc, n_pc);
    }
    float cp[PRL_LBR_MAX_CARDS], cp2[PRL_LBR_MAX_CARDS];
    return prl_lbr_reduce_range_w(g, rg, e, cp, cp2, pc, n_pc);
}
