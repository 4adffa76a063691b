// LBR check-down equity on the device (reference: LocalLBRWorker.py:379-512).
//   prl_k_lbr_classify : per hand: does LBR's hand beat / tie the hand on the first complete board (see the quirk below)
//   prl_k_lbr_board_eq : per (query range, board): the agent range with the board's cards removed and re-normalised
//                        (PokerRange.set_cards_to_zero_prob + normalize), summed over the hands LBR beats (+ half the ties)
//   prl_k_lbr_reduce   : per query: card-removal-aware board probabilities (get_lbr_checkdown_equity / _calc_eq) and the
//                        running float32 sum over the boards in the reference's enumeration order
#include <string.h>

#include <string>
#include <vector>

#include "prl_device.h"
#include "prl_host.h"
#include "prl_lbr.h"
#include "prl_rt.h"

extern "C" int32_t prl_device_available(void);

PRL_GLOBAL void prl_k_lbr_classify(PrlLbrGame g, const int8_t* __restrict__ boards, int n_boards, uint8_t* __restrict__ cls) {
    const size_t total = (size_t)n_boards * g.R;
    const int lbr_idx = g.n_hole == 1 ? g.lbr_hand[0] : prl_range_idx_2(g.lbr_hand[0], g.lbr_hand[1], g.n_cards);
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int b = (int)(t / g.R), h = (int)(t % g.R);
        const int8_t* fb = boards + (size_t)b * 5;
        const int32_t rl = prl_lbr_rank(g, lbr_idx, fb);
        const int32_t rh = prl_lbr_rank(g, h, fb);
        cls[t] = rh < rl ? 1 : (rh == rl ? 2 : 0);  // np.argwhere(handranks < lbr_rank) / (== lbr_rank), :424-425
    }
}

PRL_GLOBAL void prl_k_lbr_board_eq(PrlLbrGame g, const int8_t* __restrict__ boards, int n_boards, const uint8_t* __restrict__ cls,
                                   const float* __restrict__ ranges, int n_q, float* __restrict__ eq) {
    const int total = n_q * n_boards;
    for (int t = (int)(prl_bid() * prl_nthreads() + prl_tid()); t < total; t += (int)(prl_nblocks() * prl_nthreads())) {
        const int q = t / n_boards, b = t % n_boards;
        const float* rg = ranges + (size_t)q * g.R;
        const int8_t* fb = boards + (size_t)b * 5;
        // REFERENCE QUIRK (LocalLBRWorker.py:470, :509-510): the board counter `_i` handed down the recursion is never
        // advanced, so the win / tie index lists of the FIRST enumerated board are applied to every board. Replicated: it
        // defines the reference's LBR numbers; cls holds that one board's classification.
        const uint8_t* cl = cls;
        auto blocked = [&](int h) {
            for (int i = 0; i < g.n_board_total; ++i)
                if (prl_lbr_hand_has(g, h, fb[i])) return true;
            return false;
        };
        // PokerRange.set_cards_to_zero_prob(board) -> normalize (PokerRange.py:45-50, :67-84): an all-zero range becomes uniform
        int h0 = 0;
        auto nx = [&]() { const int h = h0++; return blocked(h) ? 0.f : rg[h]; };
        const float norm = prl_np_sum_stream<4>(g.R, nx);
        const float unif = (float)(1.0 / (double)g.R);
        auto value = [&](int h) { return norm == 0.f ? unif : (blocked(h) ? 0.f : rg[h]) / norm; };
        int n_big = 0, n_eq = 0;
        for (int h = 0; h < g.R; ++h) { n_big += cl[h] == 1; n_eq += cl[h] == 2; }
        int hb = 0, he = 0;
        auto next_big = [&]() { while (cl[hb] != 1) ++hb; return value(hb++); };
        auto next_eq = [&]() { while (cl[he] != 2) ++he; return value(he++); };
        const float s_big = prl_np_sum_stream<4>(n_big, next_big);
        const float s_eq = prl_np_sum_stream<4>(n_eq, next_eq);
        eq[t] = s_big + s_eq / 2.0f;  // :509-510
    }
}

PRL_GLOBAL void prl_k_lbr_reduce(PrlLbrGame g, int n_boards, const float* __restrict__ ranges, int n_q, const float* __restrict__ eq,
                                 float* __restrict__ out) {
    for (int q = (int)(prl_bid() * prl_nthreads() + prl_tid()); q < n_q; q += (int)(prl_nblocks() * prl_nthreads())) {
        const float* rg = ranges + (size_t)q * g.R;
        const float* e = eq + (size_t)q * n_boards;
        float cp[PRL_LBR_MAX_CARDS];
        // PokerRange.get_card_probs (:26-38) -> 1 - p, LBR's and the dealt cards zeroed, normalised if the sum is positive (:432-449)
        for (int c = 0; c < g.n_cards; ++c) {
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
            cp[c] = 1.f - p;
        }
        for (int i = 0; i < g.n_hole; ++i) cp[g.lbr_hand[i]] = 0.f;
        for (int i = 0; i < g.n_dealt; ++i) cp[g.board[i]] = 0.f;
        {
            int k = 0;
            auto nx = [&]() { return cp[k++]; };
            const float s = prl_np_sum_stream<0>(g.n_cards, nx);
            if (s > 0.f) {
                int k2 = 0;
                auto nx2 = [&]() { return cp[k2++]; };
                const float s2 = prl_np_sum_stream<0>(g.n_cards, nx2);
                for (int c = 0; c < g.n_cards; ++c) cp[c] = cp[c] / s2;
            }
        }
        // possible cards ascending; boards in the reference's enumeration order (:451-497), running float32 sum
        int8_t pc[PRL_LBR_MAX_CARDS];
        int n_pc = 0;
        for (int c = 0; c < g.n_cards; ++c) {
            bool used = false;
            for (int i = 0; i < g.n_hole; ++i) used |= g.lbr_hand[i] == c;
            for (int i = 0; i < g.n_dealt; ++i) used |= g.board[i] == c;
            if (!used) pc[n_pc++] = (int8_t)c;
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
                float cp2[PRL_LBR_MAX_CARDS];
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
        out[q] = win * fact;  // :463-468
    }
}

extern "C" int32_t prl_lbr_checkdown_equity(const PrlRules* rules, const int8_t* board_dealt, int32_t n_dealt, const int8_t* lbr_hand,
                                            const float* ranges, int32_t n_q, float* out_wp) {
    if (!rules || !lbr_hand || !ranges || !out_wp || n_q <= 0 || n_dealt < 0 || (n_dealt > 0 && !board_dealt)) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device: LBR has no CPU fallback"); return PRL_ERR_NO_DEVICE; }
    PrlLbrGame g;
    memset(&g, 0, sizeof(g));
    g.n_hole = rules->n_hole_cards; g.n_cards = rules->n_cards; g.n_suits = rules->n_suits; g.rank_rule = rules->rank_rule; g.R = rules->range_size;
    g.n_board_total = rules->n_board_cards; g.n_dealt = n_dealt; g.n_to_deal = g.n_board_total - n_dealt;
    if (g.n_hole < 1 || g.n_hole > 2 || g.n_cards > PRL_LBR_MAX_CARDS || g.n_board_total > 5 || (g.n_hole == 2 && (g.n_cards != 52 || g.n_board_total != 5))) {
        prl_set_error("LBR: 1-hole-card games or 52-card hold'em with 5 board cards"); return PRL_ERR_UNSUPPORTED;
    }
    if (g.n_to_deal < 0 || g.n_to_deal > PRL_LBR_MAX_DEAL) { prl_set_error("LBR equity: at most 2 board cards to come (set lbr_check_to_round)"); return PRL_ERR_UNSUPPORTED; }
    for (int i = 0; i < n_dealt; ++i) g.board[i] = board_dealt[i];
    for (int i = 0; i < g.n_hole; ++i) g.lbr_hand[i] = lbr_hand[i];
    if (g.n_hole == 2 && g.lbr_hand[0] > g.lbr_hand[1]) { int8_t t = g.lbr_hand[0]; g.lbr_hand[0] = g.lbr_hand[1]; g.lbr_hand[1] = t; }
    // complete boards in the reference's order: the cards still to come ascending, each board once (:408-417)
    std::vector<int8_t> pc;
    for (int c = 0; c < g.n_cards; ++c) {
        bool used = false;
        for (int i = 0; i < g.n_hole; ++i) used |= g.lbr_hand[i] == c;
        for (int i = 0; i < n_dealt; ++i) used |= g.board[i] == c;
        if (!used) pc.push_back((int8_t)c);
    }
    std::vector<int8_t> boards;
    auto push = [&](int a, int b) {
        int8_t fb[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < n_dealt; ++i) fb[i] = g.board[i];
        if (a >= 0) fb[n_dealt] = (int8_t)a;
        if (b >= 0) fb[n_dealt + 1] = (int8_t)b;
        boards.insert(boards.end(), fb, fb + 5);
    };
    if (g.n_to_deal == 0) push(-1, -1);
    else if (g.n_to_deal == 1) for (size_t i = 0; i < pc.size(); ++i) push(pc[i], -1);
    else for (size_t i = 0; i + 1 < pc.size(); ++i) for (size_t j = i + 1; j < pc.size(); ++j) push(pc[i], pc[j]);
    const int n_boards = (int)(boards.size() / 5);
    int8_t* d_boards = nullptr; uint8_t* d_cls = nullptr; float *d_rg = nullptr, *d_eq = nullptr, *d_out = nullptr;
    int rc = PRL_OK;
#define LB_TRY(x) do { if ((x) != hipSuccess) { prl_set_error("HIP error in prl_lbr_checkdown_equity"); rc = PRL_ERR_HIP; goto done; } } while (0)
    LB_TRY(hipMalloc((void**)&d_boards, boards.size()));
    LB_TRY(hipMalloc((void**)&d_cls, (size_t)g.R));
    LB_TRY(hipMalloc((void**)&d_rg, (size_t)n_q * g.R * sizeof(float)));
    LB_TRY(hipMalloc((void**)&d_eq, (size_t)n_q * n_boards * sizeof(float)));
    LB_TRY(hipMalloc((void**)&d_out, (size_t)n_q * sizeof(float)));
    LB_TRY(hipMemcpy(d_boards, boards.data(), boards.size(), hipMemcpyHostToDevice));
    LB_TRY(hipMemcpy(d_rg, ranges, (size_t)n_q * g.R * sizeof(float), hipMemcpyHostToDevice));
    {
        const size_t items = (size_t)g.R;  // first board only
        PRL_LAUNCH(prl_k_lbr_classify, (int)((items + 255) / 256), 256, 0, nullptr, g, (const int8_t*)d_boards, 1, d_cls);
        PRL_LAUNCH(prl_k_lbr_board_eq, (n_q * n_boards + 63) / 64, 64, 0, nullptr, g, (const int8_t*)d_boards, n_boards, (const uint8_t*)d_cls,
                   (const float*)d_rg, n_q, d_eq);
        PRL_LAUNCH(prl_k_lbr_reduce, (n_q + 63) / 64, 64, 0, nullptr, g, n_boards, (const float*)d_rg, n_q, (const float*)d_eq, d_out);
    }
    LB_TRY(hipDeviceSynchronize());
    LB_TRY(hipMemcpy(out_wp, d_out, (size_t)n_q * sizeof(float), hipMemcpyDeviceToHost));
#undef LB_TRY
done:
    (void)hipFree(d_boards); (void)hipFree(d_cls); (void)hipFree(d_rg); (void)hipFree(d_eq); (void)hipFree(d_out);
    return rc;
}
