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
#include "prl_kernels.h"
#include "prl_lbr.h"
#include "prl_rt.h"

extern "C" int32_t prl_device_available(void);
int prl_hole_lut_device(const uint16_t** out);  // prl_capi_device.hip: device-resident [1326] c1 | c2 << 8

PRL_GLOBAL void prl_k_lbr_classify(PrlLbrGame g, const int8_t* __restrict__ boards, int n_boards, uint8_t* __restrict__ cls) {
    const size_t total = (size_t)n_boards * g.R;
    const int lbr_idx = g.n_hole == 1 ? g.lbr_hand[0] : prl_range_idx_2(g.lbr_hand[0], g.lbr_hand[1], g.n_cards);
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads())
        cls[t] = prl_lbr_classify_hand(g, lbr_idx, (int)(t % g.R), boards + (t / g.R) * 5);
}

// REFERENCE QUIRK (LocalLBRWorker.py:470, :509-510): the board counter `_i` handed down the recursion is never advanced, so
// the win / tie index lists of the FIRST enumerated board are applied to every board. Replicated: it defines the
// reference's LBR numbers; cls holds that one board's classification.
PRL_GLOBAL void prl_k_lbr_board_eq(PrlLbrGame g, const int8_t* __restrict__ boards, int n_boards, const uint8_t* __restrict__ cls,
                                   const float* __restrict__ ranges, int n_q, float* __restrict__ eq, const uint16_t* __restrict__ hole_lut) {
    const int total = n_q * n_boards;
    for (int t = (int)(prl_bid() * prl_nthreads() + prl_tid()); t < total; t += (int)(prl_nblocks() * prl_nthreads())) {
        const int q = t / n_boards, b = t % n_boards;
        eq[t] = prl_lbr_board_equity(g, boards + (size_t)b * 5, cls, ranges + (size_t)q * g.R, hole_lut);
    }
}

PRL_GLOBAL void prl_k_lbr_reduce(PrlLbrGame g, int n_boards, const float* __restrict__ ranges, int n_q, const float* __restrict__ eq,
                                 float* __restrict__ out) {
    for (int q = (int)(prl_bid() * prl_nthreads() + prl_tid()); q < n_q; q += (int)(prl_nblocks() * prl_nthreads()))
        out[q] = prl_lbr_reduce_range(g, ranges + (size_t)q * g.R, eq + (size_t)q * n_boards);
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
    if (g.n_to_deal < 0 || g.n_to_deal > PRL_LBR_MAX_DEAL) { prl_set_error("LBR equity: at most 5 board cards to come"); return PRL_ERR_UNSUPPORTED; }
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
    {   // every combination of n_to_deal of the possible cards, lexicographic (what the nested loops of :408-417 produce)
        const int k = g.n_to_deal, m = (int)pc.size();
        boards.reserve((size_t)prl_comb(m, k) * 5);
        int idx[PRL_LBR_MAX_DEAL + 1];
        for (int i = 0; i < k; ++i) idx[i] = i;
        for (bool more = k <= m; more;) {
            int8_t fb[5] = {0, 0, 0, 0, 0};
            for (int i = 0; i < n_dealt; ++i) fb[i] = g.board[i];
            for (int i = 0; i < k; ++i) fb[n_dealt + i] = pc[idx[i]];
            boards.insert(boards.end(), fb, fb + 5);
            int i = k - 1;
            while (i >= 0 && idx[i] == m - k + i) --i;
            if (i < 0) more = false;
            else { ++idx[i]; for (int j = i + 1; j < k; ++j) idx[j] = idx[j - 1] + 1; }
        }
    }
    const int n_boards = (int)(boards.size() / 5);
    if ((long long)n_q * n_boards > 0x7FFFFFFFll) { prl_set_error("LBR equity: too many (range, board) pairs in one call"); return PRL_ERR_ARG; }
    const uint16_t* hole_lut = nullptr;  // hold'em: the process-wide (c1, c2) table of the hand evaluator
    if (g.n_hole == 2 && prl_hole_lut_device(&hole_lut) != PRL_OK) return PRL_ERR_HIP;
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
        const long long items_eq = (long long)n_q * n_boards;
        PRL_LAUNCH(prl_k_lbr_board_eq, (int)((items_eq + 63) / 64 < 262144 ? (items_eq + 63) / 64 : 262144), 64, 0, nullptr, g, (const int8_t*)d_boards, n_boards, (const uint8_t*)d_cls,
                   (const float*)d_rg, n_q, d_eq, hole_lut);
        PRL_LAUNCH(prl_k_lbr_reduce, (n_q + 63) / 64, 64, 0, nullptr, g, n_boards, (const float*)d_rg, n_q, (const float*)d_eq, d_out);
    }
    LB_TRY(hipDeviceSynchronize());
    LB_TRY(hipMemcpy(out_wp, d_out, (size_t)n_q * sizeof(float), hipMemcpyDeviceToHost));
#undef LB_TRY
done:
    (void)hipFree(d_boards); (void)hipFree(d_cls); (void)hipFree(d_rg); (void)hipFree(d_eq); (void)hipFree(d_out);
    return rc;
}
