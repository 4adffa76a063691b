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

// the b-th board of the enumeration, any number of cards to come: the lexicographic k-combination number b of the possible cards (what the host
// used to enumerate and upload: 10.6 MB per pre-flop call)
PRL_HD PRL_INLINE void prl_lbr_board_unrank(const PrlLbrGame& g, const int8_t* pc, int n_pc, long long b, int8_t* fb) {
    const int k = g.n_to_deal;
    for (int i = 0; i < 5; ++i) fb[i] = i < g.n_dealt ? g.board[i] : (int8_t)0;
    int v = 0;
    for (int j = 0; j < k; ++j) {
        for (;; ++v) {
            const long long below = prl_comb(n_pc - 1 - v, k - 1 - j);  // boards that continue with card v at position j
            if (b < below) break;
            b -= below;
        }
        const int8_t c = pc[v++];
        for (int i = 0; i < 5; ++i) fb[i] = i == g.n_dealt + j ? c : fb[i];  // (selects: a write at a run-time position would push the board into private memory)
    }
}

// (the classes as ascending index lists -- prl_lbr_board_equity_lists: element i of a class sum is a pure function of i, so NumPy's eight accumulator chains run
// side by side instead of scanning the class bytes; the same values in the same order)
PRL_GLOBAL void prl_k_lbr_board_eq_deep(PrlLbrGame g, int n_boards, const uint16_t* __restrict__ cls_list, int n_big, int n_eq, const float* __restrict__ ranges, int n_q,
                                        float* __restrict__ eq, const uint16_t* __restrict__ hole_lut) {
    int8_t pc[PRL_LBR_MAX_CARDS];
    const int n_pc = prl_lbr_possible_cards(g, pc);
    const long long total = (long long)n_q * n_boards;
    for (long long t = (long long)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (long long)prl_nblocks() * prl_nthreads()) {
        const int q = (int)(t / n_boards), b = (int)(t - (long long)q * n_boards);
        int8_t fb[5];
        prl_lbr_board_unrank(g, pc, n_pc, b, fb);
        eq[t] = prl_lbr_board_equity_lists(g, fb, cls_list, n_big, n_eq, ranges + (size_t)q * g.R, hole_lut);
    }
}

PRL_GLOBAL void prl_k_lbr_reduce(PrlLbrGame g, int n_boards, const float* __restrict__ ranges, int n_q, const float* __restrict__ eq,
                                 float* __restrict__ out) {
    for (int q = (int)(prl_bid() * prl_nthreads() + prl_tid()); q < n_q; q += (int)(prl_nblocks() * prl_nthreads()))
        out[q] = prl_lbr_reduce_range(g, ranges + (size_t)q * g.R, eq + (size_t)q * n_boards);
}

// ---- more than two cards to come (hold'em before the flop: C(50, 5) = 2 118 760 boards per range) ---------------------------------------------------------
// prl_lbr_reduce_range_deep (prl_lbr.h) is _calc_eq's recursion (LocalLBRWorker.py:470-512) as ONE sequential walk: per interior node of the deal tree the
// card probabilities re-normalised, per board the product of the dealt cards' probabilities times the board's equity, added to a running float32 sum.
// One lane doing that is 230 300 re-normalisations of 52 entries in private memory and 2.1 M dependent global reads: 1.5 s per call (round 6: the batched
// engine's pre-flop rounds and the host worker both pay it per decision). The VALUES do not need the walk: a board's term depends on its own path only.
//   prl_k_lbr_deep_terms: one lane per (range, prefix of the first k - 1 cards to come): the chain of re-normalised card probabilities down ITS path
//                         (the same operations on the same numbers as the walk), then x[b] = e[b] * reach for the boards below the prefix -- they are
//                         consecutive in the enumeration order -- written over e[b];
//   prl_k_lbr_deep_sum:   per range the running float32 sum of the terms in board order (the one step that IS sequential): the workgroup streams the
//                         terms through LDS, one lane adds.
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(256) prl_k_lbr_deep_terms(PrlLbrGame g, const float* __restrict__ ranges, int n_q, int n_boards, int n_prefix, float* __restrict__ eq) {
    float* cp_all = (float*)prl_smem();  // [n_cards][blockDim]: a lane's card probabilities, lane index fastest (no bank conflicts)
    const int nt = (int)prl_nthreads(), tid = (int)prl_tid();
    const long long t = (long long)prl_bid() * nt + tid;
    if (t >= (long long)n_q * n_prefix) return;
    const int q = (int)(t / n_prefix);
    int pr = (int)(t - (long long)q * n_prefix);
    const int k = g.n_to_deal, nc = g.n_cards;
    int8_t pc[PRL_LBR_MAX_CARDS];
    const int n_pc = prl_lbr_possible_cards(g, pc);
    auto cp = [&](int c) -> float& { return cp_all[(size_t)c * nt + tid]; };
    // the prefix: combination number `pr` of k - 1 of the first n_pc - 1 possible cards, lexicographic (the walk's order of interior paths)
    int idx[PRL_LBR_MAX_DEAL];
    {
        int v = 0;
        for (int j = 0; j < k - 1; ++j) {
            for (;; ++v) {
                const int below = (int)prl_comb(n_pc - 1 - (v + 1), k - 2 - j);  // prefixes that continue with card v at position j
                if (pr < below) break;
                pr -= below;
            }
            idx[j] = v++;
        }
    }
    // rank of the first board below the prefix among all boards (lexicographic k-combinations of n_pc cards)
    long long b0 = 0;
    {
        int prev = -1;
        for (int j = 0; j < k - 1; ++j) {
            for (int v = prev + 1; v < idx[j]; ++v) b0 += prl_comb(n_pc - 1 - v, k - 1 - j);
            prev = idx[j];
        }
    }
    // the chain of card probabilities down the path (prl_lbr_reduce_range_deep, level by level; one array, rewritten in place)
    const float* rg = ranges + (size_t)q * g.R;
    for (int c = 0; c < nc; ++c) cp(c) = prl_lbr_card_not_held(g, rg, c);
    for (int i = 0; i < g.n_hole; ++i) cp(g.lbr_hand[i]) = 0.f;
    for (int i = 0; i < g.n_dealt; ++i) cp(g.board[i]) = 0.f;
    {
        int j = 0;
        auto nx = [&]() { return cp(j++); };
        const float s = prl_np_sum_stream<0>(nc, nx);
        if (s > 0.f)
            for (int c = 0; c < nc; ++c) cp(c) = cp(c) / s;
    }
    float reach = 1.f;
    for (int l = 0; l < k - 1; ++l) {
        const int card = pc[idx[l]];
        reach = l == 0 ? cp(card) : reach * cp(card);  // 1.0 * p at depth 0
        cp(card) = 0.f;
        int j = 0;
        auto nx = [&]() { return cp(j++); };
        const float s = prl_np_sum_stream<0>(nc, nx);
        for (int c = 0; c < nc; ++c) cp(c) = cp(c) / s;
    }
    float* e = eq + (size_t)q * n_boards + b0;
    int b = 0;
    for (int i = idx[k - 2] + 1; i < n_pc; ++i, ++b) {
        const float r = k == 1 ? cp(pc[i]) : reach * cp(pc[i]);
        e[b] = e[b] * r;
    }
}

// One WAVE per range. The sum is a strictly sequential float32 chain (that is the reference's order); what can be taken off the chain is everything but
// the add itself: the wave loads 64 terms per register (coalesced, eight registers in flight), and every lane runs the same chain taking term j of a
// register by a lane broadcast (v_readlane: a scalar operand, independent of the chain) -- the chain is one dependent v_add per term.
#if defined(PRL_EMU)
#define PRL_LANE_BCAST(v, j) prl_shfl((v), (j))
#define PRL_SCHED_FENCE() do { } while (0)
#else
#define PRL_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
PRL_DEV PRL_INLINE float prl_lane_bcast_(float v, int j) { int i; __builtin_memcpy(&i, &v, 4); i = __builtin_amdgcn_readlane(i, j); float o; __builtin_memcpy(&o, &i, 4); return o; }
#define PRL_LANE_BCAST(v, j) prl_lane_bcast_((v), (j))
#endif
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(64) prl_k_lbr_deep_sum(int k, int n_boards, const float* __restrict__ terms, float* __restrict__ out) {
    const int q = (int)prl_bid(), lane = (int)prl_tid();
    const float* x = terms + (size_t)q * n_boards;
    float win = 0.f;
    bool first = true;
    for (int b0 = 0; b0 < n_boards; b0 += 8 * 64) {
        float v[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) { const int i = b0 + r * 64 + lane; v[r] = i < n_boards ? x[i] : 0.f; }
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int left = n_boards - (b0 + r * 64);
            if (left >= 64 && !first) {
                // sixteen broadcasts, a scheduling fence, sixteen adds: left to itself the compiler reads every lane into ONE scalar register right before
                // its add (v_readlane, s_nop 1, v_add: 19 clocks per term); sixteen live scalars take the broadcasts off the chain
#pragma unroll
                for (int j0 = 0; j0 < 64; j0 += 16) {
                    float t[16];
#pragma unroll
                    for (int j = 0; j < 16; ++j) t[j] = PRL_LANE_BCAST(v[r], j0 + j);
                    PRL_SCHED_FENCE();
#pragma unroll
                    for (int j = 0; j < 16; ++j) win = win + t[j];
                    PRL_SCHED_FENCE();
                }
            } else {
                for (int j = 0; j < 64; ++j) {
                    const float t = PRL_LANE_BCAST(v[r], j);
                    if (j < left) { win = first ? t : win + t; first = false; }  // 0.0 (Python float) + float32 -> float32
                }
            }
        }
    }
    if (lane == 0) {
        float fact = 1.f;
        for (int m = 2; m <= k; ++m) fact = fact * (float)m;
        out[q] = win * fact;
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
    const bool deep = g.n_to_deal > 2;  // the kernels number the boards themselves (prl_lbr_board_unrank); only the first one is needed here (classification)
    if (deep) {
        int8_t fb[5] = {0, 0, 0, 0, 0};
        for (int i = 0; i < n_dealt; ++i) fb[i] = g.board[i];
        for (int i = 0; i < g.n_to_deal; ++i) fb[n_dealt + i] = pc[i];
        boards.insert(boards.end(), fb, fb + 5);
    } else {   // every combination of n_to_deal of the possible cards, lexicographic (what the nested loops of :408-417 produce)
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
    const long long n_boards_ll = deep ? prl_comb((int)pc.size(), g.n_to_deal) : (long long)(boards.size() / 5);
    const int n_boards = (int)n_boards_ll;
    if (n_boards_ll <= 0 || n_q * n_boards_ll > 0x7FFFFFFFll) { prl_set_error("LBR equity: too many (range, board) pairs in one call"); return PRL_ERR_ARG; }
    const uint16_t* hole_lut = nullptr;  // hold'em: the process-wide (c1, c2) table of the hand evaluator
    if (g.n_hole == 2 && prl_hole_lut_device(&hole_lut) != PRL_OK) return PRL_ERR_HIP;
    int8_t* d_boards = nullptr; uint8_t* d_cls = nullptr; float *d_rg = nullptr, *d_eq = nullptr, *d_out = nullptr; uint16_t* d_list = nullptr;
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
        if (deep) {
            // the first board's classes as index lists: the hands LBR beats (ascending), then the ones it ties with
            std::vector<uint8_t> cls((size_t)g.R);
            LB_TRY(hipMemcpy(cls.data(), d_cls, (size_t)g.R, hipMemcpyDeviceToHost));
            std::vector<uint16_t> lst;
            int n_big = 0, n_eq = 0;
            for (int h = 0; h < g.R; ++h) if (cls[h] == 1) { lst.push_back((uint16_t)h); ++n_big; }
            for (int h = 0; h < g.R; ++h) if (cls[h] == 2) { lst.push_back((uint16_t)h); ++n_eq; }
            lst.resize((size_t)g.R + 8, 0);  // (the streams fetch eight entries at a time)
            LB_TRY(hipMalloc((void**)&d_list, lst.size() * sizeof(uint16_t)));
            LB_TRY(hipMemcpy(d_list, lst.data(), lst.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
            PRL_LAUNCH(prl_k_lbr_board_eq_deep, (int)((items_eq + 63) / 64 < 262144 ? (items_eq + 63) / 64 : 262144), 64, 0, nullptr, g, n_boards, (const uint16_t*)d_list, n_big, n_eq,
                       (const float*)d_rg, n_q, d_eq, hole_lut);
        }
        else PRL_LAUNCH(prl_k_lbr_board_eq, (int)((items_eq + 63) / 64 < 262144 ? (items_eq + 63) / 64 : 262144), 64, 0, nullptr, g, (const int8_t*)d_boards, n_boards, (const uint8_t*)d_cls,
                   (const float*)d_rg, n_q, d_eq, hole_lut);
        if (g.n_to_deal > 2) {  // the deal tree's terms in parallel, then the running sum in board order
            const int n_pc = (int)pc.size(), n_prefix = (int)prl_comb(n_pc - 1, g.n_to_deal - 1);
            const long long items = (long long)n_q * n_prefix;
            PRL_LAUNCH(prl_k_lbr_deep_terms, (int)((items + 255) / 256), 256, (size_t)g.n_cards * 256 * sizeof(float), nullptr, g, (const float*)d_rg, n_q, n_boards, n_prefix, d_eq);
            PRL_LAUNCH(prl_k_lbr_deep_sum, n_q, 64, 0, nullptr, (int)g.n_to_deal, n_boards, (const float*)d_eq, d_out);
        } else
            PRL_LAUNCH(prl_k_lbr_reduce, (n_q + 63) / 64, 64, 0, nullptr, g, n_boards, (const float*)d_rg, n_q, (const float*)d_eq, d_out);
    }
    LB_TRY(hipDeviceSynchronize());
    LB_TRY(hipMemcpy(out_wp, d_out, (size_t)n_q * sizeof(float), hipMemcpyDeviceToHost));
#undef LB_TRY
done:
    (void)hipFree(d_boards); (void)hipFree(d_cls); (void)hipFree(d_rg); (void)hipFree(d_eq); (void)hipFree(d_out); (void)hipFree(d_list);
    return rc;
}
