// Per-board "showdown plans" for 2-hole-card ranges, built once per tree on the GPU.
//
// A plan is everything about a board that the terminal-equity scans need and that never changes during a solve:
//   sh[i]   hand at sorted position i (live hands only, ascending (hand rank, hand index))
//   pos[h]  sorted position of hand h, -1 if a hole card is on the board
//   gs/ge   tie group [gs, ge) of every sorted position (equal hand ranks)
//   cl[c]   for every card c, the sorted positions of the live hands that contain c, ascending (46 entries on a 5-card board)
//   clx     (5-card boards of the 52-card deck only: the fused board pass) the same lists cut into the records its per-card
//           scans consume, one 8-byte record per lane of a 768-lane workgroup: lane t serves list entries 3i .. 3i+2 (i = t % 16)
//           of the (t / 16)-th live card -- see PRL_CLX_* in prl_solver_types.h
// This is the generalisation of the reference's per-terminal `handranks` loop (ValueFiller.py:140-143) to 1326-hand
// ranges: ranks come from the same evaluator as get_hand_rank_all_hands_on_given_boards (prl_handeval.h), computed here
// in-kernel so that no rank table ever round-trips through the host. Plan index n_boards is the "no board" plan
// (hand-index order, one tie group) used by pre-deal fold nodes.
//
// One workgroup (256 lanes) per plan; 2048-key bitonic sort in LDS (keys = rank << 11 | hand, unique -> deterministic).
#include "prl_device.h"
#include "prl_handeval.h"
#include "prl_kernels.h"
#include "prl_solver_types.h"

PRL_GLOBAL void prl_k_plan_build(PrlDevTree T, int n_plans, int16_t* plan_sh, int16_t* plan_pos, int16_t* plan_gs, int16_t* plan_ge,
                                 int16_t* plan_cl, int32_t* plan_nlive, int16_t* plan_hgs, int16_t* plan_hge, uint32_t* plan_clx, int32_t* plan_ndealt,
                                 uint8_t* plan_klh, int16_t* plan_pp) {
    uint32_t* keys = (uint32_t*)prl_smem();  // [2048]
    int* n_live_s = (int*)(keys + 2048);
    const int tid = (int)prl_tid(), nt = (int)prl_nthreads();
    for (int b = (int)prl_bid(); b < n_plans; b += (int)prl_nblocks()) {
        // a row of the board table is a board PREFIX: cards not dealt yet are -1. Only a complete board has hand ranks; a prefix
        // (and the "no board" plan, b == n_boards) orders the live hands by hand index in one tie group -- what fold terminals
        // before the last street need: blockers, no ranks.
        const bool has_row = b < T.n_boards;
        int n_dealt = 0;
        uint32_t s0 = 0, s1 = 0, s2 = 0, s3 = 0;
        unsigned long long on_board = 0ull;
        if (has_row) {
            for (int i = 0; i < T.board_len; ++i) {
                int c = T.boards[(size_t)b * T.board_len + i];
                if (c < 0) continue;
                ++n_dealt;
                uint32_t bit = 1u << (c >> 2);
                int su = c & 3;
                s0 |= su == 0 ? bit : 0u;
                s1 |= su == 1 ? bit : 0u;
                s2 |= su == 2 ? bit : 0u;
                s3 |= su == 3 ? bit : 0u;
                on_board |= 1ull << c;
            }
        }
        const bool has_board = has_row && n_dealt == T.board_len;  // ranks exist
        for (int h = tid; h < 2048; h += nt) {
            uint32_t key = 0xFFFFFFFFu;
            if (h < T.R) {
                int c1 = T.hole[2 * h], c2 = T.hole[2 * h + 1];
                const bool blocked = ((on_board >> c1) & 1ull) || ((on_board >> c2) & 1ull);
                if (!has_board) key = blocked ? 0xFFFFFFFFu : (uint32_t)h;
                else if (!blocked) {
                    uint32_t b1 = 1u << (c1 >> 2), b2 = 1u << (c2 >> 2);
                    int u1 = c1 & 3, u2 = c2 & 3;
                    int32_t r = prl_rank7_masks(s0 | (u1 == 0 ? b1 : 0u) | (u2 == 0 ? b2 : 0u), s1 | (u1 == 1 ? b1 : 0u) | (u2 == 1 ? b2 : 0u),
                                                s2 | (u1 == 2 ? b1 : 0u) | (u2 == 2 ? b2 : 0u), s3 | (u1 == 3 ? b1 : 0u) | (u2 == 3 ? b2 : 0u));
                    key = ((uint32_t)r << 11) | (uint32_t)h;
                }
            }
            keys[h] = key;
        }
        if (tid == 0) *n_live_s = 0;
        prl_sync();
        // bitonic sort, ascending
        for (int k = 2; k <= 2048; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < 2048; i += nt) {
                    int ixj = i ^ j;
                    if (ixj > i) {
                        uint32_t a = keys[i], c = keys[ixj];
                        bool up = (i & k) == 0;
                        if ((a > c) == up) { keys[i] = c; keys[ixj] = a; }
                    }
                }
                prl_sync();
            }
        }
        for (int i = tid; i < 2048; i += nt)
            if (keys[i] != 0xFFFFFFFFu && (i == 2047 || keys[i + 1] == 0xFFFFFFFFu)) *n_live_s = i + 1;
        prl_sync();
        const int n = *n_live_s;
        int16_t* sh = plan_sh + (size_t)b * T.plan_stride;
        int16_t* pos = plan_pos + (size_t)b * T.plan_stride;
        int16_t* gs = plan_gs + (size_t)b * T.plan_stride;
        int16_t* ge = plan_ge + (size_t)b * T.plan_stride;
        int16_t* cl = plan_cl + (size_t)b * T.cl_stride;
        int16_t* hgs = plan_hgs + (size_t)b * T.plan_stride;
        int16_t* hge = plan_hge + (size_t)b * T.plan_stride;
        for (int h = tid; h < T.R; h += nt) { pos[h] = -1; sh[h] = -1; gs[h] = 0; ge[h] = 0; hgs[h] = 0; hge[h] = 0; }
        prl_sync();
        for (int i = tid; i < n; i += nt) {
            const uint32_t key = keys[i];
            const int h = (int)(key & 0x7FFu);
            sh[i] = (int16_t)h;
            pos[h] = (int16_t)i;
            const uint32_t r = has_board ? (key >> 11) : 0u;
            int j = i;
            while (j > 0 && (has_board ? (keys[j - 1] >> 11) : 0u) == r) j--;
            gs[i] = (int16_t)j;
            j = i + 1;
            while (j < n && (has_board ? (keys[j] >> 11) : 0u) == r) j++;
            ge[i] = (int16_t)j;
            hgs[h] = gs[i];
            hge[h] = (int16_t)j;
        }
        for (int c = tid; c < T.n_cards; c += nt) {
            int16_t* row = cl + (size_t)c * (T.n_cards - 1);
            int m = 0;
            for (int i = 0; i < n; ++i) {
                const int h = (int)(keys[i] & 0x7FFu);
                if (T.hole[2 * h] == c || T.hole[2 * h + 1] == c) row[m++] = (int16_t)i;
            }
            for (; m < T.n_cards - 1; ++m) row[m] = -1;
        }
        if (tid == 0) plan_nlive[b] = n;
        prl_sync();
        if (tid == 0) plan_ndealt[b] = n_dealt;
        const int n_t = T.n_cards - 1 - n_dealt;
        // LEVELS engine: the bounds of every hand's tie group inside the lists of its two cards, found once here instead of by a
        // linear search per hand, terminal and pass (prl_terminal_equity_2card)
        if (plan_klh) {
            uint8_t* klh = plan_klh + (size_t)b * T.R * 4;
            for (int h = tid; h < T.R; h += nt) {
                const int i = pos[h];
                uint8_t out[4] = {0, 0, 0, 0};
                if (i >= 0 && has_board) {
                    const int g0 = gs[i], g1 = ge[i];
                    for (int k = 0; k < 2; ++k) {
                        const int16_t* row = cl + (size_t)T.hole[2 * h + k] * (T.n_cards - 1);
                        int lo = 0;
                        while (lo < n_t && row[lo] < g0) lo++;
                        int hi = lo;
                        while (hi < n_t && row[hi] < g1) hi++;
                        out[2 * k] = (uint8_t)lo;
                        out[2 * k + 1] = (uint8_t)hi;
                    }
                }
                for (int k = 0; k < 4; ++k) klh[4 * h + k] = out[k];
            }
        }
        // records of the fused board pass (row16 order with 3 entries per lane: lists of 33..48 entries, at most 48 live cards)
        if (plan_clx && has_board && T.n_cards - T.board_len <= PRL_CLX_SLOTS && n_t > 32 && n_t <= 48) {
            uint32_t* clx = plan_clx + (size_t)b * PRL_CLX_WORDS;
            const uint32_t inv = (uint32_t)PRL_CLX_ZERO_POS | PRL_CLX_HEAD | PRL_CLX_TAIL;
            for (int i = tid; i < PRL_CLX_WORDS / 2; i += nt) {  // slots without a card, entries past the end of a list
                clx[2 * i] = inv | (inv << 16);
                clx[2 * i + 1] = inv | ((uint32_t)(i & 15) << 16) | ((uint32_t)(i & 15) << 20) | (63u << 24);
            }
            prl_sync();
            for (int c = tid; c < T.n_cards; c += nt) {
                if ((on_board >> c) & 1ull) continue;
                const int slot = c - prl_popc64(on_board & ((1ull << c) - 1ull));
                const int16_t* row = cl + (size_t)c * (T.n_cards - 1);
                auto head = [&](int e) { return e == 0 || gs[row[e]] != gs[row[e - 1]]; };       // first of its tie group in this list
                auto tail = [&](int e) { return e == n_t - 1 || gs[row[e + 1]] != gs[row[e]]; };
                auto entry = [&](int e) -> uint32_t {
                    if (e >= n_t) return inv;
                    const int h = (int)(keys[row[e]] & 0x7FFu);
                    return (uint32_t)row[e] | (head(e) ? PRL_CLX_HEAD : 0u) | (tail(e) ? PRL_CLX_TAIL : 0u) |
                           (T.hole[2 * h] == c ? PRL_CLX_LOWER : 0u) | PRL_CLX_VALID;
                };
                for (int i = 0; i < 16; ++i) {
                    int la = i, lb = i;
                    if (3 * i < n_t) { int e = 3 * i; while (!head(e)) --e; la = e / 3; }
                    if (3 * i + 2 < n_t) { int e = 3 * i + 2; while (!tail(e)) ++e; lb = e / 3; }
                    clx[(slot * 16 + i) * 2] = entry(3 * i) | (entry(3 * i + 1) << 16);
                    clx[(slot * 16 + i) * 2 + 1] = entry(3 * i + 2) | ((uint32_t)la << 16) | ((uint32_t)lb << 20) | ((uint32_t)c << 24);
                }
            }
            prl_sync();
        }
        // position-domain plan of the single-deal fused engine (prl_solver_types.h: PRL_PP_*)
        if (plan_pp && has_board && T.R == PRL_PP_R && n <= PRL_PP_NPAD - 7) {
            int16_t* pp = plan_pp + (size_t)b * PRL_PP_STRIDE;
            for (int i = tid; i < PRL_PP_NPAD; i += nt) {
                const bool lv = i < n;
                const int h = lv ? (int)(keys[i] & 0x7FFu) : 0;
                pp[i] = (int16_t)h;  // (positions n .. R-1 are overwritten below)
                pp[PRL_PP_OFF_GM1 + i] = (int16_t)(lv && gs[i] > 0 ? gs[i] - 1 : (int)PRL_CLX_ZERO_POS);
                pp[PRL_PP_OFF_EM1 + i] = (int16_t)(lv ? ge[i] - 1 : (int)PRL_CLX_ZERO_POS);
                pp[PRL_PP_OFF_CC + i] = (int16_t)(lv ? (T.hole[2 * h] & 0xFF) | ((T.hole[2 * h + 1] & 0xFF) << 8) : 0);
            }
            for (int i = PRL_PP_OFF_CC + PRL_PP_NPAD + tid; i < PRL_PP_STRIDE; i += nt) pp[i] = 0;
            prl_sync();
            // the hands the board blocks, in hand-index order, after the live ones: sh is a permutation of all R hands
            for (int h = tid; h < T.R; h += nt) {
                if (pos[h] >= 0) continue;
                int before = 0;
                for (int g = 0; g < h; ++g) before += pos[g] < 0;
                pp[n + before] = (int16_t)h;
            }
            prl_sync();
        }
    }
}

void prl_launch_plan_build(const PrlDevTree& T, int n_plans, int16_t* plan_sh, int16_t* plan_pos, int16_t* plan_gs, int16_t* plan_ge,
                           int16_t* plan_cl, int32_t* plan_nlive, int16_t* plan_hgs, int16_t* plan_hge, uint32_t* plan_clx, int32_t* plan_ndealt,
                           uint8_t* plan_klh, int16_t* plan_pp, void* stream) {
    int grid = n_plans < 32768 ? n_plans : 32768;
    PRL_LAUNCH(prl_k_plan_build, grid, 256, 2048 * sizeof(uint32_t) + 16, stream, T, n_plans, plan_sh, plan_pos, plan_gs, plan_ge, plan_cl,
               plan_nlive, plan_hgs, plan_hge, plan_clx, plan_ndealt, plan_klh, plan_pp);
}
