// Level-synchronous public-tree kernels ("engine G"): every per-node vector of the reference lives in HBM
// (reach / ev / ev_br : [node][seat][hand]) and one kernel handles one tree level (or all terminals / all nodes of one
// seat). Works for every supported tree (Leduc family and Hold'em-sized ranges) and keeps the reference's node
// attribute protocol observable (node.reach_probs, node.ev, node.ev_br, node.data["regret"] ...). It is the general
// path and the on-GPU cross-check for the fused board-block kernels (prl_fhp_kernels.hip), which keep the per-node
// vectors on chip instead.
//
// Reference semantics implemented here (all float32 unless the strategy is float64, see prl_solver_types.h):
//   k_reach_*        StrategyFiller._update_reach_probs          StrategyFiller.py:118-146 (+ chance weights :148-169)
//   k_terminal_*     ValueFiller terminal branch                  ValueFiller.py:34-62, :103-175
//   k_ev_level       ValueFiller non-terminal branch              ValueFiller.py:64-93
//   k_exploitability ValueFiller epilogue at the root             ValueFiller.py:96-101
//   k_regret_strategy  _CFRBase._compute_regrets + variants       _CFRBase.py:146-185; VanillaCFR.py:26-52; CFRPlus.py:37-63;
//                                                                 LinearCFR.py:27-51
//   k_average        _add_strategy_to_average                     VanillaCFR.py:54-77; CFRPlus.py:65-87; LinearCFR.py:53-76
// Summation orders are the reference's (NumPy pairwise for contiguous inner axes, running add for outer axes,
// SURVEY.md Appendix A); for 2-hole-card ranges the canonical wave-64 scan order of DESIGN.md.
#include "prl_device.h"
#include "prl_handeval.h"
#include <type_traits>

#include "prl_kernels.h"
#include "prl_solver_types.h"
#include "prl_st.h"

// ---------------------------------------------------------------------------------------------------------------------
// summation helpers
// ---------------------------------------------------------------------------------------------------------------------
// NumPy float32 pairwise sum for n <= 128 (one block of the pairwise scheme), elements produced by get(i)
template <class F>
PRL_DEV PRL_INLINE float prl_np_sum(F get, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = res + get(i);
        return res;
    }
    float r0 = get(0), r1 = get(1), r2 = get(2), r3 = get(3), r4 = get(4), r5 = get(5), r6 = get(6), r7 = get(7);
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
        r0 = r0 + get(i);
        r1 = r1 + get(i + 1);
        r2 = r2 + get(i + 2);
        r3 = r3 + get(i + 3);
        r4 = r4 + get(i + 4);
        r5 = r5 + get(i + 5);
        r6 = r6 + get(i + 6);
        r7 = r7 + get(i + 7);
    }
    float res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < n; ++i) res = res + get(i);
    return res;
}

// canonical wave-64 inclusive scan (every lane of the wave must call it)
PRL_DEV PRL_INLINE float prl_wave_scan(float v) { return prl_wave_scan_canonical(v); }

// Canonical chunked exclusive prefix of y[0..n) (LDS) into P[0..n] (LDS): 64-wide scans, sequential chunk carries.
// tot / carry: LDS scratch of >= n_chunks + 1 floats each. Must be called by all threads of the block.
PRL_DEV PRL_INLINE void prl_block_prefix(const float* y, int n, float* P, float* tot, float* carry) {
    const int n_chunks = (n + 63) >> 6;
    const int wave = (int)(prl_tid() >> 6), n_waves = (int)(prl_nthreads() >> 6), lane = (int)prl_lane();
    for (int k = wave; k < n_chunks; k += n_waves) {
        int j = 64 * k + lane;
        float v = j < n ? y[j] : 0.f;
        v = prl_wave_scan(v);
        if (j + 1 <= n) P[j + 1] = v;  // inclusive value of position j, carry added below
        if (lane == 63) tot[k] = v;
    }
    prl_sync();
    if (n_chunks < 64) {  // the sequential chain of chunk carries in wave 0's registers: lane k holds tot[k], one lane broadcast and one dependent add per chunk
        if (wave == 0) {
            const float tv = lane < n_chunks ? tot[lane] : 0.f;
            float c = 0.f, mine = 0.f;
            for (int k = 0; k < n_chunks; ++k) {
                if (lane == k) mine = c;
                c = c + prl_readlane(tv, k);
            }
            if (lane == n_chunks) mine = c;
            if (lane <= n_chunks) carry[lane] = mine;
        }
    } else if (prl_tid() == 0) {
        float c = 0.f;
        for (int k = 0; k < n_chunks; ++k) {
            carry[k] = c;
            c = c + tot[k];
        }
        carry[n_chunks] = c;
    }
    prl_sync();
    // P[j] = (j % 64 == 0) ? carry[j / 64] : carry[j / 64] + incl[j - 1]
    for (int j = (int)prl_tid(); j <= n; j += (int)prl_nthreads()) {
        int k = j >> 6;
        float c = carry[k];
        P[j] = (j & 63) == 0 ? c : c + P[j];  // P[j] currently holds incl[j-1] (same chunk as j because j % 64 != 0)
    }
    prl_sync();
}

// ---------------------------------------------------------------------------------------------------------------------
// strategy fill, reach push-down
// ---------------------------------------------------------------------------------------------------------------------
PRL_GLOBAL void prl_k_fill_uniform(PrlDevTree T, PrlDevState S, const int32_t* __restrict__ col_node) {
    const size_t total = (size_t)T.n_cols * T.R;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        int col = (int)(t / T.R);
        int node = col_node[col];
        S.strategy[t] = 1.0 / (double)T.n_children[node];  // np.full(..., 1.0 / float(n_actions)) -> float64
        if (t % T.R == 0) S.strat_f64[node] = 1;
    }
}

PRL_DEV PRL_INLINE void prl_reach_root_body(const PrlDevTree& T, const PrlDevState& S) {
    const float r0 = (float)(1.0 / (double)T.R);  // PublicTree.py:122-124
    for (int t = (int)(prl_bid() * prl_nthreads() + prl_tid()); t < 2 * T.R; t += (int)(prl_nblocks() * prl_nthreads())) S.reach[t] = r0;
}

PRL_DEV PRL_INLINE void prl_reach_level_body(const PrlDevTree& T, const PrlDevState& S, int level_begin, int level_count) {
    const size_t total = (size_t)level_count * T.R;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int node = T.level_nodes[level_begin + (int)(t / T.R)];
        const int h = (int)(t % T.R);
        const int par = T.parent[node];
        const float* rp = S.reach + prl_vidx(T, par, 0);
        float* rc = S.reach + prl_vidx(T, node, 0);
        if (T.kind[par] == PRL_NODE_DECISION) {
            const int a = T.actor[par];
            const double s = S.strategy[prl_cidx(T, T.first_col[par] + T.child_idx[node]) + h];
            const float ra = rp[(size_t)a * T.R + h];
            const float v = S.strat_f64[par] ? (float)(s * (double)ra) : (float)s * ra;
            rc[(size_t)a * T.R + h] = v;
            rc[(size_t)(1 - a) * T.R + h] = rp[(size_t)(1 - a) * T.R + h];
        } else {  // chance: both seats' reach is scaled by the board probability, 0 for blocked hands
            const float w = prl_hand_blocked(T, h, T.board_id[node]) ? 0.f : T.chance_w[par];
            rc[h] = rp[h] * w;
            rc[(size_t)T.R + h] = rp[(size_t)T.R + h] * w;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// terminal values, 1-hole-card games (reference arithmetic order, R <= 128)
// ---------------------------------------------------------------------------------------------------------------------
PRL_DEV PRL_INLINE int prl_rank_1card(const PrlDevTree& T, int h, int board_card) {
    return prl_rank_leduc(h, board_card, T.n_suits, T.rank_rule == 1 ? 10000 : 100);
}

// ValueFiller._get_call_eq_final_street (ValueFiller.py:145-155): running float32 sum over ascending h_opp
PRL_DEV PRL_INLINE float prl_showdown_1card(const PrlDevTree& T, const float* x, int board_card, int h) {
    float e = 0.f;
    if (h == board_card) return e;
    const int rh = prl_rank_1card(T, h, board_card);
    for (int ho = 0; ho < T.R; ++ho) {
        if (ho == h || ho == board_card) continue;
        const int ro = prl_rank_1card(T, ho, board_card);
        if (rh > ro) e = e + x[ho];
        else if (rh < ro) e = e - x[ho];
    }
    return e;
}

PRL_DEV PRL_INLINE void prl_terminal_1card_body(const PrlDevTree& T, const PrlDevState& S, const int32_t* __restrict__ term_nodes, int n_term) {
    const size_t total = (size_t)n_term * 2 * T.R;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int node = term_nodes[t / (2 * (size_t)T.R)];
        const int p = (int)((t / T.R) & 1);
        const int h = (int)(t % T.R);
        const float* x = S.reach + prl_vidx(T, node, 1 - p);
        const int bid = T.board_id[node];
        const int bc = bid >= 0 ? (int)T.boards[(size_t)bid * T.board_len] : -1;
        const bool fold = T.kind[node] == PRL_NODE_TERM_FOLD;
        float e;
        if (fold) {  // ValueFiller.py:103-125
            float total_x = prl_np_sum([&](int i) { return x[i]; }, T.R);
            e = (total_x - x[h]) * T.eq_const;
            if (T.acted_last[node] == p) e = -e;
        } else if (bc >= 0) {
            e = prl_showdown_1card(T, x, bc, h) * T.eq_const;
        } else {  // all-in before the board card: mean over the N-2 boards (ValueFiller.py:160-175)
            float acc = 0.f;
            for (int c = 0; c < T.n_cards; ++c) acc = acc + prl_showdown_1card(T, x, c, h) * T.eq_const;
            e = acc / (float)(T.n_cards - 2);
        }
        if (h == bc) e = 0.f;  // ValueFiller.py:57-59
        const float v = (e * (float)T.main_pot[node]) / 2.f;
        S.ev[prl_vidx(T, node, p) + h] = v;
        S.ev_br[prl_vidx(T, node, p) + h] = v;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// terminal values, 2-hole-card games: one workgroup per terminal node, canonical scan order
// ---------------------------------------------------------------------------------------------------------------------
#define PRL_T2_YPAD 1344  // >= R = 1326, multiple of 64

PRL_DEV PRL_INLINE size_t prl_t2_smem_floats(int n_cards) { return (size_t)PRL_T2_YPAD + (PRL_T2_YPAD + 8) + 64 + 64 + (size_t)n_cards * 65; }

// equity of every hand against the opponent vector x (global, hand domain) on plan `plan`; result -> eq_out (global)
PRL_DEV PRL_INLINE void prl_terminal_equity_2card(const PrlDevTree& T, const float* x, int plan, bool showdown, float* smem,
                                                  float sign, float pot, float* out_ev, float* out_ev_br) {
    float* y = smem;
    float* P = y + PRL_T2_YPAD;
    float* tot = P + PRL_T2_YPAD + 8;
    float* carry = tot + 64;
    float* Q = carry + 64;  // [n_cards][65]
    const int16_t* sh = T.plan_sh + (size_t)plan * T.plan_stride;
    const int16_t* pos = T.plan_pos + (size_t)plan * T.plan_stride;
    const int16_t* gs = T.plan_gs + (size_t)plan * T.plan_stride;
    const int16_t* ge = T.plan_ge + (size_t)plan * T.plan_stride;
    const int16_t* cl = T.plan_cl + (size_t)plan * T.cl_stride;
    const uint8_t* klh = T.plan_klh ? T.plan_klh + (size_t)plan * T.R * 4 : nullptr;
    const int n = T.plan_nlive[plan];
    const int n_t = T.n_cards - 1 - T.plan_ndealt[plan];
    const int tid = (int)prl_tid(), nt = (int)prl_nthreads();
    for (int i = tid; i < n; i += nt) y[i] = x[sh[i]];
    prl_sync();
    prl_block_prefix(y, n, P, tot, carry);
    {   // per-card scans in the row16 order (prl_device.h): a wave takes four cards at a time, one per row of 16 lanes
        const int wave = tid >> 6, n_waves = nt >> 6, lane = tid & 63, row = lane >> 4, l16 = lane & 15;
        const int E = (n_t + 15) >> 4;  // entries per lane, <= 4
        for (int c0 = 4 * wave; c0 < T.n_cards; c0 += 4 * n_waves) {
            const int c = c0 + row;
            const bool okc = c < T.n_cards;
            const int16_t* lst = cl + (size_t)(okc ? c : 0) * (T.n_cards - 1);
            float l[4] = {0.f, 0.f, 0.f, 0.f};
            float run = 0.f;
            for (int k = 0; k < E; ++k) {
                const int e = l16 * E + k;
                const int q = (okc && e < n_t) ? (int)lst[e] : -1;
                const float v = q >= 0 ? y[q] : 0.f;
                run = k == 0 ? v : run + v;
                l[k] = run;
            }
            const float t = prl_row16_scan(run);
            const float carry = prl_dpp_row_shr<1>(t);
            if (okc) {
                if (l16 == 0) Q[c * 65] = 0.f;
                for (int k = 0; k < E; ++k) Q[c * 65 + l16 * E + k + 1] = carry + l[k];
            }
        }
    }
    prl_sync();
    const float Tsum = P[n];
    for (int h = tid; h < T.R; h += nt) {
        const int i = pos[h];
        float e = 0.f;
        if (i >= 0) {
            const int c1 = T.hole[2 * h], c2 = T.hole[2 * h + 1];
            if (!showdown) {
                const float m = Q[c1 * 65 + n_t] + Q[c2 * 65 + n_t];
                e = Tsum - (m - x[h]);
            } else {
                const int g0 = gs[i], g1 = ge[i];
                const float G = P[g0] - (Tsum - P[g1]);
                float K[2];
                for (int k = 0; k < 2; ++k) {
                    const int c = k == 0 ? c1 : c2;
                    int lo, hi;
                    if (klh) {  // precomputed with the plan (prl_k_plan_build)
                        lo = klh[4 * h + 2 * k];
                        hi = klh[4 * h + 2 * k + 1];
                    } else {
                        const int16_t* row = cl + (size_t)c * (T.n_cards - 1);
                        lo = 0;
                        while (lo < n_t && row[lo] < g0) lo++;
                        hi = lo;
                        while (hi < n_t && row[hi] < g1) hi++;
                    }
                    K[k] = Q[c * 65 + lo] - (Q[c * 65 + n_t] - Q[c * 65 + hi]);
                }
                e = G - (K[0] + K[1]);
            }
            e = e * T.eq_const;
            e = sign * e;  // exact (+-1)
        }
        const float v = (e * pot) / 2.f;
        out_ev[h] = v;
        out_ev_br[h] = v;
    }
    prl_sync();
}

// one workgroup per (terminal, seat): the two seats' equities of a terminal are independent
PRL_GLOBAL void prl_k_terminal_2card(PrlDevTree T, PrlDevState S, const int32_t* __restrict__ term_nodes, int n_term) {
    float* smem = (float*)prl_smem();
    for (int ti = (int)prl_bid(); ti < 2 * n_term; ti += (int)prl_nblocks()) {
        const int node = term_nodes[ti >> 1];
        const int p = ti & 1;
        const int bid = T.board_id[node];
        const bool fold = T.kind[node] == PRL_NODE_TERM_FOLD;
        const float pot = (float)T.main_pot[node];
        float* oe = S.ev + prl_vidx(T, node, p);
        float* ob = S.ev_br + prl_vidx(T, node, p);
        // (showdown terminals before the deal do not exist here: prl_build_flat_tree refuses such 2-card trees)
        const float sign = (fold && T.acted_last[node] == p) ? -1.f : 1.f;
        prl_terminal_equity_2card(T, S.reach + prl_vidx(T, node, 1 - p), bid < 0 ? T.n_boards : bid, !fold, smem, sign, pot, oe, ob);
    }
}

// the run-out chains of the per-street engine (prl_st.h): a workgroup of 256 lanes takes a BUNDLE of showdowns on one complete board for one seat. What
// prl_terminal_equity_2card reads of the board's plan per showdown -- the sorted order, the per-card position lists, a hand's tie group and its cards' bounds --
// it holds in registers across the bundle, so a showdown costs one gather of the opponent's reach (from the leaf of the all-in call above its chain, times
// the chain's outcome weights) and the scans; the arithmetic is that function's, statement for statement.
#define PRL_CT_NT 256
#define PRL_CT_K ((PRL_T2_YPAD + PRL_CT_NT - 1) / PRL_CT_NT)  // hands per lane (6)
#if defined(PRL_EMU)
#define PRL_CT_LB
#define PRL_CT_KEEP_PACKED(x) do { } while (0)
#else
#ifndef PRL_CT_WAVES
#define PRL_CT_WAVES 5
#endif
#define PRL_CT_LB __launch_bounds__(PRL_CT_NT, PRL_CT_WAVES)  // 6 waves per SIMD = the 6 workgroups per CU the 24.8 KB of LDS admit: the kernel is a chain of latencies
// the plan words stay packed across the bundle's loop: without this the compiler hoists every address it can derive from them out of the loop (48 registers)
#define PRL_CT_KEEP_PACKED(x) asm volatile("" : "+v"(x))
#endif
struct PrlCtTree {  // what this kernel reads of the forest's PrlDevTree (the whole struct by value costs ~100 scalar registers)
    const int16_t *plan_sh, *plan_pos, *plan_gs, *plan_ge, *plan_cl, *hole;
    const uint8_t* plan_klh;
    const int32_t *plan_nlive, *plan_ndealt, *main_pot;
    int32_t plan_stride, cl_stride, R, n_cards;
    float eq_const;
};
PRL_GLOBAL void PRL_CT_LB prl_k_st_chain_terminals(PrlCtTree T, const PrlStChainTerm* __restrict__ terms, const PrlStChainBundle* __restrict__ bundles,
                                                   int n_bundles, PrlStChainIo io, PrlStChainDev cd, float* __restrict__ ev, PrlStRowMap map, int seat_mask) {
    float* y = (float*)prl_smem();
    float* P = y + PRL_T2_YPAD;
    float* tot = P + PRL_T2_YPAD + 8;
    float* carry = tot + 64;
    float* Q = carry + 64;  // [n_cards][65]
    const int tid = (int)prl_tid(), wave = tid >> 6, lane = tid & 63, row = lane >> 4, l16 = lane & 15;
    const int n_seats = seat_mask == 3 ? 2 : 1;
    for (int bi = (int)prl_bid(); bi < n_seats * n_bundles; bi += (int)prl_nblocks()) {
        const PrlStChainBundle bd = bundles[bi / n_seats];
        const int p = n_seats == 2 ? bi % 2 : (seat_mask == 1 ? 0 : 1);
        const int plan = bd.plan;
        const int16_t* sh = T.plan_sh + (size_t)plan * T.plan_stride;
        const int16_t* pos = T.plan_pos + (size_t)plan * T.plan_stride;
        const int16_t* gs = T.plan_gs + (size_t)plan * T.plan_stride;
        const int16_t* ge = T.plan_ge + (size_t)plan * T.plan_stride;
        const int16_t* cl = T.plan_cl + (size_t)plan * T.cl_stride;
        const uint8_t* klh = T.plan_klh + (size_t)plan * T.R * 4;
        const int n = T.plan_nlive[plan];
        const int n_t = T.n_cards - 1 - T.plan_ndealt[plan];
        const int E = (n_t + 15) >> 4;  // entries per lane of a card's list, <= 4
        // ---- the plan's share of this lane, once per bundle (packed: the kernel wants 6 waves per SIMD) ----
        uint32_t shr[PRL_CT_K / 2];  // two 16-bit sorted-order entries per word, 0xFFFF = none
#pragma unroll
        for (int k = 0; k < PRL_CT_K / 2; ++k) {
            const int i0 = tid + (2 * k) * PRL_CT_NT, i1 = tid + (2 * k + 1) * PRL_CT_NT;
            shr[k] = (i0 < n ? (uint32_t)(uint16_t)sh[i0] : 0xFFFFu) | ((i1 < n ? (uint32_t)(uint16_t)sh[i1] : 0xFFFFu) << 16);
        }
        uint32_t lq[4][2];  // the lane's <= 4 entries of its card's position list per round, 0xFFFF = none
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int c = 4 * wave + 16 * r + row;
            const bool okc = c < T.n_cards;
            const int16_t* lst = cl + (size_t)(okc ? c : 0) * (T.n_cards - 1);
            uint32_t q[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int e = l16 * E + k;
                q[k] = (okc && k < E && e < n_t) ? (uint32_t)(uint16_t)lst[e] : 0xFFFFu;
            }
            lq[r][0] = q[0] | (q[1] << 16);
            lq[r][1] = q[2] | (q[3] << 16);
        }
        uint32_t f_g[PRL_CT_K], f_c[PRL_CT_K], f_k[PRL_CT_K];  // tie group (begin | end << 16); cards (c1 | c2 << 8 | live << 16); the cards' list bounds
#pragma unroll
        for (int k = 0; k < PRL_CT_K; ++k) {
            const int h = tid + k * PRL_CT_NT;
            f_g[k] = 0u; f_c[k] = 0u; f_k[k] = 0u;
            if (h < T.R) {
                const int i = pos[h];
                if (i >= 0) {
                    f_g[k] = (uint32_t)(uint16_t)gs[i] | ((uint32_t)(uint16_t)ge[i] << 16);
                    f_c[k] = (uint32_t)T.hole[2 * h] | ((uint32_t)T.hole[2 * h + 1] << 8) | (1u << 16);
                    f_k[k] = (uint32_t)klh[4 * h] | ((uint32_t)klh[4 * h + 1] << 8) | ((uint32_t)klh[4 * h + 2] << 16) | ((uint32_t)klh[4 * h + 3] << 24);
                }
            }
        }
        // ---- the bundle's showdowns; the next one's reach is on its way while this one is scanned ----
        PrlStChainTerm tm = terms[bd.first];
        float xr[PRL_CT_K];
        {
            const float* x = io.src[tm.src_street] + ((size_t)tm.src_slot * 2 + (1 - p)) * T.R;
#pragma unroll
            for (int k = 0; k < PRL_CT_K; ++k) {
                const uint32_t q = (shr[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                xr[k] = q != 0xFFFFu ? x[q] : 0.f;
            }
        }
        for (int t = 0; t < bd.count; ++t) {
#pragma unroll
            for (int k = 0; k < PRL_CT_K; ++k) { PRL_CT_KEEP_PACKED(f_g[k]); PRL_CT_KEEP_PACKED(f_c[k]); PRL_CT_KEEP_PACKED(f_k[k]); }
#pragma unroll
            for (int r = 0; r < 4; ++r) { PRL_CT_KEEP_PACKED(lq[r][0]); PRL_CT_KEEP_PACKED(lq[r][1]); }
#pragma unroll
            for (int k = 0; k < PRL_CT_K / 2; ++k) PRL_CT_KEEP_PACKED(shr[k]);
            float* out = ev + ((size_t)tm.node * 2 + p) * T.R;  // (prl_vidx)
            int vec_mask = 1;
            if (tm.kid >= 0) {  // a chain root: its value is a row of its street (every vector of the row that is seat p's)
                out = io.val[cd.street[tm.kid]] + (size_t)cd.val_slot[tm.kid] * map.width * T.R;
                vec_mask = 0;
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    if (v < map.width && map.seat[v] == p) vec_mask |= 1 << v;
            }
            const float pot = (float)T.main_pot[tm.node];
            const int n_w = tm.n_w;
            const float w0 = tm.w[0], w1 = tm.w[1], w2 = tm.w[2];
#pragma unroll
            for (int k = 0; k < PRL_CT_K; ++k) {
                if (((shr[k >> 1] >> (16 * (k & 1))) & 0xFFFFu) == 0xFFFFu) continue;
                float v = xr[k];
                if (n_w > 0) v = v * w0;  // (the level kernels' reach walk: the root's outcome weight, then one chance node after the other)
                if (n_w > 1) v = v * w1;
                if (n_w > 2) v = v * w2;
                y[tid + k * PRL_CT_NT] = v;
            }
            if (t + 1 < bd.count) {
                tm = terms[bd.first + t + 1];
                const float* x = io.src[tm.src_street] + ((size_t)tm.src_slot * 2 + (1 - p)) * T.R;
#pragma unroll
                for (int k = 0; k < PRL_CT_K; ++k) {
                    const uint32_t q = (shr[k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                    xr[k] = q != 0xFFFFu ? x[q] : 0.f;
                }
            }
            prl_sync();
            prl_block_prefix(y, n, P, tot, carry);
#pragma unroll
            for (int r = 0; r < 4; ++r) {  // per-card scans in the row16 order: a wave takes four cards at a time, one per row of 16 lanes
                if (4 * wave + 16 * r >= T.n_cards) continue;
                const int c = 4 * wave + 16 * r + row;
                const bool okc = c < T.n_cards;
                float l[4] = {0.f, 0.f, 0.f, 0.f};
                float run = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (k >= E) continue;
                    const uint32_t q = (lq[r][k >> 1] >> (16 * (k & 1))) & 0xFFFFu;
                    const float v = q != 0xFFFFu ? y[q] : 0.f;
                    run = k == 0 ? v : run + v;
                    l[k] = run;
                }
                const float tt = prl_row16_scan(run);
                const float cy = prl_dpp_row_shr<1>(tt);
                if (okc) {
                    if (l16 == 0) Q[c * 65] = 0.f;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (k < E) Q[c * 65 + l16 * E + k + 1] = cy + l[k];
                }
            }
            prl_sync();
            const float Tsum = P[n];
#pragma unroll
            for (int k = 0; k < PRL_CT_K; ++k) {
                const int h = tid + k * PRL_CT_NT;
                if (h >= T.R) continue;
                float e = 0.f;
                if ((f_c[k] >> 16) & 1u) {
                    const int g0 = (int)(f_g[k] & 0xFFFFu), g1 = (int)(f_g[k] >> 16);
                    const float G = P[g0] - (Tsum - P[g1]);
                    const int c1 = (int)(f_c[k] & 0xFFu), c2 = (int)((f_c[k] >> 8) & 0xFFu);
                    const int lo1 = (int)(f_k[k] & 0xFFu), hi1 = (int)((f_k[k] >> 8) & 0xFFu), lo2 = (int)((f_k[k] >> 16) & 0xFFu), hi2 = (int)(f_k[k] >> 24);
                    const float K0 = Q[c1 * 65 + lo1] - (Q[c1 * 65 + n_t] - Q[c1 * 65 + hi1]);
                    const float K1 = Q[c2 * 65 + lo2] - (Q[c2 * 65 + n_t] - Q[c2 * 65 + hi2]);
                    e = G - (K0 + K1);
                    e = e * T.eq_const;
                }
                const float v = (e * pot) / 2.f;
#pragma unroll
                for (int u = 0; u < 4; ++u)
                    if ((vec_mask >> u) & 1) out[(size_t)u * T.R + h] = v;
            }
            prl_sync();
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------------
// EV / best-response pull-up
// ---------------------------------------------------------------------------------------------------------------------
// canonical chance sum: running adds nested as blocks of 32 children, groups of 32 blocks, then the groups
PRL_DEV PRL_INLINE float prl_chance_sum(const PrlDevTree& T, const float* arr, int node, int p, int h) {
    const int A = T.n_children[node];
    const int32_t* ch = T.child_list + T.child_start[node];
    float total = 0.f;
    for (int g0 = 0, gi = 0; g0 < A; g0 += PRL_CHANCE_BLOCK * PRL_CHANCE_BLOCK, ++gi) {
        float gsum = 0.f;
        for (int b0 = g0, bi = 0; b0 < A && b0 < g0 + PRL_CHANCE_BLOCK * PRL_CHANCE_BLOCK; b0 += PRL_CHANCE_BLOCK, ++bi) {
            float bsum = arr[prl_vidx(T, ch[b0], p) + h];
            for (int i = b0 + 1; i < A && i < b0 + PRL_CHANCE_BLOCK; ++i) bsum = bsum + arr[prl_vidx(T, ch[i], p) + h];
            gsum = bi == 0 ? bsum : gsum + bsum;
        }
        total = gi == 0 ? gsum : total + gsum;
    }
    return total;
}

PRL_DEV PRL_INLINE void prl_ev_level_body(const PrlDevTree& T, const PrlDevState& S, int level_begin, int level_count) {
    const size_t total = (size_t)level_count * T.R;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int node = T.level_nodes[level_begin + (int)(t / T.R)];
        const int h = (int)(t % T.R);
        const int kind = T.kind[node];
        if (kind >= PRL_NODE_TERM_FOLD) continue;
        if (kind == PRL_NODE_CHANCE) {  // ValueFiller.py:76-78
            for (int p = 0; p < 2; ++p) {
                S.ev[prl_vidx(T, node, p) + h] = prl_chance_sum(T, S.ev, node, p, h);
                S.ev_br[prl_vidx(T, node, p) + h] = prl_chance_sum(T, S.ev_br, node, p, h);
            }
            continue;
        }
        const int A = T.n_children[node];
        const int32_t* ch = T.child_list + T.child_start[node];
        const int pl = T.actor[node], op = 1 - pl;
        const int col0 = T.first_col[node];
        float ev_pl;
        if (S.strat_f64[node]) {  // float64 strategy: float64 products and sum, rounded on store (ValueFiller.py:87)
            double acc = 0.;
            for (int i = 0; i < A; ++i) {
                double prod = S.strategy[prl_cidx(T, col0 + i) + h] * (double)S.ev[prl_vidx(T, ch[i], pl) + h];
                acc = i == 0 ? prod : acc + prod;
            }
            ev_pl = (float)acc;
        } else {
            float acc = 0.f;
            for (int i = 0; i < A; ++i) {
                float prod = (float)S.strategy[prl_cidx(T, col0 + i) + h] * S.ev[prl_vidx(T, ch[i], pl) + h];
                acc = i == 0 ? prod : acc + prod;
            }
            ev_pl = acc;
        }
        float s = S.ev[prl_vidx(T, ch[0], op) + h], sb = S.ev_br[prl_vidx(T, ch[0], op) + h];
        float mx = S.ev_br[prl_vidx(T, ch[0], pl) + h];
        int arg = 0;
        for (int i = 1; i < A; ++i) {
            s = s + S.ev[prl_vidx(T, ch[i], op) + h];
            sb = sb + S.ev_br[prl_vidx(T, ch[i], op) + h];
            float v = S.ev_br[prl_vidx(T, ch[i], pl) + h];
            if (v > mx) { mx = v; arg = i; }
        }
        S.ev[prl_vidx(T, node, pl) + h] = ev_pl;
        S.ev[prl_vidx(T, node, op) + h] = s;
        S.ev_br[prl_vidx(T, node, op) + h] = sb;
        S.ev_br[prl_vidx(T, node, pl) + h] = mx;
        if (S.br_idx) S.br_idx[(size_t)node * T.R + h] = arg;
    }
}

// root exploitability: sum_h (ev_br - ev) * reach  (ValueFiller.py:96-101). One workgroup.
PRL_DEV PRL_INLINE void prl_exploitability_body(const PrlDevTree& T, const PrlDevState& S, float* out2) {
    float* smem = (float*)prl_smem();
    if (T.n_hole == 1) {
        if (prl_tid() < 2) {
            const int p = (int)prl_tid();
            const float* ev = S.ev + prl_vidx(T, 0, p);
            const float* eb = S.ev_br + prl_vidx(T, 0, p);
            const float* rc = S.reach + prl_vidx(T, 0, p);
            out2[p] = prl_np_sum([&](int h) { return eb[h] * rc[h] - ev[h] * rc[h]; }, T.R);
        }
        return;
    }
    float* y = smem;
    float* P = y + PRL_T2_YPAD;
    float* tot = P + PRL_T2_YPAD + 8;
    float* carry = tot + 64;
    for (int p = 0; p < 2; ++p) {
        const float* ev = S.ev + prl_vidx(T, 0, p);
        const float* eb = S.ev_br + prl_vidx(T, 0, p);
        const float* rc = S.reach + prl_vidx(T, 0, p);
        for (int h = (int)prl_tid(); h < T.R; h += (int)prl_nthreads()) y[h] = eb[h] * rc[h] - ev[h] * rc[h];
        prl_sync();
        prl_block_prefix(y, T.R, P, tot, carry);
        if (prl_tid() == 0) out2[p] = P[T.R];
        prl_sync();
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// CFR updates of one seat
// ---------------------------------------------------------------------------------------------------------------------
PRL_DEV PRL_INLINE void prl_regret_strategy_body(const PrlDevTree& T, const PrlDevState& S, const int32_t* __restrict__ nodes, int n_nodes_p, int p,
                                                 int variant, int iter) {
    const size_t total = (size_t)n_nodes_p * T.R;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int node = nodes[t / T.R];
        const int h = (int)(t % T.R);
        const int A = T.n_children[node];
        const int32_t* ch = T.child_list + T.child_start[node];
        const int col0 = T.first_col[node];
        const float strat_ev = S.ev[prl_vidx(T, node, p) + h];
        for (int i = 0; i < A; ++i) {
            float* rg = S.regret + prl_cidx(T, col0 + i) + h;
            const float d = S.ev[prl_vidx(T, ch[i], p) + h] - strat_ev;
            float r;
            if (iter == 0) r = d;                                                   // *_first_it
            else if (variant == PRL_CFR_LINEAR) r = ((float)(iter + 1) * d) + *rg;  // LinearCFR.py:27-28
            else r = d + *rg;                                                       // VanillaCFR.py:26-27, CFRPlus.py:37-38
            if (variant == PRL_CFR_PLUS) r = r > 0.f ? r : 0.f;
            *rg = r;
        }
        // regret matching (CFR+ regrets are already >= 0; Vanilla / Linear clamp first)
        auto capped = [&](int i) {
            float r = S.regret[prl_cidx(T, col0 + i) + h];
            return variant == PRL_CFR_PLUS ? r : (r > 0.f ? r : 0.f);
        };
        const float s = prl_np_sum(capped, A);
        const float unif = (float)(1.0 / (double)A);
        for (int i = 0; i < A; ++i) S.strategy[prl_cidx(T, col0 + i) + h] = s > 0.f ? (double)(capped(i) / s) : (double)unif;
        if (h == 0) S.strat_f64[node] = 0;
    }
}

// mode (CFR+ only): 0 nothing yet (iter < delay), 1 copy (iter == delay), 2 blend with the float64 weights m_old / m_new
PRL_DEV PRL_INLINE void prl_average_body(const PrlDevTree& T, const PrlDevState& S, const int32_t* __restrict__ nodes, int n_nodes_p, int p, int variant,
                                        int iter, int mode, double m_old, double m_new) {
    const size_t total = (size_t)n_nodes_p * T.R;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int node = nodes[t / T.R];
        const int h = (int)(t % T.R);
        const int A = T.n_children[node];
        const int col0 = T.first_col[node];
        if (variant == PRL_CFR_PLUS) {
            if (mode == 0) continue;
            for (int i = 0; i < A; ++i) {
                const size_t k = prl_cidx(T, col0 + i) + h;
                if (mode == 2) S.avg[k] = m_old * S.avg[k] + m_new * S.strategy[k];
                else S.avg[k] = S.strategy[k];
            }
            if (h == 0) S.avg_f64[node] = mode == 2 ? 1 : S.strat_f64[node];
            continue;
        }
        const float rp = S.reach[prl_vidx(T, node, p) + h];
        for (int i = 0; i < A; ++i) {
            const size_t k = prl_cidx(T, col0 + i) + h;
            float contrib = (float)S.strategy[k] * rp;
            if (variant == PRL_CFR_LINEAR) contrib = contrib * (float)(iter + 1);
            S.avg_sum[k] = iter > 0 ? S.avg_sum[k] + contrib : contrib;
        }
        const float s = prl_np_sum([&](int i) { return S.avg_sum[prl_cidx(T, col0 + i) + h]; }, A);
        for (int i = 0; i < A; ++i) {
            const size_t k = prl_cidx(T, col0 + i) + h;
            S.avg[k] = (s == 0.f) ? 1.0 / (double)A : (double)(S.avg_sum[k] / s);
        }
        if (h == 0) S.avg_f64[node] = 1;
    }
}

PRL_GLOBAL void prl_k_reach_root(PrlDevTree T, PrlDevState S) { prl_reach_root_body(T, S); }
PRL_GLOBAL void prl_k_reach_level(PrlDevTree T, PrlDevState S, int level_begin, int level_count) { prl_reach_level_body(T, S, level_begin, level_count); }
PRL_GLOBAL void prl_k_terminal_1card(PrlDevTree T, PrlDevState S, const int32_t* __restrict__ term_nodes, int n_term) {
    prl_terminal_1card_body(T, S, term_nodes, n_term);
}
PRL_GLOBAL void prl_k_ev_level(PrlDevTree T, PrlDevState S, int level_begin, int level_count) { prl_ev_level_body(T, S, level_begin, level_count); }
// one level of the run-out forest: its chance nodes sum their children (the seats the pass wants); a chain root's sum is also a row of its street
PRL_GLOBAL void prl_k_st_chain_sum(PrlDevTree T, float* __restrict__ ev, int level_begin, int level_count, PrlStChainDev cd, PrlStChainIo io, PrlStRowMap map, int seat_mask) {
    const size_t total = (size_t)level_count * T.R;
    for (size_t t = (size_t)prl_bid() * prl_nthreads() + prl_tid(); t < total; t += (size_t)prl_nblocks() * prl_nthreads()) {
        const int node = T.level_nodes[level_begin + (int)(t / T.R)];
        const int h = (int)(t % T.R);
        if (T.kind[node] != PRL_NODE_CHANCE) continue;
        const int kid = cd.node_kid[node];
        for (int p = 0; p < 2; ++p) {
            if (!((seat_mask >> p) & 1)) continue;
            const float v = prl_chance_sum(T, ev, node, p, h);
            ev[prl_vidx(T, node, p) + h] = v;
            if (kid >= 0) {
                float* row = io.val[cd.street[kid]] + (size_t)cd.val_slot[kid] * map.width * T.R;
                for (int k = 0; k < map.width; ++k)
                    if (map.seat[k] == p) row[(size_t)k * T.R + h] = v;
            }
        }
    }
}
PRL_GLOBAL void prl_k_exploitability(PrlDevTree T, PrlDevState S, float* out2) { prl_exploitability_body(T, S, out2); }
// the same, the result also to a second place (the solver's history slot: no separate 8-byte copy on a launch-bound trunk)
PRL_GLOBAL void prl_k_exploitability2(PrlDevTree T, PrlDevState S, float* out2, float* out2b) {
    prl_exploitability_body(T, S, out2);
    prl_sync();
    if (prl_tid() < 2) out2b[prl_tid()] = out2[prl_tid()];
}
PRL_GLOBAL void prl_k_regret_strategy(PrlDevTree T, PrlDevState S, const int32_t* __restrict__ nodes, int n_nodes_p, int p, int variant, int iter) {
    prl_regret_strategy_body(T, S, nodes, n_nodes_p, p, variant, iter);
}
PRL_GLOBAL void prl_k_average(PrlDevTree T, PrlDevState S, const int32_t* __restrict__ nodes, int n_nodes_p, int p, int variant, int iter, int mode,
                              double m_old, double m_new) {
    prl_average_body(T, S, nodes, n_nodes_p, p, variant, iter, mode, m_old, m_new);
}

// ---- graph-replayable flavours: the iteration counter and the CFR+ averaging weights live in device memory, so that one
// captured hipGraph of a whole iteration can be replayed for every iteration (launch-bound Leduc-sized trees) -------------
PRL_GLOBAL void prl_k_iter_begin(PrlIterDev* ip, int variant, int delay) {
    if (prl_bid() != 0 || prl_tid() != 0) return;
    const int it = ip->iter;
    int mode = 0;
    double m_old = 0., m_new = 0.;
    if (variant == PRL_CFR_PLUS) {  // CFRPlus.py:65-87: float64 weights from integer sums
        if (it > delay) {
            const long long cw = ((long long)it * (it + 1) - (long long)delay * (delay + 1)) / 2;  // sum of delay + 1 .. it
            const long long nw = it - delay + 1;
            m_old = (double)cw / (double)(cw + nw);
            m_new = (double)nw / (double)(cw + nw);
            mode = 2;
        } else if (it == delay) mode = 1;
    }
    ip->mode = mode; ip->m_old = m_old; ip->m_new = m_new;
}
PRL_GLOBAL void prl_k_iter_end(PrlIterDev* ip, const float* __restrict__ expl) {
    if (prl_bid() != 0 || prl_tid() != 0) return;
    const int it = ip->iter + 1;
    ip->iter = it;
    ip->hist[2 * (size_t)it] = expl[0];
    ip->hist[2 * (size_t)it + 1] = expl[1];
}
PRL_GLOBAL void prl_k_regret_strategy_dev(PrlDevTree T, PrlDevState S, const int32_t* __restrict__ nodes, int n_nodes_p, int p, int variant,
                                          const PrlIterDev* __restrict__ ip) {
    prl_regret_strategy_body(T, S, nodes, n_nodes_p, p, variant, ip->iter);
}
PRL_GLOBAL void prl_k_average_dev(PrlDevTree T, PrlDevState S, const int32_t* __restrict__ nodes, int n_nodes_p, int p, int variant,
                                  const PrlIterDev* __restrict__ ip) {
    prl_average_body(T, S, nodes, n_nodes_p, p, variant, ip->iter, ip->mode, ip->m_old, ip->m_new);
}

// ---- small trees (1-hole-card games, n_nodes * R of a few thousand: StandardLeduc, DiscretizedNLLeduc): the tree state is
// L2-resident and an iteration is ~35 dependent launches of ~5 us each. ONE workgroup runs `n_iters` whole iterations in one
// launch instead: the same per-(node, hand) bodies as the level kernels, workgroup barriers where the kernel boundaries were.
struct PrlSmallIterArgs {
    const int32_t* level_start;  // device copy of the BFS level offsets, [n_levels + 1]
    const int32_t* term_nodes; int32_t n_term;
    const int32_t* nodes_p[2]; int32_t n_nodes_p[2];
    int32_t variant, delay, n_iters;
    int32_t state_in_lds;        // the whole per-node / per-column state fits in LDS: iterate there, copy back at the end
    int32_t tree_in_lds;         // ... and the tree's own arrays beside it (or alone, when the state does not fit)
    int32_t n_cols;
    PrlIterDev* ip;
};

// the iteration counter and the CFR+ averaging weights of the iteration under way, in LDS: lane 0 writes them, every lane reads them in every
// phase -- from the PrlIterDev in HBM that was a round trip per use
#define PRL_SMALL_IP_BYTES 64
struct PrlSmallIp { int32_t iter, mode; double m_old, m_new; };
// carve the solver state out of LDS (16-byte aligned pieces) and copy it in (dir = 0) or back out (dir = 1)
PRL_DEV PRL_INLINE size_t prl_small_state_lds(const PrlDevTree& T, const PrlDevState& G, int n_cols, PrlDevState& L, int dir) {  // returns the bytes it took
    char* base = prl_smem();
    size_t off = PRL_SMALL_IP_BYTES;  // LDS block: the iteration's parameters, the state, then the tree's arrays (prl_small_tree_lds)
    auto piece = [&](auto*& lp, auto* gp, size_t count) {
        typedef typename std::remove_reference<decltype(*gp)>::type E;
        if (!gp) { lp = nullptr; return; }
        E* l = (E*)(base + off);
        off += (count * sizeof(E) + 15) & ~(size_t)15;
        lp = l;
        for (size_t i = prl_tid(); i < count; i += prl_nthreads()) {
            if (dir == 0) l[i] = gp[i];
            else gp[i] = l[i];
        }
    };
    const size_t nv = (size_t)T.n_nodes * 2 * T.R, nc = (size_t)n_cols * T.R;
    piece(L.strategy, G.strategy, nc);
    piece(L.strat_f64, G.strat_f64, (size_t)T.n_nodes);
    piece(L.reach, G.reach, nv);
    piece(L.ev, G.ev, nv);
    piece(L.ev_br, G.ev_br, nv);
    piece(L.br_idx, G.br_idx, (size_t)T.n_nodes * T.R);
    piece(L.regret, G.regret, nc);
    piece(L.avg_sum, G.avg_sum, nc);
    piece(L.avg, G.avg, nc);
    piece(L.avg_f64, G.avg_f64, (size_t)T.n_nodes);
    L.expl = G.expl;  // two floats, stay in HBM
    prl_sync();
    return off;
}
// The tree's own arrays in LDS (read only). Every phase of an iteration starts with a chain of dependent look-ups in them -- node of the level, its
// parent, its first column, its children -- and from HBM / L2 each link is most of a microsecond with one workgroup and nothing to overlap:
// ~35 phases x 2-3 links was most of StandardLeduc's 78 us per iteration.
// (with the iteration's node lists: the level offsets, the terminals, each player's decision nodes)
PRL_HD PRL_INLINE size_t prl_small_tree_bytes(const PrlDevTree& T, int n_term, int n_nodes_p0, int n_nodes_p1) {
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t n = (size_t)T.n_nodes;
    return 10 * al(n * 4) + al((n + 1) * 4) + al((n > 0 ? n - 1 : 0) * 4) + (T.chance_w ? al(n * 4) : 0) + al((size_t)T.R * 2 * 2) +
           al((size_t)T.n_boards * (size_t)T.board_len) + al(((size_t)T.n_levels + 1) * 4) + al((size_t)n_term * 4) + al((size_t)n_nodes_p0 * 4) +
           al((size_t)n_nodes_p1 * 4);
}
PRL_DEV PRL_INLINE void prl_small_tree_lds(const PrlDevTree& G, PrlDevTree& L, const PrlSmallIterArgs& AG, PrlSmallIterArgs& AL, size_t off) {
    char* base = prl_smem();
    auto piece = [&](auto*& lp, auto* gp, size_t count) {
        typedef typename std::remove_const<typename std::remove_reference<decltype(*gp)>::type>::type E;
        if (!gp) { lp = nullptr; return; }
        E* l = (E*)(base + off);
        off += (count * sizeof(E) + 15) & ~(size_t)15;
        for (size_t i = prl_tid(); i < count; i += prl_nthreads()) l[i] = gp[i];
        lp = l;
    };
    const size_t n = (size_t)G.n_nodes;
    piece(L.kind, G.kind, n); piece(L.actor, G.actor, n); piece(L.parent, G.parent, n); piece(L.child_idx, G.child_idx, n);
    piece(L.acted_last, G.acted_last, n); piece(L.board_id, G.board_id, n); piece(L.main_pot, G.main_pot, n); piece(L.n_children, G.n_children, n);
    piece(L.first_col, G.first_col, n); piece(L.level_nodes, G.level_nodes, n);
    piece(L.child_start, G.child_start, n + 1);
    piece(L.child_list, G.child_list, n > 0 ? n - 1 : 0);
    piece(L.chance_w, G.chance_w, n);
    piece(L.hole, G.hole, (size_t)G.R * 2);
    piece(L.boards, G.boards, (size_t)G.n_boards * (size_t)G.board_len);
    piece(AL.level_start, AG.level_start, (size_t)G.n_levels + 1);
    piece(AL.term_nodes, AG.term_nodes, (size_t)AG.n_term);
    piece(AL.nodes_p[0], AG.nodes_p[0], (size_t)AG.n_nodes_p[0]);
    piece(AL.nodes_p[1], AG.nodes_p[1], (size_t)AG.n_nodes_p[1]);
    prl_sync();
}
PRL_DEV PRL_INLINE void prl_small_ev(const PrlDevTree& T, const PrlDevState& S, const PrlSmallIterArgs& A) {
    prl_terminal_1card_body(T, S, A.term_nodes, A.n_term);
    prl_sync();
    for (int d = T.n_levels - 2; d >= 0; --d) {
        prl_ev_level_body(T, S, A.level_start[d], A.level_start[d + 1] - A.level_start[d]);
        prl_sync();
    }
    prl_exploitability_body(T, S, S.expl);
    prl_sync();
}
// IN_LDS is a template argument so that, in the instantiation that iterates on LDS, every state pointer is VISIBLY an LDS address (base of the
// dynamic LDS + offset on every path): behind a run-time choice between the HBM and the LDS copy the pointers are generic, every access a FLAT
// instruction that waits on both memory counters, and nothing overlaps (the kernel had 212 of them).
template <bool IN_LDS, bool TREE_LDS>
PRL_DEV PRL_INLINE void prl_small_iterations_body(const PrlDevTree& TG, const PrlDevState& SG, const PrlSmallIterArgs& AG) {
    PrlSmallIterArgs A = AG;
    PrlDevState S = SG;
    size_t state_bytes = PRL_SMALL_IP_BYTES;
    PrlSmallIp& IP = *(PrlSmallIp*)prl_smem();
    if (IN_LDS) state_bytes = prl_small_state_lds(TG, SG, A.n_cols, S, 0);
    PrlDevTree T = TG;
    if (TREE_LDS) prl_small_tree_lds(TG, T, AG, A, state_bytes);
    for (int k = 0; k < A.n_iters; ++k) {
        if (prl_tid() == 0) {  // prl_k_iter_begin
            const int it = k == 0 ? A.ip->iter : IP.iter;
            int mode = 0;
            double m_old = 0., m_new = 0.;
            if (A.variant == PRL_CFR_PLUS) {
                if (it > A.delay) {
                    const long long cw = ((long long)it * (it + 1) - (long long)A.delay * (A.delay + 1)) / 2;
                    const long long nw = it - A.delay + 1;
                    m_old = (double)cw / (double)(cw + nw);
                    m_new = (double)nw / (double)(cw + nw);
                    mode = 2;
                } else if (it == A.delay) mode = 1;
            }
            IP.iter = it; IP.mode = mode; IP.m_old = m_old; IP.m_new = m_new;
        }
        prl_sync();
        for (int p = 0; p < 2; ++p) {
            if (p == 1) prl_small_ev(T, S, A);
            prl_regret_strategy_body(T, S, A.nodes_p[p], A.n_nodes_p[p], p, A.variant, IP.iter);
            prl_sync();
            prl_reach_root_body(T, S);
            prl_sync();
            for (int d = 1; d < T.n_levels; ++d) {
                prl_reach_level_body(T, S, A.level_start[d], A.level_start[d + 1] - A.level_start[d]);
                prl_sync();
            }
            prl_average_body(T, S, A.nodes_p[p], A.n_nodes_p[p], p, A.variant, IP.iter, IP.mode, IP.m_old, IP.m_new);
            prl_sync();
        }
        prl_small_ev(T, S, A);
        if (prl_tid() == 0) {  // prl_k_iter_end
            const int it = IP.iter + 1;
            IP.iter = it;
            A.ip->iter = it;
            A.ip->hist[2 * (size_t)it] = S.expl[0];
            A.ip->hist[2 * (size_t)it + 1] = S.expl[1];
        }
        prl_sync();
    }
    if (IN_LDS) prl_small_state_lds(TG, SG, A.n_cols, S, 1);
}
PRL_DEV PRL_INLINE void prl_small_iterations_dispatch(const PrlDevTree& T, const PrlDevState& SG, const PrlSmallIterArgs& A) {
    if (A.state_in_lds) {
        if (A.tree_in_lds) prl_small_iterations_body<true, true>(T, SG, A);
        else prl_small_iterations_body<true, false>(T, SG, A);
    } else {
        if (A.tree_in_lds) prl_small_iterations_body<false, true>(T, SG, A);
        else prl_small_iterations_body<false, false>(T, SG, A);
    }
}
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(1024) prl_k_small_iterations(PrlDevTree T, PrlDevState SG, PrlSmallIterArgs A) { prl_small_iterations_dispatch(T, SG, A); }
// many independent small trees at once, one workgroup (= one CU) per solve: jobs[blockIdx.y]
// A pointer LOADED from memory is a generic one to the compiler (a kernel ARGUMENT is known to be global): every access through it is a FLAT
// instruction that waits on both memory counters. The job table's pointers all address HBM: rebuild each from its bits as a global pointer.
#if defined(PRL_EMU)
template <class E> inline E* prl_global_ptr(E* p) { return p; }
#else
template <class E> PRL_DEV PRL_INLINE E* prl_global_ptr(E* p) { return (E*)(__attribute__((address_space(1))) E*)(size_t)p; }
#endif
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(1024) prl_k_small_iterations_many(const PrlSmallJob* jobs) {
    const PrlSmallJob& J = jobs[prl_bid_y()];  // launched 1 x n_jobs: the per-item bodies below see a grid of ONE workgroup
    PrlDevTree T = J.T;
    PrlDevState SG = J.S;
#define PRL_G(x) x = prl_global_ptr(x)
    PRL_G(T.kind); PRL_G(T.actor); PRL_G(T.parent); PRL_G(T.child_idx); PRL_G(T.acted_last); PRL_G(T.board_id); PRL_G(T.main_pot); PRL_G(T.n_children);
    PRL_G(T.first_col); PRL_G(T.child_start); PRL_G(T.child_list); PRL_G(T.level_nodes); PRL_G(T.boards); PRL_G(T.hole); PRL_G(T.chance_w);
    PRL_G(T.plan_sh); PRL_G(T.plan_pos); PRL_G(T.plan_gs); PRL_G(T.plan_ge); PRL_G(T.plan_cl); PRL_G(T.plan_klh); PRL_G(T.plan_nlive); PRL_G(T.plan_ndealt);
    PRL_G(T.plan_hgs); PRL_G(T.plan_hge); PRL_G(T.plan_clx); PRL_G(T.plan_pp);
    PRL_G(SG.strategy); PRL_G(SG.strat_f64); PRL_G(SG.reach); PRL_G(SG.ev); PRL_G(SG.ev_br); PRL_G(SG.br_idx); PRL_G(SG.regret); PRL_G(SG.avg_sum); PRL_G(SG.avg);
    PRL_G(SG.avg_f64); PRL_G(SG.expl);
    PrlSmallIterArgs A;
    A.level_start = J.level_start; A.term_nodes = J.term_nodes; A.n_term = J.n_term;
    A.nodes_p[0] = J.nodes_p[0]; A.nodes_p[1] = J.nodes_p[1]; A.n_nodes_p[0] = J.n_nodes_p[0]; A.n_nodes_p[1] = J.n_nodes_p[1];
    A.variant = J.variant; A.delay = J.delay; A.n_iters = J.n_iters; A.state_in_lds = J.state_in_lds; A.tree_in_lds = J.tree_in_lds; A.n_cols = J.n_cols; A.ip = J.ip;
    PRL_G(A.level_start); PRL_G(A.term_nodes); PRL_G(A.nodes_p[0]); PRL_G(A.nodes_p[1]); PRL_G(A.ip);
#undef PRL_G
    prl_small_iterations_dispatch(T, SG, A);
}

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
static inline int prl_grid_for(size_t work_items, int block) {
    size_t g = (work_items + (size_t)block - 1) / (size_t)block;
    if (g < 1) g = 1;
    if (g > 8192) g = 8192;  // grid-stride beyond 32 workgroups per CU
    return (int)g;
}

void prl_launch_fill_uniform(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_col_node, void* stream) {
    PRL_LAUNCH(prl_k_fill_uniform, prl_grid_for((size_t)T.n_cols * T.R, 256), 256, 0, stream, T, S, d_col_node);
}

void prl_launch_reach(const PrlDevTree& T, const PrlDevState& S, const int32_t* h_level_start, void* stream, bool root_is_set) {
    if (!root_is_set) PRL_LAUNCH(prl_k_reach_root, 1, 256, 0, stream, T, S);  // the root's reach is a constant: written once per state
    for (int d = 1; d < T.n_levels; ++d) {
        int cnt = h_level_start[d + 1] - h_level_start[d];
        if (cnt > 0) PRL_LAUNCH(prl_k_reach_level, prl_grid_for((size_t)cnt * T.R, 256), 256, 0, stream, T, S, h_level_start[d], cnt);
    }
}

void prl_launch_terminals(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_term_nodes, int n_term, void* stream) {
    if (n_term > 0) {
        if (T.n_hole == 1) {
            PRL_LAUNCH(prl_k_terminal_1card, prl_grid_for((size_t)n_term * 2 * T.R, 256), 256, 0, stream, T, S, d_term_nodes, n_term);
        } else {
            size_t smem = ((size_t)PRL_T2_YPAD + (PRL_T2_YPAD + 8) + 64 + 64 + (size_t)T.n_cards * 65) * sizeof(float);
            PRL_LAUNCH(prl_k_terminal_2card, 2 * n_term < 65536 ? 2 * n_term : 65536, 256, smem, stream, T, S, d_term_nodes, n_term);
        }
    }
}

// values bottom-up once the terminals (and a fused engine's chance leaves) hold theirs, and the root exploitability
void prl_launch_ev_levels(const PrlDevTree& T, const PrlDevState& S, const int32_t* h_level_start, void* stream, float* d_expl_copy) {
    for (int d = T.n_levels - 2; d >= 0; --d) {
        int cnt = h_level_start[d + 1] - h_level_start[d];
        if (cnt > 0) PRL_LAUNCH(prl_k_ev_level, prl_grid_for((size_t)cnt * T.R, 256), 256, 0, stream, T, S, h_level_start[d], cnt);
    }
    // (the root's level inside the exploitability kernel was measured: 31 us for that launch against 6 + 10 for the two; profiles/r05_experiments.txt)
    size_t smem = T.n_hole == 1 ? 0 : ((size_t)PRL_T2_YPAD + (PRL_T2_YPAD + 8) + 64 + 64) * sizeof(float);
    if (d_expl_copy) PRL_LAUNCH(prl_k_exploitability2, 1, 256, smem, stream, T, S, S.expl, d_expl_copy);
    else PRL_LAUNCH(prl_k_exploitability, 1, 256, smem, stream, T, S, S.expl);
}

// the run-out forest of the per-street engine (prl_st.h): its showdowns, then its chance levels bottom-up
void prl_launch_st_chain_eval(const PrlDevTree& Tc, const PrlStChainTerm* d_terms, const PrlStChainBundle* d_bundles, int n_bundles, const PrlStChainIo& io,
                              const PrlStChainDev& cd, float* ev_c, const int32_t* h_level_start, int mode, void* stream) {
    if (n_bundles <= 0) return;
    PrlStRowMap map = {};
    const bool both = prl_fhp_runs_seat(mode, 0) && prl_fhp_runs_seat(mode, 1), with_br = prl_fhp_with_br(mode);
    const int seat = prl_fhp_runs_seat(mode, 0) ? 0 : 1;
    map.width = prl_fhp_out_width(mode);
    if (both) {  // (ev0, ev1[, br0, br1])
        map.seat[0] = 0; map.seat[1] = 1;
        if (with_br) { map.seat[2] = 0; map.seat[3] = 1; }
    } else if (mode == PRL_FHP_UPDATE1_EVAL1) {  // seat 1's value, its value under its new strategy (nothing of seat 1 is decided below an all-in call: the same), its best response
        map.seat[0] = 1; map.seat[1] = 1; map.seat[2] = 1;
    } else {
        map.seat[0] = seat;
        if (with_br) map.seat[1] = seat;
    }
    const int seat_mask = both ? 3 : (mode == PRL_FHP_UPDATE1_EVAL1 ? 2 : 1 << seat);
    const int n_items = (seat_mask == 3 ? 2 : 1) * n_bundles;
    const size_t smem = ((size_t)PRL_T2_YPAD + (PRL_T2_YPAD + 8) + 64 + 64 + (size_t)Tc.n_cards * 65) * sizeof(float);
    PrlCtTree ct = {};
    ct.plan_sh = Tc.plan_sh; ct.plan_pos = Tc.plan_pos; ct.plan_gs = Tc.plan_gs; ct.plan_ge = Tc.plan_ge; ct.plan_cl = Tc.plan_cl; ct.plan_klh = Tc.plan_klh;
    ct.hole = Tc.hole; ct.plan_nlive = Tc.plan_nlive; ct.plan_ndealt = Tc.plan_ndealt; ct.main_pot = Tc.main_pot;
    ct.plan_stride = Tc.plan_stride; ct.cl_stride = Tc.cl_stride; ct.R = Tc.R; ct.n_cards = Tc.n_cards; ct.eq_const = Tc.eq_const;
    PRL_LAUNCH(prl_k_st_chain_terminals, n_items < 65536 ? n_items : 65536, PRL_CT_NT, smem, stream, ct, d_terms, d_bundles, n_bundles, io, cd, ev_c, map, seat_mask);
    for (int d = Tc.n_levels - 2; d >= 0; --d) {
        const int cnt = h_level_start[d + 1] - h_level_start[d];
        if (cnt > 0) PRL_LAUNCH(prl_k_st_chain_sum, prl_grid_for((size_t)cnt * Tc.R, 256), 256, 0, stream, Tc, ev_c, h_level_start[d], cnt, cd, io, map, seat_mask);
    }
}

void prl_launch_ev(const PrlDevTree& T, const PrlDevState& S, const int32_t* h_level_start, const int32_t* d_term_nodes, int n_term,
                   void* stream, float* d_expl_copy) {
    prl_launch_terminals(T, S, d_term_nodes, n_term, stream);
    prl_launch_ev_levels(T, S, h_level_start, stream, d_expl_copy);
}

void prl_launch_regret_strategy(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_nodes, int n, int p, int variant, int iter,
                                void* stream) {
    if (n > 0) PRL_LAUNCH(prl_k_regret_strategy, prl_grid_for((size_t)n * T.R, 256), 256, 0, stream, T, S, d_nodes, n, p, variant, iter);
}

void prl_launch_average(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_nodes, int n, int p, int variant, int iter, int mode,
                        double m_old, double m_new, void* stream) {
    if (n > 0) PRL_LAUNCH(prl_k_average, prl_grid_for((size_t)n * T.R, 256), 256, 0, stream, T, S, d_nodes, n, p, variant, iter, mode, m_old, m_new);
}

// bytes of LDS the whole solver state of a small tree takes (StandardLeduc: ~140 KB); if it fits, the iterations run on it there
size_t prl_small_state_bytes(const PrlDevTree& T, const PrlDevState& S) {
    auto al = [](size_t b) { return (b + 15) & ~(size_t)15; };
    const size_t nv = (size_t)T.n_nodes * 2 * T.R, nc = (size_t)T.n_cols * T.R;
    return al(nc * 8) + al(T.n_nodes) + 3 * al(nv * 4) + (S.br_idx ? al((size_t)T.n_nodes * T.R * 4) : 0) + al(nc * 4) +
           (S.avg_sum ? al(nc * 4) : 0) + al(nc * 8) + al(T.n_nodes);
}

// what of a small tree goes to LDS: the whole solver state if it fits, and the tree's arrays if they fit beside it (or alone)
void prl_small_lds_plan(const PrlDevTree& T, const PrlDevState& S, int n_term, int n_nodes_p0, int n_nodes_p1, bool* state_in_lds, bool* tree_in_lds, size_t* bytes) {
    const size_t limit = 160 * 1024 - 256, st = prl_small_state_bytes(T, S), tr = prl_small_tree_bytes(T, n_term, n_nodes_p0, n_nodes_p1);
    *state_in_lds = PRL_SMALL_IP_BYTES + st <= limit;
    const size_t base = PRL_SMALL_IP_BYTES + (*state_in_lds ? st : 0);
    *tree_in_lds = base + tr <= limit;
    *bytes = base + (*tree_in_lds ? tr : 0);
}
void prl_launch_small_iterations_many(const PrlSmallJob* d_jobs, int n_jobs, size_t lds_bytes, void* stream) {
    if (n_jobs > 0) PRL_LAUNCH_Y(prl_k_small_iterations_many, n_jobs, 1024, lds_bytes, stream, d_jobs);
}

void prl_launch_small_iterations(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_level_start, const int32_t* d_term_nodes, int n_term,
                                 const int32_t* d_nodes_p0, int n0, const int32_t* d_nodes_p1, int n1, int variant, int delay, int n_iters,
                                 PrlIterDev* d_ip, void* stream) {
    size_t lds = 0;
    bool in_lds = false, tree_lds = false;
    prl_small_lds_plan(T, S, n_term, n0, n1, &in_lds, &tree_lds, &lds);
    PrlSmallIterArgs A;
    A.state_in_lds = in_lds ? 1 : 0;
    A.tree_in_lds = tree_lds ? 1 : 0;
    A.n_cols = T.n_cols;
    A.level_start = d_level_start; A.term_nodes = d_term_nodes; A.n_term = n_term;
    A.nodes_p[0] = d_nodes_p0; A.nodes_p[1] = d_nodes_p1; A.n_nodes_p[0] = n0; A.n_nodes_p[1] = n1;
    A.variant = variant; A.delay = delay; A.n_iters = n_iters; A.ip = d_ip;
    PRL_LAUNCH(prl_k_small_iterations, 1, 1024, lds, stream, T, S, A);
}

void prl_launch_iter_begin(PrlIterDev* d_ip, int variant, int delay, void* stream) { PRL_LAUNCH(prl_k_iter_begin, 1, 64, 0, stream, d_ip, variant, delay); }
void prl_launch_iter_end(PrlIterDev* d_ip, const float* d_expl, void* stream) { PRL_LAUNCH(prl_k_iter_end, 1, 64, 0, stream, d_ip, d_expl); }
void prl_launch_regret_strategy_dev(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_nodes, int n, int p, int variant, const PrlIterDev* d_ip,
                                    void* stream) {
    if (n > 0) PRL_LAUNCH(prl_k_regret_strategy_dev, prl_grid_for((size_t)n * T.R, 256), 256, 0, stream, T, S, d_nodes, n, p, variant, d_ip);
}
void prl_launch_average_dev(const PrlDevTree& T, const PrlDevState& S, const int32_t* d_nodes, int n, int p, int variant, const PrlIterDev* d_ip, void* stream) {
    if (n > 0) PRL_LAUNCH(prl_k_average_dev, prl_grid_for((size_t)n * T.R, 256), 256, 0, stream, T, S, d_nodes, n, p, variant, d_ip);
}
