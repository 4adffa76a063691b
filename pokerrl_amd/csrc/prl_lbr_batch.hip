// Device-resident batched LBR (BASELINE.json config 5: "2^20 batched PokerEnv rollouts + 7-card eval"): every episode of
// LocalLBRWorker._run_limit / _run_no_limit (LocalLBRWorker.py:61-308) runs start to finish inside one workgroup -- betting
// engine (prl_env.h), card dealing, the agent's range (PokerRange.py), the synthetic tabular agent, LBR's look-ahead with the
// check-down equity (prl_lbr.h) and the payout. State lives in LDS while the hand is played; HBM holds only the decks in and
// the winnings out. The arithmetic is the host worker's (= the reference's) operation for operation: float32, NumPy
// summation order; tests/test_lbr.py compares the two paths hand by hand.
//
// Synthetic agents (no neural network in the timed region):
//   kind 0  uniform over the legal actions
//   kind 1  seeded hash policy: weight ((mix32(key + h * 0x9E3779B1 + a * 0x85EBCA6B) >> 8) & 0xFFFF) + 1 per (hand, legal
//           action), key = hash chain over the public betting state and the board, normalised per hand in float32;
//           tests/lbr_fixture_agent.py is the same policy as a host EvalAgent (the reference plays against that one)
//   kind 2  TABULAR policy in HBM (PrlPolicyTable: prl_policy_table_create): float32 [row][action][hand] columns addressed by a 64-bit hash of the
//           public state (the same hash chain under two seeds) through an open-addressed key table; pokerrl_amd/rl/tabular_agent.py builds it from a
//           PublicTree (a solver's average strategy) and is the same policy as a host EvalAgent. A state the table does not hold (LBR left the agent's
//           tree) plays uniformly over the legal actions -- on both sides.
// The agent's action is drawn with a counter-based hash of (seed, episode, step): no RNG state.
#include <cstdlib>
#include <string.h>

#include <string>
#include <type_traits>
#include <vector>

#include "prl_device.h"
#include "prl_env.h"
#include "prl_host.h"
#include "prl_lbr.h"
#include "prl_policy.h"
#include "prl_rt.h"

extern "C" int32_t prl_device_available(void);

// 1024 lanes = 16 waves = 4 per SIMD (128 VGPRs each): measured against 576 and 768 lanes on one box, 1.24 / 1.39 / 1.45 M hands/s (profiles/r20_lbr_threads.txt)
#ifndef LBRB_THREADS
#define LBRB_THREADS 1024
#endif
#define LBRB_MAX_Q 12      // check/call + up to 11 raise sizes considered by LBR (OFF_TREE_11): twelve candidate ranges = 64 KB of the workgroup's ~138 KB of LDS
#define LBRB_MAX_LEGAL 16  // fold, check/call and up to 14 bet sizes of either player: per-lane arrays of this size stay small (private memory
                           // per lane bounds how many waves the runtime keeps in flight)
#define LBRB_MAX_BOARDS 52  // boards per equity kept in LDS: one card to come (52-card turn: 46; >= PRL_LBR_MAX_CARDS: the rows double as work arrays)
#define LBRB_MAX_BOARDS_2 1088  // two cards to come (hold'em flop: C(47, 2) = 1081 -- the agent's cards are unknown to LBR): the equities go through an HBM scratch row

// PRL_LBRB_TIMING builds (python -m pokerrl_amd.build --variant lbrtiming PRL_LBRB_TIMING): lane 0 of every workgroup accumulates
// the shader clock between the marks below into stats[4 + i]; prl_lbr_batch_run prints the breakdown to stderr.
#ifdef PRL_LBRB_TIMING
#define LBRB_TICK(i) do { if (tid == 0) { const long long t_ = clock64(); atomicAdd(P.stats + 4 + (i), (unsigned long long)(t_ - t_prev)); t_prev = t_; } } while (0)
#else
#define LBRB_TICK(i) do { } while (0)
#endif
#define LBRB_N_STATS 16

struct PrlLbrBatchParams {
    PrlGame g_lbr, g_agent;
    PrlRules rules;
    int32_t n_envs, agent_seat, check_to_round, agent_kind, n_deal, limit;
    uint32_t seed, episode_base;
    double reward_scalar, ev_normalizer;
    const int8_t* cards;          // [n_envs][n_deal]: seat 0's hole cards, seat 1's, then the board in deal order
    float* winnings;              // [n_envs]
    unsigned long long* stats;    // [4] env steps, LBR look-ahead decisions, (range, board) equities, agent actions
    float* eq_scratch;            // [grid][LBRB_MAX_Q][LBRB_MAX_BOARDS_2] when LBR may decide with two cards to come, else NULL
    PrlPolicyTable tab;           // agent kind 2
    // PRE builds: LBR decisions with more than two board cards to come (hold'em before the flop: lbr_check_to_round = None, the reference's default,
    // LBRArgs.py:18). Such a check-down equity is C(50, 5) = 2 118 760 boards per candidate range -- no hand can afford it, and no hand has to: the
    // candidate ranges are a function of the PUBLIC history and of LBR's hand only, so the equities are cached per (history key, LBR hand) in HBM. A
    // hand that misses files a request (its candidate ranges) and stops; the host computes the requested equities with the stand-alone kernel
    // (prl_lbr_checkdown_equity: what the host worker calls at the same decision), puts them into the cache and plays the stopped hands AGAIN from
    // the start -- decks and agent draws are counter-based, so a replay reaches the same decision with the same ranges, and hits.
    const int32_t* env_list;      // [n_envs] the hands this launch plays (nullptr: 0 .. n_envs - 1)
    int32_t* status;              // [all hands] 1 = stopped at a missing equity
    const unsigned long long* pf_keys;  // [pf_mask + 1] cache: key of (history, LBR hand), 0 = empty
    const float* pf_wp;                 // [pf_mask + 1][LBRB_MAX_Q]
    unsigned long long* req_keys;       // [req_mask + 1] keys requested in this launch (claimed by atomicCAS: one request per key)
    int32_t* n_req;                     // requests filed
    int32_t* req_meta;                  // [max_req][8]: LBR's hand index, n_q, n_dealt, the cards on the table (5)
    unsigned long long* req_key_of;     // [max_req]
    float* req_ranges;                  // [max_req][LBRB_MAX_Q][R]
    uint32_t pf_mask, req_mask;
    int32_t max_req, pf_min_to_deal;
};
#define LBRB_PF_SEED 0x9E3779u  // history keys of a PRE build without a table

PRL_HD PRL_INLINE unsigned long long lbrb_pf_key(const LbrbHistKey& hk, int lbr_idx) {
    unsigned long long k = (((unsigned long long)hk.hi << 32) | hk.lo) ^ ((unsigned long long)(lbr_idx + 1) * 0x9E3779B97F4A7C15ull);
    k ^= k >> 29; k *= 0xBF58476D1CE4E5B9ull; k ^= k >> 32;
    return k == 0ull ? 1ull : k;
}

// P(action `a` | hand h) of the synthetic agent; legal[0..n_legal) ascending. 0 for an illegal action.
PRL_HD PRL_INLINE float lbrb_agent_prob(int kind, uint32_t key, int h, const int32_t* legal, int n_legal, int a) {
    bool is_legal = false;
    for (int j = 0; j < n_legal; ++j) is_legal |= legal[j] == a;
    if (!is_legal) return 0.f;
    if (kind == 0) return (float)(1.0 / (double)n_legal);
    float sum = 0.f, wa = 0.f;
    for (int j = 0; j < n_legal; ++j) {
        const uint32_t x = lbrb_mix32(key + (uint32_t)h * 0x9E3779B1u + (uint32_t)legal[j] * 0x85EBCA6Bu);
        const float w = (float)(((x >> 8) & 0xFFFFu) + 1u);
        sum = sum + w;
        if (legal[j] == a) wa = w;
    }
    return wa / sum;
}

// The agent's action for hand h given the uniform draw u: the first legal action whose cumulative probability exceeds u,
// else the last one (tests/lbr_fixture_agent.py: get_action). Same float32 values as summing lbrb_agent_prob over the legal
// actions -- the weights' sum is formed once instead of once per action.
PRL_HD PRL_INLINE int lbrb_agent_draw(int kind, uint32_t key, int h, const int32_t* legal, int n_legal, float u) {
    float sum = 0.f;
    if (kind != 0)
        for (int j = 0; j < n_legal; ++j) {
            const uint32_t x = lbrb_mix32(key + (uint32_t)h * 0x9E3779B1u + (uint32_t)legal[j] * 0x85EBCA6Bu);
            sum = sum + (float)(((x >> 8) & 0xFFFFu) + 1u);
        }
    float c = 0.f;
    for (int j = 0; j < n_legal; ++j) {
        float p;
        if (kind == 0) p = (float)(1.0 / (double)n_legal);
        else {
            const uint32_t x = lbrb_mix32(key + (uint32_t)h * 0x9E3779B1u + (uint32_t)legal[j] * 0x85EBCA6Bu);
            p = (float)(((x >> 8) & 0xFFFFu) + 1u) / sum;
        }
        c = c + p;
        if (u < c) return legal[j];
    }
    return legal[n_legal - 1];
}

// P(a | hand h) of agent `kind` in the state (key, row, legal list); TABLE builds read kind 2 from the table, the others never touch it
template <bool TABLE>
PRL_HD PRL_INLINE float lbrb_prob(const PrlPolicyTable& T, int kind, uint32_t key, int row, int h, int hm, const int32_t* legal, int n_legal, int a) {
    if (TABLE && kind == 2) {  // hm: the hand's index in the table's labelling (= h unless the table is keyed under suit-canonical boards)
        if (row >= 0) return T.probs[((size_t)row * T.n_actions + a) * T.range_size + hm];
        kind = 0;
    }
    return lbrb_agent_prob(kind, key, h, legal, n_legal, a);
}

template <bool TABLE>
PRL_HD PRL_INLINE int lbrb_draw(const PrlPolicyTable& T, int kind, uint32_t key, int row, int h, int hm, const int32_t* legal, int n_legal, float u) {
    if (TABLE && kind == 2) {
        if (row >= 0) {
            float c = 0.f;
            for (int j = 0; j < n_legal; ++j) {
                c = c + T.probs[((size_t)row * T.n_actions + legal[j]) * T.range_size + hm];
                if (u < c) return legal[j];
            }
            return legal[n_legal - 1];
        }
        kind = 0;
    }
    return lbrb_agent_draw(kind, key, h, legal, n_legal, u);
}

PRL_HD PRL_INLINE int lbrb_hand_idx(const PrlRules& r, const int8_t* hc) {
    if (r.n_hole_cards == 1) return hc[0];
    const int a = hc[0] < hc[1] ? hc[0] : hc[1], b = hc[0] < hc[1] ? hc[1] : hc[0];
    return prl_range_idx_2(a, b, r.n_cards);
}

struct LbrbShared {
    int32_t n_big, n_eq;           // class sizes of the look-ahead's first board
    PrlEnvState st;
    PrlStepInfo info;
    int32_t legal[LBRB_MAX_LEGAL];
    int32_t n_legal, action, done, n_dealt, step_ctr, n_q, n_boards, lbr_idx;
    uint32_t key;
    int32_t row, raise_row[LBRB_MAX_Q];  // tabular agent: the table rows of the state / of the states after LBR's candidate raises
    LbrbHistKey hk;                      // ... and the history key of the hand so far
    int32_t pf_slot, pf_req;             // PRE builds: the cache slot of this decision's equities (-1: a miss), the request it filed (-1: another hand did)
    int8_t cboard[5];                    // ... a table keyed under suit-canonical boards: the canonical form of the board on the table (prl_policy.h)
    int32_t canon_k;                     //     and the number of the suit permutation that makes it
    float total;
    int32_t raise_action[LBRB_MAX_Q], pot_after[LBRB_MAX_Q];
    uint32_t raise_key[LBRB_MAX_Q];
    int32_t raise_n_legal[LBRB_MAX_Q];
    int32_t raise_legal[LBRB_MAX_Q][LBRB_MAX_LEGAL];  // the agent's legal actions after raise q (the hash policy needs the whole list)
    float fold_prob[LBRB_MAX_Q], notfold_total[LBRB_MAX_Q], wp[LBRB_MAX_Q], cp_sum[LBRB_MAX_Q], util[LBRB_MAX_Q];
    int8_t board[5];
    int8_t pc[PRL_LBR_MAX_CARDS];
    int32_t n_pc;
    PrlLbrGame lg;
};

PRL_DEV PRL_INLINE float lbrb_serial_sum(const float* a, int n) {
    int k = 0;
    auto nx = [&]() { return a[k++]; };
    return prl_np_sum_stream<4>(n, nx);
}

// NumPy's pairwise sum of a[0..n) by the whole workgroup, same association as prl_np_sum_stream: the halving recursion ends
// in blocks of 8..128 elements with 8 strided accumulators each -- 8 lanes per block run the accumulator chains, one lane
// per block folds them and adds the tail, lane 0 adds the blocks in the recursion's order. Result in S.total.
#define LBRB_MAX_LEAVES 32
struct LbrbLeaves { int n_leaves; int lo[LBRB_MAX_LEAVES], n[LBRB_MAX_LEAVES]; float part[LBRB_MAX_LEAVES][8], sum[LBRB_MAX_LEAVES]; };

PRL_DEV PRL_INLINE void lbrb_build_leaves(LbrbLeaves& Lf, int n) {  // one thread; the block boundaries depend on n only
    int stack_lo[16], stack_n[16], sp = 0, nl = 0;
    stack_lo[0] = 0; stack_n[0] = n;
    while (sp >= 0) {
        const int lo = stack_lo[sp], m = stack_n[sp];
        --sp;
        if (m <= 128) { Lf.lo[nl] = lo; Lf.n[nl] = m; ++nl; continue; }
        int n2 = m / 2;
        n2 -= n2 % 8;
        ++sp; stack_lo[sp] = lo + n2; stack_n[sp] = m - n2;  // right half is visited after the left one
        ++sp; stack_lo[sp] = lo; stack_n[sp] = n2;
    }
    Lf.n_leaves = nl;
}

// sum over the recursion tree of the per-block sums (blocks in order): same halving as above, post-order adds
PRL_DEV PRL_INLINE float lbrb_combine(const LbrbLeaves& Lf, int n) {
    struct Frame { int n; int stage; float left; };
    Frame fr[8];  // n <= 128 * 2^7
    int sp = 0, leaf = 0;
    fr[0].n = n; fr[0].stage = 0; fr[0].left = 0.f;
    float ret = 0.f;
    while (sp >= 0) {
        Frame& f = fr[sp];
        int n2 = f.n / 2;
        n2 -= n2 % 8;
        if (f.stage == 0) {
            if (f.n <= 128) { ret = Lf.sum[leaf++]; --sp; }
            else { f.stage = 1; fr[sp + 1].n = n2; fr[sp + 1].stage = 0; ++sp; }
        } else if (f.stage == 1) {
            f.left = ret; f.stage = 2;
            fr[sp + 1].n = f.n - n2; fr[sp + 1].stage = 0; ++sp;
        } else { ret = f.left + ret; --sp; }
    }
    return ret;
}

PRL_DEV PRL_INLINE void lbrb_wg_sum(const float* a, int n, LbrbLeaves& Lf, LbrbShared& S) {
    const int tid = (int)prl_tid();
    prl_sync();
    if (n < 8 * 8 || Lf.n_leaves * 8 > LBRB_THREADS) {  // tiny ranges (Leduc): one lane
        if (tid == 0) S.total = lbrb_serial_sum(a, n);
        prl_sync();
        return;
    }
    if (tid < Lf.n_leaves * 8) {
        const int leaf = tid >> 3, j = tid & 7, lo = Lf.lo[leaf], m = Lf.n[leaf];
        float acc = a[lo + j];  // every block of a range >= 64 has >= 8 elements
        for (int i = 8 + j; i < m - (m % 8); i += 8) acc = acc + a[lo + i];
        Lf.part[leaf][j] = acc;
    }
    prl_sync();
    if (tid < Lf.n_leaves) {
        const float* r = Lf.part[tid];
        const int lo = Lf.lo[tid], m = Lf.n[tid];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (int i = m - (m % 8); i < m; ++i) res = res + a[lo + i];
        Lf.sum[tid] = res;
    }
    prl_sync();
    if (tid == 0) S.total = lbrb_combine(Lf, n);
    prl_sync();
}

// PokerRange.normalize (PokerRange.py:45-50), workgroup-wide
PRL_DEV PRL_INLINE void lbrb_normalize(float* rg, int R, LbrbLeaves& Lf, LbrbShared& S) {
    lbrb_wg_sum(rg, R, Lf, S);
    const float t = S.total, unif = (float)(1.0 / (double)R);
    for (int h = (int)prl_tid(); h < R; h += LBRB_THREADS) rg[h] = t == 0.f ? unif : rg[h] / t;
    prl_sync();
}

// ---- many NumPy pairwise sums of ONE length at once, by the whole workgroup (round 4) ---------------------------------------------------------------
// A look-ahead needs n_q x n_boards check-down equities, each three sums over ~1300 elements (the range with the board's hands zeroed; the
// normalised range over the hands LBR beats; over the hands it ties with). One lane per equity -- the first design -- walks 3500 elements in a
// row on 368 of the 576 lanes (turn: 8 candidates x 46 boards) and on 8 of them on the river: half of a hand's clocks (profiles/r08_lbr_phases.txt).
// NumPy's association is a tree, though: blocks of <= 128 elements with eight strided accumulators each, blocks added pairwise up the halving
// recursion. So the unit of work is one BLOCK of one sum (<= 128 elements, eight accumulators side by side, its tail): all lanes take blocks of
// all sums, a second step adds a sum's blocks in the recursion's order. Same values, same order.
struct LbrbLeafMap { int n, n_leaves; int lo[LBRB_MAX_LEAVES], m[LBRB_MAX_LEAVES]; };
// the recursion as a template (fully inlined): a stack indexed at run time would live in private memory -- a vector-memory round trip per push and pop
// on the one lane the workgroup is waiting for
template <int DEPTH>
PRL_DEV PRL_INLINE void lbrb_leaf_rec(LbrbLeafMap& M, int lo, int m, int& nl) {
    if (DEPTH == 0 || m <= 128) { M.lo[nl] = lo; M.m[nl] = m; ++nl; return; }
    int n2 = m / 2;
    n2 -= n2 % 8;
    lbrb_leaf_rec<(DEPTH > 0 ? DEPTH - 1 : 0)>(M, lo, n2, nl);
    lbrb_leaf_rec<(DEPTH > 0 ? DEPTH - 1 : 0)>(M, lo + n2, m - n2, nl);
}
PRL_DEV PRL_INLINE void lbrb_build_leaf_map(LbrbLeafMap& M, int n) {  // one thread
    M.n = n; M.n_leaves = 0;
    if (n < 8) return;  // fewer than eight elements are added one after the other
    int nl = 0;
    lbrb_leaf_rec<5>(M, 0, n, nl);  // ranges have at most 1326 entries: 128 * 2^5 covers them (static_assert at the call sites' R bound)
    M.n_leaves = nl;
}
// the blocks' sums added in the recursion's order (left half + right half, post-order); sums[leaf] is block number `leaf`
template <int DEPTH>
PRL_DEV PRL_INLINE float lbrb_comb(int n, const float* sums, int& leaf) {
    if (DEPTH == 0 || n <= 128) return sums[leaf++];
    int n2 = n / 2;
    n2 -= n2 % 8;
    const float l = lbrb_comb<(DEPTH > 0 ? DEPTH - 1 : 0)>(n2, sums, leaf);
    const float r = lbrb_comb<(DEPTH > 0 ? DEPTH - 1 : 0)>(n - n2, sums, leaf);
    return l + r;
}
// Eight elements at a time, in three explicit stages with scheduling fences between them: an element is two or three DEPENDENT LDS gathers
// (index list -> hole cards and range entry) and the compiler neither pipelines the loop nor interleaves inlined element functions -- one element
// at a time costs the full latency every time (measured: 300 clocks per element with 2.25 waves per SIMD to hide it). The element type gives
//   int index(int i)                         which entry (a hand) element i is
//   void fetch(int h, unsigned& w, float& r)  the loads
//   float value(unsigned w, float r)          straight-line arithmetic (a branch per element would fence the fetches again)
// elements first .. first + 7; slots >= valid repeat element `first` (their values are ignored by the callers)
#if defined(PRL_EMU)
#define LBRB_STAGE_FENCE() do { } while (0)
#else
#define LBRB_STAGE_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif
template <class El>
PRL_DEV PRL_INLINE void lbrb_eight(const El& el, int first, int valid, float (&v)[8]) {
    int h[8];
    unsigned w[8];
    float r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) h[k] = el.index(k < valid ? first + k : first);
    LBRB_STAGE_FENCE();
#pragma unroll
    for (int k = 0; k < 8; ++k) el.fetch(h[k], w[k], r[k]);
    LBRB_STAGE_FENCE();
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = el.value(w[k], r[k]);
}
// a block of NumPy's pairwise sum: elements lo .. lo + m - 1, 8 <= m <= 128
template <class El>
PRL_DEV PRL_INLINE float lbrb_block_sum(const El& el, int lo, int m) {
    float r[8], t[8];
    lbrb_eight(el, lo, 8, r);
    const int m8 = m & ~7;
    for (int i = 8; i < m8; i += 8) {
        lbrb_eight(el, lo + i, 8, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = r[k] + t[k];
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    const int tail = m & 7;
    if (tail) {
        lbrb_eight(el, lo + m8, tail, t);
#pragma unroll
        for (int k = 0; k < 7; ++k) res = k < tail ? res + t[k] : res;
    }
    return res;
}
// fewer than eight elements: 0 + e(0) + e(1) + ...
template <class El>
PRL_DEV PRL_INLINE float lbrb_small_sum(const El& el, int n) {
    float res = 0.f, t[8];
    if (n > 0) {
        lbrb_eight(el, 0, n, t);
#pragma unroll
        for (int k = 0; k < 7; ++k) res = k < n ? res + t[k] : res;
    }
    return res;
}
// t / d by one multiply-high: exact while t * d < 2^32 (here t < 2^16, d < 2^11)
struct LbrbMagic {
    uint32_t m;  // ceil(2^32 / d); 0 stands for d = 1 (2^32 does not fit)
    PRL_DEV PRL_INLINE int div(int t) const { return m == 0u ? t : (int)(uint32_t)(((unsigned long long)(uint32_t)t * m) >> 32); }
};
PRL_DEV PRL_INLINE LbrbMagic lbrb_magic(int d) {
    LbrbMagic g;
    g.m = d <= 1 ? 0u : (uint32_t)((0x100000000ull + (unsigned long long)d - 1ull) / (unsigned long long)d);
    return g;
}
#define LBRB_PART_FLOATS 12288  // 48 KB of block sums: every (sum, block) of a turn look-ahead in one round
// run(s, f) builds the element type of sum s (possibly one of several variants) and returns f(el); out(s, total) receives the result. Every lane of the
// workgroup calls this.
template <class Run, class Out>
PRL_DEV PRL_INLINE void lbrb_multi_sum(const LbrbLeafMap& M, int n_sums, float* part, Run run, Out out) {
    const int tid = (int)prl_tid(), n = M.n;
    if (n < 8) {
        for (int s = tid; s < n_sums; s += LBRB_THREADS) out(s, run(s, [&](const auto& el) { return lbrb_small_sum(el, n); }));
        prl_sync();
        return;
    }
    const int L = M.n_leaves, chunk = LBRB_PART_FLOATS / L;
    for (int s0 = 0; s0 < n_sums; s0 += chunk) {
        const int ns = n_sums - s0 < chunk ? n_sums - s0 : chunk;
        const LbrbMagic by_ns = lbrb_magic(ns);
        // block `leaf` of sum s0 + sl, the SUM index fastest: the lanes of a wave then walk the same block of neighbouring sums -- the same hands at
        // the same moment, so their gathers hit the same few LDS words (broadcasts). With the block index fastest, neighbouring lanes started
        // ~80 elements apart (80 mod 64 banks = 16): 16 lanes per bank, 59 % of the LDS cycles were conflicts (profiles/r15_lbr_pmc_sq.txt).
        for (int t = tid; t < ns * L; t += LBRB_THREADS) {
            const int leaf = by_ns.div(t), sl = t - leaf * ns, lo = M.lo[leaf], m = M.m[leaf];
            part[sl * L + leaf] = run(s0 + sl, [&](const auto& el) { return lbrb_block_sum(el, lo, m); });
        }
        prl_sync();
        for (int sl = tid; sl < ns; sl += LBRB_THREADS) {
            int leaf = 0;
            out(s0 + sl, lbrb_comb<6>(n, part + (size_t)sl * L, leaf));
        }
        prl_sync();
    }
}

// PokerRange.normalize on the block machinery: the range's sixteen blocks on sixteen lanes (eight elements in flight each), the blocks added in
// registers. The first version (lbrb_wg_sum) walks the recursion with a frame array indexed at run time -- private memory, a vector-memory round
// trip per push / pop on one lane while the workgroup waits -- and it runs after every agent action and every deal.
struct LbrbPlainEl {
    const float* a;
    PRL_DEV PRL_INLINE int index(int i) const { return i; }
    PRL_DEV PRL_INLINE void fetch(int i, unsigned& w, float& r) const { w = 0u; r = a[i]; }
    PRL_DEV PRL_INLINE float value(unsigned, float r) const { return r; }
};
PRL_DEV PRL_INLINE void lbrb_normalize_blocks(float* rg, int R, const LbrbLeafMap& MR, float* part, LbrbShared& S) {
    prl_sync();
    auto run = [&](int, auto f) { const LbrbPlainEl el = {rg}; return f(el); };
    auto out = [&](int, float v) { S.total = v; };
    lbrb_multi_sum(MR, 1, part, run, out);
    const float t = S.total, unif = (float)(1.0 / (double)R);
    for (int h = (int)prl_tid(); h < R; h += LBRB_THREADS) rg[h] = t == 0.f ? unif : rg[h] / t;
    prl_sync();
}

// x / b, correctly rounded, for many x and one b: the reciprocal is refined once, a quotient is a multiply and two residual corrections -- the
// instruction sequence of the generic float32 division minus its range scaling and special-case fix-up, which are the identity when b, the
// quotient and the residuals stay in the normal range (prl_fhp_div.inc has the argument and scripts/ubench/div_check.hip the exhaustive-style
// check on the GPU): b in [2^-35, 2^30] and x either 0 or >= 2^-33 b. Anything else takes the generic division.
struct LbrbDiv {
    float b, y;
    bool box;
    PRL_DEV PRL_INLINE float fast(float x) const {  // only when box
        float q = x * y;
        float r = __builtin_fmaf(-b, q, x);
        q = __builtin_fmaf(r, y, q);
        r = __builtin_fmaf(-b, q, x);
        return __builtin_fmaf(r, y, q);
    }
};
// small_ok: the caller knows that every non-zero x it will divide is >= 2^-33 b
PRL_DEV PRL_INLINE LbrbDiv lbrb_div_by(float b, bool small_ok) {
    LbrbDiv d;
    d.b = b;
    d.box = small_ok && b >= 0x1p-35f && b <= 0x1p30f;
    const float bb = d.box ? b : 1.f;
#if defined(PRL_EMU)
    float y = 1.0f / bb;
#else
    float y = __builtin_amdgcn_rcpf(bb);
#endif
    const float e = __builtin_fmaf(-bb, y, 1.0f);
    d.y = __builtin_fmaf(e, y, y);
    return d;
}

// ONE workgroup per CU (its LDS holds twelve candidate ranges); 16 waves at up to 128 VGPRs each since the sums of a look-ahead are spread over
// all lanes (12 VGPRs spill in rarely taken paths). Before that, with one lane per equity, 9 waves at 168 VGPRs were 3.3 % ahead of two workgroups
// at 96 (profiles/r06_experiments.txt).
#if defined(PRL_EMU)
#define LBRB_LB
#else
#ifndef LBRB_WAVES_PER_SIMD
#define LBRB_WAVES_PER_SIMD 4
#endif
#define LBRB_LB __launch_bounds__(LBRB_THREADS, LBRB_WAVES_PER_SIMD)
#endif
PRL_HD PRL_INLINE bool nh_is_two(const PrlRules& r) { return r.n_hole_cards == 2; }
PRL_HD PRL_INLINE size_t lbrb_smem_bytes(int R) {  // the carve-outs of prl_k_lbr_batch, in its order
    const size_t sizes[] = {(size_t)R * 4, (size_t)LBRB_MAX_Q * R * 4, (size_t)LBRB_MAX_Q * LBRB_MAX_BOARDS * 4, (size_t)R * 2, (size_t)R * 2, sizeof(LbrbShared),
                            sizeof(LbrbLeaves), (size_t)LBRB_MAX_Q * PRL_LBR_MAX_CARDS * 4, (size_t)LBRB_MAX_Q * LBRB_MAX_BOARDS * 4, 3 * sizeof(LbrbLeafMap),
                            (size_t)LBRB_PART_FLOATS * 4, LBRB_MAX_Q * sizeof(unsigned), (size_t)R * 2, (size_t)R * 2};
    size_t off = 0;
    for (size_t b : sizes) off = ((off + 15) & ~(size_t)15) + b;
    return off + 16;
}
// TABLE: the agent may be a tabular one (kind 2); the synthetic agents' kernel carries no table code. PRE: look-aheads with many cards to come go
// through the equity cache (see PrlLbrBatchParams); the kernels of the timed configurations carry none of it.
template <bool TABLE, bool PRE>
PRL_GLOBAL void LBRB_LB prl_k_lbr_batch(PrlLbrBatchParams P) {
    constexpr bool HK = TABLE || PRE;  // the history key of the hand so far is kept
    char* lbrb_smem = prl_smem();
    const int R = P.rules.range_size, tid = (int)prl_tid();
    // Every array is the LDS base plus a BYTE OFFSET (no pointer -> integer -> pointer round trips): the compiler has to see that these are LDS
    // addresses. Through an integer cast they became generic pointers and every access to them a FLAT instruction -- slower than ds_read and,
    // because FLAT operations count on both memory counters, each followed by a full s_waitcnt vmcnt(0) lgkmcnt(0): no two of them ever overlapped
    // (rounds 1-3 had the hole-card table, the shared state and the card probabilities behind such pointers).
    size_t lds_off = 0;
    auto carve = [&](size_t bytes) { const size_t at = (lds_off + 15) & ~(size_t)15; lds_off = at + bytes; return lbrb_smem + at; };
    float* rg = (float*)carve((size_t)R * 4);                        // [R] the agent's range
    float* cand = (float*)carve((size_t)LBRB_MAX_Q * R * 4);         // [LBRB_MAX_Q][R] candidate ranges of a look-ahead
    float* eq_lds = (float*)carve((size_t)LBRB_MAX_Q * LBRB_MAX_BOARDS * 4);  // [LBRB_MAX_Q][LBRB_MAX_BOARDS]
    uint16_t* cls_list = (uint16_t*)carve((size_t)R * 2);            // [R] the hands LBR beats (ascending), then the hands it ties with
    uint8_t* cls = (uint8_t*)eq_lds;  // [R] class of every hand: lives in the equity rows, which are idle until the lists are built
    static_assert(LBRB_MAX_Q * LBRB_MAX_BOARDS * 4 >= 1326, "the class bytes fit the equity rows");
    uint16_t* hole_lut = (uint16_t*)carve((size_t)R * 2);            // [R] c1 | c2 << 8
    LbrbShared& S = *(LbrbShared*)carve(sizeof(LbrbShared));
    LbrbLeaves& Lf = *(LbrbLeaves*)carve(sizeof(LbrbLeaves));
    float* cpw = (float*)carve((size_t)LBRB_MAX_Q * PRL_LBR_MAX_CARDS * 4);  // [LBRB_MAX_Q][PRL_LBR_MAX_CARDS] card probabilities of the look-ahead
    float* eq_b = (float*)carve((size_t)LBRB_MAX_Q * LBRB_MAX_BOARDS * 4);   // [LBRB_MAX_Q][LBRB_MAX_BOARDS] the tie sums of the equities under way
    LbrbLeafMap* maps = (LbrbLeafMap*)carve(3 * sizeof(LbrbLeafMap));        // blocks of sums over R / n_big / n_eq elements
    LbrbLeafMap &MR = maps[0], &MB = maps[1], &ME = maps[2];
    float* part = (float*)carve((size_t)LBRB_PART_FLOATS * 4);       // [LBRB_PART_FLOATS] accumulator slots of lbrb_multi_sum
    unsigned* minpos = (unsigned*)carve(LBRB_MAX_Q * sizeof(unsigned));  // [LBRB_MAX_Q] bits of the smallest non-zero entry of every candidate range
    uint16_t* hl2 = (uint16_t*)carve((size_t)R * 2);                 // [R] c1 | c2 << 8 of the hands the cards on the table leave, 0xFFFF for the others
    uint16_t* hmap = (uint16_t*)carve((size_t)R * 2);                // [R] tabular agent: hand -> its index in the table's labelling (identity unless suit-canonical)
    const bool canon = TABLE && P.agent_kind == 2 && P.tab.suit_canon != 0 && nh_is_two(P.rules);
    const bool coop = nh_is_two(P.rules) && R >= 64;  // the cooperative sums (hold'em ranges); tiny ranges keep one lane per sum
    if (tid == 0) lbrb_build_leaves(Lf, R);
    if (tid == 64) lbrb_build_leaf_map(MR, R);
    const int nh = P.rules.n_hole_cards, lbr_seat = 1 - P.agent_seat, n_board_total = P.rules.n_board_cards;
    unsigned long long n_steps = 0, n_look = 0, n_eq = 0, n_agent = 0;
#ifdef PRL_LBRB_TIMING
    long long t_prev = clock64();
#endif
    for (int h = tid; h < R; h += LBRB_THREADS) {
        int c1 = h, c2 = h;
        if (nh == 2) prl_hole_cards_2(h, P.rules.n_cards, &c1, &c2);
        hole_lut[h] = (uint16_t)(c1 | (c2 << 8));
    }

    for (int e_at = (int)prl_bid(); e_at < P.n_envs; e_at += (int)prl_nblocks()) {
        const int e = PRE && P.env_list ? P.env_list[e_at] : e_at;
        const int8_t* cards = P.cards + (size_t)e * P.n_deal;
        const int8_t* lbr_hand = cards + lbr_seat * nh;
        const int8_t* agent_hand = cards + P.agent_seat * nh;
        const int8_t* deck_board = cards + 2 * nh;
        const uint32_t episode = P.episode_base + (uint32_t)e + 1u;
        prl_sync();
        if (tid == 0) {
            prl_env_reset(P.g_lbr, S.st);
            S.done = 0; S.n_dealt = 0; S.step_ctr = 0;
            S.lbr_idx = lbrb_hand_idx(P.rules, lbr_hand);
            if (HK) { S.hk = lbrb_hist_step(lbrb_hist_root(TABLE && P.agent_kind == 2 ? P.tab.key_seed : LBRB_PF_SEED), S.st, S.board, 0, n_board_total, P.rules.n_suits); S.canon_k = 0; }
        }
        if (TABLE) for (int h = tid; h < R; h += LBRB_THREADS) hmap[h] = (uint16_t)h;
        // agent_range.reset(); set_cards_to_zero_prob(lbr_hand) (:73-74, :190-191)
        const float unif = (float)(1.0 / (double)R);
        PrlLbrGame hg;  // only the card geometry is needed for hand_has
        hg.n_hole = nh; hg.n_cards = P.rules.n_cards; hg.n_suits = P.rules.n_suits; hg.rank_rule = P.rules.rank_rule; hg.R = R;
        hg.n_board_total = n_board_total;
        {
            unsigned long long m = 0ull;
            for (int i = 0; i < nh; ++i) m |= 1ull << lbr_hand[i];
            for (int h = tid; h < R; h += LBRB_THREADS) rg[h] = (prl_lbr_hand_mask(hg, h, hole_lut) & m) ? 0.f : unif;
        }
        if (coop) lbrb_normalize_blocks(rg, R, MR, part, S); else lbrb_normalize(rg, R, Lf, S);
        LBRB_TICK(0);  // reset + range init

        while (true) {
            prl_sync();
            if (S.done) break;
            const int cur = S.st.cur;
            if (cur == lbr_seat) {
                int action = PRL_CHECK_CALL;
                const bool look = !(P.check_to_round >= 0 && S.st.round < P.check_to_round);
                if (look) {
                    // ---------------- LBR's one-step look-ahead (:91-154, :205-270) ------------------------------------------
                    if (tid == 0) {
                        // the look-ahead's game description is put together in REGISTERS and stored once: built in place in LDS, every store to the card
                        // list below might alias it and its fields were re-read from LDS in every iteration of the 52-card loop
                        PrlLbrGame g = hg;
                        const int n_dealt = S.n_dealt;
                        g.n_dealt = n_dealt; g.n_to_deal = n_board_total - n_dealt;
                        unsigned long long used = 0ull;  // cards on the table and in LBR's hand
                        for (int i = 0; i < 5; ++i) {
                            const int8_t c = i < n_dealt ? S.board[i] : (int8_t)0;
                            g.board[i] = c;
                            if (i < n_dealt) used |= 1ull << c;
                        }
                        int8_t h0 = lbr_hand[0], h1 = nh == 2 ? lbr_hand[1] : (int8_t)0;
                        if (nh == 2 && h0 > h1) { const int8_t t = h0; h0 = h1; h1 = t; }
                        g.lbr_hand[0] = h0; g.lbr_hand[1] = h1;
                        used |= 1ull << h0;
                        if (nh == 2) used |= 1ull << h1;
                        int n_pc = 0;  // prl_lbr_possible_cards: the cards that can still come, ascending
                        for (int c = 0; c < g.n_cards; ++c)
                            if (!((used >> c) & 1ull)) S.pc[n_pc++] = (int8_t)c;
                        S.n_pc = n_pc;
                        S.n_boards = g.n_to_deal == 0 ? 1 : (g.n_to_deal == 1 ? n_pc : (g.n_to_deal == 2 ? n_pc * (n_pc - 1) / 2 : prl_lbr_n_boards(g)));
                        S.lg = g;
                        const PrlEnvState st0 = S.st;
                        S.n_legal = prl_legal_actions(P.g_lbr, st0, S.legal);
                    }
                    prl_sync();
                    {
                        // one lane per legal raise (the legal list is ascending: fold / check-call first, then the raise sizes, so
                        // raise number q is the position in the list minus the non-raises before it)
                        const int n_legal = S.n_legal;
                        int first_raise = 0;
                        while (first_raise < n_legal && (S.legal[first_raise] == PRL_FOLD || S.legal[first_raise] == PRL_CHECK_CALL)) ++first_raise;
                        int n_raises = n_legal - first_raise;
                        if (n_raises > LBRB_MAX_Q - 1) n_raises = LBRB_MAX_Q - 1;
                        if (tid < n_raises) {
                            const int q = 1 + tid, a = S.legal[first_raise + tid];
                            // simulate LBR's raise; what the agent would answer in that state (its own bet set decides legality)
                            PrlEnvState s2 = S.st;
                            PrlStepInfo inf;
                            prl_env_step(P.g_lbr, s2, a, &inf);
                            S.raise_action[q] = a;
                            S.pot_after[q] = s2.main_pot + s2.bet[0] + s2.bet[1];
                            // straight into LDS: a per-lane array filled at run-time positions lives in private memory (a vector-memory round trip per entry)
                            int32_t* lg_q = S.raise_legal[q];
                            int nk = 0;
                            auto put = [&](int act) { lg_q[nk++] = act; };
                            const int nl2 = prl_legal_actions_to(P.g_agent, s2, put);
                            S.raise_n_legal[q] = nl2;
                            S.raise_key[q] = lbrb_state_key(P.seed, s2, S.board, S.n_dealt, n_board_total, P.rules.n_suits);
                            if (TABLE) S.raise_row[q] = P.agent_kind == 2 ? lbrb_table_row(P.tab, lbrb_hist_step(S.hk, s2, canon ? S.cboard : S.board, S.n_dealt, n_board_total, P.rules.n_suits)) : -1;
                        }
                        if (tid == 0) {
                            if (P.limit) S.step_ctr += n_raises;  // the limit branch asks get_action(step_env=False): one draw per raise is consumed (:120)
                            S.n_q = 1 + n_raises;
                        }
                    }
                    prl_sync();
                    LBRB_TICK(1);  // look-ahead set-up (scalar)
                    const PrlLbrGame g = S.lg;
                    const int n_q = S.n_q, n_boards = S.n_boards;
                    const bool big = n_boards > LBRB_MAX_BOARDS;
                    float* eq = big ? P.eq_scratch + (size_t)prl_bid() * 2 * LBRB_MAX_Q * LBRB_MAX_BOARDS_2 : eq_lds;
                    float* eqb = big ? eq + (size_t)LBRB_MAX_Q * LBRB_MAX_BOARDS_2 : eq_b;
                    const int eq_stride = big ? LBRB_MAX_BOARDS_2 : LBRB_MAX_BOARDS;
                    const bool cached = PRE && g.n_to_deal >= P.pf_min_to_deal;  // this decision's equities come from the cache (or stop the hand)
                    if (!cached && (g.n_to_deal > 2 || n_boards > LBRB_MAX_BOARDS_2 || (big && !P.eq_scratch))) {  // run() rejects configurations that get here
                        if (tid == 0) S.done = 1;
                        continue;
                    }
                    // candidate 0: the range as it is; candidate q: after "agent does not fold to raise q" (:131-141, :241-251)
                    for (int h = tid; h < R; h += LBRB_THREADS) cand[h] = rg[h];
                    for (int q = 1; q < n_q; ++q) {
                        const int nl2 = S.raise_n_legal[q];
                        const int32_t* lg2 = S.raise_legal[q];  // read from LDS where it is (the same word for every lane: a broadcast)
                        for (int h = tid; h < R; h += LBRB_THREADS)
                            cand[(size_t)q * R + h] = lbrb_prob<TABLE>(P.tab, P.agent_kind, S.raise_key[q], TABLE ? S.raise_row[q] : -1, h, TABLE ? (int)hmap[h] : h, lg2, nl2, PRL_FOLD);  // p(fold | hand)
                    }
                    // first complete board for the classification (see the quirk in prl_lbr_kernels.hip)
                    if (!cached) {
                        int8_t fb0[5];
                        prl_lbr_board_at(g, S.pc, S.n_pc, 0, fb0);
                        if (tid == 0) { S.n_big = 0; S.n_eq = 0; }
                        if (tid < LBRB_MAX_Q) minpos[tid] = 0x7F800000u;
                        prl_sync();
                        int nb1 = 0, ne1 = 0;
                        const int32_t rl = prl_lbr_rank(g, S.lbr_idx, fb0);  // LBR's own rank: once per lane, not once per hand
                        for (int h = tid; h < R; h += LBRB_THREADS) {
                            const int32_t rh = prl_lbr_rank(g, h, fb0);
                            const uint8_t c = rh < rl ? 1 : (rh == rl ? 2 : 0);  // prl_lbr_classify_hand
                            cls[h] = c;
                            nb1 += c == 1; ne1 += c == 2;
                        }
                        if (nb1) prl_lds_add_i(&S.n_big, nb1);  // class sizes: the same for every (range, board) pair of this look-ahead
                        if (ne1) prl_lds_add_i(&S.n_eq, ne1);
                    }
                    prl_sync();
                    // the two classes as ascending index lists (stable compaction by one wave: ballot + popcount of the lanes below), so that
                    // the sums over a class read element i directly instead of scanning the class bytes (prl_lbr_board_equity_lists)
                    if (!cached && tid == 64) lbrb_build_leaf_map(MB, S.n_big);
                    if (!cached && tid == 128) lbrb_build_leaf_map(ME, S.n_eq);
                    if (!cached && tid < 64) {
                        int at_big = 0, at_eq = S.n_big;
                        for (int base = 0; base < R; base += 64) {
                            const int h = base + tid;
                            const int c = h < R ? (int)cls[h] : 0;
                            const unsigned long long mb = prl_ballot(c == 1), me = prl_ballot(c == 2);
                            const unsigned long long below = tid == 0 ? 0ull : (~0ull >> (64 - tid));
                            if (c == 1) cls_list[at_big + prl_popc64(mb & below)] = (uint16_t)h;
                            if (c == 2) cls_list[at_eq + prl_popc64(me & below)] = (uint16_t)h;
                            at_big += prl_popc64(mb);
                            at_eq += prl_popc64(me);
                        }
                    }
                    prl_sync();
                    LBRB_TICK(2);  // candidate fold probabilities + classification
                    // one lane per raise: fold probability and the not-fold mass, NumPy order
                    // two lanes per raise, in different waves so that the two sums run side by side
                    if (coop) {
                        // 2 (n_q - 1) sums over the range: sum 2 (q - 1) = the fold probability of raise q, the next one its not-fold mass
                        auto run = [&](int si, auto f) {
                            const float* pf = cand + (size_t)(1 + (si >> 1)) * R;
                            const bool nf = (si & 1) != 0;
                            const float* r0 = rg;
                            struct El {
                                const float *r0, *pf; bool nf;
                                PRL_DEV PRL_INLINE int index(int k) const { return k; }
                                PRL_DEV PRL_INLINE void fetch(int k, unsigned& w, float& r) const { const float p = pf[k]; __builtin_memcpy(&w, &p, 4); r = r0[k]; }
                                PRL_DEV PRL_INLINE float value(unsigned w, float r) const { float p; __builtin_memcpy(&p, &w, 4); return nf ? r * (1.f - p) : r * p; }
                            };
                            const El el = {r0, pf, nf};
                            return f(el);
                        };
                        auto out = [&](int si, float v) { if (si & 1) S.notfold_total[1 + (si >> 1)] = v; else S.fold_prob[1 + (si >> 1)] = v; };
                        lbrb_multi_sum(MR, 2 * (n_q - 1), part, run, out);
                    } else if (tid >= 1 && tid < n_q) {
                        const float* pf = cand + (size_t)tid * R;
                        int k = 0;
                        auto nx = [&]() { const float v = rg[k] * pf[k]; ++k; return v; };
                        S.fold_prob[tid] = prl_np_sum_stream<4>(R, nx);              // np.sum(range * a_probs[:, FOLD])
                    } else if (tid >= 64 + 1 && tid < 64 + n_q) {
                        const int q = tid - 64;
                        const float* pf = cand + (size_t)q * R;
                        int k2 = 0;
                        auto nx2 = [&]() { const float v = rg[k2] * (1.f - pf[k2]); ++k2; return v; };
                        S.notfold_total[q] = prl_np_sum_stream<4>(R, nx2);           // mul_and_norm(1 - p_fold): normalisation
                    }
                    prl_sync();
                    LBRB_TICK(3);  // fold / not-fold sums (one lane per raise)
                    for (int q = 1; q < n_q; ++q) {
                        const float t = S.notfold_total[q];
                        for (int h = tid; h < R; h += LBRB_THREADS) {
                            const float v = rg[h] * (1.f - cand[(size_t)q * R + h]);
                            cand[(size_t)q * R + h] = t == 0.f ? unif : v / t;
                        }
                    }
                    prl_sync();
                    LBRB_TICK(4);  // candidate ranges
                    if (cached) {
                        // the equities of (this public history, LBR's hand): from the cache -- or the hand files a request and stops
                        if (tid == 0) {
                            const unsigned long long k = lbrb_pf_key(S.hk, S.lbr_idx);
                            int slot = -1;
                            for (uint32_t i = (uint32_t)k & P.pf_mask;; i = (i + 1u) & P.pf_mask) {
                                const unsigned long long ki = P.pf_keys[i];
                                if (ki == k) { slot = (int)i; break; }
                                if (ki == 0ull) break;
                            }
                            S.pf_slot = slot;
                            S.pf_req = -1;
                            if (slot < 0) {
                                for (uint32_t i = (uint32_t)(k >> 20) & P.req_mask, tries = 0; tries <= P.req_mask; i = (i + 1u) & P.req_mask, ++tries) {
                                    const unsigned long long was = prl_atomic_cas_u64(P.req_keys + i, 0ull, k);
                                    if (was == 0ull) {  // this hand files the request
                                        const int r = prl_atomic_add_i(P.n_req, 1);
                                        if (r < P.max_req) S.pf_req = r;
                                        break;
                                    }
                                    if (was == k) break;  // another hand did
                                }
                                P.status[e] = 1;
                                S.done = 1;
                            }
                        }
                        prl_sync();
                        if (S.pf_slot < 0) {
                            const int r = S.pf_req;
                            if (r >= 0) {
                                float* dst = P.req_ranges + (size_t)r * LBRB_MAX_Q * R;
                                for (int i = tid; i < n_q * R; i += LBRB_THREADS) dst[i] = cand[i];
                                if (tid == 0) {
                                    int32_t* m = P.req_meta + (size_t)r * 8;
                                    m[0] = S.lbr_idx; m[1] = n_q; m[2] = g.n_dealt;
                                    for (int i = 0; i < 5; ++i) m[3 + i] = i < g.n_dealt ? (int)g.board[i] : -1;
                                    P.req_key_of[r] = lbrb_pf_key(S.hk, S.lbr_idx);
                                }
                            }
                            continue;  // the hand is over for this launch (S.done): the host fills the cache and plays it again
                        }
                        if (tid < n_q) S.wp[tid] = P.pf_wp[(size_t)S.pf_slot * LBRB_MAX_Q + tid];
                    } else {
                    if (coop) {
                        // pair p = (candidate q, board b). Three rounds of sums: the pair's normaliser (the range without the board's hands) into eq,
                        // its tie sum into eqb, its win sum -- and with it the equity -- into eq (prl_lbr_board_equity_lists, term for term).
                        // Preparation, once per look-ahead: (1) the hole cards of the hands the cards ON THE TABLE leave (the others: 0xFFFF), so that
                        // "the board blocks hand h" is two byte compares with the card(s) still to come instead of two 64-bit shifts of a board mask;
                        // (2) the smallest non-zero entry of every candidate range: a candidate's entries sum to 1 within rounding and every
                        // normaliser is a sum over a subset, so entries >= 1.001 * 2^-33 are inside the shared-reciprocal division's box for every board.
                        unsigned long long base = 0ull;
                        for (int i = 0; i < g.n_dealt; ++i) base |= 1ull << g.board[i];
                        for (int h = tid; h < R; h += LBRB_THREADS) {
                            const unsigned v = hole_lut[h];
                            hl2[h] = (((base >> (v & 0xFFu)) | (base >> (v >> 8))) & 1ull) != 0ull ? (uint16_t)0xFFFFu : (uint16_t)v;
                        }
                        for (int q = 0; q < n_q; ++q) {
                            unsigned mn = 0x7F800000u;
                            for (int h = tid; h < R; h += LBRB_THREADS) {
                                const float x = cand[(size_t)q * R + h];
                                unsigned u;
                                __builtin_memcpy(&u, &x, 4);
                                mn = (u - 1u) < (mn - 1u) ? u : mn;  // non-negative floats order like their bits; 0 - 1 wraps to the maximum
                            }
                            for (int d = 32; d > 0; d >>= 1) {
                                const unsigned o = (unsigned)prl_shfl_i((int)mn, (int)prl_lane() ^ d);
                                mn = o < mn ? o : mn;
                            }
                            if (prl_lane() == 0) prl_lds_min_u(&minpos[q], mn);
                        }
                        prl_sync();
                        const int n_pairs = n_q * n_boards, n_big = S.n_big;
                        const float unif_r = (float)(1.0 / (double)R);
                        const int8_t* pcs = S.pc;
                        const int n_pc = S.n_pc, n_to_deal = g.n_to_deal;
                        // the card(s) still to come on board b, 0xFE where there is none (never a hole card)
                        auto new_cards = [&](int b, unsigned& x, unsigned& y) {
                            x = 0xFEu; y = 0xFEu;
                            if (n_to_deal == 1) x = (unsigned)pcs[b];
                            else if (n_to_deal == 2) {
                                int i = 0, left = b;
                                while (left >= n_pc - 1 - i) { left -= n_pc - 1 - i; ++i; }
                                x = (unsigned)pcs[i]; y = (unsigned)pcs[i + 1 + left];
                            }
                        };
                        auto equities = [&](auto ntd_tag) {
                            constexpr int NTD = decltype(ntd_tag)::value;  // cards to come: the element functions compare with that many
                            // MODE 0: the range entry of a hand the board leaves, else 0 (the normaliser's terms); 1: that over the pair's normaliser by the
                            // shared-reciprocal division; 2: by the generic division; 3: the constant 1 / R (an all-zero range counts as the uniform one)
                            struct ElBase {
                                const uint16_t *hl, *list; const float* r0; unsigned x, y; LbrbDiv dv; float unif;
                                PRL_DEV PRL_INLINE void fetch(int h, unsigned& w, float& r) const { w = hl[h]; r = r0[h]; }
                                PRL_DEV PRL_INLINE float live(unsigned v, float r) const {
                                    const unsigned c1 = v & 0xFFu, c2 = v >> 8;
                                    bool blocked = v == 0xFFFFu;
                                    if (NTD >= 1) blocked = blocked | (c1 == x) | (c2 == x);
                                    if (NTD >= 2) blocked = blocked | (c1 == y) | (c2 == y);
                                    return blocked ? 0.f : r;
                                }
                            };
                            struct ElNorm : ElBase {
                                PRL_DEV PRL_INLINE int index(int h) const { return h; }
                                PRL_DEV PRL_INLINE float value(unsigned w, float r) const { return this->live(w, r); }
                            };
                            struct ElFast : ElBase {
                                PRL_DEV PRL_INLINE int index(int i) const { return (int)this->list[i]; }
                                PRL_DEV PRL_INLINE float value(unsigned w, float r) const { return this->dv.fast(this->live(w, r)); }
                            };
                            struct ElSlow : ElBase {
                                PRL_DEV PRL_INLINE int index(int i) const { return (int)this->list[i]; }
                                PRL_DEV PRL_INLINE float value(unsigned w, float r) const { return this->live(w, r) / this->dv.b; }
                            };
                            struct ElUnif : ElBase {
                                PRL_DEV PRL_INLINE int index(int i) const { return (int)this->list[i]; }
                                PRL_DEV PRL_INLINE float value(unsigned, float) const { return this->unif; }
                            };
                            const LbrbMagic by_boards = lbrb_magic(n_boards);
                            auto run_norm = [&](int p, auto f) {
                                const int q = by_boards.div(p), b = p - q * n_boards;
                                ElNorm el;
                                new_cards(b, el.x, el.y);
                                el.r0 = cand + (size_t)q * R; el.hl = hl2; el.list = cls_list; el.unif = unif_r; el.dv = lbrb_div_by(1.f, false);
                                return f(el);
                            };
                            auto out_norm = [&](int p, float v) { const int q = by_boards.div(p); eq[q * eq_stride + (p - q * n_boards)] = v; };
                            lbrb_multi_sum(MR, n_pairs, part, run_norm, out_norm);
                            // the division variant is picked per call, outside the element function (a branch inside it would fence the fetches)
                            auto run_cls = [&](int p, int off, auto f) {
                                const int q = by_boards.div(p), b = p - q * n_boards;
                                const float norm = eq[q * eq_stride + b];
                                ElBase e;
                                new_cards(b, e.x, e.y);
                                e.r0 = cand + (size_t)q * R; e.hl = hl2; e.list = cls_list + off; e.unif = unif_r;
                                e.dv = lbrb_div_by(norm, norm <= 1.00125f && minpos[q] >= 0x2F0028F6u /* 1.00125 * 2^-33: >= 2^-33 norm */);
                                if (norm == 0.f) { ElUnif el; (ElBase&)el = e; return f(el); }
                                if (e.dv.box) { ElFast el; (ElBase&)el = e; return f(el); }
                                ElSlow el; (ElBase&)el = e;
                                return f(el);
                            };
                            auto run_eq = [&](int p, auto f) { return run_cls(p, n_big, f); };
                            auto out_eq = [&](int p, float v) { const int q = by_boards.div(p); eqb[q * eq_stride + (p - q * n_boards)] = v; };
                            lbrb_multi_sum(ME, n_pairs, part, run_eq, out_eq);
                            auto run_big = [&](int p, auto f) { return run_cls(p, 0, f); };
                            auto out_big = [&](int p, float v) {
                                const int q = by_boards.div(p), at = q * eq_stride + (p - q * n_boards);
                                eq[at] = v + eqb[at] / 2.0f;
                            };
                            lbrb_multi_sum(MB, n_pairs, part, run_big, out_big);
                        };
                        if (n_to_deal == 0) equities(std::integral_constant<int, 0>());
                        else if (n_to_deal == 1) equities(std::integral_constant<int, 1>());
                        else equities(std::integral_constant<int, 2>());
                    } else {
                        for (int t = tid; t < n_q * n_boards; t += LBRB_THREADS) {
                            const int q = t / n_boards, b = t % n_boards;
                            int8_t fb[5];
                            prl_lbr_board_at(g, S.pc, S.n_pc, b, fb);
                            eq[q * eq_stride + b] = prl_lbr_board_equity_lists(g, fb, cls_list, S.n_big, S.n_eq, cand + (size_t)q * R, hole_lut);
                        }
                    }
                    prl_sync();
                    LBRB_TICK(5);  // (range, board) equities
                    // work arrays in LDS: the card probabilities next to the shared state; the second set (two cards to come)
                    // in the LDS equity rows, which are idle then because those equities go through the HBM scratch row
                    for (int t = tid; t < n_q * g.n_cards; t += LBRB_THREADS) {  // one lane per (candidate, card): 51-term sums side by side
                        const int q = t / g.n_cards, c = t % g.n_cards;
                        cpw[q * PRL_LBR_MAX_CARDS + c] = prl_lbr_card_not_held(g, cand + (size_t)q * R, c);
                    }
                    prl_sync();
                    if (coop && g.n_to_deal <= 1) {
                        // prl_lbr_reduce_range_cp (:449-468) with its chains opened up: the card probabilities' sum by one lane per candidate, the products
                        // equity x board probability by one lane per (candidate, board), the running sum over the boards from LDS by one lane per candidate
                        // (one lane doing all of it walked 52 + 52 + 46 dependent LDS round trips). Same operations, same order.
                        unsigned long long used = 0ull;
                        for (int i = 0; i < 5; ++i) if (i < g.n_dealt) used |= 1ull << g.board[i];
                        used |= 1ull << g.lbr_hand[0];
                        used |= 1ull << g.lbr_hand[1];
                        for (int t = tid; t < n_q * g.n_cards; t += LBRB_THREADS) {
                            const int q = t / g.n_cards, c = t - q * g.n_cards;
                            if ((used >> c) & 1ull) cpw[q * PRL_LBR_MAX_CARDS + c] = 0.f;
                        }
                        prl_sync();
                        if (tid < n_q) {
                            const float* row = cpw + tid * PRL_LBR_MAX_CARDS;
                            int k = 0;
                            auto nx = [&]() { return row[k++]; };
                            S.cp_sum[tid] = prl_np_sum_stream<0>(g.n_cards, nx);
                        }
                        prl_sync();
                        for (int t = tid; t < n_q * n_boards; t += LBRB_THREADS) {
                            const int q = t / n_boards, b = t - q * n_boards;
                            const float e_qb = eq[q * eq_stride + b];
                            float x = e_qb * 1.0f;
                            if (g.n_to_deal == 1) {
                                const float sum = S.cp_sum[q];
                                float cpv = cpw[q * PRL_LBR_MAX_CARDS + S.pc[b]];
                                if (sum > 0.f) cpv = cpv / sum;
                                x = e_qb * cpv;
                            }
                            eqb[q * eq_stride + b] = x;
                        }
                        prl_sync();
                        if (tid < n_q) {
                            const float* row = eqb + tid * eq_stride;
                            float win = row[0];  // 0.0 (Python float) + float32 -> float32
                            for (int b0 = 1; b0 < n_boards; b0 += 8) {
                                float v[8];
#pragma unroll
                                for (int k = 0; k < 8; ++k) v[k] = row[b0 + k < n_boards ? b0 + k : 0];
#pragma unroll
                                for (int k = 0; k < 8; ++k) win = b0 + k < n_boards ? win + v[k] : win;
                            }
                            S.wp[tid] = win * 1.f;  // the factorial of the cards to come: 1
                        }
                    } else if (tid < n_q) S.wp[tid] = prl_lbr_reduce_range_cp(g, eq + tid * eq_stride, cpw + tid * PRL_LBR_MAX_CARDS, eq_lds + tid * LBRB_MAX_BOARDS,
                                                                              S.pc, S.n_pc);
                    }  // !cached
                    prl_sync();
                    LBRB_TICK(6);  // board probabilities + reduction (one lane per candidate)
                    // one lane per candidate computes its utility; lane 0 takes the arg-max in action order. Candidate 0 is check / call (action 1), the
                    // raises follow in ascending action order, fold is 0 and actions that are not candidates are -1 (:209-212): walking the candidates in
                    // their order IS np.argmax's "first maximum" over the action-indexed utility vector.
                    if (tid < n_q) {
                        const int asked = S.st.bet[P.agent_seat] - S.st.bet[lbr_seat];
                        const int pot_before = S.st.main_pot + S.st.bet[0] + S.st.bet[1];
                        const float wp = S.wp[tid];
                        float u;
                        if (tid == 0) u = wp * (float)pot_before - (1.f - wp) * (float)asked;
                        else {
                            const float fp = S.fold_prob[tid];
                            const int pot_after = S.pot_after[tid], chips_in = pot_after - pot_before;
                            const float ev_nf = (wp * (float)pot_after) - ((1.f - wp) * (float)chips_in);
                            u = fp * (float)pot_before + (1.f - fp) * ev_nf;
                        }
                        S.util[tid] = u;
                    }
                    prl_sync();
                    if (tid == 0) {
                        float best = 0.f;  // utility[FOLD] = 0
                        int best_a = PRL_FOLD;
                        float uq[LBRB_MAX_Q];
                        int aq[LBRB_MAX_Q];
#pragma unroll
                        for (int q = 0; q < LBRB_MAX_Q; ++q) { uq[q] = S.util[q < n_q ? q : 0]; aq[q] = q == 0 ? (int)PRL_CHECK_CALL : S.raise_action[q < n_q ? q : 0]; }
#pragma unroll
                        for (int q = 0; q < LBRB_MAX_Q; ++q)
                            if (q < n_q && uq[q] > best) { best = uq[q]; best_a = aq[q]; }  // np.argmax: the first maximum
                        S.action = best_a;
                        n_look += 1;
                        if (!cached) n_eq += (unsigned long long)n_q * n_boards;  // (cached equities were computed by the request that filled the cache)
                    }
                    prl_sync();
                    action = S.action;
                    LBRB_TICK(7);  // utilities + arg-max
                }
                prl_sync();  // every lane has read the state it branched on (seat to act, round) before lane 0 steps the env
                if (tid == 0) {
                    PrlEnvState st = S.st;
                    PrlStepInfo inf;
                    if (!P.limit && action >= 2) {  // step by pot fraction (:287-289)
                        const int amt = prl_fraction_of_pot_raise(st, P.g_lbr.bet_fracs[action - 2], st.cur);
                        prl_env_step_processed(P.g_lbr, st, PRL_BET_RAISE, amt, &inf);
                    } else prl_env_step(P.g_lbr, st, action, &inf);
                    S.st = st;
                    S.info = inf;
                }
            } else {
                // ---------------- the agent acts: draw its action, Bayes-update its range (:156-160, :272-281) --------------
                if (tid == 0) {
                    const PrlEnvState st = S.st;  // one batch of LDS reads; the engine then works on registers (stores to S.legal may alias S.st otherwise)
                    S.n_legal = prl_legal_actions(P.g_agent, st, S.legal);
                    S.key = lbrb_state_key(P.seed, st, S.board, S.n_dealt, n_board_total, P.rules.n_suits);
                    if (TABLE) S.row = P.agent_kind == 2 ? lbrb_table_row(P.tab, S.hk) : -1;
                    const int hi = lbrb_hand_idx(P.rules, agent_hand);
                    const uint32_t x = lbrb_mix32(P.seed * 0x51ED27u + episode * 0x9E3779B1u + (uint32_t)S.step_ctr);
                    const float u = (float)(x >> 8) / 16777216.0f;
                    S.step_ctr += 1;
                    S.action = lbrb_draw<TABLE>(P.tab, P.agent_kind, S.key, TABLE ? S.row : -1, hi, TABLE ? (int)hmap[hi] : hi, S.legal, S.n_legal, u);
                    n_agent += 1;
                }
                prl_sync();
                const int a = S.action;
                for (int h = tid; h < R; h += LBRB_THREADS) rg[h] = rg[h] * lbrb_prob<TABLE>(P.tab, P.agent_kind, S.key, TABLE ? S.row : -1, h, TABLE ? (int)hmap[h] : h, S.legal, S.n_legal, a);
                if (coop) lbrb_normalize_blocks(rg, R, MR, part, S); else lbrb_normalize(rg, R, Lf, S);
                if (tid == 0) {
                    PrlEnvState st = S.st;
                    PrlStepInfo inf;
                    if (!P.limit && a >= 2) {
                        const int amt = prl_fraction_of_pot_raise(st, P.g_agent.bet_fracs[a - 2], st.cur);
                        prl_env_step_processed(P.g_lbr, st, PRL_BET_RAISE, amt, &inf);
                    } else prl_env_step(P.g_lbr, st, a, &inf);
                    S.st = st;
                    S.info = inf;
                }
            }
            prl_sync();
            LBRB_TICK(8);  // agent action + range update + env step
            // ---------------- after the step: cards, range, payout (PokerEnv._step + the facade's _after_step) ---------------
            n_steps += tid == 0;
            const PrlStepInfo info = S.info;
            if (info.is_terminal) {
                if (tid == 0) {
                    int n_dealt = S.n_dealt;
                    if (info.rundown) {
                        for (; n_dealt < n_board_total; ++n_dealt) S.board[n_dealt] = deck_board[n_dealt];
                    }
                    const int pot = S.st.main_pot;
                    double award_lbr = 0.0;
                    if (S.st.folded[0] || S.st.folded[1]) award_lbr = S.st.folded[lbr_seat] ? 0.0 : (double)pot;
                    else {
                        PrlLbrGame g = hg;
                        int8_t fb[5];
                        for (int i = 0; i < 5; ++i) fb[i] = i < n_board_total ? S.board[i] : (int8_t)0;
                        const int32_t rl = prl_lbr_rank(g, S.lbr_idx, fb), ra = prl_lbr_rank(g, lbrb_hand_idx(P.rules, agent_hand), fb);
                        award_lbr = rl > ra ? (double)pot : (rl < ra ? 0.0 : (double)pot / 2.0);
                    }
                    const double stack_after = (double)S.st.stack[lbr_seat] + award_lbr;
                    const double rew = (stack_after - (double)P.g_lbr.start_stack[lbr_seat]) / P.reward_scalar;  // PokerEnv.py:1069-1072
                    P.winnings[e] = (float)(rew * P.reward_scalar * P.ev_normalizer);                            // :161, :304
                    S.done = 1;
                }
            } else if (info.chance_acts) {
                if (tid == 0) {
                    const int n_new = P.rules.board_cards_in_round[S.st.round];
                    for (int i = 0; i < n_new; ++i) { S.board[S.n_dealt] = deck_board[S.n_dealt]; S.n_dealt += 1; }
                    S.n_legal = n_new;  // scratch: how many cards are new
                    if (canon) S.canon_k = prl_suit_canon(S.board, S.n_dealt, P.rules.n_suits, S.cboard);
                    if (HK) S.hk = lbrb_hist_step(S.hk, S.st, canon ? S.cboard : S.board, S.n_dealt, n_board_total, P.rules.n_suits);
                }
                prl_sync();
                if (canon) {  // the hands in the canonical labelling: the same suit permutation applied to both hole cards
                    const int ck = S.canon_k;
                    for (int h = tid; h < R; h += LBRB_THREADS) {
                        const unsigned v = hole_lut[h];
                        hmap[h] = (uint16_t)prl_suit_perm_hand((int)(v & 0xFFu), (int)(v >> 8), ck, P.rules.n_suits, P.rules.n_cards);
                    }
                }
                // agent_range.update_after_new_round (PokerRange.py:60-65): the new board cards leave the range
                const int n_new = S.n_legal, nd = S.n_dealt;
                unsigned long long m = 0ull;
                for (int i = nd - n_new; i < nd; ++i) m |= 1ull << S.board[i];
                for (int h = tid; h < R; h += LBRB_THREADS)
                    if (prl_lbr_hand_mask(hg, h, hole_lut) & m) rg[h] = 0.f;
                if (coop) lbrb_normalize_blocks(rg, R, MR, part, S); else lbrb_normalize(rg, R, Lf, S);
            } else if (HK && tid == 0) S.hk = lbrb_hist_step(S.hk, S.st, canon ? S.cboard : S.board, S.n_dealt, n_board_total, P.rules.n_suits);
            LBRB_TICK(9);  // after the step: dealing / range update / payout
        }
    }
    if (tid == 0) {
        prl_atomic_add_u64(P.stats + 0, n_steps);
        prl_atomic_add_u64(P.stats + 1, n_look);
        prl_atomic_add_u64(P.stats + 2, n_eq);
        prl_atomic_add_u64(P.stats + 3, n_agent);
    }
}

// ---- tabular policies in HBM (agent kind 2) ---------------------------------------------------------------------------------------------------------
// keys / rows: the open-addressed key table as the host built it (capacity a power of two, 0 = empty); probs: [n_rows][n_actions][range_size] float32.
// probs == nullptr: the probabilities are left zeroed for the caller to fill on the device (prl_policy_table_from_solver, prl_solver.hip)
PrlPolicyTable* prl_policy_table_alloc(const uint64_t* keys, const int32_t* rows, uint32_t capacity, const float* probs, int32_t n_rows, int32_t n_actions,
                                       int32_t range_size, uint32_t key_seed) {
    if (!keys || !rows || capacity < 2 || (capacity & (capacity - 1)) || n_rows <= 0 || (uint32_t)n_rows >= capacity || n_actions < 2 || range_size <= 0) {
        prl_set_error("prl_policy_table_create: bad argument (capacity: a power of two above the number of rows)"); return nullptr;
    }
    if (!prl_device_available()) { prl_set_error("no HIP device: policy tables live in HBM"); return nullptr; }
    PrlPolicyTable* T = new PrlPolicyTable();
    memset(T, 0, sizeof(*T));
    T->mask = capacity - 1; T->key_seed = key_seed; T->n_rows = n_rows; T->n_actions = n_actions; T->range_size = range_size;
    const size_t np = (size_t)n_rows * n_actions * range_size;
    if (hipMalloc((void**)&T->keys, (size_t)capacity * 8) != hipSuccess || hipMalloc((void**)&T->rows, (size_t)capacity * 4) != hipSuccess ||
        hipMalloc((void**)&T->probs, np * 4) != hipSuccess || hipMemcpy(T->keys, keys, (size_t)capacity * 8, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(T->rows, rows, (size_t)capacity * 4, hipMemcpyHostToDevice) != hipSuccess ||
        (probs ? hipMemcpy(T->probs, probs, np * 4, hipMemcpyHostToDevice) : hipMemset(T->probs, 0, np * 4)) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(T->keys); (void)hipFree(T->rows); (void)hipFree(T->probs);
        delete T;
        prl_set_error("policy table: HIP allocation / copy failed (" + std::to_string((np * 4 + (size_t)capacity * 12) >> 20) + " MB)"); return nullptr;
    }
    return T;
}

extern "C" PrlPolicyTable* prl_policy_table_create(const uint64_t* keys, const int32_t* rows, uint32_t capacity, const float* probs, int32_t n_rows,
                                                   int32_t n_actions, int32_t range_size, uint32_t key_seed) {
    if (!probs) { prl_set_error("prl_policy_table_create: bad argument (no probabilities)"); return nullptr; }
    return prl_policy_table_alloc(keys, rows, capacity, probs, n_rows, n_actions, range_size, key_seed);
}

extern "C" int32_t prl_policy_table_info(const PrlPolicyTable* T, int64_t* out6) {
    if (!T || !out6) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    out6[0] = T->n_rows; out6[1] = T->n_actions; out6[2] = T->range_size; out6[3] = (int64_t)T->mask + 1; out6[4] = T->suit_canon; out6[5] = T->key_seed;
    return PRL_OK;
}

extern "C" int32_t prl_policy_table_export_keys(const PrlPolicyTable* T, uint64_t* out_keys, int32_t* out_rows) {
    if (!T || !out_keys || !out_rows) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    const size_t cap = (size_t)T->mask + 1;
    if (hipMemcpy(out_keys, T->keys, cap * 8, hipMemcpyDeviceToHost) != hipSuccess || hipMemcpy(out_rows, T->rows, cap * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError(); prl_set_error("HIP error in prl_policy_table_export_keys"); return PRL_ERR_HIP;
    }
    return PRL_OK;
}

extern "C" int32_t prl_policy_table_get_rows(const PrlPolicyTable* T, int32_t row_begin, int32_t n_rows, float* out) {
    if (!T || !out || row_begin < 0 || n_rows < 0 || (int64_t)row_begin + n_rows > T->n_rows) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    const size_t per_row = (size_t)T->n_actions * T->range_size;
    if (n_rows && hipMemcpy(out, T->probs + (size_t)row_begin * per_row, (size_t)n_rows * per_row * 4, hipMemcpyDeviceToHost) != hipSuccess) {
        (void)hipGetLastError(); prl_set_error("HIP error in prl_policy_table_get_rows"); return PRL_ERR_HIP;
    }
    return PRL_OK;
}

extern "C" int32_t prl_suit_canon_boards(const int8_t* boards, int32_t n, int32_t k, int32_t n_suits, int8_t* out_boards, int32_t* out_perm) {
    if (!boards || !out_boards || !out_perm || n < 0 || k < 1 || k > 5 || n_suits < 1 || n_suits > 4) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    for (int i = 0; i < n; ++i) out_perm[i] = prl_suit_canon(boards + (size_t)i * k, k, n_suits, out_boards + (size_t)i * k);
    return PRL_OK;
}

extern "C" void prl_policy_table_destroy(PrlPolicyTable* T) {
    if (!T) return;
    (void)hipFree(T->keys); (void)hipFree(T->rows); (void)hipFree(T->probs);
    delete T;
}

// n look-ups on the device, one lane each: the row of the history key (lo[i], hi[i]) and, where it exists, P(action[i] | hand[i]) of that row --
// what the batched engines read, exposed so that a table can be verified where it lives (tests; -1 / 0 for a key the table does not hold)
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(256) prl_k_policy_table_probe(PrlPolicyTable T, int n, const uint32_t* lo, const uint32_t* hi, const int32_t* action, const int32_t* hand,
                                                               int32_t* out_row, float* out_prob) {
    const int i = (int)(prl_bid() * prl_nthreads() + prl_tid());
    if (i >= n) return;
    const int row = lbrb_table_row(T, LbrbHistKey{lo[i], hi[i]});
    out_row[i] = row;
    out_prob[i] = row >= 0 ? T.probs[((size_t)row * T.n_actions + action[i]) * T.range_size + hand[i]] : 0.f;
}

extern "C" int32_t prl_policy_table_probe(const PrlPolicyTable* T, int32_t n, const uint32_t* key_lo, const uint32_t* key_hi, const int32_t* action, const int32_t* hand,
                                          int32_t* out_row, float* out_prob) {
    if (!T || n <= 0 || !key_lo || !key_hi || !action || !hand || !out_row || !out_prob) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    for (int i = 0; i < n; ++i)
        if (action[i] < 0 || action[i] >= T->n_actions || hand[i] < 0 || hand[i] >= T->range_size) { prl_set_error("prl_policy_table_probe: action / hand out of range"); return PRL_ERR_ARG; }
    void* d = nullptr;
    const size_t n4 = (size_t)n * 4;
    if (hipMalloc(&d, 6 * n4) != hipSuccess) { (void)hipGetLastError(); prl_set_error("prl_policy_table_probe: hipMalloc failed"); return PRL_ERR_HIP; }
    char* b = (char*)d;
    int rc = PRL_OK;
    if (hipMemcpy(b, key_lo, n4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(b + n4, key_hi, n4, hipMemcpyHostToDevice) != hipSuccess ||
        hipMemcpy(b + 2 * n4, action, n4, hipMemcpyHostToDevice) != hipSuccess || hipMemcpy(b + 3 * n4, hand, n4, hipMemcpyHostToDevice) != hipSuccess) rc = PRL_ERR_HIP;
    if (rc == PRL_OK) {
        PRL_LAUNCH(prl_k_policy_table_probe, (n + 255) / 256, 256, 0, nullptr, *T, (int)n, (const uint32_t*)b, (const uint32_t*)(b + n4), (const int32_t*)(b + 2 * n4),
                   (const int32_t*)(b + 3 * n4), (int32_t*)(b + 4 * n4), (float*)(b + 5 * n4));
        if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out_row, b + 4 * n4, n4, hipMemcpyDeviceToHost) != hipSuccess ||
            hipMemcpy(out_prob, b + 5 * n4, n4, hipMemcpyDeviceToHost) != hipSuccess) rc = PRL_ERR_HIP;
    }
    (void)hipFree(d);
    if (rc != PRL_OK) prl_set_error("HIP error in prl_policy_table_probe");
    return rc;
}

static int lbrb_check_table(int kind, const PrlPolicyTable* T, const PrlGame* agent_game, const PrlRules* rules, const char* who) {
    if (kind != 2) return PRL_OK;
    if (!T) { prl_set_error(std::string(who) + ": agent kind 2 needs a policy table"); return PRL_ERR_ARG; }
    const int n_act = agent_game->game_type == PRL_GAME_LIMIT ? 3 : agent_game->n_bet_sizes + 2;
    if (T->range_size != rules->range_size || T->n_actions < n_act) { prl_set_error(std::string(who) + ": the policy table was built for another game (range size / number of actions)"); return PRL_ERR_ARG; }
    return PRL_OK;
}

static int32_t lbr_batch_run_impl(const PrlGame* lbr_game, const PrlGame* agent_game, const PrlRules* rules, int32_t n_envs, int32_t agent_seat,
                                  int32_t check_to_round, int32_t agent_kind, uint32_t agent_seed, uint32_t episode_base, double reward_scalar,
                                  double ev_normalizer, const int8_t* cards, float* out_winnings, uint64_t* out_stats4, float* out_device_ms, const PrlPolicyTable* table) {
    if (!lbr_game || !agent_game || !rules || !cards || !out_winnings || n_envs <= 0 || agent_seat < 0 || agent_seat > 1 || agent_kind < 0 || agent_kind > 2) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (const int trc = lbrb_check_table(agent_kind, table, agent_game, rules, "batched LBR")) return trc;
    if (!prl_device_available()) { prl_set_error("no HIP device: batched LBR has no CPU fallback"); return PRL_ERR_NO_DEVICE; }
    const int nh = rules->n_hole_cards, nb = rules->n_board_cards;
    if (nh < 1 || nh > 2 || rules->n_cards > PRL_LBR_MAX_CARDS || nb > 5 || (nh == 2 && (rules->n_cards != 52 || nb != 5))) {
        prl_set_error("batched LBR: 1-hole-card games or 52-card hold'em"); return PRL_ERR_UNSUPPORTED;
    }
    if (lbr_game->game_type == PRL_GAME_NOLIMIT || agent_game->game_type != lbr_game->game_type) { prl_set_error("batched LBR: fixed-limit or discretized games"); return PRL_ERR_UNSUPPORTED; }
    if (lbr_game->n_bet_sizes + 2 > LBRB_MAX_LEGAL || agent_game->n_bet_sizes + 2 > LBRB_MAX_LEGAL) { prl_set_error("batched LBR: at most 14 bet sizes per player"); return PRL_ERR_UNSUPPORTED; }
    if (lbr_game->game_type == PRL_GAME_DISCRETIZED && lbr_game->n_bet_sizes + 1 > LBRB_MAX_Q) { prl_set_error("batched LBR: at most 11 LBR bet sizes"); return PRL_ERR_UNSUPPORTED; }
    // Look-aheads with at most two board cards to come are evaluated inside the kernel; with more (hold'em before the flop: lbr_check_to_round = None)
    // the equities come from the per-(history, LBR hand) cache and the request / replay rounds below (PrlLbrBatchParams). PRL_LBRB_PF_MIN (tests)
    // lowers the threshold so that small games walk the same machinery.
    int to_deal_max = 0, pf_min = 3;
    {
        int dealt_before_first_decision = 0;
        const int first_round = check_to_round >= 0 ? check_to_round : 0;
        for (int r = 0; r <= first_round && r < 4; ++r) dealt_before_first_decision += rules->board_cards_in_round[r];
        to_deal_max = nb - dealt_before_first_decision;
        if (const char* ev = getenv("PRL_LBRB_PF_MIN")) { pf_min = atoi(ev); if (pf_min < 0) pf_min = 0; }
    }
    const bool pre = to_deal_max >= pf_min;
    PrlLbrBatchParams P;
    memset(&P, 0, sizeof(P));
    P.g_lbr = *lbr_game; P.g_agent = *agent_game; P.rules = *rules;
    P.n_envs = n_envs; P.agent_seat = agent_seat; P.check_to_round = check_to_round; P.agent_kind = agent_kind;
    P.n_deal = 2 * nh + nb; P.limit = lbr_game->game_type == PRL_GAME_LIMIT;
    P.seed = agent_seed; P.episode_base = episode_base; P.reward_scalar = reward_scalar; P.ev_normalizer = ev_normalizer;
    P.pf_min_to_deal = pf_min;
    if (agent_kind == 2) P.tab = *table;
    const int R = rules->range_size;
    const size_t smem = lbrb_smem_bytes(R);
    int8_t* d_cards = nullptr; float* d_win = nullptr; unsigned long long* d_stats = nullptr; float* d_eq = nullptr;
    // PRE: the cache, the requests of a round, the hands of a round
    const int max_req = 1024;
    const uint32_t req_cap = 4096;
    int32_t *d_list = nullptr, *d_status = nullptr, *d_n_req = nullptr, *d_req_meta = nullptr;
    unsigned long long *d_pf_keys = nullptr, *d_req_keys = nullptr, *d_req_key_of = nullptr;
    float *d_pf_wp = nullptr, *d_req_ranges = nullptr;
    std::vector<unsigned long long> c_keys;
    std::vector<float> c_wp;
    uint32_t c_cap = 0, c_used = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float total_ms = 0.f;
    int n_rounds = 0;
    long long n_requests = 0;
    int rc = PRL_OK;
#define LB_TRY(x) do { if ((x) != hipSuccess) { prl_set_error("HIP error in prl_lbr_batch_run"); rc = PRL_ERR_HIP; goto done; } } while (0)
    LB_TRY(hipMalloc((void**)&d_cards, (size_t)n_envs * P.n_deal));
    LB_TRY(hipMalloc((void**)&d_win, (size_t)n_envs * sizeof(float)));
    LB_TRY(hipMalloc((void**)&d_stats, LBRB_N_STATS * sizeof(unsigned long long)));
    LB_TRY(hipMemcpy(d_cards, cards, (size_t)n_envs * P.n_deal, hipMemcpyHostToDevice));
    LB_TRY(hipMemset(d_stats, 0, LBRB_N_STATS * sizeof(unsigned long long)));
    LB_TRY(hipMemset(d_win, 0, (size_t)n_envs * sizeof(float)));
    P.cards = d_cards; P.winnings = d_win; P.stats = d_stats;
    LB_TRY(hipEventCreate(&e0));
    LB_TRY(hipEventCreate(&e1));
    {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        const int grid_max = cus * 8;  // persistent workgroups; every one plays its hands start to finish
        if (to_deal_max >= 2 && pf_min > 2) {
            const int grid = n_envs < grid_max ? n_envs : grid_max;
            LB_TRY(hipMalloc((void**)&d_eq, (size_t)grid * 2 * LBRB_MAX_Q * LBRB_MAX_BOARDS_2 * sizeof(float)));  // equities + the tie sums under way
            P.eq_scratch = d_eq;
        }
#if !defined(PRL_EMU)
        if (getenv("PRL_LBRB_DEBUG")) {
            int nb = -1;
            hipError_t oe = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, prl_k_lbr_batch<false, false>, LBRB_THREADS, smem);
            hipFuncAttributes fa;
            memset(&fa, 0, sizeof(fa));
            hipError_t fe = hipFuncGetAttributes(&fa, (const void*)prl_k_lbr_batch<false, false>);
            fprintf(stderr, "lbrb: smem %zu, occupancy %d workgroups per CU (err %d); regs %d, static smem %zu, local %zu, maxDyn %d (err %d)\n", smem, nb, (int)oe,
                    fa.numRegs, fa.sharedSizeBytes, fa.localSizeBytes, fa.maxDynamicSharedSizeBytes, (int)fe);
        }
#endif
        if (!pre) {
            const int grid = n_envs < grid_max ? n_envs : grid_max;
            LB_TRY(hipEventRecord(e0, nullptr));
            if (agent_kind == 2) PRL_LAUNCH((prl_k_lbr_batch<true, false>), grid, LBRB_THREADS, smem, nullptr, P);
            else PRL_LAUNCH((prl_k_lbr_batch<false, false>), grid, LBRB_THREADS, smem, nullptr, P);
            LB_TRY(hipEventRecord(e1, nullptr));
            LB_TRY(hipDeviceSynchronize());
            LB_TRY(hipEventElapsedTime(&total_ms, e0, e1));
        } else {
            // rounds: play the hands that are left; compute what they asked for; play the stopped ones again
            std::vector<int32_t> list((size_t)n_envs), status((size_t)n_envs), meta((size_t)max_req * 8);
            std::vector<unsigned long long> key_of((size_t)max_req);
            std::vector<float> ranges;
            for (int i = 0; i < n_envs; ++i) list[i] = i;
            LB_TRY(hipMalloc((void**)&d_list, (size_t)n_envs * 4));
            LB_TRY(hipMalloc((void**)&d_status, (size_t)n_envs * 4));
            LB_TRY(hipMalloc((void**)&d_n_req, 4));
            LB_TRY(hipMalloc((void**)&d_req_meta, (size_t)max_req * 8 * 4));
            LB_TRY(hipMalloc((void**)&d_req_keys, (size_t)req_cap * 8));
            LB_TRY(hipMalloc((void**)&d_req_key_of, (size_t)max_req * 8));
            LB_TRY(hipMalloc((void**)&d_req_ranges, (size_t)max_req * LBRB_MAX_Q * R * sizeof(float)));
            c_cap = 1u << 12;
            c_keys.assign(c_cap, 0ull);
            c_wp.assign((size_t)c_cap * LBRB_MAX_Q, 0.f);
            auto insert = [&](unsigned long long k, const float* wp) {
                for (uint32_t i = (uint32_t)k & (c_cap - 1);; i = (i + 1u) & (c_cap - 1)) {
                    if (c_keys[i] == k) return;
                    if (c_keys[i] == 0ull) { c_keys[i] = k; memcpy(&c_wp[(size_t)i * LBRB_MAX_Q], wp, LBRB_MAX_Q * sizeof(float)); ++c_used; return; }
                }
            };
            uint32_t d_cap = 0;
            while (!list.empty()) {
                if (2 * c_used + 2 * (uint32_t)max_req > c_cap) {  // keep the table at most half full after this round's insertions
                    std::vector<unsigned long long> ok;
                    std::vector<float> ow;
                    ok.swap(c_keys); ow.swap(c_wp);
                    const uint32_t old = c_cap;
                    while (2 * c_used + 2 * (uint32_t)max_req > c_cap) c_cap *= 2;
                    c_keys.assign(c_cap, 0ull); c_wp.assign((size_t)c_cap * LBRB_MAX_Q, 0.f); c_used = 0;
                    for (uint32_t i = 0; i < old; ++i) if (ok[i]) insert(ok[i], &ow[(size_t)i * LBRB_MAX_Q]);
                }
                if (d_cap != c_cap) {
                    (void)hipFree(d_pf_keys); (void)hipFree(d_pf_wp); d_pf_keys = nullptr; d_pf_wp = nullptr;
                    LB_TRY(hipMalloc((void**)&d_pf_keys, (size_t)c_cap * 8));
                    LB_TRY(hipMalloc((void**)&d_pf_wp, (size_t)c_cap * LBRB_MAX_Q * sizeof(float)));
                    d_cap = c_cap;
                }
                LB_TRY(hipMemcpy(d_pf_keys, c_keys.data(), (size_t)c_cap * 8, hipMemcpyHostToDevice));
                LB_TRY(hipMemcpy(d_pf_wp, c_wp.data(), (size_t)c_cap * LBRB_MAX_Q * sizeof(float), hipMemcpyHostToDevice));
                LB_TRY(hipMemcpy(d_list, list.data(), list.size() * 4, hipMemcpyHostToDevice));
                LB_TRY(hipMemset(d_status, 0, (size_t)n_envs * 4));
                LB_TRY(hipMemset(d_n_req, 0, 4));
                LB_TRY(hipMemset(d_req_keys, 0, (size_t)req_cap * 8));
                P.n_envs = (int)list.size(); P.env_list = d_list; P.status = d_status;
                P.pf_keys = d_pf_keys; P.pf_wp = d_pf_wp; P.pf_mask = c_cap - 1;
                P.req_keys = d_req_keys; P.req_mask = req_cap - 1; P.n_req = d_n_req; P.req_meta = d_req_meta; P.req_key_of = d_req_key_of; P.req_ranges = d_req_ranges;
                P.max_req = max_req;
                const int grid = P.n_envs < grid_max ? P.n_envs : grid_max;
                LB_TRY(hipEventRecord(e0, nullptr));
                if (agent_kind == 2) PRL_LAUNCH((prl_k_lbr_batch<true, true>), grid, LBRB_THREADS, smem, nullptr, P);
                else PRL_LAUNCH((prl_k_lbr_batch<false, true>), grid, LBRB_THREADS, smem, nullptr, P);
                LB_TRY(hipEventRecord(e1, nullptr));
                LB_TRY(hipDeviceSynchronize());
                float ms = 0.f;
                LB_TRY(hipEventElapsedTime(&ms, e0, e1));
                total_ms += ms;
                ++n_rounds;
                int n_req = 0;
                LB_TRY(hipMemcpy(&n_req, d_n_req, 4, hipMemcpyDeviceToHost));
                if (n_req > max_req) n_req = max_req;
                LB_TRY(hipMemcpy(status.data(), d_status, (size_t)n_envs * 4, hipMemcpyDeviceToHost));
                std::vector<int32_t> next;
                for (int32_t e : list) if (status[e]) next.push_back(e);
                if (!next.empty() && n_req == 0) { prl_set_error("batched LBR: hands wait for an equity nobody requested"); rc = PRL_ERR_STATE; goto done; }
                if (n_req > 0) {
                    LB_TRY(hipMemcpy(meta.data(), d_req_meta, (size_t)n_req * 8 * 4, hipMemcpyDeviceToHost));
                    LB_TRY(hipMemcpy(key_of.data(), d_req_key_of, (size_t)n_req * 8, hipMemcpyDeviceToHost));
                    ranges.resize((size_t)n_req * LBRB_MAX_Q * R);
                    LB_TRY(hipMemcpy(ranges.data(), d_req_ranges, ranges.size() * sizeof(float), hipMemcpyDeviceToHost));
                    for (int r = 0; r < n_req; ++r) {
                        const int32_t* m = &meta[(size_t)r * 8];
                        int c1 = m[0], c2 = 0;
                        if (nh == 2) prl_hole_cards_2(m[0], rules->n_cards, &c1, &c2);
                        const int8_t hand[2] = {(int8_t)c1, (int8_t)c2};
                        int8_t board[5];
                        for (int i = 0; i < 5; ++i) board[i] = (int8_t)(m[3 + i] < 0 ? 0 : m[3 + i]);
                        float wp[LBRB_MAX_Q];
                        memset(wp, 0, sizeof(wp));
                        // the stand-alone equity of the host worker's own call at this decision (LocalLBRWorker._checkdown_equity -> prl_lbr_checkdown_equity)
                        rc = prl_lbr_checkdown_equity(rules, board, m[2], hand, &ranges[(size_t)r * LBRB_MAX_Q * R], m[1], wp);
                        if (rc != PRL_OK) goto done;
                        insert(key_of[r], wp);
                    }
                    n_requests += n_req;
                }
                list.swap(next);
            }
            if (getenv("PRL_LBRB_DEBUG")) fprintf(stderr, "lbrb: %d rounds, %lld equity requests, %u cached keys\n", n_rounds, n_requests, c_used);
        }
    }
    if (out_device_ms) *out_device_ms = total_ms;
    LB_TRY(hipMemcpy(out_winnings, d_win, (size_t)n_envs * sizeof(float), hipMemcpyDeviceToHost));
    if (out_stats4) LB_TRY(hipMemcpy(out_stats4, d_stats, 4 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#ifdef PRL_LBRB_TIMING
    {
        unsigned long long h[LBRB_N_STATS];
        LB_TRY(hipMemcpy(h, d_stats, sizeof(h), hipMemcpyDeviceToHost));
        static const char* names[10] = {"reset + range init", "look-ahead set-up", "fold probs + classify", "fold / not-fold sums", "candidate ranges",
                                        "(range, board) equities", "board probs + reduction", "utilities + arg-max", "agent action + range + step", "after step"};
        double tot = 0;
        for (int i = 0; i < 10; ++i) tot += (double)h[4 + i];
        for (int i = 0; i < 10; ++i) fprintf(stderr, "lbrb phase %-28s %6.2f %%  %12.0f clk per hand\n", names[i], 100.0 * h[4 + i] / tot, (double)h[4 + i] / n_envs);
    }
#endif
#undef LB_TRY
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_cards); (void)hipFree(d_win); (void)hipFree(d_stats); (void)hipFree(d_eq);
    (void)hipFree(d_list); (void)hipFree(d_status); (void)hipFree(d_n_req); (void)hipFree(d_req_meta); (void)hipFree(d_req_keys); (void)hipFree(d_req_key_of);
    (void)hipFree(d_req_ranges); (void)hipFree(d_pf_keys); (void)hipFree(d_pf_wp);
    return rc;
}

extern "C" int32_t prl_lbr_batch_run(const PrlGame* lbr_game, const PrlGame* agent_game, const PrlRules* rules, int32_t n_envs, int32_t agent_seat,
                                     int32_t check_to_round, int32_t agent_kind, uint32_t agent_seed, uint32_t episode_base, double reward_scalar,
                                     double ev_normalizer, const int8_t* cards, float* out_winnings, uint64_t* out_stats4, float* out_device_ms) {
    if (agent_kind == 2) { prl_set_error("prl_lbr_batch_run: a tabular agent goes through prl_lbr_batch_run_table"); return PRL_ERR_ARG; }
    return lbr_batch_run_impl(lbr_game, agent_game, rules, n_envs, agent_seat, check_to_round, agent_kind, agent_seed, episode_base, reward_scalar, ev_normalizer, cards,
                              out_winnings, out_stats4, out_device_ms, nullptr);
}

// LBR against a tabular agent (kind 2): `table` holds the agent's policy (a solver's average strategy); agent_seed drives the action draws as before
extern "C" int32_t prl_lbr_batch_run_table(const PrlGame* lbr_game, const PrlGame* agent_game, const PrlRules* rules, int32_t n_envs, int32_t agent_seat,
                                           int32_t check_to_round, const PrlPolicyTable* table, uint32_t agent_seed, uint32_t episode_base, double reward_scalar,
                                           double ev_normalizer, const int8_t* cards, float* out_winnings, uint64_t* out_stats4, float* out_device_ms) {
    return lbr_batch_run_impl(lbr_game, agent_game, rules, n_envs, agent_seat, check_to_round, 2, agent_seed, episode_base, reward_scalar, ev_normalizer, cards,
                              out_winnings, out_stats4, out_device_ms, table);
}

// ---------------------------------------------------------------------------------------------------------------------
// Batched head-to-head (LocalHead2HeadMaster.py:82-126 on the batched env): two synthetic agents
// play n_envs hands, ONE LANE PER HAND -- a head-to-head hand touches only the two players' own rows of the policy, so the
// whole episode (betting engine, the acting agent's probabilities for its hand, the action draw, dealing, payout with the
// 7-card ranks) is scalar work; hands of a wave diverge freely. Same agents, draws and float32 arithmetic as two
// tests/lbr_fixture_agent.py HashAgents (modes "HASH" / "HASH2") under pokerrl_amd.eval.head_to_head.LocalHead2HeadMaster.
// ---------------------------------------------------------------------------------------------------------------------
struct PrlH2hBatchParams {
    PrlGame game;
    PrlRules rules;
    int32_t n_envs, ref_seat, n_deal, limit;
    int32_t kind[2];              // [0] the reference agent (its winnings are reported), [1] the opponent
    uint32_t seed[2];
    uint32_t episode_base;
    double reward_scalar, ev_normalizer;
    const int8_t* cards;          // [n_envs][n_deal]: seat 0's hole cards, seat 1's, then the board in deal order
    float* winnings;              // [n_envs]
    unsigned long long* stats;    // [2] env steps, showdowns
    PrlPolicyTable tab[2];        // agents of kind 2
};

template <bool TABLE>
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(256) prl_k_h2h_batch(PrlH2hBatchParams P) {
    const int e = (int)(prl_bid() * prl_nthreads() + prl_tid());
    unsigned long long n_steps = 0, n_show = 0;
    if (e < P.n_envs) {
        const int nh = P.rules.n_hole_cards, nb = P.rules.n_board_cards;
        const int8_t* cards = P.cards + (size_t)e * P.n_deal;
        const int8_t* deck_board = cards + 2 * nh;
        const uint32_t episode = P.episode_base + (uint32_t)e + 1u;
        PrlEnvState st;
        prl_env_reset(P.game, st);
        int8_t board[5] = {0, 0, 0, 0, 0};
        int n_dealt = 0;
        int step_ctr[2] = {0, 0};  // per agent: its own get_action calls of this episode
        LbrbHistKey hk[2];         // per tabular agent: the history key of the hand so far under its table's seed
        if (TABLE)
            for (int w = 0; w < 2; ++w) hk[w] = lbrb_hist_step(lbrb_hist_root(P.tab[w].key_seed), st, board, 0, nb, P.rules.n_suits);
        int hand_idx[2], hand_idx_c[2];  // ... _c: in the labelling of a table keyed under suit-canonical boards (prl_policy.h), once the board is out
        for (int p = 0; p < 2; ++p) hand_idx_c[p] = hand_idx[p] = lbrb_hand_idx(P.rules, cards + p * nh);
        int8_t cboard[5] = {0, 0, 0, 0, 0};
        const bool any_canon = TABLE && nh == 2 && ((P.kind[0] == 2 && P.tab[0].suit_canon) || (P.kind[1] == 2 && P.tab[1].suit_canon));
        PrlLbrGame hg;
        hg.n_hole = nh; hg.n_cards = P.rules.n_cards; hg.n_suits = P.rules.n_suits; hg.rank_rule = P.rules.rank_rule; hg.R = P.rules.range_size;
        hg.n_board_total = nb;
        for (bool done = false; !done;) {
            const int seat = st.cur;
            const int who = seat == P.ref_seat ? 0 : 1;
            int32_t legal[LBRB_MAX_LEGAL];
            const int n_legal = prl_legal_actions(P.game, st, legal);
            const uint32_t key = lbrb_state_key(P.seed[who], st, board, n_dealt, nb, P.rules.n_suits);
            const uint32_t x = lbrb_mix32(P.seed[who] * 0x51ED27u + episode * 0x9E3779B1u + (uint32_t)step_ctr[who]);
            const float u = (float)(x >> 8) / 16777216.0f;
            step_ctr[who] += 1;
            const int row = TABLE && P.kind[who] == 2 ? lbrb_table_row(P.tab[who], hk[who]) : -1;
            const int a = lbrb_draw<TABLE>(P.tab[who], P.kind[who], key, row, hand_idx[seat], TABLE && P.tab[who].suit_canon ? hand_idx_c[seat] : hand_idx[seat], legal, n_legal, u);
            PrlStepInfo info;
            if (!P.limit && a >= 2) {  // discretized games step by pot fraction
                const int amt = prl_fraction_of_pot_raise(st, P.game.bet_fracs[a - 2], st.cur);
                prl_env_step_processed(P.game, st, PRL_BET_RAISE, amt, &info);
            } else prl_env_step(P.game, st, a, &info);
            n_steps += 1;
            if (info.is_terminal) {
                if (info.rundown)
                    for (; n_dealt < nb; ++n_dealt) board[n_dealt] = deck_board[n_dealt];
                const int pot = st.main_pot, rs = P.ref_seat;
                double award = 0.0;
                if (st.folded[0] || st.folded[1]) award = st.folded[rs] ? 0.0 : (double)pot;
                else {
                    const int32_t r0 = prl_lbr_rank(hg, hand_idx[rs], board), r1 = prl_lbr_rank(hg, hand_idx[1 - rs], board);
                    award = r0 > r1 ? (double)pot : (r0 < r1 ? 0.0 : (double)pot / 2.0);
                    n_show += 1;
                }
                const double rew = ((double)st.stack[rs] + award - (double)P.game.start_stack[rs]) / P.reward_scalar;  // PokerEnv.py:1069-1072
                P.winnings[e] = (float)(rew * P.reward_scalar * P.ev_normalizer);                                     // LocalHead2HeadMaster.py:122-124
                done = true;
            } else if (info.chance_acts) {
                const int n_new = P.rules.board_cards_in_round[st.round];
                for (int i = 0; i < n_new; ++i) { board[n_dealt] = deck_board[n_dealt]; n_dealt += 1; }
                if (any_canon) {
                    const int ck = prl_suit_canon(board, n_dealt, P.rules.n_suits, cboard);
                    for (int p = 0; p < 2; ++p) {
                        const int8_t* hc = cards + p * nh;
                        hand_idx_c[p] = prl_suit_perm_hand(hc[0], hc[1], ck, P.rules.n_suits, P.rules.n_cards);
                    }
                }
            }
            if (TABLE && !info.is_terminal)
                for (int w = 0; w < 2; ++w) hk[w] = lbrb_hist_step(hk[w], st, P.tab[w].suit_canon ? cboard : board, n_dealt, nb, P.rules.n_suits);
        }
    }
    // one pair of atomics per wave: butterfly sum of the per-lane counts (each far below 2^31)
    int cs = (int)n_steps, cw = (int)n_show;
    const int lane = (int)(prl_tid() & 63);
    for (int off = 32; off > 0; off >>= 1) { cs += prl_shfl_i(cs, lane ^ off); cw += prl_shfl_i(cw, lane ^ off); }
    if (lane == 0) { prl_atomic_add_u64(P.stats + 0, (unsigned long long)cs); prl_atomic_add_u64(P.stats + 1, (unsigned long long)cw); }
}

static int32_t h2h_batch_run_impl(const PrlGame* game, const PrlRules* rules, int32_t n_envs, int32_t ref_seat, int32_t ref_kind, uint32_t ref_seed,
                                  int32_t opp_kind, uint32_t opp_seed, uint32_t episode_base, double reward_scalar, double ev_normalizer,
                                  const int8_t* cards, float* out_winnings, uint64_t* out_stats2, float* out_device_ms, const PrlPolicyTable* ref_table,
                                  const PrlPolicyTable* opp_table) {
    if (!game || !rules || !cards || !out_winnings || n_envs <= 0 || ref_seat < 0 || ref_seat > 1 || ref_kind < 0 || ref_kind > 2 || opp_kind < 0 || opp_kind > 2) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (const int trc = lbrb_check_table(ref_kind, ref_table, game, rules, "batched head-to-head")) return trc;
    if (const int trc = lbrb_check_table(opp_kind, opp_table, game, rules, "batched head-to-head")) return trc;
    if (!prl_device_available()) { prl_set_error("no HIP device: batched head-to-head has no CPU fallback"); return PRL_ERR_NO_DEVICE; }
    const int nh = rules->n_hole_cards, nb = rules->n_board_cards;
    if (nh < 1 || nh > 2 || rules->n_cards > PRL_LBR_MAX_CARDS || nb > 5 || (nh == 2 && (rules->n_cards != 52 || nb != 5))) {
        prl_set_error("batched head-to-head: 1-hole-card games or 52-card hold'em"); return PRL_ERR_UNSUPPORTED;
    }
    if (game->game_type == PRL_GAME_NOLIMIT) { prl_set_error("batched head-to-head: fixed-limit or discretized games"); return PRL_ERR_UNSUPPORTED; }
    if (game->n_bet_sizes + 2 > LBRB_MAX_LEGAL) { prl_set_error("batched head-to-head: at most 14 bet sizes"); return PRL_ERR_UNSUPPORTED; }
    PrlH2hBatchParams P;
    memset(&P, 0, sizeof(P));
    P.game = *game; P.rules = *rules; P.n_envs = n_envs; P.ref_seat = ref_seat; P.n_deal = 2 * nh + nb; P.limit = game->game_type == PRL_GAME_LIMIT;
    P.kind[0] = ref_kind; P.kind[1] = opp_kind; P.seed[0] = ref_seed; P.seed[1] = opp_seed;
    if (ref_kind == 2) P.tab[0] = *ref_table;
    if (opp_kind == 2) P.tab[1] = *opp_table;
    P.episode_base = episode_base; P.reward_scalar = reward_scalar; P.ev_normalizer = ev_normalizer;
    int8_t* d_cards = nullptr; float* d_win = nullptr; unsigned long long* d_stats = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PRL_OK;
#define HB_TRY(x) do { if ((x) != hipSuccess) { prl_set_error("HIP error in prl_h2h_batch_run"); rc = PRL_ERR_HIP; goto done; } } while (0)
    HB_TRY(hipMalloc((void**)&d_cards, (size_t)n_envs * P.n_deal));
    HB_TRY(hipMalloc((void**)&d_win, (size_t)n_envs * sizeof(float)));
    HB_TRY(hipMalloc((void**)&d_stats, 2 * sizeof(unsigned long long)));
    HB_TRY(hipMemcpy(d_cards, cards, (size_t)n_envs * P.n_deal, hipMemcpyHostToDevice));
    HB_TRY(hipMemset(d_stats, 0, 2 * sizeof(unsigned long long)));
    P.cards = d_cards; P.winnings = d_win; P.stats = d_stats;
    HB_TRY(hipEventCreate(&e0));
    HB_TRY(hipEventCreate(&e1));
    HB_TRY(hipEventRecord(e0, nullptr));
    if (ref_kind == 2 || opp_kind == 2) PRL_LAUNCH(prl_k_h2h_batch<true>, (n_envs + 255) / 256, 256, 0, nullptr, P);
    else PRL_LAUNCH(prl_k_h2h_batch<false>, (n_envs + 255) / 256, 256, 0, nullptr, P);
    HB_TRY(hipEventRecord(e1, nullptr));
    HB_TRY(hipDeviceSynchronize());
    if (out_device_ms) HB_TRY(hipEventElapsedTime(out_device_ms, e0, e1));
    HB_TRY(hipMemcpy(out_winnings, d_win, (size_t)n_envs * sizeof(float), hipMemcpyDeviceToHost));
    if (out_stats2) HB_TRY(hipMemcpy(out_stats2, d_stats, 2 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
#undef HB_TRY
done:
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    (void)hipFree(d_cards); (void)hipFree(d_win); (void)hipFree(d_stats);
    return rc;
}

extern "C" int32_t prl_h2h_batch_run(const PrlGame* game, const PrlRules* rules, int32_t n_envs, int32_t ref_seat, int32_t ref_kind, uint32_t ref_seed,
                                     int32_t opp_kind, uint32_t opp_seed, uint32_t episode_base, double reward_scalar, double ev_normalizer,
                                     const int8_t* cards, float* out_winnings, uint64_t* out_stats2, float* out_device_ms) {
    if (ref_kind == 2 || opp_kind == 2) { prl_set_error("prl_h2h_batch_run: tabular agents go through prl_h2h_batch_run_tables"); return PRL_ERR_ARG; }
    return h2h_batch_run_impl(game, rules, n_envs, ref_seat, ref_kind, ref_seed, opp_kind, opp_seed, episode_base, reward_scalar, ev_normalizer, cards, out_winnings,
                              out_stats2, out_device_ms, nullptr, nullptr);
}

// head-to-head with tabular agents: a NULL table keeps that side's synthetic agent (its kind argument), a table makes it kind 2
extern "C" int32_t prl_h2h_batch_run_tables(const PrlGame* game, const PrlRules* rules, int32_t n_envs, int32_t ref_seat, int32_t ref_kind, uint32_t ref_seed,
                                            const PrlPolicyTable* ref_table, int32_t opp_kind, uint32_t opp_seed, const PrlPolicyTable* opp_table,
                                            uint32_t episode_base, double reward_scalar, double ev_normalizer, const int8_t* cards, float* out_winnings,
                                            uint64_t* out_stats2, float* out_device_ms) {
    return h2h_batch_run_impl(game, rules, n_envs, ref_seat, ref_table ? 2 : ref_kind, ref_seed, opp_table ? 2 : opp_kind, opp_seed, episode_base, reward_scalar,
                              ev_normalizer, cards, out_winnings, out_stats2, out_device_ms, ref_table, opp_table);
}

// ---------------------------------------------------------------------------------------------------------------------
// Counter-based decks for the batched engines: hand i's cards depend on (seed, first_hand + i) only, so any split of the
// hands over GPUs deals the same cards. A partial Fisher-Yates shuffle of 0..n_cards-1 driven by a SplitMix-style hash, one
// lane per hand (pokerrl_amd/eval/lbr/BatchedLBR.py: deal_decks_host is the same algorithm in NumPy).
// ---------------------------------------------------------------------------------------------------------------------
PRL_GLOBAL void PRL_LAUNCH_BOUNDS(256) prl_k_deal_decks(int n_hands, int n_cards, int n_deal, unsigned long long seed, unsigned long long first_hand, int8_t* out) {
    const int i = (int)(prl_bid() * prl_nthreads() + prl_tid());
    if (i >= n_hands) return;
    prl_deal_hand(n_cards, n_deal, seed, first_hand + (unsigned long long)i, out + (size_t)i * n_deal);  // prl_cards.h
}

extern "C" int32_t prl_deal_decks(int32_t n_hands, int32_t n_cards_in_deck, int32_t n_deal, uint64_t seed, uint64_t first_hand, int8_t* out_cards) {
    if (!out_cards || n_hands <= 0 || n_deal <= 0 || n_deal > 8 * 2 || n_deal > n_cards_in_deck || n_cards_in_deck > 127) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device"); return PRL_ERR_NO_DEVICE; }
    int8_t* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)n_hands * n_deal) != hipSuccess) { prl_set_error("hipMalloc failed"); return PRL_ERR_HIP; }
    PRL_LAUNCH(prl_k_deal_decks, (n_hands + 255) / 256, 256, 0, nullptr, n_hands, n_cards_in_deck, n_deal, (unsigned long long)seed, (unsigned long long)first_hand, d);
    const hipError_t e = hipMemcpy(out_cards, d, (size_t)n_hands * n_deal, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    if (e != hipSuccess) { prl_set_error("HIP error in prl_deal_decks"); return PRL_ERR_HIP; }
    return PRL_OK;
}
