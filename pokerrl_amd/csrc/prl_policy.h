// Tabular policies in HBM (agent kind 2 of the batched evaluators): the table, the history-key chain that addresses its rows, and the suit
// canonicalisation of boards for tables that hold one row set per SUIT CLASS of boards (a table made from a whole-game solve of a single-deal
// hold'em game: prl_policy_table_from_solver on a prl_solver_create_weighted solver). Shared by the evaluator kernels (prl_lbr_batch.hip) and the
// table builder (prl_solver.hip); host and device. What the reference's evaluators get from EvalAgentBase.get_a_probs_for_each_hand / get_action
// (PokerRL/rl/base_cls/EvalAgentBase.py:35-62) when the agent is tabular.
#pragma once
#include "prl_cards.h"
#include "prl_defs.h"
#include "prl_env.h"

struct PrlPolicyTable {
    unsigned long long* keys;     // [mask + 1] 64-bit state keys, 0 = empty slot (linear probing)
    int32_t* rows;                // [mask + 1] row of the key in the same slot
    float* probs;                 // [n_rows][n_actions][range_size] P(action | hand); 0 for actions that are not legal in the row's state
    uint32_t mask, key_seed;
    int32_t n_rows, n_actions, range_size;
    int32_t suit_canon;           // 1: rows are keyed under the suit-canonical board and hold the hands of THAT labelling (prl_suit_canon)
};
#define LBRB_KEY_SEED_HI 0x5BD1E995u  // the high word of a state's key is the hash chain under key_seed ^ this

PRL_HD PRL_INLINE uint32_t lbrb_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}

// hash chain over the public state (tests/lbr_fixture_agent.py: state_key)
PRL_HD PRL_INLINE uint32_t lbrb_state_key(uint32_t seed, const PrlEnvState& s, const int8_t* board, int n_dealt, int n_board_total, int n_suits) {
    uint32_t k = seed;
    const int vals[7] = {s.round, s.main_pot, s.bet[0], s.bet[1], s.stack[0], s.stack[1], (int)s.cur};
    for (int i = 0; i < 7; ++i) k = lbrb_mix32(k * 31u + (uint32_t)vals[i]);
    for (int i = 0; i < n_board_total; ++i) {  // 2-D cards (rank, suit); the not-dealt token is -127 in both fields
        const int r = i < n_dealt ? board[i] / n_suits : -127, su = i < n_dealt ? board[i] % n_suits : -127;
        k = lbrb_mix32(k * 31u + (uint32_t)(r & 0xFF));
        k = lbrb_mix32(k * 31u + (uint32_t)(su & 0xFF));
    }
    return k;
}

// A policy row belongs to a node of the agent's public TREE, not to a public state (two betting histories can meet in one state), so the
// key is a chain over the states of the hand so far: key(root) = lbrb_state_key(key_seed, root state), key(next) = lbrb_state_key(key(now), next state)
// after every env step (with the new board cards on the table when the step ends a round) -- two 32-bit chains make the 64-bit key. States, not action
// ids: LBR raising by a pot fraction of ITS bet set reaches the agent's node whenever the agent's tree has a raise to the same amount.
struct LbrbHistKey { uint32_t lo, hi; };

PRL_HD PRL_INLINE LbrbHistKey lbrb_hist_root(uint32_t key_seed) { return LbrbHistKey{key_seed, key_seed ^ LBRB_KEY_SEED_HI}; }

PRL_HD PRL_INLINE LbrbHistKey lbrb_hist_step(const LbrbHistKey& k, const PrlEnvState& s, const int8_t* board, int n_dealt, int n_board_total, int n_suits) {
    return LbrbHistKey{lbrb_state_key(k.lo, s, board, n_dealt, n_board_total, n_suits), lbrb_state_key(k.hi, s, board, n_dealt, n_board_total, n_suits)};
}

PRL_HD PRL_INLINE unsigned long long lbrb_key64(const LbrbHistKey& hk) {
    const unsigned long long k = ((unsigned long long)hk.hi << 32) | hk.lo;
    return k == 0ull ? 1ull : k;
}
PRL_HD PRL_INLINE uint32_t lbrb_first_slot(const LbrbHistKey& hk, uint32_t mask) { return (hk.lo ^ (hk.hi * 0x9E3779B1u)) & mask; }

// the row of a history key in the table, -1 if the table does not hold it (one lane; a handful of dependent HBM reads per agent decision)
PRL_HD PRL_INLINE int lbrb_table_row(const PrlPolicyTable& T, const LbrbHistKey& hk) {
    const unsigned long long k = lbrb_key64(hk);
    for (uint32_t i = lbrb_first_slot(hk, T.mask);; i = (i + 1u) & T.mask) {  // the table is never full: the loop ends at an empty slot
        const unsigned long long ki = T.keys[i];
        if (ki == k) return T.rows[i];
        if (ki == 0ull) return -1;
    }
}

// ---- suit canonicalisation ---------------------------------------------------------------------------------------------------------------------
// A suit permutation p relabels card (rank, suit) as (rank, p[suit]). The CANONICAL form of a board is the lexicographically smallest board --
// cards ascending -- among its n_suits! relabellings (pokerrl_amd/game/board_enum.py: single_deal_board_classes lists exactly these as the class
// representatives); the permutation chosen is the FIRST one, in lexicographic order of (p[0], p[1], ...), that attains it. (A board with a non-trivial
// stabiliser has several; a solver's columns are not bit-symmetric under it -- the summation order differs between a hand and its image -- so both
// sides of every comparison must take the same one.) Permutation number k in that order, n_suits <= 4:
PRL_HD PRL_INLINE void prl_suit_perm(int k, int n_suits, int* p /*[n_suits]*/) {
    int fact = 1;
    for (int i = 2; i < n_suits; ++i) fact *= i;  // (n_suits - 1)!
    unsigned used = 0u;
    for (int i = 0; i < n_suits; ++i) {
        int d = k / fact;
        k -= d * fact;
        if (n_suits - 1 - i > 0) fact /= (n_suits - 1 - i);
        int s = 0;
        for (;; ++s) {  // the d-th suit not used yet
            if ((used >> s) & 1u) continue;
            if (d-- == 0) break;
        }
        p[i] = s;
        used |= 1u << s;
    }
}
PRL_HD PRL_INLINE int prl_n_suit_perms(int n_suits) { int f = 1; for (int i = 2; i <= n_suits; ++i) f *= i; return f; }

// board[0..n): 1d cards in any order -> out[0..n): the canonical board, cards ascending; returns the number of the permutation taken. n <= 5.
PRL_HD PRL_INLINE int prl_suit_canon(const int8_t* board, int n, int n_suits, int8_t* out) {
    int best_k = 0;
    int best[5] = {127, 127, 127, 127, 127};
    const int n_perm = prl_n_suit_perms(n_suits);
    for (int k = 0; k < n_perm; ++k) {
        int p[4] = {0, 1, 2, 3};
        prl_suit_perm(k, n_suits, p);
        int c[5] = {127, 127, 127, 127, 127};
        for (int i = 0; i < 5; ++i) {
            if (i >= n) continue;
            const int card = board[i], r = card / n_suits, s = card - r * n_suits;
            int su = p[0];
            if (s == 1) su = p[1];
            if (s == 2) su = p[2];
            if (s == 3) su = p[3];
            c[i] = r * n_suits + su;
        }
        // five entries (unused ones 127): a fixed sorting network on registers
#define PRL_CSWAP(a, b) do { const int lo_ = c[a] < c[b] ? c[a] : c[b], hi_ = c[a] < c[b] ? c[b] : c[a]; c[a] = lo_; c[b] = hi_; } while (0)
        PRL_CSWAP(0, 1); PRL_CSWAP(3, 4); PRL_CSWAP(2, 4); PRL_CSWAP(2, 3); PRL_CSWAP(1, 4); PRL_CSWAP(0, 3); PRL_CSWAP(0, 2); PRL_CSWAP(1, 3); PRL_CSWAP(1, 2);
#undef PRL_CSWAP
        bool less = false, decided = false;
        for (int i = 0; i < 5; ++i) {
            if (!decided && c[i] != best[i]) { less = c[i] < best[i]; decided = true; }
        }
        if (less) {
            for (int i = 0; i < 5; ++i) best[i] = c[i];
            best_k = k;
        }
    }
    for (int i = 0; i < 5; ++i)
        if (i < n) out[i] = (int8_t)best[i];
    return best_k;
}

// the range index of 2-card hand h under suit permutation number k (cards relabelled, the pair re-sorted)
PRL_HD PRL_INLINE int prl_suit_perm_hand(int c1, int c2, int k, int n_suits, int n_cards) {
    int p[4] = {0, 1, 2, 3};
    prl_suit_perm(k, n_suits, p);
    const int r1 = c1 / n_suits, s1 = c1 - r1 * n_suits, r2 = c2 / n_suits, s2 = c2 - r2 * n_suits;
    const int q1 = s1 == 0 ? p[0] : (s1 == 1 ? p[1] : (s1 == 2 ? p[2] : p[3])), q2 = s2 == 0 ? p[0] : (s2 == 1 ? p[1] : (s2 == 2 ? p[2] : p[3]));
    const int a = r1 * n_suits + q1, b = r2 * n_suits + q2;
    return a < b ? prl_range_idx_2(a, b, n_cards) : prl_range_idx_2(b, a, n_cards);
}

// host: the device arrays of a table from its host key table; probs == nullptr leaves the probabilities zeroed for the caller to fill on the device
// (prl_lbr_batch.hip; prl_policy_table_create is this with probabilities, prl_policy_table_from_solver fills them from a solver's columns)
PrlPolicyTable* prl_policy_table_alloc(const uint64_t* keys, const int32_t* rows, uint32_t capacity, const float* probs, int32_t n_rows, int32_t n_actions,
                                       int32_t range_size, uint32_t key_seed);
