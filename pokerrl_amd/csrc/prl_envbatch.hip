// Batched heads-up PokerEnv, device resident: the public betting state of n_envs independent envs as struct-of-arrays in
// HBM, one lane per env.
//
// Reference semantics (integer-exact, the same POD engine as the single-env host entry points -- prl_env.h):
//   reset                PokerEnv.reset (public part)                            PokerEnv.py:1075-1122
//   step / step_processed  PokerEnv._step + _get_fixed_action + env-specific action formulation
//                          PokerEnv.py:681-789,885-941; LimitPokerEnv.py:27-59; DiscretizedPokerEnv.py:44-135
//   legal_masks          get_legal_actions                                       LimitPokerEnv.py:41-59, DiscretizedPokerEnv.py:99-135,
//                                                                                PokerEnv.py:1313-1330
//   get_state            the public part of state_dict()                         PokerEnv.py:1161-1197
// Cards are not part of this object (as in prl_env.h: on `chance_acts` the caller deals, on a showdown the caller ranks the
// hands -- prl_hand_rank_boards_device): what is batched here is the branchy integer state machine.
//
// Layout: PRL_EB_N_COLS int32 columns of n_envs entries each (column-major: the lanes of a wave touch consecutive words of one
// column -> every access is one coalesced 256-byte transaction). Small fields are packed (flag bits, the three "who raised"
// seats), so a step moves 13 words in and 13 out per env.
// Wave-level primitives: the list of still-running envs is compacted with ballot + popcount prefix (one atomic per wave), the
// usual front half of an agent query over a ragged batch; legal actions come back as a 128-bit mask per env.
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "prl_cards.h"
#include "prl_device.h"
#include "prl_env.h"
#include "prl_lbr.h"
#include "prl_rt.h"

enum {
    EB_ROUND = 0, EB_POT = 1, EB_BET0 = 2, EB_BET1 = 3, EB_STACK0 = 4, EB_STACK1 = 5,
    EB_FLAGS = 6,    // bit 0,1 allin; 2,3 folded; 4,5 acted; 6 cur; 7 capped_happened; 8 done (episode over)
    EB_SEATS = 7,    // (last_raiser + 1) | (capped_raiser + 1) << 8 | (capped_cant_reopen + 1) << 16
    EB_NACT = 8, EB_NRAISES = 9, EB_LA_TYPE = 10, EB_LA_AMOUNT = 11, EB_LA_SEAT = 12
};
static_assert(PRL_EB_N_COLS == 13, "include/pokerrl_hip.h");

struct prl_envbatch {
    PrlGame game;
    int32_t n = 0;
    int32_t* d_state = nullptr;   // [PRL_EB_N_COLS][n]
    PrlGame* d_game = nullptr;
    int32_t *d_a = nullptr, *d_b = nullptr, *d_info = nullptr;  // staging for the host-pointer entry points
    uint32_t* d_mask = nullptr;
    int32_t* d_count = nullptr;
    unsigned long long* d_stats = nullptr;  // [EB_STAT_SLOTS][3]
    hipStream_t stream = nullptr;
    // ---- the whole PokerEnv.step (prl_envbatch_create_with_cards): cards, payouts, rewards, observations ----
    bool with_cards = false;
    PrlRules rules{};
    int32_t n_deal = 0, obs_dim = 0;
    uint64_t deck_seed = 0;
    double reward_scalar = 1.0;
    int8_t* d_cards = nullptr;      // [n][n_deal]: hole cards of seat 0, of seat 1, the board in deal order (1-d cards)
    uint32_t* d_episode = nullptr;  // [n] episodes dealt so far (the deck counter of env i: episode * n + i)
    float* d_obs = nullptr;         // [n][obs_dim] staging for the host-pointer entry points
    double* d_rew = nullptr;        // [n][2]
    uint8_t* d_done = nullptr;      // [n]
    int32_t full_cap = 0;           // workgroups of the kernels with observation rows the device holds at once (eb_grid_full)
};

// the 13 words of env i as they lie in HBM, and their unpacking: two steps, so that a persistent workgroup can have its NEXT chunk's words
// in flight while it writes this chunk's observation vectors (prl_k_ebf_random_step)
PRL_DEV PRL_INLINE void eb_load_raw(const int32_t* st, int n, int i, int32_t (&w)[PRL_EB_N_COLS]) {
#if defined(__clang__)
#pragma unroll
#endif
    for (int c = 0; c < PRL_EB_N_COLS; ++c) w[c] = st[(size_t)c * n + i];
}
PRL_HD PRL_INLINE void eb_unpack(const int32_t (&w)[PRL_EB_N_COLS], PrlEnvState& s, bool* done) {
    s.round = w[EB_ROUND];
    s.main_pot = w[EB_POT];
    s.bet[0] = w[EB_BET0];
    s.bet[1] = w[EB_BET1];
    s.stack[0] = w[EB_STACK0];
    s.stack[1] = w[EB_STACK1];
    const int f = w[EB_FLAGS];
    s.allin[0] = f & 1; s.allin[1] = (f >> 1) & 1;
    s.folded[0] = (f >> 2) & 1; s.folded[1] = (f >> 3) & 1;
    s.acted[0] = (f >> 4) & 1; s.acted[1] = (f >> 5) & 1;
    s.cur = (int8_t)((f >> 6) & 1);
    s.capped_happened = (int8_t)((f >> 7) & 1);
    *done = ((f >> 8) & 1) != 0;
    const int x = w[EB_SEATS];
    s.last_raiser = (int8_t)((x & 0xFF) - 1);
    s.capped_raiser = (int8_t)(((x >> 8) & 0xFF) - 1);
    s.capped_cant_reopen = (int8_t)(((x >> 16) & 0xFF) - 1);
    s.pad0 = 0;
    s.n_actions_ep = w[EB_NACT];
    s.n_raises_round = w[EB_NRAISES];
    s.last_action[0] = w[EB_LA_TYPE];
    s.last_action[1] = w[EB_LA_AMOUNT];
    s.last_action[2] = w[EB_LA_SEAT];
}
PRL_DEV PRL_INLINE void eb_load(const int32_t* st, int n, int i, PrlEnvState& s, bool* done) {
    int32_t w[PRL_EB_N_COLS];
    eb_load_raw(st, n, i, w);
    eb_unpack(w, s, done);
}

PRL_DEV PRL_INLINE void eb_store(int32_t* st, int n, int i, const PrlEnvState& s, bool done) {
    st[(size_t)EB_ROUND * n + i] = s.round;
    st[(size_t)EB_POT * n + i] = s.main_pot;
    st[(size_t)EB_BET0 * n + i] = s.bet[0];
    st[(size_t)EB_BET1 * n + i] = s.bet[1];
    st[(size_t)EB_STACK0 * n + i] = s.stack[0];
    st[(size_t)EB_STACK1 * n + i] = s.stack[1];
    st[(size_t)EB_FLAGS * n + i] = (s.allin[0] & 1) | ((s.allin[1] & 1) << 1) | ((s.folded[0] & 1) << 2) | ((s.folded[1] & 1) << 3) |
                                   ((s.acted[0] & 1) << 4) | ((s.acted[1] & 1) << 5) | ((s.cur & 1) << 6) | ((s.capped_happened & 1) << 7) |
                                   ((done ? 1 : 0) << 8);
    st[(size_t)EB_SEATS * n + i] = (s.last_raiser + 1) | ((s.capped_raiser + 1) << 8) | ((s.capped_cant_reopen + 1) << 16);
    st[(size_t)EB_NACT * n + i] = s.n_actions_ep;
    st[(size_t)EB_NRAISES * n + i] = s.n_raises_round;
    st[(size_t)EB_LA_TYPE * n + i] = s.last_action[0];
    st[(size_t)EB_LA_AMOUNT * n + i] = s.last_action[1];
    st[(size_t)EB_LA_SEAT * n + i] = s.last_action[2];
}

PRL_HD PRL_INLINE uint32_t eb_mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// legal actions of a state as a 128-bit mask over the env's action ints + their count
PRL_HD PRL_INLINE int eb_legal_mask(const PrlGame& g, const PrlEnvState& s, uint32_t m[4]) {
    int32_t legal[PRL_MAX_BET_SIZES + 2];
    const int n = prl_legal_actions(g, s, legal);
    m[0] = m[1] = m[2] = m[3] = 0u;
    for (int k = 0; k < n; ++k) m[legal[k] >> 5] |= 1u << (legal[k] & 31);
    return n;
}

// play statistics: every wave adds its lanes up with shuffles and does ONE atomic triple into one of EB_STAT_SLOTS slot triples
// (2^20 lanes adding into three words would serialise on them: ~12 ns per atomic); the host sums the slots
#define EB_STAT_SLOTS 256
PRL_DEV PRL_INLINE void eb_stats_add(unsigned long long* stats, unsigned long long steps, unsigned long long hands, unsigned long long pots) {
    for (int d = 32; d > 0; d >>= 1) {
        const int src = (int)prl_lane() ^ d;
        auto add64 = [&](unsigned long long& v) {  // 64-bit butterfly add through two 32-bit shuffles (no truncation of long rollouts)
            const unsigned lo = (unsigned)prl_shfl_i((int)(unsigned)(v & 0xFFFFFFFFull), src);
            const unsigned hi = (unsigned)prl_shfl_i((int)(unsigned)(v >> 32), src);
            v += ((unsigned long long)hi << 32) | lo;
        };
        add64(steps); add64(hands); add64(pots);
    }
    if (prl_lane() == 0) {
        unsigned long long* s = stats + 3 * (((prl_bid() * prl_nthreads() + prl_tid()) >> 6) & (EB_STAT_SLOTS - 1));
        prl_atomic_add_u64(s, steps);
        prl_atomic_add_u64(s + 1, hands);
        prl_atomic_add_u64(s + 2, pots);
    }
}

PRL_GLOBAL void prl_k_eb_reset(const PrlGame* __restrict__ g, int32_t* __restrict__ st, int n, const uint8_t* __restrict__ mask) {
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n; i += (int)(prl_nblocks() * prl_nthreads())) {
        if (mask && !mask[i]) continue;
        PrlEnvState s;
        prl_env_reset(*g, s);
        eb_store(st, n, i, s, false);
    }
}

// actions: env action ints (processed == 0) or (type, amount) pairs; an env whose episode is over, or whose action is < 0, is
// left untouched and reports info = {-1, 0, 0, 0}. info[4][n]: is_terminal, chance_acts, pot_before_payout, terminal kind
// (0 none, 1 fold, 2 showdown on the last street, 3 all-in run-out)
PRL_GLOBAL void prl_k_eb_step(const PrlGame* __restrict__ g, int32_t* __restrict__ st, int n, const int32_t* __restrict__ a0, const int32_t* __restrict__ a1, int processed, int32_t* __restrict__ info) {
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n; i += (int)(prl_nblocks() * prl_nthreads())) {
        PrlEnvState s;
        bool done;
        eb_load(st, n, i, s, &done);
        const int act = a0[i];
        int o0 = -1, o1 = 0, o2 = 0, o3 = 0;
        const int n_act = g->game_type == PRL_GAME_DISCRETIZED ? g->n_bet_sizes + 2 : 3;
        if (!done && act >= 0 && act < (processed ? 3 : n_act)) {
            PrlStepInfo si;
            if (processed) prl_env_step_processed(*g, s, act, a1[i], &si);
            else prl_env_step(*g, s, act, &si);
            done = si.is_terminal != 0;
            eb_store(st, n, i, s, done);
            o0 = si.is_terminal; o1 = si.chance_acts; o2 = si.pot_before_payout;
            o3 = si.is_terminal ? (si.terminal_is_fold ? 1 : (si.rundown ? 3 : 2)) : 0;
        }
        if (info) {
            info[i] = o0; info[(size_t)n + i] = o1; info[(size_t)2 * n + i] = o2; info[(size_t)3 * n + i] = o3;
        }
    }
}

PRL_GLOBAL void prl_k_eb_legal(const PrlGame* __restrict__ g, const int32_t* __restrict__ st, int n, uint32_t* __restrict__ mask4, int32_t* __restrict__ count) {
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n; i += (int)(prl_nblocks() * prl_nthreads())) {
        PrlEnvState s;
        bool done;
        eb_load(st, n, i, s, &done);
        uint32_t m[4] = {0u, 0u, 0u, 0u};
        const int c = done ? 0 : eb_legal_mask(*g, s, m);
        for (int k = 0; k < 4; ++k) mask4[(size_t)k * n + i] = m[k];
        count[i] = c;
    }
}

// ids of the envs whose episode is still running, ascending inside a wave's 64 envs, waves in arrival order: ballot of the
// predicate, popcount of the lanes below = the lane's slot, one atomic per wave for the wave's base
PRL_GLOBAL void prl_k_eb_active(const int32_t* __restrict__ st, int n, int32_t* __restrict__ out_idx, int32_t* __restrict__ out_count) {
    const int n_pad = (n + 63) & ~63;
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n_pad; i += (int)(prl_nblocks() * prl_nthreads())) {
        const bool live = i < n && ((st[(size_t)EB_FLAGS * n + i] >> 8) & 1) == 0;
        const unsigned long long b = prl_ballot(live);
        const int lane = (int)prl_lane();
        const int below = prl_popc64(b & ((1ull << lane) - 1ull));
        int base = 0;
        if (lane == 0 && b) base = prl_atomic_add_i(out_count, prl_popc64(b));
        base = prl_shfl_i(base, 0);
        if (live) out_idx[base + below] = i;
    }
}

// uniform-random legal play with a counter-based generator keyed by (seed, env, step): n_steps steps per env, an env that ends
// its hand is reset and keeps playing. stats[0] += steps, [1] += finished hands, [2] += sum of terminal pots (a checksum).
PRL_GLOBAL void prl_k_eb_rollout(const PrlGame* __restrict__ g, int32_t* __restrict__ st, int n, int n_steps, uint32_t seed, unsigned long long* __restrict__ stats) {
    unsigned long long steps = 0, hands = 0, pots = 0;
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n; i += (int)(prl_nblocks() * prl_nthreads())) {
        PrlEnvState s;
        bool done;
        eb_load(st, n, i, s, &done);
        if (done) { prl_env_reset(*g, s); done = false; }
        for (int k = 0; k < n_steps; ++k) {
            const uint32_t r = eb_mix32(seed ^ eb_mix32((uint32_t)i * 0x9E3779B9u + (uint32_t)k));
            PrlStepInfo si;
            const int a = prl_legal_action_pick(*g, s, r);
            if (g->game_type == PRL_GAME_NOLIMIT) prl_env_step_processed(*g, s, a, a == PRL_BET_RAISE ? (int)(eb_mix32(r) % (uint32_t)(prl_at2(s.stack, s.cur) + prl_at2(s.bet, s.cur) + 1)) : -1, &si);
            else prl_env_step(*g, s, a, &si);
            ++steps;
            if (si.is_terminal) {
                ++hands;
                pots += (unsigned long long)si.pot_before_payout;
                prl_env_reset(*g, s);
            }
        }
        eb_store(st, n, i, s, false);
    }
    eb_stats_add(stats, steps, hands, pots);
}

// the same play with the state in HBM between steps: ONE step per env and launch (what a rollout driven by an external agent
// costs per step: 13 words in, 13 out per env). Step k of env i draws the same number as step k of prl_k_eb_rollout.
PRL_GLOBAL void prl_k_eb_random_step(const PrlGame* __restrict__ g, int32_t* __restrict__ st, int n, int k, uint32_t seed, unsigned long long* __restrict__ stats) {
    unsigned long long hands = 0, pots = 0, steps = 0;
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n; i += (int)(prl_nblocks() * prl_nthreads())) {
        PrlEnvState s;
        bool done;
        eb_load(st, n, i, s, &done);
        if (done) prl_env_reset(*g, s);
        const uint32_t r = eb_mix32(seed ^ eb_mix32((uint32_t)i * 0x9E3779B9u + (uint32_t)k));
        PrlStepInfo si;
        const int a = prl_legal_action_pick(*g, s, r);
        if (g->game_type == PRL_GAME_NOLIMIT) prl_env_step_processed(*g, s, a, a == PRL_BET_RAISE ? (int)(eb_mix32(r) % (uint32_t)(prl_at2(s.stack, s.cur) + prl_at2(s.bet, s.cur) + 1)) : -1, &si);
        else prl_env_step(*g, s, a, &si);
        ++steps;
        if (si.is_terminal) { ++hands; pots += (unsigned long long)si.pot_before_payout; }
        eb_store(st, n, i, s, si.is_terminal != 0);
    }
    eb_stats_add(stats, steps, hands, pots);
}

// ---------------------------------------------------------------------------------------------------------------------
// the whole PokerEnv.step for n envs: cards, payout, rewards, observation vector
// ---------------------------------------------------------------------------------------------------------------------
struct EbFull {
    PrlRules rules;
    int32_t n_deal, obs_dim, suits_matter;
    double reward_scalar;
};
PRL_HD PRL_INLINE int eb_obs_dim(const PrlRules& r) { return 7 + 3 + 2 + 2 + r.n_rounds + 3 * 2 + r.n_board_cards * (r.n_ranks + r.n_suits); }
PRL_HD PRL_INLINE int eb_cards_out(const PrlRules& r, int round) {  // board cards on the table in `round`
    int n = 0;
    for (int k = 1; k <= round && k < 4; ++k) n += r.board_cards_in_round[k];
    return n;
}
PRL_HD PRL_INLINE int eb_hand_idx(const PrlRules& r, const int8_t* hc) {
    if (r.n_hole_cards == 1) return hc[0];
    const int a = hc[0] < hc[1] ? hc[0] : hc[1], b = hc[0] < hc[1] ? hc[1] : hc[0];
    return prl_range_idx_2(a, b, r.n_cards);
}
// The cards of one env in registers (round 6): the flat [n_deal] row of d_cards -- seat 0's hole cards, seat 1's, the board in deal order -- read
// ONCE, together with the state words (one memory round trip instead of one per consumer), every later index a compile-time constant
// (an array indexed at run time would live in scratch memory). Entries past n_deal are -1.
PRL_HD PRL_INLINE void eb_cards_load(const int8_t* p, int n_deal, int8_t (&c)[16]) {
#if defined(__clang__)
#pragma unroll
#endif
    for (int d = 0; d < 16; ++d) {  // no branch: the loads stay in one block with the state loads (a branch per card made the compiler drain them first)
        const int8_t v = p[d < n_deal ? d : 0];
        c[d] = d < n_deal ? v : (int8_t)-1;
    }
}
PRL_HD PRL_INLINE void eb_cards_store(int8_t* p, int n_deal, const int8_t (&c)[16]) {
#if defined(__clang__)
#pragma unroll
#endif
    for (int d = 0; d < 16; ++d)
        if (d < n_deal) p[d] = c[d];
}
// board card i (a constant after unrolling) of a 1- or 2-hole-card game
#define EB_BOARD(c, n_hole, i) ((n_hole) == 2 ? (c)[4 + (i)] : (c)[2 + (i)])
// PokerEnv._payout_pots, heads-up (PokerEnv.py:468-481) + the rewards of PokerEnv.py:1069-1072: (stack after - starting stack) / REWARD_SCALAR.
// The showdown ranks straight from the cards: what prl_lbr_rank(range index of the hole cards, board) returns (-1 for a hand that shares a
// card with the board), without the trip through the range index and back.
PRL_HD PRL_INLINE void eb_payout(const PrlGame& g, const EbFull& F, const PrlEnvState& s, const int8_t (&c)[16], double rew[2], bool* showdown) {
    const int pot = s.main_pot;
    double award0 = 0.0, award1 = 0.0;
    *showdown = false;
    if (s.folded[0]) award1 = (double)pot;
    else if (s.folded[1]) award0 = (double)pot;
    else {
        int32_t r0, r1;
        if (F.rules.n_hole_cards == 1) {
            const int bonus = F.rules.rank_rule == 1 ? 10000 : 100;
            r0 = prl_rank_leduc(c[0], c[2], F.rules.n_suits, bonus);
            r1 = prl_rank_leduc(c[1], c[2], F.rules.n_suits, bonus);
        } else {
            const int8_t board[5] = {c[4], c[5], c[6], c[7], c[8]};
            bool hit0 = false, hit1 = false;
            for (int i = 0; i < 5; ++i) { hit0 |= board[i] == c[0] || board[i] == c[1]; hit1 |= board[i] == c[2] || board[i] == c[3]; }
            r0 = hit0 ? -1 : prl_rank7_cards_52(board, c[0], c[1]);
            r1 = hit1 ? -1 : prl_rank7_cards_52(board, c[2], c[3]);
        }
        if (r0 > r1) award0 = (double)pot;
        else if (r0 < r1) award1 = (double)pot;
        else award0 = award1 = (double)pot / 2.0;
        *showdown = true;
    }
    rew[0] = ((double)s.stack[0] + award0 - (double)g.start_stack[0]) / F.reward_scalar;
    rew[1] = ((double)s.stack[1] + award1 - (double)g.start_stack[1]) / F.reward_scalar;
}
// the heads-up "simple" observation (PokerEnv.py:199-261, :989-1031, :1253-1271): float64 quotients rounded to float32 like the
// reference's np.array(list of Python floats, dtype=float32)
PRL_HD PRL_INLINE void eb_observation(const PrlGame& g, const EbFull& F, const PrlEnvState& s, const int8_t* cards, float* o, size_t stride) {
    const double norm = (double)(g.start_stack[0] + g.start_stack[1]) / 2.0;
    const int small = s.bet[0] < s.bet[1] ? s.bet[0] : s.bet[1], big = s.bet[0] < s.bet[1] ? s.bet[1] : s.bet[0];
    const int min_raise = big + ((big - small) > g.big_blind ? (big - small) : g.big_blind);
    const bool have_la = s.last_action[0] >= 0;
    int k = 0;
    auto put = [&](double v) { o[(size_t)k * stride] = (float)v; ++k; };
    put((double)g.ante / norm); put((double)g.small_blind / norm); put((double)g.big_blind / norm); put((double)min_raise / norm);
    put((double)s.main_pot / norm); put((double)big / norm); put(have_la ? (double)s.last_action[1] / norm : 0.0);
    for (int a = 0; a < 3; ++a) put(have_la && s.last_action[0] == a ? 1.0 : 0.0);
    for (int p = 0; p < 2; ++p) put(have_la && s.last_action[2] == p ? 1.0 : 0.0);
    for (int p = 0; p < 2; ++p) put(s.cur == p ? 1.0 : 0.0);
    for (int r = 0; r < F.rules.n_rounds; ++r) put(s.round == r ? 1.0 : 0.0);
    for (int p = 0; p < 2; ++p) { put((double)s.stack[p] / norm); put((double)s.bet[p] / norm); put(s.allin[p] ? 1.0 : 0.0); }
    const int n_out = eb_cards_out(F.rules, s.round);
    const int8_t* board = cards + 2 * F.rules.n_hole_cards;
    for (int i = 0; i < F.rules.n_board_cards; ++i) {
        const int c = i < n_out ? board[i] : -1;
        const int rank = c >= 0 ? c / F.rules.n_suits : -1, suit = c >= 0 ? c % F.rules.n_suits : -1;
        for (int j = 0; j < F.rules.n_ranks; ++j) put(j == rank ? 1.0 : 0.0);
        for (int j = 0; j < F.rules.n_suits; ++j) put(F.suits_matter && j == suit ? 1.0 : 0.0);
    }
}

// ---- the observation vectors of a workgroup's envs, written as ONE linear stream (round 4) -------------------------------------------------------
// A lane that writes its own env's vector stores 4 bytes at a stride of obs_dim floats: every store instruction of a wave touches 64 cache lines and
// the 436-byte vector of a hold'em env costs 109 of them. Every element of the vector is a function of ONE small word of the env, though: a float
// copied (seven pot / bet quotients, two stacks, two bets) or 1.0 where an integer (last action, who acted, whose turn, round, all-in flag, a
// board card's rank / suit) equals the element's own value. So each lane leaves its env's <= 27 SOURCE WORDS in an LDS row, and the workgroup then
// writes the vectors of its 256 consecutive envs -- one contiguous piece of memory -- with linear, fully coalesced stores: element m of the piece
// belongs to env m / obs_dim, entry m % obs_dim, whose table entry says which word and which comparison.
#define EB_OBS_ROW 29  // words per LDS row (odd: the lanes' row writes hit 64 different banks); [27] = 1: leave this env's vector alone
#define EB_OBS_SKIP 27
PRL_HD PRL_INLINE size_t eb_obs_tab_bytes(int obs_dim) { return (((size_t)obs_dim * 4 + 15) & ~(size_t)15) + 16; }  // the entry table + eb_obs_const(0..2)
PRL_HD PRL_INLINE size_t eb_obs_smem(int obs_dim, int n_threads) { return eb_obs_tab_bytes(obs_dim) + (size_t)n_threads * EB_OBS_ROW * 4; }
// table entry of element j: source word | (value + 1) << 8, value + 1 == 0 for a float that is copied. Source words: 0..6 the seven quotients of
// eb_observation's first line, 7 last action, 8 who did it, 9 whose turn, 10 round, 11 + 3 p: stack, bet, all-in flag of seat p, 17 + 2 i: rank and
// suit of board card i (-1: not dealt yet / suits do not matter)
PRL_HD PRL_INLINE int32_t eb_obs_entry(const EbFull& F, int j) {
    if (j < 7) return j;
    j -= 7;
    if (j < 3) return 7 | ((j + 1) << 8);
    j -= 3;
    if (j < 2) return 8 | ((j + 1) << 8);
    j -= 2;
    if (j < 2) return 9 | ((j + 1) << 8);
    j -= 2;
    if (j < F.rules.n_rounds) return 10 | ((j + 1) << 8);
    j -= F.rules.n_rounds;
    if (j < 6) { const int p = j / 3, q = j % 3; return q < 2 ? 11 + 3 * p + q : ((13 + 3 * p) | (2 << 8)); }
    j -= 6;
    const int per = F.rules.n_ranks + F.rules.n_suits, i = j / per, q = j % per;
    return q < F.rules.n_ranks ? ((17 + 2 * i) | ((q + 1) << 8)) : ((18 + 2 * i) | ((q - F.rules.n_ranks + 1) << 8));
}
PRL_HD PRL_INLINE uint32_t eb_f32_bits(double v) { const float f = (float)v; uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }
// the three quotients every vector starts with (ante, blinds): the game's, not the env's -- three lanes compute them once per workgroup
// (eb_obs_setup), every lane copies them (round 6; they were three float64 divisions per env and step)
PRL_HD PRL_INLINE uint32_t eb_obs_const(const PrlGame& g, int k) {
    const double norm = (double)(g.start_stack[0] + g.start_stack[1]) / 2.0;
    return eb_f32_bits((double)(k == 0 ? g.ante : (k == 1 ? g.small_blind : g.big_blind)) / norm);
}
// the source words of one env (live = false: a finished episode, the reference's all-zero observation); k3: eb_obs_const(0..2)
PRL_HD PRL_INLINE void eb_obs_words(const PrlGame& g, const EbFull& F, const PrlEnvState& s, const int8_t (&c)[16], bool live, const uint32_t* k3, uint32_t* w) {
    w[EB_OBS_SKIP] = 0u;
    if (!live) {
        for (int k = 0; k < 7; ++k) w[k] = 0u;
        for (int k = 7; k < 27; ++k) w[k] = 0xFFFFFFFFu;
        for (int p = 0; p < 2; ++p) { w[11 + 3 * p] = 0u; w[12 + 3 * p] = 0u; }
        return;
    }
    const double norm = (double)(g.start_stack[0] + g.start_stack[1]) / 2.0;
    const int small = s.bet[0] < s.bet[1] ? s.bet[0] : s.bet[1], big = s.bet[0] < s.bet[1] ? s.bet[1] : s.bet[0];
    const int min_raise = big + ((big - small) > g.big_blind ? (big - small) : g.big_blind);
    const bool have_la = s.last_action[0] >= 0;
    const uint32_t q_bet0 = eb_f32_bits((double)s.bet[0] / norm), q_bet1 = eb_f32_bits((double)s.bet[1] / norm);
    w[0] = k3[0]; w[1] = k3[1]; w[2] = k3[2];
    w[3] = eb_f32_bits((double)min_raise / norm); w[4] = eb_f32_bits((double)s.main_pot / norm);
    w[5] = s.bet[0] < s.bet[1] ? q_bet1 : q_bet0;  // big / norm: the larger bet's quotient
    w[6] = have_la ? eb_f32_bits((double)s.last_action[1] / norm) : 0u;
    w[7] = have_la ? (uint32_t)s.last_action[0] : 0xFFFFFFFFu;
    w[8] = have_la ? (uint32_t)s.last_action[2] : 0xFFFFFFFFu;
    w[9] = (uint32_t)s.cur;
    w[10] = (uint32_t)s.round;
    w[11] = eb_f32_bits((double)s.stack[0] / norm); w[12] = q_bet0; w[13] = s.allin[0] ? 1u : 0u;
    w[14] = eb_f32_bits((double)s.stack[1] / norm); w[15] = q_bet1; w[16] = s.allin[1] ? 1u : 0u;
    const int n_out = eb_cards_out(F.rules, s.round);
    for (int i = 0; i < 5; ++i) {
        const int cb = EB_BOARD(c, F.rules.n_hole_cards, i);
        const int cc = (i < F.rules.n_board_cards && i < n_out) ? cb : -1;
        w[17 + 2 * i] = cc >= 0 ? (uint32_t)(cc / F.rules.n_suits) : 0xFFFFFFFFu;
        w[18 + 2 * i] = (cc >= 0 && F.suits_matter) ? (uint32_t)(cc % F.rules.n_suits) : 0xFFFFFFFFu;
    }
}
// LDS: the entry table, the game's three constant quotients, then one row per lane (valid after the next workgroup barrier)
PRL_DEV PRL_INLINE void eb_obs_setup(const PrlGame& g, const EbFull& F, int32_t** tab, const uint32_t** k3, uint32_t** rows) {
    char* sm = prl_smem();
    *tab = (int32_t*)sm;
    uint32_t* kc = (uint32_t*)(sm + eb_obs_tab_bytes(F.obs_dim) - 16);
    *k3 = kc;
    *rows = (uint32_t*)(sm + eb_obs_tab_bytes(F.obs_dim));
    for (int j = (int)prl_tid(); j < F.obs_dim; j += (int)prl_nthreads()) (*tab)[j] = eb_obs_entry(F, j);
    if (prl_tid() < 3) kc[prl_tid()] = eb_obs_const(g, (int)prl_tid());
}
// after a workgroup barrier: the vectors of the n_rows envs whose rows the lanes filled, to out (= the first of these envs' vectors).
// The table-driven loop: element m of the piece by lane m % T -- any obs_dim, ~19 vector instructions per element (entry look-up, row look-up,
// the running (env, entry) pair). Kept for vectors longer than the workgroup; the games of this package take eb_obs_emit_cols.
template <bool SKIPPABLE>
PRL_DEV PRL_INLINE void eb_obs_emit_any(const int32_t* tab, const uint32_t* rows, int obs_dim, int n_rows, float* out) {
    const int T = (int)prl_nthreads(), total = n_rows * obs_dim, de = T / obs_dim, dj = T % obs_dim;
    int e = (int)prl_tid() / obs_dim, j = (int)prl_tid() % obs_dim;
    for (int m = (int)prl_tid(); m < total; m += T) {
        const int t = tab[j];
        const uint32_t w = rows[e * EB_OBS_ROW + (t & 255)];
        const int c = t >> 8;
        float one = 1.f, v;
        __builtin_memcpy(&v, &w, 4);
        if (c != 0) v = (int)w == c - 1 ? one : 0.f;
        if (!SKIPPABLE || rows[e * EB_OBS_ROW + EB_OBS_SKIP] == 0u) out[m] = v;
        e += de; j += dj;
        if (j >= obs_dim) { j -= obs_dim; ++e; }
    }
}
// Round 6: a lane keeps ONE entry of the vector for the whole kernel. The emit loop above was 38 % of the step kernel's vector instructions
// (19 per element x 109 elements per env; SQ_INSTS_VALU 5.4 k per wave, profiles/r07_env_pmc.txt) although an element is one LDS word, one
// comparison and one store. With T / obs_dim envs side by side (2 for hold'em's 109 entries in a 256-lane workgroup) lane t owns entry
// t % obs_dim of env t / obs_dim, + k envs every round: its table entry lives in registers as (source word, value to match, result on a match,
// mask of the word otherwise) -- a copied float is "match 0 -> 0, else the word itself" -- and the row and output addresses advance by constants.
// The stores of a round are still one linear piece of k vectors.
// the lane id, opaque to the optimiser: what is derived from it is computed where it is asked for (after the step), not hoisted out of the
// kernel's grid-stride loop to live in registers across the betting code (spills at the 128-register budget)
PRL_DEV PRL_INLINE int eb_tid_here() {
    int t = (int)prl_tid();
#if !defined(PRL_EMU)
    asm volatile("" : "+v"(t));
#endif
    return t;
}
struct EbObsLane {
    int32_t src, e0, k, j;
    uint32_t match, hit, keep;
    bool on;
};
PRL_DEV PRL_INLINE EbObsLane eb_obs_lane(const EbFull& F) {
    EbObsLane L;
    const int T = (int)prl_nthreads(), D = F.obs_dim, t = eb_tid_here();
    L.k = T / D;  // 0: vectors longer than the workgroup -> eb_obs_emit_any
#if defined(PRL_EB_EMIT_ANY)  // A/B builds only (python -m pokerrl_amd.build --variant ...): round 4's loop for every game
    L.k = 0;
#endif
    L.e0 = L.k ? t / D : 0;
    L.j = L.k ? t % D : 0;
    L.on = L.k != 0 && L.e0 < L.k;
    const int32_t ent = eb_obs_entry(F, L.j);
    const int c = ent >> 8;
    L.src = ent & 255;
    L.match = c ? (uint32_t)(c - 1) : 0u;
    L.hit = c ? 0x3F800000u : 0u;
    L.keep = c ? 0u : 0xFFFFFFFFu;
    return L;
}
template <bool SKIPPABLE>
PRL_DEV PRL_INLINE void eb_obs_emit_cols(const EbObsLane& L, const uint32_t* rows, int obs_dim, int n_rows, float* out) {
    if (!L.on) return;
    const uint32_t* r = rows + (size_t)L.e0 * EB_OBS_ROW;
    float* o = out + (size_t)L.e0 * obs_dim + L.j;
    const int dr = L.k * EB_OBS_ROW, dout = L.k * obs_dim;
    auto put = [&](uint32_t w, uint32_t skip, float* dst) {
        const uint32_t u = w == L.match ? L.hit : (w & L.keep);
        float v;
        __builtin_memcpy(&v, &u, 4);
        if (!SKIPPABLE || skip == 0u) *dst = v;
    };
    int e = L.e0;
    for (; e + 3 * L.k < n_rows; e += 4 * L.k) {  // four rounds' LDS words in flight, then four stores
        uint32_t w[4], sk[4] = {0u, 0u, 0u, 0u};
        for (int q = 0; q < 4; ++q) { w[q] = r[q * dr + L.src]; if (SKIPPABLE) sk[q] = r[q * dr + EB_OBS_SKIP]; }
        for (int q = 0; q < 4; ++q) put(w[q], sk[q], o + (size_t)q * dout);
        r += 4 * dr; o += (size_t)4 * dout;
    }
    for (; e < n_rows; e += L.k) {
        put(r[L.src], SKIPPABLE ? r[EB_OBS_SKIP] : 0u, o);
        r += dr; o += dout;
    }
}
// ... and four entries per store: the vectors of four consecutive envs are obs_dim 16-byte pieces; a lane owns piece t % obs_dim of the group
// t / obs_dim (+ k groups every round) = four (env of the group, entry) pairs with their table entries in registers: four LDS words, ONE
// global_store_dwordx4. 32 stores per lane and chunk instead of 128 -- fewer than the 63 vector-memory operations a wave may have in flight, so
// the next chunk's loads, issued ahead of them (prl_k_ebf_random_step), are not held up behind the stores. Needs a 16-byte aligned `out` (a
// chunk's first vector lies 256 x 4 obs_dim bytes into the buffer); rows past the last whole group go through the per-entry loop.
struct EbObsLane4 {
    int32_t roff[4];  // word offset of piece entry q's source inside the group's four rows
    uint32_t match[4], hit[4], keep[4];
    int32_t g0, k, piece;
    bool on;
};
PRL_DEV PRL_INLINE EbObsLane4 eb_obs_lane4(const EbFull& F) {
    EbObsLane4 L;
    const int T = (int)prl_nthreads(), D = F.obs_dim, t = eb_tid_here();
    L.k = T / D;
    L.g0 = L.k ? t / D : 0;
    L.piece = L.k ? t % D : 0;
    L.on = L.k != 0 && L.g0 < L.k;
    for (int q = 0; q < 4; ++q) {
        const int m = 4 * L.piece + q, e = m / D, j = m - e * D;
        const int32_t ent = eb_obs_entry(F, j);
        const int c = ent >> 8;
        L.roff[q] = e * EB_OBS_ROW + (ent & 255);
        L.match[q] = c ? (uint32_t)(c - 1) : 0u;
        L.hit[q] = c ? 0x3F800000u : 0u;
        L.keep[q] = c ? 0u : 0xFFFFFFFFu;
    }
    return L;
}
struct alignas(16) EbU4 { uint32_t x, y, z, w; };
// returns the number of rows written (a multiple of 4)
PRL_DEV PRL_INLINE int eb_obs_emit_vec4(const EbObsLane4& L, const uint32_t* rows, int obs_dim, int n_rows, float* out) {
    const int n_groups = n_rows >> 2;
    if (L.on) {
        const uint32_t* r = rows + (size_t)L.g0 * 4 * EB_OBS_ROW;
        EbU4* o = (EbU4*)out + (size_t)L.g0 * obs_dim + L.piece;
        const int dr = L.k * 4 * EB_OBS_ROW, dout = L.k * obs_dim;
        auto val = [&](uint32_t w, int q) { return w == L.match[q] ? L.hit[q] : (w & L.keep[q]); };
        int g = L.g0;
        for (; g + L.k < n_groups; g += 2 * L.k) {  // two rounds' LDS words in flight
            uint32_t a[4], b[4];
            for (int q = 0; q < 4; ++q) { a[q] = r[L.roff[q]]; b[q] = r[dr + L.roff[q]]; }
            EbU4 va, vb;
            va.x = val(a[0], 0); va.y = val(a[1], 1); va.z = val(a[2], 2); va.w = val(a[3], 3);
            vb.x = val(b[0], 0); vb.y = val(b[1], 1); vb.z = val(b[2], 2); vb.w = val(b[3], 3);
            o[0] = va; o[dout] = vb;
            r += 2 * dr; o += (size_t)2 * dout;
        }
        for (; g < n_groups; g += L.k) {
            EbU4 va;
            va.x = val(r[L.roff[0]], 0); va.y = val(r[L.roff[1]], 1); va.z = val(r[L.roff[2]], 2); va.w = val(r[L.roff[3]], 3);
            o[0] = va;
            r += dr; o += dout;
        }
    }
    return n_groups << 2;
}
// The same for the common case -- a whole chunk of 256 envs, 256 lanes, two groups of four envs per round (2 obs_dim <= 256 < 3 obs_dim: hold'em's
// 109 entries) -- as STRAIGHT-LINE code: 16 x 2 stores every wave issues whatever its lanes are (the lanes past 2 obs_dim repeat the last piece:
// same address, same value). That the stores are unconditional is the point: `vmcnt` counts loads and stores in order, and only when the compiler
// can count 32 stores behind the next chunk's loads does its wait at the top of the next step let them drain underneath it
// (prl_k_ebf_random_step<true>). LDS offsets and the rounds are compile-time constants.
PRL_DEV PRL_INLINE void eb_obs_emit_whole_chunk_k2(const EbFull& F, const uint32_t* rows, float* out) {
    const int D = F.obs_dim;
    int t = eb_tid_here();
    t = t < 2 * D ? t : 2 * D - 1;
    const int g0 = t >= D ? 1 : 0, piece = t - g0 * D;
    int32_t roff[4];
    uint32_t match[4], hit[4], keep[4];
    for (int q = 0; q < 4; ++q) {
        const int m = 4 * piece + q, e = m / D, j = m - e * D;
        const int32_t ent = eb_obs_entry(F, j);
        const int c = ent >> 8;
        roff[q] = (g0 * 4 + e) * EB_OBS_ROW + (ent & 255);
        match[q] = c ? (uint32_t)(c - 1) : 0u;
        hit[q] = c ? 0x3F800000u : 0u;
        keep[q] = c ? 0u : 0xFFFFFFFFu;
    }
    EbU4* o = (EbU4*)out + (size_t)g0 * D + piece;
    const int dout = 2 * D;
    auto val = [&](uint32_t w, int q) { return w == match[q] ? hit[q] : (w & keep[q]); };
    constexpr int DR = 8 * EB_OBS_ROW;  // two groups of four rows per round
#if defined(__clang__)
#pragma unroll
#endif
    for (int it = 0; it < 16; ++it) {
        uint32_t a[4], b[4];
        for (int q = 0; q < 4; ++q) { a[q] = rows[(2 * it) * DR + roff[q]]; b[q] = rows[(2 * it + 1) * DR + roff[q]]; }
        EbU4 va, vb;
        va.x = val(a[0], 0); va.y = val(a[1], 1); va.z = val(a[2], 2); va.w = val(a[3], 3);
        vb.x = val(b[0], 0); vb.y = val(b[1], 1); vb.z = val(b[2], 2); vb.w = val(b[3], 3);
        o[(size_t)(2 * it) * dout] = va;
        o[(size_t)(2 * it + 1) * dout] = vb;
    }
}
template <bool SKIPPABLE>
PRL_DEV PRL_INLINE void eb_obs_emit(const EbFull& F, const int32_t* tab, const uint32_t* rows, int obs_dim, int n_rows, float* out) {
    int first = 0;
#if !defined(PRL_EB_NO_VEC4)
    if (!SKIPPABLE && (int)prl_nthreads() >= obs_dim && (((uintptr_t)out) & 15u) == 0) {
        const EbObsLane4 L4 = eb_obs_lane4(F);
        first = eb_obs_emit_vec4(L4, rows, obs_dim, n_rows, out);
        if (first == n_rows) return;
        rows += (size_t)first * EB_OBS_ROW; out += (size_t)first * obs_dim; n_rows -= first;
    }
#endif
    const EbObsLane L = eb_obs_lane(F);  // here, not ahead of the step: its registers would be live across the betting code (spills at 128 VGPRs)
    if (L.k) eb_obs_emit_cols<SKIPPABLE>(L, rows, obs_dim, n_rows, out);
    else eb_obs_emit_any<SKIPPABLE>(tab, rows, obs_dim, n_rows, out);
}

#if defined(PRL_EB_TIMELINE) && !defined(PRL_EMU)  // instrumented variant builds only: clocks of every workgroup's phases (scripts/r6_env_timeline.py)
__device__ unsigned long long eb_timeline[4096 * 4];
extern "C" int32_t prl_debug_eb_timeline(unsigned long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(eb_timeline), sizeof(eb_timeline)) == hipSuccess ? 0 : -1;
}
#define EB_TL(slot) do { if (prl_tid() == 0 && prl_bid() < 4096u) eb_timeline[prl_bid() * 4 + (slot)] = (slot) == 3 ? (unsigned long long)__builtin_amdgcn_s_getreg(0xF804 /* HW_ID[15:0] */) : (unsigned long long)__builtin_readcyclecounter(); } while (0)
#else
#define EB_TL(slot) do { } while (0)
#endif
#ifndef PRL_EB_KO
#define PRL_EB_KO 0  // knock-out builds of prl_k_ebf_random_step for timing experiments: 1 no observation stores, 2 observation stores only
#endif
// reset of the masked envs: public state, a fresh hand from the env's counter-based deck, the observation of the new hand
PRL_GLOBAL void prl_k_ebf_reset(const PrlGame* __restrict__ g, EbFull F, int32_t* __restrict__ st, int n, const uint8_t* __restrict__ mask, int8_t* __restrict__ cards, uint32_t* __restrict__ episode, uint64_t deck_seed,
                                int deal, float* __restrict__ obs) {
    int32_t* tab;
    const uint32_t* k3;
    uint32_t* rows;
    eb_obs_setup(*g, F, &tab, &k3, &rows);
    const int T = (int)prl_nthreads();
    for (int i0 = (int)prl_bid() * T; i0 < n; i0 += (int)prl_nblocks() * T) {  // workgroup-uniform: the vectors of T consecutive envs leave together
        const int i = i0 + (int)prl_tid();
        uint32_t* w = rows + (size_t)prl_tid() * EB_OBS_ROW;
        prl_sync();
        if (i < n) {
            if (mask && !mask[i]) w[EB_OBS_SKIP] = 1u;
            else {
                PrlEnvState s;
                prl_env_reset(*g, s);
                eb_store(st, n, i, s, false);
                int8_t* c = cards + (size_t)i * F.n_deal;
                int8_t cl[16];
                if (deal) {
                    const uint32_t ep = episode[i];
                    episode[i] = ep + 1u;
                    prl_deal_hand(F.rules.n_cards, F.n_deal, deck_seed, (unsigned long long)ep * (unsigned long long)n + (unsigned long long)i, cl);
                    eb_cards_store(c, F.n_deal, cl);
                } else eb_cards_load(c, F.n_deal, cl);
                if (obs) eb_obs_words(*g, F, s, cl, true, k3, w);
            }
        }
        prl_sync();
        if (obs) eb_obs_emit<true>(F, tab, rows, F.obs_dim, n - i0 < T ? n - i0 : T, obs + (size_t)i0 * F.obs_dim);
    }
}

// PokerEnv.step of every env: (obs, reward, done, info). An env whose episode is over, or whose action is < 0, is skipped
// (done = 1, zero observation, zero reward, info -1 as prl_k_eb_step).
PRL_GLOBAL void prl_k_ebf_step(const PrlGame* __restrict__ g, EbFull F, int32_t* __restrict__ st, int n, const int32_t* __restrict__ a0, const int32_t* __restrict__ a1, int processed, const int8_t* __restrict__ cards,
                               float* __restrict__ obs, double* __restrict__ rew, uint8_t* __restrict__ done_out, int32_t* __restrict__ info) {
    int32_t* tab;
    const uint32_t* k3;
    uint32_t* rows;
    eb_obs_setup(*g, F, &tab, &k3, &rows);
    const int T = (int)prl_nthreads(), stride = (int)prl_nblocks() * T;
    // the raw words of a chunk -- state, cards, actions -- one chunk ahead, issued before the observation stores of the chunk in hand
    // (as prl_k_ebf_random_step, where the scheme is described)
    int32_t raw[PRL_EB_N_COLS], cw[16], act_in = -1, amt_in = -1;
    auto fetch = [&](int i) {
        eb_load_raw(st, n, i, raw);
        act_in = a0[i];
        if (processed) amt_in = a1[i];
        const int8_t* p = cards + (size_t)i * F.n_deal;
#if defined(__clang__)
#pragma unroll
#endif
        for (int d = 0; d < 16; ++d) cw[d] = p[d < F.n_deal ? d : 0];
    };
    if ((int)prl_bid() * T + (int)prl_tid() < n) fetch((int)prl_bid() * T + (int)prl_tid());
    for (int i0 = (int)prl_bid() * T; i0 < n; i0 += stride) {
        const int i = i0 + (int)prl_tid();
        prl_sync();
        if (i < n) {
            PrlEnvState s;
            bool done;
            int8_t cl[16];
            for (int d = 0; d < 16; ++d) cl[d] = d < F.n_deal ? (int8_t)cw[d] : (int8_t)-1;
            eb_unpack(raw, s, &done);
            const int act = act_in;
            int o0 = -1, o1 = 0, o2 = 0, o3 = 0;
            double r[2] = {0.0, 0.0};
            const int n_act = g->game_type == PRL_GAME_DISCRETIZED ? g->n_bet_sizes + 2 : 3;
            const bool stepped = !done && act >= 0 && act < (processed ? 3 : n_act);
            if (stepped) {
                PrlStepInfo si;
                if (processed) prl_env_step_processed(*g, s, act, amt_in, &si);
                else prl_env_step(*g, s, act, &si);
                done = si.is_terminal != 0;
                eb_store(st, n, i, s, done);
                o0 = si.is_terminal; o1 = si.chance_acts; o2 = si.pot_before_payout;
                o3 = si.is_terminal ? (si.terminal_is_fold ? 1 : (si.rundown ? 3 : 2)) : 0;
                if (done) {
                    bool sd;
                    eb_payout(*g, F, s, cl, r, &sd);
                }
            }
            // PokerEnv.get_current_obs(is_terminal=True): zeros
            eb_obs_words(*g, F, s, cl, stepped && !done, k3, rows + (size_t)prl_tid() * EB_OBS_ROW);
            rew[2 * (size_t)i] = r[0];
            rew[2 * (size_t)i + 1] = r[1];
            done_out[i] = done ? 1 : 0;
            if (info) { info[i] = o0; info[(size_t)n + i] = o1; info[(size_t)2 * n + i] = o2; info[(size_t)3 * n + i] = o3; }
        }
        prl_sync();
        if ((long long)i + stride < (long long)n) fetch(i + stride);
        eb_obs_emit<false>(F, tab, rows, F.obs_dim, n - i0 < T ? n - i0 : T, obs + (size_t)i0 * F.obs_dim);
    }
}

// the observation of every env's CURRENT state (PokerEnv.get_current_obs; zeros for a finished episode)
PRL_GLOBAL void prl_k_ebf_observe(const PrlGame* __restrict__ g, EbFull F, const int32_t* __restrict__ st, int n, const int8_t* __restrict__ cards, float* __restrict__ obs) {
    int32_t* tab;
    const uint32_t* k3;
    uint32_t* rows;
    eb_obs_setup(*g, F, &tab, &k3, &rows);
    const int T = (int)prl_nthreads();
    for (int i0 = (int)prl_bid() * T; i0 < n; i0 += (int)prl_nblocks() * T) {
        const int i = i0 + (int)prl_tid();
        prl_sync();
        if (i < n) {
            PrlEnvState s;
            bool done;
            int8_t cl[16];
            eb_cards_load(cards + (size_t)i * F.n_deal, F.n_deal, cl);
            eb_load(st, n, i, s, &done);
            eb_obs_words(*g, F, s, cl, !done, k3, rows + (size_t)prl_tid() * EB_OBS_ROW);
        }
        prl_sync();
        eb_obs_emit<false>(F, tab, rows, F.obs_dim, n - i0 < T ? n - i0 : T, obs + (size_t)i0 * F.obs_dim);
    }
}

// uniform-random legal play of WHOLE hands with the state in registers: deal, bet, show down, pay out, deal again. stats: steps, finished
// hands, showdowns, sum over hands of 2 x seat 0's chip winnings + a large offset (an integer checksum of the payouts)
PRL_HD PRL_INLINE void eb_play_full(const PrlGame& g, const EbFull& F, int n, int i, int n_steps, uint32_t seed, uint64_t deck_seed, unsigned long long acc[4]) {
    PrlEnvState s;
    prl_env_reset(g, s);
    int8_t cards[16];
    for (int d = 0; d < 16; ++d) cards[d] = -1;
    uint32_t ep = 0;
    prl_deal_hand(F.rules.n_cards, F.n_deal, deck_seed, (unsigned long long)ep * (unsigned long long)n + (unsigned long long)i, cards);
    for (int k = 0; k < n_steps; ++k) {
        const uint32_t r = eb_mix32(seed ^ eb_mix32((uint32_t)i * 0x9E3779B9u + (uint32_t)k));
        PrlStepInfo si;
        const int a = prl_legal_action_pick(g, s, r);
        if (g.game_type == PRL_GAME_NOLIMIT) prl_env_step_processed(g, s, a, a == PRL_BET_RAISE ? (int)(eb_mix32(r) % (uint32_t)(prl_at2(s.stack, s.cur) + prl_at2(s.bet, s.cur) + 1)) : -1, &si);
        else prl_env_step(g, s, a, &si);
        acc[0] += 1;
        if (si.is_terminal) {
            double rew[2];
            bool sd;
            eb_payout(g, F, s, cards, rew, &sd);
            acc[1] += 1;
            acc[2] += sd ? 1 : 0;
            acc[3] += (unsigned long long)(long long)(2.0 * rew[0] * F.reward_scalar) + (1ull << 20);
            prl_env_reset(g, s);
            ++ep;
            prl_deal_hand(F.rules.n_cards, F.n_deal, deck_seed, (unsigned long long)ep * (unsigned long long)n + (unsigned long long)i, cards);
        }
    }
}
PRL_GLOBAL void prl_k_ebf_rollout(const PrlGame* __restrict__ g, EbFull F, int n, int n_steps, uint32_t seed, uint64_t deck_seed, unsigned long long* __restrict__ stats) {
    unsigned long long acc[4] = {0, 0, 0, 0};
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n; i += (int)(prl_nblocks() * prl_nthreads())) eb_play_full(*g, F, n, i, n_steps, seed, deck_seed, acc);
    // per wave: butterfly sums in 32-bit halves, one atomic per counter
    for (int c = 0; c < 4; ++c) {
        unsigned long long v = acc[c];
        for (int d = 32; d > 0; d >>= 1) {
            const int src = (int)prl_lane() ^ d;
            const unsigned lo = (unsigned)prl_shfl_i((int)(unsigned)(v & 0xFFFFFFFFull), src);
            const unsigned hi = (unsigned)prl_shfl_i((int)(unsigned)(v >> 32), src);
            v += ((unsigned long long)hi << 32) | lo;
        }
        if (prl_lane() == 0) prl_atomic_add_u64(stats + 4 * (((prl_bid() * prl_nthreads() + prl_tid()) >> 6) & (EB_STAT_SLOTS - 1)) + c, v);
    }
}

// one whole PokerEnv.step per env and launch with the state in HBM between the launches, driven by uniform-random legal actions: what an
// agent-driven rollout costs per step -- 13 state words in and out, the observation vector, two rewards and the done flag out; a finished
// hand is reset and dealt again at the next launch. Step k of env i draws the same number as step k of the other rollouts.
// WHOLE: every chunk is 256 envs (n % 256 == 0), 256 lanes, two groups of four observation vectors per store round, `obs` 16-byte aligned
// (the host checks): the first chunk peeled and the straight-line emit, so that both ways into the loop carry "loads, then 32 stores"
template <bool WHOLE>
PRL_GLOBAL void prl_k_ebf_random_step(const PrlGame* __restrict__ g, EbFull F, int32_t* __restrict__ st, int n, int k, uint32_t seed, int8_t* __restrict__ cards,
                                      uint32_t* __restrict__ episode, uint64_t deck_seed, float* __restrict__ obs, double* __restrict__ rew, uint8_t* __restrict__ done_out,
                                      unsigned long long* __restrict__ stats) {
    unsigned long long hands = 0, pots = 0, steps = 0;
    int32_t* tab;
    const uint32_t* k3;
    uint32_t* rows;
    eb_obs_setup(*g, F, &tab, &k3, &rows);
    const int T = (int)prl_nthreads(), stride = (int)prl_nblocks() * T;
    // A chunk's raw words -- state, deck counter, cards -- are loaded ONE CHUNK AHEAD: issued after the step of the chunk before, ahead of that
    // chunk's observation stores, which then drain while this chunk is stepped (a persistent workgroup walks n / (256 x grid) chunks).
    int32_t raw[PRL_EB_N_COLS];
    int32_t cw[16];  // a card per register while in flight (bytes carried around the loop get packed, and every load then waits for the one before)
    uint32_t ep = 0u;
    auto fetch = [&](int i) {
        eb_load_raw(st, n, i, raw);
        ep = episode[i];
        const int8_t* p = cards + (size_t)i * F.n_deal;
#if defined(__clang__)
#pragma unroll
#endif
        for (int d = 0; d < 16; ++d) cw[d] = p[d < F.n_deal ? d : 0];
    };
    if ((int)prl_bid() * T + (int)prl_tid() < n) fetch((int)prl_bid() * T + (int)prl_tid());
    auto chunk = [&](int i0) __attribute__((always_inline)) {
        const int i = i0 + (int)prl_tid();
        prl_sync();
        EB_TL(0); EB_TL(3);
#if PRL_EB_KO == 2
        for (int q = 0; q < EB_OBS_ROW; ++q) rows[(size_t)prl_tid() * EB_OBS_ROW + q] = (uint32_t)(prl_tid() + q) & 3u;
        if (false) {
#else
        if (WHOLE || i < n) {
#endif
            PrlEnvState s;
            bool done;
            int8_t* c = cards + (size_t)i * F.n_deal;
            int8_t cl[16];
            for (int d = 0; d < 16; ++d) cl[d] = d < F.n_deal ? (int8_t)cw[d] : (int8_t)-1;
            eb_unpack(raw, s, &done);
            if (done) {
                prl_env_reset(*g, s);
                episode[i] = ep + 1u;
                prl_deal_hand(F.rules.n_cards, F.n_deal, deck_seed, (unsigned long long)ep * (unsigned long long)n + (unsigned long long)i, cl);
                eb_cards_store(c, F.n_deal, cl);
            }
            const uint32_t r = eb_mix32(seed ^ eb_mix32((uint32_t)i * 0x9E3779B9u + (uint32_t)k));
            PrlStepInfo si;
            const int a = prl_legal_action_pick(*g, s, r);
            if (g->game_type == PRL_GAME_NOLIMIT) prl_env_step_processed(*g, s, a, a == PRL_BET_RAISE ? (int)(eb_mix32(r) % (uint32_t)(prl_at2(s.stack, s.cur) + prl_at2(s.bet, s.cur) + 1)) : -1, &si);
            else prl_env_step(*g, s, a, &si);
            ++steps;
            double rw[2] = {0.0, 0.0};
            if (si.is_terminal) {
                bool sd;
                eb_payout(*g, F, s, cl, rw, &sd);
                ++hands;
                pots += (unsigned long long)si.pot_before_payout;
            }
            eb_obs_words(*g, F, s, cl, !si.is_terminal, k3, rows + (size_t)prl_tid() * EB_OBS_ROW);
            rew[2 * (size_t)i] = rw[0];
            rew[2 * (size_t)i + 1] = rw[1];
            done_out[i] = si.is_terminal ? 1 : 0;
            eb_store(st, n, i, s, si.is_terminal != 0);
        }
        prl_sync();
        EB_TL(1);
        if ((long long)i + stride < (long long)n) fetch(i + stride);
#if PRL_EB_KO == 1
        if (rows[prl_tid()] == 0x12345678u) obs[i0] = 1.f;
#else
        if constexpr (WHOLE) eb_obs_emit_whole_chunk_k2(F, rows, obs + (size_t)i0 * F.obs_dim);
        else eb_obs_emit<false>(F, tab, rows, F.obs_dim, n - i0 < T ? n - i0 : T, obs + (size_t)i0 * F.obs_dim);
#endif
#if defined(PRL_EB_TIMELINE) && !defined(PRL_EMU)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the clock after the stores have left
        EB_TL(2);
#endif
    };
    int i0 = (int)prl_bid() * T;
    if constexpr (WHOLE) {  // (the grid never exceeds the number of chunks: every workgroup has a first one)
        chunk(i0);
        i0 += stride;
    }
    for (; i0 < n; i0 += stride) chunk(i0);
    eb_stats_add(stats, steps, hands, pots);
}

static int eb_grid(int n) {
    int g = (n + 255) / 256;
    return g < 1 ? 1 : (g > 4096 ? 4096 : g);
}
// The kernels with observation rows in LDS as PERSISTENT workgroups (round 6): as many as the device holds at once (4 per CU: 128 registers,
// 30 KB of LDS), each walking its chunks of 256 envs -- a slot is not idle between a workgroup's last store and its successor's first load
// (~3 us of a ~45 us workgroup life, profiles/r103_env_timeline.txt; 2.4 % of the step kernel). PRL_EB_GRID_CAP overrides (experiments).
static int eb_grid_full(prl_envbatch* b) {
    int& cap = b->full_cap;
    if (!cap) {
        const char* e = getenv("PRL_EB_GRID_CAP");
        if (e && atoi(e) > 0) cap = atoi(e);
        else {
            int dev = 0, cus = 256, per_cu = 4;
            if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
#if !defined(PRL_EMU)
            int nb = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, prl_k_ebf_random_step<false>, 256, eb_obs_smem(b->obs_dim, 256)) == hipSuccess && nb > 0) per_cu = nb;
#endif
            cap = cus * per_cu;
        }
    }
    const int g = (b->n + 255) / 256;
    return g < 1 ? 1 : (g > cap ? cap : g);
}

extern "C" {

int32_t prl_envbatch_create(const PrlGame* game, int32_t n_envs, prl_envbatch_t** out) {
    if (!game || !out || n_envs <= 0) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (game->n_bet_sizes < 0 || game->n_bet_sizes > PRL_MAX_BET_SIZES) { prl_set_error("too many bet sizes"); return PRL_ERR_ARG; }
    if (!prl_device_available()) { prl_set_error("no HIP device"); return PRL_ERR_NO_DEVICE; }
    prl_envbatch* b = new prl_envbatch();
    b->game = *game;
    b->n = n_envs;
    auto fail = [&](const char* what) { prl_set_error(std::string("prl_envbatch_create: ") + what); prl_envbatch_destroy(b); return PRL_ERR_OOM; };
    if (hipStreamCreate(&b->stream) != hipSuccess) return fail("stream");
    if (hipMalloc((void**)&b->d_state, (size_t)PRL_EB_N_COLS * n_envs * sizeof(int32_t)) != hipSuccess) return fail("state");
    if (hipMalloc((void**)&b->d_game, sizeof(PrlGame)) != hipSuccess) return fail("game");
    if (hipMalloc((void**)&b->d_a, (size_t)n_envs * sizeof(int32_t)) != hipSuccess) return fail("staging");
    if (hipMalloc((void**)&b->d_b, (size_t)n_envs * sizeof(int32_t)) != hipSuccess) return fail("staging");
    if (hipMalloc((void**)&b->d_info, (size_t)4 * n_envs * sizeof(int32_t)) != hipSuccess) return fail("staging");
    if (hipMalloc((void**)&b->d_mask, (size_t)4 * n_envs * sizeof(uint32_t)) != hipSuccess) return fail("staging");
    if (hipMalloc((void**)&b->d_count, (size_t)(n_envs + 64) * sizeof(int32_t)) != hipSuccess) return fail("staging");
    if (hipMalloc((void**)&b->d_stats, (size_t)4 * EB_STAT_SLOTS * sizeof(unsigned long long)) != hipSuccess) return fail("stats");
    if (hipMemcpy(b->d_game, game, sizeof(PrlGame), hipMemcpyHostToDevice) != hipSuccess) return fail("upload");
    *out = b;
    return prl_envbatch_reset(b, nullptr);
}

static EbFull eb_full(const prl_envbatch* b) {
    EbFull F;
    F.rules = b->rules; F.n_deal = b->n_deal; F.obs_dim = b->obs_dim; F.suits_matter = b->rules.rank_rule == 2 ? 1 : 0;  // game_rules.py: SUITS_MATTER
    F.reward_scalar = b->reward_scalar;
    return F;
}

int32_t prl_envbatch_create_with_cards(const PrlGame* game, const PrlRules* rules, int32_t n_envs, uint64_t deck_seed, double reward_scalar,
                                       prl_envbatch_t** out) {
    if (!rules || !(reward_scalar > 0.0)) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    const int nh = rules->n_hole_cards, nb = rules->n_board_cards;
    if (nh < 1 || nh > 2 || nb < 1 || nb > 5 || rules->n_cards > PRL_LBR_MAX_CARDS || rules->n_rounds > 4 || (nh == 2 && (rules->n_cards != 52 || nb != 5))) {
        prl_set_error("batched env with cards: 1-hole-card games or 52-card hold'em");
        return PRL_ERR_UNSUPPORTED;
    }
    prl_envbatch* b = nullptr;
    int rc = prl_envbatch_create(game, n_envs, &b);
    if (rc) return rc;
    b->with_cards = true;
    b->rules = *rules;
    b->n_deal = 2 * nh + nb;
    b->obs_dim = eb_obs_dim(*rules);
    b->deck_seed = deck_seed;
    b->reward_scalar = reward_scalar;
    auto fail = [&](const char* what) { prl_set_error(std::string("prl_envbatch_create_with_cards: ") + what); prl_envbatch_destroy(b); return PRL_ERR_OOM; };
    if (hipMalloc((void**)&b->d_cards, (size_t)n_envs * b->n_deal) != hipSuccess) return fail("cards");
    if (hipMalloc((void**)&b->d_episode, (size_t)n_envs * sizeof(uint32_t)) != hipSuccess) return fail("episode counters");
    if (hipMalloc((void**)&b->d_obs, (size_t)n_envs * b->obs_dim * sizeof(float)) != hipSuccess) return fail("observations");
    if (hipMalloc((void**)&b->d_rew, (size_t)n_envs * 2 * sizeof(double)) != hipSuccess) return fail("rewards");
    if (hipMalloc((void**)&b->d_done, (size_t)n_envs) != hipSuccess) return fail("done flags");
    if (hipMemset(b->d_episode, 0, (size_t)n_envs * sizeof(uint32_t)) != hipSuccess) return fail("memset");
    *out = b;
    return prl_envbatch_reset_full(b, nullptr, nullptr);
}

int32_t prl_envbatch_obs_dim(prl_envbatch_t* b, int32_t* out) {
    if (!b || !out || !b->with_cards) { prl_set_error("not a batch with cards"); return PRL_ERR_ARG; }
    *out = b->obs_dim;
    return PRL_OK;
}

int32_t prl_envbatch_reset_full(prl_envbatch_t* b, const uint8_t* mask, float* out_obs) {
    if (!b || !b->with_cards) { prl_set_error("not a batch with cards"); return PRL_ERR_ARG; }
    uint8_t* d_m = nullptr;
    if (mask) {
        d_m = (uint8_t*)b->d_mask;
        PRL_HIP_TRY(hipMemcpyAsync(d_m, mask, (size_t)b->n, hipMemcpyHostToDevice, b->stream));
    }
    PRL_LAUNCH(prl_k_ebf_reset, eb_grid_full(b), 256, eb_obs_smem(b->obs_dim, 256), b->stream, (const PrlGame*)b->d_game, eb_full(b), b->d_state, b->n, (const uint8_t*)d_m, b->d_cards,
               b->d_episode, b->deck_seed, 1, b->d_obs);
    PRL_HIP_TRY(hipGetLastError());
    if (out_obs) PRL_HIP_TRY(hipMemcpyAsync(out_obs, b->d_obs, (size_t)b->n * b->obs_dim * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    return PRL_OK;
}

int32_t prl_envbatch_set_cards(prl_envbatch_t* b, const int8_t* cards) {
    if (!b || !b->with_cards || !cards) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipMemcpyAsync(b->d_cards, cards, (size_t)b->n * b->n_deal, hipMemcpyHostToDevice, b->stream));
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    return PRL_OK;
}

int32_t prl_envbatch_get_cards(prl_envbatch_t* b, int8_t* out_cards) {
    if (!b || !b->with_cards || !out_cards) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    PRL_HIP_TRY(hipMemcpy(out_cards, b->d_cards, (size_t)b->n * b->n_deal, hipMemcpyDeviceToHost));
    return PRL_OK;
}

int32_t prl_envbatch_observe(prl_envbatch_t* b, float* out_obs) {
    if (!b || !b->with_cards || !out_obs) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_LAUNCH(prl_k_ebf_observe, eb_grid_full(b), 256, eb_obs_smem(b->obs_dim, 256), b->stream, (const PrlGame*)b->d_game, eb_full(b), (const int32_t*)b->d_state, b->n, (const int8_t*)b->d_cards, b->d_obs);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipMemcpyAsync(out_obs, b->d_obs, (size_t)b->n * b->obs_dim * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    return PRL_OK;
}

int32_t prl_envbatch_step_full_device(prl_envbatch_t* b, const int32_t* d_actions, const int32_t* d_amounts, float* d_obs, double* d_reward2, uint8_t* d_done,
                                      int32_t* d_info4) {
    if (!b || !b->with_cards || !d_actions || !d_obs || !d_reward2 || !d_done) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_LAUNCH(prl_k_ebf_step, eb_grid_full(b), 256, eb_obs_smem(b->obs_dim, 256), b->stream, (const PrlGame*)b->d_game, eb_full(b), b->d_state, b->n, d_actions, d_amounts, d_amounts ? 1 : 0,
               (const int8_t*)b->d_cards, d_obs, d_reward2, d_done, d_info4);
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int32_t prl_envbatch_step_full(prl_envbatch_t* b, const int32_t* actions, const int32_t* amounts, float* out_obs, double* out_reward2, uint8_t* out_done,
                               int32_t* out_info4) {
    if (!b || !b->with_cards || !actions || !out_obs || !out_reward2 || !out_done) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipMemcpyAsync(b->d_a, actions, (size_t)b->n * 4, hipMemcpyHostToDevice, b->stream));
    if (amounts) PRL_HIP_TRY(hipMemcpyAsync(b->d_b, amounts, (size_t)b->n * 4, hipMemcpyHostToDevice, b->stream));
    int rc = prl_envbatch_step_full_device(b, b->d_a, amounts ? b->d_b : nullptr, b->d_obs, b->d_rew, b->d_done, b->d_info);
    if (rc) return rc;
    PRL_HIP_TRY(hipMemcpyAsync(out_obs, b->d_obs, (size_t)b->n * b->obs_dim * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipMemcpyAsync(out_reward2, b->d_rew, (size_t)b->n * 2 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipMemcpyAsync(out_done, b->d_done, (size_t)b->n, hipMemcpyDeviceToHost, b->stream));
    if (out_info4) PRL_HIP_TRY(hipMemcpyAsync(out_info4, b->d_info, (size_t)4 * b->n * 4, hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    return PRL_OK;
}

int32_t prl_envbatch_random_rollout_full(prl_envbatch_t* b, int32_t n_steps, uint32_t seed, uint64_t* out_stats4, float* out_device_ms) {
    if (!b || !b->with_cards || n_steps < 0 || !out_stats4) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    unsigned long long* d_stats = b->d_stats;
    PRL_HIP_TRY(hipMemsetAsync(d_stats, 0, 4 * EB_STAT_SLOTS * sizeof(unsigned long long), b->stream));
    hipEvent_t e0, e1;
    PRL_HIP_TRY(hipEventCreate(&e0));
    PRL_HIP_TRY(hipEventCreate(&e1));
    PRL_HIP_TRY(hipEventRecord(e0, b->stream));
    PRL_LAUNCH(prl_k_ebf_rollout, eb_grid(b->n), 256, 0, b->stream, (const PrlGame*)b->d_game, eb_full(b), b->n, n_steps, seed, b->deck_seed, d_stats);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipEventRecord(e1, b->stream));
    PRL_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    PRL_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (out_device_ms) *out_device_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    unsigned long long h[4 * EB_STAT_SLOTS];
    PRL_HIP_TRY(hipMemcpy(h, d_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 4; ++k) out_stats4[k] = 0;
    for (int j = 0; j < EB_STAT_SLOTS; ++j)
        for (int k = 0; k < 4; ++k) out_stats4[k] += (uint64_t)h[4 * j + k];
    return PRL_OK;
}

// what the LAST launch of prl_envbatch_random_steps_full (or prl_envbatch_step_full) left in the batch's own output buffers
int32_t prl_envbatch_last_outputs(prl_envbatch_t* b, float* out_obs, double* out_reward2, uint8_t* out_done) {
    if (!b || !b->with_cards) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    if (out_obs) PRL_HIP_TRY(hipMemcpyAsync(out_obs, b->d_obs, (size_t)b->n * b->obs_dim * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    if (out_reward2) PRL_HIP_TRY(hipMemcpyAsync(out_reward2, b->d_rew, (size_t)b->n * 2 * sizeof(double), hipMemcpyDeviceToHost, b->stream));
    if (out_done) PRL_HIP_TRY(hipMemcpyAsync(out_done, b->d_done, (size_t)b->n, hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    return PRL_OK;
}

int32_t prl_envbatch_random_steps_full(prl_envbatch_t* b, int32_t n_launches, uint32_t seed, uint64_t* out_stats3, float* out_device_ms) {
    if (!b || !b->with_cards || n_launches < 0 || !out_stats3) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    unsigned long long* d_stats = b->d_stats;
    PRL_HIP_TRY(hipMemsetAsync(d_stats, 0, 4 * EB_STAT_SLOTS * sizeof(unsigned long long), b->stream));
    hipEvent_t e0, e1;
    PRL_HIP_TRY(hipEventCreate(&e0));
    PRL_HIP_TRY(hipEventCreate(&e1));
    PRL_HIP_TRY(hipEventRecord(e0, b->stream));
    const bool whole = b->n % 256 == 0 && 2 * b->obs_dim <= 256 && 3 * b->obs_dim > 256 && (((uintptr_t)b->d_obs) & 15u) == 0 && !getenv("PRL_EB_NO_WHOLE");
    for (int k = 0; k < n_launches; ++k)
        if (whole)
            PRL_LAUNCH((prl_k_ebf_random_step<true>), eb_grid_full(b), 256, eb_obs_smem(b->obs_dim, 256), b->stream, (const PrlGame*)b->d_game, eb_full(b), b->d_state, b->n, k, seed, b->d_cards,
                       b->d_episode, b->deck_seed, b->d_obs, b->d_rew, b->d_done, d_stats);
        else
            PRL_LAUNCH((prl_k_ebf_random_step<false>), eb_grid_full(b), 256, eb_obs_smem(b->obs_dim, 256), b->stream, (const PrlGame*)b->d_game, eb_full(b), b->d_state, b->n, k, seed, b->d_cards,
                       b->d_episode, b->deck_seed, b->d_obs, b->d_rew, b->d_done, d_stats);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipEventRecord(e1, b->stream));
    PRL_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    PRL_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (out_device_ms) *out_device_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    unsigned long long h[3 * EB_STAT_SLOTS];
    PRL_HIP_TRY(hipMemcpy(h, d_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) out_stats3[k] = 0;
    for (int j = 0; j < EB_STAT_SLOTS; ++j)
        for (int k = 0; k < 3; ++k) out_stats3[k] += (uint64_t)h[3 * j + k];
    return PRL_OK;
}

// the same hands on the host, one env after the other (the checker of prl_k_ebf_rollout and the CPU leg of bench_env.py --full)
int32_t prl_env_random_rollout_full_host(const PrlGame* game, const PrlRules* rules, int32_t n_envs, int32_t n_steps, uint32_t seed, uint64_t deck_seed,
                                         double reward_scalar, uint64_t* out_stats4) {
    if (!game || !rules || n_envs < 0 || n_steps < 0 || !out_stats4 || !(reward_scalar > 0.0)) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    EbFull F;
    F.rules = *rules; F.n_deal = 2 * rules->n_hole_cards + rules->n_board_cards; F.obs_dim = eb_obs_dim(*rules); F.suits_matter = rules->rank_rule == 2; F.reward_scalar = reward_scalar;
    unsigned long long acc[4] = {0, 0, 0, 0};
    for (int i = 0; i < n_envs; ++i) eb_play_full(*game, F, n_envs, i, n_steps, seed, deck_seed, acc);
    for (int k = 0; k < 4; ++k) out_stats4[k] = (uint64_t)acc[k];
    return PRL_OK;
}

void prl_envbatch_destroy(prl_envbatch_t* b) {
    if (!b) return;
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    void* ptrs[] = {b->d_state, b->d_game, b->d_a, b->d_b, b->d_info, b->d_mask, b->d_count, b->d_stats, b->d_cards, b->d_episode, b->d_obs, b->d_rew, b->d_done};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int32_t prl_envbatch_reset(prl_envbatch_t* b, const uint8_t* mask) {
    if (!b) { prl_set_error("NULL batch"); return PRL_ERR_ARG; }
    uint8_t* d_m = nullptr;
    if (mask) {
        d_m = (uint8_t*)b->d_mask;  // staging: n bytes fit in the 16 n bytes of the mask buffer
        PRL_HIP_TRY(hipMemcpyAsync(d_m, mask, (size_t)b->n, hipMemcpyHostToDevice, b->stream));
    }
    PRL_LAUNCH(prl_k_eb_reset, eb_grid(b->n), 256, 0, b->stream, (const PrlGame*)b->d_game, b->d_state, b->n, (const uint8_t*)d_m);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    return PRL_OK;
}

int32_t prl_envbatch_step_device(prl_envbatch_t* b, const int32_t* d_actions, const int32_t* d_amounts, int32_t* d_info4) {
    if (!b || !d_actions) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_LAUNCH(prl_k_eb_step, eb_grid(b->n), 256, 0, b->stream, (const PrlGame*)b->d_game, b->d_state, b->n, d_actions, d_amounts,
               d_amounts ? 1 : 0, d_info4);
    PRL_HIP_TRY(hipGetLastError());
    return PRL_OK;
}

int32_t prl_envbatch_step(prl_envbatch_t* b, const int32_t* actions, const int32_t* amounts, int32_t* out_info4) {
    if (!b || !actions) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipMemcpyAsync(b->d_a, actions, (size_t)b->n * 4, hipMemcpyHostToDevice, b->stream));
    if (amounts) PRL_HIP_TRY(hipMemcpyAsync(b->d_b, amounts, (size_t)b->n * 4, hipMemcpyHostToDevice, b->stream));
    int rc = prl_envbatch_step_device(b, b->d_a, amounts ? b->d_b : nullptr, b->d_info);
    if (rc) return rc;
    if (out_info4) PRL_HIP_TRY(hipMemcpyAsync(out_info4, b->d_info, (size_t)4 * b->n * 4, hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    return PRL_OK;
}

int32_t prl_envbatch_legal_masks(prl_envbatch_t* b, uint32_t* out_mask4, int32_t* out_count) {
    if (!b || !out_mask4 || !out_count) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_LAUNCH(prl_k_eb_legal, eb_grid(b->n), 256, 0, b->stream, (const PrlGame*)b->d_game, (const int32_t*)b->d_state, b->n, b->d_mask, b->d_count);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipMemcpyAsync(out_mask4, b->d_mask, (size_t)4 * b->n * 4, hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipMemcpyAsync(out_count, b->d_count, (size_t)b->n * 4, hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    return PRL_OK;
}

int32_t prl_envbatch_active(prl_envbatch_t* b, int32_t* out_idx, int32_t* out_count) {
    if (!b || !out_idx || !out_count) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    int32_t* d_cnt = b->d_count + b->n;  // the spare words behind the per-env counts
    PRL_HIP_TRY(hipMemsetAsync(d_cnt, 0, sizeof(int32_t), b->stream));
    PRL_LAUNCH(prl_k_eb_active, eb_grid(b->n), 256, 0, b->stream, (const int32_t*)b->d_state, b->n, b->d_a, d_cnt);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipMemcpyAsync(out_count, d_cnt, sizeof(int32_t), hipMemcpyDeviceToHost, b->stream));
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    if (*out_count > 0) PRL_HIP_TRY(hipMemcpy(out_idx, b->d_a, (size_t)*out_count * 4, hipMemcpyDeviceToHost));
    return PRL_OK;
}

int32_t prl_envbatch_get_state(prl_envbatch_t* b, int32_t* out_cols) {
    if (!b || !out_cols) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    PRL_HIP_TRY(hipStreamSynchronize(b->stream));
    PRL_HIP_TRY(hipMemcpy(out_cols, b->d_state, (size_t)PRL_EB_N_COLS * b->n * 4, hipMemcpyDeviceToHost));
    return PRL_OK;
}

int32_t prl_envbatch_state_device(prl_envbatch_t* b, void** out_d_cols, void** out_hip_stream) {
    if (!b) { prl_set_error("NULL batch"); return PRL_ERR_ARG; }
    if (out_d_cols) *out_d_cols = b->d_state;
    if (out_hip_stream) *out_hip_stream = (void*)b->stream;
    return PRL_OK;
}

int32_t prl_envbatch_random_rollout(prl_envbatch_t* b, int32_t n_steps, uint32_t seed, uint64_t* out_stats3, float* out_device_ms) {
    if (!b || n_steps < 0 || !out_stats3) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    unsigned long long* d_stats = b->d_stats;
    PRL_HIP_TRY(hipMemsetAsync(d_stats, 0, 3 * EB_STAT_SLOTS * sizeof(unsigned long long), b->stream));
    hipEvent_t e0, e1;
    PRL_HIP_TRY(hipEventCreate(&e0));
    PRL_HIP_TRY(hipEventCreate(&e1));
    PRL_HIP_TRY(hipEventRecord(e0, b->stream));
    PRL_LAUNCH(prl_k_eb_rollout, eb_grid(b->n), 256, 0, b->stream, (const PrlGame*)b->d_game, b->d_state, b->n, n_steps, seed, d_stats);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipEventRecord(e1, b->stream));
    PRL_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    PRL_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (out_device_ms) *out_device_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    unsigned long long h[3 * EB_STAT_SLOTS];
    PRL_HIP_TRY(hipMemcpy(h, d_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) out_stats3[k] = 0;
    for (int j = 0; j < EB_STAT_SLOTS; ++j)
        for (int k = 0; k < 3; ++k) out_stats3[k] += (uint64_t)h[3 * j + k];
    return PRL_OK;
}

int32_t prl_envbatch_random_steps(prl_envbatch_t* b, int32_t n_launches, uint32_t seed, uint64_t* out_stats3, float* out_device_ms) {
    if (!b || n_launches < 0 || !out_stats3) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    unsigned long long* d_stats = b->d_stats;
    PRL_HIP_TRY(hipMemsetAsync(d_stats, 0, 3 * EB_STAT_SLOTS * sizeof(unsigned long long), b->stream));
    hipEvent_t e0, e1;
    PRL_HIP_TRY(hipEventCreate(&e0));
    PRL_HIP_TRY(hipEventCreate(&e1));
    PRL_HIP_TRY(hipEventRecord(e0, b->stream));
    for (int k = 0; k < n_launches; ++k)
        PRL_LAUNCH(prl_k_eb_random_step, eb_grid(b->n), 256, 0, b->stream, (const PrlGame*)b->d_game, b->d_state, b->n, k, seed, d_stats);
    PRL_HIP_TRY(hipGetLastError());
    PRL_HIP_TRY(hipEventRecord(e1, b->stream));
    PRL_HIP_TRY(hipEventSynchronize(e1));
    float ms = 0.f;
    PRL_HIP_TRY(hipEventElapsedTime(&ms, e0, e1));
    if (out_device_ms) *out_device_ms = ms;
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    unsigned long long h[3 * EB_STAT_SLOTS];
    PRL_HIP_TRY(hipMemcpy(h, d_stats, sizeof(h), hipMemcpyDeviceToHost));
    for (int k = 0; k < 3; ++k) out_stats3[k] = 0;
    for (int j = 0; j < EB_STAT_SLOTS; ++j)
        for (int k = 0; k < 3; ++k) out_stats3[k] += (uint64_t)h[3 * j + k];
    return PRL_OK;
}

// the same random play on the host, one env after the other (the CPU leg of bench_env.py; also the checker of the kernel above)
int32_t prl_env_random_rollout_host(const PrlGame* game, int32_t n_envs, int32_t n_steps, uint32_t seed, uint64_t* out_stats3) {
    if (!game || n_envs < 0 || n_steps < 0 || !out_stats3) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    uint64_t steps = 0, hands = 0, pots = 0;
    for (int i = 0; i < n_envs; ++i) {
        PrlEnvState s;
        prl_env_reset(*game, s);
        for (int k = 0; k < n_steps; ++k) {
            const uint32_t r = eb_mix32(seed ^ eb_mix32((uint32_t)i * 0x9E3779B9u + (uint32_t)k));
            PrlStepInfo si;
            const int a = prl_legal_action_pick(*game, s, r);
            if (game->game_type == PRL_GAME_NOLIMIT) prl_env_step_processed(*game, s, a, a == PRL_BET_RAISE ? (int)(eb_mix32(r) % (uint32_t)(prl_at2(s.stack, s.cur) + prl_at2(s.bet, s.cur) + 1)) : -1, &si);
            else prl_env_step(*game, s, a, &si);
            ++steps;
            if (si.is_terminal) { ++hands; pots += (uint64_t)si.pot_before_payout; prl_env_reset(*game, s); }
        }
    }
    out_stats3[0] = steps; out_stats3[1] = hands; out_stats3[2] = pots;
    return PRL_OK;
}

}  // extern "C"
