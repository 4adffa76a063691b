// Heads-up poker betting state machine (public state only), host + device.
//
// Integer-exact restatement of the heads-up subset of the reference engine
//   PokerRL/game/_/rl_env/base/PokerEnv.py:681-789 (_step), :885-941 (_get_fixed_action/_process_*), :943-954
//   (_should_continue_in_this_round), :539-551 (HU bet sweep), :803-812 (HU min raise), :1376-1396 (pot fraction),
//   :1075-1122 (reset);  _PokerPlayer.py:68-100;  LimitPokerEnv.py:27-59;  DiscretizedPokerEnv.py:44-135;
//   NoLimitPokerEnv.py:32-33;  games.py:222-254 (Flop5Holdem pot-size raise through the Limit env).
// BR / CFR / LBR in the reference assert two seats (_CFRBase.py:40, LocalBRMaster.py:23, LocalLBRWorker.py:18), so the
// N>2 side-pot machinery is deliberately absent.
//
// The struct is a POD so that the same code steps one env on the host (public-tree construction) and a batch of envs
// on the GPU (one lane per env).
#pragma once
#include "prl_defs.h"

enum { PRL_GAME_LIMIT = 0, PRL_GAME_DISCRETIZED = 1, PRL_GAME_NOLIMIT = 2 };

PRL_HD PRL_INLINE int prl_imax(int a, int b) { return a > b ? a : b; }
PRL_HD PRL_INLINE int prl_imin(int a, int b) { return a < b ? a : b; }

// The two seats' fields are arrays of two; the seat is often a run-time value (the player to act). An array indexed by a run-time value
// cannot live in registers: the whole state then sits in private (scratch) memory on the GPU and every access is a vector-memory round trip
// (the batched step kernels carried 400-464 bytes of scratch per lane). Reads and writes by seat go through these select forms instead.
template <class T>
PRL_HD PRL_INLINE T prl_at2(const T (&a)[2], int p) { return p ? a[1] : a[0]; }
template <class T, class V>
PRL_HD PRL_INLINE void prl_put2(T (&a)[2], int p, V v) {
    const T x = (T)v, a0 = a[0], a1 = a[1];
    a[0] = p ? a0 : x;
    a[1] = p ? x : a1;
}

PRL_HD PRL_INLINE void prl_player_bet_raise(PrlEnvState& s, int p, int total) {  // _PokerPlayer.py:68-81
    prl_put2(s.acted, p, 1);
    const int stack = prl_at2(s.stack, p) - (total - prl_at2(s.bet, p));
    prl_put2(s.stack, p, stack);
    prl_put2(s.bet, p, total);
    if (stack == 0) prl_put2(s.allin, p, 1);
}

// HU sweep of the current bets into the main pot (PokerEnv.py:539-551): refund the uncalled excess first
PRL_HD PRL_INLINE void prl_sweep_bets(PrlEnvState& s) {
    int dif = s.bet[0] - s.bet[1];
    if (dif > 0) { s.stack[0] += dif; s.bet[0] -= dif; }
    else if (dif < 0) { s.stack[1] += -dif; s.bet[1] -= -dif; }
    s.main_pot += s.bet[0];
    s.main_pot += s.bet[1];
    s.bet[0] = 0;
    s.bet[1] = 0;
}

PRL_HD PRL_INLINE void prl_env_reset(const PrlGame& g, PrlEnvState& s) {  // PokerEnv.py:1075-1122 (public part)
    s.n_raises_round = (g.game_type == PRL_GAME_LIMIT && g.big_blind > 0) ? 1 : 0;
    s.main_pot = 0;
    s.round = PRL_PREFLOP;
    s.capped_happened = 0; s.capped_raiser = -1; s.capped_cant_reopen = -1;
    s.last_raiser = -1;
    s.n_actions_ep = 0;
    s.last_action[0] = s.last_action[1] = s.last_action[2] = -1;
    s.pad0 = 0;
    s.stack[0] = g.start_stack[0]; s.stack[1] = g.start_stack[1];
    s.bet[0] = s.bet[1] = 0; s.allin[0] = s.allin[1] = 0; s.folded[0] = s.folded[1] = 0; s.acted[0] = s.acted[1] = 0;
    prl_player_bet_raise(s, 0, g.ante); s.acted[0] = 0;
    prl_player_bet_raise(s, 1, g.ante); s.acted[1] = 0;
    prl_sweep_bets(s);                                   // antes do not count as current bet
    prl_player_bet_raise(s, 0, g.small_blind); s.acted[0] = 0;   // HU: seat 0 = BTN = SB (PokerEnv.py:337-340)
    prl_player_bet_raise(s, 1, g.big_blind); s.acted[1] = 0;
    s.cur = 0;                                           // first to act pre-flop (PokerEnv.py:834-840)
}

PRL_HD PRL_INLINE int prl_total_to_call(const PrlEnvState& s) { return prl_imax(s.bet[0], s.bet[1]); }

PRL_HD PRL_INLINE int prl_min_raise_total(const PrlGame& g, const PrlEnvState& s) {  // PokerEnv.py:809-812
    int small = prl_imin(s.bet[0], s.bet[1]), big = prl_imax(s.bet[0], s.bet[1]);
    return big + prl_imax(big - small, g.big_blind);
}

// pot fraction -> total chips in front (PokerEnv.py:1376-1396). The only float op of the engine: Python computes
// int(to_call + pot_after_call * fraction) in float64 with separate multiply and add (build with -ffp-contract=off).
PRL_HD PRL_INLINE int prl_fraction_of_pot_raise(const PrlEnvState& s, double fraction, int p) {
    const int bet_p = prl_at2(s.bet, p);
    int to_call = prl_total_to_call(s) - bet_p;
    int pot_after_call = s.main_pot + s.bet[0] + s.bet[1] + to_call;
    double prod = (double)pot_after_call * fraction;
    double sum = (double)to_call + prod;
    int delta = (int)sum;  // truncation toward zero == Python int()
    return delta + bet_p;
}

// env-specific action int -> (type, amount) (LimitPokerEnv.py:27-35, DiscretizedPokerEnv.py:47-62, evaluation mode)
PRL_HD PRL_INLINE void prl_adjust_action(const PrlGame& g, const PrlEnvState& s, int action_int, int* type, int* amount) {
    if (action_int == 0) { *type = PRL_FOLD; *amount = -1; return; }
    if (action_int == 1) { *type = PRL_CHECK_CALL; *amount = -1; return; }
    *type = PRL_BET_RAISE;
    if (g.game_type == PRL_GAME_DISCRETIZED) *amount = prl_fraction_of_pot_raise(s, g.bet_fracs[action_int - 2], s.cur);
    else *amount = -1;  // fixed in prl_fixed_action
}

PRL_HD PRL_INLINE void prl_process_check_call(const PrlEnvState& s, int total_to_call, int* type, int* amount) {
    const int p = s.cur, bet_p = prl_at2(s.bet, p);
    int delta = prl_imin(total_to_call - bet_p, prl_at2(s.stack, p));
    *type = PRL_CHECK_CALL;
    *amount = delta + bet_p;
}

// PokerEnv.py:885-941
PRL_HD PRL_INLINE void prl_fixed_action(const PrlGame& g, const PrlEnvState& s, int type, int amount, int* ftype, int* famount) {
    const int p = s.cur, bet_p = prl_at2(s.bet, p), stack_p = prl_at2(s.stack, p);
    int ttc = prl_total_to_call(s);
    if (type == PRL_FOLD) {
        if (ttc <= bet_p) { prl_process_check_call(s, ttc, ftype, famount); return; }
        *ftype = PRL_FOLD; *famount = -1; return;
    }
    if (type == PRL_CHECK_CALL) {
        if (g.first_action_no_call && s.n_actions_ep == 0 && s.round == PRL_PREFLOP) { *ftype = PRL_FOLD; *famount = -1; return; }
        prl_process_check_call(s, ttc, ftype, famount);
        return;
    }
    // BET_RAISE
    if (g.game_type == PRL_GAME_LIMIT && s.n_raises_round >= g.max_raises[s.round]) { prl_process_check_call(s, ttc, ftype, famount); return; }
    if (stack_p + bet_p <= ttc || s.capped_cant_reopen == p) { prl_process_check_call(s, ttc, ftype, famount); return; }
    int raise_to;
    if (g.pot_size_raise) raise_to = prl_fraction_of_pot_raise(s, 1.0, p);                       // games.py:253-254
    else if (g.game_type == PRL_GAME_LIMIT)
        raise_to = (s.n_raises_round + 1) * (s.round >= g.round_big_bet_starts ? g.big_bet : g.small_bet);  // LimitPokerEnv.py:37-39
    else raise_to = prl_imax(prl_min_raise_total(g, s), amount);                                 // Discretized/NoLimit _adjust_raise
    if (bet_p + stack_p < raise_to) raise_to = stack_p + bet_p;
    *ftype = PRL_BET_RAISE;
    *famount = raise_to;
}

PRL_HD PRL_INLINE int prl_should_continue(const PrlEnvState& s) {  // PokerEnv.py:943-954
    int n_nonfold = (!s.folded[0]) + (!s.folded[1]);
    if (n_nonfold < 2) return 0;
    int largest = prl_imax(s.bet[0], s.bet[1]);
    int n_ok = 0, n_unacted = 0;
    if (!s.folded[0]) { n_ok += (s.allin[0] || s.bet[0] == largest); n_unacted += (!s.allin[0] && !s.acted[0]); }
    if (!s.folded[1]) { n_ok += (s.allin[1] || s.bet[1] == largest); n_unacted += (!s.allin[1] && !s.acted[1]); }
    if (n_ok == n_nonfold && n_unacted == 0) return 0;
    return 1;
}

// The first half of a betting step: the PROCESSED action (type, amount) of the current player is fixed and applied, nothing
// else moves -- bets stay in front of the players, round / current player / raise caps unchanged. This is the state
// PokerEnv._step hands out as info["state_dict_before_money_move"] on a round transition (PokerEnv.py:681-728,761-766).
PRL_HD PRL_INLINE void prl_env_apply_action(const PrlGame& g, PrlEnvState& s, int type, int amount, int* out_ftype, int* out_famount) {
    int ftype, famount;
    prl_fixed_action(g, s, type, amount, &ftype, &famount);
    int p = s.cur;
    if (ftype == PRL_CHECK_CALL) {               // _PokerPlayer.py:83-95
        prl_put2(s.acted, p, 1);
        const int stack = prl_at2(s.stack, p) - (famount - prl_at2(s.bet, p));
        prl_put2(s.stack, p, stack);
        prl_put2(s.bet, p, famount);
        if (stack == 0) prl_put2(s.allin, p, 1);
    } else if (ftype == PRL_FOLD) {
        prl_put2(s.acted, p, 1);
        prl_put2(s.folded, p, 1);
    } else {                                     // PokerEnv.py:706-728
        if (famount < prl_min_raise_total(g, s)) {
            s.capped_happened = 1;
            s.capped_raiser = (int8_t)p;
            s.capped_cant_reopen = s.last_raiser;
        } else if (s.capped_happened) {
            if (s.capped_cant_reopen != p) { s.capped_happened = 0; s.capped_raiser = -1; s.capped_cant_reopen = -1; }
        }
        s.last_raiser = (int8_t)p;
        prl_player_bet_raise(s, p, famount);
        s.n_actions_ep += 1;
        if (g.game_type == PRL_GAME_LIMIT) s.n_raises_round += 1;
    }
    s.last_action[0] = ftype; s.last_action[1] = famount; s.last_action[2] = p;
    *out_ftype = ftype;
    *out_famount = famount;
}

// One betting step with a PROCESSED action (type, amount) for the current player. Cards are not handled here: on a
// round transition the caller deals (chance_acts), on a showdown the caller evaluates hands. Mirrors PokerEnv._step.
PRL_HD PRL_INLINE void prl_env_step_processed(const PrlGame& g, PrlEnvState& s, int type, int amount, PrlStepInfo* info) {
    int ftype, famount;
    const int p = s.cur;
    prl_env_apply_action(g, s, type, amount, &ftype, &famount);

    info->fixed_type = ftype; info->fixed_amount = famount;
    info->is_terminal = 0; info->chance_acts = 0; info->terminal_is_fold = 0; info->rundown = 0; info->pot_before_payout = 0;

    int n_nonfold = (!s.folded[0]) + (!s.folded[1]);
    int n_active = (!s.folded[0] && !s.allin[0]) + (!s.folded[1] && !s.allin[1]);
    if (prl_should_continue(s)) {
        int q = 1 - p;                           // HU: the other seat, if it can act (PokerEnv.py:871-883)
        if (prl_at2(s.allin, q) || prl_at2(s.folded, q)) q = p;
        s.cur = (int8_t)q;
    } else if (n_active > 1) {
        if (s.round == g.n_rounds - 1) {         // showdown on the last street
            info->is_terminal = 1;
            prl_sweep_bets(s);
            info->pot_before_payout = s.main_pot;
        } else {                                 // next round (PokerEnv.py:661-679); caller deals the cards
            info->chance_acts = 1;
            if (g.game_type == PRL_GAME_LIMIT) s.n_raises_round = 0;
            s.capped_happened = 0; s.capped_raiser = -1; s.capped_cant_reopen = -1;
            prl_sweep_bets(s);
            s.cur = (int8_t)(g.btn_first_postflop ? 0 : 1);
            s.acted[0] = 0; s.acted[1] = 0;
            s.round += 1;
        }
    } else if (n_nonfold > 1) {                  // all-in run-out (PokerEnv.py:620-644)
        info->is_terminal = 1;
        info->rundown = 1;
        prl_sweep_bets(s);
        info->pot_before_payout = s.main_pot;
        s.round = g.n_rounds - 1;
    } else {                                     // everybody else folded
        info->is_terminal = 1;
        info->terminal_is_fold = 1;
        prl_sweep_bets(s);
        info->pot_before_payout = s.main_pot;
    }
}

PRL_HD PRL_INLINE void prl_env_step(const PrlGame& g, PrlEnvState& s, int action_int, PrlStepInfo* info) {
    int type, amount;
    prl_adjust_action(g, s, action_int, &type, &amount);
    prl_env_step_processed(g, s, type, amount, info);
}

// Legal actions (LimitPokerEnv.py:41-59, DiscretizedPokerEnv.py:99-135). Returns the count, fills `out` (env action ints)
// `put(a)` receives the legal actions in ascending order
template <class Put>
PRL_HD PRL_INLINE int prl_legal_actions_to(const PrlGame& g, const PrlEnvState& s, Put& put) {
    int n = 0, ft, fa;
    prl_fixed_action(g, s, PRL_FOLD, -1, &ft, &fa);
    if (ft == PRL_FOLD) { put(PRL_FOLD); ++n; }
    prl_fixed_action(g, s, PRL_CHECK_CALL, -1, &ft, &fa);
    if (ft == PRL_CHECK_CALL) { put(PRL_CHECK_CALL); ++n; }
    if (g.game_type == PRL_GAME_LIMIT) {
        prl_fixed_action(g, s, PRL_BET_RAISE, -1, &ft, &fa);
        if (s.n_raises_round < g.max_raises[s.round] && ft == PRL_BET_RAISE) { put(PRL_BET_RAISE); ++n; }
        return n;
    }
    if (g.game_type == PRL_GAME_NOLIMIT) {  // PokerEnv.get_legal_actions (PokerEnv.py:1313-1330): probe raise amount 1
        prl_fixed_action(g, s, PRL_BET_RAISE, 1, &ft, &fa);
        if (ft == PRL_BET_RAISE) { put(PRL_BET_RAISE); ++n; }
        return n;
    }
    int last_too_small = -1;
    int n_actions = g.n_bet_sizes + 2;
    for (int a = 2; a < n_actions; ++a) {
        int t, amt;
        prl_adjust_action(g, s, a, &t, &amt);
        prl_fixed_action(g, s, t, amt, &ft, &fa);
        if (ft != t) break;                       // env turned the raise into a call: no raises at all
        if (amt < fa) {
            last_too_small = a;                   // below min-raise: remember the largest such size
        } else {
            if (last_too_small >= 0) { put(last_too_small); ++n; last_too_small = -1; }
            put(a); ++n;
        }
        if (amt > fa) break;                      // clamped to all-in: bigger sizes collapse to the same raise
    }
    return n;
}
PRL_HD PRL_INLINE int prl_legal_actions(const PrlGame& g, const PrlEnvState& s, int32_t* out) {
    int k = 0;
    auto put = [&](int a) { out[k++] = a; };
    return prl_legal_actions_to(g, s, put);
}
// legal[r % n_legal] without the list (no array indexed at run time). Up to 64 action ints: ONE enumeration into a bit mask, then the
// (r % n)-th set bit (round 6; the enumeration -- a float64 pot fraction and the fixed-action rules per bet size -- was the random
// player's main cost). More action ints: one pass counts, one picks.
PRL_HD PRL_INLINE int prl_legal_action_pick(const PrlGame& g, const PrlEnvState& s, uint32_t r) {
    if (g.game_type != PRL_GAME_DISCRETIZED || g.n_bet_sizes + 2 <= 64) {
        uint32_t lo = 0u, hi = 0u;
        auto mark = [&](int a) { lo |= a < 32 ? 1u << (a & 31) : 0u; hi |= a < 32 ? 0u : 1u << (a & 31); };
        const int n = prl_legal_actions_to(g, s, mark);
        const int want = (int)(r % (uint32_t)n);
        for (int q = 0; q < want; ++q) {  // drop the lowest set bit
            const uint32_t l2 = lo & (lo - 1u);
            hi = lo ? hi : hi & (hi - 1u);
            lo = l2;
        }
        if (!(lo | hi)) return -1;
        return lo ? __builtin_ctz(lo) : 32 + __builtin_ctz(hi);
    }
    auto nothing = [](int) {};
    const int n = prl_legal_actions_to(g, s, nothing);
    int want = (int)(r % (uint32_t)n), k = 0, picked = -1;
    auto pick = [&](int a) { picked = k == want ? a : picked; ++k; };
    prl_legal_actions_to(g, s, pick);
    return picked;
}
