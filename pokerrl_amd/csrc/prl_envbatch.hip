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
#include <string.h>

#include <string>
#include <vector>

#include "prl_device.h"
#include "prl_env.h"
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
};

PRL_DEV PRL_INLINE void eb_load(const int32_t* st, int n, int i, PrlEnvState& s, bool* done) {
    s.round = st[(size_t)EB_ROUND * n + i];
    s.main_pot = st[(size_t)EB_POT * n + i];
    s.bet[0] = st[(size_t)EB_BET0 * n + i];
    s.bet[1] = st[(size_t)EB_BET1 * n + i];
    s.stack[0] = st[(size_t)EB_STACK0 * n + i];
    s.stack[1] = st[(size_t)EB_STACK1 * n + i];
    const int f = st[(size_t)EB_FLAGS * n + i];
    s.allin[0] = f & 1; s.allin[1] = (f >> 1) & 1;
    s.folded[0] = (f >> 2) & 1; s.folded[1] = (f >> 3) & 1;
    s.acted[0] = (f >> 4) & 1; s.acted[1] = (f >> 5) & 1;
    s.cur = (int8_t)((f >> 6) & 1);
    s.capped_happened = (int8_t)((f >> 7) & 1);
    *done = ((f >> 8) & 1) != 0;
    const int w = st[(size_t)EB_SEATS * n + i];
    s.last_raiser = (int8_t)((w & 0xFF) - 1);
    s.capped_raiser = (int8_t)(((w >> 8) & 0xFF) - 1);
    s.capped_cant_reopen = (int8_t)(((w >> 16) & 0xFF) - 1);
    s.pad0 = 0;
    s.n_actions_ep = st[(size_t)EB_NACT * n + i];
    s.n_raises_round = st[(size_t)EB_NRAISES * n + i];
    s.last_action[0] = st[(size_t)EB_LA_TYPE * n + i];
    s.last_action[1] = st[(size_t)EB_LA_AMOUNT * n + i];
    s.last_action[2] = st[(size_t)EB_LA_SEAT * n + i];
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
        steps += (unsigned long long)(unsigned)prl_shfl_i((int)(unsigned)steps, src);
        hands += (unsigned long long)(unsigned)prl_shfl_i((int)(unsigned)hands, src);
        const unsigned lo = (unsigned)prl_shfl_i((int)(unsigned)(pots & 0xFFFFFFFFull), src);
        const unsigned hi = (unsigned)prl_shfl_i((int)(unsigned)(pots >> 32), src);
        pots += ((unsigned long long)hi << 32) | lo;
    }
    if (prl_lane() == 0) {
        unsigned long long* s = stats + 3 * (((prl_bid() * prl_nthreads() + prl_tid()) >> 6) & (EB_STAT_SLOTS - 1));
        prl_atomic_add_u64(s, steps);
        prl_atomic_add_u64(s + 1, hands);
        prl_atomic_add_u64(s + 2, pots);
    }
}

PRL_GLOBAL void prl_k_eb_reset(const PrlGame* g, int32_t* st, int n, const uint8_t* mask) {
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
PRL_GLOBAL void prl_k_eb_step(const PrlGame* g, int32_t* st, int n, const int32_t* a0, const int32_t* a1, int processed, int32_t* info) {
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

PRL_GLOBAL void prl_k_eb_legal(const PrlGame* g, const int32_t* st, int n, uint32_t* mask4, int32_t* count) {
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
PRL_GLOBAL void prl_k_eb_active(const int32_t* st, int n, int32_t* out_idx, int32_t* out_count) {
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
PRL_GLOBAL void prl_k_eb_rollout(const PrlGame* g, int32_t* st, int n, int n_steps, uint32_t seed, unsigned long long* stats) {
    unsigned long long steps = 0, hands = 0, pots = 0;
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n; i += (int)(prl_nblocks() * prl_nthreads())) {
        PrlEnvState s;
        bool done;
        eb_load(st, n, i, s, &done);
        if (done) { prl_env_reset(*g, s); done = false; }
        for (int k = 0; k < n_steps; ++k) {
            int32_t legal[PRL_MAX_BET_SIZES + 2];
            const int nl = prl_legal_actions(*g, s, legal);
            const uint32_t r = eb_mix32(seed ^ eb_mix32((uint32_t)i * 0x9E3779B9u + (uint32_t)k));
            PrlStepInfo si;
            const int a = legal[r % (uint32_t)nl];
            if (g->game_type == PRL_GAME_NOLIMIT) prl_env_step_processed(*g, s, a, a == PRL_BET_RAISE ? (int)(eb_mix32(r) % (uint32_t)(s.stack[s.cur] + s.bet[s.cur] + 1)) : -1, &si);
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
PRL_GLOBAL void prl_k_eb_random_step(const PrlGame* g, int32_t* st, int n, int k, uint32_t seed, unsigned long long* stats) {
    unsigned long long hands = 0, pots = 0, steps = 0;
    for (int i = (int)(prl_bid() * prl_nthreads() + prl_tid()); i < n; i += (int)(prl_nblocks() * prl_nthreads())) {
        PrlEnvState s;
        bool done;
        eb_load(st, n, i, s, &done);
        if (done) prl_env_reset(*g, s);
        int32_t legal[PRL_MAX_BET_SIZES + 2];
        const int nl = prl_legal_actions(*g, s, legal);
        const uint32_t r = eb_mix32(seed ^ eb_mix32((uint32_t)i * 0x9E3779B9u + (uint32_t)k));
        PrlStepInfo si;
        const int a = legal[r % (uint32_t)nl];
        if (g->game_type == PRL_GAME_NOLIMIT) prl_env_step_processed(*g, s, a, a == PRL_BET_RAISE ? (int)(eb_mix32(r) % (uint32_t)(s.stack[s.cur] + s.bet[s.cur] + 1)) : -1, &si);
        else prl_env_step(*g, s, a, &si);
        ++steps;
        if (si.is_terminal) { ++hands; pots += (unsigned long long)si.pot_before_payout; }
        eb_store(st, n, i, s, si.is_terminal != 0);
    }
    eb_stats_add(stats, steps, hands, pots);
}

static int eb_grid(int n) {
    int g = (n + 255) / 256;
    return g < 1 ? 1 : (g > 4096 ? 4096 : g);
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
    if (hipMalloc((void**)&b->d_stats, (size_t)3 * 256 * sizeof(unsigned long long)) != hipSuccess) return fail("stats");
    PRL_HIP_TRY(hipMemcpy(b->d_game, game, sizeof(PrlGame), hipMemcpyHostToDevice));
    *out = b;
    return prl_envbatch_reset(b, nullptr);
}

void prl_envbatch_destroy(prl_envbatch_t* b) {
    if (!b) return;
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    void* ptrs[] = {b->d_state, b->d_game, b->d_a, b->d_b, b->d_info, b->d_mask, b->d_count, b->d_stats};
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
            int32_t legal[PRL_MAX_BET_SIZES + 2];
            const int nl = prl_legal_actions(*game, s, legal);
            const uint32_t r = eb_mix32(seed ^ eb_mix32((uint32_t)i * 0x9E3779B9u + (uint32_t)k));
            PrlStepInfo si;
            const int a = legal[r % (uint32_t)nl];
            if (game->game_type == PRL_GAME_NOLIMIT) prl_env_step_processed(*game, s, a, a == PRL_BET_RAISE ? (int)(eb_mix32(r) % (uint32_t)(s.stack[s.cur] + s.bet[s.cur] + 1)) : -1, &si);
            else prl_env_step(*game, s, a, &si);
            ++steps;
            if (si.is_terminal) { ++hands; pots += (uint64_t)si.pot_before_payout; prl_env_reset(*game, s); }
        }
    }
    out_stats3[0] = steps; out_stats3[1] = hands; out_stats3[2] = pots;
    return PRL_OK;
}

}  // extern "C"
