/*
 * pokerrl_hip.h -- C ABI of libpokerrl_hip.so, the MI355X (gfx950) native replacement for PokerRL's tabular hot path:
 * public-tree CFR / exact best response, range-vs-range terminal equity, the 7-card hand evaluator and the card/hand
 * index look-up tables.
 *
 * Boundary rules:
 *   - plain C, `extern "C"`, pointers + sizes only; no torch / numpy / C++ types cross it;
 *   - every host buffer is allocated AND owned by the caller, the callee only reads inputs and fills outputs;
 *   - device memory lives behind opaque handles created / destroyed explicitly;
 *   - new entry points return an int32 status (PRL_OK == 0, negative = error) and set prl_last_error(); nothing aborts;
 *   - a handle is not thread-safe, different handles are; each handle owns its HIP stream.
 *
 * Each group below cites the reference interface it replaces (paths under the upstream PokerRL repository).
 */
#ifndef POKERRL_HIP_H
#define POKERRL_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------------------------------------------------------- */
/* constants                                                                                                         */
/* ---------------------------------------------------------------------------------------------------------------- */
#define PRL_MAX_BET_SIZES 96

/* status codes */
#define PRL_OK 0
#define PRL_ERR_ARG (-1)
#define PRL_ERR_NO_DEVICE (-2)
#define PRL_ERR_HIP (-3)
#define PRL_ERR_UNSUPPORTED (-4)
#define PRL_ERR_OOM (-5)
#define PRL_ERR_STATE (-6)

/* ---------------------------------------------------------------------------------------------------------------- */
/* plain-old-data descriptions of a game (reference: PokerRL/game/games.py:18-269, game_rules.py:15-312)              */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct PrlRules {
    int32_t n_hole_cards;            /* 1 (Leduc family) or 2 (Hold'em family) */
    int32_t n_ranks;
    int32_t n_suits;
    int32_t n_cards;                 /* n_ranks * n_suits */
    int32_t range_size;              /* C(n_cards, n_hole_cards) */
    int32_t n_rounds;                /* len(ALL_ROUNDS_LIST) */
    int32_t board_cards_in_round[4]; /* cards dealt in the transition TO round r */
    int32_t n_board_cards;
    int32_t btn_first_postflop;
    int32_t rank_rule;               /* 0 Leduc (100 + rank), 1 BigLeduc (10000 + rank), 2 52-card hold'em evaluator */
} PrlRules;

typedef struct PrlGame {
    int32_t game_type;               /* 0 fixed-limit, 1 discretized no-limit, 2 no-limit */
    int32_t n_rounds;
    int32_t small_blind, big_blind, ante;
    int32_t small_bet, big_bet;
    int32_t round_big_bet_starts;
    int32_t max_raises[4];
    int32_t first_action_no_call;
    int32_t btn_first_postflop;
    int32_t pot_size_raise;          /* Flop5Holdem: every raise is pot-sized (games.py:253-254) */
    int32_t n_bet_sizes;
    int32_t start_stack[2];
    double bet_fracs[PRL_MAX_BET_SIZES]; /* ascending */
} PrlGame;

/* public betting state of one heads-up env (reference: PokerEnv.state_dict, PokerEnv.py:1161-1197, HU subset) */
typedef struct PrlEnvState {
    int32_t round;
    int32_t main_pot;
    int32_t bet[2];
    int32_t stack[2];
    int8_t allin[2];
    int8_t folded[2];
    int8_t acted[2];
    int8_t cur;                 /* seat to act */
    int8_t last_raiser;         /* -1 = None */
    int8_t capped_happened;
    int8_t capped_raiser;       /* -1 = None */
    int8_t capped_cant_reopen;  /* -1 = None */
    int8_t pad0;
    int32_t n_actions_ep;       /* counts raises only (PokerEnv.py:724) */
    int32_t n_raises_round;
    int32_t last_action[3];     /* type, amount, seat (-1 = None) */
} PrlEnvState;

/* result of one step: the `info` dict of PokerEnv._step with RETURN_PRE_TRANSITION_STATE_IN_INFO (PokerEnv.py:737-787) */
typedef struct PrlStepInfo {
    int32_t is_terminal;
    int32_t chance_acts;        /* round transition: cards must be dealt before play continues */
    int32_t terminal_is_fold;   /* only one player left */
    int32_t rundown;            /* terminal reached through an all-in run-out */
    int32_t pot_before_payout;  /* main pot after the bet sweep, before any payout */
    int32_t fixed_type, fixed_amount; /* the action after _get_fixed_action */
} PrlStepInfo;

const char* prl_last_error(void);
/* 1 if a HIP device is usable by this library build, else 0 (never throws) */
int32_t prl_device_available(void);
/* one process per GPU: make HIP device `ordinal` (the launcher's LOCAL_RANK) the device of every handle this thread creates
   from now on; handles keep the device they were created on. Returns PRL_ERR_NO_DEVICE if the ordinal does not exist. */
int32_t prl_set_device(int32_t ordinal);
/* compile-time identification: "hip-gfx950" for the product build */
const char* prl_build_flavor(void);

/* ---------------------------------------------------------------------------------------------------------------- */
/* 1. Legacy symbols: drop-in for the reference's binary-only lib_luts.so / lib_hand_eval.so                           */
/*    (ctypes call sites: PokerRL/game/_/cpp_wrappers/CppLUT.py:16-94, CppHandeval.py:19-65;                           */
/*     2-D arrays arrive as vectors of ROW POINTERS, PokerRL/_/CppWrapper.py:14,24-27).                                 */
/*    Signatures are the reference's; sizes are fixed by the 52-card deck.                                             */
/* ---------------------------------------------------------------------------------------------------------------- */
int8_t get_1d_card(const int8_t* card_2d);                       /* CppLUT.py:73-82   rank*4+suit          */
void get_2d_card(int8_t card_1d, int8_t* out_card_2d);           /* CppLUT.py:84-94   (c/4, c%4)           */
void get_idx_2_hole_card_lut(int8_t** out_1326x2);               /* CppLUT.py:38-41                          */
void get_hole_card_2_idx_lut(int16_t** out_52x52);               /* CppLUT.py:43-47   upper triangle only   */
/* CppLUT.py:27-34,49-71: bound by CppLibHoldemLuts.__init__, never called by the reference (its binary's versions crash); row i =
 * the i-th k-card board in ascending lexicographic order of ascending 1d cards: 22 100 x 3, 270 725 x 4, 2 598 960 x 5 */
void get_idx_2_flop_lut(int8_t** out_22100x3);
void get_idx_2_turn_lut(int8_t** out_270725x4);
void get_idx_2_river_lut(int8_t** out_2598960x5);
/* CppHandeval.py:34-43: hand_2d[2][2], board_2d[5][2] as (rank, suit) rows. Scalar call -> host evaluator. */
int32_t get_hand_rank_52_holdem(int8_t** hand_2d, int8_t** board_2d);
/* CppHandeval.py:45-65: out[N][1326] pre-filled with -1 by the caller; boards_1d[N][5]; the two LUT arguments of the
 * reference are accepted and ignored (the library owns identical tables). Runs on the GPU. */
void get_hand_rank_all_hands_on_given_boards_52_holdem(int32_t** out, int8_t** boards_1d, int32_t n_boards,
                                                       int8_t** lut_idx_2_hole_cards, int8_t** lut_1d_2_2d);

/* ---------------------------------------------------------------------------------------------------------------- */
/* 2. Flat-pointer LUT / evaluator API (what new callers bind)                                                        */
/*    replaces look_up_table.py:95-134,191-220 and game_rules.py:213-223                                               */
/* ---------------------------------------------------------------------------------------------------------------- */
int32_t prl_lut_idx_2_hole_cards(const PrlRules* rules, int8_t* out /* [range_size][n_hole_cards] */);
int32_t prl_lut_hole_cards_2_idx(const PrlRules* rules, int16_t* out /* [n_cards][n_cards], -2 where undefined */);
int32_t prl_lut_card_in_what_range_idxs(const PrlRules* rules, int32_t* out /* [n_cards][n_cards-1] or [n_cards][1] */);
/* host scalar evaluator: 7 cards of the 52-card deck as 1d cards */
int32_t prl_hand_rank_7(const int8_t* board_1d /*[5]*/, int8_t c1, int8_t c2);
/* batched evaluator on the GPU, host buffers in and out: ranks[n][1326] int32, -1 for hands blocked by the board */
int32_t prl_hand_rank_boards(const int8_t* boards_1d /*[n][5]*/, int32_t n_boards, int32_t* out_ranks);
/* verification aid: one order-sensitive u64 checksum of the rank table per `chunk` boards, computed on the GPU without
 * materialising the table (an exhaustive C(52,5) sweep is 13.8 GB); definition in csrc/prl_handeval_kernels.hip */
int32_t prl_hand_rank_checksums(const int8_t* boards_1d, int32_t n_boards, int32_t chunk, uint64_t* out_checksums);
/* same, device pointers in and out (no copies); stream = hipStream_t or NULL */
int32_t prl_hand_rank_boards_device(const void* d_boards_1d, int32_t n_boards, void* d_out_ranks, void* stream);

/* ---------------------------------------------------------------------------------------------------------------- */
/* 3. Public tree (host): replaces PublicTree.build_tree (PokerRL/game/_/tree/PublicTree.py:111-126,161-293)           */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct prl_tree prl_tree_t;

/* boards: the run-outs, one row of `board_len` 1d cards per run-out in deal order. A game that deals once (Leduc, Flop5Holdem) has one
 * chance level whose children are these rows in order; a game that deals on several streets (hold'em: 3 + 1 + 1) gets one chance level
 * per street whose children are the distinct prefixes of the listed run-outs. An all-in before the last street of a 2-hole-card game
 * is dealt out as a chain of chance nodes down to showdown leaves (1-hole-card games keep the reference's run-out terminal). */
int32_t prl_tree_build(const PrlGame* game, const PrlRules* rules, const int8_t* boards, int32_t n_boards,
                       int32_t board_len, prl_tree_t** out_tree);
/* PublicTree(stop_at_street = stop_at_round) (PublicTree.py:72,173,185): nodes of a betting round >= stop_at_round are not expanded
 * -- decision nodes without children. Structure and states only: prl_solver_create* refuses a partial tree. stop_at_round < 0 = full. */
int32_t prl_tree_build_partial(const PrlGame* game, const PrlRules* rules, const int8_t* boards, int32_t n_boards, int32_t board_len,
                               int32_t stop_at_round, prl_tree_t** out_tree);
void prl_tree_destroy(prl_tree_t* tree);

enum {
    PRL_TI_N_NODES = 0, PRL_TI_N_COLS = 1, PRL_TI_N_BOARDS = 2, PRL_TI_BOARD_LEN = 3, PRL_TI_N_LEVELS = 4,
    PRL_TI_RANGE_SIZE = 5, PRL_TI_N_DECISION = 6, PRL_TI_N_TERMINAL = 7, PRL_TI_COUNT = 8
};
int32_t prl_tree_info(const prl_tree_t* tree, int32_t* out_info /* [PRL_TI_COUNT] */);

enum {
    PRL_TF_KIND = 0,        /* 0 decision, 1 chance, 2 terminal fold, 3 terminal showdown            [n_nodes] */
    PRL_TF_ACTOR = 1,       /* seat to act, -1 for chance / terminal                                 [n_nodes] */
    PRL_TF_PARENT = 2,
    PRL_TF_CHILD_IDX = 3,   /* position in the parent's child list                                    */
    PRL_TF_ACTION = 4,      /* env action int that produced the node, -1 for "CHANCE"/root            */
    PRL_TF_ACTED_LAST = 5,  /* seat that produced the node, -2 chance, -1 root                        */
    PRL_TF_ROUND = 6,
    PRL_TF_BOARD_ID = 7,    /* row of the board table, -1 before the deal                             */
    PRL_TF_MAIN_POT = 8,
    PRL_TF_DEPTH = 9,
    PRL_TF_N_CHILDREN = 10,
    PRL_TF_FIRST_COL = 11,  /* first action column of a decision node, -1 otherwise                   */
    PRL_TF_SUBTREE_SIZE = 12,
    PRL_TF_CHILD_START = 13, /* CSR offsets                                                   [n_nodes+1] */
    PRL_TF_CHILD_LIST = 14,  /* CSR payload                                                   [n_nodes-1] */
    PRL_TF_COL_ACTION = 15,  /* env action int of each action column                            [n_cols] */
    PRL_TF_COL_NODE = 16,    /* owning decision node of each action column                      [n_cols] */
    PRL_TF_LEVEL_START = 17, /*                                                               [n_levels+1] */
    PRL_TF_LEVEL_NODES = 18  /* node ids grouped by depth                                       [n_nodes] */
};
int32_t prl_tree_get(const prl_tree_t* tree, int32_t field, int32_t* out);
/* the board table node.board_id indexes: [n_boards][board_len] 1d cards, one row per board PREFIX of every dealing street (cards
 * not dealt yet = -1); for a game that deals once these are the caller's rows */
int32_t prl_tree_get_boards(const prl_tree_t* tree, int8_t* out_rows);


/* ---------------------------------------------------------------------------------------------------------------- */
/* 4. Heads-up betting engine, one env on the host (tree construction, tests, Python PokerEnv facade).                 */
/*    replaces PokerEnv.{reset,step,get_legal_actions} public-state semantics (PokerEnv.py:681-789,885-941,1075-1122;  */
/*    LimitPokerEnv.py:27-59; DiscretizedPokerEnv.py:44-135). Cards are dealt by the caller.                           */
/* ---------------------------------------------------------------------------------------------------------------- */
int32_t prl_env_reset_host(const PrlGame* game, PrlEnvState* state);
int32_t prl_env_step_host(const PrlGame* game, PrlEnvState* state, int32_t action_int, PrlStepInfo* out_info);
/* processed (type, amount) form == PokerEnv.step_from_processed_tuple */
int32_t prl_env_step_processed_host(const PrlGame* game, PrlEnvState* state, int32_t type, int32_t amount, PrlStepInfo* out_info);
/* the action half of a step only: the current player's action is fixed and applied, no bet sweep / round transition / payout.
 * This is PokerEnv._step's info["state_dict_before_money_move"] state (PokerEnv.py:761-766). action_int form if !is_processed. */
int32_t prl_env_apply_action_host(const PrlGame* game, PrlEnvState* state, int32_t action_int, int32_t is_processed, int32_t type, int32_t amount,
                                  PrlStepInfo* out_info);
int32_t prl_env_legal_actions_host(const PrlGame* game, const PrlEnvState* state, int32_t* out_actions, int32_t* out_n);
int32_t prl_env_fraction_of_pot_raise_host(const PrlEnvState* state, double fraction, int32_t seat, int32_t* out_total);

/* 4b. The same engine batched on the device: n_envs independent heads-up envs, public betting state as struct-of-arrays in    */
/*    HBM (PRL_EB_N_COLS int32 columns of n_envs entries), one GPU lane per env. replaces stepping n PokerEnv objects one by one   */
/*    (PokerEnv.py:681-789,885-941,1075-1122,1161-1197,1313-1330; LimitPokerEnv.py:27-59; DiscretizedPokerEnv.py:44-135).       */
/*    Cards stay with the caller as in section 4 (deal on chance_acts, rank the hands on a showdown).                            */
/*    Host-pointer calls copy in / out and synchronise; the *_device forms take device pointers and only enqueue on the batch's   */
/*    stream (prl_envbatch_state_device hands out the stream and the state columns).                                              */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct prl_envbatch prl_envbatch_t;
#define PRL_EB_N_COLS 13
/* state columns (prl_envbatch_get_state: out_cols[PRL_EB_N_COLS][n_envs]) */
enum {
    PRL_EB_ROUND = 0, PRL_EB_MAIN_POT = 1, PRL_EB_BET0 = 2, PRL_EB_BET1 = 3, PRL_EB_STACK0 = 4, PRL_EB_STACK1 = 5,
    PRL_EB_FLAGS = 6,      /* bit 0,1 is_allin[seat]; 2,3 folded; 4,5 has_acted_this_round; 6 current_player; 7 capped raise pending; 8 episode over */
    PRL_EB_SEATS = 7,      /* (last_raiser + 1) | (capped_raise[0] + 1) << 8 | (capped_raise[1] + 1) << 16; 0 = None */
    PRL_EB_N_ACTIONS_EP = 8, PRL_EB_N_RAISES_ROUND = 9, PRL_EB_LAST_ACTION_TYPE = 10, PRL_EB_LAST_ACTION_AMOUNT = 11, PRL_EB_LAST_ACTION_SEAT = 12
};
int32_t prl_envbatch_create(const PrlGame* game, int32_t n_envs, prl_envbatch_t** out_batch);  /* all envs reset */
void prl_envbatch_destroy(prl_envbatch_t* batch);
/* reset the envs with mask[i] != 0 (NULL = all) */
int32_t prl_envbatch_reset(prl_envbatch_t* batch, const uint8_t* mask);
/* one step of every env: actions[i] = env action int (amounts == NULL; PokerEnv.step) or action type with amounts[i] chips
 * (step_from_processed_tuple). An env whose episode is over, or with actions[i] < 0, is skipped. out_info4[4][n_envs] (may be NULL):
 * is_terminal (-1 = skipped), chance_acts, pot before the payout, terminal kind (1 fold, 2 showdown on the last street, 3 all-in run-out) */
int32_t prl_envbatch_step(prl_envbatch_t* batch, const int32_t* actions, const int32_t* amounts, int32_t* out_info4);
int32_t prl_envbatch_step_device(prl_envbatch_t* batch, const int32_t* d_actions, const int32_t* d_amounts, int32_t* d_info4);
/* get_legal_actions of every env: out_mask4[4][n_envs] = 128-bit set of legal action ints (word k covers actions 32k..32k+31),
 * out_count[n_envs] = how many (0 when the episode is over) */
int32_t prl_envbatch_legal_masks(prl_envbatch_t* batch, uint32_t* out_mask4, int32_t* out_count);
/* ids of the envs whose episode is still running (wave ballot + prefix compaction; order: ascending within each group of 64 envs) */
int32_t prl_envbatch_active(prl_envbatch_t* batch, int32_t* out_idx, int32_t* out_count);
int32_t prl_envbatch_get_state(prl_envbatch_t* batch, int32_t* out_cols);
int32_t prl_envbatch_state_device(prl_envbatch_t* batch, void** out_d_cols, void** out_hip_stream);
/* n_steps uniform-random legal steps per env, finished hands restart (counter-based generator keyed by seed, env, step);
 * out_stats3 = steps, finished hands, sum of their pots; prl_env_random_rollout_host plays the same hands on the host */
int32_t prl_envbatch_random_rollout(prl_envbatch_t* batch, int32_t n_steps, uint32_t seed, uint64_t* out_stats3, float* out_device_ms);
/* the same play with the state in HBM between steps: n_launches launches of ONE step per env (13 words in, 13 out per env and step) */
int32_t prl_envbatch_random_steps(prl_envbatch_t* batch, int32_t n_launches, uint32_t seed, uint64_t* out_stats3, float* out_device_ms);
int32_t prl_env_random_rollout_host(const PrlGame* game, int32_t n_envs, int32_t n_steps, uint32_t seed, uint64_t* out_stats3);
/* The WHOLE PokerEnv.step for n envs (PokerEnv.py:737-789: obs, reward, done, info; state incl. deck / board / hands :1161-1197): a batch
 * created with cards also holds every env's hole cards and board (1-d cards [n][2 * n_hole + n_board]: seat 0's hand, seat 1's hand, the
 * board in deal order), dealt from a counter-based deck keyed by (deck_seed, episode * n_envs + env) at every reset -- or set by the caller
 * (prl_envbatch_set_cards: replays, an external dealer). step_full returns per env the observation vector of the reference's heads-up
 * layout (PokerEnv.py:199-261, 1253-1271: float64 quotients rounded to float32; zeros once the episode is over), the two seats' rewards
 * ((stack after the payout - starting stack) / reward_scalar, PokerEnv.py:468-481, 1069-1072; showdowns ranked on the device, a tie pays
 * half the pot each), the done flag and the info words of prl_envbatch_step. Heads-up; 1-hole-card games and 52-card hold'em. */
int32_t prl_envbatch_create_with_cards(const PrlGame* game, const PrlRules* rules, int32_t n_envs, uint64_t deck_seed, double reward_scalar,
                                       prl_envbatch_t** out_batch);
int32_t prl_envbatch_obs_dim(prl_envbatch_t* batch, int32_t* out_dim);
int32_t prl_envbatch_reset_full(prl_envbatch_t* batch, const uint8_t* mask, float* out_obs /* [n][obs_dim], may be NULL */);
int32_t prl_envbatch_set_cards(prl_envbatch_t* batch, const int8_t* cards);
int32_t prl_envbatch_get_cards(prl_envbatch_t* batch, int8_t* out_cards);
int32_t prl_envbatch_observe(prl_envbatch_t* batch, float* out_obs);
int32_t prl_envbatch_step_full(prl_envbatch_t* batch, const int32_t* actions, const int32_t* amounts, float* out_obs, double* out_reward2,
                               uint8_t* out_done, int32_t* out_info4);
int32_t prl_envbatch_step_full_device(prl_envbatch_t* batch, const int32_t* d_actions, const int32_t* d_amounts, float* d_obs, double* d_reward2,
                                      uint8_t* d_done, int32_t* d_info4);
/* whole hands with the state in registers: deal, uniform-random legal betting, showdown ranks, payout, deal again; n_steps steps per env.
 * out_stats4 = steps, finished hands, showdowns, sum over hands of (2 x seat 0's chip winnings + 2^20) (an integer checksum of the
 * payouts); prl_env_random_rollout_full_host plays the same hands on the host */
int32_t prl_envbatch_random_rollout_full(prl_envbatch_t* batch, int32_t n_steps, uint32_t seed, uint64_t* out_stats4, float* out_device_ms);
/* the same play with the state in HBM between steps: n_launches launches of ONE whole step per env -- 13 state words in and out, the
 * observation vector, two rewards and the done flag out per env and step (what an agent-driven rollout moves); out_stats3 as random_steps */
int32_t prl_envbatch_random_steps_full(prl_envbatch_t* batch, int32_t n_launches, uint32_t seed, uint64_t* out_stats3, float* out_device_ms);
/* The observation vectors [n][obs_dim], rewards [n][2] and done flags [n] the last launch of prl_envbatch_random_steps_full left in the batch's own
 * output buffers (any pointer may be NULL). Test / inspection entry point: an agent-driven rollout uses prl_envbatch_step_full_device. */
int32_t prl_envbatch_last_outputs(prl_envbatch_t* batch, float* out_obs, double* out_reward2, uint8_t* out_done);
int32_t prl_env_random_rollout_full_host(const PrlGame* game, const PrlRules* rules, int32_t n_envs, int32_t n_steps, uint32_t seed, uint64_t deck_seed,
                                         double reward_scalar, uint64_t* out_stats4);

/* ---------------------------------------------------------------------------------------------------------------- */
/* 5. Device-resident tabular solver: public-tree CFR / CFR+ / Linear CFR and exact best response on one GPU.          */
/*    replaces  PokerRL/cfr/_CFRBase.py:110-262 (+ VanillaCFR.py, CFRPlus.py, LinearCFR.py),                           */
/*              PokerRL/game/_/tree/_/StrategyFiller.py:17-169, ValueFiller.py:21-175,                                 */
/*              PokerRL/eval/br/LocalBRMaster.py:67-80 (fill strategy -> compute_ev -> root exploitability).           */
/*    Array layout: per-node vectors float32 [n_nodes][2][R]; per-action-column arrays [n_cols][R] where column        */
/*    first_col[node] + a is the reference's node.strategy[:, a]; nodes in DFS pre-order (prl_tree_get).               */
/*    All calls are asynchronous on the solver's stream except the ones that copy results to the host.                 */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct prl_solver prl_solver_t;

enum { PRL_VARIANT_VANILLA = 0, PRL_VARIANT_PLUS = 1, PRL_VARIANT_LINEAR = 2 };

/* create = upload tree + build showdown plans + CFRBase.reset(); `delay` is CFR+'s linear-averaging delay */
int32_t prl_solver_create(const prl_tree_t* tree, int32_t variant, int32_t delay, prl_solver_t** out_solver);
/* engine selection: AUTO = FUSED where applicable, else LEVELS.
 *   LEVELS keeps every per-node vector in HBM (any supported tree; node.reach_probs / ev / ev_br readable);
 *   FUSED walks each board subtree on chip and only keeps regrets / averages / root values in HBM. It takes 2-hole-card trees whose betting
 *   after every deal matches a registered subtree shape (fold / call / one raise size, up to five raises per round: 9 / 15 / 21 / 27 / 33 nodes):
 *   one deal (Flop5Holdem: the board pass) or several (LimitHoldem, DiscretizedNLHoldem with pot-sized raises, 3 + 1 + 1: one pass per street, shape
 *   and direction over its street instances, sharded over the first deal's outcomes). Since round 6 the subtrees of one street may have different
 *   shapes and an all-in call may be dealt out without further decisions (run-out chains). Trees it does not take (several raise sizes, 1-hole-card
 *   games) run on LEVELS; asking for FUSED on one of them is an error that says why. PRL_SF_ENGINE tells which engine a solver runs on. */
enum { PRL_ENGINE_AUTO = 0, PRL_ENGINE_LEVELS = 1, PRL_ENGINE_FUSED = 2 };
int32_t prl_solver_create_ex(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, prl_solver_t** out_solver);
/* Options. PRL_SOLVER_AVG_F32 (opt-in, NOT the reference's numerics): CFR+'s running average strategy of the board columns is STORED as
 * float32 -- read, widened, blended in float64 with the reference's weights (CFRPlus.py:65-87), rounded on the store. The reference's average
 * is float64 (a NumPy-2 promotion of its integer weights) and is 54 % of the board pass's HBM traffic; with float32 storage the average
 * differs from the reference's by one rounding per iteration (bench.py --avg-f32 reports the average-strategy exploitability of both).
 * Regrets, current strategies and the current-strategy exploitability history are unaffected. Fused engines (the single-deal board pass and, since
 * round 5, the per-street passes: the trunk's few columns stay float64), CFR+, no checkpoints; prl_solver_get(AVG) widens, prl_solver_get_cols(AVG) refuses. */
enum { PRL_SOLVER_AVG_F32 = 1 };
int32_t prl_solver_create_opts(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, int32_t flags, prl_solver_t** out_solver);
/* WEIGHTED BOARDS / SUIT ISOMORPHISM (round 5; an algorithmic extension the reference lacks -- its public tree lists every board, PokerRL/game/_/tree/
 * PublicTree.py:188-210, and cannot build 2-hole-card trees at all). `board_mult[i]` >= 1 says how many boards of the game the i-th listed board
 * stands for; the chance probability counts sum(board_mult) boards. With `symmetrize` != 0 the listed boards are representatives of the classes of
 * boards under the 24 suit permutations (board_mult = orbit sizes: Flop5Holdem has 134 459 classes for its 2 598 960 boards, ~28 GB of HBM: the WHOLE
 * game on one GPU) and the chance node's value of a hand is the mean over the hand's own suit orbit of the multiplicity-weighted sum:
 *     ev_chance[h] = (1 / |O(h)|) * sum_{h' in O(h)} sum_classes c  mult_c * v_c[h']        (O(h): the 4 / 6 / 12 hands h maps to under suit permutations)
 * which equals the full game's sum over all boards in exact arithmetic (suits do not enter the rules; regret matching keeps strategies suit-symmetric).
 * A class subtree works with reach scaled by mult_c (values are linear in the opponent's reach, strategies do not depend on the scale), so its
 * regrets are mult_c x those of one member board. Summation order: the canonical chance sum over the listed boards, then the orbit sum in ascending
 * hand index, then one correctly rounded division -- oracle/prl_oracle.c restates it (orc_set_board_weights, orc_set_symmetrize).
 * `symmetrize` is checked, not trusted (the orbit mean is only right for suit classes): every listed board must be the representative of its class
 * (the lexicographically smallest of its 24 relabellings, cards ascending), board_mult[i] the size of its orbit, and with symmetrize = 1 the list must
 * cover the game (multiplicities adding up to C(n_cards, k)); PRL_SYMMETRIZE_SUBSET accepts a subset of the classes (tests, partial solves). Any other
 * weighting of boards (importance-sampled boards ...) is symmetrize = 0. PRL_ERR_ARG otherwise, prl_last_error() naming the board.
 * Single-deal FUSED engine, one GPU (no exchange); flags as prl_solver_create_opts. */
enum { PRL_SYMMETRIZE_SUBSET = 2 };
int32_t prl_solver_create_weighted(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t flags, const int32_t* board_mult, int32_t symmetrize,
                                   prl_solver_t** out_solver);
/* Placement selection. The fused board pass streams within ~15 % of what HBM sustains and its speed depends on WHERE its arrays land
 * physically: solver objects of one process differ by up to 15 % and keep their speed for life ("Spread"). This entry
 * point does what a careful user would do by hand: it builds up to `n_candidates` solvers of the tree side by side (all alive until the
 * choice: a freed set's place would just be taken again; it stops early when HBM runs out), runs `probe_iters` steady-state iterations on
 * each (after three warm-up iterations), keeps the fastest, destroys the others and resets the winner -- the state handed back is exactly
 * that of prl_solver_create_opts. out_ms (may be NULL) receives the candidates' milliseconds per iteration (0 for candidates not built),
 * out_chosen (may be NULL) the index kept. Engines other than FUSED have nothing to choose: one solver is built, out_ms[0] = 0.
 * reference: the solver object of PokerRL/cfr/_CFRBase.py:18-62 (what CFRPlus(...) builds), which has no such concern on the CPU. */
int32_t prl_solver_create_placed(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t engine, int32_t flags, int32_t n_candidates,
                                 int32_t probe_iters, float* out_ms, int32_t* out_chosen, prl_solver_t** out_solver);
/* The same for weighted boards / suit classes (prl_solver_create_weighted's arguments, then prl_solver_create_placed's): the whole Flop5Holdem game is
 * 30 GB, three candidates fit side by side. */
int32_t prl_solver_create_weighted_placed(const prl_tree_t* tree, int32_t variant, int32_t delay, int32_t flags, const int32_t* board_mult, int32_t symmetrize,
                                          int32_t n_candidates, int32_t probe_iters, float* out_ms, int32_t* out_chosen, prl_solver_t** out_solver);
/* Sharded solve over `world_size` GPUs, one process per GPU: the tree handed to rank r holds the
 * r-th contiguous block of the global board list (every rank the same number of boards; see _ragged below); the pre-chance trunk is
 * replicated; regrets / averages of a board live on its owner only. The one exchange per EV pass is the pull-up of the
 * chance node's values (ValueFiller.py:76-78 is a plain sum over the chance children): every rank reduces its boards to
 * whole canonical summation units, `exchange` all-gathers them (rank-major), and every rank finishes the sum over all
 * units in global order -- so the result is bit-identical to the single-GPU solve of the whole board list, for any
 * world size. `exchange(user, local_dev, gathered_dev, bytes_per_rank)` must fill gathered_dev[rank * bytes_per_rank ...]
 * with rank's local_dev for every rank (e.g. ncclAllGather / torch.distributed.all_gather_into_tensor) and return 0
 * once gathered_dev is complete and visible to work submitted afterwards; the solver's own stream is idle while it
 * runs. FUSED engine only. Every solver call that evaluates the tree is collective (all ranks must make it). */
typedef int32_t (*prl_exchange_fn)(void* user, const void* local_dev, void* gathered_dev, uint64_t bytes_per_rank);
int32_t prl_solver_create_sharded(const prl_tree_t* local_tree, int32_t variant, int32_t delay, int32_t world_size, int32_t rank,
                                  prl_exchange_fn exchange, void* user, prl_solver_t** out_solver);
/* Ragged shards (a board list that does not divide by the world size, e.g. all C(52,5) = 2 598 960 flops of Flop5Holdem on 8 GPUs):
 * every rank before the last holds `shard_boards` boards, the last one the remaining total_boards - (world_size-1)*shard_boards
 * (> 0, <= shard_boards). The exchange moves whole canonical summation units of the highest level shard_boards is a multiple of
 * (1024-board groups, 32-board blocks, else single boards), bytes_per_rank is the same on every rank (the last rank pads with
 * zeros that are never summed), and the result is still bit-identical to the single-GPU solve of the whole list. */
int32_t prl_solver_create_sharded_ragged(const prl_tree_t* local_tree, int32_t variant, int32_t delay, int32_t world_size, int32_t rank,
                                         int64_t shard_boards, int64_t total_boards, prl_exchange_fn exchange, void* user,
                                         prl_solver_t** out_solver);
/* The same sharded solve with the exchange inside the library: ncclAllGather (RCCL over xGMI) enqueued on the solver's own HIP stream,
 * no callback, no host synchronisation per pass. Rank 0 draws a 128-byte communicator id (prl_rccl_unique_id = ncclGetUniqueId) and
 * hands it to every rank by any means (a file, MPI, torch.distributed.broadcast ...); every rank then calls this with the device of its
 * GPU current (prl_set_device). Collective: returns once all ranks have joined. shard_boards = 0: equal shards (every rank's tree holds
 * the same number of first-deal outcomes); else the ragged geometry of prl_solver_create_sharded_ragged. RCCL is bound at run time
 * (dlopen; the one already in the process if PyTorch-ROCm is loaded): single-GPU users never load it. */
int32_t prl_rccl_unique_id(void* out_id128);
/* The RCCL the library binds at run time: the path of the file its symbols come from, or -- PRL_ERR_UNSUPPORTED -- why there is none.
 * Order: PRL_RCCL_LIB=<path> if set, the RCCL already loaded in the process (PyTorch-ROCm's), the loader's search path, /opt/rocm/lib. */
int32_t prl_rccl_info(char* out, int32_t n);
int32_t prl_solver_create_sharded_rccl(const prl_tree_t* local_tree, int32_t variant, int32_t delay, int32_t world_size, int32_t rank,
                                       const void* unique_id128, int64_t shard_boards, int64_t total_boards, prl_solver_t** out_solver);
/* Checkpoint / resume (the reference's CFR has none): the solver's persistent state -- iteration
 * counter, regrets, average strategy (+ sum), current trunk strategy with its dtype flags, exploitability history -- as one
 * opaque host blob. load_state needs a solver created on the same tree with the same variant / delay / engine; a resumed
 * run continues bit-identically. Sharded solves: every rank saves / loads its own blob. */
int32_t prl_solver_state_size(prl_solver_t* solver, uint64_t* out_bytes);
int32_t prl_solver_save_state(prl_solver_t* solver, void* out_blob, uint64_t bytes);
int32_t prl_solver_load_state(prl_solver_t* solver, const void* blob, uint64_t bytes);
/* Stream-ordered exchange: by default the solver drains its stream before it calls `exchange` and expects the gathered
 * buffer to be complete when the callback returns. A callback that ENQUEUES the collective on the solver's own stream
 * (prl_solver_get_stream; e.g. ncclAllGather(..., stream), or torch.distributed under torch.cuda.ExternalStream) needs
 * neither: prl_solver_set_exchange_async(solver, 1) removes the host synchronisation on both sides. */
int32_t prl_solver_get_stream(prl_solver_t* solver, void** out_hip_stream);
int32_t prl_solver_set_exchange_async(prl_solver_t* solver, int32_t stream_ordered);
/* The canonical chance-node sum on its own (host buffers in / out, device kernels inside): values [n_boards][2][R] ->
 * out [2][R]. world_size > 1 replays the sharded path on one device (every rank's partial units at the level the shard
 * size allows, rank-major gather, finish); the result must not depend on world_size. n_boards % world_size == 0. */
int32_t prl_chance_sum_host(const float* board_values, int32_t n_boards, int32_t R, int32_t world_size, float* out);
/* the same with ragged shards: ranks before the last hold shard_boards boards, the last one the rest */
int32_t prl_chance_sum_host_ragged(const float* board_values, int32_t n_boards, int32_t R, int32_t world_size, int32_t shard_boards, float* out);
void prl_solver_destroy(prl_solver_t* solver);
int32_t prl_solver_reset(prl_solver_t* solver);                        /* _CFRBase.reset            :110-120 */
int32_t prl_solver_iteration(prl_solver_t* solver);                    /* _CFRBase.iteration        :122-134 (w/o avg eval) */
int32_t prl_solver_iterations(prl_solver_t* solver, int32_t n);
/* n iterations of MANY independent solvers in one launch, one workgroup (one CU) per solver: the way to fill a GPU with
 * Leduc-sized trees (a sweep over games / stack sizes / bet sets; BASELINE configs 1-2). Every solver must be a small
 * 1-hole-card tree on the LEVELS engine (n_nodes * range_size <= 32768); each advances exactly as prl_solver_iterations(s, n). */
int32_t prl_solver_iterations_many(prl_solver_t** solvers, int32_t n_solvers, int32_t n);
int32_t prl_solver_eval_avg(prl_solver_t* solver, float* out_expl2);   /* _evaluate_avg_strats      :218-262 */
int32_t prl_solver_fill_uniform(prl_solver_t* solver);                 /* PublicTree.fill_uniform_random     */
int32_t prl_solver_set_strategy(prl_solver_t* solver, const void* strategy_cols, int32_t is_f64); /* fill_with_agent_policy */
/* float64 data with a per-NODE dtype flag (NULL = all float64): the reference mixes float64 strategies (uniform fill, averages) with
 * float32 ones (after regret matching) in one tree and the arithmetic dtype follows the node's strategy. LEVELS engine. */
int32_t prl_solver_set_strategy_mixed(prl_solver_t* solver, const void* strategy_cols, int32_t is_f64, const uint8_t* node_is_f64);
int32_t prl_solver_update_reach(prl_solver_t* solver);                 /* PublicTree.update_reach_probs      */
int32_t prl_solver_compute_ev(prl_solver_t* solver);                   /* PublicTree.compute_ev              */
int32_t prl_solver_exploitability(prl_solver_t* solver, float* out_expl2); /* root.exploitability, raw float32 per seat */
int32_t prl_solver_sync(prl_solver_t* solver);
/* runs n iterations between two HIP events recorded on the solver's stream; elapsed device time in milliseconds */
int32_t prl_solver_time_iterations(prl_solver_t* solver, int32_t n, float* out_ms);
/* same, and every launch of the FUSED engine's board-pass kernel is bracketed by its own pair of events: their summed
 * duration and count (the per-kernel figure bench.py's roofline is computed from; out_pass_* may be NULL) */
int32_t prl_solver_time_iterations_ex(prl_solver_t* solver, int32_t n, float* out_ms, float* out_pass_ms, int32_t* out_n_pass);

/* n x (prl_solver_update_reach + prl_solver_compute_ev) of the strategy the solver holds -- one exact best-response evaluation each
 * (LocalBRMaster.py:67-80 without the agent query) -- bracketed by HIP events like prl_solver_time_iterations_ex */
int32_t prl_solver_time_evaluations(prl_solver_t* solver, int32_t n, float* out_ms, float* out_pass_ms, int32_t* out_n_pass);

enum {
    PRL_SF_REACH = 0,        /* float32 [n_nodes][2][R]  node.reach_probs                      */
    PRL_SF_EV = 1,           /* float32 [n_nodes][2][R]  node.ev                               */
    PRL_SF_EV_BR = 2,        /* float32 [n_nodes][2][R]  node.ev_br                            */
    PRL_SF_STRATEGY = 3,     /* float64 [n_cols][R]      node.strategy.T                       */
    PRL_SF_STRAT_F64 = 4,    /* uint8   [n_nodes]        1 where the strategy dtype is float64 */
    PRL_SF_REGRET = 5,       /* float32 [n_cols][R]      node.data["regret"].T                 */
    PRL_SF_AVG = 6,          /* float64 [n_cols][R]      node.data["avg_strat"].T              */
    PRL_SF_AVG_F64 = 7,      /* uint8   [n_nodes]                                              */
    PRL_SF_AVG_SUM = 8,      /* float32 [n_cols][R]      node.data["avg_strat_sum"].T          */
    PRL_SF_BR_IDX = 9,       /* int32   [n_nodes][R]     br_a_idx_in_child_arr_for_each_hand   */
    PRL_SF_EXPL_HISTORY = 10, /* float32 [iter+1][2]     current-strategy exploitability after every iteration */
    PRL_SF_ITER = 11,        /* int32                    iteration counter                     */
    PRL_SF_CONSTANTS = 12,   /* float32 [2]              chance probability, equity constant   */
    PRL_SF_BYTES_ALLOCATED = 13, /* int64                HBM bytes held by the solver          */
    PRL_SF_ENGINE = 14,      /* int32                    PRL_ENGINE_LEVELS or PRL_ENGINE_FUSED  */
    PRL_SF_GRAPH_REPLAY = 15, /* int32                   1 if iterations are replays of a captured hipGraph (LEVELS engine) */
    PRL_SF_EXCHANGES = 17,   /* int64                    all-gathers of a sharded solve so far (0 for an unsharded one) */
    PRL_SF_VMM_RANGES = 18,  /* int64 [2]                arrays backed by shuffled virtual-memory ranges (the default of a sharded solve, PRL_VMM_SHUFFLE_MB
                                                         elsewhere) and the bytes they hold; {0, 0}: plain hipMalloc */
    PRL_SF_EXPLICIT_STRATEGY = 16 /* int32               fused engines: -1 strategy follows regrets / uniform fill, 0 an explicit float32
                                                         strategy is loaded (prl_solver_set_strategy), 1 an explicit float64 one; LEVELS: -1 */
};
int32_t prl_solver_get(prl_solver_t* solver, int32_t field, void* out);
/* The strategy of every decision node from a DEVICE array -- PublicTree.fill_with_agent_policy (StrategyFiller.py:88-116) without the host in between:
 * d_probs float32 [n_decision_nodes][range_size][n_actions] in HBM (decision nodes in node order = the order the reference visits them in; the agent's
 * probabilities over ALL action ints, e.g. the output tensor of one batched network forward); column first_col[node] + j takes
 * d_probs[k][:][col_action[first_col[node] + j]]. Equal, bit for bit, to prl_solver_set_strategy with the same float32 values gathered on the host
 * (float32 strategy: float32 arithmetic at every node); the fused engines scatter into their own storage (sorted board columns: through the staging
 * buffer, a chunk of boards at a time). `tree`: the tree the solver was created on. Ends with the reach pass, like prl_solver_set_strategy. The caller
 * synchronises whatever produced d_probs before the call (the solver works on its own stream). */
int32_t prl_solver_set_strategy_device(prl_solver_t* solver, const prl_tree_t* tree, const float* d_probs, int32_t n_actions);
/* n_cols action columns of a per-action-column field (REGRET, AVG, AVG_SUM) starting at flat-tree column col_begin: what lets a caller
 * stream a 20-60 GB array through a small host buffer (hashing, checkpoints to disk). Trees on the per-street engine keep their columns
 * in another order internally: the call gathers them (one copy per run of columns that are neighbours in both orders). */
int32_t prl_solver_get_cols(prl_solver_t* solver, int32_t field, int64_t col_begin, int64_t n_cols, void* out);

/* ---------------------------------------------------------------------------------------------------------------- */
/* 6. Local best response (LBR): the check-down equity of LBR's hand against agent ranges.                           */
/*    replaces  PokerRL/eval/lbr/LocalLBRWorker.py:379-512 (_LBRRolloutManager.__init__/_build_eq_vecs,               */
/*              get_lbr_checkdown_equity, _calc_eq) incl. PokerRange.get_card_probs / set_cards_to_zero_prob /         */
/*              normalize (PokerRange.py:26-84) as they are used there. float32, NumPy's summation order: bit-exact.   */
/*    board_dealt: the n_dealt board cards on the table (1d cards, deal order); lbr_hand: n_hole_cards 1d cards;       */
/*    ranges: [n_q][R] candidate agent ranges (the current one, and the one after "agent does not fold" per raise);    */
/*    out_wp[n_q]: P(LBR wins the check-down) per range. Any number of board cards to come: hold'em before the flop     */
/*    enumerates all C(50, 5) = 2 118 760 run-outs (LocalLBRWorker.py:388-425), about a second per decision.           */
/* ---------------------------------------------------------------------------------------------------------------- */
int32_t prl_lbr_checkdown_equity(const PrlRules* rules, const int8_t* board_dealt, int32_t n_dealt, const int8_t* lbr_hand,
                                 const float* ranges, int32_t n_q, float* out_wp);

/* Device-resident batched LBR (BASELINE config 5): n_envs hands of LocalLBRWorker.run (LocalLBRWorker.py:61-308) against a
 * synthetic tabular agent, each played start to finish by one workgroup: betting engine (PokerEnv._step), dealing, the
 * agent's PokerRange, LBR's look-ahead with check-down equities, payout. Same float32 arithmetic as the host worker.
 *   lbr_game / agent_game: the same game with LBR's and the agent's bet sizes; cards: [n_envs][2*n_hole + n_board] 1d cards
 *   (seat 0's hole cards, seat 1's, then the board in deal order = the top of the reference's shuffled deck);
 *   check_to_round: -1 = None; agent_kind 0 uniform, 1 seeded hash policy (csrc/prl_lbr_batch.hip); episode_base: hand e
 *   draws the agent's actions as episode episode_base + e + 1; reward_scalar / ev_normalizer: env.REWARD_SCALAR, EV_NORMALIZER.
 *   out_winnings[n_envs] float32 = reward[lbr_seat] * REWARD_SCALAR * EV_NORMALIZER; out_stats4: env steps, LBR look-ahead
 *   decisions, (range, board) equities, agent actions; out_device_ms: kernel time (HIP events, summed over the rounds below).
 *   LBR decisions with at most two board cards to come are evaluated inside the kernel. Decisions with more -- hold'em BEFORE THE FLOP, i.e.
 *   lbr_check_to_round = None, the reference's default (LBRArgs.py:18; enumeration LocalLBRWorker.py:388-425) -- need C(50, 5) boards per candidate
 *   range, which are a function of the public history and LBR's hand only: they are cached per (history key, LBR hand) in HBM. A hand that misses
 *   files a request and stops; the call computes the requested equities with prl_lbr_checkdown_equity (the host worker's own call at that decision),
 *   fills the cache and plays the stopped hands again from the start (counter-based decks and draws: a replay reaches the same decision with the
 *   same ranges) -- rounds of launches until no hand waits; the counters include the replays. Same per-hand winnings as the host worker. */
int32_t prl_lbr_batch_run(const PrlGame* lbr_game, const PrlGame* agent_game, const PrlRules* rules, int32_t n_envs, int32_t agent_seat,
                          int32_t check_to_round, int32_t agent_kind, uint32_t agent_seed, uint32_t episode_base, double reward_scalar,
                          double ev_normalizer, const int8_t* cards, float* out_winnings, uint64_t* out_stats4, float* out_device_ms);

/* Batched head-to-head (PokerRL/eval/head_to_head/LocalHead2HeadMaster.py:82-126 on the batched env): n_envs
 * hands between two of the library's synthetic agents, one GPU lane per hand (betting engine, the acting agent's row of its
 * policy, action draw, dealing, payout with the hand ranks). ref_*: the agent whose winnings are reported, sitting in ref_seat;
 * opp_*: its opponent; kinds / seeds / episode_base / cards / reward_scalar / ev_normalizer as in prl_lbr_batch_run; every
 * agent counts its own action draws per hand. out_winnings[n_envs] float32 = reward[ref_seat] * REWARD_SCALAR * EV_NORMALIZER;
 * out_stats2: env steps, showdowns. Same game for both agents (the reference assumes one action space). */
int32_t prl_h2h_batch_run(const PrlGame* game, const PrlRules* rules, int32_t n_envs, int32_t ref_seat, int32_t ref_kind, uint32_t ref_seed,
                          int32_t opp_kind, uint32_t opp_seed, uint32_t episode_base, double reward_scalar, double ev_normalizer,
                          const int8_t* cards, float* out_winnings, uint64_t* out_stats2, float* out_device_ms);

/* Tabular agents for the batched evaluators (agent kind 2): a policy addressed by the public state, resident in HBM. This is what the reference's
 * evaluators get from EvalAgentBase.get_a_probs_for_each_hand / get_action (PokerRL/rl/base_cls/EvalAgentBase.py:35-62; used by
 * LocalLBRWorker.py:120-160,241-281 and LocalHead2HeadMaster.py:100-118) when the agent is tabular -- e.g. the average strategy a CFR solver left
 * in its PublicTree (pokerrl_amd/rl/tabular_agent.py builds the arrays from a tree and is the same policy as a host EvalAgent).
 *   keys / rows [capacity]: open-addressed table (linear probing, capacity a power of two > n_rows, key 0 = empty slot) from a state's 64-bit key to
 *     its row; the key's low word is the hash chain of csrc/prl_lbr_batch.hip (lbrb_state_key: round, pot, bets, stacks, seat to act, board) under
 *     key_seed, the high word the same chain under key_seed ^ 0x5BD1E995; slot of first probe: (lo ^ hi * 0x9E3779B1) & (capacity - 1);
 *   probs [n_rows][n_actions][range_size] float32: P(action | hand), 0 for actions that are not legal in the row's state.
 * A state the table does not hold (LBR stepped outside the agent's tree) plays uniformly over the legal actions. Returns NULL on error. */
typedef struct PrlPolicyTable prl_policy_table_t;
prl_policy_table_t* prl_policy_table_create(const uint64_t* keys, const int32_t* rows, uint32_t capacity, const float* probs, int32_t n_rows,
                                            int32_t n_actions, int32_t range_size, uint32_t key_seed);
void prl_policy_table_destroy(prl_policy_table_t* table);
/* n look-ups on the device (host arrays in and out): out_row[i] = the row of history key (key_lo[i], key_hi[i]) or -1, out_prob[i] = P(action[i] | hand[i])
 * of that row (0 without one) -- what the batched engines read, so that a table can be verified where it lives. */
int32_t prl_policy_table_probe(const prl_policy_table_t* table, int32_t n, const uint32_t* key_lo, const uint32_t* key_hi, const int32_t* action,
                               const int32_t* hand, int32_t* out_row, float* out_prob);

/* The table of a SOLVER's average strategy, built where the strategy lives (no host copy of the columns; the reference's evaluators are pointed at the
 * trained agent: PokerRL/eval/lbr/LocalLBRWorker.py:316-374, PokerRL/rl/base_cls/EvalAgentBase.py:39-44). `tree` is the tree the solver was created on.
 * One row per decision node, rows in node order; the row's history key is the chain of prl_policy_table_create over the states on the node's path
 * (the env replayed along the tree). Every engine: LEVELS; the single-deal FUSED engine (its board columns are expanded from sorted storage to hand
 * order on the device, a chunk of boards at a time); the per-street FUSED engine (its internal column order, mixed street shapes included: no decision
 * below an all-in call, so run-out chains have no rows). The average is what prl_solver_get(PRL_SF_AVG) returns, rounded to float32.
 * A solver made by prl_solver_create_weighted with `symmetrize` (suit classes) yields a SUIT-CANONICAL table: rows exist for the class representatives
 * only; the evaluators relabel the dealt board to its class representative -- the lexicographically smallest of its 24 relabellings, cards ascending,
 * taking the FIRST permutation in lexicographic order of (p[0], p[1], p[2], p[3]) that attains it (csrc/prl_policy.h: prl_suit_canon) -- hash THAT
 * board into the history key and read the table at the hand relabelled by the same permutation: the whole game's 2 598 960 boards through 134 459 x 6
 * rows (12.8 GB). Errors: PRL_ERR_STATE before the first average exists, PRL_ERR_UNSUPPORTED for a rank of a sharded solve. */
int32_t prl_policy_table_from_solver(prl_solver_t* solver, const prl_tree_t* tree, uint32_t key_seed, prl_policy_table_t** out_table);
/* out6: n_rows, n_actions, range_size, capacity of the key table, suit_canon (0 / 1), key_seed */
int32_t prl_policy_table_info(const prl_policy_table_t* table, int64_t* out6);
/* the key table as it lives on the device: keys[capacity] (0 = empty), rows[capacity] */
int32_t prl_policy_table_export_keys(const prl_policy_table_t* table, uint64_t* out_keys, int32_t* out_rows);
/* rows [row_begin, row_begin + n_rows) of the table: out float32 [n_rows][n_actions][range_size] (hands in the table's own labelling) */
int32_t prl_policy_table_get_rows(const prl_policy_table_t* table, int32_t row_begin, int32_t n_rows, float* out);
/* host helper (tests, host twins of the evaluators): the canonical form of n boards of k cards each (prl_suit_canon) and the permutation numbers taken */
int32_t prl_suit_canon_boards(const int8_t* boards, int32_t n, int32_t k, int32_t n_suits, int8_t* out_boards, int32_t* out_perm);

/* prl_lbr_batch_run against a tabular agent: `table` is the agent's policy; agent_seed drives its action draws as for the synthetic agents. */
int32_t prl_lbr_batch_run_table(const PrlGame* lbr_game, const PrlGame* agent_game, const PrlRules* rules, int32_t n_envs, int32_t agent_seat,
                                int32_t check_to_round, const prl_policy_table_t* table, uint32_t agent_seed, uint32_t episode_base, double reward_scalar,
                                double ev_normalizer, const int8_t* cards, float* out_winnings, uint64_t* out_stats4, float* out_device_ms);

/* prl_h2h_batch_run with tabular agents: a non-NULL table makes that side a tabular agent (its kind argument is ignored), NULL keeps the synthetic one. */
int32_t prl_h2h_batch_run_tables(const PrlGame* game, const PrlRules* rules, int32_t n_envs, int32_t ref_seat, int32_t ref_kind, uint32_t ref_seed,
                                 const prl_policy_table_t* ref_table, int32_t opp_kind, uint32_t opp_seed, const prl_policy_table_t* opp_table,
                                 uint32_t episode_base, double reward_scalar, double ev_normalizer, const int8_t* cards, float* out_winnings,
                                 uint64_t* out_stats2, float* out_device_ms);

/* Counter-based decks for the batched engines (no reference counterpart: the reference shuffles with np.random, one hand at a
 * time): hand i gets the first n_deal cards of a Fisher-Yates shuffle of 0..n_cards_in_deck-1 keyed by (seed, first_hand + i),
 * so any split of the hands over GPUs deals the same cards. out_cards: host int8 [n_hands][n_deal] (n_deal <= 16). */
int32_t prl_deal_decks(int32_t n_hands, int32_t n_cards_in_deck, int32_t n_deal, uint64_t seed, uint64_t first_hand, int8_t* out_cards);

#ifdef __cplusplus
}
#endif
#endif /* POKERRL_HIP_H */
