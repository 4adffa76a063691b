// C-ABI entry points that are host code by nature: index LUT generators, the scalar hand evaluator, public-tree
// construction. (The reference's equivalents are host C++ too: lib_luts.so, get_hand_rank_52_holdem.)
#include <string.h>

#include <string>

#include "prl_cards.h"
#include "prl_defs.h"
#include "prl_env.h"
#include "prl_handeval.h"
#include "prl_host.h"
#include "prl_tree.h"

static thread_local std::string g_last_error;

void prl_set_error(const std::string& msg) { g_last_error = msg; }

extern "C" {

const char* prl_last_error(void) { return g_last_error.c_str(); }

// ---- legacy lib_luts.so symbols (CppLUT.py:38-47,73-94) ---------------------------------------------------------------
int8_t get_1d_card(const int8_t* card_2d) { return (int8_t)prl_card_1d(card_2d[0], card_2d[1], 4); }

void get_2d_card(int8_t card_1d, int8_t* out_card_2d) {
    out_card_2d[0] = (int8_t)prl_card_rank(card_1d, 4);
    out_card_2d[1] = (int8_t)prl_card_suit(card_1d, 4);
}

void get_idx_2_hole_card_lut(int8_t** out) {
    int idx = 0;
    for (int c1 = 0; c1 < 52; ++c1)
        for (int c2 = c1 + 1; c2 < 52; ++c2) {
            out[idx][0] = (int8_t)c1;
            out[idx][1] = (int8_t)c2;
            idx++;
        }
}

void get_hole_card_2_idx_lut(int16_t** out) {  // only the upper triangle is written (the caller pre-fills -2)
    for (int c1 = 0; c1 < 52; ++c1)
        for (int c2 = c1 + 1; c2 < 52; ++c2) out[c1][c2] = (int16_t)prl_range_idx_2(c1, c2, 52);
}

// The three board-index tables of lib_luts.so (CppLUT.py:49-71). The reference never calls them -- its binary's versions crash
//  -- but CppLibHoldemLuts.__init__ binds their argtypes (CppLUT.py:27-34), so a drop-in must EXPORT them
// or the wrapper's constructor raises. Implemented with the obvious contract: row i = the i-th k-card board of the 52-card
// deck in ascending lexicographic order of ascending 1d cards (22 100 / 270 725 / 2 598 960 rows, caller-allocated).
static void prl_fill_board_lut(int8_t** out, int k) {
    int c[5] = {0, 1, 2, 3, 4};
    for (size_t row = 0;; ++row) {
        for (int i = 0; i < k; ++i) out[row][i] = (int8_t)c[i];
        int i = k - 1;
        while (i >= 0 && c[i] == 52 - k + i) --i;
        if (i < 0) break;
        ++c[i];
        for (int j = i + 1; j < k; ++j) c[j] = c[j - 1] + 1;
    }
}
void get_idx_2_flop_lut(int8_t** out) { prl_fill_board_lut(out, 3); }
void get_idx_2_turn_lut(int8_t** out) { prl_fill_board_lut(out, 4); }
void get_idx_2_river_lut(int8_t** out) { prl_fill_board_lut(out, 5); }

int32_t get_hand_rank_52_holdem(int8_t** hand_2d, int8_t** board_2d) {
    uint32_t s[4] = {0, 0, 0, 0};
    for (int i = 0; i < 2; ++i) s[hand_2d[i][1] & 3] |= 1u << hand_2d[i][0];
    for (int i = 0; i < 5; ++i) s[board_2d[i][1] & 3] |= 1u << board_2d[i][0];
    return prl_rank7_masks(s[0], s[1], s[2], s[3]);
}

// ---- flat LUT API ----------------------------------------------------------------------------------------------------
static int check_rules(const PrlRules* r) {
    if (!r) { prl_set_error("rules is NULL"); return PRL_ERR_ARG; }
    if (r->n_hole_cards != 1 && r->n_hole_cards != 2) { prl_set_error("n_hole_cards must be 1 or 2"); return PRL_ERR_UNSUPPORTED; }
    if (r->n_cards != r->n_ranks * r->n_suits || r->n_cards < 2 || r->n_cards > 64) { prl_set_error("bad deck"); return PRL_ERR_ARG; }
    if (r->range_size != (int)prl_comb(r->n_cards, r->n_hole_cards)) { prl_set_error("range_size != C(n_cards, n_hole_cards)"); return PRL_ERR_ARG; }
    return PRL_OK;
}

int32_t prl_lut_idx_2_hole_cards(const PrlRules* rules, int8_t* out) {
    int e = check_rules(rules);
    if (e) return e;
    for (int i = 0; i < rules->range_size; ++i) {
        int c1, c2;
        prl_hand_cards(*rules, i, &c1, &c2);
        if (rules->n_hole_cards == 1) out[i] = (int8_t)c1;
        else { out[2 * i] = (int8_t)c1; out[2 * i + 1] = (int8_t)c2; }
    }
    return PRL_OK;
}

int32_t prl_lut_hole_cards_2_idx(const PrlRules* rules, int16_t* out) {
    int e = check_rules(rules);
    if (e) return e;
    const int n = rules->n_cards;
    if (rules->n_hole_cards == 1) {  // look_up_table.py:153-155: [c1] -> idx, shape [n, 1]
        for (int c = 0; c < n; ++c) out[c] = (int16_t)c;
        return PRL_OK;
    }
    for (int i = 0; i < n * n; ++i) out[i] = -2;
    for (int c1 = 0; c1 < n; ++c1)
        for (int c2 = c1 + 1; c2 < n; ++c2) out[c1 * n + c2] = (int16_t)prl_range_idx_2(c1, c2, n);
    return PRL_OK;
}

int32_t prl_lut_card_in_what_range_idxs(const PrlRules* rules, int32_t* out) {
    int e = check_rules(rules);
    if (e) return e;
    const int n = rules->n_cards;
    if (rules->n_hole_cards == 1) {  // look_up_table.py:157-158
        for (int c = 0; c < n; ++c) out[c] = c;
        return PRL_OK;
    }
    for (int c = 0; c < n; ++c) {  // ascending range idx, look_up_table.py:121-134
        int k = 0;
        for (int a = 0; a < c; ++a) out[c * (n - 1) + k++] = prl_range_idx_2(a, c, n);
        for (int b = c + 1; b < n; ++b) out[c * (n - 1) + k++] = prl_range_idx_2(c, b, n);
    }
    return PRL_OK;
}

int32_t prl_hand_rank_7(const int8_t* board_1d, int8_t c1, int8_t c2) { return prl_rank7_cards_52(board_1d, c1, c2); }

// ---- public tree -----------------------------------------------------------------------------------------------------
struct prl_tree {
    PrlFlatTree t;
};

int32_t prl_tree_build(const PrlGame* game, const PrlRules* rules, const int8_t* boards, int32_t n_boards,
                       int32_t board_len, prl_tree_t** out_tree) {
    return prl_tree_build_partial(game, rules, boards, n_boards, board_len, -1, out_tree);
}

int32_t prl_tree_build_partial(const PrlGame* game, const PrlRules* rules, const int8_t* boards, int32_t n_boards, int32_t board_len,
                               int32_t stop_at_round, prl_tree_t** out_tree) {
    if (!game || !rules || !boards || !out_tree) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    int e = check_rules(rules);
    if (e) return e;
    prl_tree* h = new prl_tree();
    e = prl_build_flat_tree(*game, *rules, boards, n_boards, board_len, &h->t, stop_at_round);
    if (e) {
        prl_set_error(h->t.error);
        delete h;
        return e;
    }
    *out_tree = h;
    return PRL_OK;
}

void prl_tree_destroy(prl_tree_t* tree) { delete tree; }

const PrlFlatTree* prl_tree_flat(const prl_tree_t* tree) { return tree ? &tree->t : nullptr; }

int32_t prl_tree_get_boards(const prl_tree_t* tree, int8_t* out) {
    if (!tree || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    memcpy(out, tree->t.boards.data(), tree->t.boards.size());
    return PRL_OK;
}

int32_t prl_tree_info(const prl_tree_t* tree, int32_t* out) {
    if (!tree || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    const PrlFlatTree& t = tree->t;
    int n_dec = 0, n_term = 0;
    for (int i = 0; i < t.n_nodes; ++i) {
        n_dec += t.kind[i] == PRL_NODE_DECISION;
        n_term += t.kind[i] >= PRL_NODE_TERM_FOLD;
    }
    out[PRL_TI_N_NODES] = t.n_nodes;
    out[PRL_TI_N_COLS] = t.n_cols;
    out[PRL_TI_N_BOARDS] = t.n_boards;
    out[PRL_TI_BOARD_LEN] = t.board_len;
    out[PRL_TI_N_LEVELS] = t.n_levels;
    out[PRL_TI_RANGE_SIZE] = t.rules.range_size;
    out[PRL_TI_N_DECISION] = n_dec;
    out[PRL_TI_N_TERMINAL] = n_term;
    return PRL_OK;
}

int32_t prl_tree_get(const prl_tree_t* tree, int32_t field, int32_t* out) {
    if (!tree || !out) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    const PrlFlatTree& t = tree->t;
    const std::vector<int32_t>* v = nullptr;
    switch (field) {
        case PRL_TF_KIND: v = &t.kind; break;
        case PRL_TF_ACTOR: v = &t.actor; break;
        case PRL_TF_PARENT: v = &t.parent; break;
        case PRL_TF_CHILD_IDX: v = &t.child_idx; break;
        case PRL_TF_ACTION: v = &t.action; break;
        case PRL_TF_ACTED_LAST: v = &t.acted_last; break;
        case PRL_TF_ROUND: v = &t.round; break;
        case PRL_TF_BOARD_ID: v = &t.board_id; break;
        case PRL_TF_MAIN_POT: v = &t.main_pot; break;
        case PRL_TF_DEPTH: v = &t.depth; break;
        case PRL_TF_N_CHILDREN: v = &t.n_children; break;
        case PRL_TF_FIRST_COL: v = &t.first_col; break;
        case PRL_TF_SUBTREE_SIZE: v = &t.subtree_size; break;
        case PRL_TF_CHILD_START: v = &t.child_start; break;
        case PRL_TF_CHILD_LIST: v = &t.child_list; break;
        case PRL_TF_COL_ACTION: v = &t.col_action; break;
        case PRL_TF_COL_NODE: v = &t.col_node; break;
        case PRL_TF_LEVEL_START: v = &t.level_start; break;
        case PRL_TF_LEVEL_NODES: v = &t.level_nodes; break;
        default: prl_set_error("unknown tree field"); return PRL_ERR_ARG;
    }
    if (!v->empty()) memcpy(out, v->data(), v->size() * sizeof(int32_t));
    return PRL_OK;
}

// ---- heads-up betting engine on the host -----------------------------------------------------------------------------
int32_t prl_env_reset_host(const PrlGame* game, PrlEnvState* state) {
    if (!game || !state) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    prl_env_reset(*game, *state);
    return PRL_OK;
}

int32_t prl_env_step_host(const PrlGame* game, PrlEnvState* state, int32_t action_int, PrlStepInfo* out_info) {
    if (!game || !state || !out_info) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    int n_act = game->game_type == PRL_GAME_DISCRETIZED ? game->n_bet_sizes + 2 : 3;
    if (action_int < 0 || action_int >= n_act) { prl_set_error("action out of range"); return PRL_ERR_ARG; }
    prl_env_step(*game, *state, action_int, out_info);
    return PRL_OK;
}

int32_t prl_env_step_processed_host(const PrlGame* game, PrlEnvState* state, int32_t type, int32_t amount, PrlStepInfo* out_info) {
    if (!game || !state || !out_info) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (type < 0 || type > 2) { prl_set_error("action type out of range"); return PRL_ERR_ARG; }
    prl_env_step_processed(*game, *state, type, amount, out_info);
    return PRL_OK;
}

int32_t prl_env_apply_action_host(const PrlGame* game, PrlEnvState* state, int32_t action_int, int32_t is_processed, int32_t type, int32_t amount,
                                  PrlStepInfo* out_info) {
    if (!game || !state || !out_info) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    if (is_processed) {
        if (type < 0 || type > 2) { prl_set_error("action type out of range"); return PRL_ERR_ARG; }
    } else {
        const int n_act = game->game_type == PRL_GAME_DISCRETIZED ? game->n_bet_sizes + 2 : 3;
        if (action_int < 0 || action_int >= n_act) { prl_set_error("action out of range"); return PRL_ERR_ARG; }
        prl_adjust_action(*game, *state, action_int, &type, &amount);
    }
    *out_info = PrlStepInfo();
    int ft, fa;
    prl_env_apply_action(*game, *state, type, amount, &ft, &fa);
    out_info->fixed_type = ft;
    out_info->fixed_amount = fa;
    return PRL_OK;
}

int32_t prl_env_legal_actions_host(const PrlGame* game, const PrlEnvState* state, int32_t* out_actions, int32_t* out_n) {
    if (!game || !state || !out_actions || !out_n) { prl_set_error("NULL argument"); return PRL_ERR_ARG; }
    *out_n = prl_legal_actions(*game, *state, out_actions);
    return PRL_OK;
}

int32_t prl_env_fraction_of_pot_raise_host(const PrlEnvState* state, double fraction, int32_t seat, int32_t* out_total) {
    if (!state || !out_total || seat < 0 || seat > 1) { prl_set_error("bad argument"); return PRL_ERR_ARG; }
    *out_total = prl_fraction_of_pot_raise(*state, fraction, seat);
    return PRL_OK;
}

}  // extern "C"
