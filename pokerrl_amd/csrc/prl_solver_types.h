// Device-resident public tree + solver state (struct-of-arrays in HBM), passed BY VALUE to the kernels.
//
// Layout (DESIGN.md "data layout"):
//   per-node vectors   reach / ev / ev_br : float32 [n_nodes][2 seats][R]      hand index fastest -> coalesced
//   per-action columns strategy (float64), regret (float32), avg (float64), avg_sum (float32) : [n_cols][R]
//       column id = first_col[node] + a  (children order), i.e. the reference's strategy[:, a] (nodes.py:35) is one
//       contiguous R-vector; a node's A columns are adjacent; a board subtree's columns are one contiguous block.
//   strategy is kept in float64 storage with a per-node dtype flag because the reference mixes float64 strategies
//   (uniform fill, average strategy) with float32 ones (after regret matching) and the arithmetic dtype follows the
//   strategy's dtype (SURVEY.md 8a dtype ledger); when the flag is 0 the stored values are exactly float32.
#pragma once
#include "prl_defs.h"

// iteration state of a graph-replayed iteration (prl_tree_kernels.hip: *_dev kernels)
struct PrlIterDev { int32_t iter, mode; double m_old, m_new; float* hist; };

#define PRL_CHANCE_BLOCK 32   // canonical chance-sum order: blocks of 32 children, groups of 32 blocks

// Records of the fused board pass's per-card scans (prl_plan_kernels.hip builds them, prl_fhp_pass.inc consumes them): lane t of
// a 768-lane workgroup serves entries 3i, 3i+1, 3i+2 (i = t % 16) of the list of the (t / 16)-th live card, cards ascending.
//   word 0: entry 3i | entry 3i+1 << 16        word 1: entry 3i+2 | laneA << 16 | laneB << 20 | card << 24
//   entry : sorted position (11 bits) | HEAD (first of its tie group within this list) | TAIL (last of it) | LOWER (the card is
//           the lower card of that hand) | VALID; entries past the end of a list / slots without a card point at the always-zero
//           position PRL_CLX_ZERO_POS and are their own group
//   laneA : lane (0..15 of the row) holding the head of the tie group that entry 3i belongs to; laneB: lane holding the tail of
//           the group of entry 3i+2
#define PRL_CLX_SLOTS 48
#define PRL_CLX_WORDS (PRL_CLX_SLOTS * 16 * 2)
#define PRL_CLX_ZERO_POS 1087u   // = FHP_NPAD - 1
#define PRL_CLX_HEAD 0x0800u
#define PRL_CLX_TAIL 0x1000u
#define PRL_CLX_LOWER 0x2000u
#define PRL_CLX_VALID 0x4000u

// Position-domain plan of a 5-card board for the fused board pass (round 4: a lane owns two ADJACENT SORTED POSITIONS, and the
// board's action columns are stored in HBM in that order -- prl_fhp.h "sorted storage"). int16 units, one block per board:
//   sh  [PRL_PP_R]     the permutation position -> hand: the live hands in rank order (positions 0 .. n_live-1), then the hands the
//                      board blocks, in hand-index order
//   gm1 [PRL_PP_NPAD]  per live position: tie-group start - 1 = index of P[gs] in the inclusive prefix array (PRL_CLX_ZERO_POS when
//                      gs == 0: that slot always reads 0); positions past the live ones: PRL_CLX_ZERO_POS
//   em1 [PRL_PP_NPAD]  tie-group end - 1
//   cc  [PRL_PP_NPAD]  c1 | c2 << 8 of the hand at the position (0 past the live ones)
#define PRL_PP_R 1326
#define PRL_PP_NPAD 1088
#define PRL_PP_OFF_GM1 PRL_PP_R
#define PRL_PP_OFF_EM1 (PRL_PP_R + PRL_PP_NPAD)
#define PRL_PP_OFF_CC (PRL_PP_R + 2 * PRL_PP_NPAD)
#define PRL_PP_STRIDE 4592   // int16 per board: 1326 + 3 * 1088 = 4590, rounded up to a multiple of 8 (16-byte aligned blocks)

struct PrlDevTree {
    int32_t n_nodes, n_cols, R, n_hole, n_cards, n_suits, rank_rule, n_boards, board_len, n_levels;
    const int32_t *kind, *actor, *parent, *child_idx, *acted_last, *board_id, *main_pot, *n_children, *first_col,
        *child_start, *child_list, *level_nodes;
    const int8_t* boards;   // [n_boards][board_len]
    const int16_t* hole;    // [R][2] 1d cards of every hand (second = -1 for 1-card games)
    float chance_prob;      // generalised StrategyFiller.py:166 constant of the FIRST chance node (what the fused engine's one level uses)
    const float* chance_w;  // [n_nodes] the same per chance node (trees that deal on several streets: fewer cards left, other k); 0 elsewhere
    float eq_const;         // generalised ValueFiller.py:19 constant
    // showdown plans of 2-card games, one per board + one "no board" plan at index n_boards (see prl_plan_kernels.hip)
    int32_t plan_stride;         // = R
    int32_t cl_stride;           // = n_cards * (n_cards - 1)
    const int16_t* plan_sh;      // [n_plans][R]   hand at sorted position (only the first n_live entries are valid)
    const int16_t* plan_pos;     // [n_plans][R]   sorted position of a hand, -1 if blocked by the board
    const int16_t* plan_gs;      // [n_plans][R]   first position of the tie group
    const int16_t* plan_ge;      // [n_plans][R]   one past the last position of the tie group
    const int16_t* plan_cl;      // [n_plans][n_cards][n_cards-1] positions of the hands containing card c, ascending; -1 pad
    const uint8_t* plan_klh;     // [n_plans][R][4] LEVELS engine (nullptr otherwise): per hand and for each of its two cards, how many
                                 // entries of that card's list lie before the hand's tie group / before its end (lo1, hi1, lo2, hi2)
    const int32_t* plan_nlive;   // [n_plans]
    const int32_t* plan_ndealt;  // [n_plans]   board cards of the plan's row (0 for the no-board plan): its card lists hold n_cards - 1 - that many hands
    // hand-domain / flagged copies used by the fused board kernels (prl_fhp_kernels.hip)
    const int16_t* plan_hgs;     // [n_plans][R]   gs[pos[h]] (0 for blocked hands)
    const int16_t* plan_hge;     // [n_plans][R]   ge[pos[h]]
    const uint32_t* plan_clx;    // [n_plans][PRL_CLX_WORDS] per-lane records of the fused board pass (below); 5-card boards only
    const int16_t* plan_pp;      // [n_plans][PRL_PP_STRIDE] position-domain plan of the single-deal fused engine (above); nullptr otherwise
};

struct PrlDevState {
    double* strategy;    // [n_cols][R]
    uint8_t* strat_f64;  // [n_nodes]
    float* reach;        // [n_nodes][2][R]
    float* ev;
    float* ev_br;
    int32_t* br_idx;     // [n_nodes][R] first arg-max child of the actor's best response (nodes.py:39)
    float* regret;       // [n_cols][R]
    float* avg_sum;      // [n_cols][R]  (Vanilla / Linear)
    double* avg;         // [n_cols][R]
    uint8_t* avg_f64;    // [n_nodes]
    float* expl;         // [2] root exploitability of the last EV pass
};

// one entry per independent small-tree solve of a batched launch (prl_solver_iterations_many)
struct PrlSmallJob {
    PrlDevTree T;
    PrlDevState S;
    const int32_t* level_start;
    const int32_t* term_nodes; int32_t n_term;
    const int32_t* nodes_p[2]; int32_t n_nodes_p[2];
    int32_t variant, delay, n_iters, state_in_lds, tree_in_lds, n_cols;
    PrlIterDev* ip;
};

PRL_HD PRL_INLINE size_t prl_vidx(const PrlDevTree& T, int node, int p) { return ((size_t)node * 2 + (size_t)p) * (size_t)T.R; }
PRL_HD PRL_INLINE size_t prl_cidx(const PrlDevTree& T, int col) { return (size_t)col * (size_t)T.R; }

PRL_HD PRL_INLINE bool prl_hand_blocked(const PrlDevTree& T, int h, int board_id) {
    if (board_id < 0) return false;
    const int8_t* b = T.boards + (size_t)board_id * T.board_len;
    int c1 = T.hole[2 * h], c2 = T.hole[2 * h + 1];
    bool blk = false;
    for (int i = 0; i < T.board_len; ++i) blk = blk || (b[i] == c1) || (b[i] == c2);
    return blk;
}
