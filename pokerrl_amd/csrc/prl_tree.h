// Flat (struct-of-arrays) public game tree, built once on the host and uploaded to HBM.
//
// Replaces the reference's tree of Python objects (PokerRL/game/_/tree/_/nodes.py:8-62) and its builder
// (PublicTree.py:111-126, :161-293). Same semantics:
//   * children of a decision node are in `allowed_actions` order (test/game/test_tree.py:43,57,71);
//   * a node whose action closed the betting round with cards still to come is a CHANCE node whose children are one
//     decision node per board, in the order of the board list handed in (reference: cards ascending, test_tree.py:58-59);
//   * terminal nodes carry the pot AFTER the bet sweep and BEFORE the payout (PublicTree.py:244-251);
//   * the root is the first actor's decision node.
// Node ids are DFS pre-order, so every board subtree is one contiguous id range and all board subtrees below one chance
// node have the same size and shape (betting never depends on the cards) -- the property the board-block kernels use.
#pragma once
#include <vector>
#include <string>
#include "prl_defs.h"
#include "prl_cards.h"
#include "prl_env.h"

struct PrlFlatTree {
    PrlRules rules;
    PrlGame game;
    int32_t n_nodes = 0;
    int32_t n_cols = 0;       // sum over decision nodes of their number of actions ("action columns")
    int32_t n_boards = 0;     // rows of the board table: one per board PREFIX of every dealing round (cards not dealt yet = -1)
    int32_t n_runouts = 0;    // the caller's rows (complete run-outs); == n_boards when the game deals once
    int32_t board_len = 0;    // cards per board row
    int32_t n_levels = 0;
    // per node
    std::vector<int32_t> kind, actor, parent, child_idx, action, acted_last, round, board_id, main_pot, depth;
    std::vector<int32_t> n_children, first_col, subtree_size, child_start;
    std::vector<int32_t> child_list;          // CSR payload, n_nodes - 1 entries
    // per action column
    std::vector<int32_t> col_action, col_node;
    // board table
    std::vector<int8_t> boards;
    // levels (BFS depth) for the level-synchronous kernels
    std::vector<int32_t> level_start, level_nodes;
    bool is_partial = false;  // built with stop_at_round: decision nodes without children exist (structure only: no solver)
    std::string error;
};

// Builds the public tree. `boards` lists run-outs (board_len cards each, in deal order); a game that deals on several streets
// (LimitHoldem: 3 + 1 + 1) gets one chance level per street whose children are the distinct prefixes of the listed run-outs.
// An all-in before the last street of a 2-hole-card game becomes a chain of chance nodes down to showdown leaves.
// Returns 0 or PRL_ERR_*.
// stop_at_round >= 0: nodes whose betting round is >= stop_at_round are not expanded (PublicTree's stop_at_street,
// PublicTree.py:72,173,185); such a partial tree carries structure and states only -- the solver refuses it.
int prl_build_flat_tree(const PrlGame& game, const PrlRules& rules, const int8_t* boards, int n_boards, int board_len,
                        PrlFlatTree* out, int stop_at_round = -1);
