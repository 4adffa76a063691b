// Host-side construction of the flat public tree. See prl_tree.h for the contract and the reference anchors.
#include "prl_tree.h"

#include <algorithm>

namespace {

struct Builder {
    PrlFlatTree* t;
    const int8_t* boards;
    int n_boards, board_len;
    int stop_at_round;  // nodes of a round >= this are left unexpanded (PublicTree.py:173: stop_at_street); INT_MAX = full tree
    int err = 0;

    int new_node(int kind, int actor, int parent, int child_idx, int action, int acted_last, int round, int board_id,
                 int main_pot, int depth) {
        int id = t->n_nodes++;
        t->kind.push_back(kind);
        t->actor.push_back(actor);
        t->parent.push_back(parent);
        t->child_idx.push_back(child_idx);
        t->action.push_back(action);
        t->acted_last.push_back(acted_last);
        t->round.push_back(round);
        t->board_id.push_back(board_id);
        t->main_pot.push_back(main_pot);
        t->depth.push_back(depth);
        t->n_children.push_back(0);
        t->first_col.push_back(-1);
        t->subtree_size.push_back(1);
        return id;
    }

    // replicate the node range [s, e) (one board subtree, root s, parent chance node `ch`) for board b as child k
    void replicate(int s, int e, int col_s, int col_e, int ch, int b, int k) {
        int shift = t->n_nodes - s;
        int cshift = t->n_cols - col_s;
        for (int i = s; i < e; ++i) {
            int id = new_node(t->kind[i], t->actor[i], i == s ? ch : t->parent[i] + shift, i == s ? k : t->child_idx[i],
                              t->action[i], t->acted_last[i], t->round[i], b, t->main_pot[i], t->depth[i]);
            t->n_children[id] = t->n_children[i];
            t->first_col[id] = t->first_col[i] < 0 ? -1 : t->first_col[i] + cshift;
            t->subtree_size[id] = t->subtree_size[i];
        }
        for (int c = col_s; c < col_e; ++c) {
            t->col_action.push_back(t->col_action[c]);
            t->col_node.push_back(t->col_node[c] + shift);
        }
        t->n_cols += col_e - col_s;
    }

    void expand(int id, const PrlEnvState& st) {
        if (err) return;
        if (st.round >= stop_at_round) {  // a non-terminal LEAF: it keeps its actor, has no children and no action columns
            t->is_partial = true;
            return;
        }
        int32_t legal[PRL_MAX_BET_SIZES + 2];
        int n = prl_legal_actions(t->game, st, legal);
        if (n <= 0) { err = PRL_ERR_STATE; t->error = "decision node without legal actions"; return; }
        t->n_children[id] = n;
        t->first_col[id] = t->n_cols;
        for (int i = 0; i < n; ++i) { t->col_action.push_back(legal[i]); t->col_node.push_back(id); }
        t->n_cols += n;
        const int actor = st.cur;
        const int depth = t->depth[id];
        for (int i = 0; i < n; ++i) {
            PrlEnvState s2 = st;
            PrlStepInfo info;
            prl_env_step(t->game, s2, legal[i], &info);
            if (info.is_terminal) {
                // state before payouts; round / board stay the parent's (PublicTree.py:244-251)
                int kind = (legal[i] == PRL_FOLD) ? PRL_NODE_TERM_FOLD : PRL_NODE_TERM_SHOWDOWN;
                if (kind == PRL_NODE_TERM_SHOWDOWN && t->board_id[id] < 0 && t->rules.n_hole_cards != 1) {
                    // all-in before the deal: the reference averages the showdown over every run-out (ValueFiller.py:160-175,
                    // 1-card ranges only); for 2-card ranges that needs run-out equity tables, which this engine does not have.
                    // Refuse the tree instead of valuing the terminal at 0.
                    err = PRL_ERR_UNSUPPORTED;
                    t->error = "showdown terminal before the deal (all-in run-out) on a 2-hole-card tree is not supported";
                    return;
                }
                new_node(kind, -1, id, i, legal[i], actor, t->round[id], t->board_id[id], info.pot_before_payout, depth + 1);
            } else if (info.chance_acts) {
                if (t->game.n_rounds != 2 || s2.round != 1) {
                    err = PRL_ERR_UNSUPPORTED;
                    t->error = "public trees with more than one chance level are not supported yet";
                    return;
                }
                int ch = new_node(PRL_NODE_CHANCE, -1, id, i, legal[i], actor, t->round[id], t->board_id[id], st.main_pot, depth + 1);
                t->n_children[ch] = n_boards;
                int s = -1, e = -1, col_s = -1, col_e = -1;
                for (int b = 0; b < n_boards; ++b) {
                    if (b == 0) {
                        s = t->n_nodes;
                        col_s = t->n_cols;
                        int c = new_node(PRL_NODE_DECISION, s2.cur, ch, 0, -1, -2, s2.round, 0, s2.main_pot, depth + 2);
                        expand(c, s2);
                        if (err) return;
                        e = t->n_nodes;
                        col_e = t->n_cols;
                    } else {
                        replicate(s, e, col_s, col_e, ch, b, b);
                    }
                }
                t->subtree_size[ch] = t->n_nodes - ch;
            } else {
                int c = new_node(PRL_NODE_DECISION, s2.cur, id, i, legal[i], actor, s2.round, t->board_id[id], s2.main_pot, depth + 1);
                expand(c, s2);
                if (err) return;
            }
        }
        t->subtree_size[id] = t->n_nodes - id;
    }
};

}  // namespace

int prl_build_flat_tree(const PrlGame& game, const PrlRules& rules, const int8_t* boards, int n_boards, int board_len,
                        PrlFlatTree* out, int stop_at_round) {
    PrlFlatTree& t = *out;
    t = PrlFlatTree();
    t.rules = rules;
    t.game = game;
    if (n_boards <= 0 || board_len <= 0 || board_len > PRL_MAX_BOARD_CARDS) { t.error = "bad board table"; return PRL_ERR_ARG; }
    if (rules.n_hole_cards != 1 && rules.n_hole_cards != 2) { t.error = "n_hole_cards must be 1 or 2"; return PRL_ERR_UNSUPPORTED; }
    t.n_boards = n_boards;
    t.board_len = board_len;
    t.boards.assign(boards, boards + (size_t)n_boards * board_len);

    Builder b{&t, boards, n_boards, board_len, stop_at_round < 0 ? 0x7FFFFFFF : stop_at_round};
    PrlEnvState st;
    prl_env_reset(game, st);
    // the root is the first actor's decision node (PublicTree.py:111-124); its `action` is the reference's "CHANCE"
    int root = b.new_node(PRL_NODE_DECISION, st.cur, -1, 0, -1, -1, st.round, -1, st.main_pot, 0);
    b.expand(root, st);
    if (b.err) return b.err;

    // CSR children + BFS levels
    t.child_start.assign(t.n_nodes + 1, 0);
    for (int i = 0; i < t.n_nodes; ++i) t.child_start[i + 1] = t.child_start[i] + t.n_children[i];
    t.child_list.assign(t.n_nodes > 0 ? t.n_nodes - 1 : 0, -1);
    int max_depth = 0;
    for (int i = 1; i < t.n_nodes; ++i) {
        t.child_list[t.child_start[t.parent[i]] + t.child_idx[i]] = i;
        max_depth = std::max(max_depth, t.depth[i]);
    }
    t.n_levels = max_depth + 1;
    t.level_start.assign(t.n_levels + 1, 0);
    for (int i = 0; i < t.n_nodes; ++i) t.level_start[t.depth[i] + 1]++;
    for (int d = 0; d < t.n_levels; ++d) t.level_start[d + 1] += t.level_start[d];
    t.level_nodes.assign(t.n_nodes, 0);
    std::vector<int32_t> fill(t.level_start.begin(), t.level_start.end() - 1);
    for (int i = 0; i < t.n_nodes; ++i) t.level_nodes[fill[t.depth[i]]++] = i;
    return PRL_OK;
}
