// Host-side construction of the flat public tree. See prl_tree.h for the contract and the reference anchors.
#include "prl_tree.h"

#include <algorithm>
#include <map>

namespace {

struct Builder {
    PrlFlatTree* t;
    const int8_t* boards;
    int n_boards, board_len;
    int stop_at_round;  // nodes of a round >= this are left unexpanded (PublicTree.py:173: stop_at_street); INT_MAX = full tree
    int err = 0;
    // chance outcomes: the caller's run-outs cut into per-street prefixes (prl_tree.h). kids[(round, parent row)] = the rows of the
    // distinct extensions dealt in the transition to `round`, in order of first appearance
    std::map<std::pair<int, int>, std::vector<int>> kids;
    int n_deal_levels = 0;
    bool is_last_round(int r) const { return r == t->game.n_rounds - 1; }

    // an all-in before the last street on a 2-hole-card tree: the reference pays the run-out average (ValueFiller.py:160-175, 1-card
    // only). Here the remaining streets are dealt as a chain of chance nodes without decisions, showdown leaves at the end: the same
    // expectation under the same chance weights a checked-down hand would see, and no special equity kernel.
    void expand_runout(int parent, int child_idx, int action, int acted_last, int round_now, int row, int pot, int depth) {
        if (err) return;
        if (is_last_round(round_now)) {
            new_node(PRL_NODE_TERM_SHOWDOWN, -1, parent, child_idx, action, acted_last, round_now, row, pot, depth);
            return;
        }
        const int ch = new_node(PRL_NODE_CHANCE, -1, parent, child_idx, action, acted_last, round_now, row, pot, depth);
        auto it = kids.find({round_now + 1, row});
        if (it == kids.end() || it->second.empty()) { err = PRL_ERR_ARG; t->error = "no run-out listed below a board prefix"; return; }
        t->n_children[ch] = (int)it->second.size();
        for (size_t k = 0; k < it->second.size(); ++k) expand_runout(ch, (int)k, -1, -2, round_now + 1, it->second[k], pot, depth + 1);
        t->subtree_size[ch] = t->n_nodes - ch;
    }

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
                if (kind == PRL_NODE_TERM_SHOWDOWN && !is_last_round(t->round[id]) && t->rules.n_hole_cards != 1) {
                    expand_runout(id, i, legal[i], actor, t->round[id], t->board_id[id], info.pot_before_payout, depth + 1);
                    if (err) return;
                    continue;
                }
                new_node(kind, -1, id, i, legal[i], actor, t->round[id], t->board_id[id], info.pot_before_payout, depth + 1);
            } else if (info.chance_acts) {
                auto it = kids.find({(int)s2.round, t->board_id[id]});
                if (it == kids.end() || it->second.empty()) { err = PRL_ERR_ARG; t->error = "no run-out listed below a board prefix"; return; }
                const std::vector<int>& rows = it->second;
                int ch = new_node(PRL_NODE_CHANCE, -1, id, i, legal[i], actor, t->round[id], t->board_id[id], st.main_pot, depth + 1);
                t->n_children[ch] = (int)rows.size();
                int s = -1, e = -1, col_s = -1, col_e = -1;
                for (size_t b = 0; b < rows.size(); ++b) {
                    if (b == 0 || n_deal_levels > 1) {  // with deeper chance levels the subtree depends on the row: expand every one
                        if (b == 0) { s = t->n_nodes; col_s = t->n_cols; }
                        int c = new_node(PRL_NODE_DECISION, s2.cur, ch, (int)b, -1, -2, s2.round, rows[b], s2.main_pot, depth + 2);
                        expand(c, s2);
                        if (err) return;
                        if (b == 0) { e = t->n_nodes; col_e = t->n_cols; }
                    } else {
                        replicate(s, e, col_s, col_e, ch, rows[b], (int)b);  // betting never looks at the cards: copy the first subtree
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
    // The caller lists RUN-OUTS: one row of board_len cards per run-out, in deal order. The chance outcomes of the transition to
    // round r are the distinct prefixes of length (cards out after r) below the current prefix. Every prefix of every dealing round
    // gets a row of the board table (cards not dealt yet = -1); with one dealing round the rows are the caller's rows, in order.
    for (int i = 0; i < n_boards; ++i) {  // every run-out: cards of this deck, no card twice (the kernels count on C(n - len, 2) live hands)
        unsigned long long seen = 0;
        for (int c = 0; c < board_len; ++c) {
            const int card = boards[(size_t)i * board_len + c];
            if (card < 0 || card >= rules.n_cards || card >= 64 || (seen >> card & 1ull)) { t.error = "a run-out holds a card outside the deck, or the same card twice"; return PRL_ERR_ARG; }
            seen |= 1ull << card;
        }
    }
    t.board_len = board_len;
    t.n_runouts = n_boards;
    Builder b{&t, boards, n_boards, board_len, stop_at_round < 0 ? 0x7FFFFFFF : stop_at_round};
    {
        int out = 0, prev_len = 0;
        std::map<std::vector<int8_t>, int> row_of;  // prefix -> row
        std::vector<int> parent_row(n_boards, -1);  // per run-out: the row of its prefix at the previous dealing round
        for (int r = 1; r < game.n_rounds; ++r) {
            const int k = rules.board_cards_in_round[r];
            if (k <= 0) continue;
            out += k;
            if (out > board_len) { t.error = "run-outs are shorter than the cards the game deals"; return PRL_ERR_ARG; }
            b.n_deal_levels++;
            for (int i = 0; i < n_boards; ++i) {
                std::vector<int8_t> key(boards + (size_t)i * board_len, boards + (size_t)i * board_len + out);
                key.push_back((int8_t)r);
                auto it = row_of.find(key);
                int row;
                if (it == row_of.end()) {
                    row = (int)(t.boards.size() / board_len);
                    row_of.emplace(key, row);
                    for (int c = 0; c < board_len; ++c) t.boards.push_back(c < out ? boards[(size_t)i * board_len + c] : (int8_t)-1);
                    b.kids[{r, parent_row[i]}].push_back(row);
                } else {
                    row = it->second;
                }
                parent_row[i] = row;
            }
            prev_len = out;
        }
        (void)prev_len;
        if (out != board_len) { t.error = "run-outs are longer than the cards the game deals"; return PRL_ERR_ARG; }
        t.n_boards = (int)(t.boards.size() / board_len);
        if (b.n_deal_levels == 1 && t.n_boards != n_boards) { t.error = "duplicate boards"; return PRL_ERR_ARG; }
    }
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
