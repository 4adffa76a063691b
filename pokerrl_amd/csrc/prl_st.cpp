// Host side of the per-street fused engine (prl_st.h): cuts a flat multi-street public tree into trunk + street instances, checks
// every instance against the registered street shapes and lays out the internal action-column order.
#include "prl_st.h"

#include <string>

#include "prl_cards.h"
#include "prl_tree.h"

namespace {

struct Listing {  // DFS pre-order listing of one street instance (chance nodes / showdowns = leaves)
    std::vector<int> node, kind, actor, nch;
};

// leaves: chance nodes (kind 3 in the listing, "the street goes on") or showdown terminals; returns false on a node the engine
// cannot walk (a chance node directly below a chance node = an all-in run-out chain)
bool list_instance(const PrlFlatTree& t, int root, Listing* out, bool* has_chance_leaf, bool* has_showdown_leaf) {
    std::vector<int> stack{root};
    while (!stack.empty()) {
        const int n = stack.back();
        stack.pop_back();
        const int k = t.kind[n];
        out->node.push_back(n);
        if (k == PRL_NODE_CHANCE) {
            *has_chance_leaf = true;
            out->kind.push_back(PRL_NODE_TERM_SHOWDOWN); out->actor.push_back(-1); out->nch.push_back(0);
        } else if (k == PRL_NODE_TERM_SHOWDOWN) {
            *has_showdown_leaf = true;
            out->kind.push_back(k); out->actor.push_back(-1); out->nch.push_back(0);
        } else if (k == PRL_NODE_TERM_FOLD) {
            out->kind.push_back(k); out->actor.push_back(-1); out->nch.push_back(0);
        } else {
            out->kind.push_back(k); out->actor.push_back(t.actor[n]); out->nch.push_back(t.n_children[n]);
            for (int i = t.n_children[n] - 1; i >= 0; --i) stack.push_back(t.child_list[t.child_start[n] + i]);
        }
        if (out->node.size() > PRL_FHP_MAX_NODES) return false;
    }
    return true;
}

int match_spec(const PrlFlatTree& t, const Listing& L) {
    for (int sid = 0; sid < PRL_ST_N_SPECS; ++sid) {
        const PrlFhpShapeDesc& d = prl_st_spec_desc(sid);
        if (d.n_nodes != (int)L.node.size()) continue;
        bool ok = true;
        for (int n = 0; n < d.n_nodes && ok; ++n) {
            if (L.kind[n] != d.kind[n] || L.nch[n] != d.nch[n]) ok = false;
            else if (L.kind[n] == PRL_NODE_DECISION && L.actor[n] != d.actor[n]) ok = false;
            else if (n > 0 && t.parent[L.node[n]] != L.node[d.parent[n]]) ok = false;
            else if (L.kind[n] == PRL_NODE_TERM_FOLD && t.acted_last[L.node[n]] != d.folder[n]) ok = false;
        }
        if (ok) return sid;
    }
    return -1;
}

float chance_weight(const PrlFlatTree& t, int chance_node, long long n_children_override) {
    auto dealt = [&](int row) { int n = 0; if (row >= 0) for (int c = 0; c < t.board_len; ++c) n += t.boards[(size_t)row * t.board_len + c] >= 0; return n; };
    const int before = dealt(t.board_id[chance_node]);
    const int child = t.child_list[t.child_start[chance_node]];
    const int k = dealt(t.board_id[child]) - before;
    const long long nc = n_children_override > 0 ? n_children_override : t.n_children[chance_node];
    const int N = t.rules.n_cards - before, H = t.rules.n_hole_cards;
    const double denom = (double)nc * (double)prl_comb(N - 2 * H, k) / (double)prl_comb(N, k);  // StrategyFiller.py:166 generalised
    return (float)(1.0 / denom);
}

}  // namespace

int prl_st_build(const PrlFlatTree& t, long long top_weight_children, PrlStPlanHost* out, std::string* why) {
    PrlStPlanHost& P = *out;
    P = PrlStPlanHost();
    auto fail = [&](const char* m) { if (why) *why = m; return PRL_ERR_UNSUPPORTED; };
    if (t.rules.n_hole_cards != 2 || t.rules.n_cards != 52 || t.board_len != 5) return fail("2-hole-card games on the 52-card deck with 5-card run-outs only");
    // ---- trunk: the nodes above every chance node -------------------------------------------------------------------------------
    P.trunk_col_of_node.assign(t.n_nodes, -1);
    long long trunk_nodes = 0;
    for (int n = 0; n < t.n_nodes;) {
        ++trunk_nodes;
        if (t.kind[n] == PRL_NODE_CHANCE) {
            P.trunk_leaf_node.push_back(n);
            n += t.subtree_size[n];
            continue;
        }
        if (t.kind[n] == PRL_NODE_DECISION) {
            P.trunk_col_of_node[n] = P.n_trunk_cols;
            for (int a = 0; a < t.n_children[n]; ++a) P.col_dfs.push_back(t.first_col[n] + a);
            P.n_trunk_cols += t.n_children[n];
        } else if (t.kind[n] == PRL_NODE_TERM_SHOWDOWN) {
            return fail("a showdown before the first deal (all-in run-out)");
        }
        ++n;
    }
    if (P.trunk_leaf_node.empty()) return fail("no chance node");
    P.n_top = t.n_children[P.trunk_leaf_node[0]];
    for (int c : P.trunk_leaf_node)
        if (t.n_children[c] != P.n_top) return fail("the trunk's chance nodes have different numbers of outcomes");
    // ---- streets ------------------------------------------------------------------------------------------------------------------
    // the chance nodes the next level's instances hang below, grouped per parent instance (the trunk = one pseudo instance)
    std::vector<std::vector<int>> parent_leaves{P.trunk_leaf_node};
    int next_col = P.n_trunk_cols;
    for (int lv = 0;; ++lv) {
        if (lv >= PRL_ST_MAX_LEVELS) return fail("more dealing streets than PRL_ST_MAX_LEVELS");
        PrlStLevelHost& L = P.level[lv];
        std::vector<std::vector<int>> my_leaves;  // per instance of this level: its chance leaves
        bool any_chance = false, any_show = false;
        for (size_t pi = 0; pi < parent_leaves.size(); ++pi) {
            const std::vector<int>& leaves = parent_leaves[pi];
            const int nl = (int)leaves.size();
            const int nk = t.n_children[leaves[0]];
            for (int c : leaves)
                if (t.n_children[c] != nk) return fail("the chance nodes of one street instance have different numbers of outcomes");
            if (lv > 0) {
                PrlStInst& par = P.level[lv - 1].inst[pi];
                par.n_kids = nk;
                par.kid_base = (int)L.inst.size();
            }
            for (int k = 0; k < nk; ++k)
                for (int j = 0; j < nl; ++j) {
                    const int ch = leaves[j];
                    const int root = t.child_list[t.child_start[ch] + k];
                    if (t.kind[root] != PRL_NODE_DECISION) return fail("a chance outcome that is not followed by a decision (all-in run-out chain)");
                    {   // the street pass is laid out for >= 3 board cards on every street (st_npad: <= 1209 live hands; per-card lists of <= 48
                        // entries): a first deal of one or two cards (custom rules, e.g. 2 + 2 + 1) would overrun both
                        int dealt = 0;
                        const int row = t.board_id[root];
                        if (row >= 0) for (int c = 0; c < t.board_len; ++c) dealt += t.boards[(size_t)row * t.board_len + c] >= 0;
                        if (dealt < 3) return fail("a street with fewer than 3 board cards out (the street pass holds <= 1209 live hands and per-card lists of <= 48 entries)");
                    }
                    if (j > 0 && t.board_id[root] != t.board_id[t.child_list[t.child_start[leaves[0]] + k]]) return fail("chance outcomes differ between the leaves of one instance");
                    Listing ls;
                    bool hc = false, hs = false;
                    if (!list_instance(t, root, &ls, &hc, &hs)) return fail("a street subtree larger than PRL_FHP_MAX_NODES");
                    any_chance |= hc; any_show |= hs;
                    const int spec = match_spec(t, ls);
                    if (spec < 0) return fail("a street subtree that is not one of the registered shapes (prl_st.h)");
                    if (L.spec < 0) L.spec = spec;
                    else if (L.spec != spec) return fail("street instances of one street with different shapes");
                    PrlStInst in = {};
                    in.row = t.board_id[root];
                    in.parent_slot = (int)pi * nl + j;
                    in.w = chance_weight(t, ch, lv == 0 ? top_weight_children : 0);
                    for (size_t n = 0; n < ls.node.size(); ++n) in.pot[n] = (float)t.main_pot[ls.node[n]];
                    L.inst.push_back(in);
                    L.root_node.push_back(root);
                    std::vector<int> lv_leaves;
                    for (size_t n = 0; n < ls.node.size(); ++n)
                        if (t.kind[ls.node[n]] == PRL_NODE_CHANCE) lv_leaves.push_back(ls.node[n]);
                    my_leaves.push_back(lv_leaves);
                    // internal columns of this instance: adjacent, local DFS order
                    for (size_t n = 0; n < ls.node.size(); ++n)
                        if (ls.kind[n] == PRL_NODE_DECISION)
                            for (int a = 0; a < ls.nch[n]; ++a) P.col_dfs.push_back(t.first_col[ls.node[n]] + a);
                }
        }
        if (any_chance && any_show) return fail("a street with both showdowns and further deals below it (all-in run-outs)");
        const PrlFhpShapeDesc& d = prl_st_spec_desc(L.spec);
        L.n_inst = (int)L.inst.size();
        L.n_cols_inst = d.n_cols;
        L.n_nodes_inst = d.n_nodes;
        L.n_leaves = 0;
        for (int n = 0; n < d.n_nodes; ++n) L.n_leaves += d.kind[n] == PRL_NODE_TERM_SHOWDOWN;
        L.last = !any_chance;
        L.col_base = next_col;
        next_col += L.n_inst * L.n_cols_inst;
        P.n_levels = lv + 1;
        if (L.last) break;
        parent_leaves.swap(my_leaves);
    }
    if ((int)P.col_dfs.size() != t.n_cols) return fail("internal error: column count");
    // every node belongs to the trunk (its chance nodes included) or to exactly one instance (its chance leaves included)
    long long inst_nodes = 0;
    for (int lv = 0; lv < P.n_levels; ++lv) inst_nodes += (long long)P.level[lv].n_inst * P.level[lv].n_nodes_inst;
    if (trunk_nodes + inst_nodes != t.n_nodes) return fail("internal error: node count");
    // complete boards on the last street only (showdowns need ranks)
    {
        const PrlStLevelHost& L = P.level[P.n_levels - 1];
        for (const PrlStInst& in : L.inst) {
            int n = 0;
            for (int c = 0; c < t.board_len; ++c) n += t.boards[(size_t)in.row * t.board_len + c] >= 0;
            if (n != t.board_len) return fail("a showdown on an incomplete board");
        }
    }
    return PRL_OK;
}
