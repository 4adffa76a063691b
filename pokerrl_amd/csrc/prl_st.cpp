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
    // A PARENT is what the instances of a street hang below: the trunk (street 0) or an instance of the previous street, with its chance leaves.
    // Its children occupy the rows kid_base + k * n_leaves + j (outcome k, leaf j) of the street's buffer, whatever group they belong to.
    struct Parent { std::vector<int> leaves; int leaf_slot0; int group, index; };  // group < 0: the trunk
    std::vector<Parent> parents(1);
    parents[0].leaves = P.trunk_leaf_node; parents[0].leaf_slot0 = 0; parents[0].group = -1; parents[0].index = 0;
    std::vector<std::vector<int32_t>> group_cols;  // per group: the flat-tree columns of its instances, in instance order (the internal order is assembled below)
    long long chain_nodes = 0;
    for (int lv = 0;; ++lv) {
        if (lv >= PRL_ST_MAX_LEVELS) return fail("more dealing streets than PRL_ST_MAX_LEVELS");
        int group_of_spec[PRL_ST_N_SPECS];
        for (int& g : group_of_spec) g = -1;
        std::vector<Parent> next;
        bool any_chance = false, any_show = false, any_inst = false;
        for (size_t pi = 0; pi < parents.size(); ++pi) {
            const Parent& par = parents[pi];
            const int nl = (int)par.leaves.size();
            if (nl == 0) continue;
            const int nk = t.n_children[par.leaves[0]];
            for (int c : par.leaves)
                if (t.n_children[c] != nk) return fail("the chance nodes of one street instance have different numbers of outcomes");
            const int kid_base = P.n_val_slots[lv];
            P.n_val_slots[lv] += nk * nl;
            if (par.group >= 0) {
                PrlStInst& pin = P.group[par.group].inst[par.index];
                pin.n_kids = nk;
                pin.kid_base = kid_base;
            }
            for (int k = 0; k < nk; ++k)
                for (int j = 0; j < nl; ++j) {
                    const int ch = par.leaves[j];
                    const int root = t.child_list[t.child_start[ch] + k];
                    const int val_slot = kid_base + k * nl + j;
                    if (j > 0 && t.board_id[root] != t.board_id[t.child_list[t.child_start[par.leaves[0]] + k]]) return fail("chance outcomes differ between the leaves of one instance");
                    if (t.kind[root] != PRL_NODE_DECISION) {
                        // a chance outcome without a decision below it: the all-in call above was the last decision of the hand (a run-out chain)
                        if (t.kind[root] != PRL_NODE_CHANCE && t.kind[root] != PRL_NODE_TERM_SHOWDOWN) return fail("a chance outcome that is neither a decision, a chance node nor a showdown");
                        for (int n = root; n < root + t.subtree_size[root]; ++n)
                            if (t.kind[n] == PRL_NODE_DECISION || t.kind[n] == PRL_NODE_TERM_FOLD) return fail("a decision below a run-out chain");
                        PrlStChainKid ck;
                        ck.node = root; ck.street = lv; ck.parent_slot = par.leaf_slot0 + j; ck.val_slot = val_slot;
                        ck.w = chance_weight(t, ch, lv == 0 ? top_weight_children : 0);
                        P.chain.push_back(ck);
                        chain_nodes += t.subtree_size[root];
                        continue;
                    }
                    {   // the street pass is laid out for >= 3 board cards on every street (st_npad: <= 1209 live hands; per-card lists of <= 48
                        // entries): a first deal of one or two cards (custom rules, e.g. 2 + 2 + 1) would overrun both
                        int dealt = 0;
                        const int row = t.board_id[root];
                        if (row >= 0) for (int c = 0; c < t.board_len; ++c) dealt += t.boards[(size_t)row * t.board_len + c] >= 0;
                        if (dealt < 3) return fail("a street with fewer than 3 board cards out (the street pass holds <= 1209 live hands and per-card lists of <= 48 entries)");
                    }
                    Listing ls;
                    bool hc = false, hs = false;
                    if (!list_instance(t, root, &ls, &hc, &hs)) return fail("a street subtree larger than PRL_FHP_MAX_NODES");
                    any_chance |= hc; any_show |= hs; any_inst = true;
                    const int spec = match_spec(t, ls);
                    if (spec < 0) return fail("a street subtree that is not one of the registered shapes (prl_st.h)");
                    if (group_of_spec[spec] < 0) {
                        if (P.n_groups >= PRL_ST_MAX_GROUPS) return fail("more (street, shape) groups than PRL_ST_MAX_GROUPS");
                        group_of_spec[spec] = P.n_groups++;
                        PrlStLevelHost& G = P.group[group_of_spec[spec]];
                        const PrlFhpShapeDesc& d = prl_st_spec_desc(spec);
                        G.street = lv; G.spec = spec; G.n_cols_inst = d.n_cols; G.n_nodes_inst = d.n_nodes; G.n_leaves = 0;
                        for (int n = 0; n < d.n_nodes; ++n) G.n_leaves += d.kind[n] == PRL_NODE_TERM_SHOWDOWN;
                        group_cols.emplace_back();
                    }
                    const int gi = group_of_spec[spec];
                    PrlStLevelHost& G = P.group[gi];
                    PrlStInst in = {};
                    in.row = t.board_id[root];
                    in.parent_slot = par.leaf_slot0 + j;
                    in.w = chance_weight(t, ch, lv == 0 ? top_weight_children : 0);
                    in.val_slot = val_slot;
                    in.leaf_slot0 = P.n_leaf_slots[lv];
                    for (size_t n = 0; n < ls.node.size(); ++n) in.pot[n] = (float)t.main_pot[ls.node[n]];
                    Parent me;
                    me.leaf_slot0 = in.leaf_slot0; me.group = gi; me.index = (int)G.inst.size();
                    for (size_t n = 0; n < ls.node.size(); ++n)
                        if (t.kind[ls.node[n]] == PRL_NODE_CHANCE) me.leaves.push_back(ls.node[n]);
                    P.n_leaf_slots[lv] += (int)me.leaves.size();
                    G.inst.push_back(in);
                    G.root_node.push_back(root);
                    next.push_back(me);
                    // the columns of this instance: adjacent, local DFS order
                    for (size_t n = 0; n < ls.node.size(); ++n)
                        if (ls.kind[n] == PRL_NODE_DECISION)
                            for (int a = 0; a < ls.nch[n]; ++a) group_cols[gi].push_back(t.first_col[ls.node[n]] + a);
                }
        }
        if (!any_inst) {  // nothing but run-out chains below the previous street: the streets end there
            if (lv == 0) return fail("no decision after the first deal");
            P.n_levels = lv;
            break;
        }
        if (any_chance && any_show) return fail("a street with both showdowns and further deals below it");
        const bool last = !any_chance;
        for (int g = 0; g < PRL_ST_N_SPECS; ++g)
            if (group_of_spec[g] >= 0) { PrlStLevelHost& G = P.group[group_of_spec[g]]; G.n_inst = (int)G.inst.size(); G.last = last; }
        P.n_levels = lv + 1;
        if (last) {
            P.n_leaf_slots[lv] = 0;
            break;
        }
        parents.swap(next);
    }
    for (int g = 0; g + 1 < P.n_groups; ++g)
        if (P.group[g].street > P.group[g + 1].street) return fail("internal error: group order");
    if (!P.group[P.n_groups - 1].last) {
        // the deepest street with decisions is not the last dealing street (every path was all-in by then): its passes take kind-3 leaves for "goes on",
        // which they are (chance nodes with run-out chains below) -- fine as long as no group claims showdowns
        for (int g = 0; g < P.n_groups; ++g) if (P.group[g].last) return fail("internal error: a last street before the deepest one");
    }
    // internal column order: the trunk's columns, then group by group, instance by instance
    {
        int next_col = P.n_trunk_cols;
        for (int g = 0; g < P.n_groups; ++g) {
            P.group[g].col_base = next_col;
            next_col += P.group[g].n_inst * P.group[g].n_cols_inst;
            P.col_dfs.insert(P.col_dfs.end(), group_cols[g].begin(), group_cols[g].end());
        }
    }
    if ((int)P.col_dfs.size() != t.n_cols) return fail("internal error: column count");
    // every node belongs to the trunk (its chance nodes included), to exactly one instance (its chance leaves included) or to a run-out chain
    long long inst_nodes = 0;
    for (int g = 0; g < P.n_groups; ++g) inst_nodes += (long long)P.group[g].n_inst * P.group[g].n_nodes_inst;
    if (trunk_nodes + inst_nodes + chain_nodes != t.n_nodes) return fail("internal error: node count");
    // complete boards on the last street only (showdowns need ranks)
    for (int g = 0; g < P.n_groups; ++g) {
        const PrlStLevelHost& L = P.group[g];
        if (!L.last) continue;
        for (const PrlStInst& in : L.inst) {
            int n = 0;
            for (int c = 0; c < t.board_len; ++c) n += t.boards[(size_t)in.row * t.board_len + c] >= 0;
            if (n != t.board_len) return fail("a showdown on an incomplete board");
        }
    }
    return PRL_OK;
}
