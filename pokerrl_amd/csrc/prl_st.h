// Per-street fused engine ("STREETS") for public trees that deal on several streets (LimitHoldem: 3 + 1 + 1 board cards;
// reference anchors: PokerRL/game/_/tree/PublicTree.py:188-210 one chance level per dealing round, games.py:134-167,
// StrategyFiller.py:148-169 per-chance-node weights, ValueFiller.py:76-78 chance sums).
//
// The flat tree of such a game is a TRUNK (the betting before the first deal) whose chance nodes each fan out into one STREET
// INSTANCE per chance outcome; a street instance is the betting subtree of one street on one board prefix, entered through one
// particular leaf of its parent -- its own "go on" leaves are chance nodes again (or showdowns on the last street). Betting never
// looks at the cards, so all instances of a street share ONE shape (a spec of prl_fhp.h, leaves = kind 3). The engine walks every
// instance on chip like the board pass of the single-deal engine (prl_fhp_pass.inc) and never materialises per-node vectors:
//
//   trunk (LEVELS kernels, chance nodes as leaves)
//     DOWN  street 1 .. L-1   hand-local: both seats' reach at every "go on" leaf of every instance -> leaf_reach (HBM)
//     PASS  street L (last)   reach at the terminals, showdown / fold equity scans, values (+ best response), regret / average
//                             update of the seat being updated; one row of root vectors per instance -> val (HBM)
//     PASS  street L-1 .. 1   the same with fold terminals only; a "go on" leaf's value is the canonical chance sum (blocks of 32
//                             children) of its child instances' root vectors
//   trunk: chance-leaf values = canonical sum over the street-1 instances of that leaf (the one step a sharded solve exchanges)
//
// HBM per instance: its action columns (regret f32, average f64 [+ sum f32]) + 2 R floats per leaf + <= 4 R floats of root
// vectors: ~1/4 of what the level-synchronous engine keeps ([n_nodes][2][R] reach / ev / ev_br + float64 strategies).
//
// Instance order inside a street: (parent instance, chance outcome k, parent leaf j), j fastest -- the instances of one board
// prefix are adjacent (one plan, read once), the children of (parent, leaf j) sit at stride n_leaves(parent shape), and the street-1
// instances are outcome-major, i.e. one row per flop: what a sharded solve splits over ranks.
// Action columns are kept in an INTERNAL order: trunk columns, then street by street, instance by instance, each instance's columns
// adjacent in its local DFS order; prl_solver_get / set translate to the flat tree's DFS column order (col_dfs).
#pragma once
#include <vector>

#include "prl_fhp.h"

// Round 6: MIXED STREETS. The instances of one street need not share a shape any more (a discretized no-limit game: the betting left before somebody is
// all-in depends on the pot the street is entered with), and a leaf may be an all-in call whose children hold no decision at all (a run-out chain: chance
// nodes down to showdowns, PublicTree.py:244-251 / ValueFiller.py:160-175). The instances of a street are cut into GROUPS, one per registered shape, each
// walked by its own launches of the same kernels; every street has ONE leaf-reach buffer and ONE buffer of root-vector rows, addressed by explicit slots
// (PrlStInst::leaf_slot0 / val_slot), so that the children of one leaf may live in any group: child (leaf j, outcome k) of an instance sits at row
// kid_base + k * n_leaves + j of the next street's buffer whatever it is. The decision-free subtrees (the chance outcomes below an all-in call) form a small
// forest with two kernels of its own (prl_launch_st_chain_eval below): its showdowns read the reach of the all-in leaf in place and write their values into
// the rows the parent's pass sums.
enum { PRL_ST_SPEC_9 = 0, PRL_ST_SPEC_15 = 1, PRL_ST_SPEC_21 = 2, PRL_ST_SPEC_27 = 3, PRL_ST_SPEC_33 = 4, PRL_ST_N_SPECS = 5 };
#define PRL_ST_MAX_LEVELS 4   // dealing streets
#define PRL_ST_MAX_GROUPS 20  // (street, shape) groups: PRL_ST_MAX_LEVELS x PRL_ST_N_SPECS

// per street instance (uniform over the workgroup that walks it: read through scalar loads)
struct PrlStInst {
    int32_t row;          // row of the board table = showdown plan of this instance
    int32_t parent_slot;  // slot of its root's reach in the previous street's leaf-reach buffer (street 0: the trunk id of its chance node)
    float w;              // chance weight of the outcome (StrategyFiller.py:159-166 generalised, per chance node)
    int32_t n_kids;       // chance outcomes below each of its leaves (0 on the last street)
    int32_t kid_base;     // first child row (next street's buffer): child of (leaf j, outcome k) = kid_base + k * n_leaves + j
    int32_t val_slot;     // this instance's row of root vectors in its street's buffer (= where its parent looks for it)
    int32_t leaf_slot0;   // its first leaf's slot in its street's leaf-reach buffer (leaf j: leaf_slot0 + j)
    int32_t pad;
    float pot[PRL_FHP_MAX_NODES];  // main pot of every node of the instance (terminals use theirs)
};

// runtime parameters of one street's kernels (by value)
struct PrlStParams {
    int32_t n_inst, R;
    int32_t col_base;            // internal column of instance 0, local column 0; instance i: col_base + i * N_COLS
    int32_t variant, iter;
    int32_t max_grid;
    float eq_const;
    const PrlStInst* inst;       // [n_inst]
    const float* parent_reach;   // [parent slots][2][R] reach at the chance node above every instance root (previous level / trunk gather)
    float* leaf_reach;           // [n_inst * NL][2][R]   DOWN output (not on the last street)
    const float* child_val;      // [n_child_inst][child_w][R]  root vectors of the next street's instances (PASS input, not on the last street)
    int32_t child_w;             // vectors per child row (prl_fhp_out_width(mode))
    float* val;                  // [n_inst][prl_fhp_out_width(mode)][R] PASS output
    float* regret;               // [n_cols][R] internal column order (PRL_SRC_STRAT32: the float32 strategy array, read only)
    double* avg;
    float* avg32;                // opt-in PRL_SOLVER_AVG_F32 (as PrlFhpParams::avg32): the running average of the street columns STORED as float32; `avg` unused then
    float* avg_sum;
    const double* strat_arr;     // PRL_SRC_ARR64 / ARR32
    int32_t avg_mode;            // as PrlFhpParams
    double m_old, m_new;
    int32_t avgsum_mask, avgsum_iter[2];
    const uint16_t* hole_packed;
    int32_t plan_stride, cl_stride, n_cards;
    const int16_t *plan_pos, *plan_hgs, *plan_hge, *plan_cl;
    const uint32_t* plan_clx;
    const int32_t *plan_nlive, *plan_ndealt;
    unsigned long long* timing;  // PRL_ST_TIMING builds (scripts/gpu_st_phases.sh): [8] shader-clock accumulators per phase of the pass; else unused
};

// host description of one GROUP: the instances of one street that have one shape (prl_st_build)
struct PrlStLevelHost {
    int street = 0;                   // 0 = the first dealing street
    int spec = -1, n_inst = 0, n_leaves = 0, n_cols_inst = 0, n_nodes_inst = 0;
    bool last = false;                // the last street: its leaves are showdowns
    int col_base = 0;                 // internal column of instance 0
    std::vector<PrlStInst> inst;
    std::vector<int32_t> root_node;   // flat-tree node id of every instance root
};
// a chance outcome below an all-in call: the root of a decision-free subtree (run-out chain), a child row of its parent like any instance
struct PrlStChainKid {
    int32_t node;         // flat-tree node id (a chance node, or a showdown on the last street)
    int32_t street;       // the street it belongs to (its row lives in that street's buffer)
    int32_t parent_slot;  // leaf slot of the all-in call above it in the previous street's leaf-reach buffer (street 0: index of the trunk's chance leaf)
    int32_t val_slot;     // its row in its street's buffer
    float w;              // chance weight of the outcome
};
struct PrlStPlanHost {
    int n_levels = 0;                       // dealing streets (>= 2 for this engine)
    int n_groups = 0;                       // (street, shape) groups, ordered by street
    PrlStLevelHost group[PRL_ST_MAX_GROUPS];
    int n_val_slots[PRL_ST_MAX_LEVELS] = {0, 0, 0, 0};   // rows of root vectors per street (instances and run-out chain roots)
    int n_leaf_slots[PRL_ST_MAX_LEVELS] = {0, 0, 0, 0};  // leaves per street (not the last one)
    std::vector<PrlStChainKid> chain;       // run-out chain roots, all streets
    std::vector<int32_t> trunk_leaf_node;   // flat-tree ids of the trunk's chance nodes, DFS order (= leaf index j of "level 0")
    int n_top = 0;                          // chance outcomes of the first deal (the unit a sharded solve splits)
    int n_trunk_cols = 0;
    std::vector<int32_t> col_dfs;           // internal column -> flat-tree (DFS) column
    std::vector<int32_t> trunk_col_of_node; // flat-tree node -> internal first column (trunk decision nodes), -1 elsewhere
};

const PrlFhpShapeDesc& prl_st_spec_desc(int spec);
// Does the tree have >= 2 dealing streets whose instances all match registered specs? Fills `out`; returns PRL_OK or PRL_ERR_UNSUPPORTED
// (with the reason in *why). top_weight_children: the number of first-deal outcomes the street-1 chance weight counts (the global
// number in a sharded solve, else 0 = this tree's own).
int prl_st_build(const PrlFlatTree& t, long long top_weight_children, PrlStPlanHost* out, std::string* why);

int prl_launch_st_down(int spec, const PrlStParams& prm, int src0, int src1, void* stream);
int prl_launch_st_pass(int spec, bool last, const PrlStParams& prm, int mode, int src0, int src1, void* stream);
// trunk glue: reach of the trunk's chance leaves -> [n_leaves][2][R]; summed street-1 rows -> the trunk's leaf nodes
struct PrlStScatter {  // copies out of a summed row of n_vec vectors: vector src_vec[d] -> array dst_arr[d] (0 ev, 1 ev_br, 2 half buffer), seat / slot dst_seat[d]
    int32_t n_vec, n_dst;
    int32_t src_vec[6], dst_arr[6], dst_seat[6];
};
void prl_launch_st_scatter_trunk(const float* d_summed, const int32_t* d_leaf_nodes, int n_leaves, int R, const PrlStScatter& sc, float* d_ev, float* d_ev_br,
                                 float* d_half, void* stream);
void prl_launch_st_half_to_trunk(const float* d_half, const int32_t* d_leaf_nodes, int n_leaves, int R, float* d_ev, float* d_ev_br, void* stream);
// run-out chains: the decision-free forest below the all-in calls, evaluated by two kernels of its own (prl_tree_kernels.hip: they share the showdown
// arithmetic of the level kernels). A showdown of the forest reads the opponent's reach straight from the leaf of the all-in call above its chain (the
// trunk's reach array or a street's leaf-reach buffer), multiplies the outcome weights of the chain in (the products the level kernels' reach walk would
// make, in its order) and writes its value where it is read next: the street's row buffer if the showdown is itself a chain root (an all-in call on the
// turn: most of them), the forest's value array otherwise; the chance nodes of the forest then sum their children level by level (canonical chance sum).
struct PrlStChainDev {             // per chain root (device arrays), and per forest node the root it is (or -1)
    const int32_t *street, *val_slot, *node_kid;
};
struct PrlStChainTerm {            // per showdown of the forest
    int32_t node;                  // forest node (board row, pot)
    int32_t src_street, src_slot;  // where the reach above its chain lives: PrlStChainIo::src[src_street] + src_slot * 2 R
    int32_t kid;                   // the chain root it is, or -1
    int32_t n_w;                   // weights of the chain above it: the root's outcome weight, then one per chance node on the way down
    float w[3];
};
struct PrlStChainBundle { int32_t plan, first, count; };  // showdowns [first, first + count) of the list (sorted by board) sit on board row `plan`: one workgroup's work
struct PrlStChainIo { const float* src[PRL_ST_MAX_LEVELS + 1]; float* val[PRL_ST_MAX_LEVELS + 1]; };
struct PrlStRowMap { int32_t width, seat[4]; };  // vector v of a row holds seat[v]'s value (value = best response where nothing is decided)
void prl_launch_st_chain_eval(const PrlDevTree& Tc, const PrlStChainTerm* d_terms, const PrlStChainBundle* d_bundles, int n_bundles, const PrlStChainIo& io,
                              const PrlStChainDev& cd, float* ev_c, const int32_t* h_level_start, int mode, void* stream);
// materialise strategies / averages of one street's columns (prl_solver_get, Vanilla / Linear averages)
void prl_launch_st_strategy_from_regret(const PrlStParams& prm, int spec, double* out_cols, void* stream);
void prl_launch_st_avg_from_sum(const PrlStParams& prm, int spec, void* stream);
